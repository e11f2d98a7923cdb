"""Index ingest at a measured rate (bench.py's `ingest` leg): write a snapshot of a synthetic index in the
device layout, drop the index, load the file back, verify sampled rows against the oracle's generator.  One JSON line.

    python scripts/ingest_bench.py [--gb 32] [--cols 100000] [--dir /dev/shm] [--threads 0]

What bounds it: the file system (here page cache / tmpfs: memcpy-speed preads by `threads` host threads) and PCIe (pinned
buffers, hipMemcpyAsync); both are reported next to the rate: `file_GBps` = bytes / time inside pread or pwrite, `pcie_GBps` = a
plain pinned-to-device copy of one 256 MB buffer on this box."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20260928


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gb", type=float, default=32.0, help="size of the matrix to write and load back")
    p.add_argument("--cols", type=int, default=100_000)
    p.add_argument("--hashes", type=int, default=4)
    p.add_argument("--dir", default=None, help="where the snapshot goes (default: /dev/shm if it has room, else the system temp dir)")
    p.add_argument("--threads", type=int, default=0)
    p.add_argument("--sample-rows", type=int, default=64)
    p.add_argument("--group-gb", type=float, default=24.0, help="size of the two-shard group whose snapshot is also written and loaded (0 = skip)")
    a = p.parse_args()
    import tempfile
    from bigsi_amd import _lib
    from bigsi_amd.storage import get_storage
    from bigsi_amd.storage.hip_hbm import HipHbmStorage
    from oracle.ref_model import SynthOracle

    words = -(-a.cols // 64)
    stride = max(16, -(-words // 16) * 16) * 8                       # the library's row pitch: 128-byte multiples
    m = int(a.gb * 1e9 // stride)
    need = m * stride
    d = a.dir
    if d is None:
        d = tempfile.gettempdir()
        for cand in ("/dev/shm", tempfile.gettempdir()):
            try:
                sv = os.statvfs(cand)
                if sv.f_bavail * sv.f_frsize > need * 1.25 + (8 << 30):
                    d = cand
                    break
            except OSError:
                pass
    sv = os.statvfs(d)
    free = sv.f_bavail * sv.f_frsize
    if free < need * 1.1 + (2 << 30):
        m = int(max((free - (2 << 30)) / 1.1, 1 << 30) // stride)      # never fill the file system of a shared box
        need = m * stride
    fn = os.path.join(d, "bigsi_ingest_bench_%d.hbm" % os.getpid())
    cfg = {"storage-engine": "hip-hbm", "k": 31, "m": m, "h": a.hashes, "storage-config": {"name": "ingest", "max_cols": a.cols}}
    st = get_storage(cfg)
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", a.cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", a.hashes)):
        st.set_integer(key, v)
    st.fill_synthetic(SEED, 0, 2)
    out = {"what": "snapshot (device layout) of a synthetic %d-row x %d-sample index written, index dropped, file loaded back" % (m, a.cols),
           "gb": need / 1e9, "dir": d, "fs_free_gb_before": free / 1e9}
    try:
        t0 = time.perf_counter()
        s_ = st.save_snapshot(fn, a.threads)
        out.update(save_s=time.perf_counter() - t0, save_GBps=s_.bytes / s_.seconds / 1e9, save_file_GBps=s_.bytes / max(s_.file_seconds, 1e-9) / 1e9, threads=int(s_.threads))
        st.delete_all()
        t0 = time.perf_counter()
        st2, l_ = HipHbmStorage.load_snapshot(cfg["storage-config"], fn, a.threads)
        out.update(load_s=time.perf_counter() - t0, load_GBps=l_.bytes / l_.seconds / 1e9, load_file_GBps=l_.bytes / max(l_.file_seconds, 1e-9) / 1e9,
                   direct=int(l_.direct))
        # a second load: the file is certainly in the page cache now
        st2.delete_all()
        t0 = time.perf_counter()
        st2, l2 = HipHbmStorage.load_snapshot(cfg["storage-config"], fn, a.threads)
        out.update(load2_GBps=l2.bytes / l2.seconds / 1e9, load2_file_GBps=l2.bytes / max(l2.file_seconds, 1e-9) / 1e9)
        # verification: sampled rows of the loaded index == the generator's (and the four integers came back)
        orc = SynthOracle(SEED, 0, m, a.cols, a.hashes, 31, 2)
        rng = np.random.default_rng(5)
        rows = np.unique(np.concatenate([[0, m - 1], rng.integers(0, m, size=a.sample_rows)])).astype(np.uint64)
        got = st2.get_rows_packed(rows)
        want = np.stack([orc.row(int(r)) for r in rows])
        assert np.array_equal(np.asarray(got), want), "loaded rows differ from the generator"
        assert st2.get_integer("number_of_cols") == a.cols and st2.get_integer("number_of_rows") == m
        out["verified"] = "%d sampled rows of the loaded index == the oracle's generator" % rows.size
        # the PCIe bound of this box: one pinned 256 MB buffer, host -> device, plain hipMemcpy through torch
        try:
            import torch
            h = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
            dv = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
            dv.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                dv.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
            out["pcie_h2d_GBps"] = 8 * (256 << 20) / (time.perf_counter() - t0) / 1e9
        except Exception as e:  # noqa: BLE001
            out["pcie_h2d_GBps"] = None
            out["pcie_note"] = str(e)[:80]
        # why the snapshot's matrix is a directory of 16 striped part files: the same save into ONE file (every pwrite of a file takes
        # its inode lock, every new page goes into its one page-cache tree), a quarter of the rows
        one = fn + ".one"
        s1 = _lib.IoStats()
        n1 = max(m // 4, 1)
        _lib.check(_lib.lib().bigsi_hip_save_rows_file(st2.handle, one.encode(), 0, 0, n1, stride, a.threads, _lib.C.byref(s1)))
        out["save_one_file_GBps"] = s1.bytes / s1.seconds / 1e9
        os.remove(one)
        st2.delete_all()
        # the same for an index spread over several GPUs of this process (storage-config devices=[...]; here two shards on this
        # one device): whole rows, one 2-D copy per shard and 256 MB chunk (bigsi_hip_group_save / load_rows_file)
        if a.group_gb > 0:
            mg = int(a.group_gb * 1e9 // stride)
            gcfg = {"storage-engine": "hip-hbm", "k": 31, "m": mg, "h": a.hashes, "storage-config": {"name": "ingest-group", "max_cols": a.cols, "devices": [0, 0]}}
            sg = get_storage(gcfg)
            sg.delete_all()
            for key, v in (("number_of_rows", mg), ("number_of_cols", a.cols), ("ksi:bloomfilter_size", mg), ("ksi:num_hashes", a.hashes)):
                sg.set_integer(key, v)
            sg.fill_synthetic(SEED, 0, 2)
            sg.res.written[:] = True
            ids = np.arange(0, mg, max(mg // 50, 1), dtype=np.uint64)
            before = np.asarray(sg.get_rows_packed(ids)).copy()
            gs = sg.save_snapshot(fn + ".grp", a.threads)
            sg.delete_all()
            sg2, gl = HipHbmStorage.load_snapshot(gcfg["storage-config"], fn + ".grp", a.threads)
            assert np.array_equal(np.asarray(sg2.get_rows_packed(ids)), before), "group snapshot round trip differs"
            out.update(group_gb=gs.bytes / 1e9, group_save_GBps=gs.bytes / gs.seconds / 1e9, group_load_GBps=gl.bytes / gl.seconds / 1e9,
                       group_load_file_GBps=gl.bytes / max(gl.file_seconds, 1e-9) / 1e9, group_shards=2)
            sg2.delete_all()
    finally:
        import shutil
        for f in (fn, fn + ".tmp", fn + ".one", fn + ".grp", fn + ".grp.tmp"):
            if os.path.exists(f):
                os.remove(f)
            shutil.rmtree(f + ".d", ignore_errors=True)
            shutil.rmtree(f + ".1.d", ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
