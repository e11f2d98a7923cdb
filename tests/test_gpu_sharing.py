"""One resident matrix, several handles (include/bigsi_hip.h "SHARING"): another process attaches over hipIpc
(bigsi_hip_export_ipc / bigsi_hip_open_ipc, storage-config `export` / `attach`) and other threads of the owner take views
(bigsi_hip_open_view); a handle used by two threads at once fails with BIGSI_ERR_STATE instead of racing.  The reference's
counterpart is a store any process opens (bigsi/__main__.py:75-80, 204-205; bigsi/storage/berkeleydb.py:12-19).
Needs a real MI355X: `pytest -m gpu`."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np
import pytest

from conftest import ROOT, check_search, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def owner():
    """A second process holding golden G7's index and a 17.6 GB synthetic one, both exported for attach."""
    d = tempfile.mkdtemp(prefix="bigsi_ipc_")
    rows, cols = 1_400_000, 100_000              # 12.5 KB rows at a 12544-byte pitch: 17.6 GB
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "ipc_owner.py"), d, str(rows), str(cols)],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    line = p.stdout.readline()
    assert line, "owner process died: exit code %r" % p.poll()
    info = json.loads(line)

    def ask(cmd):
        p.stdin.write(cmd + "\n")
        p.stdin.flush()
        return json.loads(p.stdout.readline())
    yield {"dir": d, "rows": rows, "cols": cols, "ask": ask, "proc": p, **info}
    try:
        p.stdin.write("quit\n")
        p.stdin.flush()
        p.wait(timeout=60)
    except Exception:  # noqa: BLE001
        p.kill()


def free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def test_attach_to_another_processes_index(owner):
    """Process B (this one) attaches to A's 17.6 GB index in milliseconds, without a second copy in HBM, answers like the oracle,
    sees what A writes afterwards, survives A's sync(), cannot write, and detaches without harming A."""
    import bigsi_amd
    from bigsi_amd import _lib
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    assert _lib.device_count() >= 1
    before = free_bytes()
    t0 = time.perf_counter()
    st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": owner["rows"], "h": 3,
                      "storage-config": {"name": "big-attached", "attach": os.path.join(owner["dir"], "big.attach")}})
    attach_s = time.perf_counter() - t0
    assert st.res.attached and int(st.res.info().index_bytes) == owner["index_bytes"] > 16e9
    assert attach_s < 0.5, "attach took %.3f s" % attach_s       # (2-4 ms measured; the first IPC open of a cold box has taken longer)
    assert before - free_bytes() < 1 << 30, "attaching took %d bytes of HBM: a second copy?" % (before - free_bytes())
    seqs = owner["seqs"]
    orc = SynthOracle(4242, 0, owner["rows"], owner["cols"], 3, 31, 2)
    orc.insert_kmers(12345, seqs[0])

    def check(threshold):
        batch = st.new_batch(seqs, 31)
        batch.run(threshold)
        _, nu, mk = batch.unique()
        off, colours, counts = batch.hits()
        for i, s in enumerate(seqs):
            u, cnt = orc.counts(s)
            want = np.flatnonzero(cnt >= (u if threshold == 1.0 else mk[i]))
            assert nu[i] == u and np.array_equal(colours[int(off[i]):int(off[i + 1])], want), (threshold, i)
            assert np.array_equal(counts[int(off[i]):int(off[i + 1])], cnt[want].astype(np.uint32))
        batch.close()
        return colours[int(off[0]):int(off[1])].tolist(), colours[int(off[1]):int(off[2])].tolist()
    h0, h1 = check(1.0)
    assert 12345 in h0 and 777 not in h1
    check(0.4)
    # read-only: the storage contract's writes and the device-side builders are refused, nothing is corrupted
    for write in (lambda: st.insert_kmers(5, [seqs[2]], 31), lambda: st.fill_synthetic(1, 0, 2),
                  lambda: st.set_rows_packed(0, np.zeros((1, (owner["cols"] + 7) // 8), np.uint8))):
        with pytest.raises(_lib.BigsiHipError) as ei:
            write()
        assert ei.value.code == _lib.ERR_STATE
    # what the owner writes next is what this handle reads next; its sync() does not disturb the attachment
    assert owner["ask"]("plant") == {"ok": "plant"}
    orc.insert_kmers(777, seqs[1])
    h0, h1 = check(1.0)
    assert 777 in h1
    assert owner["ask"]("sync") == {"ok": "sync"}
    check(1.0)
    st.delete_all()                   # detaches; the owner's index and files stay
    assert os.path.exists(os.path.join(owner["dir"], "big.attach")) and owner["proc"].poll() is None
    st2 = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": owner["rows"], "h": 3,
                       "storage-config": {"name": "big-attached-again", "attach": os.path.join(owner["dir"], "big.attach")}})
    assert st2.res.attached
    st2.delete_all()


def test_attached_bigsi_answers_golden_g7(owner):
    """The whole BIGSI object over an attached index: golden G7's searches (scored ones included) and lookups, sample names from
    the attach file's records."""
    import bigsi_amd
    g = load_golden("g7_random.json")
    k = g["k"]
    b = bigsi_amd.BIGSI({"storage-engine": "hip-hbm", "k": k, "m": g["m"], "h": g["h"],
                         "storage-config": {"name": "g7-attached", "attach": os.path.join(owner["dir"], "g7.attach")}})
    assert b.storage.res.attached and b.num_samples == len(g["sample_names"])
    for s in g["searches"]:
        check_search(lambda: b.search(g["queries"][s["q"]], s["threshold"], s["score"]), s, "attached q%d t=%r" % (s["q"], s["threshold"]))
    for rec in g["lookups"]:
        got = b.lookup([rec["seq"][i:i + k] for i in range(len(rec["seq"]) - k + 1)], remove_trailing_zeros=False)
        assert {km: v.tobytes().hex() for km, v in got.items()} == rec["lookup"]
    b.storage.delete_all()


def test_attached_device_group_answers_golden_g7(owner):
    """A multi-GPU index (here three shards of the owner process on one device) is exported shard by shard; this process attaches
    to all of them (bigsi_hip_group_open_ipc) and the BIGSI object on top answers golden G7 -- exchange between the shards, scores
    on the owning shard and all -- without a copy of its own; writes are refused."""
    import bigsi_amd
    from bigsi_amd import _lib
    g = load_golden("g7_random.json")
    k = g["k"]
    b = bigsi_amd.BIGSI({"storage-engine": "hip-hbm", "k": k, "m": g["m"], "h": g["h"],
                         "storage-config": {"name": "g7group-attached", "attach": os.path.join(owner["dir"], "g7group.attach")}})
    res = b.storage.res
    assert res.attached and res.is_group and int(res.info().n_shards) == 3 and b.num_samples == len(g["sample_names"])
    for s in g["searches"]:
        check_search(lambda: b.search(g["queries"][s["q"]], s["threshold"], s["score"]), s, "attached group q%d t=%r" % (s["q"], s["threshold"]))
    multi = b.search_batch(g["queries"][:10], 0.4)
    for qi in range(10):
        assert multi[qi] == b.search(g["queries"][qi], 0.4)
    with pytest.raises(_lib.BigsiHipError) as ei:
        b.storage.insert_kmers(3, [g["queries"][0]], k)
    assert ei.value.code == _lib.ERR_STATE
    b.storage.delete_all()


def test_stale_attach_file_falls_back_to_the_snapshot():
    """An attach file whose owner has exited is ignored: the index loads from its snapshot as if `attach` were not there."""
    import bigsi_amd
    d = tempfile.mkdtemp(prefix="bigsi_stale_")
    cfg = {"storage-engine": "hip-hbm", "k": 3, "m": 1000, "h": 3, "storage-config": {"name": "stale-src", "filename": os.path.join(d, "snap")}}
    b = bigsi_amd.BIGSI.build(cfg, [bigsi_amd.BIGSI.bloom(cfg, ["ATA", "TAC"]), bigsi_amd.BIGSI.bloom(cfg, ["ACA"])], ["a", "b"])
    b.storage.sync()
    want = b.search("ATAC", 1.0)
    with open(os.path.join(d, "gone.attach"), "w") as f:
        json.dump({"format": "bigsi-hip-attach-1", "handle": "00" * 64, "pid": 2 ** 22 - 3, "device": 0, "m": 1000, "num_cols": 2, "col_capacity": 1024,
                   "num_hashes": 3, "kv": {}, "uniform_len": None, "written": None, "rowlen": None}, f)
    from bigsi_amd.storage.hip_hbm import HipHbmStorage
    HipHbmStorage.drop("stale-src")
    b2 = bigsi_amd.BIGSI({"storage-engine": "hip-hbm", "k": 3, "m": 1000, "h": 3,
                          "storage-config": {"name": "stale-dst", "filename": os.path.join(d, "snap"), "attach": os.path.join(d, "gone.attach")}})
    assert not b2.storage.res.attached and b2.search("ATAC", 1.0) == want == [{"percent_kmers_found": 100.0, "num_kmers": 2, "num_kmers_found": 2, "sample_name": "a"}]
    b2.storage.res.free()
    # ... and so is a file whose handle does not open (its owner died and the pid now belongs to some other live process -- here: init)
    doc = json.load(open(os.path.join(d, "gone.attach")))
    doc["pid"] = 1
    with open(os.path.join(d, "reused.attach"), "w") as f:
        json.dump(doc, f)
    HipHbmStorage.drop("stale-dst")
    with pytest.warns(UserWarning, match="could not attach"):
        b3 = bigsi_amd.BIGSI({"storage-engine": "hip-hbm", "k": 3, "m": 1000, "h": 3,
                              "storage-config": {"name": "stale-dst2", "filename": os.path.join(d, "snap"), "attach": os.path.join(d, "reused.attach")}})
    assert not b3.storage.res.attached and b3.search("ATAC", 1.0) == want
    b3.storage.res.free()


# ----------------------------------------------------------------------------------------------- threads of one process
def _synthetic_index(name, rows, cols):
    from bigsi_amd.storage import get_storage
    st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": rows, "h": 3, "storage-config": {"name": name, "max_cols": cols}})
    st.delete_all()
    for key, v in (("number_of_rows", rows), ("number_of_cols", cols), ("ksi:bloomfilter_size", rows), ("ksi:num_hashes", 3)):
        st.set_integer(key, v)
    st.fill_synthetic(99, 0, 2)
    return st


def _search(L, handle, blob, soff, n, k, threshold, cap):
    from bigsi_amd import _lib
    nk, nu, off = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n + 1, np.uint64)
    col, cnt = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    rc = L.bigsi_hip_search_batch(handle, blob, _lib.ptr(soff), n, k, float(threshold), 0, _lib.ptr(nk), _lib.ptr(nu), None, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
    return rc, nu, off, col, cnt


def test_four_threads_with_views_of_one_index_against_the_oracle():
    """A serving host: four threads, each with its OWN handle (bigsi_hip_open_view) onto one resident matrix, each hammering the
    one-call search for two seconds with its own queries; every answer equals the oracle's.  The owner refuses to close or
    re-stride while views exist; a view refuses writes."""
    from bigsi_amd import _lib
    from oracle.ref_model import SynthOracle
    L = _lib.lib()
    rows, cols = 200_003, 20_000
    st = _synthetic_index("views", rows, cols)
    rng = np.random.default_rng(5)
    n_thr, per = 4, 6
    seqs = [["".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 300)))) for _ in range(per)] for _ in range(n_thr)]
    orc = SynthOracle(99, 0, rows, cols, 3, 31, 2)
    for t in range(n_thr):
        st.insert_kmers(100 + t, [seqs[t][0]], 31)
        orc.insert_kmers(100 + t, seqs[t][0])
    want = {}
    for t in range(n_thr):
        for thr in (1.0, 0.5):
            exp = []
            for s in seqs[t]:
                u, cnt = orc.counts(s)
                mk = int(np.ceil(u * thr))
                sel = np.flatnonzero(cnt >= (u if thr == 1.0 else mk))
                exp.append((u, sel, cnt[sel].astype(np.uint32)))
            want[(t, thr)] = exp
    views = []
    for _ in range(n_thr):
        v = C.c_void_p()
        _lib.check(L.bigsi_hip_open_view(st.handle, C.byref(v)))
        views.append(v)
    # the owner cannot go away or move the matrix under its views; a view cannot write
    assert L.bigsi_hip_close(st.handle) == _lib.ERR_STATE and L.bigsi_hip_reserve_cols(st.handle, 10 * cols) == _lib.ERR_STATE
    assert L.bigsi_hip_fill_synthetic(views[0], 1, 0, 2) == _lib.ERR_STATE and L.bigsi_hip_clear(views[0]) == _lib.ERR_STATE
    problems, calls = [], [0] * n_thr

    def worker(t):
        blob, soff = _lib.pack_seqs(seqs[t])
        t_end = time.time() + 2.0
        i = 0
        while time.time() < t_end:
            thr = (1.0, 0.5)[i % 2]
            i += 1
            rc, nu, off, col, cnt = _search(L, views[t], blob, soff, per, 31, thr, 1 << 16)
            if rc != 0:
                problems.append((t, "rc %d: %s" % (rc, L.bigsi_hip_last_error().decode())))
                return
            for q, (u, sel, c) in enumerate(want[(t, thr)]):
                lo, hi = int(off[q]), int(off[q + 1])
                if nu[q] != u or not np.array_equal(col[lo:hi], sel) or not np.array_equal(cnt[lo:hi], c):
                    problems.append((t, "query %d at %r differs from the oracle" % (q, thr)))
                    return
            calls[t] += 1
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not problems, problems
    assert min(calls) > 50, calls
    for v in views:
        _lib.check(L.bigsi_hip_close(v))
    st.delete_all()


def test_one_handle_in_two_threads_fails_cleanly():
    """Misuse: two threads inside the SAME handle.  One of them gets BIGSI_ERR_STATE (with a message that names the remedy), the
    other's results stay correct -- no race, no corruption."""
    from bigsi_amd import _lib
    from oracle.ref_model import SynthOracle
    L = _lib.lib()
    rows, cols = 200_003, 20_000
    st = _synthetic_index("misuse", rows, cols)
    rng = np.random.default_rng(6)
    many = ["".join(rng.choice(list("ACGT"), size=1000)) for _ in range(4096)]       # a long streaming call: tens of ms inside the library
    blob, soff = _lib.pack_seqs(many)
    n = len(many)
    out = dict(nk=np.zeros(n, np.uint32), nu=np.zeros(n, np.uint32), off=np.zeros(n + 1, np.uint64), col=np.zeros(1 << 20, np.uint32), cnt=np.zeros(1 << 20, np.uint32))
    rc_long, started = [], threading.Event()

    def long_call():
        started.set()
        done = 0
        while done < 20 and len(rc_long) < 100000:      # (either thread may be the one that is refused: this one just tries again)
            rc = L.bigsi_hip_search_stream(st.handle, blob, _lib.ptr(soff), n, 31, 1.0, 0, _lib.ptr(out["nk"]), _lib.ptr(out["nu"]), None,
                                           _lib.ptr(out["off"]), _lib.ptr(out["col"]), _lib.ptr(out["cnt"]), 1 << 20)
            rc_long.append(rc)
            done += rc == 0
    th = threading.Thread(target=long_call)
    th.start()
    started.wait()
    one = [many[0]]
    b1, o1 = _lib.pack_seqs(one)
    refused = ok = 0
    orc = SynthOracle(99, 0, rows, cols, 3, 31, 2)
    u0, cnt0 = orc.counts(many[0])
    t_end = time.time() + 20.0
    while th.is_alive() and time.time() < t_end:
        rc, nu, off, col, cnt = _search(L, st.handle, b1, o1, 1, 31, 1.0, 1 << 12)
        if rc == _lib.ERR_STATE:
            refused += 1
            assert b"another host thread" in L.bigsi_hip_last_error() and b"bigsi_hip_open_view" in L.bigsi_hip_last_error()
        else:
            assert rc == 0 and nu[0] == u0 and np.array_equal(col[: int(off[1])], np.flatnonzero(cnt0 >= u0))      # a call that went through is right
            ok += 1
    th.join()
    assert set(rc_long) <= {0, _lib.ERR_STATE} and rc_long.count(0) == 20, (rc_long.count(0), set(rc_long))
    assert refused + rc_long.count(_lib.ERR_STATE) > 0, "the two threads never met inside the handle (%d + %d calls went through)" % (ok, len(rc_long))
    for q in (0, n // 2, n - 1):
        u, cnt = orc.counts(many[q])
        assert out["nu"][q] == u and np.array_equal(out["col"][int(out["off"][q]):int(out["off"][q + 1])], np.flatnonzero(cnt >= u))
    st.delete_all()


def test_batch_close_waits_for_a_busy_handle():
    """A QueryBatch finalised (close(), or the garbage collector) while another thread of the process is inside a call on the same
    handle: the library refuses the second thread, the host side waits for the handle instead of leaking the device buffers."""
    from bigsi_amd import _lib
    L = _lib.lib()
    st = _synthetic_index("closebusy", 200_003, 20_000)
    rng = np.random.default_rng(8)
    many = ["".join(rng.choice(list("ACGT"), size=1000)) for _ in range(4096)]
    blob, soff = _lib.pack_seqs(many)
    n = len(many)
    out = dict(nk=np.zeros(n, np.uint32), nu=np.zeros(n, np.uint32), off=np.zeros(n + 1, np.uint64), col=np.zeros(1 << 20, np.uint32), cnt=np.zeros(1 << 20, np.uint32))
    batches = [st.new_batch(many[:4], 31) for _ in range(200)]
    stop, rcs = threading.Event(), []

    def long_calls():
        while not stop.is_set():
            rcs.append(L.bigsi_hip_search_stream(st.handle, blob, _lib.ptr(soff), n, 31, 1.0, 0, _lib.ptr(out["nk"]), _lib.ptr(out["nu"]), None,
                                                 _lib.ptr(out["off"]), _lib.ptr(out["col"]), _lib.ptr(out["cnt"]), 1 << 20))
    th = threading.Thread(target=long_calls)
    th.start()
    try:
        while not rcs:
            time.sleep(0.001)
        for b in batches:
            b.close()                          # never raises: waits out the stream call when it meets one
            assert b.b is None
    finally:
        stop.set()
        th.join()
    assert set(rcs) <= {0, _lib.ERR_STATE} and rcs.count(0) >= 1
    st.delete_all()
