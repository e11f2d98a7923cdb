"""Device-side timeline of ONE-call searches of a single gene-length query (bigsi_hip_search_batch on the C3 index): run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python scripts/one_call_timeline.py run
then  python scripts/one_call_timeline.py report DIR  prints, per threshold, the median duration of every kernel of a call and the
gaps between them (end of one kernel to the start of the next) -- what the chain of dependent launches costs beside the kernels."""
import csv
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from bigsi_amd import _lib
    from bigsi_amd.storage import get_storage
    reads = os.environ.get("MODE", "") == "reads"          # MODE=reads: 1000 x 61-mers in one call on the C2 index instead
    m, n_cols, h = (1_000_000, 10_000, 3) if reads else (int(os.environ.get("ROWS", 10_000_000)), 100_000, 4)
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "tl", "max_cols": n_cols}, "k": 31, "m": m, "h": h}
    st = get_storage(cfg)
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(1, 0, 2)
    rng = np.random.default_rng(0)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    nq, ql = (1000, 61) if reads else (1, 1000)
    sets = [[lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(nq, ql), dtype=np.uint8)] for _ in range(8)]
    packs = [_lib.pack_seqs(s_) for s_ in sets]
    nk, nu, off = np.zeros(nq, np.uint32), np.zeros(nq, np.uint32), np.zeros(nq + 1, np.uint64)
    col, cnt = np.zeros(1 << 16, np.uint32), np.zeros(1 << 16, np.uint32)
    fn = _lib.lib().bigsi_hip_search_batch
    for thr in (1.0, 0.4):
        argv = [(st.handle, blob, _lib.ptr(soff), nq, 31, float(thr), 0, _lib.ptr(nk), _lib.ptr(nu), None, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), col.size) for blob, soff in packs]
        ts = []
        for i in range(60):
            t0 = time.perf_counter()
            _lib.check(fn(*argv[i % 8]))
            ts.append(time.perf_counter() - t0)
            time.sleep(0.002)          # (calls well apart: the trace is cut into calls by the idle time between them)
        print("threshold %.1f: %.1f us median per call (under the profiler)" % (thr, float(np.median(ts[10:])) * 1e6), flush=True)
    st.delete_all()


def report(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bigsi::", "")))
    rows.sort()
    calls, cur = [], []
    for s, e, n in rows:
        if cur and s - cur[-1][1] > 500_000:          # > 0.5 ms idle: a new call
            calls.append(cur)
            cur = []
        cur.append((s, e, n))
    calls.append(cur)
    by_shape = {}
    for c in calls:
        if any("fill_synth" in n for _, _, n in c):
            continue
        by_shape.setdefault(tuple(n for _, _, n in c), []).append(c)
    for shape, cs in by_shape.items():
        if len(cs) < 20:
            continue
        cs = cs[5:]
        print("%d calls of %d kernels; first kernel start -> last kernel end: %.1f us median" % (len(cs), len(shape), float(np.median([c[-1][1] - c[0][0] for c in cs])) / 1e3))
        for i, n in enumerate(shape):
            dur = float(np.median([c[i][1] - c[i][0] for c in cs])) / 1e3
            gap = float(np.median([c[i][0] - c[i - 1][1] for c in cs])) / 1e3 if i else 0.0
            print("   %-46s gap before %5.1f us   runs %5.1f us" % (n[:46], gap, dur))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
