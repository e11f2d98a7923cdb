"""Tuning build only: where the time of k_kmerize_lds goes for ONE 1 kbp query of a one-call search (phase timestamps of the
workgroup, 100 MHz wall clock).  BIGSI_HIP_LIB=bigsi_amd/libbigsi_hip_tuning.so python scripts/ab_k1_phases.py [threshold]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigsi_amd import _lib  # noqa: E402
from bigsi_amd._lib import check  # noqa: E402
from scripts.measure import open_index, rand_seqs  # noqa: E402

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
st, _ = open_index("k1ph", 1_000_000, 100_000, 4)
rng = np.random.default_rng(7)
qs = [rand_seqs(rng, 1, 1000) for _ in range(8)]
L = _lib.lib()
L.bigsi_hip_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
acc, acc4, between, andk = [], [], [], []
import time
calls = []
for i in range(200):
    t0 = time.perf_counter()
    st.search_batch(qs[i % 8], 31, thr)
    calls.append(time.perf_counter() - t0)
    if i >= 20:
        ph = np.zeros((1024, 8), np.uint64)
        check(L.bigsi_hip_debug_phases(st.handle, ph.ctypes.data, 1024))
        acc.append(np.diff(ph[0, :7].astype(np.int64)) / 100.0)
        acc4.append(np.diff(ph[1000, :7].astype(np.int64)) / 100.0)          # k_hits_write's single-workgroup route (group 1000)
        between.append((int(ph[1000, 0]) - int(ph[0, 6])) / 100.0)
        andk.append([(int(ph[1001, 0]) - int(ph[0, 6])) / 100.0, (int(ph[1001, 1]) - int(ph[1001, 0])) / 100.0, (int(ph[1001, 2]) - int(ph[1001, 1])) / 100.0,
                     (int(ph[1002, 0]) - int(ph[1001, 0])) / 100.0, (int(ph[1002, 2]) - int(ph[1002, 0])) / 100.0, (int(ph[1000, 0]) - int(ph[1002, 2])) / 100.0])
a = np.median(np.array(acc), axis=0)
names = ["table + sequence in", "fingerprints", "insert", "resolve + scan + hash + rows out", "sort (off)", "pos_unique"]
print({"threshold": thr, "phases_us": {n_: round(float(x), 2) for n_, x in zip(names, a)}, "sum_us": round(float(a.sum()), 2)})
a4 = np.median(np.array(acc4), axis=0)
names4 = ["words in", "scan", "hits out", "block header (stores issued)", "system fence (incl. waiting for the stores)", "barrier + flag"]
print({"k_hits_write_us": {n_: round(float(x), 2) for n_, x in zip(names4, a4)}, "sum_us": round(float(a4.sum()), 2),
       "end of K1 -> start of K4 (the row kernel and two launch boundaries)": round(float(np.median(between)), 2)})
st.delete_all()
ak = np.median(np.array(andk), axis=0)
print({"k_and_exact_us": dict(zip(["end of K1 -> first workgroup starts", "first workgroup: its query's numbers in", "first workgroup: its rows streamed",
                                   "first -> one of the last workgroups starts", "that workgroup: start -> rows streamed", "its rows streamed -> K4 starts"],
                                  [round(float(x), 2) for x in ak]))})
print({"storage.search_batch_us (Python call included)": round(float(np.median(calls[20:]) * 1e6), 1)})
