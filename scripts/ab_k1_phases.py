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
acc = []
for i in range(200):
    st.search_batch(qs[i % 8], 31, thr)
    if i >= 20:
        ph = np.zeros((1, 8), np.uint64)
        check(L.bigsi_hip_debug_phases(st.handle, ph.ctypes.data, 1))
        acc.append(np.diff(ph[0, :7].astype(np.int64)) / 100.0)
a = np.median(np.array(acc), axis=0)
names = ["table + sequence in", "fingerprints", "insert", "resolve + scan + hash + rows out", "sort (off)", "pos_unique"]
print({"threshold": thr, "phases_us": {n_: round(float(x), 2) for n_, x in zip(names, a)}, "sum_us": round(float(a.sum()), 2)})
st.delete_all()
