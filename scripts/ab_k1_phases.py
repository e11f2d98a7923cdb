"""Tuning build only: phase breakdown of k_kmerize_lds (K1 of gene-length queries), 8192 x 1 kbp.
BIGSI_HIP_LIB=bigsi_amd/libbigsi_hip_tuning.so python scripts/ab_k1_phases.py [n_queries] [threshold]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigsi_amd import _lib  # noqa: E402
from bigsi_amd._lib import check  # noqa: E402
from scripts.measure import open_index, rand_seqs, stats  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
st, _ = open_index("k1ph", 1_000_000, 12_500, 4)
b = st.new_batch(rand_seqs(np.random.default_rng(1), nq, 1000), 31)
L = _lib.lib()
for _ in range(3):
    b.run(thr, sparse_counts=True)
check(L.bigsi_hip_synchronize(st.handle))
check(L.bigsi_hip_set_profiling(st.handle, 1))
stats(st)
for _ in range(5):
    b.run(thr, sparse_counts=True)
s = stats(st)
out = {"n_queries": nq, "threshold": thr, "k1_ms": s.kmerize_ms / max(s.kmerize_launches, 1)}
ph = np.zeros((1024, 8), np.uint64)
L.bigsi_hip_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
check(L.bigsi_hip_debug_phases(st.handle, ph.ctypes.data, 1024))
t = ph.astype(np.int64) / 100.0
names = ["stage", "fingerprints", "insert", "rank+hash", "sort", "pos_unique"]
for i, nm in enumerate(names):
    out[nm + "_us"] = float(np.median(t[:, i + 1] - t[:, i]))
out["workgroup_us"] = float(np.median(t[:, 6] - t[:, 0]))
print(out)
b.close()
st.delete_all()
