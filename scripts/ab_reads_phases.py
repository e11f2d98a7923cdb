"""Tuning build only: phase breakdown of k_reads_fused on BASELINE configs[1] (1M x 10k, h=3, 1000 x 61-mers), 32 distinct
batches cycling.  BIGSI_HIP_LIB=bigsi_amd/libbigsi_hip_tuning.so [BIGSI_HIP_READ_GROUPS=n] python scripts/ab_reads_phases.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigsi_amd import _lib  # noqa: E402
from bigsi_amd._lib import check  # noqa: E402
from scripts.measure import open_index, rand_seqs, stats  # noqa: E402

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
m, n, h = 1_000_000, 10_000, 3
st, _ = open_index("c2ph", m, n, h)
rng = np.random.default_rng(7)
bs = [st.new_batch(rand_seqs(rng, 1000, 61), 31) for _ in range(32)]
L = _lib.lib()
for i in range(64):
    bs[i % 32].run(thr, sparse_counts=True)
check(L.bigsi_hip_synchronize(st.handle))
t0 = time.perf_counter()
for i in range(320):
    bs[i % 32].run(thr, sparse_counts=True)
check(L.bigsi_hip_synchronize(st.handle))
wall = (time.perf_counter() - t0) / 320 * 1e6
out = {"threshold": thr, "groups": os.environ.get("BIGSI_HIP_READ_GROUPS", "default"), "step_us": round(wall, 2)}
if hasattr(L, "bigsi_hip_debug_phases"):
    ph = np.zeros((1000, 8), np.uint64)
    L.bigsi_hip_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    check(L.bigsi_hip_debug_phases(st.handle, ph.ctypes.data, 1000))
    ph = ph.astype(np.int64)
    t = (ph - ph[:, :1].min()) / 100.0          # us since the first workgroup started
    out.update(start_spread_us=float(t[:, 0].max()), k1_us=float(np.median(t[:, 1] - t[:, 0])), k2_us=float(np.median(t[:, 2] - t[:, 1])),
               k4_us=float(np.median(t[:, 3] - t[:, 2])), k1_offsets=float(np.median(t[:256, 4] - t[:256, 0])), k1_fp=float(np.median(t[:256, 5] - t[:256, 4])),
               k1_dedupe=float(np.median(t[:256, 6] - t[:256, 5])), k1_hash=float(np.median(t[:256, 7] - t[:256, 6])), k1_rest=float(np.median(t[:256, 1] - t[:256, 7])),
               k2_end_pct=[float(x) for x in np.percentile(t[:, 2], [10, 50, 90, 99])], k1_end_max=float(t[:, 1].max()), k2_end_max=float(t[:, 2].max()), end_max=float(t[:, 3].max()))
print(out)
for b in bs:
    b.close()
st.delete_all()
