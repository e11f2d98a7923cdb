import sys, time, numpy as np
sys.path.insert(0, ".")
from bigsi_amd import _lib
from bigsi_amd._lib import check
from scripts.measure import open_index, rand_seqs
st, _ = open_index("ee", 10_000_000, 100_000, 4)
L = _lib.lib()
b = st.new_batch(rand_seqs(np.random.default_rng(1), 2048, 1000), 31)
for thr in (0.4, 0.7, 0.9):
    for ee in (False, True):
        for _ in range(2): b.run(thr, sparse_counts=True, early_exit=ee)
        check(L.bigsi_hip_synchronize(st.handle))
        t0 = time.perf_counter()
        for _ in range(4): b.run(thr, sparse_counts=True, early_exit=ee)
        check(L.bigsi_hip_synchronize(st.handle))
        dt = (time.perf_counter() - t0) / 4
        _, nu, _ = b.unique()
        print("threshold %.1f early_exit=%d: %.2f ms per 2048 x 1 kbp, %.1f M lookups/s" % (thr, ee, dt * 1e3, nu.sum() / dt / 1e6))
b.close(); st.delete_all()
