"""Interleaved in-process A/B of the address-ordered row lists (k_sort_rows) on the C3 workload: alternates runs with and
without BIGSI_RUN_NO_SORT on one resident index and reports the row-AND kernel time (HIP events) per arm."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd import _lib
from bigsi_amd._lib import check
from bigsi_amd.storage import get_storage

rows, cols, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": rows, "h": h, "storage-config": {"name": "ab", "max_cols": cols}})
st.delete_all()
for key, v in (("number_of_rows", rows), ("number_of_cols", cols), ("ksi:bloomfilter_size", rows), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
st.fill_synthetic(20260928, 0, 2)
rng = np.random.default_rng(1)
seqs = ["".join(rng.choice(list("ACGT"), size=1000)) for _ in range(256)]
batch = st.new_batch(seqs, 31)
L = _lib.lib()
check(L.bigsi_hip_set_profiling(st.handle, 1))
stats = _lib.Stats()
for thr in (1.0, 0.4):
    res = {0: [], 1: []}
    for rnd in range(12):
        for nosort in (0, 1):
            flags = (_lib.RUN_NO_SORT if nosort else 0) | _lib.RUN_SPARSE_COUNTS
            for _ in range(5):
                check(L.bigsi_hip_batch_run(batch.b, thr, flags))
            check(L.bigsi_hip_stats(st.handle, _lib.C.byref(stats), 1))
            if rnd >= 2:
                res[nosort].append(stats.and_ms / stats.and_launches)
    a, b = np.array(res[0]), np.array(res[1])
    print("threshold %.1f  sorted: median %.4f ms (min %.4f)   hash order: median %.4f ms (min %.4f)   speedup %.3fx"
          % (thr, np.median(a), a.min(), np.median(b), b.min(), np.median(b) / np.median(a)))
batch.close(); st.delete_all()
