"""200 one-query calls of storage.search_batch (1 kbp against the 10 M x 100 k index), exact and at threshold 0.4: run under
scripts/prof.sh to get the kernel chain of a latency-bound call (profiles/r02_single_query_kernel_stats.csv)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd.storage import get_storage
m, n_cols, h = 10_000_000, 100_000, 4
cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "sq", "max_cols": n_cols}, "k": 31, "m": m, "h": h}
st = get_storage(cfg); st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
st.fill_synthetic(1, 0, 2)
rng = np.random.default_rng(0)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
seqs = [lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(64, 1000), dtype=np.uint8)]
for thr in (1.0, 0.4):
    for i in range(200):
        st.search_batch([seqs[i % 64]], 31, thr)
st.delete_all()
