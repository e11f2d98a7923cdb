import cProfile, pstats, sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from bigsi_amd import BIGSI
from bigsi_amd.storage import get_storage
m, n_cols, h = 1_000_000, 100_000, 4
cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "lat", "max_cols": n_cols}, "k": 31, "m": m, "h": h}
st = get_storage(cfg); st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
st.set_integer("metadata:colour_count", n_cols)
st.fill_synthetic(1, 0, 2)
b = BIGSI(cfg)
rng = np.random.default_rng(0)
seqs = ["".join(rng.choice(list("ACGT"), size=1000)) for _ in range(50)]
for s in seqs: b.search(s)
t0 = time.perf_counter()
for _ in range(20):
    for s in seqs: b.search(s)
print("us per search", (time.perf_counter() - t0) / 1000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    for s in seqs: b.search(s)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
