"""Where BIGSI.search_stream spends host time on read-length queries (cProfile on the C2 index shape).
    python scripts/stream_profile.py [n_reads]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd import BIGSI  # noqa: E402
from bigsi_amd.storage import get_storage  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m, n_cols, h = 1_000_000, 10_000, 3
cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "sp", "max_cols": n_cols}, "k": 31, "m": m, "h": h}
st = get_storage(cfg)
st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
for c in range(n_cols):
    st.set_string("metadata:%d" % c, "s%d" % c)
st.set_integer("metadata:colour_count", n_cols)
st.fill_synthetic(1, 0, 2)
rng = np.random.default_rng(0)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
seqs = [lut[r].tobytes().decode("ascii") for r in rng.integers(0, 4, size=(nq, 61), dtype=np.uint8)]
for i in range(0, nq, 997):
    st.insert_kmers(i % n_cols, [seqs[i]], 31)
b = BIGSI(cfg)
list(b.search_stream(seqs[:20000], 1.0))
t0 = time.perf_counter()
n_hit = sum(1 for _, r in b.search_stream(seqs, 1.0) if r)
dt = time.perf_counter() - t0
print("search_stream: %.2f M reads/s, %.1f M lookups/s, %d reads with hits" % (nq / dt / 1e6, nq * 31 / dt / 1e6, n_hit))
pr = cProfile.Profile()
pr.enable()
for _ in b.search_stream(seqs[: nq // 4], 1.0):
    pass
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
b.delete()
