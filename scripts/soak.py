"""Soak: thousands of runs, reloads and batch create/destroy cycles on one index; device memory must return to its
starting level (no leaks) and results must stay identical."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd.storage import get_storage
import torch

def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e6

m, n_cols, h = 2_000_000, 20_000, 3
st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": m, "h": h, "storage-config": {"name": "soak", "max_cols": n_cols}})
st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
st.fill_synthetic(1, 0, 2)
rng = np.random.default_rng(0)
pool = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(31, 2000)))) for _ in range(400)]
st.insert_kmers(5, pool[:3], 31)
base = used()
ref = None
b = st.new_batch(pool[:64], 31)
for it in range(3000):
    thr = (1.0, 0.4, 0.0)[it % 3] if it % 50 else 0.0
    b.run(thr, sparse_counts=bool(it % 2))
    if it % 100 == 0:
        b.hits()
for it in range(300):
    seqs = [pool[(it * 7 + j) % len(pool)] for j in range(1 + it % 90)]
    b.reload(seqs)
    b.run(0.5)
    off, col, cnt = b.hits()
    if it % 60 == 0:
        b2 = st.new_batch(seqs, 31); b2.run(0.5); o2, c2, n2 = b2.hits(); b2.close()
        assert np.array_equal(off, o2) and np.array_equal(col, c2) and np.array_equal(cnt, n2)
for it in range(300):
    bb = st.new_batch(pool[it % 300: it % 300 + 5], 31); bb.run(1.0); bb.hits(); bb.close()
b.close()
after = used()
print("device MB used: before batches %.0f, after %.0f (delta %.1f)" % (base, after, after - base))
assert after - base < 64, "device memory grew"
st.delete_all()
print("soak OK")
