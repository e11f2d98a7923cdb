"""Application-level throughput: Python strings in -> result dicts out, through BIGSI.search_stream (pipelined) and
through sequential search_batch calls, on the C3 index shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd import BIGSI
from bigsi_amd.storage import get_storage

m, n_cols, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
qlen, nq = int(sys.argv[4]), int(sys.argv[5])
cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "bulk", "max_cols": n_cols}, "k": 31, "m": m, "h": h}
st = get_storage(cfg); st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
for c in range(0, n_cols, max(1, n_cols // 50)):
    st.set_string("metadata:%d" % c, "s%d" % c)
st.set_integer("metadata:colour_count", n_cols)
st.fill_synthetic(1, 0, 2)
rng = np.random.default_rng(0)
seqs = ["".join(rng.choice(list("ACGT"), size=qlen)) for _ in range(nq)]
for i in range(0, nq, 97):
    st.insert_kmers((i % 50) * max(1, n_cols // 50), [seqs[i]], 31)
b = BIGSI(cfg)
nk = nq * (qlen - 30)
for thr in (1.0, 0.4):
    b.search_batch(seqs[:256], thr)
    t0 = time.perf_counter()
    r1 = [r for i in range(0, nq, 256) for r in b.search_batch(seqs[i:i + 256], thr)]
    t1 = time.perf_counter()
    r2 = [r for _, r in b.search_stream(seqs, thr)]
    t2 = time.perf_counter()
    assert r1 == r2
    print("threshold %.1f: sequential batches %.1f M lookups/s, pipelined stream %.1f M lookups/s, %d queries with hits"
          % (thr, nk / (t1 - t0) / 1e6, nk / (t2 - t1) / 1e6, sum(1 for r in r1 if r)))
b.delete()
