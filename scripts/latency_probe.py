"""Single-query latency of BIGSI.search() through the whole Python + C-ABI stack (not a throughput benchmark)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd import BIGSI
from bigsi_amd.storage import get_storage

def probe(m, n_cols, h, qlen, reps=30):
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "lat", "max_cols": n_cols}, "k": 31, "m": m, "h": h}
    st = get_storage(cfg); st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    for c in range(min(n_cols, 4)):
        st.set_string("metadata:%d" % c, "s%d" % c)
    st.set_integer("metadata:colour_count", n_cols)
    st.fill_synthetic(1, 0, 2)
    b = BIGSI(cfg)
    rng = np.random.default_rng(0)
    seqs = ["".join(rng.choice(list("ACGT"), size=qlen)) for _ in range(reps)]
    for thr in (1.0, 0.4):
        b.search(seqs[0], thr)
        t0 = time.perf_counter()
        for s in seqs:
            b.search(s, thr)
        dt = (time.perf_counter() - t0) / reps
        print("m=%d N=%d h=%d qlen=%d threshold=%g: %.3f ms per search()" % (m, n_cols, h, qlen, thr, dt * 1e3))
    st.delete_all()

probe(1000, 3, 3, 61)
probe(1_000_000, 10_000, 3, 61)
probe(10_000_000, 100_000, 4, 1000)
