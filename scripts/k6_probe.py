"""K6 latency probe (scripts/, not part of the product): the scored-search device calls of one c5-shaped batch, alone on the
device -- synchronous score_hits, begin/end on the score stream, begin/end ordered on the index stream."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bigsi_amd import _lib
from bigsi_amd.storage import get_storage
m, n, h, k = 8_000_000, 62_500, 3, 31
st = get_storage({"storage-engine": "hip-hbm", "storage-config": {"name": "probe", "max_cols": n}, "m": m, "h": h, "k": k})
st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
st.fill_synthetic(1, 0, 2)
rng = np.random.default_rng(0)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
seqs = [lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(256, 1000), dtype=np.uint8)]
for qi in range(16):
    for t in range(16):
        st.insert_kmers((7919 * (16 * qi + t) + 11) % n, [seqs[qi][:710]], k)
b = st.new_batch(seqs, k)
b.run(0.4, sparse_counts=True)
nk, nu, _ = b.unique()
off, col, cnt = b.hits()
print("hits", int(off[-1]))
def T(f, reps=50):
    f(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps * 1e3
_lib.check(_lib.lib().bigsi_hip_set_profiling(st.handle, 1))
print("score_hits sync      %.3f ms" % T(lambda: b.score_hits(off, col, cnt, nk)))
s = _lib.Stats(); _lib.check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(s), 1))
print("  kernels %.3f ms per call (%d calls)" % (s.presence_ms / max(s.presence_launches, 1), s.presence_launches))
for ordered in (False, True):
    tb, te = [], []
    for _ in range(50):
        t0 = time.perf_counter(); b.score_hits_begin(off, col, cnt, nk, ordered=ordered); t1 = time.perf_counter()
        b.score_hits_end(); t2 = time.perf_counter()
        tb.append(t1 - t0); te.append(t2 - t1)
    print("begin/end ordered=%s: begin %.3f ms, end %.3f ms" % (ordered, np.median(tb) * 1e3, np.median(te) * 1e3))
    s = _lib.Stats(); _lib.check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(s), 1))
    print("  kernels %.3f ms per call" % (s.presence_ms / max(s.presence_launches, 1)))
print("presence_hits (ASCII) %.3f ms" % T(lambda: b.presence_hits(off, col, nk)))
print("hits() %.3f ms, unique() %.3f ms" % (T(b.hits), T(b.unique)))
# does end() of an ORDERED request wait for work queued behind it?  (a second batch run right after begin)
b2 = st.new_batch(seqs, k)
b2.run(0.4, sparse_counts=True); b2.hits()
for label, runs in (("nothing behind", 0), ("one run behind", 1), ("three runs behind", 3)):
    te = []
    for _ in range(20):
        b.score_hits_begin(off, col, cnt, nk, ordered=True)
        for _ in range(runs):
            b2.run(0.4, sparse_counts=True)
        t1 = time.perf_counter(); b.score_hits_end(); te.append(time.perf_counter() - t1)
        t1 = time.perf_counter(); b2.hits(); tr = time.perf_counter() - t1
    print("ordered end() with %s: %.3f ms (then waiting for the runs: %.3f ms)" % (label, np.median(te) * 1e3, tr * 1e3))
# the bench's three-deep loop: step k = run X(k); hits of Y(k-1) + begin(Y); end(job begun at step k-1, on X)
X, Y = b, b2
hX = (off, col, cnt)
Y.run(0.4, sparse_counts=True); hY = Y.hits()
X.run(0.4, sparse_counts=True)
X.hits()
X.score_hits_begin(*hX, nk, ordered=True)
job_on, cur, other, hc, ho = X, Y, X, hY, hX
t_run, t_hits, t_begin, t_end = [], [], [], []
t00 = time.perf_counter()
for step in range(200):
    t0 = time.perf_counter(); cur.run(0.4, sparse_counts=True); t1 = time.perf_counter()
    # the batch launched one step ago is `other`... (X and Y alternate)
    prev = other
    # job begun last step is on `cur`'s object (two staged batches): finish it AFTER begin of prev
    if step:
        hprev = prev.hits(); t2 = time.perf_counter()
        pending_end = job_on
        prev.score_hits_begin(*hprev, nk, ordered=True); t3 = time.perf_counter()
        pending_end.score_hits_end(); t4 = time.perf_counter()
        job_on = prev
        t_run.append(t1 - t0); t_hits.append(t2 - t1); t_begin.append(t3 - t2); t_end.append(t4 - t3)
    cur, other = other, cur
tot = (time.perf_counter() - t00) / 200 * 1e3
print("three-deep loop: %.3f ms/step; run %.3f, hits %.3f, begin %.3f, end %.3f ms" % (tot, np.median(t_run) * 1e3, np.median(t_hits) * 1e3, np.median(t_begin) * 1e3, np.median(t_end) * 1e3))
b2.close()
b.close(); st.delete_all()
