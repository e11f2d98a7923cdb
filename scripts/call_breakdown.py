"""Where the host time of one search goes, call by call of the C ABI (reload / run / fetch_unique / fetch_hits), for a
one-read, a 1000-read and a one-gene batch.  The device work of these batches is tens of microseconds; what a caller
with host buffers waits for is the copies and their synchronisations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd.storage import get_storage

def index(m, n_cols, h):
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "cb%d" % m, "max_cols": n_cols}, "k": 31, "m": m, "h": h}
    st = get_storage(cfg); st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(1, 0, 2)
    return st

def breakdown(st, label, nq, qlen, thr, reps=200):
    rng = np.random.default_rng(0)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    sets = [[lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(nq, qlen), dtype=np.uint8)] for _ in range(8)]
    b = st.new_batch(sets[0], 31)
    t = dict(reload=0.0, run=0.0, unique=0.0, hits=0.0)
    for i in range(reps + 5):
        if i == 5:
            t = dict.fromkeys(t, 0.0)
        t0 = time.perf_counter(); b.reload(sets[i % 8]); t1 = time.perf_counter()
        b.run(thr, sparse_counts=True); t2 = time.perf_counter()
        b.unique(); t3 = time.perf_counter()
        b.hits(); t4 = time.perf_counter()
        t["reload"] += t1 - t0; t["run"] += t2 - t1; t["unique"] += t3 - t2; t["hits"] += t4 - t3
    tot = sum(t.values())
    print("%-28s thr=%.1f  total %6.1f us | " % (label, thr, tot / reps * 1e6) + "  ".join("%s %5.1f" % (k, v / reps * 1e6) for k, v in t.items()))
    # the one-call entry point
    st.search_batch(sets[0], 31, thr)
    t0 = time.perf_counter()
    for i in range(reps):
        st.search_batch(sets[i % 8], 31, thr)
    print("%-28s thr=%.1f  search_batch (one call, Python wrapper included) %6.1f us" % (label, thr, (time.perf_counter() - t0) / reps * 1e6))
    # the C entry point alone (arguments prepared once: what a non-Python binder pays)
    from bigsi_amd import _lib
    packs = [_lib.pack_seqs(s_) for s_ in sets]
    nk, nu, off = np.zeros(nq, np.uint32), np.zeros(nq, np.uint32), np.zeros(nq + 1, np.uint64)
    col, cnt = np.zeros(1 << 16, np.uint32), np.zeros(1 << 16, np.uint32)
    fn = _lib.lib().bigsi_hip_search_batch
    # argument conversion (numpy .ctypes.data: ~1 us per pointer) outside the timed loop: what a C / C++ / Go binder pays is the call
    argv = [(st.handle, blob, _lib.ptr(soff), nq, 31, float(thr), 0, _lib.ptr(nk), _lib.ptr(nu), None, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), col.size) for blob, soff in packs]
    ts = []
    trace = getattr(_lib.lib(), "bigsi_hip_debug_call_trace", None) if "tuning" in os.environ.get("BIGSI_HIP_LIB", "") else None
    if trace:
        trace(None, 1)
    for i in range(reps):
        t0 = time.perf_counter()
        rc = fn(*argv[i % 8])
        ts.append(time.perf_counter() - t0)
        _lib.check(rc)
    if trace:          # tuning library: where inside the call the host's time goes
        tr = (_lib.C.c_uint64 * 16)()
        trace(tr, 1)
        print("%-28s thr=%.1f  inside the call (host clock): " % (label, thr) +
              "  ".join("%s %.1f" % (nm, tr[j] / tr[15] / 1e3) for j, nm in ((1, "stage"), (2, "run"), (3, "export"), (4, "wait"), (5, "collect"))) + " us")
    ts.sort()
    print("%-28s thr=%.1f  bigsi_hip_search_batch (the C call alone)            %6.1f us median  (p10 %.1f, p90 %.1f, mean %.1f)"
          % (label, thr, ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6, ts[len(ts) * 9 // 10] * 1e6, sum(ts) / len(ts) * 1e6))
    b.close()

st = index(1_000_000, 10_000, 3)
for thr in (1.0, 0.4):
    breakdown(st, "C2 index, 1 x 61 bp", 1, 61, thr)
    breakdown(st, "C2 index, 1000 x 61 bp", 1000, 61, thr)
# host-visible streaming rate: 64 x 1000 reads in ONE bigsi_hip_search_stream call (host sequences in, host hit lists out)
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
n_many = int(os.environ.get("BIGSI_STREAM_READS", "64000"))
many = [lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(n_many, 61), dtype=np.uint8)]
from bigsi_amd import _lib
blob, soff = _lib.pack_seqs(many)
nk, nu, off = np.zeros(len(many), np.uint32), np.zeros(len(many), np.uint32), np.zeros(len(many) + 1, np.uint64)
col, cnt = np.zeros(1 << 20, np.uint32), np.zeros(1 << 20, np.uint32)
for thr in (1.0, 0.4):
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        _lib.check(_lib.lib().bigsi_hip_search_stream(st.handle, blob, _lib.ptr(soff), len(many), 31, thr, 0, _lib.ptr(nk), _lib.ptr(nu), None,
                                                      _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), col.size))
        best = min(best, time.perf_counter() - t0)
    print("C2 index, %d x 61 bp in one bigsi_hip_search_stream call thr=%.1f: %.2f ms = %.0f M k-mer lookups/s host-visible (%d hits)"
          % (len(many), thr, best * 1e3, float(nu.sum()) / best / 1e6, int(off[-1])))
st.delete_all()
st = index(10_000_000, 100_000, 4)
for thr in (1.0, 0.4):
    breakdown(st, "C3 index, 1 x 1 kbp", 1, 1000, thr)
st.delete_all()
