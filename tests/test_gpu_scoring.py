"""K6 -- BIGSI.score on the device (bigsi_hip_batch_score_hits / bigsi_hip_score_presence, csrc/bigsi_score.hpp) -- against
the reference's golden scores (tests/golden/g5_scoring.json: the reference's own Scorer.score outputs), CPython's round(), the
scalar restatement of scoring/score.py, and the presence strings the single-sequence kernel (k_presence) and the oracle give.
Everything exact except evalue / pvalue (tolerances of conftest.py).  Needs a real MI355X: `pytest -m gpu`."""
import contextlib
import itertools
import os

import numpy as np
import pytest

from conftest import assert_results_equal

pytestmark = pytest.mark.gpu

_counter = itertools.count()


def cfg(k, m, h, **sc):
    sc.setdefault("name", "k6_%d" % next(_counter))
    return {"storage-engine": "hip-hbm", "storage-config": sc, "k": k, "m": m, "h": h}


def device_score(strings, found=None, unique=None):
    from bigsi_amd import _lib
    from bigsi_amd.scoring import HIT_SCORE_DTYPE, pack_presence
    bits, off, lens = pack_presence(strings)
    rec = np.zeros(max(len(strings), 1), HIT_SCORE_DTYPE)
    _lib.check(_lib.lib().bigsi_hip_score_presence(0, _lib.ptr(bits), _lib.ptr(off), _lib.ptr(lens), _lib.ptr(found), _lib.ptr(unique),
                                                   len(strings), _lib.ptr(rec)))
    return rec[:len(strings)]


def test_k6_device_scores_equal_golden_scores_and_scalar_scorer():
    """All 438 golden Scorer.score cases (incl. the reference's known answer bigsi/tests/scoring.py:10-31) and ~500 seeded strings
    (lengths around the 64-position word edges, gaps of SNP length) through k_score_packed: tallies, the rounded score chain,
    SNP totals and percent_kmers_found bit-equal."""
    from test_abi_and_host import check_records_against_golden_and_scalar
    check_records_against_golden_and_scalar(device_score)


def test_k6_round_on_the_device_is_cpythons_round():
    """py_round2 as compiled for gfx950, seen through percent_kmers_found = round(100 * found / unique, 2) for every
    0 <= found <= unique <= 1200 (720 k quotients; the x.xx5 cases among them are the hard ones) and through single-gap strings,
    whose scores are round(n - gap + gap', 2)-style chains over every gap length up to 700."""
    uniq = np.repeat(np.arange(1, 1201, dtype=np.uint32), np.arange(2, 1202))
    found = np.concatenate([np.arange(u + 1, dtype=np.uint32) for u in range(1, 1201)])
    rec = device_score(["1"] * uniq.size, found, uniq)
    want = [round(100 * float(f) / u, 2) for f, u in zip(found.tolist(), uniq.tolist())]
    assert rec["percent_kmers_found"].tolist() == want
    from bigsi_amd.scoring import SCORE_KEYS, Scorer, score_columns
    strings = ["1" * a + "0" * g + "1" * b for g in range(1, 700, 3) for a, b in ((40, 40), (3, 500), (0, 37))]
    cols = score_columns(device_score(strings), 1000)
    sc = Scorer(1000)
    for i, s in enumerate(strings):
        want = sc.score(s)
        assert {k: c[i] for k, c in zip(SCORE_KEYS, cols)} == want, s[:50]


def build_index(hip_cfg, samples):
    from bigsi_amd import BIGSI
    return BIGSI.build_from_sequences(hip_cfg, samples)


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), size=n))


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_k6_score_hits_equals_strings_and_scalar_scorer(devices):
    """bigsi_hip_batch_score_hits on a small index whose samples share pieces of the queries: the packed bits must be the
    presence strings of k_presence (one sequence at a time, the round-1 kernel) and the records the scalar Scorer's values for
    those strings.  Queries with repeated k-mers (listed pieces), lengths that end inside a 16-position piece / a 64-position
    word, one sequence without hits.  devices = [0, 0, 0]: the same through a three-shard group (each shard scores its hits)."""
    from bigsi_amd.scoring import SCORE_KEYS, Scorer, score_columns, unpack_presence
    rng = np.random.default_rng(11)
    k, m, h = 31, 50021, 3
    queries = [rand_seq(rng, 1000), rand_seq(rng, 31 + 15), rand_seq(rng, 31 + 16), rand_seq(rng, 31 + 63), rand_seq(rng, 31 + 64),
               rand_seq(rng, 300, "AC"), rand_seq(rng, 200) * 3, rand_seq(rng, 2500), rand_seq(rng, 97)]
    samples = {}
    for c in range(150):
        q = queries[c % (len(queries) - 1)]                    # the last query matches nothing
        lo = int(rng.integers(0, max(len(q) - 60, 1)))
        samples["s%d" % c] = [q[lo:lo + int(rng.integers(31, len(q)))], rand_seq(rng, 400)] + ([q[:45]] if c % 3 == 0 else []) + ([q] if c % 4 == 1 else [])
    sc = {"max_cols": 150}
    if devices:
        sc["devices"] = devices
    index = build_index(cfg(k, m, h, **sc), samples)
    batch = index.storage.new_batch(queries, k)
    scalar = Scorer(150)
    for thr in (0.05, 1.0):
        batch.run(thr)
        nk, nu, _ = batch.unique()
        off, colours, counts = batch.hits()
        assert int(off[-1]) > (30 if thr == 1.0 else 100)
        rec, bits, boff = batch.score_hits(off, colours, None if thr == 1.0 else counts, nk)
        text = unpack_presence(bits, boff)
        cols = score_columns(rec, 150)
        t = 0
        for i in range(len(queries)):
            hits = colours[int(off[i]):int(off[i + 1])]
            strs = batch.presence(i, hits, int(nk[i]))
            for c, s in zip(hits.tolist(), strs):
                assert text[8 * int(boff[t]):8 * int(boff[t]) + int(nk[i])] == s, (thr, i, c)
                assert {key: col[t] for key, col in zip(SCORE_KEYS, cols)} == scalar.score(s), (thr, i, c)
                f = int(nu[i]) if thr == 1.0 else int(counts[t])
                assert rec["percent_kmers_found"][t] == round(100 * float(f) / int(nu[i]), 2) and rec["num_kmers"][t] == nk[i]
                t += 1
        assert t == int(off[-1])
        # any subset / order of a sequence's colours, and hit lists that start past zero (what a sliced caller passes)
        part = np.clip(off, 7, int(off[-1]) - 5).astype(np.uint64)
        rec2, bits2, boff2 = batch.score_hits(part, colours, None if thr == 1.0 else counts, nk)
        assert np.array_equal(rec2, rec[7:int(off[-1]) - 5])
        assert unpack_presence(bits2, boff2) == text[8 * int(boff[7]):8 * int(boff[int(off[-1]) - 5])]
    batch.close()
    index.delete()


def test_k6_search_with_scores_in_slices_equals_scalar_assembly():
    """BIGSI.search_batch(score=True) assembles its dicts from K6's records; with a tiny slice budget (several device passes per
    batch, slices cutting through a sequence's hits) the results must equal the round-2 assembly: presence string per hit ->
    scalar Scorer.score -> dict, in the reference's order (count descending, colour ascending), deleted samples dropped."""
    import bigsi_amd.graph.bigsi as gb
    from bigsi_amd.scoring import Scorer
    rng = np.random.default_rng(12)
    k, m, h = 31, 30011, 3
    queries = [rand_seq(rng, 400), rand_seq(rng, 61), rand_seq(rng, 700)]
    samples = {"s%d" % c: [queries[c % 3][: int(rng.integers(40, len(queries[c % 3])))], rand_seq(rng, 300)] + ([queries[c % 3]] if c % 5 < 2 else [])
               for c in range(40)}
    index = build_index(cfg(k, m, h, max_cols=64), samples)
    index.delete_sample("s4")
    old = gb.SCORE_SLICE_CHARS
    try:
        results = {}
        for budget in (old, 2000):
            gb.SCORE_SLICE_CHARS = budget
            results[budget] = {thr: index.search_batch(queries, thr, score=True) for thr in (1.0, 0.3)}
    finally:
        gb.SCORE_SLICE_CHARS = old
    scalar = Scorer(index.scorer.DB_SIZE)
    batch = index.storage.new_batch(queries, k)
    for thr in (1.0, 0.3):
        batch.run(thr)
        nk, nu, _ = batch.unique()
        off, colours, counts = batch.hits()
        for i in range(len(queries)):
            lo, hi = int(off[i]), int(off[i + 1])
            hits = sorted(zip(colours[lo:hi].tolist(), counts[lo:hi].tolist()), key=lambda x: -x[1]) if thr != 1.0 else list(zip(colours[lo:hi].tolist(), counts[lo:hi].tolist()))
            want = []
            for c, f in hits:
                name = index.colour_to_sample(c)
                if name == "D3L3T3D":
                    continue
                s = batch.presence(i, np.array([c], np.uint32), int(nk[i]))[0]
                d = {"percent_kmers_found": round(100 * float(f) / int(nu[i]), 2), "num_kmers": int(nu[i]), "num_kmers_found": f, "sample_name": name}
                d.update(scalar.score(s))
                d["kmer-presence"] = s
                want.append(d)
            assert len(want) >= (3 if thr == 1.0 else 8), (thr, i, len(want))
            for budget in results:
                assert_results_equal(results[budget][thr][i], want, "thr=%r q%d budget=%d" % (thr, i, budget))
                assert [list(r) for r in results[budget][thr][i]] == [list(w) for w in want]
    batch.close()
    index.delete()


@pytest.mark.parametrize("ordered", [False, True])
def test_k6_begin_end_equals_the_synchronous_call_even_when_the_batch_runs_again(ordered):
    """bigsi_hip_batch_score_hits_begin / _end: the request may be collected after the batch has been RUN AGAIN (a serving loop
    three batches deep does exactly that) -- the re-run is ordered behind the request on the device, the results are staged on the
    host.  Same records and bits as the synchronous call, on the score stream and on the index stream."""
    rng = np.random.default_rng(13)
    k, m, h = 31, 40009, 3
    queries = [rand_seq(rng, 500), rand_seq(rng, 95), rand_seq(rng, 1200)]
    samples = {"s%d" % c: [queries[c % 3][: int(rng.integers(60, len(queries[c % 3])))], rand_seq(rng, 200)] for c in range(90)}
    index = build_index(cfg(k, m, h, max_cols=128), samples)
    batch = index.storage.new_batch(queries, k)
    other = index.storage.new_batch([rand_seq(rng, 3000) for _ in range(64)], k)
    batch.run(0.2)
    nk, nu, _ = batch.unique()
    off, colours, counts = batch.hits()
    assert int(off[-1]) > 30
    want = batch.score_hits(off, colours, counts, nk)
    for rounds in range(4):
        batch.score_hits_begin(off, colours, counts, nk, ordered=ordered)
        other.run(0.5)                                   # something else on the index stream
        batch.run(0.9 if rounds % 2 else 0.2)            # the same batch again: K1 rewrites what the request reads
        got = batch.score_hits_end()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[1][: int(got[2][-1])], want[1][: int(want[2][-1])])
        batch.run(0.2)
        o2, c2, n2 = batch.hits()
        assert np.array_equal(o2, off) and np.array_equal(c2, colours) and np.array_equal(n2, counts)
    with pytest.raises(Exception):
        batch.score_hits_begin(off, colours, counts, nk)
        batch.score_hits_begin(off, colours, counts, nk)          # one request per batch at a time
    batch.score_hits_end()
    batch.close()
    other.close()
    index.delete()


def test_scored_search_stream_three_batches_deep_equals_search():
    """BIGSI.search_stream(score=True) queues each batch's K5 + K6 beside the next batch and collects them a step later (three
    workspaces); the results must be those of search(..., score=True) one query at a time, in order, for every batch size."""
    rng = np.random.default_rng(14)
    k, m, h = 31, 30011, 3
    queries = [rand_seq(rng, int(n)) for n in rng.integers(61, 700, size=23)]
    samples = {"s%d" % c: [queries[c % 23][: int(rng.integers(45, len(queries[c % 23])))], rand_seq(rng, 250)] + ([queries[c % 23]] if c % 4 == 0 else [])
               for c in range(60)}
    index = build_index(cfg(k, m, h, max_cols=64), samples)
    for thr in (0.35, 1.0):
        want = [index.search(q, thr, score=True) for q in queries]
        assert sum(len(w) for w in want) > (10 if thr == 1.0 else 40)
        for bs in (1, 2, 5, 23, 100):
            got = list(index.search_stream(iter(queries), thr, score=True, batch_size=bs))
            assert [s for s, _ in got] == queries
            for (s, r), w in zip(got, want):
                assert_results_equal(r, w, "thr=%r bs=%d" % (thr, bs))
        got = list(index.search_stream(iter(queries), thr, score=True, batch_kmers=2000))
        assert [r for _, r in got] == want
    index.delete()


@pytest.mark.parametrize("threshold", [1.0, 0.3])
def test_search_stream_scored_equals_batch_score_hits_over_many_device_batches(threshold):
    """bigsi_hip_search_stream_scored -- score=True for any number of sequences in one C call: 20 000 reads (five device batches,
    three workspaces, each batch's K5 + K6 beside the next batch's row-AND) with planted matches, genes among them (sequences of
    several 64-position words), sequences shorter than k.  Hit lists equal search_stream's; every hit's bits and record equal
    bigsi_hip_batch_score_hits on a batch of that hit's sequence alone for a sample, and the scalar Scorer for every hit; the
    capacity protocol (a sizing call, then one retry with what it reported) returns the same."""
    from bigsi_amd import _lib
    from bigsi_amd.scoring import HIT_SCORE_DTYPE, SCORE_KEYS, Scorer, score_columns, unpack_presence
    rng = np.random.default_rng(21 + int(threshold * 10))
    k, m, h = 31, 200003, 3
    genes = [rand_seq(rng, int(n)) for n in (1000, 450, 95, 31, 2000)]
    samples = {}
    for c in range(96):
        g = genes[c % len(genes)]
        cut = int(rng.integers(31, len(g) + 1))
        samples["s%d" % c] = [g if c % 4 == 0 else g[:cut], rand_seq(rng, 500)]
    index = build_index(cfg(k, m, h, max_cols=128), samples)
    seqs = []
    for i in range(20000):
        r = i % 97
        if r == 0:
            seqs.append(genes[(i // 97) % len(genes)])
        elif r == 1:
            g = genes[0]
            lo = int(rng.integers(0, len(g) - 61))
            seqs.append(g[lo:lo + 61])
        elif r == 2:
            seqs.append(rand_seq(rng, int(rng.integers(0, 31))))       # shorter than k: no k-mers, no hits
        else:
            seqs.append(rand_seq(rng, 61))
    st = index.storage
    nk, nu, off, col, cnt, bits, boff, rec = st.search_many_scored(seqs, k, threshold)
    nk2, nu2, off2, col2, cnt2 = st.search_many(seqs, k, threshold)
    assert np.array_equal(nk, nk2) and np.array_equal(nu, nu2) and np.array_equal(off, off2) and np.array_equal(col, col2) and np.array_equal(cnt, cnt2)
    n_hits = int(off[-1])
    assert int((nk[np.repeat(np.arange(len(seqs)), np.diff(off).astype(np.int64))] > 0).sum()) > 4000 and boff.size == n_hits + 1 and rec.size == n_hits and int(boff[-1]) == bits.size
    # every hit: the scalar scorer on its unpacked string
    text = unpack_presence(bits, boff)
    hit_seq = np.repeat(np.arange(len(seqs)), np.diff(off).astype(np.int64))
    # (a sequence without k-mers: exact search -> no hits; thresholded -> count 0 >= ceil(0 * t) in every column, as in
    # bigsi_hip_search_stream -- the host layer raises the reference's error for it, graph/bigsi.py -- with all-zero records
    # and no bits)
    empty = nk[hit_seq] == 0
    assert (int(empty.sum()) == 0) == (threshold == 1.0)
    assert not rec[empty].tobytes().strip(b"\0") and np.array_equal(boff[:-1][empty], boff[1:][empty])
    with pytest.raises(ZeroDivisionError) if empty.any() else contextlib.nullcontext():
        score_columns(rec, 96)                      # (score.py:99-100 divides by the k-mer count)
    filled = np.where(empty, 1, rec["num_kmers"])
    rec_nz = rec.copy()
    rec_nz["num_kmers"] = filled
    cols = score_columns(rec_nz, 96)
    scalar = Scorer(96)
    cache = {}
    for t in np.flatnonzero(~empty).tolist():
        i = int(hit_seq[t])
        s = text[8 * int(boff[t]):8 * int(boff[t]) + int(nk[i])]
        assert len(s) == int(nk[i])
        if s not in cache:
            cache[s] = scalar.score(s)
        assert {key: c[t] for key, c in zip(SCORE_KEYS, cols)} == cache[s], (t, i)
        assert rec["percent_kmers_found"][t] == round(100 * float(cnt[t]) / int(nu[i]), 2) and rec["num_kmers"][t] == nk[i]
    # a sample of sequences with hits: the same bits and records from a one-sequence batch
    with_hits = np.flatnonzero(np.diff(off).astype(np.int64))
    for i in with_hits[:: max(1, with_hits.size // 60)].tolist():
        b = st.new_batch([seqs[i]], k)
        b.run(threshold)
        o1, c1, n1 = b.hits()
        lo, hi = int(off[i]), int(off[i + 1])
        assert np.array_equal(c1, col[lo:hi]) and np.array_equal(n1, cnt[lo:hi])
        r1, b1, bo1 = b.score_hits(o1, c1, n1, b.unique()[0])
        assert np.array_equal(r1, rec[lo:hi])
        assert np.array_equal(b1[: int(bo1[-1])], bits[int(boff[lo]):int(boff[hi])])
        b.close()
    # the capacity protocol by hand: sizing call -> CAPACITY with complete offsets and byte count -> one retry
    blob, soff = _lib.pack_seqs(seqs)
    o3, need = np.zeros(len(seqs) + 1, np.uint64), np.zeros(1, np.uint64)
    rc = _lib.lib().bigsi_hip_search_stream_scored(st.handle, blob, _lib.ptr(soff), len(seqs), k, threshold, 0, None, None, None, _lib.ptr(o3),
                                                   None, None, 0, None, 0, _lib.ptr(np.zeros(1, np.uint64)), None, _lib.ptr(need))
    assert rc == _lib.ERR_CAPACITY and np.array_equal(o3, off) and int(need[0]) == bits.size
    # hits fit, bits do not (half the bytes): CAPACITY again, lists complete
    c3, n3 = np.zeros(n_hits, np.uint32), np.zeros(n_hits, np.uint32)
    bo3, r3, b3 = np.zeros(n_hits + 1, np.uint64), np.zeros(n_hits, HIT_SCORE_DTYPE), np.zeros(bits.size // 2, np.uint8)
    rc = _lib.lib().bigsi_hip_search_stream_scored(st.handle, blob, _lib.ptr(soff), len(seqs), k, threshold, 0, None, None, None, _lib.ptr(o3),
                                                   _lib.ptr(c3), _lib.ptr(n3), n_hits, _lib.ptr(b3), b3.size, _lib.ptr(bo3), _lib.ptr(r3), _lib.ptr(need))
    assert rc == _lib.ERR_CAPACITY and np.array_equal(c3, col) and np.array_equal(bo3, boff) and int(need[0]) == bits.size
    b3 = np.zeros(bits.size, np.uint8)
    rc = _lib.lib().bigsi_hip_search_stream_scored(st.handle, blob, _lib.ptr(soff), len(seqs), k, threshold, 0, None, None, None, _lib.ptr(o3),
                                                   _lib.ptr(c3), _lib.ptr(n3), n_hits, _lib.ptr(b3), b3.size, _lib.ptr(bo3), _lib.ptr(r3), _lib.ptr(need))
    _lib.check(rc)
    assert np.array_equal(b3, bits) and np.array_equal(r3, rec) and np.array_equal(n3, cnt)
    index.delete()


@pytest.mark.parametrize("threshold", [1.0, 0.25])
def test_search_stream_scored_equals_the_cpu_twins(threshold):
    """The same index (filled through insert_kmers on both sides) and the same 3000 sequences through
    bigsi_hip_search_stream_scored and the CPU twin's bigsi_cpu_search_stream_scored (reference-shaped loops, the scoring header
    compiled for the host): every output array identical."""
    import ctypes as C
    import test_cpu_twin as tw
    assert os.path.exists(tw.LIB), "libbigsi_cpu.so has not been built"
    L = C.CDLL(tw.LIB)
    L.bigsi_cpu_last_error.restype = C.c_char_p
    rng = np.random.default_rng(31)
    k, m, h, n = 31, 60013, 3, 70
    genes = [rand_seq(rng, int(x)) for x in (900, 64 + 30, 128 + 30, 31 + 31, 1500)]
    samples = {}
    for c in range(n):
        g = genes[c % len(genes)]
        lo = int(rng.integers(0, max(len(g) - 50, 1)))
        samples["s%d" % c] = [g[lo:lo + int(rng.integers(31, len(g) + 1))], rand_seq(rng, 200)] + ([g] if c % 6 == 0 else [])
    index = build_index(cfg(k, m, h, max_cols=128), samples)
    twin = tw.Index(L, m, h, 128)
    for c in range(n):
        twin.add_sample(c, samples["s%d" % c], k)
    seqs = [genes[i % len(genes)] if i % 41 == 0 else genes[0][(7 * i) % 700:(7 * i) % 700 + 70] if i % 41 == 1 else rand_seq(rng, int(rng.integers(20, 90)))
            for i in range(3000)]
    got = index.storage.search_many_scored(seqs, k, threshold)
    want = tw.scored_stream(L, twin.ix, seqs, k, threshold)
    assert int(got[2][-1]) > (100 if threshold == 1.0 else 400)
    for name, a, b in zip(("num_kmers", "num_unique", "hit_offsets", "colours", "counts", "bits", "bit_offsets", "records"), got, want):
        assert np.array_equal(a, b), name
    twin.close()
    index.delete()
