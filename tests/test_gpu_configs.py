"""BASELINE.json's configurations at their own workload sizes, through the product path, against the oracle.

configs[1] C2 (1M x 10k, h=3, 1000 x 61-mers): every query, exact and threshold 0.4.
configs[2] C3 (10M x 100k, h=4, 1 kbp): 32 sampled queries of a 256-query batch, AND bitmap + per-sample counts.
configs[3] C4 (25M x 500k over 8 GPUs): the per-GPU shard (25M x 62.5k, h=3, 195 GB), 32 sampled queries.
configs[4] C5 (C4 at threshold 0.4 with scores): the same shard through BIGSI.search_batch(score=True) on >= 64 planted hits
           -- counts, order, presence strings and score fields against the oracle's restatement of graph/bigsi.py:211-239.
The oracle recomputes any row of the seeded synthetic index (oracle/bigsi_oracle.c: orc_synth_row), so nothing of these
sizes is ever held on the host."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20260928
_counter = itertools.count()


def synth_storage(m, n_cols, h, shard=0):
    from bigsi_amd._lib import BigsiHipError
    from bigsi_amd.storage import get_storage
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "cfgtest%d" % next(_counter), "max_cols": n_cols}, "k": 31, "m": m, "h": h}
    st = get_storage(cfg)
    st.delete_all()
    try:
        for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
            st.set_integer(key, v)
        st.fill_synthetic(SEED, shard, 2)
    except BigsiHipError as e:
        st.delete_all()
        pytest.skip("this device cannot hold a %d x %d index: %s" % (m, n_cols, e))
    return cfg, st


def rand_seqs(rng, n, qlen):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[r].tobytes().decode("ascii") for r in rng.integers(0, 4, size=(n, qlen), dtype=np.uint8)]


def check_queries(st, orc, seqs, sample, thresholds=(1.0, 0.4)):
    """exact: hits == set bits of the oracle's AND bitmap; thresholded: colours, counts and (for sampled queries) the whole
    counter vector == unpack_and_sum of the oracle's rows."""
    batch = st.new_batch(seqs, 31)
    for thr in thresholds:
        batch.run(thr)
        nk, nu, mk = batch.unique()
        off, colours, counts = batch.hits()
        for i in sample:
            u, cnt = orc.counts(seqs[i])
            assert nu[i] == u and nk[i] == len(seqs[i]) - 30
            assert mk[i] == int(np.ceil(u * thr))
            want = np.flatnonzero(cnt >= (u if thr == 1.0 else mk[i]))
            lo, hi = int(off[i]), int(off[i + 1])
            assert np.array_equal(colours[lo:hi], want), (thr, i)
            assert np.array_equal(counts[lo:hi], cnt[want].astype(np.uint32)), (thr, i)
            if thr == 1.0:
                _, bm = orc.exact_bitmap(seqs[i])
                assert np.array_equal(batch.bitmap(i), bm), (thr, i)
            else:
                assert np.array_equal(batch.counts(i), cnt.astype(np.uint32)), (thr, i)
    batch.close()


def test_c2_full_workload_every_query():
    from oracle.ref_model import SynthOracle
    m, n, h = 1_000_000, 10_000, 3
    cfg, st = synth_storage(m, n, h)
    orc = SynthOracle(SEED, 0, m, n, h, 31, 2)
    seqs = rand_seqs(np.random.default_rng(1), 1000, 61)
    for j, qi in enumerate(range(0, 1000, 97)):
        col = (1009 * (j + 1)) % n
        part = seqs[qi] if j % 2 == 0 else seqs[qi][:50]          # whole query, or 20 of its 31 k-mers
        st.insert_kmers(col, [part], 31)
        orc.insert_kmers(col, part)
    st.insert_kmers(n - 1, [seqs[3]], 31)                         # last column: the ragged final word
    orc.insert_kmers(n - 1, seqs[3])
    check_queries(st, orc, seqs, range(1000))
    # the route BIGSI.search takes for such a batch (hit lists only): K1 + K2 + K4 in one launch (k_reads_fused)
    batch = st.new_batch(seqs, 31)
    for thr in (1.0, 0.4):
        batch.run(thr, sparse_counts=True)
        _, nu, mk = batch.unique()
        off, colours, counts = batch.hits()
        for i in range(0, 1000, 7):
            u, cnt = orc.counts(seqs[i])
            want = np.flatnonzero(cnt >= (u if thr == 1.0 else mk[i]))
            assert nu[i] == u and np.array_equal(colours[int(off[i]):int(off[i + 1])], want), (thr, i)
            assert np.array_equal(counts[int(off[i]):int(off[i + 1])], cnt[want].astype(np.uint32)), (thr, i)
    batch.close()
    st.delete_all()


def test_c3_full_size_32_sampled_queries():
    from oracle.ref_model import SynthOracle
    m, n, h = 10_000_000, 100_000, 4
    cfg, st = synth_storage(m, n, h)
    orc = SynthOracle(SEED, 0, m, n, h, 31, 2)
    seqs = rand_seqs(np.random.default_rng(1), 256, 1000)
    sample = list(range(0, 256, 8))
    for j, qi in enumerate(sample[:6]):
        col = [99_999, 0, 63, 64, 31_337, 50_000][j]
        part = seqs[qi] if j % 2 == 0 else seqs[qi][:700]
        st.insert_kmers(col, [part], 31)
        orc.insert_kmers(col, part)
    check_queries(st, orc, seqs, sample)
    st.delete_all()


def test_c4_shard_32_sampled_queries():
    """One GPU's shard of BASELINE configs[3]: 25M rows x 62 500 samples (shard 3 of 8), h=3."""
    from oracle.ref_model import SynthOracle
    m, n, h = 25_000_000, 62_500, 3
    cfg, st = synth_storage(m, n, h, shard=3)
    orc = SynthOracle(SEED, 3, m, n, h, 31, 2)
    seqs = rand_seqs(np.random.default_rng(1), 256, 1000)
    sample = list(range(0, 256, 8))
    for j, qi in enumerate(sample[:4]):
        col = [62_499, 0, 4_097, 33_333][j]
        part = seqs[qi] if j % 2 == 0 else seqs[qi][:600]
        st.insert_kmers(col, [part], 31)
        orc.insert_kmers(col, part)
    check_queries(st, orc, seqs, sample)
    st.delete_all()


def test_c5_shard_threshold_and_scores_on_planted_hits():
    """BASELINE configs[4] on the same shard shape: threshold 0.4 with score=True through BIGSI.search_batch.  64 planted
    hits (8 queries x 8 samples, each holding the first 50-100 % of the query's k-mers) plus every random query."""
    from bigsi_amd import BIGSI
    from oracle.ref_model import Scorer, SynthOracle
    m, n, h = 25_000_000, 62_500, 3
    cfg, st = synth_storage(m, n, h)
    for c in range(n):
        st.set_string("metadata:%d" % c, "s%d" % c)
    st.set_integer("metadata:colour_count", n)
    orc = SynthOracle(SEED, 0, m, n, h, 31, 2)
    seqs = rand_seqs(np.random.default_rng(5), 64, 1000)
    planted = {}
    for j in range(8):
        for t in range(8):
            col = (7919 * (8 * j + t) + 11) % n
            part = seqs[j][: 500 + 71 * t]                       # 470 .. 967 of the 970 k-mers
            st.insert_kmers(col, [part], 31)
            orc.insert_kmers(col, part)
            planted.setdefault(j, []).append(col)
    index = BIGSI(cfg)
    got = index.search_batch(seqs, 0.4, score=True)
    scorer = Scorer(n)
    n_hits = 0
    for i, s in enumerate(seqs):
        kmers, uniq, rows = orc.per_kmer_rows(s)
        from oracle import coracle
        cnt = coracle.unpack_and_sum(rows)[:n]
        u = len(uniq)
        want_cols = [int(c) for c in np.flatnonzero(cnt >= int(np.ceil(u * 0.4)))]
        want_cols.sort(key=lambda c: -int(cnt[c]))               # stable: count desc, colour asc (graph/bigsi.py:215-229)
        assert [r["sample_name"] for r in got[i]] == ["s%d" % c for c in want_cols], i
        if i < 8:
            assert set(planted[i]) <= set(want_cols)
        bits = np.unpackbits(rows, axis=1)
        idx = {km: t for t, km in enumerate(uniq)}
        for r, c in zip(got[i], want_cols):
            assert r["num_kmers"] == u and r["num_kmers_found"] == int(cnt[c])
            assert r["percent_kmers_found"] == round(100 * float(cnt[c]) / u, 2)
            col = "".join("1" if bits[idx[km], c] else "0" for km in kmers)
            assert r["kmer-presence"] == col, (i, c)
            want = scorer.score(col)
            for key, v in want.items():
                if key in ("evalue", "pvalue"):
                    assert r[key] == pytest.approx(v, rel=1e-12, abs=2.5e-16)
                else:
                    assert r[key] == v, (i, c, key)
            n_hits += 1
    assert n_hits >= 64
    index.delete()
