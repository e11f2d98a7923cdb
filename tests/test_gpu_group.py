"""The multi-GPU boundary of the C ABI on a one-GPU box.

* device groups (bigsi_hip_group_*, storage-config {"devices": [...]}) with a device listed several times: every shard is a
  real index with its own streams and batch objects, uneven and EMPTY shards included; the exchange runs through shared
  device memory instead of RCCL.  Everything is checked against the oracle on the concatenated index, and the reference's
  own G7 results (goldens) must come out of BIGSI on top of a three-shard group.
* the RCCL calls themselves with the one rank a one-GPU box allows: a one-device group (ncclCommInitAll over [0]) and a
  one-rank communicator (bigsi_hip_comm_init_rank + bigsi_hip_batch_run_sharded): ncclAllGather / ncclAllReduce are really
  issued, on the library's own communicator stream.
The N-rank RCCL run is the driver's scaling bench (bench.py --gpus N), which verifies planted hits on every shard."""
import itertools
import os

import numpy as np
import pytest

from conftest import check_search, load_golden

pytestmark = pytest.mark.gpu
_counter = itertools.count()


def rand_seqs(rng, n, lo, hi):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[rng.integers(0, 4, size=int(rng.integers(lo, hi + 1)))].tobytes().decode("ascii") for _ in range(n)]


def group_storage(m, total_cols, h, devices, seed=77, draws=1):
    from bigsi_amd.storage import get_storage
    cfg = {"storage-engine": "hip-hbm", "k": 31, "m": m, "h": h,
           "storage-config": {"name": "grp%d" % next(_counter), "devices": devices, "max_cols": total_cols}}
    st = get_storage(cfg)
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", total_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(seed, 0, draws)
    return cfg, st


def shard_oracles(st, m, h, seed, draws):
    from oracle.ref_model import SynthOracle
    inf = st.res.info()
    sc, total = int(inf.shard_cols), int(inf.num_cols)
    orcs = []
    for i in range(int(inf.n_shards)):
        n_i = max(0, min(sc, total - i * sc))
        orcs.append(SynthOracle(seed, i, m, n_i, h, 31, draws) if n_i else None)
    return sc, orcs


def whole_counts(orcs, sc, seq):
    """per-sample counts on the concatenated index, indexed by GLOBAL colour (shard * shard_cols + local)."""
    u, out = 0, np.zeros(len(orcs) * sc, np.int64)
    for i, o in enumerate(orcs):
        if o is not None:
            u, c = o.counts(seq)
            out[i * sc: i * sc + c.size] = c
    return u, out


@pytest.mark.parametrize("devices,total", [([0, 0], 200), ([0, 0, 0], 200), ([0, 0], 129), ([0, 0, 0, 0], 1000), ([0], 150)])
def test_group_vs_oracle(devices, total):
    """shard widths: 200/2 -> 128 + 72; 200/3 -> 128 + 72 + 0 (an empty shard); 129/2 -> 128 + 1; 1000/4 -> 256 x 3 + 232;
    a one-device group exchanges through RCCL (one rank)."""
    from bigsi_amd import _lib
    m, h, seed = 20011, 3, 77
    cfg, st = group_storage(m, total, h, devices, seed)
    inf = st.res.info()
    assert inf.n_shards == len(devices) and inf.rccl == (1 if len(devices) == 1 else 0)
    sc, orcs = shard_oracles(st, m, h, seed, 1)
    rng = np.random.default_rng(total)
    seqs = rand_seqs(rng, 9, 31, 400) + ["ACGT" * 20, "N" * 40]
    plant = [0, total - 1, min(total - 1, sc), total // 2]
    for j, c in enumerate(plant):
        st.insert_kmers(c, [seqs[j][: max(31, len(seqs[j]) * (3 if j % 2 else 4) // 4)]], 31)
        orcs[c // sc].insert_kmers(c % sc, seqs[j][: max(31, len(seqs[j]) * (3 if j % 2 else 4) // 4)])
    # storage contract over whole rows: what the device holds == the shards' rows side by side
    ids = np.array([0, 1, m // 2, m - 1], dtype=np.uint64)
    got = st.get_rows_packed(ids, (total + 7) // 8)
    for t, r in enumerate(ids):
        want = np.zeros(len(orcs) * sc // 8, np.uint8)
        for i, o in enumerate(orcs):
            if o is not None:
                row = o.row(int(r))
                want[i * sc // 8: i * sc // 8 + row.size] = row
        assert np.array_equal(got[t], want[: (total + 7) // 8])
    batch = st.new_batch(seqs, 31)
    for thr in (1.0, 0.3, 0.0):
        batch.run(thr)
        nk, nu, mk = batch.unique()
        off, col, cnt = batch.hits()
        for i, s in enumerate(seqs):
            u, wc = whole_counts(orcs, sc, s)
            assert nu[i] == u
            valid = np.zeros(wc.size, bool)
            for g, o in enumerate(orcs):
                if o is not None:
                    valid[g * sc: g * sc + o.n_cols] = True
            want = np.flatnonzero(valid & (wc >= (u if thr == 1.0 else mk[i])))
            lo, hi = int(off[i]), int(off[i + 1])
            assert np.array_equal(col[lo:hi], want), (thr, i, col[lo:hi][:5], want[:5])
            assert np.array_equal(cnt[lo:hi], wc[want].astype(np.uint32)), (thr, i)
        ref_lists = (off.copy(), col.copy(), cnt.copy())
        batch.run(thr, early_exit=True)                  # opt-in early exit on every shard: the same gathered hit lists
        for a, b_ in zip(ref_lists, batch.hits()):
            assert np.array_equal(a, b_), thr
        if thr == 0.3:      # presence strings, each produced on the shard that owns the column
            i = 0
            hits = col[int(off[0]):int(off[1])]
            strs = batch.presence(0, hits, int(nk[0]))
            for c, sgot in zip(hits.tolist(), strs):
                kmers, uniq, rows = orcs[c // sc].per_kmer_rows(seqs[0])
                bits = np.unpackbits(rows, axis=1)[:, c % sc]
                idx = {km: t for t, km in enumerate(uniq)}
                assert sgot == "".join("1" if bits[idx[km]] else "0" for km in kmers)
    batch.close()
    # lookup of explicit k-mers: AND of the h rows, whole-index bytes
    kms = [seqs[0][:31], seqs[1][:31]]
    got = st.lookup_kmers(kms)
    for km in kms:
        want = np.zeros(len(orcs) * sc // 8, np.uint8)
        for i, o in enumerate(orcs):
            if o is not None:
                _, _, rows = o.per_kmer_rows(km)
                want[i * sc // 8: i * sc // 8 + rows.shape[1]] = rows[0]
        assert got[km] == want[: (total + 7) // 8].tobytes()
    # the one-call entry point of the C ABI
    blob, offs = _lib.pack_seqs(seqs)
    nk2, nu2, mk2 = (np.zeros(len(seqs), np.uint32) for _ in range(3))
    hoff = np.zeros(len(seqs) + 1, np.uint64)
    hc, hn = np.zeros(1 << 16, np.uint32), np.zeros(1 << 16, np.uint32)
    _lib.check(_lib.lib().bigsi_hip_group_search_batch(st.handle, blob, _lib.ptr(offs), len(seqs), 31, 0.3, 0, _lib.ptr(nk2), _lib.ptr(nu2),
                                                       _lib.ptr(mk2), _lib.ptr(hoff), _lib.ptr(hc), _lib.ptr(hn), hc.size))
    batch = st.new_batch(seqs, 31)
    batch.run(0.3)
    off, col, cnt = batch.hits()
    assert np.array_equal(hoff, off) and np.array_equal(hc[: int(off[-1])], col) and np.array_equal(hn[: int(off[-1])], cnt)
    batch.close()
    st.delete_all()


@pytest.mark.parametrize("devices", [[0, 0], [0]])
def test_group_hit_lists_regrow(devices):
    """threshold 0 returns every sample of every query: more hits than the shards' gathered hit buffers start with (65536),
    so every shard must grow + rewrite its lists and the per-hit counts must be reduced again over the larger arrays."""
    m, h, total, seed = 5003, 2, 1500, 9
    cfg, st = group_storage(m, total, h, devices, seed)
    sc, orcs = shard_oracles(st, m, h, seed, 1)
    seqs = rand_seqs(np.random.default_rng(3), 64, 40, 70)
    batch = st.new_batch(seqs, 31)
    batch.run(0.0)
    off, col, cnt = batch.hits()
    assert int(off[-1]) == total * len(seqs) > 65536
    for i in (0, 17, 63):
        u, wc = whole_counts(orcs, sc, seqs[i])
        valid = np.concatenate([np.arange(g * sc, g * sc + o.n_cols) for g, o in enumerate(orcs) if o is not None])
        assert np.array_equal(col[int(off[i]):int(off[i + 1])], valid)
        assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], wc[valid].astype(np.uint32))
    batch.run(0.6)          # back to short lists with the large buffers still around
    off, col, cnt = batch.hits()
    _, nu, mk = batch.unique()
    for i in (0, 17, 63):
        u, wc = whole_counts(orcs, sc, seqs[i])
        valid = np.zeros(wc.size, bool)
        for g, o in enumerate(orcs):
            if o is not None:
                valid[g * sc: g * sc + o.n_cols] = True
        want = np.flatnonzero(valid & (wc >= mk[i]))
        assert np.array_equal(col[int(off[i]):int(off[i + 1])], want)
        assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], wc[want].astype(np.uint32))
    batch.close()
    st.delete_all()


def test_bigsi_over_three_shards_reproduces_reference_results():
    """The reference's own answers on the 200-sample G7 index (goldens: searches with and without scores, exceptions,
    lookups) from BIGSI on top of ONE get_storage() call that spreads the index over three shards (128 + 72 + 0 columns)."""
    from bigsi_amd import BIGSI
    g = load_golden("g7_random.json")
    k, m, h = g["k"], g["m"], g["h"]
    cfg = {"storage-engine": "hip-hbm", "k": k, "m": m, "h": h,
           "storage-config": {"name": "grp%d" % next(_counter), "devices": [0, 0, 0], "max_cols": len(g["sample_names"])}}
    b = BIGSI.build_from_sequences(cfg, {nm: list(g["sample_seqs"][i]) for i, nm in enumerate(g["sample_names"])})
    assert b.storage.res.info().n_shards == 3 and b.num_samples == 200
    for rec in g["lookups"]:
        kms = [rec["seq"][i:i + k] for i in range(len(rec["seq"]) - k + 1)]
        got = b.lookup(kms, remove_trailing_zeros=False)
        assert {km: v.tobytes().hex() for km, v in got.items()} == rec["lookup"]
    for s in g["searches"]:
        check_search(lambda: b.search(g["queries"][s["q"]], s["threshold"], s["score"]), s, "q%d t=%r" % (s["q"], s["threshold"]))
    multi = b.search_batch(g["queries"][:10], 0.4)
    for qi in range(10):
        assert multi[qi] == b.search(g["queries"][qi], 0.4)
    # the same index built through the Bloom-filter route (device transpose routed to the owning shards) holds the same rows
    cfg2 = {"storage-engine": "hip-hbm", "k": k, "m": m, "h": h,
            "storage-config": {"name": "grp%d" % next(_counter), "devices": [0, 0, 0], "max_cols": 200}}
    blooms = [BIGSI.bloom({"k": k, "m": m, "h": h, "storage-config": {}}, [a[i:i + k] for i in range(len(a) - k + 1)] + [c[i:i + k] for i in range(len(c) - k + 1)])
              for a, c in g["sample_seqs"]]
    b2 = BIGSI.build(cfg2, blooms, g["sample_names"])
    ids = np.arange(m)
    assert np.array_equal(b2.storage.get_rows_packed(ids), b.storage.get_rows_packed(ids))
    assert b2.search(g["queries"][0], 0.4) == b.search(g["queries"][0], 0.4)
    b2.delete()
    b.delete()


def test_non_ascii_queries_over_three_shards():
    """Golden G13's Greek / accented / 4-byte samples at colours 0, 64, 65 and 129 of a 130-sample index over three shards
    (64 + 64 + 2 columns): lookups and searches (with scores, and the reference's exceptions) equal the oracle, which
    tests/test_oracle_golden.py::test_g13_non_ascii_text pins to the reference's own answers on these strings."""
    from bigsi_amd import BIGSI
    from oracle.ref_model import OracleBIGSI, seq_to_kmers
    g = load_golden("g13_unicode.json")
    k, m, h = g["k"], g["m"], g["h"]
    rng = np.random.default_rng(13)
    seqs = ["".join(rng.choice(list("ACGT"), size=9)) for _ in range(130)]
    for col, s in zip((0, 64, 65, 129), g["samples"].values()):
        seqs[col] = s
    names = ["s%d" % i for i in range(130)]
    cfg = {"storage-engine": "hip-hbm", "k": k, "m": m, "h": h,
           "storage-config": {"name": "grp%d" % next(_counter), "devices": [0, 0, 0], "max_cols": 130}}
    blooms = [BIGSI.bloom(cfg, seq_to_kmers(s, k)) for s in seqs]
    b = BIGSI.build(cfg, blooms, names)
    assert b.storage.res.info().n_shards == 3
    o = OracleBIGSI.build([OracleBIGSI.bloom(seq_to_kmers(s, k), m, h) for s in seqs], names, k, m, h)
    assert np.array_equal(b.storage.get_rows_packed(np.arange(m)), o.rows)
    for lk in g["lookups"]:
        got = b.lookup(lk["kmers"], remove_trailing_zeros=lk["remove_trailing_zeros"])
        assert {x: v.to01() for x, v in got.items()} == o.lookup(lk["kmers"], lk["remove_trailing_zeros"])
    for c in g["searches"]:
        try:
            want = {"results": o.search(c["seq"], c["threshold"], c["score"])}
        except (TypeError, UnboundLocalError, IndexError) as e:
            want = {"raises": type(e).__name__}
        assert ("raises" in want) == ("raises" in c["out"])
        check_search(lambda: b.search(c["seq"], c["threshold"], c["score"]), {"out": want}, "%s t=%r" % (c["seq"], c["threshold"]))
    b.delete()


@pytest.mark.parametrize("threshold", [1.0, 0.3])
def test_one_rank_rccl_communicator(threshold):
    """bigsi_hip_comm_init_rank + bigsi_hip_batch_run_sharded with world = 1: the library's own ncclAllGather (in place) and
    ncclAllReduce on its communicator stream, two batches alternating so that exchanges overlap the next run."""
    from bigsi_amd.parallel import ShardedSearch
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    m, n, h, seed = 30011, 5000, 3, 5
    cfg = {"storage-engine": "hip-hbm", "k": 31, "m": m, "h": h, "storage-config": {"name": "comm%d" % next(_counter), "max_cols": n}}
    st = get_storage(cfg)
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(seed, 0, 1)
    orc = SynthOracle(seed, 0, m, n, h, 31, 1)
    rng = np.random.default_rng(8)
    sets = [rand_seqs(rng, 12, 31, 300) for _ in range(2)]
    st.insert_kmers(4999, [sets[0][0]], 31)
    orc.insert_kmers(4999, sets[0][0])
    sh = ShardedSearch(st, n + 120, force_gather=True)          # a shard width larger than this shard's own columns
    assert sh.exchange == "rccl" and sh.comm_ranks() == (0, 1)
    batches = [st.new_batch(s, 31) for s in sets]
    sh.prepare(batches, threshold == 1.0)
    for _ in range(5):
        sh.step(batches, threshold)
    for b, seqs in zip(batches, sets):
        off, col, cnt = sh.fetch(b)
        _, nu, mk = b.unique()
        for i, s in enumerate(seqs):
            u, c = orc.counts(s)
            want = np.flatnonzero(c >= (u if threshold == 1.0 else mk[i]))
            assert np.array_equal(col[int(off[i]):int(off[i + 1])], want)
            assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], c[want].astype(np.uint32))
    for b in batches:
        b.close()
    sh.close()
    st.delete_all()


def test_group_at_c4_column_geometry_eight_shards_with_scores():
    """BASELINE configs[3] / [4]'s exchange at its REAL column geometry on the one GPU a test box has: 500 000 samples as eight
    shards of 62 528 columns (977 words each; rows cut to 1 M so that the eight shards share one device: 63 GB), through the
    group entry points -- every shard runs K1-K3 on the queries, the masks are exchanged and compacted, counts reduced, and
    with score=True each hit is scored (K5 + K6) on the shard that owns its column.  Exact and threshold 0.4, against the
    oracle's per-shard restatement: global colours, counts, order, presence strings, all 17 score fields."""
    from bigsi_amd import BIGSI
    from oracle import coracle
    from oracle.ref_model import Scorer
    m, total, h, seed = 1_000_000, 500_000, 3, 20260928
    cfg, st = group_storage(m, total, h, [0] * 8, seed, draws=2)
    inf = st.res.info()
    assert inf.n_shards == 8 and inf.shard_cols == 62_528          # ceil(500000 / 8) rounded up to 64 columns
    sc, orcs = shard_oracles(st, m, h, seed, 2)
    assert [o.n_cols for o in orcs] == [62_528] * 7 + [500_000 - 7 * 62_528]
    rng = np.random.default_rng(3)
    seqs = rand_seqs(rng, 24, 1000, 1000)
    planted = {}
    for j in range(6):                                   # each planted query into one sample of EVERY shard, partly
        for g in range(8):
            c = g * sc + (7919 * (8 * j + g) + 11) % orcs[g].n_cols
            part = seqs[j] if g == 7 else seqs[j][: 450 + 70 * g]          # (the last shard's sample holds the whole query: an exact hit)
            st.insert_kmers(c, [part], 31)
            orcs[g].insert_kmers(c % sc, part)
            planted.setdefault(j, []).append(c)
    names = {c: "s%d" % c for cs in planted.values() for c in cs}
    st.set_integer("metadata:colour_count", total)
    index = BIGSI(cfg)
    index.colour_to_sample = lambda c: names.get(int(c), "s%d" % int(c))          # (500 000 metadata records are not what is tested here)
    scorer = Scorer(index.scorer.DB_SIZE)
    for thr in (1.0, 0.4):
        got = index.search_batch(seqs, thr, score=True)
        n_hits = 0
        for i, s in enumerate(seqs):
            u, wc = whole_counts(orcs, sc, s)
            valid = np.zeros(wc.size, bool)
            for g, o in enumerate(orcs):
                valid[g * sc: g * sc + o.n_cols] = True
            want = [int(c) for c in np.flatnonzero(valid & (wc >= (u if thr == 1.0 else int(np.ceil(u * thr)))))]
            if thr != 1.0:
                want.sort(key=lambda c: -int(wc[c]))
            assert [r["sample_name"] for r in got[i]] == ["s%d" % c for c in want], (thr, i)
            if i < 6:
                assert set(planted[i]) <= set(want) if thr != 1.0 else True
            for r, c in zip(got[i], want):
                kmers, uniq, rows = orcs[c // sc].per_kmer_rows(s)
                bits = np.unpackbits(rows, axis=1)[:, c % sc]
                idx = {km: t for t, km in enumerate(uniq)}
                col = "".join("1" if bits[idx[km]] else "0" for km in kmers)
                assert r["kmer-presence"] == col and r["num_kmers"] == u and r["num_kmers_found"] == int(wc[c])
                assert r["percent_kmers_found"] == round(100 * float(wc[c]) / u, 2)
                for key, v in scorer.score(col).items():
                    if key in ("evalue", "pvalue"):
                        assert r[key] == pytest.approx(v, rel=1e-12, abs=2.5e-16)
                    else:
                        assert r[key] == v, (thr, i, c, key)
                n_hits += 1
        assert n_hits >= (6 if thr == 1.0 else 48), (thr, n_hits)
    index.delete()


def test_group_snapshot_round_trip_and_cross_loading(tmp_path):
    """sync() of a multi-GPU index writes whole rows through bigsi_hip_group_save_rows_file (one 2-D copy per shard and chunk) and a
    fresh open loads them through bigsi_hip_group_load_rows_file; the file also loads into a single-GPU index and a single-GPU
    snapshot (rows at its padded 128-byte-multiple pitch) loads into a group: same rows, same answers."""
    from bigsi_amd.storage import get_storage, hip_hbm
    m, total, h, seed = 40_009, 1000, 3, 31
    fn = str(tmp_path / "grp.hbm")
    cfg, st = group_storage(m, total, h, [0, 0, 0], seed)
    sc, orcs = shard_oracles(st, m, h, seed, 1)
    rng = np.random.default_rng(2)
    seqs = rand_seqs(rng, 6, 40, 200)
    for j, c in enumerate((5, sc - 1, sc, 2 * sc + 7, total - 1)):
        st.insert_kmers(c, [seqs[j]], 31)
        orcs[c // sc].insert_kmers(c % sc, seqs[j])
    st.set_string("metadata:5:string", "five")
    ids = np.arange(0, m, 37, dtype=np.uint64)
    st.res.written[:] = True
    rows = np.asarray(st.get_rows_packed(ids)).copy()
    stats = st.save_snapshot(fn)
    assert stats is not None and stats.bytes == m * 128           # 1000 columns: 125 bytes, pitch of 8-byte multiples
    assert open(fn, "rb").read(10) == b"BIGSIHBM2\n"

    def answers(s_):
        batch = s_.new_batch(seqs, 31)
        out = []
        for thr in (1.0, 0.4):
            batch.run(thr)
            off, colours, counts = batch.hits()
            out.append((off.tolist(), colours[: int(off[-1])].tolist(), counts[: int(off[-1])].tolist()))
        batch.close()
        return out
    want = answers(st)
    for i, s in enumerate(seqs):          # ... which are the oracle's
        u, cnt = whole_counts(orcs, sc, s)
        lo, hi = want[0][0][i], want[0][0][i + 1]
        assert want[0][1][lo:hi] == np.flatnonzero(cnt >= u).tolist()
    st.delete_all()
    # a group of another shape, and one GPU, from the same file
    for devs in ([0, 0], [0, 0, 0, 0], None):
        sc_ = {"name": "grpload%d" % next(_counter), "max_cols": total}
        if devs:
            sc_["devices"] = devs
        s2, _ = hip_hbm.HipHbmStorage.load_snapshot(sc_, fn)
        assert s2.get_string("metadata:5:string") == "five" and s2.get_integer("number_of_cols") == total
        assert np.array_equal(np.asarray(s2.get_rows_packed(ids)), rows)
        got = answers(s2)
        assert got == want                 # (a colour is a column of the whole row, whatever the shard width)
        if devs is None:
            fn1 = str(tmp_path / "single.hbm")
            s2.save_snapshot(fn1)          # rows at the single GPU's 128-byte pitch
            s3, _ = hip_hbm.HipHbmStorage.load_snapshot({"name": "grpload%d" % next(_counter), "max_cols": total, "devices": [0, 0, 0]}, fn1)
            assert np.array_equal(np.asarray(s3.get_rows_packed(ids)), rows) and answers(s3) == want
            s3.delete_all()
        s2.delete_all()


def test_group_rows_file_rate(tmp_path):
    """A 2-shard group on one device: 6 GB of whole rows saved and loaded back through the group entry points; the load runs at
    the file <-> HBM rate of the single-GPU route (>= 15 GB/s from page cache), sampled rows identical."""
    import shutil
    from bigsi_amd import _lib
    d = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 16e9 else str(tmp_path)
    fn = os.path.join(d, "bigsi_group_rate_%d.bin" % os.getpid())
    m, total = 480_000, 100_000
    cfg, st = group_storage(m, total, 3, [0, 0], 5, draws=2)
    try:
        rb = 12_504
        ids = np.arange(0, m, 4801, dtype=np.uint64)
        st.res.written[:] = True
        rows = np.asarray(st.get_rows_packed(ids, rb)).copy()
        ss, ls = _lib.IoStats(), _lib.IoStats()
        _lib.check(_lib.lib().bigsi_hip_group_save_rows_file(st.handle, fn.encode(), 4096, 0, m, rb, 0, _lib.C.byref(ss)))
        assert os.path.getsize(fn) == 4096 + m * rb
        _lib.check(_lib.lib().bigsi_hip_group_clear(st.handle))
        assert not np.asarray(st.get_rows_packed(ids[:3], rb)).any()
        _lib.check(_lib.lib().bigsi_hip_group_load_rows_file(st.handle, fn.encode(), 4096, 0, m, rb, 0, _lib.C.byref(ls)))
        assert np.array_equal(np.asarray(st.get_rows_packed(ids, rb)), rows)
        rate = ls.bytes / ls.seconds / 1e9
        print("group rows file: save %.1f GB/s, load %.1f GB/s (file side %.1f GB/s, %d threads, %s)"
              % (ss.bytes / ss.seconds / 1e9, rate, ls.bytes / max(ls.file_seconds, 1e-9) / 1e9, ls.threads, d))
        assert rate >= 15.0, "group load ran at %.1f GB/s" % rate
    finally:
        if os.path.exists(fn):
            os.remove(fn)
        st.delete_all()
