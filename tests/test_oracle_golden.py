"""Pins the CPU oracle (oracle/) to golden vectors produced by running the unmodified reference
(tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest

from conftest import GOLDEN, assert_result_equal, check_search, load_golden, unjson
from oracle import coracle
from oracle.ref_model import OracleBIGSI, Scorer, remove_short_ones, seq_to_kmers, tabulate_score


def hex_rows(rows_hex):
    return np.array([list(bytes.fromhex(r)) for r in rows_hex], dtype=np.uint8)


def test_g1_mmh3_and_rows():
    g = load_golden("g1_hash.json")
    for rec in g["mmh3"]:
        for seed, want in zip(g["seeds"], rec["hashes"]):
            assert coracle.mmh3_hash(rec["s"], seed) == want, (rec["s"], seed)
    for rec in g["generate_hashes"]:
        s = rec["s"]
        rows = [coracle.row_of(s, sd, rec["m"]) for sd in range(rec["h"])]
        assert rows == rec["rows_in_seed_order"], rec
        assert sorted(set(rows)) == rec["set"]
    # the reference's own known answers (bigsi/tests/bloom/test_create_bloomfilter.py:6-8)
    assert {coracle.row_of("ATT", s, 25) for s in range(3)} == {2, 15, 17}
    assert {coracle.row_of("ATT", s, 25) for s in range(1)} == {15}
    assert {coracle.row_of("ATT", s, 50) for s in range(2)} == {15, 27}


def test_g1_canonical_and_kmers():
    g = load_golden("g1_hash.json")
    for rec in g["canonical"]:
        assert coracle.reverse_comp(rec["s"]) == rec["reverse_comp"]
        assert coracle.canonical(rec["s"]) == rec["canonical"]
    for rec in g["seq_to_kmers"]:
        assert seq_to_kmers(rec["seq"], rec["k"]) == rec["kmers"]
        first, p2u = coracle.unique_kmers(rec["seq"], rec["k"])
        uniq = list(dict.fromkeys(rec["kmers"]))
        assert [rec["seq"][p:p + rec["k"]] for p in first] == uniq
        assert [uniq[j] for j in p2u] == rec["kmers"]


def test_g2_lookup_and_build():
    for case in load_golden("g2_lookup.json"):
        m, h, k = case["m"], case["h"], case["k"]
        blooms = [OracleBIGSI.bloom(ks, m, h) for ks in case["samples"]]
        for b, want in zip(blooms, case["blooms"]):
            assert b.tobytes().hex() == want
        o = OracleBIGSI.build(blooms, ["s1", "s2"], k, m, h)
        assert [r.tobytes().hex() for r in o.rows] == case["rows"]
        for lk in case["lookups"]:
            assert o.lookup(lk["kmers"], lk["remove_trailing_zeros"]) == lk["result"], lk


@pytest.mark.parametrize("name", ["g3_search.json", "g4_config1.json"])
def test_g3_g4_search(name):
    case = load_golden(name)
    k, m, h = case["k"], case["m"], case["h"]
    names = list(case["samples"].keys())
    o = OracleBIGSI(hex_rows(case["rows"]), names, k, h)
    # also check that the oracle's own bloom/build reproduces the reference's stored rows
    kms = [seq_to_kmers(v, k) if isinstance(v, str) else v for v in case["samples"].values()]
    o2 = OracleBIGSI.build([OracleBIGSI.bloom(x, m, h) for x in kms], names, k, m, h)
    assert np.array_equal(o.rows, o2.rows)
    for s in case["searches"]:
        t = int(s["threshold"]) if s.get("threshold_is_int") else s["threshold"]
        check_search(lambda: o.search(s["seq"], t, s["score"]), s, "%s t=%r score=%r" % (s["seq"][:20], t, s["score"]))
    if "after_delete_a" in case:
        o.names[0] = "D3L3T3D"
        for s in case["after_delete_a"]["searches"]:
            check_search(lambda: o.search(s["seq"], s["threshold"], s["score"]), s, "deleted")


def test_g13_non_ascii_text():
    """k CHARACTERS per k-mer, character-wise canonical form, UTF-8 bytes hashed: the reference run on Greek / accented /
    4-byte text (tests/golden/make_golden.py: g13_unicode)."""
    from oracle.ref_model import canonical_chars, kmer_rows_chars
    g = load_golden("g13_unicode.json")
    k, m, h = g["k"], g["m"], g["h"]
    for rec in g["canonical"]:
        assert canonical_chars(rec["s"]) == rec["canonical"]
        assert kmer_rows_chars(rec["s"], h, m) == rec["rows_in_seed_order"]
    names = list(g["samples"])
    blooms = [OracleBIGSI.bloom(seq_to_kmers(s, k), m, h) for s in g["samples"].values()]
    for b, want in zip(blooms, g["blooms"]):
        assert b.tobytes().hex() == want
    o = OracleBIGSI.build(blooms, names, k, m, h)
    assert [r.tobytes().hex() for r in o.rows] == g["rows"]
    for lk in g["lookups"]:
        assert o.lookup(lk["kmers"], lk["remove_trailing_zeros"]) == lk["lookup"], lk
    for case in g["searches"]:
        check_search(lambda: o.search(case["seq"], case["threshold"], case["score"]), case)


def test_g5_scoring():
    g = load_golden("g5_scoring.json")
    for rec in g["helpers"]["remove_short_ones"]:
        assert remove_short_ones(rec["s"]) == rec["out"], rec["s"]
    for rec in g["helpers"]["tabulate_score"]:
        assert tabulate_score(rec["s"]) == rec["out"], rec["s"]
    for rec in g["cases"]:
        if "raises" in rec:
            with pytest.raises(BaseException) as ei:
                Scorer(rec["db_size"]).score(rec["s"])
            assert type(ei.value).__name__ == rec["raises"]
        else:
            got = Scorer(rec["db_size"]).score(rec["s"])
            assert_result_equal(got, unjson(rec["score"]), "db=%d s=%s" % (rec["db_size"], rec["s"][:24]))


def test_g6_threshold_arithmetic():
    g = load_golden("g6_arith.json")
    for rec in g["min_kmers"]:
        assert math.ceil(rec["n"] * rec["t"]) == rec["min_kmers"]
    for rec in g["percent"]:
        assert round(100 * float(rec["found"]) / rec["n"], 2) == rec["percent"]


def test_g7_random_index():
    g = load_golden("g7_random.json")
    z = np.load(GOLDEN + "/g7_random.npz")
    k, m, h, N = g["k"], g["m"], g["h"], g["n_cols"]
    kms = [seq_to_kmers(a, k) + seq_to_kmers(b, k) for a, b in g["sample_seqs"]]
    o = OracleBIGSI.build([OracleBIGSI.bloom(x, m, h) for x in kms], g["sample_names"], k, m, h)
    assert np.array_equal(o.rows, z["rows"])
    for qi, s in enumerate(g["queries"]):
        u, cnt = o.counts(s)
        assert u == len(set(seq_to_kmers(s, k)))
        assert np.array_equal(cnt, z["counts"][qi][:N])
        assert not z["counts"][qi][N:].any()
    for rec in g["lookups"]:
        got = o.lookup(seq_to_kmers(rec["seq"], k), remove_trailing_zeros=False)
        assert {km: np.packbits(np.array(list(v), dtype="U1") == "1").tobytes().hex() for km, v in got.items()} == rec["lookup"]
    for s in g["searches"]:
        check_search(lambda: o.search(g["queries"][s["q"]], s["threshold"], s["score"]), s, "q%d t=%r" % (s["q"], s["threshold"]))


def test_synthetic_generator_self_consistency():
    # valid-mask and byte order of the shared synthetic generator
    for n_cols in (1, 7, 8, 9, 63, 64, 65, 100, 200, 1000):
        row = coracle.synth_row(1234, 0, 17, n_cols, 1)
        bits = np.unpackbits(row)
        assert not bits[n_cols:].any()
        w0 = coracle.synth_word(1234, 0, 17, 0, 1) & coracle.lib().orc_valid_mask(0, n_cols)
        assert bytes(row[:8]) == int(w0).to_bytes(8, "little")[: len(row[:8])]
    a = coracle.synth_fill(5, 2, 10, 4, 130, 2)
    for r in range(4):
        assert np.array_equal(a[r], coracle.synth_row(5, 2, 10 + r, 130, 2))
    dens = np.unpackbits(coracle.synth_fill(9, 0, 0, 64, 4096, 2)).mean()
    assert 0.23 < dens < 0.27
