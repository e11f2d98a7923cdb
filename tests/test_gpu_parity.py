"""Parity of the HIP path (through libbigsi_hip.so) with the reference: golden vectors produced by running the
unmodified reference (tests/golden) and the CPU oracle (oracle/) on seeded inputs.  Bit-exact for everything
integer / byte / index; float score fields as stated in conftest.py.  Needs a real MI355X: `pytest -m gpu`."""
import itertools
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_results_equal, check_search, load_golden, unjson

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import bigsi_amd
    from bigsi_amd import _lib
    assert _lib.device_count() >= 1, "no HIP device visible"
    return bigsi_amd


_counter = itertools.count()


def cfg(k, m, h, **sc):
    sc.setdefault("name", "t%d" % next(_counter))
    return {"storage-engine": "hip-hbm", "storage-config": sc, "k": k, "m": m, "h": h}


def rows_hex(b):
    return [bytes(r).hex() for r in b.storage.get_rows_packed(np.arange(b.bloomfilter_size))]


def seq_kmers(s, k):
    return [s[i:i + k] for i in range(len(s) - k + 1)]


# --------------------------------------------------------------------------------------------- hashing / bloom
def test_g1_device_hashing(hip):
    from bigsi_amd.bloom import BloomFilter, generate_hashes
    g = load_golden("g1_hash.json")
    assert generate_hashes("ATT", 3, 25) == {2, 15, 17}          # bigsi/tests/bloom/test_create_bloomfilter.py:6-8
    assert generate_hashes("ATT", 1, 25) == {15}
    assert generate_hashes("ATT", 2, 50) == {15, 27}
    big = 0
    for rec in g["generate_hashes"]:
        if rec["m"] > 1 << 26:                                    # 2^31 and 2^32+15: a 0.25-0.5 GB filter per call
            big += 1
            if big > 6:
                continue
        assert generate_hashes(rec["s"], rec["h"], rec["m"]) == set(rec["set"]), rec
    # canonicalisation on the device: BIGSI.bloom of a k-mer == raw filter of its golden canonical form
    for rec in g["canonical"]:
        c = {"m": 5003, "h": 4, "storage-config": {}}
        a = hip.BIGSI.bloom(c, [rec["s"]])
        b = BloomFilter(5003, 4).update([rec["canonical"]]).bitarray
        assert a == b, rec


def test_g1_row_ids_from_k1(hip):
    """Row ids straight out of K1 (signed hash, Python floor-mod) at real moduli, vs the pinned oracle."""
    from oracle import coracle
    g = load_golden("g1_hash.json")
    for m, h in [(1000, 3), (25000000, 4), (7, 3)]:
        c = cfg(31, m, h, max_cols=64)
        b = hip.BIGSI.build(c, [hip.BIGSI.bloom(c, ["A" * 31])], ["s"])
        seqs = [r["s"] for r in g["canonical"] if len(r["s"]) == 31]
        batch = b.storage.new_batch(seqs, 31)
        batch.run(1.0)
        _, nu, _ = batch.unique()
        for i, s in enumerate(seqs):
            assert nu[i] == 1
            assert [int(x) for x in batch.rows(i, 1)[0]] == coracle.kmer_rows(s, h, m), (s, m)
        batch.close()
        b.delete()


# --------------------------------------------------------------------------------------------- lookup
def test_g2_lookup(hip):
    from bigsi_amd import BitRow
    for case in load_golden("g2_lookup.json"):
        c = cfg(case["k"], case["m"], case["h"])
        blooms = [hip.BIGSI.bloom(c, ks) for ks in case["samples"]]
        for bl, want in zip(blooms, case["blooms"]):
            assert bl.tobytes().hex() == want
        b = hip.BIGSI.build(c, blooms, ["s1", "s2"])
        assert rows_hex(b) == case["rows"]
        for lk in case["lookups"]:
            got = b.lookup(lk["kmers"], remove_trailing_zeros=lk["remove_trailing_zeros"])
            assert {k: v.to01() for k, v in got.items()} == lk["result"], lk
        # the reference's own assertions (bigsi/tests/graph/test_index.py:34-45)
        assert b.lookup(["ATC", "ATC", "ATT", "TTT"]) == {"ATC": BitRow("11"), "ATT": BitRow("10"), "TTT": BitRow("01")}
        b.delete()


# --------------------------------------------------------------------------------------------- search
@pytest.mark.parametrize("name", ["g3_search.json", "g4_config1.json"])
def test_g3_g4_search(hip, name):
    case = load_golden(name)
    k, m, h = case["k"], case["m"], case["h"]
    c = cfg(k, m, h)
    names = list(case["samples"].keys())
    kms = [seq_kmers(v, k) if isinstance(v, str) else v for v in case["samples"].values()]
    blooms = [hip.BIGSI.bloom(c, x) for x in kms]
    for bl, want in zip(blooms, case["blooms"]):
        assert bl.tobytes().hex() == want
    b = hip.BIGSI.build(c, blooms, names)
    assert rows_hex(b) == case["rows"]
    for s in case["searches"]:
        t = int(s["threshold"]) if s.get("threshold_is_int") else s["threshold"]
        check_search(lambda: b.search(s["seq"], t, s["score"]), s, "%s t=%r score=%r" % (s["seq"][:20], t, s["score"]))
    if "after_delete_a" in case:
        d = case["after_delete_a"]
        b.delete_sample("a")
        assert b.num_samples == d["num_samples"]
        assert b.colour_to_sample(0) == d["colour_to_sample_0"]
        assert b.sample_to_colour("a") == d["sample_to_colour_a"]
        for s in d["searches"]:
            check_search(lambda: b.search(s["seq"], s["threshold"], s["score"]), s, "deleted")
    b.delete()


def test_g13_non_ascii_text(hip):
    """Greek / accented / 4-byte characters, as the unmodified reference answered them (golden G13): k CHARACTERS per
    k-mer, canonical form character by character on the host, UTF-8 bytes hashed + rows fetched + combined on the device
    (bigsi_hip_batch_create_elements / bigsi_hip_lookup_raw / BLOOM_RAW)."""
    from bigsi_amd.bloom import generate_hashes
    g = load_golden("g13_unicode.json")
    k, m, h = g["k"], g["m"], g["h"]
    c = cfg(k, m, h)
    for rec in g["canonical"]:
        assert generate_hashes(rec["canonical"], h, m) == set(rec["rows_in_seed_order"]), rec
    names = list(g["samples"])
    blooms = [hip.BIGSI.bloom(c, seq_kmers(s, k)) for s in g["samples"].values()]
    for bl, want in zip(blooms, g["blooms"]):
        assert bl.tobytes().hex() == want
    b = hip.BIGSI.build(c, blooms, names)
    assert rows_hex(b) == g["rows"]
    for lk in g["lookups"]:
        got = b.lookup(lk["kmers"], remove_trailing_zeros=lk["remove_trailing_zeros"])
        assert {x: v.to01() for x, v in got.items()} == lk["lookup"], lk
    for s in g["searches"]:
        check_search(lambda: b.search(s["seq"], s["threshold"], s["score"]), s, "%s t=%r score=%r" % (s["seq"], s["threshold"], s["score"]))
    # a batch mixing ASCII and non-ASCII queries, and the pipelined stream, answer like one search() each
    seqs = [x["seq"] for x in g["searches"] if x["threshold"] == 0.3 and not x["score"] and "results" in x["out"]]
    seqs = list(dict.fromkeys(seqs)) + ["ATACACAAT", "CAAT"]
    one = [b.search(q, 0.3) for q in seqs]
    assert b.search_batch(seqs, 0.3) == one
    assert [r for _, r in b.search_stream(seqs, 0.3, batch_size=2)] == one
    scored = [q for q in seqs if len(q) > k]                     # one k-mer + score=True is the reference's IndexError
    assert b.search_batch(scored, 0.3, score=True) == [b.search(q, 0.3, score=True) for q in scored]
    b.delete()


def test_reference_end_to_end_asserts(hip):
    """The reference's own end-to-end expectations (bigsi/tests/graph/test_end_to_end.py:12-131)."""
    import json
    from bigsi_amd import BitRow
    c = cfg(3, 1000, 3)
    b = hip.BIGSI.build(c, [hip.BIGSI.bloom(c, ["ATC", "ATA"])], ["1"])
    assert (b.kmer_size, b.bloomfilter_size, b.num_hashes, b.num_samples) == (3, 1000, 3, 1)
    assert b.lookup("ATC") == {"ATC": BitRow("1")}
    assert b.colour_to_sample(0) == "1" and b.sample_to_colour("1") == 0
    b.insert(hip.BIGSI.bloom(c, ["ATC", "ATT"]), "2")
    assert b.num_samples == 2
    assert b.lookup(["ATC", "ATA", "ATT"]) == {"ATC": BitRow("11"), "ATA": BitRow("10"), "ATT": BitRow("01")}
    assert b.colour_to_sample(1) == "2" and b.sample_to_colour("2") == 1
    with pytest.raises(ValueError):
        b.insert(hip.BIGSI.bloom(c, ["ATC"]), "1")
    b.delete()
    with pytest.raises(BaseException):
        hip.BIGSI(c)                       # empty store
    k1, k2 = seq_kmers("ATACACAAT", 3), seq_kmers("ATACACAAC", 3)
    b = hip.BIGSI.build(c, [hip.BIGSI.bloom(c, k1), hip.BIGSI.bloom(c, k2)], ["a", "b"])
    assert b.search("ACAGTTAAC", 0.5) == []
    assert b.lookup("AAT") == {"AAT": BitRow("10")}
    res = b.search("ATACACAAT", 0.5)
    assert res[0] == {"percent_kmers_found": 100.0, "num_kmers": 6, "num_kmers_found": 6, "sample_name": "a"}
    assert json.dumps(res[0]) == '{"percent_kmers_found": 100.0, "num_kmers": 6, "num_kmers_found": 6, "sample_name": "a"}'
    assert res[1] == {"percent_kmers_found": 83.33, "num_kmers": 6, "num_kmers_found": 5, "sample_name": "b"}
    b.delete()


def test_g7_random_index(hip):
    g = load_golden("g7_random.json")
    z = np.load(GOLDEN + "/g7_random.npz")
    k, m, h, N = g["k"], g["m"], g["h"], g["n_cols"]
    c = cfg(k, m, h)
    blooms = [hip.BIGSI.bloom(c, seq_kmers(a, k) + seq_kmers(b_, k)) for a, b_ in g["sample_seqs"]]
    b = hip.BIGSI.build(c, blooms, g["sample_names"])
    assert np.array_equal(b.storage.get_rows_packed(np.arange(m)), z["rows"])
    # per-sample counts (unpack_and_sum) and exact bitmaps for all 48 queries in ONE batch
    batch = b.storage.new_batch(g["queries"], k)
    batch.run(0.5)
    nk, nu, mk = batch.unique()
    for qi, s in enumerate(g["queries"]):
        assert nk[qi] == len(s) - k + 1 and nu[qi] == len(set(seq_kmers(s, k)))
        assert np.array_equal(batch.counts(qi), z["counts"][qi][:N].astype(np.uint32))
    batch.run(1.0)
    for qi in range(len(g["queries"])):
        want = np.packbits(z["counts"][qi][: 8 * ((N + 7) // 8)] == nu[qi])
        assert np.array_equal(batch.bitmap(qi), want)
    batch.close()
    for rec in g["lookups"]:
        got = b.lookup(seq_kmers(rec["seq"], k), remove_trailing_zeros=False)
        assert {km: v.tobytes().hex() for km, v in got.items()} == rec["lookup"]
    for s in g["searches"]:
        check_search(lambda: b.search(g["queries"][s["q"]], s["threshold"], s["score"]), s, "q%d t=%r" % (s["q"], s["threshold"]))
    # batched front-end gives the same lists as one-by-one calls
    multi = b.search_batch(g["queries"][:10], 0.4)
    for qi in range(10):
        assert multi[qi] == b.search(g["queries"][qi], 0.4)
    b.delete()


# --------------------------------------------------------------------------------------------- storage contract
def test_g8_storage_contract(hip):
    from bigsi_amd import BitRow
    from bigsi_amd.matrix import BitMatrix
    from bigsi_amd.storage import get_storage
    g = load_golden("g8_storage.json")
    st = get_storage(cfg(3, 25, 1))
    st.delete_all()
    # bigsi/tests/storage/test_storage.py
    st["test"] = b"123"
    assert st["test"] == b"123"
    st.set_integer("test", 112)
    assert st.get_integer("test") == 112
    st.set_string("test", "abc")
    assert st.get_string("test") == "abc"
    ba = BitRow("110101111010")
    st.set_bitarray("test", ba)
    assert st["test:bitarray"].hex() == g["bitarray_bytes"]["stored_hex"]
    assert st.get_bitarray("test")[:12] == ba
    assert st.get_bit("test", 1) is True and st.get_bit("test", 2) is False
    st.set_bit("test", 0, 0)
    assert st.get_bitarray("test").to01() == g["after_set_bit_0_0"]
    assert [st.incr("testinc"), st.incr("testinc")] == g["incr"]
    st.delete_all()
    with pytest.raises(BaseException):
        st.get_string("test")
    # bigsi/tests/matrix/test_bitmatrix.py on device rows
    rows = [BitRow("001"), BitRow("001"), BitRow("111"), BitRow("001"), BitRow("111")] * 5
    bm = BitMatrix.create(st, rows, len(rows), 3)
    bm.set_rows(range(25), rows)
    assert list(bm.get_rows(range(3))) == rows[:3]
    s = g["bitmatrix"]
    assert bm.get_column(0).to01() == s["col0"] and bm.get_column(2).to01() == s["col2"]
    bm.insert_column(BitRow("1" * 25), 0)
    assert bm.get_column(0).to01() == s["col0_after_insert"]
    assert bm.get_row(1).to01() == s["row1_after_insert0"]
    bm.insert_column(BitRow("1" * 25), 3)
    assert bm.get_row(1).to01() == s["row1_after_insert3"] and bm.num_cols == s["num_cols_after"]
    with pytest.raises(KeyError):
        st.get_bitarray(25)            # a row that was never stored
    st.delete_all()
    # insert + merge end to end
    c = cfg(3, 1000, 3)
    b = hip.BIGSI.build(c, [hip.BIGSI.bloom(c, ["ATC", "ATA"])], ["1"])
    b.insert(hip.BIGSI.bloom(c, ["ATC", "ATT"]), "2")
    assert b.num_samples == g["insert"]["num_samples"]
    assert {k: v.to01() for k, v in b.lookup(["ATC", "ATA", "ATT"]).items()} == g["insert"]["lookup"]
    assert rows_hex(b) == g["insert"]["rows"]
    b.delete()
    c1, c2 = cfg(3, 1000, 3), cfg(3, 1000, 3)
    b1 = hip.BIGSI.build(c1, [hip.BIGSI.bloom(c1, seq_kmers("ATACACAAT", 3))], ["a"])
    b2 = hip.BIGSI.build(c2, [hip.BIGSI.bloom(c2, seq_kmers("ATACACAAC", 3))], ["b"])
    b1.merge(b2)
    assert b1.num_samples == g["merge"]["num_samples"]
    assert_results_equal(b1.search("ATACACAAT", 0.5), unjson(g["merge"]["search"]["results"]), "merge")
    assert rows_hex(b1) == g["merge"]["rows"]
    b1.delete()
    b2.delete()


def test_snapshot_roundtrip(hip, tmp_path):
    from bigsi_amd.storage import hip_hbm
    fn = str(tmp_path / "idx.hbm")
    c = cfg(3, 1000, 3, filename=fn)
    k1 = seq_kmers("ATACACAAT", 3)
    b = hip.BIGSI.build(c, [hip.BIGSI.bloom(c, k1)], ["a"])     # build() syncs -> snapshot written
    want = b.search("ATACACAAT", 1.0)
    rows = rows_hex(b)
    hip_hbm._RESIDENT.pop(c["storage-config"]["name"]).free()    # "restart": the resident index is gone
    b2 = hip.BIGSI(c)
    assert rows_hex(b2) == rows and b2.search("ATACACAAT", 1.0) == want
    b2.delete()


def write_v1_snapshot(st, fn, rb):
    """A version-1 snapshot file as rounds 1-3 wrote them (the product only reads them now): JSON header with the written-row bitmap in
    hex, then every row at rb bytes."""
    import struct
    res = st.res
    header = {"kv": {k.decode("latin-1"): v.decode("latin-1") for k, v in res.kv.items()}, "m": res.m, "rb": rb,
              "written": np.packbits(res.written).tobytes().hex(), "uniform_len": res.uniform_len}
    if res.rowlen is not None:
        header["rowlen"] = res.rowlen.tobytes().hex()
    hb = json.dumps(header).encode("utf-8")
    with open(fn, "wb") as f:
        f.write(b"BIGSIHBM1\n" + struct.pack("<Q", len(hb)) + hb)
        ids = np.arange(res.m, dtype=np.uint64)
        out = np.zeros((res.m, rb), np.uint8)
        from bigsi_amd import _lib
        _lib.check(res.fn("get_rows")(res.ix, _lib.ptr(ids), ids.size, _lib.ptr(out), rb))
        f.write(out.tobytes())


def test_snapshot_v2_is_the_device_layout_and_round_trips(hip, tmp_path):
    """sync() writes the device layout (rows at the device pitch behind a 4096-byte aligned header, bitmap of written rows raw) through
    bigsi_hip_save_rows_file; a fresh process' open goes through bigsi_hip_load_rows_file.  Rows never stored stay KeyErrors, rows
    stored shorter keep their length, and a version-1 file (rows at ceil(N/8) bytes, hex bitmap) still loads."""
    import struct
    from bigsi_amd.storage import get_storage, hip_hbm
    fn = str(tmp_path / "v2.hbm")
    c = cfg(31, 5003, 3, filename=fn, max_cols=300, name="snapv2")
    st = get_storage(c)
    st.delete_all()
    st.set_integer("number_of_rows", 5003)
    st.set_integer("number_of_cols", 300)
    rng = np.random.default_rng(11)
    ids = np.sort(rng.choice(5003, size=3000, replace=False)).astype(np.uint64)
    rows = rng.integers(0, 256, size=(3000, 38), dtype=np.uint8)
    rows[:, 37] &= 0xF0                                   # 300 columns: 4 pad bits
    st.res.put_rows(ids, rows)
    st.res.put_rows([int(ids[5])], [bytes(rows[5, :20])])            # one row stored shorter
    st.sync()
    raw = open(fn, "rb").read()
    assert raw.startswith(b"BIGSIHBM2\n")
    (hl,) = struct.unpack("<Q", raw[10:18])
    header = json.loads(raw[18:18 + hl])
    assert header["stride"] == 128 and header["m"] == 5003 and not header["all_written"]
    data_off = len(raw) - 5003 * 128
    assert data_off % 4096 == 0 and raw[data_off + int(ids[0]) * 128: data_off + int(ids[0]) * 128 + 38] == rows[0].tobytes()
    hip_hbm._RESIDENT.pop("snapv2").free()
    st2 = get_storage(c)
    got = st2.get_rows_packed(ids[:5], 38)
    assert np.array_equal(np.asarray(got), rows[:5])
    assert st2.get_bitarray(int(ids[5])).tobytes() == bytes(rows[5, :20])
    missing = sorted(set(range(5003)) - set(ids.tolist()))[0]
    with pytest.raises(KeyError):
        st2.get_bitarray(missing)
    # version 1 of the same index still loads (through the scatter route: its rows are 38 bytes, not 128)
    fn1 = str(tmp_path / "v1.hbm")
    write_v1_snapshot(st2, fn1, 38)
    st2.delete_all()
    c1 = cfg(31, 5003, 3, filename=fn1, max_cols=300, name="snapv1")
    st3 = get_storage(c1)
    assert np.array_equal(np.asarray(st3.get_rows_packed(ids[7:60], 38)), rows[7:60])
    st3.delete_all()


def test_rows_file_io_with_any_row_length(hip, tmp_path):
    """bigsi_hip_save_rows_file / load_rows_file: a row range at the device pitch (no kernel on the way) and at the reference's row length
    (gather / scatter kernels), several pinned buffers' worth, written by one index and read back by another."""
    from bigsi_amd import _lib
    from bigsi_amd.storage import get_storage
    m, n_cols = 70001, 40000
    _, a = synth_index(hip, m, n_cols, 3, 99)
    ref_rows = np.asarray(a.get_rows_packed(np.arange(0, m, 997, dtype=np.uint64)))
    stride, rb = int(a.res.info().row_stride_bytes), int(a.res.info().row_bytes)
    for row_bytes, r0, n in ((stride, 0, m), (rb, 1000, 60000)):
        fn = str(tmp_path / ("rows_%d.bin" % row_bytes))
        st = _lib.IoStats()
        _lib.check(_lib.lib().bigsi_hip_save_rows_file(a.handle, fn.encode(), 4096, r0, n, row_bytes, 3, _lib.C.byref(st)))
        assert st.bytes == n * row_bytes and os.path.getsize(fn) == 4096 + n * row_bytes and st.direct == int(row_bytes == stride)
        c = cfg(31, m, 3, max_cols=n_cols, name="rowsfile%d" % row_bytes)
        b = get_storage(c)
        b.delete_all()
        b.set_integer("number_of_rows", m)
        b.set_integer("number_of_cols", n_cols)
        _lib.check(_lib.lib().bigsi_hip_load_rows_file(b.handle, fn.encode(), 4096, r0, n, row_bytes, 2, None))
        b.res.written[:] = True
        sel = np.arange(0, m, 997, dtype=np.uint64)
        inside = (sel >= r0) & (sel < r0 + n)
        got = np.asarray(b.get_rows_packed(sel))
        assert np.array_equal(got[inside], ref_rows[inside]) and not got[~inside].any()
        rc = _lib.lib().bigsi_hip_load_rows_file(b.handle, fn.encode(), 4096 + row_bytes, r0, n, row_bytes, 2, None)      # one row beyond the end of the file
        assert rc == _lib.ERR_INVALID and b"too short" in _lib.lib().bigsi_hip_last_error()
        assert _lib.lib().bigsi_hip_load_rows_file(b.handle, fn.encode(), 4096, m - 1, 2, row_bytes, 2, None) == _lib.ERR_RANGE
        b.delete_all()
    a.delete_all()


def test_rows_file_striped_over_part_files(hip, tmp_path):
    """A path that ends in '/' is a DIRECTORY of 16 part files over which the rows are striped (stripes of ~4 MB; one inode takes a
    few GB/s of writes, sixteen take the PCIe rate): both row lengths, a range that starts and ends inside stripes, written by one
    index, read back by another -- and by plain file arithmetic from the part files themselves.  sync() uses it for matrices from
    256 MB on."""
    from bigsi_amd import _lib
    from bigsi_amd.storage import get_storage, hip_hbm
    m, n_cols = 250_007, 40000
    _, a = synth_index(hip, m, n_cols, 3, 98)
    sel = np.arange(0, m, 1009, dtype=np.uint64)
    ref_rows = np.asarray(a.get_rows_packed(sel))
    stride, rb = int(a.res.info().row_stride_bytes), int(a.res.info().row_bytes)
    for row_bytes, r0, n in ((stride, 0, m), (rb, 1234, 200_000)):
        d = str(tmp_path / ("striped_%d" % row_bytes)) + "/"
        st = _lib.IoStats()
        _lib.check(_lib.lib().bigsi_hip_save_rows_file(a.handle, d.encode(), 0, r0, n, row_bytes, 0, _lib.C.byref(st)))
        lay = dict(l.split() for l in open(d + "layout").read().splitlines()[1:])
        P, S = int(lay["parts"]), int(lay["stripe_rows"])
        assert P == 16 and S == (4 << 20) // row_bytes and int(lay["row_bytes"]) == row_bytes and int(lay["rows"]) == n
        assert sorted(os.listdir(d)) == ["layout"] + ["part.%03d" % p for p in range(16)]
        assert sum(os.path.getsize(d + "part.%03d" % p) for p in range(16)) == n * row_bytes
        for i in np.flatnonzero((sel >= r0) & (sel < r0 + n))[::7]:       # row r of the range: stripe s = r // S, part s % P, row (s // P) * S + r % S there
            r = int(sel[i]) - r0
            s_ = r // S
            with open(d + "part.%03d" % (s_ % P), "rb") as f:
                f.seek(((s_ // P) * S + r % S) * row_bytes)
                assert f.read(rb) == ref_rows[i].tobytes()
        b = get_storage(cfg(31, m, 3, max_cols=n_cols, name="striped%d" % row_bytes))
        b.delete_all()
        b.set_integer("number_of_rows", m)
        b.set_integer("number_of_cols", n_cols)
        _lib.check(_lib.lib().bigsi_hip_load_rows_file(b.handle, d.encode(), 0, r0, n, row_bytes, 5, None))
        b.res.written[:] = True
        inside = (sel >= r0) & (sel < r0 + n)
        got = np.asarray(b.get_rows_packed(sel))
        assert np.array_equal(got[inside], ref_rows[inside]) and not got[~inside].any()
        assert _lib.lib().bigsi_hip_load_rows_file(b.handle, d.encode(), 8, r0, n, row_bytes, 2, None) == _lib.ERR_INVALID       # not what the layout says
        b.delete_all()
    # the snapshot route: forced to stripe a small matrix
    old = hip_hbm._STRIPE_FROM
    hip_hbm._STRIPE_FROM = 1
    try:
        fn = str(tmp_path / "snap.hbm")
        a.set_string("metadata:0:string", "zero")
        a.res.written[:] = True
        a.save_snapshot(fn)
        assert os.path.isdir(fn + ".d") and os.path.getsize(fn) < 1 << 20
        a.save_snapshot(fn)                                                 # over an existing snapshot: a directory of its own, the old one goes
        assert os.path.isdir(fn + ".1.d") and not os.path.exists(fn + ".d") and not os.path.exists(fn + ".tmp.d")
        # a save that dies between its part files and the header (the one step that switches snapshots is the header's atomic
        # replace) leaves the previous snapshot whole: header and part files still belong together
        real_replace = hip_hbm.os.replace
        hip_hbm.os.replace = lambda *a_: (_ for _ in ()).throw(OSError("simulated crash before the header is replaced"))
        try:
            a.set_string("metadata:0:string", "never committed")
            with pytest.raises(OSError):
                a.save_snapshot(fn)
        finally:
            hip_hbm.os.replace = real_replace
            a.set_string("metadata:0:string", "zero")
        assert os.path.isdir(fn + ".1.d") and hip_hbm._current_data_dir(fn) == os.path.basename(fn) + ".1.d"
        s2, _ = hip_hbm.HipHbmStorage.load_snapshot({"name": "striped-snap", "max_cols": n_cols}, fn)
        assert s2.get_string("metadata:0:string") == "zero" and np.array_equal(np.asarray(s2.get_rows_packed(sel)), ref_rows)
        s3, _ = hip_hbm.HipHbmStorage.load_snapshot({"name": "striped-snap-g", "max_cols": n_cols, "devices": [0, 0]}, fn)      # ... and into two shards
        assert np.array_equal(np.asarray(s3.get_rows_packed(sel)), ref_rows)
        s3.storage_config["filename"] = str(tmp_path / "snapg.hbm")
        s3.sync()
        assert os.path.isdir(s3.storage_config["filename"] + ".d")
        s3.sync()
        assert os.path.isdir(s3.storage_config["filename"] + ".1.d") and not os.path.exists(s3.storage_config["filename"] + ".d")
        s3.delete_all()
        assert not any(os.path.exists(s3.storage_config["filename"] + e) for e in (".d", ".1.d", ""))
        s2.delete_all()
    finally:
        hip_hbm._STRIPE_FROM = old
    a.delete_all()


@pytest.mark.parametrize("n_cols,h", [(333, 2), (5000, 3), (70016, 4)])
def test_one_call_searches_of_one_read_equal_the_batch_route(hip, n_cols, h):
    """ONE read per bigsi_hip_search_batch call: the read kernel's only workgroup writes the caller's block itself and raises the
    flag (no export kernel).  Reads of 31..93 bp, planted in a few samples or not, exact and thresholded -- threshold 0.0 returns
    every sample: more hits than the block carries (1024) and, on the widest index, than the hit buffers hold (65 536), which take
    the fetch route and the regrow route -- against the batch objects; other shapes through the same workspace in between."""
    m = 200003
    _, st = synth_index(hip, m, n_cols, h, 777)
    rng = np.random.default_rng(n_cols + 1)
    reads = ["".join(rng.choice(list("ACGT"), size=L)) for L in (31, 32, 61, 61, 75, 93, 93, 40)]
    reads += [reads[2][:45] + "N" + reads[2][46:], reads[3].lower(), reads[4][:35] + reads[4][:35]]
    for i, q in enumerate(reads[:6]):
        for c in rng.choice(n_cols, size=4, replace=False):
            st.insert_kmers(int(c), [q if i % 2 == 0 else q[:50]], 31)

    def general(seqs, thr):
        b = st.new_batch(seqs, 31)
        b.run(thr, sparse_counts=thr < 1.0)
        nk, nu, _ = b.unique()
        off, col, cnt = b.hits()
        b.close()
        return [(int(nk[i]), int(nu[i]), col[int(off[i]):int(off[i + 1])].tolist(), cnt[int(off[i]):int(off[i + 1])].tolist()) for i in range(len(seqs))]

    found = 0
    for rounds in range(2):
        for thr in (1.0, 0.5, 0.0):
            for i, q in enumerate(reads):
                (k_, u_, col, cnt), = st.search_batch([q], 31, thr)
                want = general([q], thr)[0]
                assert (k_, u_, col.tolist(), cnt.tolist()) == want, (thr, i, len(q))
                found += len(want[2]) if thr == 1.0 else 0
                if i % 4 == 3:
                    got = st.search_batch(reads[:5], 31, thr)
                    assert [(a, b_, c.tolist(), d.tolist()) for a, b_, c, d in got] == general(reads[:5], thr), (thr, i)
    assert found >= 2 * 3 * 4
    st.delete_all()


@pytest.mark.parametrize("n_cols,h", [(333, 2), (5000, 3), (70016, 4)])
def test_one_call_searches_of_one_query_equal_the_batch_route(hip, n_cols, h):
    """bigsi_hip_search_batch -- the serving call: staged input read by K1 in place (zero copy), no completion event, the export
    kernel's flag -- for ONE query at a time: same numbers and hit lists as the batch objects (create / run / fetch) for lengths on both
    sides of the K1 routes' limits, repeated k-mers, N and lowercase, planted and absent queries, calls of other shapes in between
    (pairs, thresholded, batches of reads cut from the query: other bytes in the same staging area every call), and a query with
    more hits than the hit buffers hold."""
    m = 200003
    _, st = synth_index(hip, m, n_cols, h, 4242)
    rng = np.random.default_rng(n_cols)
    qs = random_seqs(rng, 6, 94, 1500) + ["".join(rng.choice(list("ACGT"), size=L)) for L in (93, 94, 95, 96, 97, 156, 157, 4062, 4063, 1000, 1023, 1024, 1025, 3070, 3072, 3073)]      # (also around the sizes a query travels in the kernel arguments)
    rep = "".join(rng.choice(list("ACGT"), size=70))
    qs += [rep * 9, qs[0][:300] + "N" + qs[0][300:], qs[1].lower(), qs[2][:200] + qs[2][:200]]
    for i, q in enumerate(qs[:8]):
        for c in rng.choice(n_cols, size=3, replace=False):
            st.insert_kmers(int(c), [q], 31)

    def general(seqs):
        b = st.new_batch(seqs, 31)
        b.run(1.0)
        nk, nu, _ = b.unique()
        off, col, cnt = b.hits()
        b.close()
        return [(int(nk[i]), int(nu[i]), col[int(off[i]):int(off[i + 1])].tolist(), cnt[int(off[i]):int(off[i + 1])].tolist()) for i in range(len(seqs))]

    for rounds in range(2):
        for i, q in enumerate(qs):
            (k_, u_, col, cnt), = st.search_batch([q], 31, 1.0)
            assert (k_, u_, col.tolist(), cnt.tolist()) == general([q])[0], (i, len(q))
            if i % 5 == 4:          # other shapes through the same workspace: two queries, thresholded, reads
                pair = st.search_batch([qs[0], qs[3]], 31, 1.0)
                assert [(a, b_, c.tolist(), d.tolist()) for a, b_, c, d in pair] == general([qs[0], qs[3]])
                st.search_batch([q], 31, 0.5)
                # reads through the same staging area, other bytes every call (the kernels read the pinned input in place: a stale
                # cached copy of an earlier call's bytes would show here)
                reads = [q[j:j + 61] for j in range(0, 300, 7) if len(q[j:j + 61]) >= 40]
                got = st.search_batch(reads, 31, 1.0)
                assert [(a, b_, c.tolist(), d.tolist()) for a, b_, c, d in got] == general(reads), i
    assert sum(len(general([q])[0][2]) for q in qs[:8]) >= 24           # the planted samples are found
    # more hits than the hit buffers hold (65 536): every row of a query set to all ones -> every column hits
    if n_cols > 65536:
        q = qs[3]
        b = st.new_batch([q], 31)
        b.run(1.0)
        _, nu, _ = b.unique()
        rows = np.unique(b.rows(0, nu[0]))
        b.close()
        full = np.full((rows.size, (n_cols + 7) // 8), 0xFF, np.uint8)
        if n_cols % 8:
            full[:, -1] = (0xFF << (8 - n_cols % 8)) & 0xFF
        st.res.put_rows(rows.astype(np.uint64), full)
        (k_, u_, col, cnt), = st.search_batch([q], 31, 1.0)
        assert col.tolist() == list(range(n_cols)) and (cnt == u_).all() and u_ == int(nu[0])
        (k2, u2, col2, cnt2), = st.search_batch([qs[4]], 31, 1.0)           # and the workspace works again afterwards
        assert (k2, u2, col2.tolist(), cnt2.tolist()) == general([qs[4]])[0]
    st.delete_all()


# --------------------------------------------------------------------------------------------- synthetic index vs oracle
def synth_index(hip, m, n_cols, h, seed, shard=0, draws=2):
    from bigsi_amd.storage import get_storage
    c = cfg(31, m, h, max_cols=n_cols)
    st = get_storage(c)
    st.delete_all()
    st.set_integer("number_of_rows", m)
    st.set_integer("number_of_cols", n_cols)
    st.set_integer("ksi:bloomfilter_size", m)
    st.set_integer("ksi:num_hashes", h)
    st.fill_synthetic(seed, shard, draws)
    return c, st


def random_seqs(rng, n, lo, hi):
    return ["".join(rng.choice(list("ACGT"), size=int(rng.integers(lo, hi + 1)))) for _ in range(n)]


@pytest.mark.parametrize("m,n_cols,h,maxlen", [
    (5003, 1000, 3, 90),          # P=6  (< 64 k-mers), ragged last word, 2 waves
    (20011, 9999, 4, 700),        # P=10, C2-like row width (157 words)
    (3001, 70, 1, 1500),          # P=16, single hash, two words
    (2003, 130, 2, 400),          # h=2
    (2003, 64, 5, 200),           # h=5, exactly one word
    (2003, 65, 7, 100),           # runtime-h kernel (h > 5)
    (50021, 40000, 3, 300),       # 625 words: two 256-thread tiles
])
def test_synthetic_vs_oracle(hip, m, n_cols, h, maxlen):
    from oracle import coracle
    from oracle.ref_model import SynthOracle
    seed = 20260928 + m
    c, st = synth_index(hip, m, n_cols, h, seed)
    orc = SynthOracle(seed, 0, m, n_cols, h, 31, 2)
    # the fill kernel against the CPU generator, sampled rows + first/last
    ids = np.unique(np.concatenate([[0, m - 1], np.random.default_rng(1).integers(0, m, 40)])).astype(np.uint64)
    got = st.get_rows_packed(ids)
    for i, r in enumerate(ids):
        assert np.array_equal(got[i], coracle.synth_row(seed, 0, int(r), n_cols, 2)), r
    rng = np.random.default_rng(m)
    seqs = random_seqs(rng, 10, 31, maxlen) + ["A" * 40, "ACGT" * 20, "N" * 31, "ACGTN" * 12, "acgt" * 10, "AC", ""]
    # plant three of them into a few samples (Bloom-add on the transposed matrix)
    plants = [(0, seqs[0]), (n_cols - 1, seqs[0]), (n_cols // 2, seqs[1]), (min(63, n_cols - 1), seqs[2][:60])]
    for col, s in plants:
        st.insert_kmers(col, [s], 31)
        orc.insert_kmers(col, s)
    batch = st.new_batch(seqs, 31)
    for thr in (0.35, 1.0):
        batch.run(thr, force_counts=(thr == 1.0 and h == 2))
        nk, nu, mk = batch.unique()
        off, colours, counts = batch.hits()
        inf = batch.info()
        for i, s in enumerate(seqs):
            u, cnt = orc.counts(s)
            assert nu[i] == u and nk[i] == max(len(s) - 30, 0)
            assert mk[i] == int(np.ceil(u * thr))
            lo, hi = int(off[i]), int(off[i + 1])
            if inf.exact:
                _, bm = orc.exact_bitmap(s) if u else (0, np.zeros(orc.rb, np.uint8))
                assert np.array_equal(batch.bitmap(i), bm)
                want = np.flatnonzero(np.unpackbits(bm)[:n_cols]) if u else np.zeros(0, int)
                assert np.array_equal(colours[lo:hi], want) and (counts[lo:hi] == u).all()
            else:
                assert np.array_equal(batch.counts(i), cnt.astype(np.uint32))
                want = np.flatnonzero(cnt >= mk[i])
                assert np.array_equal(colours[lo:hi], want)
                assert np.array_equal(counts[lo:hi], cnt[want].astype(np.uint32))
    # planted sequences are exact hits of their samples
    batch.run(1.0)
    off, colours, counts = batch.hits()
    assert {0, n_cols - 1} <= set(colours[int(off[0]):int(off[1])].tolist())
    assert n_cols // 2 in set(colours[int(off[1]):int(off[2])].tolist())
    # lookup rows and presence strings of one query against the oracle
    kmers, uniq, rows = orc.per_kmer_rows(seqs[0])
    first, got_rows = batch.lookup(0, len(uniq))
    assert [seqs[0][p:p + 31] for p in first] == uniq
    assert np.array_equal(got_rows, rows)
    cols = np.array([0, n_cols - 1, n_cols // 3], dtype=np.uint32)
    pres = batch.presence(0, cols, len(kmers))
    bits = np.unpackbits(rows, axis=1)
    idx = {km: j for j, km in enumerate(uniq)}
    for cc, p in zip(cols, pres):
        assert p == "".join("1" if bits[idx[km], cc] else "0" for km in kmers)
    batch.close()
    st.delete_all()


def test_long_query_uint32_counters(hip):
    """> 65535 k-mers in one query: P=32 planes, uint32 counters."""
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 1009, 100, 2
    c, st = synth_index(hip, m, n_cols, h, 77, draws=1)
    orc = SynthOracle(77, 0, m, n_cols, h, 31, 1)
    rng = np.random.default_rng(3)
    s = "".join(rng.choice(list("ACGT"), size=66000))
    batch = st.new_batch([s, "ACGT" * 10], 31)
    batch.run(0.5)
    assert batch.info().count_bytes == 4
    for i, q in enumerate([s, "ACGT" * 10]):
        u, cnt = orc.counts(q)
        assert batch.unique()[1][i] == u
        assert np.array_equal(batch.counts(i), cnt.astype(np.uint32))
    batch.close()
    st.delete_all()


def test_size_independent_properties(hip):
    """Properties that hold at any size: strand symmetry, idempotence under repetition, exact == (count == u),
    planted round trip."""
    m, n_cols, h = 200003, 30000, 4
    c, st = synth_index(hip, m, n_cols, h, 5)
    rng = np.random.default_rng(9)
    base = random_seqs(rng, 6, 200, 400)
    comp = str.maketrans("ACGT", "TGCA")
    seqs = base + [s[::-1].translate(comp) for s in base]
    for i, s in enumerate(base):
        st.insert_kmers(100 * i + 7, [s], 31)
    batch = st.new_batch(seqs, 31)
    batch.run(0.5)
    _, nu, _ = batch.unique()
    cnts = [batch.counts(i) for i in range(len(seqs))]
    for i in range(len(base)):
        assert nu[i] == nu[i + len(base)]
        assert np.array_equal(cnts[i], cnts[i + len(base)])          # reverse complement: same canonical k-mers
        assert cnts[i][100 * i + 7] == nu[i]                          # planted sample holds every k-mer
    batch.run(1.0)
    off, colours, _ = batch.hits()
    for i in range(len(seqs)):
        assert np.array_equal(colours[int(off[i]):int(off[i + 1])], np.flatnonzero(cnts[i] == nu[i]))
    batch.close()
    # tandem repeat s+s: the unique k-mers of s plus at most 30 junction windows -> counts move by at most 30
    rep = st.new_batch([s + s for s in base], 31)
    rep.run(0.5)
    _, nu2, _ = rep.unique()
    for i in range(len(base)):
        d = rep.counts(i).astype(np.int64) - cnts[i].astype(np.int64)
        assert nu[i] <= nu2[i] <= nu[i] + 30 and d.min() >= 0 and d.max() <= 30
    rep.close()
    st.delete_all()


def test_gathered_compaction_two_shards_on_one_gpu(hip):
    """Everything of the multi-GPU exchange except the RCCL calls: two column shards run one after the other on this GPU,
    each writing its per-sample bit vector (exact AND bitmap / thresholded hit mask) into its slot of a [shard][seq][stride]
    buffer (bigsi_hip_batch_set_outputs); then, as every rank would, each shard's batch compacts the whole buffer
    (compact_gathered / compact_gathered_masks with its own shard id) and the per-hit counts are summed over "ranks".
    Hits must equal the oracle's on the concatenated index.  Also the dense variant (gathered uint16 counters)."""
    import torch
    from bigsi_amd import _lib
    from oracle.ref_model import SynthOracle
    m, shard_cols, h, world = 6007, 1000, 3, 2
    rng = np.random.default_rng(11)
    seqs = random_seqs(rng, 6, 31, 300) + ["ACGT" * 30]
    shards, orcs = [], []
    for g in range(world):
        c, st = synth_index(hip, m, shard_cols, h, 321, shard=g, draws=1)
        orc = SynthOracle(321, g, m, shard_cols, h, 31, 1)
        st.insert_kmers(5 + 3 * g, [seqs[0]], 31)
        orc.insert_kmers(5 + 3 * g, seqs[0])
        shards.append(st)
        orcs.append(orc)
    wv_pad = (-(-shard_cols // 64) + 1) // 2 * 2
    L = _lib.lib()

    def fetch(b):
        off = np.zeros(len(seqs) + 1, np.uint64)
        col = np.zeros(1 << 16, np.uint32)
        cnt = np.zeros(1 << 16, np.uint32)
        _lib.check(L.bigsi_hip_batch_fetch_gathered_hits(b.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), col.size))
        return off, col[: int(off[-1])], cnt[: int(off[-1])]

    for thr, dense in ((1.0, False), (0.3, False), (0.3, True)):
        exact = thr == 1.0
        stride = wv_pad * 64 * 2 if dense else wv_pad * 8
        buf = torch.zeros((world, len(seqs) * stride), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        batches = []
        for g, st in enumerate(shards):
            b = st.new_batch(seqs, 31)
            slot = buf[g].data_ptr()
            _lib.check(L.bigsi_hip_batch_set_outputs(b.b, None if dense else slot, slot if dense else None))
            b.run(thr, skip_compact=True, sparse_counts=not dense)
            _lib.check(L.bigsi_hip_synchronize(st.handle))
            batches.append(b)
        per_rank = []
        for g, b in enumerate(batches):
            if exact or dense:
                _lib.check(L.bigsi_hip_batch_compact_gathered(b.b, buf.data_ptr(), world, shard_cols))
            else:
                _lib.check(L.bigsi_hip_batch_compact_gathered_masks(b.b, buf.data_ptr(), world, shard_cols, g))
            per_rank.append(fetch(b))
        off, col = per_rank[0][0], per_rank[0][1]
        for r in per_rank[1:]:
            assert np.array_equal(r[0], off) and np.array_equal(r[1], col)        # every rank derives the same lists
        if exact or dense:
            cnt = per_rank[0][2]
            assert all(np.array_equal(r[2], cnt) for r in per_rank)
        else:
            cnt = sum(r[2].astype(np.uint64) for r in per_rank).astype(np.uint32)  # the all-reduce
            for g, r in enumerate(per_rank):                                       # each rank only knows its own shard's counts
                other = (col // shard_cols) != g
                assert not r[2][other].any()
        _, nu, mk = batches[0].unique()
        for i, s in enumerate(seqs):
            want_cnt = np.concatenate([o.counts(s)[1] for o in orcs])
            want = np.flatnonzero(want_cnt >= (nu[i] if exact else mk[i]))
            lo, hi = int(off[i]), int(off[i + 1])
            assert np.array_equal(col[lo:hi], want), (thr, dense, i)
            assert np.array_equal(cnt[lo:hi], want_cnt[want].astype(np.uint32)), (thr, dense, i)
        # a shard's own (local) hit list is still available after a skip_compact run
        loff, lcol, lcnt = batches[1].hits()
        c1 = orcs[1].counts(seqs[0])[1]
        want1 = np.flatnonzero(c1 >= (nu[0] if exact else mk[0]))
        assert np.array_equal(lcol[int(loff[0]):int(loff[1])], want1)
        assert np.array_equal(lcnt[int(loff[0]):int(loff[1])], c1[want1].astype(np.uint32))
        for b in batches:
            b.close()
    for st in shards:
        st.delete_all()


def test_full_size_c3_properties(hip):
    """BASELINE configs[2] at full size (10M rows x 100k samples, h=4, 1 kbp queries; 125 GB of HBM): size-independent
    properties + one query recomputed by the oracle from the synthetic generator."""
    from bigsi_amd._lib import BigsiHipError
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 10_000_000, 100_000, 4
    try:
        c, st = synth_index(hip, m, n_cols, h, 20260928)
    except BigsiHipError as e:
        pytest.skip("cannot hold the 125 GB index on this device: %s" % e)
    rng = np.random.default_rng(1)
    base = random_seqs(rng, 8, 1000, 1000)
    comp = str.maketrans("ACGT", "TGCA")
    seqs = base + [s[::-1].translate(comp) for s in base[:4]]
    st.insert_kmers(99_999, [base[0]], 31)      # last column: the ragged final word
    st.insert_kmers(31_337, [base[1]], 31)
    batch = st.new_batch(seqs, 31)
    batch.run(0.4)
    _, nu, mk = batch.unique()
    assert (nu[:8] == 970).all() and (mk == 388).all()
    cnts = [batch.counts(i) for i in range(len(seqs))]
    off, colours, counts = batch.hits()
    assert colours[int(off[0]):int(off[1])].tolist() == [99_999] and counts[int(off[0])] == 970
    assert colours[int(off[1]):int(off[2])].tolist() == [31_337]
    for i in range(4):
        assert np.array_equal(cnts[i], cnts[8 + i])                     # strand symmetry
    for i in range(2, 8):
        assert off[i + 1] == off[i] and cnts[i].max() < 388            # random queries: no sample near the threshold
    batch.run(1.0)
    off, colours, _ = batch.hits()
    for i in range(len(seqs)):
        assert np.array_equal(colours[int(off[i]):int(off[i + 1])], np.flatnonzero(cnts[i] == nu[i]))
    orc = SynthOracle(20260928, 0, m, n_cols, h, 31, 2)
    orc.insert_kmers(99_999, base[0])
    u, cnt = orc.counts(base[0])
    assert u == 970 and np.array_equal(cnts[0], cnt.astype(np.uint32))
    batch.close()
    st.delete_all()


def test_many_hits_regrow_and_threshold_zero(hip):
    """threshold 0 returns EVERY sample (count >= 0), far beyond the device hit buffers' initial capacity (65536): the
    write pass must be re-run after growing them, and the lists must still be complete and ordered."""
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 5003, 40000, 2
    c, st = synth_index(hip, m, n_cols, h, 8, draws=1)
    orc = SynthOracle(8, 0, m, n_cols, h, 31, 1)
    seqs = random_seqs(np.random.default_rng(2), 5, 40, 60)
    batch = st.new_batch(seqs, 31)
    batch.run(0.0)
    off, colours, counts = batch.hits()
    assert int(off[-1]) == n_cols * len(seqs)
    for i, s in enumerate(seqs):
        _, cnt = orc.counts(s)
        assert np.array_equal(colours[int(off[i]):int(off[i + 1])], np.arange(n_cols))
        assert np.array_equal(counts[int(off[i]):int(off[i + 1])], cnt.astype(np.uint32))
    batch.run(0.5)          # and back to a small list with the big buffers still around
    off, colours, counts = batch.hits()
    for i, s in enumerate(seqs):
        u, cnt = orc.counts(s)
        assert np.array_equal(colours[int(off[i]):int(off[i + 1])], np.flatnonzero(cnt >= int(np.ceil(u * 0.5))))
    batch.close()
    st.delete_all()


@pytest.mark.parametrize("k", [1, 4, 32, 33, 45, 64])
def test_other_kmer_sizes_vs_oracle(hip, k):
    """run-time-k kernels (everything except the k=31 specialisation), including k > 32 and the 4-byte-block/tail
    boundaries of MurmurHash3."""
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 3001, 300, 3
    c = cfg(k, m, h, max_cols=n_cols)
    st = get_storage(c)
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(k, 0, 1)
    orc = SynthOracle(k, 0, m, n_cols, h, k, 1)
    seqs = random_seqs(np.random.default_rng(k), 6, max(k, 2), 150) + ["ACGTN" * 30, "A" * (k + 5)]
    st.insert_kmers(3, [seqs[0]], k)
    orc.insert_kmers(3, seqs[0])
    batch = st.new_batch(seqs, k)
    batch.run(0.6)
    _, nu, mk = batch.unique()
    for i, s in enumerate(seqs):
        u, cnt = orc.counts(s)
        assert nu[i] == u and np.array_equal(batch.counts(i), cnt.astype(np.uint32)), (k, i)
    batch.close()
    st.delete_all()


def test_device_build_paths(hip):
    """Three ways to the same matrix: (1) Bloom filters + device transpose (BIGSI.build), (2) straight from sequences
    (build_from_sequences), (3) the golden rows the reference stored; plus unaligned multi-column inserts and the
    device-to-device merge at column offsets that are not multiples of 8."""
    from bigsi_amd.storage import get_storage
    g = load_golden("g7_random.json")
    z = np.load(GOLDEN + "/g7_random.npz")
    k, m, h, N = g["k"], g["m"], g["h"], g["n_cols"]
    c2 = cfg(k, m, h)
    b2 = hip.BIGSI.build_from_sequences(c2, {n: [a, b_] for n, (a, b_) in zip(g["sample_names"], g["sample_seqs"])})
    assert np.array_equal(b2.storage.get_rows_packed(np.arange(m)), z["rows"])
    assert b2.num_samples == N and b2.colour_to_sample(N - 1) == g["sample_names"][-1]
    s0 = g["searches"][0]
    check_search(lambda: b2.search(g["queries"][s0["q"]], s0["threshold"], s0["score"]), s0, "from_sequences")
    # columns inserted in odd-sized groups at odd offsets == one-shot build
    blooms = np.stack([np.frombuffer(hip.BIGSI.bloom(c2, seq_kmers(a, k) + seq_kmers(b_, k)).tobytes(), np.uint8)
                       for a, b_ in g["sample_seqs"]])
    st = get_storage(cfg(k, m, h, max_cols=8))       # tiny capacity: forces re-striding while columns arrive
    st.delete_all()
    st.set_integer("number_of_rows", m)
    st.set_integer("number_of_cols", 0)
    c0 = 0
    for n in (1, 7, 64, 3, 61, 64):
        st.insert_columns(c0, blooms[c0:c0 + n])
        c0 += n
    assert c0 == N and int(st.res.info().num_cols) == N
    assert np.array_equal(st.get_rows_packed(np.arange(m)), z["rows"])
    st.insert_columns(5, blooms[100:103])          # overwrite in the middle
    want = np.unpackbits(z["rows"], axis=1)
    want[:, 5:8] = np.unpackbits(blooms[100:103], axis=1)[:, :m].T
    assert np.array_equal(st.get_rows_packed(np.arange(m)), np.packbits(want, axis=1))
    st.delete_all()
    # merge: 13 + 200 + 5 columns
    parts = []
    for lo, hi in ((0, 13), (13, 200), (195, 200)):
        cc = cfg(k, m, h)
        parts.append(hip.BIGSI.build(cc, [hip.BitRow.frombytes(blooms[i].tobytes(), m) for i in range(lo, hi)],
                                     ["p%d_%d" % (lo, i) for i in range(lo, hi)]))
    parts[0].merge(parts[1])
    parts[0].merge(parts[2])
    cols = list(range(0, 200)) + list(range(195, 200))
    want = np.packbits(np.unpackbits(z["rows"], axis=1)[:, cols], axis=1)
    assert parts[0].num_samples == 205 and parts[0].bitmatrix.num_cols == 205
    assert np.array_equal(parts[0].storage.get_rows_packed(np.arange(m)), want)
    assert parts[0].colour_to_sample(204) == "p195_199"
    for p in parts:
        p.delete()
    b2.delete()


@pytest.mark.parametrize("k,lens", [(31, [31, 61, 500, 4126]), (31, [4127, 100]), (5, [5, 4100, 9, 4]), (33, [33, 2000, 40]),
                                    (31, [31, 61, 94, 40, 30, 93]), (4, [4, 10, 67, 5, 3, 66]), (1, [64, 1, 2]),
                                    # a handful of queries of 256 .. 1024 positions: K1 runs as 8 workgroups per query (dedupe replicated,
                                    # hashing shared out) -- and on both sides of that window
                                    (31, [1000, 500, 286]), (31, [1054]), (31, [1055]), (31, [286] + [900] * 31), (31, [285, 100]),
                                    (5, [900, 260, 9]), (33, [1054, 300, 33]), (21, [1044, 1044, 20])])
def test_k1_fused_lds_equals_global_path(hip, k, lens):
    """K1 has three routes -- one wavefront per query (all queries <= 64 positions), one fused launch with the dedupe table
    in LDS (all queries <= 4096 positions) and the four-kernel global-table route (longer queries, or forced by
    BIGSI_RUN_K1_GLOBAL).  Same row ids, unique counts,
    position->unique maps (seen through presence strings) and hits from both, and both equal the oracle."""
    from bigsi_amd.storage import get_storage
    from oracle import coracle
    m, n_cols, h = 2503, 128, 3
    st = get_storage(cfg(k, m, h, max_cols=n_cols))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(3, 0, 1)
    rng = np.random.default_rng(sum(lens) + k)
    seqs = ["".join(rng.choice(list("ACGT" if i % 2 else "AC"), size=L)) for i, L in enumerate(lens)]   # "AC" strings: many duplicate k-mers
    batch = st.new_batch(seqs, k)
    outs = []
    for forced in (False, True):
        batch.run(0.5, k1_global=forced)
        nk, nu, mk = batch.unique()
        off, col, cnt = batch.hits()
        rows = [batch.rows(i, nu[i]).copy() for i in range(len(seqs))]
        pres = [batch.presence(i, np.arange(4, dtype=np.uint32), nk[i]) for i in range(len(seqs))]
        first = [batch.lookup(i, nu[i])[0].copy() for i in range(len(seqs))]
        outs.append((nk.copy(), nu.copy(), mk.copy(), off.copy(), col.copy(), cnt.copy(), rows, pres, first))
    a, b = outs
    for x, y in zip(a[:6], b[:6]):
        assert np.array_equal(x, y)
    for i in range(len(seqs)):
        assert np.array_equal(a[6][i], b[6][i]) and a[7][i] == b[7][i] and np.array_equal(a[8][i], b[8][i])
        fp, p2u = coracle.unique_kmers(seqs[i], k)
        assert np.array_equal(a[8][i], fp)
        want_rows = np.array([coracle.kmer_rows(seqs[i][p:p + k], h, m) for p in fp], dtype=np.uint64).reshape(len(fp), h)
        assert np.array_equal(a[6][i], want_rows)
    batch.close()
    st.delete_all()


def test_sparse_counts_run_gives_same_hits(hip):
    """BIGSI_RUN_SPARSE_COUNTS (what BIGSI.search uses): counters are stored only for words containing a hit, the hit
    lists (colours AND counts) must be identical to a full-counter run, for thresholds from 0 to 1."""
    from bigsi_amd._lib import BigsiHipError
    m, n_cols, h = 7001, 3000, 3
    c, st = synth_index(hip, m, n_cols, h, 17, draws=1)
    seqs = random_seqs(np.random.default_rng(4), 12, 31, 200)
    for i in (0, 3):
        st.insert_kmers(100 * i + 1, [seqs[i]], 31)
        st.insert_kmers(2999, [seqs[i][:50]], 31)
    batch = st.new_batch(seqs, 31)
    for thr in (0.0, 0.2, 0.5, 0.9, 1.0):
        batch.run(thr, force_counts=True)
        full = [x.copy() for x in batch.hits()]
        cnt0 = batch.counts(0).copy()
        batch.run(thr, force_counts=True, sparse_counts=True)
        sparse = batch.hits()
        for a, b in zip(full, sparse):
            assert np.array_equal(a, b), thr
        try:        # large batches store counters sparsely (then fetch_counts refuses); small, row-sliced ones keep them all
            assert np.array_equal(batch.counts(0), cnt0)
        except BigsiHipError as e:
            assert e.code == -6
        lo, hi = int(full[0][0]), int(full[0][1])
        assert np.array_equal(full[2][lo:hi], cnt0[full[1][lo:hi]])
    batch.close()
    st.delete_all()


def test_error_behaviour(hip):
    """Error codes of the C ABI and the exceptions the host shim turns them into (and the reference's own assertion
    for threshold > 1, bigsi/graph/bigsi.py:176)."""
    from bigsi_amd import _lib
    from bigsi_amd._lib import BigsiHipError, C
    L = _lib.lib()
    out = C.c_void_p()
    assert L.bigsi_hip_open(100, 8, 8, 3, 99, C.byref(out)) == _lib.ERR_INVALID and b"device" in L.bigsi_hip_last_error()
    assert L.bigsi_hip_open(0, 8, 8, 3, 0, C.byref(out)) == _lib.ERR_INVALID
    assert L.bigsi_hip_open(100, 8, 8, 0, 0, C.byref(out)) == _lib.ERR_INVALID
    _lib.check(L.bigsi_hip_open(100, 70, 70, 3, 0, C.byref(out)))
    ix = out
    ids = np.array([5, 100], dtype=np.uint64)
    buf = np.zeros((2, 9), np.uint8)
    assert L.bigsi_hip_set_rows(ix, _lib.ptr(ids), 2, _lib.ptr(buf), 9) == _lib.ERR_RANGE
    assert L.bigsi_hip_get_rows(ix, _lib.ptr(ids), 2, _lib.ptr(buf), 9) == _lib.ERR_RANGE
    big = np.zeros((1, 4096), np.uint8)
    assert L.bigsi_hip_set_rows(ix, _lib.ptr(ids), 1, _lib.ptr(big), 4096) == _lib.ERR_CAPACITY
    assert L.bigsi_hip_set_num_cols(ix, 10 ** 6) == _lib.ERR_CAPACITY
    assert L.bigsi_hip_insert_kmers(ix, 70, b"ACGT", _lib.ptr(np.array([0, 4], np.uint64)), 1, 3) == _lib.ERR_RANGE
    b = C.c_void_p()
    off = np.array([0, 4], np.uint64)
    assert L.bigsi_hip_batch_create(ix, b"ACGT", _lib.ptr(off), 1, 0, C.byref(b)) == _lib.ERR_INVALID        # k = 0
    assert L.bigsi_hip_batch_create(ix, b"ACGT", _lib.ptr(off), 0, 3, C.byref(b)) == _lib.ERR_INVALID        # empty batch
    assert L.bigsi_hip_batch_create(ix, b"ACGT", _lib.ptr(np.array([4, 0], np.uint64)), 1, 3, C.byref(b)) == _lib.ERR_INVALID
    _lib.check(L.bigsi_hip_batch_create(ix, b"ACGT", _lib.ptr(off), 1, 3, C.byref(b)))
    cnt = np.zeros(70, np.uint32)
    assert L.bigsi_hip_batch_fetch_counts(b, 0, _lib.ptr(cnt)) == _lib.ERR_STATE                              # before run
    assert L.bigsi_hip_batch_run(b, 1.5, 0) == _lib.ERR_INVALID                                                # threshold > 1
    assert L.bigsi_hip_batch_run(b, float("nan"), 0) == _lib.ERR_INVALID
    _lib.check(L.bigsi_hip_batch_run(b, 1.0, 0))
    assert L.bigsi_hip_batch_fetch_counts(b, 0, _lib.ptr(cnt)) == _lib.ERR_STATE                              # exact run
    assert L.bigsi_hip_batch_fetch_bitmap(b, 3, _lib.ptr(buf)) == _lib.ERR_RANGE
    hoff = np.zeros(2, np.uint64)
    _lib.check(L.bigsi_hip_batch_run(b, -0.5, 0))            # negative threshold: every sample is a hit, like count >= ceil(<0)
    assert L.bigsi_hip_batch_fetch_hits(b, _lib.ptr(hoff), None, None, 0) == _lib.ERR_CAPACITY and hoff[1] == 70
    _lib.check(L.bigsi_hip_batch_destroy(b))
    _lib.check(L.bigsi_hip_close(ix))
    assert L.bigsi_hip_close(None) == 0 and L.bigsi_hip_batch_destroy(None) == 0
    # host shim
    c = cfg(3, 1000, 3)
    bb = hip.BIGSI.build(c, [hip.BIGSI.bloom(c, ["ATC", "ATA"])], ["1"])
    with pytest.raises(AssertionError):
        bb.search("ATCATA", 1.5)
    assert bb.search("ATCéTA", 1.0) == []     # non-ASCII query: answered (test_g13_non_ascii_text), no k-mer of it is in "1"
    with pytest.raises(ValueError):
        bb.lookup([""])
    with pytest.raises(ValueError):
        hip.BIGSI.build(c, [hip.BIGSI.bloom(c, ["ATC"])], ["a", "b"])
    with pytest.raises(TypeError):
        bb.search("AT", 1.0)
    with pytest.raises(UnboundLocalError):
        bb.search("AT", 0.5)
    assert bb.search("ATCATA", -1) == bb.search("ATCATA", 0)      # the reference accepts any threshold <= 1
    bb.delete()


def test_migrate_from_reference_style_storage(hip):
    """An index held by a plain KV store in the reference's record format (here: the golden G3 rows in a dict) moves
    into HBM through the contract and answers like the reference."""
    from bigsi_amd.migrate import migrate_index
    from bigsi_amd.storage import get_storage
    from bigsi_amd.storage.contract import BaseStorage

    class Dict(BaseStorage):
        def __init__(self):
            self.storage = {}

        def delete_all(self):
            self.storage = {}

    case = load_golden("g3_search.json")
    src = Dict()
    names = list(case["samples"].keys())
    for r, hx in enumerate(case["rows"]):
        src.storage[("%d:bitarray" % r).encode()] = bytes.fromhex(hx)
    for key, v in (("number_of_rows", case["m"]), ("number_of_cols", len(names)), ("ksi:bloomfilter_size", case["m"]), ("ksi:num_hashes", case["h"])):
        src.set_integer(key, v)
    for c, nme in enumerate(names):
        src.set_string("metadata:%d" % c, nme)
        src.set_integer("metadata:%s" % nme, c)
    src.set_integer("metadata:colour_count", len(names))
    c = cfg(case["k"], case["m"], case["h"])
    assert migrate_index(src, get_storage(c)) == (case["m"], len(names), len(names))
    b = hip.BIGSI(c)
    assert rows_hex(b) == case["rows"]
    for s in case["searches"][:60]:
        t = int(s["threshold"]) if s.get("threshold_is_int") else s["threshold"]
        check_search(lambda: b.search(s["seq"], t, s["score"]), s, "migrated")
    b.delete()


def test_sparse_counters_refuse_fetch_counts_on_large_batches(hip):
    """A batch big enough to run unsliced (>= 1024 wavefronts) with BIGSI_RUN_SPARSE_COUNTS: hit lists complete, counters
    of non-hit words never stored, fetch_counts -> BIGSI_ERR_STATE."""
    from bigsi_amd._lib import BigsiHipError
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 4001, 64 * 40, 3         # 40 words -> 1 wave per query: 1100 queries = 1100 waves
    c, st = synth_index(hip, m, n_cols, h, 23, draws=1)
    orc = SynthOracle(23, 0, m, n_cols, h, 31, 1)
    seqs = random_seqs(np.random.default_rng(6), 1100, 31, 45)
    st.insert_kmers(77, [seqs[5]], 31)
    orc.insert_kmers(77, seqs[5])
    batch = st.new_batch(seqs, 31)
    batch.run(0.8, sparse_counts=True)
    off, col, cnt = batch.hits()
    with pytest.raises(BigsiHipError):
        batch.counts(5)
    for i in (0, 5, 700, 1099):
        u, want_cnt = orc.counts(seqs[i])
        want = np.flatnonzero(want_cnt >= int(np.ceil(u * 0.8)))
        assert np.array_equal(col[int(off[i]):int(off[i + 1])], want)
        assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], want_cnt[want].astype(np.uint32))
    assert 77 in col[int(off[5]):int(off[6])].tolist()
    batch.close()
    st.delete_all()


def test_early_exit_gives_identical_results(hip):
    """BIGSI_RUN_EARLY_EXIT skips rows once a segment's running AND is zero; bitmaps and hit lists must not change."""
    m, n_cols, h = 100003, 20000, 3
    c, st = synth_index(hip, m, n_cols, h, 31)
    seqs = random_seqs(np.random.default_rng(8), 300, 200, 600)     # >= 1024 wavefronts: the unsliced path
    for i in (0, 17, 299):
        st.insert_kmers(1000 + i, [seqs[i]], 31)
        st.insert_kmers(n_cols - 1, [seqs[i]], 31)
    batch = st.new_batch(seqs, 31)
    batch.run(1.0)
    ref = [x.copy() for x in batch.hits()]
    ref_bm = [batch.bitmap(i).copy() for i in (0, 1, 17, 299)]
    batch.run(1.0, early_exit=True)
    got = batch.hits()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for j, i in enumerate((0, 1, 17, 299)):
        assert np.array_equal(batch.bitmap(i), ref_bm[j])
    assert int(ref[0][-1]) == 6
    # thresholded searches: a wavefront stops once no column of its segment can reach min_kmers any more -- partial plants
    # (60 % and 35 % of a query's k-mers), thresholds on both sides of them, same hit lists and counts as without the flag
    st.insert_kmers(77, [seqs[5][: 30 + int(0.6 * (len(seqs[5]) - 30))]], 31)
    st.insert_kmers(4242, [seqs[6][: 30 + int(0.35 * (len(seqs[6]) - 30))]], 31)
    for thr in (0.9, 0.5, 0.3, 0.05, 0.0):
        batch.run(thr, sparse_counts=True)
        want = [x.copy() for x in batch.hits()]
        batch.run(thr, sparse_counts=True, early_exit=True)
        for a, b in zip(want, batch.hits()):
            assert np.array_equal(a, b), thr
        if thr == 0.5:
            off = want[0]
            assert 77 in want[1][int(off[5]):int(off[6])] and 4242 not in want[1][int(off[6]):int(off[7])]
    batch.run(1.0, force_counts=True, sparse_counts=True, early_exit=True)      # the exact answer through the counting kernel
    for a, b in zip(ref, batch.hits()):
        assert np.array_equal(a, b)
    small = st.new_batch(seqs[:2], 31)                                  # sliced (atomicAnd) path
    small.run(1.0, early_exit=True)
    off, col, _ = small.hits()
    assert col[int(off[0]):int(off[1])].tolist() == [1000, n_cols - 1] and off[2] == off[1]
    small.close()
    batch.close()
    st.delete_all()


def test_batch_reload_reuses_workspace(hip):
    """bigsi_hip_batch_reload: new sequences (more, fewer, longer, a different k, long enough to switch K1 route) in the
    same batch object give the same results as a fresh batch."""
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 9001, 500, 3
    c, st = synth_index(hip, m, n_cols, h, 3, draws=1)
    rng = np.random.default_rng(12)
    sets = [(31, random_seqs(rng, 5, 40, 80)), (31, random_seqs(rng, 40, 31, 300)), (31, random_seqs(rng, 2, 5000, 6000)),
            (15, random_seqs(rng, 7, 15, 100)), (31, ["ACGT" * 20])]
    batch = st.new_batch(sets[0][1], 31)
    for k, seqs in sets:
        batch.reload(seqs, k)
        orc = SynthOracle(3, 0, m, n_cols, h, k, 1)
        for thr in (1.0, 0.5):
            batch.run(thr)
            _, nu, mk = batch.unique()
            off, col, cnt = batch.hits()
            assert batch.n == len(seqs) and len(off) == len(seqs) + 1
            for i, s in enumerate(seqs):
                u, want_cnt = orc.counts(s)
                want = np.flatnonzero(want_cnt >= (u if thr == 1.0 else mk[i]))
                assert nu[i] == u and np.array_equal(col[int(off[i]):int(off[i + 1])], want), (k, thr, i)
    batch.close()
    # the index object's cached workspace: many searches of different shapes through one BIGSI
    b = hip.BIGSI.build(cfg(3, 1000, 3), [hip.BIGSI.bloom({"m": 1000, "h": 3, "storage-config": {}}, ["ATC", "ATA"])], ["1"])
    r1 = b.search("ATCATA", 0.5)
    r2 = b.search_batch(["ATC", "ATCATAATC", "GGG"], 0.5)
    assert b.search("ATCATA", 0.5) == r1 and r2[0] == b.search("ATC", 0.5) and r2[2] == []
    b.delete()
    st.delete_all()


def test_huge_k_takes_the_global_k1_route(hip):
    """k so large that sequence + table exceed the LDS window even though there are few positions."""
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    k, m, n_cols, h = 70000, 1009, 64, 2
    st = get_storage(cfg(k, m, h, max_cols=n_cols))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(1, 0, 1)
    s = "".join(np.random.default_rng(0).choice(list("ACGT"), size=k + 40))
    orc = SynthOracle(1, 0, m, n_cols, h, k, 1)
    batch = st.new_batch([s, s[:k]], k)
    batch.run(0.5)
    _, nu, _ = batch.unique()
    for i, q in enumerate([s, s[:k]]):
        u, cnt = orc.counts(q)
        assert nu[i] == u and np.array_equal(batch.counts(i), cnt.astype(np.uint32))
    batch.close()
    st.delete_all()


def test_search_stream_pipelined_equals_one_by_one(hip):
    """search_stream (two workspaces, batch i assembled while batch i+1 runs) == search() per sequence, in order, for batch
    sizes that do and do not divide the input, with and without score."""
    g = load_golden("g7_random.json")
    k, m, h = g["k"], g["m"], g["h"]
    c = cfg(k, m, h)
    b = hip.BIGSI.build_from_sequences(c, {n: [a, b_] for n, (a, b_) in zip(g["sample_names"], g["sample_seqs"])})
    qs = g["queries"]
    for thr, score in ((1.0, False), (0.4, False), (0.7, True)):
        want = [b.search(q, thr, score) for q in qs]
        for bs in (1, 5, 16, 48, 100):
            got = list(b.search_stream(iter(qs), thr, score, batch_size=bs))
            assert [s for s, _ in got] == qs
            assert [r for _, r in got] == want, (thr, score, bs)
    assert list(b.search_stream([], 1.0)) == []
    # the stream defers full garbage-collector passes while it runs (pause_gc): the collector is off between its first and its
    # last yield and back as it was afterwards -- whether the stream is consumed, abandoned half way, or ends in one of the
    # reference's errors -- and a caller that had it off keeps it off; pause_gc=False never touches it
    import gc
    assert gc.isenabled()
    it = b.search_stream(iter(qs), 1.0, batch_size=5)
    next(it)
    assert not gc.isenabled()
    it.close()
    assert gc.isenabled()
    with pytest.raises(TypeError):
        list(b.search_stream(qs[:7] + ["ACGT"] + qs[7:], 1.0, batch_size=5))      # a query without k-mers: reduce() of nothing
    assert gc.isenabled()
    it = b.search_stream(iter(qs), 1.0, batch_size=5, pause_gc=False)
    next(it)
    assert gc.isenabled()
    it.close()
    gc.disable()
    try:
        assert len(list(b.search_stream(iter(qs), 0.4, batch_size=16))) == len(qs) and not gc.isenabled()
    finally:
        gc.enable()
    b.delete()


@pytest.mark.parametrize("seed", range(int(os.environ.get("BIGSI_FUZZ_SEEDS", "48"))))      # a longer campaign: BIGSI_FUZZ_SEEDS=1000
def test_fuzz_random_shapes_vs_oracle(hip, seed):
    """Seeded random shapes: column counts around byte / word / tile boundaries, h 1..8, k 1..40, batch sizes from 1 (row
    sliced, atomics) to hundreds (unsliced), thresholds incl. 0 / tiny / 1, queries with duplicates, N, lowercase, too short."""
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    rng = np.random.default_rng(1000 + seed)
    n_cols = int(rng.choice([1, 7, 8, 9, 63, 64, 65, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 33000]))
    m = int(rng.choice([1, 2, 97, 1009, 65537, 300007]))
    h = int(rng.integers(1, 9))
    k = int(rng.choice([1, 2, 5, 15, 31, 31, 31, 32, 40]))
    draws = int(rng.integers(1, 4))
    st = get_storage(cfg(k, m, h, max_cols=n_cols))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(seed, 0, draws)
    orc = SynthOracle(seed, 0, m, n_cols, h, k, draws)
    nq = int(rng.choice([1, 2, 9, 40, 300])) if n_cols * m < 2e9 else 3
    alphabet = list("ACGT") if seed % 3 else list("ACGTNacgt")
    seqs = []
    for i in range(nq):
        L = int(rng.integers(max(k - 2, 0), k + int(rng.choice([1, 8, 70, 300] + ([1000] if nq <= 9 else [])))))      # (few and long: K1's several-workgroups-per-query window)
        s = "".join(rng.choice(alphabet, size=L))
        if i % 4 == 1 and L > 2 * k:
            s = s[: L // 2] + s[: L // 2]                      # repeats -> duplicate k-mers
        seqs.append(s)
    for j in range(min(3, nq)):
        col = int(rng.integers(0, n_cols))
        st.insert_kmers(col, [seqs[j]], k)
        orc.insert_kmers(col, seqs[j])
    batch = st.new_batch(seqs, k)
    check = list(range(nq)) if nq <= 40 else sorted(set([0, 1, 2] + rng.integers(0, nq, 12).tolist()))
    for thr in (1.0, float(rng.choice([0.0, 0.05, 0.3, 0.5, 0.77, 0.999]))):
        batch.run(thr, sparse_counts=bool(seed % 2), early_exit=bool((seed // 2) % 2))      # (opt-in early exit in half of the campaign: same hit lists, same counts)
        _, nu, mk = batch.unique()
        off, col, cnt = batch.hits()
        for i in check:
            u, want_cnt = orc.counts(seqs[i])
            assert nu[i] == u, (seed, i)
            if u == 0:
                want = np.zeros(0, int) if thr == 1.0 else np.flatnonzero(want_cnt >= 0) if mk[i] == 0 else np.zeros(0, int)
            else:
                want = np.flatnonzero(want_cnt >= (u if thr == 1.0 else mk[i]))
            lo, hi = int(off[i]), int(off[i + 1])
            assert np.array_equal(col[lo:hi], want), (seed, thr, i, n_cols, m, h, k)
            if u:
                assert np.array_equal(cnt[lo:hi], want_cnt[want].astype(np.uint32)), (seed, thr, i)
        if 0 < int(off[nq]) <= 200000:
            _FUZZ_SCORED[0] += _fuzz_scored(st, orc, batch, seqs, k, thr, check, off, col, cnt, (seed, thr))
    batch.close()
    st.delete_all()


_FUZZ_SCORED = [0]


def test_fuzz_checked_scored_hits_too(hip):
    """(runs after the campaign above) it compared presence strings and score records of a few hits per seed"""
    if int(os.environ.get("BIGSI_FUZZ_SEEDS", "48")) >= 48 and _FUZZ_SCORED[0] >= 0:
        print("scored hits checked by the fuzz campaign:", _FUZZ_SCORED[0])
        assert _FUZZ_SCORED[0] >= 100


def _fuzz_scored(st, orc, batch, seqs, k, thr, check, off, col, cnt, what):
    """score=True over the same random shapes: K5's presence bits against the oracle's per-k-mer rows and K6's records against
    the scalar restatement of scoring/score.py, for some hits of the checked sequences; the one-call
    bigsi_hip_search_stream_scored must return exactly what the batch calls return."""
    from bigsi_amd.scoring import SCORE_KEYS, score_columns, unpack_presence
    from oracle.ref_model import Scorer as OracleScorer
    nk, nu, _ = batch.unique()
    rec, bits, boff = batch.score_hits(off, col, None if thr == 1.0 else cnt, nk)
    text = unpack_presence(bits, boff)
    hit_seq = np.repeat(np.arange(len(seqs)), np.diff(off.astype(np.int64)))
    nonempty = nk[hit_seq] > 0
    assert not rec[~nonempty].tobytes().strip(b"\0")
    rec_nz = rec.copy()
    rec_nz["num_kmers"] = np.where(nonempty, rec["num_kmers"], 1)
    cols = score_columns(rec_nz, 1000)
    scalar = OracleScorer(1000)
    checked = 0
    for i in check:
        lo, hi = int(off[i]), int(off[i + 1])
        if hi == lo or nk[i] == 0:
            continue
        kmers, uniq, rows = orc.per_kmer_rows(seqs[i])
        where = {km: j for j, km in enumerate(uniq)}
        idx = np.array([where[km] for km in kmers])
        for t in sorted(set([lo, hi - 1, (lo + hi) // 2])):
            c = int(col[t])
            want = "".join("1" if x else "0" for x in (rows[idx, c >> 3] & (0x80 >> (c & 7))))
            got = text[8 * int(boff[t]):8 * int(boff[t]) + int(nk[i])]
            assert got == want, what + (i, c)
            ref = scalar.score(want)
            exact_keys = [key for key in SCORE_KEYS if key in ref and "value" not in key]      # (evalue / pvalue: libm's, conftest's tolerances)
            assert len(exact_keys) >= 13
            assert {key: col_[t] for key, col_ in zip(SCORE_KEYS, cols) if key in exact_keys} == {key: ref[key] for key in exact_keys}, what + (i, c)
            assert rec["percent_kmers_found"][t] == round(100 * float(cnt[t]) / int(nu[i]), 2)
            checked += 1
    # the one-call entry point: the same arrays
    nk2, nu2, off2, col2, cnt2, bits2, boff2, rec2 = st.search_many_scored(seqs, k, thr)
    assert np.array_equal(off2, off) and np.array_equal(col2, col[: int(off[-1])]) and np.array_equal(nk2, nk[: len(seqs)])
    assert np.array_equal(rec2, rec) and np.array_equal(boff2, boff) and np.array_equal(bits2, bits[: int(boff[-1])]), what
    return checked


@pytest.mark.parametrize("qlen", [200, 1100])
def test_batches_of_a_few_thousand_wavefronts(hip, qlen):
    """300 queries x 5 wavefronts (33 000 columns): one unsliced launch that is not a whole number of 4-wavefront workgroups
    per CU, so the row-AND kernels run with one-wavefront workgroups (qlen 1100: 12 counter planes and the pipelined counting
    loop as well).  Exact and thresholded, sampled queries against the oracle."""
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    m, n_cols, h, k, seed, nq = 65537, 33000, 3, 31, 91, 300
    st = get_storage(cfg(k, m, h, max_cols=n_cols))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(seed, 0, 1)
    orc = SynthOracle(seed, 0, m, n_cols, h, k, 1)
    rng = np.random.default_rng(qlen)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [lut[r].tobytes().decode("ascii") for r in rng.integers(0, 4, size=(nq, qlen), dtype=np.uint8)]
    check = sorted(set([0, 1, 7, 8, nq - 1] + rng.integers(0, nq, 8).tolist()))
    for j, i in enumerate(check[:4]):
        st.insert_kmers(64 * j + 3, [seqs[i]], k)
        orc.insert_kmers(64 * j + 3, seqs[i])
    batch = st.new_batch(seqs, k)
    for thr, sparse in ((1.0, False), (0.3, True), (0.3, False)):
        batch.run(thr, sparse_counts=sparse)
        _, nu, mk = batch.unique()
        off, col, cnt = batch.hits()
        for i in check:
            u, want_cnt = orc.counts(seqs[i])
            want = np.flatnonzero(want_cnt >= (u if thr == 1.0 else mk[i]))
            lo, hi = int(off[i]), int(off[i + 1])
            assert nu[i] == u and np.array_equal(col[lo:hi], want), (qlen, thr, i)
            assert np.array_equal(cnt[lo:hi], want_cnt[want].astype(np.uint32)), (qlen, thr, i)
        assert int(off[nq]) >= 4
    batch.close()
    st.delete_all()


@pytest.mark.parametrize("n_queries, kind", [(4000, "sliced tail"), (4784, "one-wavefront tail"), (3584, "no tail")])
def test_exact_batches_beyond_one_launch_and_their_last_launch(hip, n_queries, kind):
    """A large exact batch goes out as several co-resident launches; the last one, when the batch is not a multiple of the
    launch size, is launched like a batch of its own size (sliced with atomics below ~1000 wavefronts, one-wavefront
    workgroups otherwise).  8192 columns = one wavefront per query: 1792 queries per launch.  Sampled queries of every
    launch, and all of the last one's first and final queries, against the oracle."""
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import SynthOracle
    m, n_cols, h, k, seed = 100003, 8192, 3, 31, 77
    st = get_storage(cfg(k, m, h, max_cols=n_cols))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(seed, 0, 1)
    orc = SynthOracle(seed, 0, m, n_cols, h, k, 1)
    rng = np.random.default_rng(n_queries)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [lut[r].tobytes().decode("ascii") for r in rng.integers(0, 4, size=(n_queries, 100), dtype=np.uint8)]
    check = sorted(set([0, 1, 1791, 1792, 3583, n_queries - 1, n_queries - 2] + list(range(3584, min(3584 + 12, n_queries))) +
                       rng.integers(0, n_queries, 16).tolist()))
    check = [i for i in check if i < n_queries]
    for j, i in enumerate(check[:6]):                      # plant: some checked queries are found in a sample
        st.insert_kmers(100 + j, [seqs[i]], k)
        orc.insert_kmers(100 + j, seqs[i])
    batch = st.new_batch(seqs, k)
    batch.run(1.0)
    _, nu, _ = batch.unique()
    off, col, cnt = batch.hits()
    found = 0
    for i in check:
        u, want_cnt = orc.counts(seqs[i])
        want = np.flatnonzero(want_cnt >= u)
        lo, hi = int(off[i]), int(off[i + 1])
        assert nu[i] == u and np.array_equal(col[lo:hi], want), (kind, i)
        assert np.array_equal(cnt[lo:hi], np.full(hi - lo, u, np.uint32)), (kind, i)
        found += hi - lo
    assert found >= 6
    batch.close()
    st.delete_all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("BIGSI_API_FUZZ_SEEDS", "16"))))
def test_fuzz_api_vs_oracle_model(hip, seed):
    """The whole API against the oracle's restatement of BIGSI.search (pinned to the reference by test_oracle_golden.py):
    random samples built three ways (bloom+build, build_from_sequences, build then insert / merge), a soft-deleted
    sample, and random queries -- substrings, mutated, repeated, reverse strand, junk -- at random thresholds with and
    without scores; result lists (names, counts, order, percentages, score dicts, exceptions) must be identical."""
    import bigsi_amd
    from bigsi_amd.utils import seq_to_kmers
    from oracle.ref_model import OracleBIGSI
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.choice([5, 11, 21, 31]))
    m = int(rng.choice([251, 1000, 4099, 20011]))
    h = int(rng.integers(1, 5))
    n = int(rng.choice([1, 3, 8, 9, 33, 70]))
    genomes = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(k + 5, 400)))) for _ in range(max(2, n // 3))]
    samples = {}
    for i in range(n):
        g = genomes[int(rng.integers(0, len(genomes)))]
        a = int(rng.integers(0, max(len(g) - k - 4, 1)))
        frag = g[a:a + int(rng.integers(k, 250))]
        extra = genomes[int(rng.integers(0, len(genomes)))][: int(rng.integers(k, 3 * k))]
        samples["s%d" % i] = [frag, extra] if i % 3 else [frag]
    names = list(samples)
    kmers_of = lambda seqs: [km for s in seqs for km in seq_to_kmers(s, k)]      # noqa: E731
    how = seed % 3
    c = cfg(k, m, h)
    if how == 0:
        b = bigsi_amd.BIGSI.build(c, [bigsi_amd.BIGSI.bloom(c, kmers_of(samples[nm])) for nm in names], names)
    elif how == 1:
        b = bigsi_amd.BIGSI.build_from_sequences(c, samples)
    else:                                            # first part built, one inserted, the rest merged in from a second index
        cut = max(1, n // 2)
        first = names[:cut]
        b = bigsi_amd.BIGSI.build(c, [bigsi_amd.BIGSI.bloom(c, kmers_of(samples[nm])) for nm in first], first)
        rest = names[cut:]
        if rest:
            b.insert(bigsi_amd.BIGSI.bloom(c, kmers_of(samples[rest[0]])), rest[0])
        if len(rest) > 1:
            c2 = cfg(k, m, h)
            b2 = bigsi_amd.BIGSI.build(c2, [bigsi_amd.BIGSI.bloom(c2, kmers_of(samples[nm])) for nm in rest[1:]], rest[1:])
            b.merge(b2)
            b2.delete()
        # like the reference, a BIGSI object sizes its Scorer when it is constructed (graph/bigsi.py:140): open the grown
        # index afresh, as the next process would
        b = bigsi_amd.BIGSI(c)
    orc = OracleBIGSI.build([OracleBIGSI.bloom(kmers_of(samples[nm]), m, h) for nm in names], list(names), k, m, h)
    assert b.num_samples == n
    if n > 2 and seed % 2:
        b.delete_sample(names[1])
        orc.names[1] = "D3L3T3D"
    try:
        queries = []
        for qi in range(14):
            g = genomes[qi % len(genomes)]
            a = int(rng.integers(0, max(len(g) - k, 1)))
            s = g[a:a + int(rng.integers(k - 1, 4 * k + 40))]
            if qi % 5 == 1 and len(s) > 2:
                p = int(rng.integers(0, len(s)))
                s = s[:p] + "ACGT"[("ACGT".index(s[p]) + 1) % 4] + s[p + 1:]
            if qi % 5 == 2:
                s = s + s
            if qi % 5 == 3:
                s = bigsi_amd.utils.reverse_comp(s)
            if qi % 7 == 6:
                s = s[: len(s) // 2] + "N" + s[len(s) // 2:].lower()
            queries.append(s)
        queries += [genomes[0][:k], genomes[0][: k + 1], "A" * (k - 1)]
        for s in queries:
            for thr in (1.0, float(rng.choice([0.0, 0.2, 0.5, 0.9])), 1):
                for score in (False, True):
                    try:
                        want = {"results": orc.search(s, thr, score)}
                    except BaseException as e:  # noqa: BLE001
                        want = {"raises": type(e).__name__}
                    check_search(lambda: b.search(s, thr, score), {"out": want}, "seed %d q=%s t=%r score=%r" % (seed, s[:12], thr, score))
        got = b.search_batch([q for q in queries if len(q) >= k], 0.5)
        for q, r in zip([q for q in queries if len(q) >= k], got):
            assert_results_equal(r, orc.search(q, 0.5), "batch " + q[:12])
        km = list(dict.fromkeys(kmers_of([queries[0]])))[:20]
        if km:
            lk, want_lk = b.lookup(km), orc.lookup(km)
            assert {x: v.to01() for x, v in lk.items()} == want_lk
    finally:
        b.delete()


def test_c_host_of_the_abi(hip, tmp_path):
    """A host that is not Python (tests/c_host/search_host.c: C99, links libbigsi_hip.so only) builds the G7 index through
    bigsi_hip_insert_kmers and searches through the one-call bigsi_hip_search_batch; its printed hit lists must be the
    reference's G7 results (exact and threshold 0.4), including the grow-and-retry protocol for the hit buffers; the records and
    presence strings of bigsi_hip_search_stream_scored it prints must be the reference's score=True outputs."""
    import subprocess
    from test_abi_and_host import build_c_host
    g = load_golden("g7_random.json")
    names = g["sample_names"]
    exe = build_c_host(tmp_path)
    lines = ["%d %d %d %d %d 0.4" % (g["m"], g["h"], g["k"], len(names), len(g["queries"]))]
    lines += ["%d %s" % (len(seqs), " ".join(seqs)) for seqs in g["sample_seqs"]]
    lines += list(g["queries"])
    r = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    from test_cpu_twin import check_c_host_against_g7
    check_c_host_against_g7(r.stdout, at_least=60)


def test_storage_search_batch_entry_point(hip):
    """HipHbmStorage.search_batch -- the one call INTEGRATION.md dispatches BIGSI.search to -- against the G7 goldens."""
    import bigsi_amd
    g = load_golden("g7_random.json")
    c = cfg(g["k"], g["m"], g["h"])
    b = bigsi_amd.BIGSI.build_from_sequences(c, {nm: list(s) for nm, s in zip(g["sample_names"], g["sample_seqs"])})
    try:
        names = g["sample_names"]
        for thr in (1.0, 0.4):
            cases = [s for s in g["searches"] if s["threshold"] == thr and not s["score"] and "results" in s["out"]]
            got = b.storage.search_batch([g["queries"][s["q"]] for s in cases], g["k"], thr)
            for s, (nk, nu, col, cnt) in zip(cases, got):
                want = s["out"]["results"]
                assert sorted(names[int(x)] for x in col) == sorted(r["sample_name"] for r in want), (s["q"], thr)
                by_name = {r["sample_name"]: r for r in want}
                assert list(col) == sorted(col)
                for x, f in zip(col, cnt):
                    r = by_name[names[int(x)]]
                    assert (int(f), nu) == (r["num_kmers_found"], r["num_kmers"])
        # the entry point keeps its workspace inside the index: calls of other sizes and k, an invalid call in between,
        # and an index change must each give what a fresh batch gives
        qs = g["queries"]
        for batch_qs, k2 in ((qs[:3], g["k"]), (qs[:40] * 3, g["k"]), ([q[:25] for q in qs[:5]], 11), (qs[:1], g["k"])):
            got = b.storage.search_batch(batch_qs, k2, 0.4)
            fresh = b.storage.new_batch(batch_qs, k2)
            fresh.run(0.4, sparse_counts=True)
            _, nu, _ = fresh.unique()
            off, col, cnt = fresh.hits()
            for i, (nk_i, nu_i, col_i, cnt_i) in enumerate(got):
                assert nu_i == nu[i] and np.array_equal(col_i, col[int(off[i]):int(off[i + 1])]) and np.array_equal(cnt_i, cnt[int(off[i]):int(off[i + 1])])
            fresh.close()
            with pytest.raises(Exception):
                b.storage.search_batch(batch_qs, 0, 0.4)              # k = 0: refused, the workspace is dropped and rebuilt
        b.insert(bigsi_amd.BIGSI.bloom(c, [qs[0][i:i + g["k"]] for i in range(len(qs[0]) - g["k"] + 1)]), "extra")
        nk, nu, col, cnt = b.storage.search_batch(qs[:1], g["k"], 1.0)[0]
        assert b.sample_to_colour("extra") in col
    finally:
        b.delete()


def test_single_query_arrays_route_keeps_its_buffers_between_calls(hip):
    """HipHbmStorage.search_batch_arrays with ONE str query (what BIGSI.search calls): the argument arrays and their addresses are
    kept between calls.  Same numbers as the general route (a bytes query, a pair), results handed out are copies unless the caller
    borrows them, a query with more hits than the kept buffers hold grows them, non-ASCII text is refused as everywhere."""
    m, n_cols, h = 200003, 5000, 3
    _, st = synth_index(hip, m, n_cols, h, 99)
    st._search_cap = 64                                  # (small on purpose: the regrow path below)
    st._one_ws = None
    rng = np.random.default_rng(5)
    qs = ["".join(rng.choice(list("ACGT"), size=L)) for L in (61, 1000, 31, 300, 30)]
    for i, q in enumerate(qs[:3]):
        for c in rng.choice(n_cols, size=3 + i, replace=False):
            st.insert_kmers(int(c), [q], 31)
    kept = []
    for rounds in range(2):
        for thr in (1.0, 0.5):
            for q in qs:
                one = st.search_batch_arrays([q], 31, thr)
                gen = st.search_batch_arrays([q.encode()], 31, thr)          # (bytes: the general route)
                pair = st.search_batch_arrays([q, qs[0]], 31, thr)
                for a, b_ in zip(one, gen):
                    assert a.dtype == b_.dtype and np.array_equal(a, b_)
                assert one[0][0] == pair[0][0] and one[1][0] == pair[1][0] and np.array_equal(one[3], pair[3][: int(pair[2][1])])
                kept.append((q, thr, [a.copy() for a in one], one))
    for q, thr, snap, handed in kept:                      # copies: later calls did not change what earlier ones returned
        for a, b_ in zip(snap, handed):
            assert np.array_equal(a, b_)
    assert sum(len(x[2][3]) for x in kept) >= 2 * (3 + 4 + 5)
    b1 = st.search_batch_arrays([qs[0]], 31, 1.0, borrow=True)
    want = [a.copy() for a in st.search_batch_arrays([qs[1]], 31, 1.0)]
    b2 = st.search_batch_arrays([qs[1]], 31, 1.0, borrow=True)
    assert b1[0] is b2[0] and all(np.array_equal(a, w) for a, w in zip(b2, want))      # borrowed: the kept buffers themselves
    # more hits than the kept buffers hold (64): threshold 0 returns every sample
    nk, nu, off, col, cnt = st.search_batch_arrays([qs[3]], 31, 0.0)
    assert int(off[1]) == n_cols and np.array_equal(col, np.arange(n_cols)) and st._search_cap >= n_cols
    nk2, nu2, off2, col2, cnt2 = st.search_batch_arrays([qs[3]], 31, 0.0)             # ... and the next call has buffers of that size
    assert np.array_equal(col2, col) and np.array_equal(cnt2, cnt) and st._one_ws[0] >= n_cols
    with pytest.raises(ValueError):
        st.search_batch_arrays(["ACGT" * 10 + "\u00e9"], 31, 1.0)
    nk, nu, off, col, cnt = st.search_batch_arrays([qs[4]], 31, 1.0)                    # shorter than k: no k-mers, no hits
    assert nk[0] == 0 and nu[0] == 0 and int(off[1]) == 0
    st.delete_all()


def test_transpose_on_device_matches_numpy(hip):
    from bigsi_amd import BitRow
    from bigsi_amd.matrix.transpose import transpose, transpose_packed
    rng = np.random.default_rng(0)
    for n, m in [(5, 10), (10, 10), (7, 33), (1, 8), (9, 65)]:
        a = rng.integers(0, 2, size=(n, m)).astype(bool)
        rows = list(transpose([BitRow(r) for r in a]))
        assert [r.tolist() for r in rows] == a.T.tolist()
        assert np.array_equal(np.unpackbits(transpose_packed([BitRow(r) for r in a]), axis=1)[:, :n], a.T.astype(np.uint8))




@pytest.mark.parametrize("m,first,second", [(5003, 700, 300), (4096, 1024, 64), (1000, 129, 1), (70000, 64, 1000), (513, 5, 250), (3001, 2304, 333), (1025, 3, 2110)])
def test_tiled_transpose_matches_numpy(hip, m, first, second):
    """bigsi_hip_insert_columns: whole 64-column words go through the tiled transpose (k_transpose_regs: 1024 x 1024 tiles, bit level
    between registers, byte level in the transposing LDS read), ragged heads / tails through the column-at-a-time kernel; appending at a
    column that is not a multiple of 128, row counts that are not multiples of 8, 512 or 1024, several tile columns with a partial last
    one, filters shorter than a vector load.  Rows must equal numpy's transpose of the filters bit for bit."""
    from bigsi_amd.storage import get_storage
    rng = np.random.default_rng(m + first)
    n = first + second
    bits = rng.integers(0, 2, size=(n, m), dtype=np.uint8)
    blooms = np.packbits(bits, axis=1)                      # filter c = row c, MSB first: the reference's Bloom file format
    st = get_storage(cfg(31, m, 3, max_cols=n))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", 0), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", 3)):
        st.set_integer(key, v)
    st.insert_columns(0, blooms[:first])
    st.insert_columns(first, blooms[first:])
    got = st.get_rows_packed(np.arange(m), (n + 7) // 8)
    assert np.array_equal(got, np.packbits(bits.T, axis=1))
    # overwrite a middle range in place (col0 < num_cols): the neighbours must survive
    lo = min(70, n - 1)
    hi = min(n, lo + 200)
    bits2 = bits.copy()
    bits2[lo:hi] ^= 1
    st.insert_columns(lo, np.packbits(bits2[lo:hi], axis=1))
    assert np.array_equal(st.get_rows_packed(np.arange(m), (n + 7) // 8), np.packbits(bits2.T, axis=1))
    st.delete_all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("BIGSI_TRANSPOSE_FUZZ", "6"))))      # (campaigns: BIGSI_TRANSPOSE_FUZZ=300)
def test_transpose_fuzz_matches_numpy(hip, seed):
    """Seeded shapes around the edges of k_transpose_regs' tiles (1024 rows x 1024 columns, passes of 512 rows, 128-column heads):
    row counts from 1 to a few thousand, columns appended in two or three slabs of arbitrary widths (each slab: ragged head up to the next
    multiple of 128 columns, whole words through the tiled kernel, ragged tail), dense and sparse filters."""
    from bigsi_amd.storage import get_storage
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([1, 7, 8, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, int(rng.integers(1, 6000))]))
    slabs = [int(rng.integers(1, 2600)) for _ in range(int(rng.integers(2, 4)))]
    n = sum(slabs)
    dens = float(rng.choice([0.5, 0.03]))
    bits = (rng.random((n, m)) < dens).astype(np.uint8)
    st = get_storage(cfg(31, m, 3, max_cols=n))
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", 0), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", 3)):
        st.set_integer(key, v)
    c0 = 0
    for w in slabs:
        st.insert_columns(c0, np.packbits(bits[c0:c0 + w], axis=1))
        c0 += w
    assert np.array_equal(st.get_rows_packed(np.arange(m), (n + 7) // 8), np.packbits(bits.T, axis=1)), (m, slabs)
    st.delete_all()


@pytest.mark.parametrize("h,n_cols", [(3, 10000), (2, 700), (4, 32768), (3, 65)])
def test_one_launch_read_path_equals_three_launch_path(hip, h, n_cols):
    """Batches of reads (< 64 k-mers each, k = 31, at most 1024 queries, rows of at most 512 words) take k_reads_fused: K1 + K2 +
    K4 in one launch.  Everything observable -- k-mer counts, row ids, hit lists, AND bitmaps, presence strings -- must equal
    the three-launch route (forced here with the K1_GLOBAL test flag) and the oracle, for exact and thresholded searches,
    including reads shorter than k, duplicates, N and lowercase, and a hit list that outgrows its initial buffers."""
    from oracle.ref_model import SynthOracle
    m, seed = 30011, 77 + h
    c, st = synth_index(hip, m, n_cols, h, seed, draws=1)
    orc = SynthOracle(seed, 0, m, n_cols, h, 31, 1)
    rng = np.random.default_rng(n_cols)
    seqs = random_seqs(rng, 300, 31, 93) + ["A" * 60, "ACGT" * 15, "N" * 40, "acgtacgtac" * 5, "AC", "", "ACGTN" * 12]
    st.insert_kmers(n_cols - 1, [seqs[0]], 31)
    orc.insert_kmers(n_cols - 1, seqs[0])
    st.insert_kmers(3, [seqs[1][:50]], 31)
    orc.insert_kmers(3, seqs[1][:50])
    fused, plain, weak = st.new_batch(seqs, 31), st.new_batch(seqs, 31), st.new_batch(seqs, 31)
    for thr in (1.0, 0.4, 0.0):
        fused.run(thr, sparse_counts=True)
        plain.run(thr, sparse_counts=True, k1_global=True)
        a, b_ = fused.unique(), plain.unique()
        assert all(np.array_equal(x, y) for x, y in zip(a, b_))
        # the dedupe's exact pairwise route (taken on a fingerprint collision; forced with 1-bit fingerprints): same everything
        weak.run(thr, sparse_counts=True, weak_fingerprint=True)
        assert all(np.array_equal(x, y) for x, y in zip(a, weak.unique()))
        assert all(np.array_equal(x, y) for x, y in zip(fused.hits(), weak.hits()))
        for i in range(len(seqs)):
            assert np.array_equal(fused.rows(i, a[1][i]), weak.rows(i, a[1][i])), i
        fo, fc, fn = fused.hits()
        po, pc, pn = plain.hits()
        assert np.array_equal(fo, po) and np.array_equal(fc, pc) and np.array_equal(fn, pn), thr
        if thr == 0.0:
            assert int(fo[-1]) > 65536 or n_cols * len(seqs) <= 65536
        for i in (0, 1, 5, 300, 301, 302, 303, 306):
            assert np.array_equal(fused.rows(i, a[1][i]), plain.rows(i, a[1][i]))
            u, cnt = orc.counts(seqs[i])
            want = np.flatnonzero(cnt >= (u if thr == 1.0 else a[2][i]))
            assert a[1][i] == u and np.array_equal(fc[int(fo[i]):int(fo[i + 1])], want), (thr, i)
            assert np.array_equal(fn[int(fo[i]):int(fo[i + 1])], cnt[want].astype(np.uint32))
            if thr == 1.0:
                assert np.array_equal(fused.bitmap(i), plain.bitmap(i))
        if thr == 0.4:
            hits = fc[int(fo[1]):int(fo[2])]
            assert fused.presence(1, hits, int(a[0][1])) == plain.presence(1, hits, int(a[0][1]))
    fused.close()
    plain.close()
    weak.close()
    st.delete_all()


def test_one_launch_read_path_places_hits_without_an_order_between_workgroups(hip):
    """k_reads_fused allocates every query's place in the hit buffers with one atomic add (no workgroup waits for another), so
    the lists lie in no particular order on the device; what the caller gets is in QUERY order whichever route brings it:
    fetch_hits (ordered on the host), the export of a one-call search (k_export_reads: ordered on the device), and the route taken
    when the lists outgrow the buffers (launched again).  2600 reads in one launch against the three-launch route, at thresholds
    that give none, some and (0.0: every sample, lists regrown) all hits; then the same batch run again and again (the two
    allocation counters alternate) and a one-call search of the same reads."""
    from bigsi_amd import _lib
    m, n_cols, h, seed = 30011, 1500, 3, 91
    c, st = synth_index(hip, m, n_cols, h, seed, draws=1)
    rng = np.random.default_rng(5)
    seqs = random_seqs(rng, 2600, 31, 93)
    for i in (0, 1023, 1024, 2047, 2048, 2599):
        st.insert_kmers((7 * i + 3) % n_cols, [seqs[i]], 31)
    fused, plain = st.new_batch(seqs, 31), st.new_batch(seqs, 31)
    for thr in (1.0, 0.4, 0.0):
        plain.run(thr, sparse_counts=True, k1_global=True)
        po, pc, pn = plain.hits()
        for again in range(3):
            fused.run(thr, sparse_counts=True)
            assert fused.info().one_launch == 1
            assert fused.info().total_hits == int(po[-1])
            assert all(np.array_equal(x, y) for x, y in zip(fused.unique(), plain.unique()))
            fo, fc, fn = fused.hits()
            assert np.array_equal(fo, po) and np.array_equal(fc, pc) and np.array_equal(fn, pn), (thr, again)
        if thr == 1.0:
            for i in (0, 1023, 1024, 2047, 2048, 2599):
                assert (7 * i + 3) % n_cols in fc[int(fo[i]):int(fo[i + 1])]
        if thr > 0.0:
            nk, nu, off, col, cnt = st.search_many(seqs, 31, thr)
            assert np.array_equal(off, po) and np.array_equal(col, pc) and np.array_equal(cnt, pn), thr
    # 40 000 reads in ONE call: one launch of the read kernel, the export's 64 workgroups own 625 queries each
    many = random_seqs(rng, 40000, 31, 61)
    for i in (0, 624, 625, 20000, 39999):
        many[i] = seqs[0]                         # (planted above: its sample must come back at these places)
    got = st.search_batch(many, 31, 1.0)
    nk, nu, off, col, cnt = st.search_many(many, 31, 1.0)
    assert len(got) == 40000 and int(off[-1]) >= 5
    for i in list(range(0, 40000, 997)) + [0, 624, 625, 20000, 39999]:
        assert got[i][2].tolist() == col[int(off[i]):int(off[i + 1])].tolist() and got[i][3].tolist() == cnt[int(off[i]):int(off[i + 1])].tolist(), i
    assert (7 * 0 + 3) % n_cols in got[625][2] and (7 * 0 + 3) % n_cols in got[39999][2]
    s_ = _lib.Stats()
    _lib.check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(s_), 1))
    assert s_.read_launches_repeated == 0          # nothing waits, nothing is repeated
    fused.close()
    plain.close()
    st.delete_all()


@pytest.mark.parametrize("n_cols", [100, 4097, 16383, 16384, 16385, 20000, 32768])
@pytest.mark.parametrize("h", [2, 3, 4])
def test_read_kernel_widths_around_its_lane_layouts(hip, n_cols, h):
    """k_reads_fused over row widths on both sides of its layouts: rows of at most 256 words take one word per lane on the counting
    route (8-byte loads), wider ones -- up to the kernel's 512 -- two; the exact route always two.  Reads of 31..93 bp, some planted in
    a few samples, at thresholds 1.0 / 0.7 / 0.3 / 0.0, through batch objects and the one-call search, against the three-launch route."""
    m = 60013
    _, st = synth_index(hip, m, n_cols, h, 1000 + n_cols + h, draws=1)
    rng = np.random.default_rng(n_cols * 7 + h)
    reads = random_seqs(rng, 300, 31, 93)
    for i in range(0, 300, 9):
        for c in rng.choice(n_cols, size=min(5, n_cols), replace=False):
            st.insert_kmers(int(c), [reads[i] if i % 2 else reads[i][:55]], 31)
    fused, plain = st.new_batch(reads, 31), st.new_batch(reads, 31)
    found = 0
    for thr in (1.0, 0.7, 0.3, 0.0):
        fused.run(thr, sparse_counts=True)
        assert fused.info().one_launch == 1
        plain.run(thr, sparse_counts=True, k1_global=True)
        assert plain.info().one_launch == 0
        fo, fc, fn = fused.hits()
        po, pc, pn = plain.hits()
        assert np.array_equal(fo, po) and np.array_equal(fc, pc) and np.array_equal(fn, pn), thr
        assert all(np.array_equal(x, y) for x, y in zip(fused.unique(), plain.unique()))
        found += int(po[-1]) if thr == 1.0 else 0
        if thr in (1.0, 0.3):
            # 40 reads: read in place from the pinned staging (zero copy); all 300 at 1.0: ~20 KB, copied in by k_stage_in first
            got = st.search_batch(reads[:40] if thr < 1.0 else reads, 31, thr)
            for i, (nk, nu, col, cnt) in enumerate(got):
                assert col.tolist() == pc[int(po[i]):int(po[i + 1])].tolist() and cnt.tolist() == pn[int(po[i]):int(po[i + 1])].tolist(), (thr, i)
        if thr in (0.3, 0.0):
            # the streaming entry point over the same reads: at these thresholds a device batch's lists outgrow the hit buffers the
            # workspace starts with (and what the export carries along): the regrow route, inside the stream
            nk_, nu_, so, sc, sn = st.search_many(reads, 31, thr)
            assert np.array_equal(so, po) and np.array_equal(sc, pc) and np.array_equal(sn, pn), thr
    assert found >= 30
    fused.close()
    plain.close()
    st.delete_all()


def test_read_run_hit_buffers_too_small_for_the_caller(hip):
    """The CAPACITY protocol on the read routes, whose lists the device holds in no particular order: fetch_hits of a read batch and
    the one-call search of many reads / of one read with a caller's buffer one entry short fill the offsets (the total says what to
    bring), return BIGSI_ERR_CAPACITY, and give the full lists on the next call with room."""
    from bigsi_amd import _lib
    m, n_cols, h = 30011, 700, 3
    _, st = synth_index(hip, m, n_cols, h, 41, draws=1)
    rng = np.random.default_rng(41)
    reads = random_seqs(rng, 50, 40, 93)
    for i in range(0, 50, 3):
        for c in rng.choice(n_cols, size=3, replace=False):
            st.insert_kmers(int(c), [reads[i]], 31)
    L = _lib.lib()
    b = st.new_batch(reads, 31)
    b.run(1.0, sparse_counts=True)
    assert b.info().one_launch == 1
    off_ok, col_ok, cnt_ok = b.hits()
    total = int(off_ok[-1])
    assert total >= 17 * 3
    off = np.zeros(len(reads) + 1, np.uint64)
    col, cnt = np.zeros(total, np.uint32), np.zeros(total, np.uint32)
    assert L.bigsi_hip_batch_fetch_hits(b.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), total - 1) == _lib.ERR_CAPACITY
    assert np.array_equal(off, off_ok)
    _lib.check(L.bigsi_hip_batch_fetch_hits(b.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), total))
    assert np.array_equal(col, col_ok) and np.array_equal(cnt, cnt_ok)
    b.close()
    for qs in (reads, reads[:1], reads[3:4]):
        blob, soff = _lib.pack_seqs(qs)
        n = len(qs)
        want = [(col_ok[int(off_ok[i]):int(off_ok[i + 1])].tolist()) for i in (range(n) if n > 1 else [reads.index(qs[0])])]
        t = sum(len(w) for w in want)
        nk, nu, off = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n + 1, np.uint64)
        col, cnt = np.zeros(max(t, 1), np.uint32), np.zeros(max(t, 1), np.uint32)
        assert t >= 3
        rc = L.bigsi_hip_search_batch(st.handle, blob, _lib.ptr(soff), n, 31, 1.0, 0, _lib.ptr(nk), _lib.ptr(nu), None, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), t - 1)
        assert rc == _lib.ERR_CAPACITY and int(off[-1]) == t
        _lib.check(L.bigsi_hip_search_batch(st.handle, blob, _lib.ptr(soff), n, 31, 1.0, 0, _lib.ptr(nk), _lib.ptr(nu), None, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), t))
        assert [col[int(off[i]):int(off[i + 1])].tolist() for i in range(n)] == want
    st.delete_all()


def test_read_batches_on_library_streams_keep_their_order(hip):
    """Batches of reads run on three library streams (consecutive batches overlap).  What must still hold: a batch answers for the
    index as it was when the batch was launched even if the index is changed right after the (asynchronous) launch; a batch
    re-run after the change sees it; interleaving read batches with a gene-length batch on the index stream, reloading a batch
    while others are in flight, and stats / synchronize all give what a one-stream run gives."""
    from bigsi_amd import _lib
    m, n_cols, h, seed = 30011, 3000, 3, 7
    c, st = synth_index(hip, m, n_cols, h, seed, draws=1)
    rng = np.random.default_rng(3)
    reads = [random_seqs(rng, 1500, 61, 61) for _ in range(6)]
    genes = random_seqs(rng, 8, 900, 1200)
    ref = []
    for rs in reads:                                    # reference answers, one batch at a time, fully synchronised
        b = st.new_batch(rs, 31)
        b.run(0.3, sparse_counts=True, k1_global=True)
        ref.append([x.copy() for x in b.hits()])
        b.close()
    batches = [st.new_batch(rs, 31) for rs in reads]
    gene = st.new_batch(genes, 31)
    check = _lib.check
    L = _lib.lib()
    check(L.bigsi_hip_set_profiling(st.handle, 3))
    for i, b in enumerate(batches):                     # six one-launch runs in flight, a long batch in between
        b.run(0.3, sparse_counts=True)
        assert b.info().one_launch == 1
        if i == 2:
            gene.run(0.3, sparse_counts=True)
    # change the index while they may still be running: every read of batch 0 now matches sample 5 in full
    st.insert_kmers(5, reads[0], 31)
    for b, want in zip(batches, ref):                   # ... but the launched runs answered for the index as it was
        assert all(np.array_equal(x, y) for x, y in zip(b.hits(), want))
    s_ = _lib.Stats()
    check(L.bigsi_hip_stats(st.handle, _lib.C.byref(s_), 1))
    assert s_.and_launches_total >= 7 and 1 <= s_.and_launches < s_.and_launches_total      # every third run timed
    check(L.bigsi_hip_set_profiling(st.handle, 0))
    batches[0].run(0.3, sparse_counts=True)             # a re-run sees the inserted k-mers
    off, col, cnt = batches[0].hits()
    for i in range(0, 1500, 97):
        lo, hi = int(off[i]), int(off[i + 1])
        assert 5 in col[lo:hi] and cnt[lo:hi][list(col[lo:hi]).index(5)] == 31
    batches[1].reload(reads[2], 31)                     # reload + run while others could be in flight
    batches[3].run(0.3, sparse_counts=True)
    batches[1].run(0.3, sparse_counts=True)
    batches[2].run(0.3, sparse_counts=True)             # the same reads, same (changed) index
    got, want = batches[1].hits(), batches[2].hits()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    for b in batches + [gene]:
        b.close()
    st.delete_all()


def test_k1_row_sort_in_registers_and_beyond(hip):
    """The exact path's row lists are sorted by address inside K1.  Batches of >= 1024 queries use 256-thread workgroups, which
    sort up to 16 row ids per thread from registers and longer lists (here a 3000-bp query at h = 4: 11 880 rows) through the
    general loop; both must leave the AND results untouched: hits equal the thresholded counting route (which does not sort)
    and the oracle's bitmaps."""
    from oracle.ref_model import SynthOracle
    m, n_cols, h, seed = 65537, 700, 4, 41
    c, st = synth_index(hip, m, n_cols, h, seed, draws=1)
    orc = SynthOracle(seed, 0, m, n_cols, h, 31, 1)
    rng = np.random.default_rng(8)
    lens = [40] * 1020 + [3000, 1000, 300, 31, 2500, 64, 900, 1800]
    seqs = ["".join(rng.choice(list("ACGT"), size=L)) for L in lens]
    for i in (1020, 1021, 1024, 1027):
        st.insert_kmers((13 * i) % n_cols, [seqs[i]], 31)
        orc.insert_kmers((13 * i) % n_cols, seqs[i])
    batch = st.new_batch(seqs, 31)
    batch.run(1.0)
    off, col, cnt = [x.copy() for x in batch.hits()]
    for i in (0, 500, 1020, 1021, 1022, 1023, 1024, 1025, 1026, 1027):
        u, bm = orc.exact_bitmap(seqs[i])
        assert np.array_equal(batch.bitmap(i), bm), i
        assert (13 * i) % n_cols in col[int(off[i]):int(off[i + 1])] or i not in (1020, 1021, 1024, 1027)
    batch.run(1.0, force_counts=True)                    # the counting route: unsorted row lists
    off2, col2, cnt2 = batch.hits()
    assert np.array_equal(off, off2) and np.array_equal(col, col2) and np.array_equal(cnt, cnt2)
    batch.close()
    st.delete_all()


def test_search_stream_arrays_equals_search(hip):
    """The array-level stream (an extension for bulk callers) carries exactly what search() turns into dicts."""
    g = load_golden("g7_random.json")
    c = cfg(g["k"], g["m"], g["h"])
    b = hip.BIGSI.build_from_sequences(c, {nm: list(sq) for nm, sq in zip(g["sample_names"], g["sample_seqs"])})
    qs = g["queries"][:37]
    for thr in (1.0, 0.4):
        seen = 0
        for chunk, nu, off, col, cnt in b.search_stream_arrays(qs, thr, batch_size=10):
            for i, q in enumerate(chunk):
                want = b.search(q, thr)
                got = sorted(zip(cnt[int(off[i]):int(off[i + 1])].tolist(), col[int(off[i]):int(off[i + 1])].tolist()), key=lambda t: (-t[0], t[1]))
                assert [(r["num_kmers_found"], r["sample_name"]) for r in want] == [(f, b.colour_to_sample(cc)) for f, cc in got]
                assert all(r["num_kmers"] == int(nu[i]) for r in want)
                seen += 1
        assert seen == len(qs)
    b.delete()


def test_k4_of_several_indexes_beside_a_saturating_row_and_kernel(hip):
    """K4 (threshold + compaction) used to publish workgroup totals and spin on its predecessors' -- correct only while its whole
    grid was resident.  It is two launches without any waiting now; this runs what would have been the dangerous mix: one index
    streaming a long exact batch (every CU busy) while three other indexes -- each with a stream of its own -- compact
    thresholded batches with thousands of hits at the same time.  All hit lists against the oracle."""
    from oracle.ref_model import SynthOracle
    rng = np.random.default_rng(21)
    big_c, big = synth_index(hip, 400_000, 60_000, 3, 5)
    long_seqs = random_seqs(rng, 1500, 1000, 1000)
    big_batch = big.new_batch(long_seqs, 31)
    small = []
    for j in range(3):
        m, n_cols, h = 30_011 + 1000 * j, 20_000 + 777 * j, 3
        c, st = synth_index(hip, m, n_cols, h, 30 + j, draws=1)
        seqs = random_seqs(rng, 40, 45, 120)
        small.append((st, SynthOracle(30 + j, 0, m, n_cols, h, 31, 1), seqs, st.new_batch(seqs, 31)))
    for rounds in range(3):
        big_batch.run(1.0)                                     # ~10 ms of row streaming on the big index's stream
        for st, orc, seqs, batch in small:
            batch.run(0.3)                                     # P=6/10 counting + K4 with many hits, beside it
        for st, orc, seqs, batch in small:
            _, nu, mk = batch.unique()
            off, colours, counts = batch.hits()
            assert int(off[-1]) > 1000
            for i, s in enumerate(seqs):
                u, cnt = orc.counts(s)
                want = np.flatnonzero(cnt >= mk[i])
                lo, hi = int(off[i]), int(off[i + 1])
                assert nu[i] == u and np.array_equal(colours[lo:hi], want) and np.array_equal(counts[lo:hi], cnt[want].astype(np.uint32)), (rounds, i)
        off, colours, _ = big_batch.hits()
        assert int(off[-1]) == 0 or colours.size == int(off[-1])
    big_batch.close()
    big.delete_all()
    for st, _, _, batch in small:
        batch.close()
        st.delete_all()


def test_replace_and_drop_free_a_resident_index_of_another_shape(hip):
    """One name = one resident index, and a second config under the same name must describe the same thing (BigsiHipError);
    storage-config {"replace": true} and HipHbmStorage.drop(name) are the ways out for a long-lived process."""
    from bigsi_amd._lib import BigsiHipError
    from bigsi_amd.storage import get_storage
    from bigsi_amd.storage.hip_hbm import HipHbmStorage
    a = cfg(31, 1009, 3, name="same_name", max_cols=64)
    st = get_storage(a)
    st.delete_all()
    for key, v in (("number_of_rows", 1009), ("number_of_cols", 10), ("ksi:bloomfilter_size", 1009), ("ksi:num_hashes", 3)):
        st.set_integer(key, v)
    st.fill_synthetic(1, 0, 1)
    b = cfg(31, 2003, 4, name="same_name", max_cols=64)
    b["storage-config"]["m"], b["storage-config"]["h"] = 2003, 4
    a["storage-config"]["m"], a["storage-config"]["h"] = 1009, 3
    st2 = get_storage(a)
    assert st2.get_integer("number_of_rows") == 1009
    with pytest.raises(BigsiHipError):
        get_storage(b)
    b["storage-config"]["replace"] = True
    st3 = get_storage(b)
    with pytest.raises(KeyError):
        st3.get_integer("number_of_rows")                       # a fresh, empty store under the old name
    assert HipHbmStorage.drop("same_name") is True and HipHbmStorage.drop("same_name") is False


def test_search_stream_entry_point_equals_batch_runs(hip):
    """bigsi_hip_search_stream (any number of sequences in one call, three workspaces in flight, pinned staging + export kernel):
    reads (the one-launch kernel, several chunks), gene-length queries (chunks cut by k-mer positions), a mix with sequences
    shorter than k, and a hit buffer that is too small at first -- all equal to plain batch runs and to the oracle; the one-call
    bigsi_hip_search_batch goes through the same staging and must agree too."""
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 200_003, 9_000, 3
    c, st = synth_index(hip, m, n_cols, h, 44, draws=1)
    orc = SynthOracle(44, 0, m, n_cols, h, 31, 1)
    rng = np.random.default_rng(8)
    reads = random_seqs(rng, 70_000, 61, 61)                  # > 2 chunks of 2^15 sequences
    genes = random_seqs(rng, 1300, 900, 1100)                 # > 2 chunks of 2^19 positions
    mixed = random_seqs(rng, 300, 10, 200) + ["ACGT", ""] + random_seqs(rng, 50, 1000, 3000)
    for col_, s_ in ((5, reads[3]), (8999, reads[40_000]), (77, genes[700]), (4000, mixed[320])):
        st.insert_kmers(col_, [s_], 31)
        orc.insert_kmers(col_, s_)
    for seqs, thr in ((reads, 1.0), (reads[:40_000], 0.6), (genes, 1.0), (genes[:600], 0.45), (mixed, 0.5)):
        st._search_cap = 4                                   # force the grow-and-retry protocol
        nk, nu, off, col, cnt = st.search_many(seqs, 31, thr)
        assert off[0] == 0 and int(off[-1]) == col.size == cnt.size
        # against plain batches of 5000 sequences
        pos = 0
        for lo in range(0, len(seqs), 5000):
            part = seqs[lo:lo + 5000]
            b = st.new_batch(part, 31)
            b.run(thr, sparse_counts=True)
            bnk, bnu, _ = b.unique()
            boff, bcol, bcnt = b.hits()
            b.close()
            assert np.array_equal(nk[lo:lo + len(part)], bnk) and np.array_equal(nu[lo:lo + len(part)], bnu)
            assert np.array_equal(off[lo:lo + len(part) + 1].astype(np.int64) - int(off[lo]), boff.astype(np.int64))
            assert np.array_equal(col[int(off[lo]):int(off[lo + len(part)])], bcol) and np.array_equal(cnt[int(off[lo]):int(off[lo + len(part)])], bcnt)
            pos += len(part)
        # sampled against the oracle
        for i in list(range(0, len(seqs), max(len(seqs) // 25, 1))) + [len(seqs) - 1]:
            u, want_cnt = orc.counts(seqs[i])
            assert nu[i] == u
            if u == 0:                  # (no k-mers: the reference raises; the device's answer is whatever plain batches give, checked above)
                continue
            want = np.flatnonzero(want_cnt >= (u if thr == 1.0 else int(np.ceil(u * thr))))
            assert np.array_equal(col[int(off[i]):int(off[i + 1])], want), (thr, i)
            assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], want_cnt[want].astype(np.uint32))
    assert int(off[-1]) > 0
    # threshold 0: EVERY sample is a hit of every query (graph/bigsi.py:241-242) -- hit lists far beyond what the export kernel
    # carries along and beyond the device buffers' first size: the regrow / plain-copy routes of the collection
    few = reads[:700] + genes[:40]
    st._search_cap = 4
    nk, nu, off, col, cnt = st.search_many(few, 31, 0.0)
    assert int(off[-1]) == len(few) * n_cols and np.array_equal(np.diff(off.astype(np.int64)), np.full(len(few), n_cols))
    assert np.array_equal(col.reshape(len(few), n_cols), np.tile(np.arange(n_cols, dtype=np.uint32), (len(few), 1)))
    for i in (0, 699, 700, 739):
        u, want_cnt = orc.counts(few[i])
        assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], want_cnt.astype(np.uint32))
    # planted reads are found
    nk, nu, off, col, cnt = st.search_many(reads, 31, 1.0)
    assert 5 in col[int(off[3]):int(off[4])].tolist() and 8999 in col[int(off[40_000]):int(off[40_001])].tolist()
    # the one-call entry point: same staging, one workspace
    for part, thr in ((reads[:1000], 1.0), (reads[:1], 1.0), (genes[699:702], 0.45), (mixed, 0.5)):
        res = st.search_batch(part, 31, thr)
        nk2, nu2, off2, col2, cnt2 = st.search_many(part, 31, thr)
        for i, (a, b_, c_, d_) in enumerate(res):
            assert a == nk2[i] and b_ == nu2[i] and np.array_equal(c_, col2[int(off2[i]):int(off2[i + 1])]) and np.array_equal(d_, cnt2[int(off2[i]):int(off2[i + 1])])
    st.delete_all()


def test_search_stream_chunks_cut_at_whole_row_and_launches(hip):
    """Round 6: bigsi_hip_search_stream ends an exact chunk of gene-length queries on a multiple of the row-AND launch size (no chunk closes
    with a part launch) and gives thresholded searches four times the positions per chunk.  On 20 000 samples a launch takes 768 queries
    and 2^20 positions are 2093 queries of 531 bp: exact chunks of 1536, thresholded ones of 8371 -- several of each here, against plain
    batches (hit lists whole) and the oracle (sampled)."""
    from oracle.ref_model import SynthOracle
    m, n_cols, h = 200_003, 20_000, 3
    c, st = synth_index(hip, m, n_cols, h, 45, draws=1)
    orc = SynthOracle(45, 0, m, n_cols, h, 31, 1)
    rng = np.random.default_rng(9)
    genes = random_seqs(rng, 9000, 531, 531)
    for col_, i in ((7, 0), (19_999, 1535), (640, 1536), (12_345, 3071), (3, 3072), (9_000, 8370), (9_001, 8371), (5, 8999)):      # either side of every cut
        st.insert_kmers(col_, [genes[i]], 31)
        orc.insert_kmers(col_, genes[i])
    for seqs, thr in ((genes[:5000], 1.0), (genes, 0.45)):
        nk, nu, off, col, cnt = st.search_many(seqs, 31, thr)
        assert off[0] == 0 and int(off[-1]) == col.size == cnt.size and int(off[-1]) >= 4
        for lo in range(0, len(seqs), 3000):
            part = seqs[lo:lo + 3000]
            b = st.new_batch(part, 31)
            b.run(thr, sparse_counts=True)
            bnk, bnu, _ = b.unique()
            boff, bcol, bcnt = b.hits()
            b.close()
            assert np.array_equal(nk[lo:lo + len(part)], bnk) and np.array_equal(nu[lo:lo + len(part)], bnu)
            assert np.array_equal(off[lo:lo + len(part) + 1].astype(np.int64) - int(off[lo]), boff.astype(np.int64))
            assert np.array_equal(col[int(off[lo]):int(off[lo + len(part)])], bcol) and np.array_equal(cnt[int(off[lo]):int(off[lo + len(part)])], bcnt)
        for i in [0, 1535, 1536, 3071, 3072, len(seqs) - 1] + ([8370, 8371] if len(seqs) > 8371 else []) + list(range(11, len(seqs), 997)):
            u, want_cnt = orc.counts(seqs[i])
            want = np.flatnonzero(want_cnt >= (u if thr == 1.0 else int(np.ceil(u * thr))))
            assert nu[i] == u and np.array_equal(col[int(off[i]):int(off[i + 1])], want), (thr, i)
            assert np.array_equal(cnt[int(off[i]):int(off[i + 1])], want_cnt[want].astype(np.uint32))
    st.delete_all()
