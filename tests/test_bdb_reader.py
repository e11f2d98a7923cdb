"""The pure-Python BerkeleyDB hash-file reader (bigsi_amd/bdb.py), against files written by libdb itself (through the
standard library's dbm.ndbm, which is Berkeley DB on this image) and against the reference's own example index files."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, check_search, load_golden

ndbm = pytest.importorskip("dbm.ndbm")
if getattr(ndbm, "library", "") != "Berkeley DB":
    pytest.skip("dbm.ndbm is not backed by Berkeley DB here", allow_module_level=True)


def write_bdb(path_noext, records):
    db = ndbm.open(path_noext, "n")
    for k, v in records.items():
        db[k] = v
    db.close()
    return path_noext + ".db"


def test_roundtrip_small_and_overflow_values(tmp_path):
    from bigsi_amd.bdb import BdbHashFile, read_all
    rng = np.random.default_rng(0)
    rec = {b"number_of_rows:int": b"1000", b"empty": b"", b"k" * 300: b"v" * 3}
    for i in range(3000):                                   # many buckets / page splits
        rec[b"%d:bitarray" % i] = rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
    for i, n in enumerate((1000, 4071, 4096, 5000, 20000, 123457)):   # around and far beyond the 4 KiB page: overflow chains
        rec[b"big%d:bitarray" % i] = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    rec[b"K" * 9000] = b"long key"                            # overflow key
    fn = write_bdb(str(tmp_path / "t"), rec)
    got = read_all(fn)
    assert got == rec
    with BdbHashFile(fn) as db:
        assert db.pagesize == 4096 and db.version >= 7
        only = dict(db.items(want_key=lambda k: k.startswith(b"big")))
        assert sorted(only) == sorted(k for k in rec if k.startswith(b"big"))


def test_rejects_other_files(tmp_path):
    from bigsi_amd.bdb import BdbFormatError, BdbHashFile
    p = tmp_path / "x"
    p.write_bytes(b"\0" * 4096)
    with pytest.raises(BdbFormatError):
        BdbHashFile(str(p))


def test_reference_example_index_files():
    """example-data/test-bigsi/{graph,metadata} of the reference (BerkeleyDB hash v9, 16 KiB pages): contents as
    scripts/convert_v01_to_v03.py:23-55 describes them."""
    from bigsi_amd.bdb import read_all
    g = read_all(os.path.join(GOLDEN, "bdb_v01_graph.db"))
    m = read_all(os.path.join(GOLDEN, "bdb_v01_metadata.db"))
    assert sorted(g) == [struct.pack(">I", i) for i in range(1000)] and {len(v) for v in g.values()} == {1}
    assert int.from_bytes(m[b"bloom_filter_size"], "big") == 1000 and int.from_bytes(m[b"kmer_size"], "big") == 31
    assert int.from_bytes(m[b"num_hashes"], "big") == 1 and int.from_bytes(m[b"num_colours"], "big") == 2
    assert (m[b"colour0"], m[b"colour1"]) == (b"s1", b"s2")


@pytest.mark.gpu
def test_import_v03_file_and_v01_directory(tmp_path):
    import shutil

    import bigsi_amd
    from bigsi_amd import bdb
    from bigsi_amd.storage import get_storage
    from oracle.ref_model import OracleBIGSI
    # v0.3: the golden G3 index written as the reference's BerkeleyDB backend would store it
    case = load_golden("g3_search.json")
    names = list(case["samples"].keys())
    rec = {b"%d:bitarray" % r: bytes.fromhex(hx) for r, hx in enumerate(case["rows"])}
    rec.update({b"number_of_rows:int": b"%d" % case["m"], b"number_of_cols:int": b"%d" % len(names),
                b"ksi:bloomfilter_size:int": b"%d" % case["m"], b"ksi:num_hashes:int": b"%d" % case["h"],
                b"metadata:colour_count:int": b"%d" % len(names)})
    for c, nme in enumerate(names):
        rec[b"metadata:%d:string" % c] = nme.encode()
        rec[("metadata:%s:int" % nme).encode()] = b"%d" % c
    fn = write_bdb(str(tmp_path / "v03"), rec)
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "bdb3"}, "k": case["k"], "m": case["m"], "h": case["h"]}
    assert bdb.import_index(fn, get_storage(cfg)) == (case["m"], len(names))
    b = bigsi_amd.BIGSI(cfg)
    assert [bytes(r).hex() for r in b.storage.get_rows_packed(np.arange(case["m"]))] == case["rows"]
    for s in case["searches"][:40]:
        t = int(s["threshold"]) if s.get("threshold_is_int") else s["threshold"]
        check_search(lambda: b.search(s["seq"], t, s["score"]), s, "bdb v0.3")
    b.delete()
    # v0.1: the reference's own example index, straight from its two BerkeleyDB files
    d = tmp_path / "test-bigsi"
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, "bdb_v01_graph.db"), d / "graph")
    shutil.copy(os.path.join(GOLDEN, "bdb_v01_metadata.db"), d / "metadata")
    cfg1 = {"storage-engine": "hip-hbm", "storage-config": {"name": "bdb1"}, "k": 31, "m": 1000, "h": 1}
    assert bdb.import_v01_index(str(d), get_storage(cfg1)) == (1000, 2, 31)
    b1 = bigsi_amd.BIGSI(cfg1)
    rows = b1.storage.get_rows_packed(np.arange(1000))
    graph = bdb.read_all(str(d / "graph"))
    assert [bytes(r) for r in rows] == [graph[struct.pack(">I", i)] for i in range(1000)]
    assert b1.num_samples == 2 and b1.colour_to_sample(1) == "s2" and b1.num_hashes == 1
    orc = OracleBIGSI(rows, ["s1", "s2"], 31, 1)
    g = load_golden("g9_frontend.json")
    seqs = [l for l in g["fasta_text"]["example"].splitlines() if l and not l.startswith(">")][:5] + ["GATCGTTTGCGGCCACAGTTGCCAGAGATGAAAG"]
    for s in seqs:
        for t in (1.0, 0.3):
            assert b1.search(s, t) == orc.search(s, t)
    b1.delete()


def test_native_small_records_equal_the_python_page_walk(tmp_path):
    """bigsi_hip_bdb_small_records (the library's threaded page scan, no device) against bigsi_amd/bdb.py on a file libdb wrote:
    every non-row record, values in overflow chains included; the row records counted and measured."""
    from bigsi_amd import _lib, bdb
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libbigsi_hip.so not built")
    rng = np.random.default_rng(1)
    rec = {b"number_of_rows:int": b"700", b"number_of_cols:int": b"40000", b"empty:string": b"", b"metadata:7:string": "séven".encode(),
           b"blob:string": rng.integers(0, 256, size=30000, dtype=np.uint8).tobytes(), b"12:bitarrayX": b"not a row", b"x12:bitarray": b"nor this"}
    for i in range(700):
        rec[b"%d:bitarray" % i] = rng.integers(0, 256, size=int(rng.choice([3, 5000, 4071, 4096])), dtype=np.uint8).tobytes()
    rec[b"K" * 9000] = b"long key"                          # an overflow KEY: no index record, skipped by both
    fn = write_bdb(str(tmp_path / "s"), rec)
    small, n_rows, widest = bdb.small_records(fn, threads=3)
    want = {k: v for k, v in bdb.read_all(fn).items() if not (k.endswith(b":bitarray") and k[:-9].isdigit()) and len(k) < 4000}
    assert small == want and n_rows == 700 and widest == 5000
    with pytest.raises(_lib.BigsiHipError):
        bdb.small_records(str(tmp_path / "nope"))
    p = tmp_path / "junk"
    p.write_bytes(b"\1" * 8192)
    with pytest.raises(_lib.BigsiHipError):
        bdb.small_records(str(p))


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_native_import_equals_the_python_route(tmp_path, devices):
    """import_index below Python (bigsi_hip_load_rows_file on the BerkeleyDB file itself: rows inline and in overflow chains, shorter
    and longer than ceil(N/8)) == the Python page walk, on one GPU and over three shards; a store with a missing row falls back."""
    from bigsi_amd import bdb
    from bigsi_amd.storage import get_storage
    rng = np.random.default_rng(2)
    m, n = 3001, 70_001
    rb = (n + 7) // 8
    rows = rng.integers(0, 256, size=(m, rb), dtype=np.uint8)
    rows[:, rb - 1] &= 0x80                                  # 70001 columns: 7 pad bits
    rec = {b"number_of_rows:int": b"%d" % m, b"number_of_cols:int": b"%d" % n, b"ksi:bloomfilter_size:int": b"%d" % m, b"ksi:num_hashes:int": b"3",
           b"metadata:colour_count:int": b"2", b"metadata:0:string": b"zero", b"metadata:zero:int": b"0"}
    for r in range(m):
        raw = rows[r].tobytes()
        rec[b"%d:bitarray" % r] = raw[:100] if r % 50 == 7 else raw + b"\0\0\0" if r % 50 == 9 else raw      # some stored shorter / longer
        if r % 50 == 7:
            rows[r, 100:] = 0
    fn = write_bdb(str(tmp_path / "big"), rec)
    got = {}
    for native in (True, False):
        sc = {"name": "bdbnat%d%s" % (native, "g" if devices else ""), "max_cols": n}
        if devices:
            sc["devices"] = devices
        st = get_storage({"storage-engine": "hip-hbm", "storage-config": sc, "k": 31, "m": m, "h": 3})
        assert bdb.import_index(fn, st, native=native) == (m, n)
        got[native] = np.asarray(st.get_rows_packed(np.arange(m), rb)).copy()
        assert st.get_string("metadata:0") == "zero" and st.get_integer("number_of_cols") == n and st.get_integer("ksi:num_hashes") == 3
        st.delete_all()
    assert np.array_equal(got[True], rows) and np.array_equal(got[False], rows)
    del rec[b"5:bitarray"]                                   # not every row present: the Python route (KeyError semantics per row) takes over
    fn2 = write_bdb(str(tmp_path / "holey"), rec)
    st = get_storage({"storage-engine": "hip-hbm", "storage-config": {"name": "bdbholey", "max_cols": n}, "k": 31, "m": m, "h": 3})
    assert bdb.import_index(fn2, st) == (m, n)
    with pytest.raises(KeyError):
        st.get_bitarray(5)
    assert st.get_bitarray(6).tobytes() == rows[6].tobytes()
    st.delete_all()
