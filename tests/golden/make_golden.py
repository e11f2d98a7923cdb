#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE.

Run (in the build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

This script is test infrastructure.  It imports the reference package from
/root/reference and records input -> output pairs as JSON / NPZ.  It copies no
reference source.  Stand-ins for third-party modules that are not installed
here:

  * ``mmh3``  -> ``sklearn.utils.murmurhash3_32(bytes, seed, positive=False)``
    (MurmurHash3_x86_32, signed result), validated below against the
    reference's own known answers (bigsi/tests/bloom/test_create_bloomfilter.py:6-8)
    and mmh3's documented vectors (.conda/mmh3/meta.yaml:35,44-45).
  * ``redis`` -> empty stub (bigsi/storage/__init__.py:1 imports it unconditionally).
  * a dict-backed ``BaseStorage`` subclass registered in STORAGE_DICT (bsddb3 /
    rocksdb bindings are not installable offline).

Harness patches for two hazards of running the 2018 code on a modern stack
(neither edits the reference; both restore the behaviour its tests pin):

  * H1  ``BloomFilter.__init__`` leaves ``bitarray(m)`` uninitialised
        (bloom/bloomfilter.py:20) -> wrapped to ``setall(False)``.
  * H2  ``np.where(bitarray)`` sees the byte buffer with bitarray>=1.x, so
        ``non_zero_bitarrary_positions`` (utils/fncts.py:28-29) returns byte
        indices -> replaced in bigsi.graph.bigsi with the bit-iterating
        equivalent (semantics pinned by tests/graph/test_end_to_end.py:78-89).
"""
import json
import math
import os
import sys
import types
import warnings

warnings.filterwarnings("ignore")

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------- stand-ins
from sklearn.utils import murmurhash3_32  # noqa: E402

mmh3 = types.ModuleType("mmh3")


def _mmh3_hash(key, seed=0):
    if isinstance(key, str):
        key = key.encode("utf-8")
    return int(murmurhash3_32(key, seed, positive=False))


mmh3.hash = _mmh3_hash
sys.modules["mmh3"] = mmh3

redis = types.ModuleType("redis")
redis.StrictRedis = object
sys.modules["redis"] = redis

sys.path.insert(0, REF)
import numpy as np  # noqa: E402
from bitarray import bitarray  # noqa: E402

import bigsi  # noqa: E402
import bigsi.bloom.bloomfilter as ref_bloom  # noqa: E402
import bigsi.graph.bigsi as ref_graph  # noqa: E402
import bigsi.storage as ref_storage  # noqa: E402
from bigsi import BIGSI  # noqa: E402
from bigsi.bloom import generate_hashes  # noqa: E402
from bigsi.scoring.score import Scorer, remove_short_ones, tabulate_score  # noqa: E402
from bigsi.storage.base import BaseStorage  # noqa: E402
from bigsi.utils import canonical, reverse_comp, seq_to_kmers  # noqa: E402

# mmh3 stand-in validation against the reference's and mmh3's own known answers
assert generate_hashes("ATT", 3, 25) == {2, 15, 17}
assert generate_hashes("ATT", 1, 25) == {15}
assert generate_hashes("ATT", 2, 50) == {15, 27}
assert mmh3.hash("foo") == -156908512
assert mmh3.hash("foo", 42) == -1322301282
assert mmh3.hash("aaaa", 0x9747B28C) == 1519878282


# H1
_orig_bf_init = ref_bloom.BloomFilter.__init__


def _bf_init(self, m, h):
    _orig_bf_init(self, m, h)
    self.bitarray.setall(False)


ref_bloom.BloomFilter.__init__ = _bf_init


# H2
def _nonzero_positions(ba):
    return [i for i, b in enumerate(ba) if b]


ref_graph.non_zero_bitarrary_positions = _nonzero_positions

_STORES = {}


class DictStorage(BaseStorage):
    """In-memory stand-in for a local KV file: contents survive close()/reopen by name."""

    def __init__(self, storage_config):
        self.name = storage_config.get("name", "default")
        self.storage = _STORES.setdefault(self.name, {})

    def delete_all(self):
        _STORES[self.name] = {}
        self.storage = _STORES[self.name]

    def close(self):
        pass


ref_storage.STORAGE_DICT["dict"] = DictStorage


def cfg(name, k, m, h):
    return {"storage-engine": "dict", "storage-config": {"name": name}, "k": k, "m": m, "h": h}


def jsonable(x):
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        x = float(x)
    if isinstance(x, float):
        if math.isinf(x) or math.isnan(x):
            return {"__float__": repr(x)}
        return x
    if isinstance(x, bitarray):
        return x.to01()
    return x


def lookup_dict(d):
    """lookup() returns a dict built from a set: its order depends on the interpreter's string hash seed -> sort for a
    reproducible fixture (the tests compare these as dicts)."""
    return {k: jsonable(d[k]) for k in sorted(d)}


def run_search(b, seq, threshold, score):
    """search() outcome as {'results': [...]} or {'raises': 'ExcName'}."""
    try:
        return {"results": jsonable(b.search(seq, threshold, score))}
    except BaseException as e:  # noqa: BLE001 - the exception type is the golden
        return {"raises": type(e).__name__}


def dump(name, obj, indent=1):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=indent, sort_keys=False)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


def rows_of(b):
    """All stored rows of an index, as the raw bytes the storage contract holds (hex)."""
    return [b.storage[("%d:bitarray" % r)].hex() for r in range(b.bloomfilter_size)]


# --------------------------------------------------------------- G1: hashing
def g1_hash():
    rng = np.random.default_rng(101)
    strings = ["ATT", "A", "", "AT", "ATTA", "ATTAC", "ACGTACGTAC", "acgt", "ACGTN", "NNNN",
               "GATCGTTTGCGGCCACAGTTGCCAGAGATGA", "TCATCTCTGGCAACTGTGGCCGCAAACGATC", "foo", "aaaa"]
    for L in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 61, 64):
        for _ in range(3):
            strings.append("".join(rng.choice(list("ACGT"), size=L)))
    for _ in range(8):
        strings.append("".join(rng.choice(list("ACGTNacgtRY-"), size=31)))
    seeds = [0, 1, 2, 3, 4, 7, 42, 0x9747B28C & 0x7FFFFFFF]
    raw = [{"s": s, "hashes": [mmh3.hash(s, sd) for sd in seeds]} for s in strings]
    gen = []
    for s in strings:
        if not s:
            continue
        for (h, m) in [(3, 25), (1, 25), (2, 50), (3, 1000), (4, 25000000), (3, 1 << 31), (5, (1 << 32) + 15), (3, 7)]:
            gen.append({"s": s, "h": h, "m": m,
                        "rows_in_seed_order": [ref_bloom._hash(s, sd, m) for sd in range(h)],
                        "set": sorted(generate_hashes(s, h, m))})
    canon = [{"s": s, "canonical": canonical(s), "reverse_comp": reverse_comp(s)} for s in strings if s]
    kmers = []
    for (seq, k) in [("ATACACAAT", 3), ("AT", 3), ("ATA", 3), ("", 3), ("ACGTNACGT", 4)]:
        kmers.append({"seq": seq, "k": k, "kmers": list(seq_to_kmers(seq, k))})
    dump("g1_hash.json", {"seeds": seeds, "mmh3": raw, "generate_hashes": gen, "canonical": canon,
                          "seq_to_kmers": kmers})


# ------------------------------------------------------- G2: lookup semantics
def g2_lookup():
    out = []
    kmers1 = ["ATC", "ATG", "ATA", "ATT"]
    kmers2 = ["ATC", "ATG", "ATA", "TTT"]
    for (m, h) in [(250, 3), (2500, 2), (250, 1)]:
        c = cfg("g2_%d_%d" % (m, h), 3, m, h)
        ref_storage.get_storage(c).delete_all()
        b = BIGSI.build(c, [BIGSI.bloom(c, kmers1), BIGSI.bloom(c, kmers2)], ["s1", "s2"])
        case = {"m": m, "h": h, "k": 3, "samples": [kmers1, kmers2],
                "blooms": [BIGSI.bloom(c, kmers1).tobytes().hex(), BIGSI.bloom(c, kmers2).tobytes().hex()],
                "rows": rows_of(b), "lookups": []}
        for q in [["ATC"], ["ATC", "ATC", "ATT"], ["ATC", "ATC", "ATT", "TTT"], "ATC", ["AAT"], ["GGG"], ["acg", "ANT"]]:
            for rtz in (True, False):
                case["lookups"].append({"kmers": q, "remove_trailing_zeros": rtz,
                                        "result": lookup_dict(b.lookup(q, remove_trailing_zeros=rtz))})
        out.append(case)
    dump("g2_lookup.json", out)


# ------------------------------------------- G3: search semantics + edge cases
def g3_search():
    c = cfg("g3", 3, 1000, 3)
    ref_storage.get_storage(c).delete_all()
    samples = {"a": "ATACACAAT", "b": "ATACACAAC", "c": "ACAGAGAAC", "d": "ATACACAAT"}
    blooms = [BIGSI.bloom(c, seq_to_kmers(s, 3)) for s in samples.values()]
    b = BIGSI.build(c, blooms, list(samples.keys()))
    case = {"k": 3, "m": 1000, "h": 3, "samples": samples,
            "blooms": [x.tobytes().hex() for x in blooms], "rows": rows_of(b), "searches": []}
    seqs = ["ATACACAAT", "ACAGAGAAC", "ACAGTTAAC", "ATACACAAC", "ATAT", "ATANACAAT", "atacacaat",
            "AT", "ATA", "ATAC", "ATACA", "", "ACAGAGAACATACACAAT", "TTGTGTATTGTGTAT", "GGGGGGG"]
    thresholds = [1.0, 1, 0.5, 0.4, 0.1, 0.0, 0, 0.29, 0.8333333333333334, 0.99, 1.5]
    for s in seqs:
        for t in thresholds:
            for sc in (False, True):
                case["searches"].append({"seq": s, "threshold": t, "threshold_is_int": isinstance(t, int),
                                         "score": sc, "out": run_search(b, s, t, sc)})
    # deleted sample semantics (graph/metadata.py:33-38, graph/bigsi.py:186-190)
    b.delete_sample("a")
    case["after_delete_a"] = {
        "num_samples": b.num_samples,
        "colour_to_sample_0": b.colour_to_sample(0),
        "sample_to_colour_a": b.sample_to_colour("a"),
        "searches": [{"seq": s, "threshold": t, "score": False, "out": run_search(b, s, t, False)}
                     for s in ["ATACACAAT", "ACAGAGAAC"] for t in (1.0, 0.5, 0.0)],
    }
    dump("g3_search.json", case)


# ------------------------- G4: config #1 (k=31, m=1000, h=3, test_kmers fixture)
def read_fasta(path):
    recs, name, buf = [], None, []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(buf)))
            name, buf = line[1:], []
        elif line:
            buf.append(line)
    if name is not None:
        recs.append((name, "".join(buf)))
    return recs


def g4_config1():
    c = cfg("g4", 31, 1000, 3)
    ref_storage.get_storage(c).delete_all()
    base = [l.strip() for l in open(os.path.join(REF, "bigsi/tests/data/test_kmers.txt")) if l.strip()]
    rng = np.random.default_rng(4)
    q_example = read_fasta(os.path.join(REF, "example-data/query.fasta"))
    q_test = read_fasta(os.path.join(REF, "bigsi/tests/data/query.fasta"))
    # s1: the reference's 100 fixture k-mers; s2: 60 of them + k-mers of the first example query;
    # s3: seeded random 31-mers + k-mers of the tests/data query
    s1 = list(base)
    s2 = [base[i] for i in rng.permutation(100)[:60]] + list(seq_to_kmers(q_example[0][1], 31))
    s3 = ["".join(rng.choice(list("ACGT"), size=31)) for _ in range(80)] + list(seq_to_kmers(q_test[0][1], 31))
    samples = {"s1": s1, "s2": s2, "s3": s3}
    blooms = [BIGSI.bloom(c, ks) for ks in samples.values()]
    b = BIGSI.build(c, blooms, list(samples.keys()))
    case = {"k": 31, "m": 1000, "h": 3, "samples": samples,
            "blooms": [x.tobytes().hex() for x in blooms], "rows": rows_of(b), "searches": [],
            "fixture_bloom_is_superset_of_h3": None}
    fixture = bitarray()
    with open(os.path.join(REF, "bigsi/tests/data/test_kmers.bloom"), "rb") as f:
        fixture.fromfile(f)
    case["fixture_bloom_is_superset_of_h3"] = bool((blooms[0] & fixture[:1000]) == blooms[0])
    queries = [("example:%s:%d" % (n, i), s) for i, (n, s) in enumerate(q_example)] + \
              [("tests:%s" % n, s) for n, s in q_test] + \
              [("kmer0", base[0]), ("kmer0+3", base[0] + "AAG"), ("two-kmers", base[5] + base[6]),
               ("rc-kmer0", reverse_comp(base[0]))]
    for name, s in queries:
        for t in (1.0, 0.4, 0.1):
            for sc in (False, True):
                case["searches"].append({"name": name, "seq": s, "threshold": t, "score": sc,
                                         "out": run_search(b, s, t, sc)})
    dump("g4_config1.json", case)


# ------------------------------------------------------------- G5: scoring
def g5_scoring():
    rng = np.random.default_rng(5)
    kat = open(os.path.join(REF, "bigsi/tests/scoring.py")).read().split('s = "')[1].split('"')[0]
    strs = [kat, "11", "1", "0", "00", "01", "10", "111", "000", "101", "1101110", "1100111", "0" * 40, "1" * 40,
            "1" * 31 + "0" * 31 + "1" * 31, "0" * 33 + "1" + "0" * 34, "1" * 100 + "0" * 35 + "1" * 100]
    for n in (2, 3, 4, 5, 8, 31, 34, 35, 36, 64, 100, 300, 970, 2000):
        for p in (0.05, 0.5, 0.9, 0.99):
            strs.append("".join("1" if x else "0" for x in (rng.random(n) < p)))
    cases = []
    for db in (0, 1, 3, 4, 10000, 500000):
        sc = Scorer(db)
        for s in strs:
            try:
                cases.append({"db_size": db, "s": s, "score": jsonable(sc.score(s))})
            except BaseException as e:  # noqa: BLE001
                cases.append({"db_size": db, "s": s, "raises": type(e).__name__})
    helpers = {"remove_short_ones": [{"s": s, "out": remove_short_ones(s)} for s in strs if s],
               "tabulate_score": [{"s": s, "out": tabulate_score(s)} for s in strs if s]}
    dump("g5_scoring.json", {"cases": cases, "helpers": helpers})


# ----------------------------------------------- G6: threshold / percent maths
def g6_arith():
    thr = [0.0, 0.1, 0.29, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8333333333333334, 0.9, 0.99, 0.999999, 1.0, 1e-9, 1 / 3, 2 / 3]
    ceil_tab = [{"n": n, "t": t, "min_kmers": math.ceil(n * t)}
                for n in list(range(0, 130)) + [970, 1000, 4096, 65535, 65536, 100000, 1234567] for t in thr]
    pct = [{"found": f, "n": n, "percent": round(100 * float(f) / n, 2)}
           for n in [1, 2, 3, 6, 7, 9, 11, 31, 64, 100, 970, 1000, 65536] for f in sorted({0, 1, n // 3, n // 2, n - 1, n}) if f <= n]
    dump("g6_arith.json", {"min_kmers": ceil_tab, "percent": pct})


# ----------------------------- G7: medium random index (k=31) rows/lookups/counts
def g7_random():
    rng = np.random.default_rng(7)
    m, N, k, h = 4096, 200, 31, 3   # N not a multiple of 8 or 64 on purpose
    c = cfg("g7", k, m, h)
    ref_storage.get_storage(c).delete_all()
    genomes = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(200, 420)))) for _ in range(24)]
    sample_seqs, sample_kmers = [], []   # a sample = the k-mers of two sequence fragments
    for i in range(N):
        g = genomes[int(rng.integers(0, len(genomes)))]
        g2 = genomes[int(rng.integers(0, len(genomes)))]
        a = int(rng.integers(0, 80))
        fa, fb = g[a:a + int(rng.integers(60, 200))], g2[:int(rng.integers(31, 120))]
        sample_seqs.append([fa, fb])
        sample_kmers.append(list(seq_to_kmers(fa, k)) + list(seq_to_kmers(fb, k)))
    blooms = [BIGSI.bloom(c, ks) for ks in sample_kmers]
    names = ["s%03d" % i for i in range(N)]
    b = BIGSI.build(c, blooms, names)
    rows = np.frombuffer(b"".join(b.storage[("%d:bitarray" % r)] for r in range(m)), dtype=np.uint8).reshape(m, -1)
    queries = []
    for qi in range(48):
        g = genomes[qi % len(genomes)]
        a = int(rng.integers(0, 100))
        s = g[a:a + int(rng.integers(31, 160))]
        if qi % 5 == 1:   # mutate a base
            p = int(rng.integers(0, len(s)))
            s = s[:p] + "ACGT"[("ACGT".index(s[p]) + 1) % 4] + s[p + 1:]
        if qi % 7 == 3:   # tandem repeat -> duplicate k-mers
            s = s[:40] + s[:40] + s[:40]
        if qi % 11 == 5:  # reverse complement strand
            s = reverse_comp(s)
        if qi % 13 == 6:
            s = s[:20] + "N" + s[21:]
        queries.append(s)
    searches, lookups, counts = [], [], []
    for qi, s in enumerate(queries):
        kms = list(seq_to_kmers(s, k))
        lk = b.lookup(kms, remove_trailing_zeros=False)
        cnt = ref_graph.unpack_and_sum(list(lk.values()))
        counts.append(np.asarray(cnt, dtype=np.int32))
        if qi < 6:
            lookups.append({"seq": s, "lookup": {km: lk[km].tobytes().hex() for km in sorted(lk)}})
        for t in ((1.0, 0.7, 0.4, 0.0) if qi < 3 else (1.0, 0.7, 0.4)):
            for sc in ((False, True) if qi < 8 else (False,)):
                searches.append({"q": qi, "threshold": t, "score": sc, "out": run_search(b, s, t, sc)})
    np.savez_compressed(os.path.join(HERE, "g7_random.npz"), rows=rows,
                        counts=np.stack(counts), n_cols=np.int64(N))
    print("wrote g7_random.npz")
    dump("g7_random.json", {"k": k, "m": m, "h": h, "n_cols": N, "sample_names": names,
                            "sample_seqs": sample_seqs, "queries": queries,
                            "lookups": lookups, "searches": searches}, indent=None)


# ------------------------------- G8: storage/bitmatrix contract (insert, bytes)
def g8_storage():
    from bigsi.matrix import BitMatrix
    out = {}
    st = DictStorage({"name": "g8"})
    st.delete_all()
    ba = bitarray("110101111010")
    st.set_bitarray("test", ba)
    out["bitarray_bytes"] = {"bits": ba.to01(), "stored_hex": st["test:bitarray"].hex(),
                             "get_bitarray": st.get_bitarray("test").to01()}
    st.set_bit("test", 0, 0)
    out["after_set_bit_0_0"] = st.get_bitarray("test").to01()
    st.set_integer("x", 112)
    out["integer_bytes"] = st["x:int"].decode()
    out["incr"] = [st.incr("ctr"), st.incr("ctr")]
    rows = [bitarray("001"), bitarray("001"), bitarray("111"), bitarray("001"), bitarray("111")] * 5
    st.delete_all()
    bm = BitMatrix.create(st, rows, len(rows), len(rows[0]))
    steps = {"col0": bm.get_column(0).to01(), "col2": bm.get_column(2).to01()}
    bm.insert_column(bitarray("1" * 25), 0)
    steps["col0_after_insert"] = bm.get_column(0).to01()
    steps["row1_after_insert0"] = bm.get_row(1).to01()
    bm.insert_column(bitarray("1" * 25), 3)
    steps["row1_after_insert3"] = bm.get_row(1).to01()
    steps["num_cols_after"] = bm.num_cols
    out["bitmatrix"] = steps
    # BIGSI.insert / merge end-to-end (tests/graph/test_end_to_end.py:28-49, :135-154 bodies)
    c = cfg("g8i", 3, 1000, 3)
    ref_storage.get_storage(c).delete_all()
    b = BIGSI.build(c, [BIGSI.bloom(c, ["ATC", "ATA"])], ["1"])
    b.insert(BIGSI.bloom(c, ["ATC", "ATT"]), "2")
    out["insert"] = {"num_samples": b.num_samples,
                     "lookup": lookup_dict(b.lookup(["ATC", "ATA", "ATT"])),
                     "rows": rows_of(b)}
    c1, c2 = cfg("g8m1", 3, 1000, 3), cfg("g8m2", 3, 1000, 3)
    for cc in (c1, c2):
        ref_storage.get_storage(cc).delete_all()
    k1, k2 = list(seq_to_kmers("ATACACAAT", 3)), list(seq_to_kmers("ATACACAAC", 3))
    b1 = BIGSI.build(c1, [BIGSI.bloom(c1, k1)], ["a"])
    b2 = BIGSI.build(c2, [BIGSI.bloom(c2, k2)], ["b"])
    b1.merge(b2)
    out["merge"] = {"num_samples": b1.num_samples, "search": run_search(b1, "ATACACAAT", 0.5, False),
                    "rows": rows_of(b1)}
    dump("g8_storage.json", out)


# ------------------------- G9: the reference's own front-end (bigsi/__main__.py) -- search / bulk_search text output
def g9_frontend():
    """Runs bigsi.__main__.bigsi().search / .bulk_search unmodified.  Extra stand-ins, import-time only:
    `hug` (used purely as decorators / type annotations: a permissive object whose calls return the decorated function),
    `pyfasta.Fasta` (dict of record name -> sequence in file order), `humanfriendly` (unused on this path)."""
    import contextlib
    import io
    import tempfile

    import yaml

    class _Any(object):
        def __getattr__(self, name):
            return _Any()

        def __call__(self, *a, **k):
            if len(a) == 1 and callable(a[0]) and not k and not isinstance(a[0], _Any):
                return a[0]
            return _Any()

    sys.modules["hug"] = _Any()
    sys.modules["humanfriendly"] = types.ModuleType("humanfriendly")
    pyfasta = types.ModuleType("pyfasta")

    class Fasta(dict):
        def __init__(self, path):
            dict.__init__(self, read_fasta(path))

    pyfasta.Fasta = Fasta
    sys.modules["pyfasta"] = pyfasta
    import bigsi.__main__ as ref_main

    c = cfg("g9", 31, 1000, 3)
    c["nproc"] = 1
    ref_storage.get_storage(c).delete_all()
    base = [l.strip() for l in open(os.path.join(REF, "bigsi/tests/data/test_kmers.txt")) if l.strip()]
    q_example = read_fasta(os.path.join(REF, "example-data/query.fasta"))
    q_test = read_fasta(os.path.join(REF, "bigsi/tests/data/query.fasta"))
    samples = {"s1": base, "s2": base[:50] + list(seq_to_kmers(q_example[0][1], 31)),
               'we,ird "name"': list(seq_to_kmers(q_test[0][1], 31)) + list(seq_to_kmers(q_example[3][1], 31))}
    blooms = [BIGSI.bloom(c, ks) for ks in samples.values()]
    BIGSI.build(c, blooms, list(samples.keys()))
    out = {"k": 31, "m": 1000, "h": 3, "samples": samples, "cases": []}
    fastas = {"example": os.path.join(REF, "example-data/query.fasta"), "tests": os.path.join(REF, "bigsi/tests/data/query.fasta")}
    out["fasta_text"] = {k: open(v).read() for k, v in fastas.items()}
    with tempfile.TemporaryDirectory() as td:
        cf = os.path.join(td, "c.yaml")
        with open(cf, "w") as f:
            yaml.safe_dump(c, f)
        api = ref_main.bigsi()
        for thr in (1.0, 0.4):
            for score in (False, True):
                for fmt in ("json", "csv"):
                    for seq in (base[0] + base[1][:5], q_test[0][1], q_example[0][1][:80]):
                        out["cases"].append({"cmd": "search", "seq": seq, "threshold": thr, "score": score, "format": fmt,
                                             "out": api.search(seq, thr, cf, score, fmt)})
                    for fname, fpath in fastas.items():
                        if score and fname == "example" and thr != 0.4:
                            continue      # keep the fixture small: 20 records x presence strings
                        out["cases"].append({"cmd": "bulk_search", "fasta": fname, "threshold": thr, "score": score, "format": fmt,
                                             "stream": False, "out": api.bulk_search(fpath, thr, cf, score, fmt, False)})
                        buf = io.StringIO()
                        with contextlib.redirect_stdout(buf):
                            ret = api.bulk_search(fpath, thr, cf, score, fmt, True)
                        out["cases"].append({"cmd": "bulk_search", "fasta": fname, "threshold": thr, "score": score, "format": fmt,
                                             "stream": True, "out": ret, "stdout": buf.getvalue()})
    dump("g9_frontend.json", out, indent=None)


# ------------------------------- G10: k-mers of the reference's Cortex (.ctx) graph files
def g10_cortex():
    import base64
    from bigsi.utils.cortex import GraphReader, extract_kmers_from_ctx
    out = []
    for rel in ("bigsi/tests/data/test_kmers.ctx", "example-data/test1.ctx", "example-data/test2.ctx"):
        path = os.path.join(REF, rel)
        gr = GraphReader(path)
        out.append({"file": rel, "ctx_base64": base64.b64encode(open(path, "rb").read()).decode(),
                    "kmer_size": gr.kmer_size, "num_colours": gr.num_colours, "num_records": gr.num_records,
                    "kmers_k31": list(extract_kmers_from_ctx(path, 31)), "kmers_k21": list(extract_kmers_from_ctx(path, 21))})
    dump("g10_cortex.json", out, indent=None)


# ------------------------------- G11: the reference's example index files (data, copied verbatim)
def g11_example_bdb_files():
    """example-data/test-bigsi/{graph,metadata}: a v0.1-format BerkeleyDB pair the reference ships as example data
    (scripts/convert_v01_to_v03.py documents its layout).  Used to pin bigsi_amd/bdb.py on real BerkeleyDB files."""
    import shutil
    for name in ("graph", "metadata"):
        dst = os.path.join(HERE, "bdb_v01_%s.db" % name)
        shutil.copyfile(os.path.join(REF, "example-data/test-bigsi", name), dst)
        os.chmod(dst, 0o644)
        print("copied", dst)


# ------------------------------- G12: variant genotyping (cmds/variant_search.py) over exact searches
def g12_variant():
    """Runs BIGSIVariantSearch / BIGSIAminoAcidMutationSearch and the `variant_search` command of bigsi.__main__
    unmodified.  The external probe generator (`mykrobe variants make-probes`, a subprocess in the reference) is not
    installed, so `create_variant_probe_set` is patched on the instance/class to return a canned probe FASTA -- the
    searches, the ref/alt split by record name, genotype calls and output text are the reference's.  The reference emits
    results in set-iteration order (varies with PYTHONHASHSEED); the fixture stores them sorted by sample name."""
    import tempfile

    import yaml
    import bigsi.__main__ as ref_main          # stand-ins for hug / pyfasta installed by g9_frontend()
    from bigsi.cmds import variant_search as ref_vs
    rng = np.random.default_rng(12)
    k, m, h = 21, 5000, 3
    c = cfg("g12", k, m, h)
    ref_storage.get_storage(c).delete_all()
    genome = "".join(rng.choice(list("ACGT"), size=400))
    other = "".join(rng.choice(list("ACGT"), size=400))

    def mutate(g, pos, base):
        return g[:pos] + base + g[pos + 1:]

    variants = []
    for pos in (60, 150, 151, 300):
        refb = genome[pos]
        alts = [b for b in "ACGT" if b != refb]
        variants.append((pos, refb, alts))
    p0, r0, a0 = variants[0]
    p1, r1, a1 = variants[1]
    p3, r3, a3 = variants[3]
    samples = {
        "wildtype": [genome],
        "snp60": [mutate(genome, p0, a0[0])],
        "het60": [genome, mutate(genome, p0, a0[0])],
        "snp150_second_alt": [mutate(genome, p1, a1[1])],
        "double": [mutate(mutate(genome, p0, a0[0]), p3, a3[2])],
        "unrelated": [other],
        "partial": [genome[:p3 - 3]],
    }
    blooms = [BIGSI.bloom(c, [km for s in seqs for km in seq_to_kmers(s, k)]) for seqs in samples.values()]
    b = BIGSI.build(c, blooms, list(samples.keys()))
    b.delete_sample("unrelated")

    def probe(g, pos):
        return g[pos - k + 1: pos + k]

    def norm(res):
        return sorted(res, key=lambda r: r["sample_name"])

    out = {"k": k, "m": m, "h": h, "samples": samples, "deleted": ["unrelated"], "cases": [], "cli": []}
    with tempfile.TemporaryDirectory() as td:
        cf = os.path.join(td, "c.yaml")
        with open(cf, "w") as f:
            yaml.safe_dump(c, f)
        for pos, refb, alts in variants:
            refs = [probe(genome, pos)]
            for alt_base, alt_list in [(a, [a]) for a in alts] + [("X", alts)]:
                alt_seqs = [probe(mutate(genome, pos, a), pos) for a in alt_list]
                fasta = "".join(">ref-%s%d%s?var_name=%s%d%s&num_alts=%d\n%s\n" % (refb, pos, alt_base, refb, pos, alt_base, len(alt_seqs), r)
                                for r in refs)
                fasta += "".join(">alt-%s%d%s-%d\n%s\n" % (refb, pos, alt_base, i, a) for i, a in enumerate(alt_seqs))
                vs = ref_vs.BIGSIVariantSearch(b, "ref.fa")
                vs.create_variant_probe_set = lambda var_name, _f=fasta: _f.encode()
                d = vs.search(refb, pos, alt_base)
                assert norm(d["results"]) == norm(vs.genotype_alleles(refs, alt_seqs))
                out["cases"].append({"ref_base": refb, "pos": pos, "alt_base": alt_base, "probes_fasta": fasta,
                                     "refs": refs, "alts": alt_seqs, "query": d["query"], "results": norm(d["results"])})
        # the command (json and csv), and the amino-acid flavour (gene + genbank), probe sets patched at class level
        case = out["cases"][0]
        saved = (ref_vs.BIGSIVariantSearch.create_variant_probe_set, ref_vs.BIGSIAminoAcidMutationSearch.create_variant_probe_set)
        ref_vs.BIGSIVariantSearch.create_variant_probe_set = lambda self, var_name: case["probes_fasta"].encode()
        ref_vs.BIGSIAminoAcidMutationSearch.create_variant_probe_set = lambda self, var_name: case["probes_fasta"].encode()
        try:
            api = ref_main.bigsi()
            for fmt in ("json", "csv"):
                for gene, genbank in ((None, None), ("rpoB", "x.gb")):
                    text = api.variant_search("ref.fa", case["ref_base"], case["pos"], case["alt_base"], gene, genbank, cf, fmt)
                    if fmt == "json":
                        d = json.loads(text)
                        d["results"] = norm(d["results"])
                        keys = list(json.loads(text).keys())
                        rec = {"parsed": d, "key_order": keys, "indent4": text.startswith('{\n    "')}
                    else:
                        lines = text.split("\r\n")
                        rec = {"header": lines[0], "rows_sorted": sorted(lines[1:-1]), "tail": lines[-1]}
                    out["cli"].append({"format": fmt, "gene": gene, "genbank": genbank, "case": 0, "out": rec})
            try:
                api.variant_search("ref.fa", "A", 1, "T", "rpoB", None, cf, "json")
            except Exception as e:
                out["gene_without_genbank"] = type(e).__name__
        finally:
            ref_vs.BIGSIVariantSearch.create_variant_probe_set, ref_vs.BIGSIAminoAcidMutationSearch.create_variant_probe_set = saved
    dump("g12_variant.json", out, indent=None)


# ------------------------------------------------ G13: non-ASCII sequences (k CHARACTERS, hashed as their UTF-8 bytes)
def g13_unicode():
    """The reference slices k characters (utils/fncts.py:63-65), reverse-complements character by character
    (fncts.py:12,38-39), compares Python strings (fncts.py:51-54) and hashes the UTF-8 bytes (bloom/bloomfilter.py:5-6)."""
    c = cfg("g13", 3, 1000, 3)
    ref_storage.get_storage(c).delete_all()
    samples = {"u1": "AT\u03b1CA\u03b2AT", "u2": "\u03a9ATACA\u03a9", "a": "ATACACAAT", "e": "G\u00e9N\u00f4ME\U0001F9ECGAT"}
    blooms = [BIGSI.bloom(c, seq_to_kmers(s, 3)) for s in samples.values()]
    b = BIGSI.build(c, blooms, list(samples.keys()))
    strings = ["AT\u03b1", "\u03b1TA", "\u03a9AT", "G\u00e9N", "ME\U0001F9EC", "\U0001F9ECGA", "ATA", "\u00e9\u00e9\u00e9"]
    case = {"k": 3, "m": 1000, "h": 3, "samples": samples, "blooms": [x.tobytes().hex() for x in blooms], "rows": rows_of(b),
            "canonical": [{"s": s, "canonical": canonical(s), "reverse_comp": reverse_comp(s),
                           "rows_in_seed_order": [ref_bloom._hash(canonical(s), sd, 1000) for sd in range(3)]} for s in strings],
            "lookups": [], "searches": []}
    for kms in (["AT\u03b1", "ATA"], ["\u03a9AT", "TA\u03a9", "G\u00e9N"], ["\U0001F9ECGA"]):
        for rtz in (True, False):
            case["lookups"].append({"kmers": kms, "remove_trailing_zeros": rtz, "lookup": lookup_dict(b.lookup(kms, rtz))})
    seqs = ["AT\u03b1CA\u03b2AT", "\u03a9ATACA\u03a9", "AT\u03b1CA", "G\u00e9N\u00f4ME\U0001F9ECGAT", "\u00f4ME\U0001F9ECGATACA", "\u03b1\u03b2",
            "AT\u03b1", "ATACACAAT\u03b1", "\u00e9\u00e9\u00e9\u00e9\u00e9"]
    for s in seqs:
        for t in (1.0, 0.5, 0.3, 0.0):
            for sc in (False, True):
                case["searches"].append({"seq": s, "threshold": t, "score": sc, "out": run_search(b, s, t, sc)})
    dump("g13_unicode.json", case)


if __name__ == "__main__":
    if sys.argv[1:] == ["g13"]:       # only the fixture added in round 2 (the others are byte-identical re-runs)
        g13_unicode()
        sys.exit(0)
    g1_hash()
    g2_lookup()
    g3_search()
    g4_config1()
    g5_scoring()
    g6_arith()
    g7_random()
    g8_storage()
    g9_frontend()
    g10_cortex()
    g11_example_bdb_files()
    g12_variant()
    g13_unicode()
