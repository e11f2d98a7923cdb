#!/opt/conda/bin/python3.9
"""Run the UNMODIFIED reference -- its own test-suite and its own classes -- on top of the C ABI, through the ctypes stub exactly
as INTEGRATION.md prints it, bound to the CPU twin (libbigsi_cpu.so: the same entry points, no GPU), and record what crossed the
storage boundary as golden G14.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/run_reference_suite.py [--write]

Build container only (/root/reference does not exist on the GPU box; neither this script nor the reference travel -- G14 does).
Test infrastructure: it imports the reference package, copies none of its source.  Same interpreter, stand-ins and harness patches
as make_golden.py (mmh3 -> sklearn's MurmurHash3, empty `redis`, H1 zeroed BloomFilter, H2 bit positions), plus an empty
`hypothesis.strategies` (bigsi/tests/base.py:1 builds module-level strategies none of the suites below draws from).

What runs, twice -- PLAIN (the backend registered, nothing else changed: every layer of the reference
above storage/base.py is the reference's) and FUSED (step 3's two dispatch edits applied as monkeypatches: BIGSI.search and
KmerSignatureIndex.lookup hand whole queries to the library):

  A. the reference's suites, collected and run by pytest from where they lie:
       bigsi/tests/storage/test_storage.py, bigsi/tests/matrix/test_bitmatrix.py,
       bigsi/tests/graph/test_index.py, test_metadata.py, test_end_to_end.py        with bigsi.tests.base.CONFIGS = [hip-hbm]
     and the two tests the reference itself skips (test_inexact_search "passes in isolation", test_merge "TODO single config"),
     called directly -- in isolation, with three hip-hbm configs -- because on this backend they can run;
  B. the reference's BIGSI over the stub against what the same BIGSI produced over the dict store when G3 / G4 / G7 were made
     (tests/golden/g3_search.json, g4_config1.json, g7_random.json): every search, threshold, score=True included, every lookup,
     the stored rows, the deleted-sample behaviour -- results must be EQUAL (floats included: same interpreter, same scorer).

Every call that reaches the backend's primitives (set / get / batch_set / batch_get / delete_all / lookup_kmers / search_batch) is
recorded with its result: tests/golden/g14_reference_suite.json.gz.  tests/test_reference_suite_replay.py replays that trace
through the same stub text -- on the CPU twin in the CPU suite, on libbigsi_hip.so in the `-m gpu` suite -- and every result must
come back byte for byte.
"""
import gzip
import json
import os
import re
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)

hyp, strat = types.ModuleType("hypothesis"), types.ModuleType("hypothesis.strategies")
for _n in ("sampled_from", "just", "text", "integers"):
    setattr(strat, _n, lambda *a, **k: None)
hyp.strategies = strat
sys.modules["hypothesis"], sys.modules["hypothesis.strategies"] = hyp, strat

import make_golden as mg  # noqa: E402  (stand-ins + harness patches + the reference on sys.path)

import numpy as np  # noqa: E402
import pytest  # noqa: E402
from bitarray import bitarray  # noqa: E402

import bigsi.storage as ref_storage  # noqa: E402
import bigsi.tests.base as ref_base  # noqa: E402
from bigsi import BIGSI  # noqa: E402
from bigsi.graph.index import KmerSignatureIndex  # noqa: E402
from bigsi.utils import seq_to_kmers  # noqa: E402

REF = mg.REF
TWIN = os.path.join(ROOT, "bigsi_amd", "libbigsi_cpu.so")


# ------------------------------------------------------------------ the stub, as INTEGRATION.md prints it
def integration_block(marker):
    """The fenced code block that follows `<!-- marker -->` in INTEGRATION.md."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- %s -->\s*```python\n(.*?)\n```" % re.escape(marker), text, re.S)
    assert m, "INTEGRATION.md has no block marked %s" % marker
    return m.group(1)


def load_stub():
    os.environ["BIGSI_HIPHBM_LIBRARY"] = TWIN
    mod = types.ModuleType("bigsi.storage.hiphbm")
    mod.__file__ = "INTEGRATION.md#hiphbm.py"
    sys.modules["bigsi.storage.hiphbm"] = mod
    exec(compile(integration_block("stub:bigsi/storage/hiphbm.py"), mod.__file__, "exec"), mod.__dict__)
    ref_storage.hiphbm = mod
    # step 1: the registration lines, executed inside bigsi/storage/__init__.py's namespace
    exec(compile(integration_block("edit:bigsi/storage/__init__.py"), "INTEGRATION.md#register", "exec"), ref_storage.__dict__)
    assert ref_storage.STORAGE_DICT["hip-hbm"] is mod.HipHbmStorage
    return mod


# ------------------------------------------------------------------ recording what crosses the boundary
TRACE = []
STORES = {}          # name -> storage-config, as the reference's get_storage() passed it
_depth = [0]


def _ids(rows):
    rows = [int(r) for r in rows]
    if len(rows) > 2 and rows == list(range(rows[0], rows[0] + len(rows))):
        return {"from": rows[0], "n": len(rows)}
    return rows


def _key(k):
    return k.decode("utf-8") if isinstance(k, bytes) else k


def _outcome(fn):
    try:
        return fn(), None
    except BaseException as e:  # noqa: BLE001 -- the exception type is part of the contract
        return None, e


def record(stub):
    """Wrap the backend's primitives: outermost calls only (batch_set falls back to __setitem__ for non-row keys)."""
    cls, row = stub.HipHbmStorage, stub._ROW

    def wrap(name, encode):
        inner = getattr(cls, name)

        def outer(self, *args):
            if _depth[0]:
                return inner(self, *args)
            STORES.setdefault(self.storage_config.get("name", "default"), dict(self.storage_config))
            args = tuple(list(a) if name.startswith("batch_") else a for a in args)      # (generators: consumed once)
            _depth[0] += 1
            try:
                res, err = _outcome(lambda: inner(self, *args))
            finally:
                _depth[0] -= 1
            TRACE.append(encode(self.storage_config.get("name", "default"), args, res if err is None else {"raises": type(err).__name__}))
            if err is not None:
                raise err
            return res
        setattr(cls, name, outer)

    def enc_rows(op, store, keys, blobs):
        ms = [row.match(k if isinstance(k, bytes) else k.encode()) for k in keys]
        if keys and all(ms) and not isinstance(blobs, dict) and len({len(b) for b in blobs}) == 1:
            return [op + "_rows", store, _ids(m.group(1) for m in ms), len(blobs[0]), b"".join(blobs).hex()]
        return [op, store, [_key(k) for k in keys], blobs if isinstance(blobs, dict) else [b.hex() for b in blobs]]

    wrap("__setitem__", lambda s, a, r: ["set", s, _key(a[0]), bytes(a[1]).hex()] + ([r] if r else []))
    wrap("__getitem__", lambda s, a, r: ["get", s, _key(a[0]), r if isinstance(r, dict) else r.hex()])
    wrap("batch_set", lambda s, a, r: enc_rows("mset", s, a[0], [bytes(v) for v in a[1]]) + ([r] if r else []))
    wrap("batch_get", lambda s, a, r: enc_rows("mget", s, a[0], r if isinstance(r, dict) else [bytes(v) for v in r]))
    wrap("delete_all", lambda s, a, r: ["delete_all", s] + ([r] if r else []))
    wrap("lookup_kmers", lambda s, a, r: ["lookup", s, list(a[0]), a[1], r if "raises" in r else {k: v.hex() for k, v in r.items()}])
    wrap("search_batch", lambda s, a, r: ["search", s, list(a[0]), a[1], a[2], bool(a[3]) if len(a) > 3 else False,
                                          r if isinstance(r, dict) else [list(x) for x in r]])


# ------------------------------------------------------------------ step 3 as monkeypatches (the three added lines of each edit)
def apply_fused_dispatch():
    orig_lookup, orig_search = KmerSignatureIndex.lookup, BIGSI.search

    def lookup(self, kmers, remove_trailing_zeros=True):
        if isinstance(kmers, str):
            kmers = [kmers]
        kmers = set(kmers)
        fused = getattr(self.storage, "fused_lookup", None)                               # + graph/index.py, after line 46
        found = fused(self, kmers, remove_trailing_zeros) if fused else None              # +
        if found is not None:                                                             # +
            return found                                                                  # +
        return orig_lookup(self, kmers, remove_trailing_zeros)

    def search(self, seq, threshold=1.0, score=False):
        self._BIGSI__validate_search_query(seq)
        assert threshold <= 1
        fused = getattr(self.storage, "fused_search", None)                               # + graph/bigsi.py, after line 176
        found = fused(self, seq, threshold, score) if fused else None                     # +
        if found is not None:                                                             # +
            return found                                                                  # +
        return orig_search(self, seq, threshold, score)

    KmerSignatureIndex.lookup, BIGSI.search = lookup, search

    def undo():
        KmerSignatureIndex.lookup, BIGSI.search = orig_lookup, orig_search
    return undo


# ------------------------------------------------------------------ A: the reference's own suites
SUITES = ["bigsi/tests/storage/test_storage.py", "bigsi/tests/matrix/test_bitmatrix.py", "bigsi/tests/graph/test_index.py",
          "bigsi/tests/graph/test_metadata.py", "bigsi/tests/graph/test_end_to_end.py"]


def hip_config(name):
    return {"storage-engine": "hip-hbm", "storage-config": {"name": name, "max_cols": 64}, **ref_base.PARAMETERS}


class Outcomes(object):
    def __init__(self):
        self.rows = []

    def pytest_runtest_logreport(self, report):
        if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
            self.rows.append([report.nodeid.replace("::", ":"), report.outcome])


def run_suites(tag):
    ref_base.CONFIGS[:] = [hip_config("ref-%s" % tag)]
    assert [repr(s) for s in ref_base.get_test_storages()] == ["hip-hbm Storage"]
    out = Outcomes()
    rc = pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", REF, "-o", "python_files=test_*.py"] + [os.path.join(REF, s) for s in SUITES],
                     plugins=[out])
    rows = [[r[0].split("bigsi/tests/")[-1], r[1]] for r in out.rows]
    assert int(rc) == 0, "the reference's suites failed on the hip-hbm stub (%s): %r" % (tag, rows)
    # the two tests the reference skips, in isolation (their own words), on three hip-hbm stores
    import bigsi.tests.graph.test_end_to_end as e2e
    ref_base.CONFIGS[:] = [hip_config("ref-%s-%d" % (tag, i)) for i in range(3)]
    for fn in (e2e.test_inexact_search, e2e.test_merge):
        fn()
        rows.append(["graph/test_end_to_end.py:%s (skipped by the reference; called directly)" % fn.__name__, "passed"])
    ref_base.CONFIGS[:] = []
    return rows


# ------------------------------------------------------------------ B: the reference's BIGSI over the stub == over the dict store
def golden(name):
    return json.load(open(os.path.join(HERE, name)))


def same(a, b, what):
    assert a == b, "%s differs\n  stub: %r\n  dict: %r" % (what, a, b)


def check_rows(b, case, what):
    n = (len(case["blooms"]) + 7) // 8
    got = [v[:n].hex() for v in b.storage.batch_get(b.storage.convert_bitarray_batch_keys(range(b.bloomfilter_size)))]
    same(got, [x[:2 * n] for x in case["rows"]], what + " stored rows")


def replay_goldens(tag):
    n = 0
    # G3: search semantics + edge cases + a deleted sample
    g = golden("g3_search.json")
    c = hip_config("g3-" + tag)
    c.update(k=g["k"], m=g["m"], h=g["h"])
    ref_storage.get_storage(c).delete_all()
    b = BIGSI.build(c, [BIGSI.bloom(c, seq_to_kmers(s, g["k"])) for s in g["samples"].values()], list(g["samples"]))
    check_rows(b, g, "G3")
    for s in g["searches"]:
        t = int(s["threshold"]) if s["threshold_is_int"] else s["threshold"]
        same(mg.run_search(b, s["seq"], t, s["score"]), s["out"], "G3 search %r t=%r score=%r" % (s["seq"], t, s["score"]))
        n += 1
    b.delete_sample("a")
    d = g["after_delete_a"]
    same([b.num_samples, b.colour_to_sample(0), b.sample_to_colour("a")], [d["num_samples"], d["colour_to_sample_0"], d["sample_to_colour_a"]], "G3 delete")
    for s in d["searches"]:
        same(mg.run_search(b, s["seq"], s["threshold"], s["score"]), s["out"], "G3 after delete %r" % s["seq"])
        n += 1
    b.delete()
    # G4: BASELINE configs[0]'s shape (k = 31, m = 1000, h = 3), the reference's fixture k-mers and query files
    g = golden("g4_config1.json")
    c = hip_config("g4-" + tag)
    c.update(k=g["k"], m=g["m"], h=g["h"])
    ref_storage.get_storage(c).delete_all()
    b = BIGSI.build(c, [BIGSI.bloom(c, ks) for ks in g["samples"].values()], list(g["samples"]))
    check_rows(b, g, "G4")
    for s in g["searches"]:
        same(mg.run_search(b, s["seq"], s["threshold"], s["score"]), s["out"], "G4 search %s t=%r score=%r" % (s["name"], s["threshold"], s["score"]))
        n += 1
    b.delete()
    # G7: 200 samples (not a multiple of 8 or 64), 4096 rows, k = 31: lookups as stored bytes, searches incl. scores
    g = golden("g7_random.json")
    c = hip_config("g7-" + tag)
    c.update(k=g["k"], m=g["m"], h=g["h"])
    ref_storage.get_storage(c).delete_all()
    blooms = [BIGSI.bloom(c, [km for f in frags for km in seq_to_kmers(f, g["k"])]) for frags in g["sample_seqs"]]
    b = BIGSI.build(c, blooms, g["sample_names"])
    nb = (g["n_cols"] + 7) // 8
    for lk in g["lookups"]:
        got = b.lookup(list(seq_to_kmers(lk["seq"], g["k"])), remove_trailing_zeros=False)
        same({km: got[km].tobytes()[:nb].hex() for km in sorted(got)}, {km: v[:2 * nb] for km, v in lk["lookup"].items()}, "G7 lookup")
        assert all(not any(got[km][g["n_cols"]:]) for km in got)
        n += 1
    for s in g["searches"]:
        same(mg.run_search(b, g["queries"][s["q"]], s["threshold"], s["score"]), s["out"], "G7 search q%d t=%r score=%r" % (s["q"], s["threshold"], s["score"]))
        n += 1
    b.delete()
    # G2: lookup dicts incl. remove_trailing_zeros both ways, non-ACGT and lowercase k-mers
    for g in golden("g2_lookup.json"):
        c = hip_config("g2-" + tag)
        c.update(k=g["k"], m=g["m"], h=g["h"])
        ref_storage.get_storage(c).delete_all()
        b = BIGSI.build(c, [BIGSI.bloom(c, ks) for ks in g["samples"]], ["s1", "s2"])
        check_rows(b, g, "G2")
        for lk in g["lookups"]:
            got = b.lookup(lk["kmers"], remove_trailing_zeros=lk["remove_trailing_zeros"])
            want = lk["result"]
            if lk["remove_trailing_zeros"]:
                same(mg.lookup_dict(got), want, "G2 lookup %r" % (lk["kmers"],))
            else:       # rows as wide as the store hands them out: the columns that exist must agree, the padding must be zero
                same({k: v.to01()[:2] for k, v in got.items()}, {k: v[:2] for k, v in want.items()}, "G2 lookup (untrimmed) %r" % (lk["kmers"],))
                assert all(not any(v[2:]) for v in got.values())
            n += 1
        b.delete()
    return n


def main():
    import logging
    logging.disable(logging.CRITICAL)       # (the reference logs every build step at DEBUG and warns about every short query)
    stub = load_stub()
    record(stub)
    import hashlib
    report = {"library": os.path.basename(TWIN),
              "integration_md_sha256": {m: hashlib.sha256(integration_block(m).encode()).hexdigest()
                                        for m in ("stub:bigsi/storage/hiphbm.py", "edit:bigsi/storage/__init__.py")},
              "interpreter": "python %d.%d, numpy %s, bitarray %s, pytest %s" % (
        sys.version_info[0], sys.version_info[1], np.__version__, __import__("bitarray").__version__, pytest.__version__), "phases": {}}
    for tag in ("plain", "fused"):
        undo = apply_fused_dispatch() if tag == "fused" else (lambda: None)
        first = len(TRACE)
        try:
            suites = run_suites(tag)
            compared = replay_goldens(tag)
        finally:
            undo()
        ops = TRACE[first:]
        calls = {}
        for op in ops:
            calls[op[0]] = calls.get(op[0], 0) + 1
        if tag == "plain":
            assert not calls.get("search") and not calls.get("lookup")
        else:
            assert calls.get("search", 0) > 300 and calls.get("lookup", 0) > 20, calls
        report["phases"][tag] = {"reference_tests": suites, "golden_results_compared_equal": compared, "boundary_calls": calls}
        tally = {o: sum(1 for r in suites if r[1] == o) for o in sorted({r[1] for r in suites})}
        assert set(tally) <= {"passed", "skipped"} and tally.get("skipped", 0) == 2, tally      # (the 2 skips are the reference's own marks; both are then called directly)
        print("%s: the reference's tests %s, %d golden results equal, boundary calls %s" % (tag, tally, compared, calls))
    fixture = {"about": "what crossed the hip-hbm storage boundary while the unmodified reference ran its own suites and the G2/G3/G4/G7 "
                        "workloads over INTEGRATION.md's stub bound to libbigsi_cpu.so (tests/golden/run_reference_suite.py)",
               "report": report, "stores": STORES, "trace": TRACE}
    raw = json.dumps(fixture, separators=(",", ":")).encode()
    print("trace: %d boundary calls, %.1f KB of JSON" % (len(TRACE), len(raw) / 1e3))
    if "--write" in sys.argv:
        path = os.path.join(HERE, "g14_reference_suite.json.gz")
        with open(path, "wb") as f:
            with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as z:
                z.write(raw)
        json.dump(report, open(os.path.join(HERE, "g14_report.json"), "w"), indent=1)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
