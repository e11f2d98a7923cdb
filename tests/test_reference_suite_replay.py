"""G14: the reference's own suites, replayed at the storage boundary.

tests/golden/run_reference_suite.py ran the UNMODIFIED reference -- bigsi/tests/storage, tests/matrix/test_bitmatrix.py,
tests/graph/{test_index,test_metadata,test_end_to_end}.py, then its BIGSI class over the G2 / G3 / G4 / G7 workloads -- on top of
the ctypes stub that INTEGRATION.md prints (bound to libbigsi_cpu.so), plain and with the two fused dispatch edits, and recorded
every call that reached the backend's primitives together with its result (tests/golden/g14_reference_suite.json.gz).  Here the SAME
stub text, cut out of INTEGRATION.md again, is driven through that trace and must give every result back byte for byte:

  * `-m "not gpu"`: bound to the CPU twin (the recording replays under this interpreter / numpy too, and the printed stub is still
    the text the recording was made with);
  * `-m gpu`: bound to libbigsi_hip.so -- rows, gathers, lookups and hit lists (exact, thresholded, scored presence strings) of
    the reference's suites from the HIP path.

The reference itself is not needed (and does not exist on the GPU box): the one name the stub imports from it, BaseStorage of
bigsi/storage/base.py, is provided by this package's mirror of the contract (bigsi_amd/storage/contract.py)."""
import gzip
import hashlib
import json
import os
import re
import sys
import types

import pytest

from conftest import GOLDEN, ROOT

TWIN = os.path.join(ROOT, "bigsi_amd", "libbigsi_cpu.so")
HIP = os.path.join(ROOT, "bigsi_amd", "libbigsi_hip.so")


def integration_block(marker):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- %s -->\s*```python\n(.*?)\n```" % re.escape(marker), text, re.S)
    assert m, "INTEGRATION.md has no block marked %s" % marker
    return m.group(1)


def load_stub(library):
    """INTEGRATION.md's bigsi/storage/hiphbm.py as a module bound to `library`."""
    from bigsi_amd.storage.contract import BaseStorage
    fake = {"bigsi": types.ModuleType("bigsi"), "bigsi.storage": types.ModuleType("bigsi.storage"), "bigsi.storage.base": types.ModuleType("bigsi.storage.base")}
    fake["bigsi.storage.base"].BaseStorage = BaseStorage
    saved = {k: sys.modules.get(k) for k in fake}
    old_env = os.environ.get("BIGSI_HIPHBM_LIBRARY")
    sys.modules.update(fake)
    os.environ["BIGSI_HIPHBM_LIBRARY"] = library
    try:
        mod = types.ModuleType("hiphbm_stub_%s" % os.path.basename(library).split(".")[0])
        exec(compile(integration_block("stub:bigsi/storage/hiphbm.py"), "INTEGRATION.md#hiphbm.py", "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        if old_env is None:
            os.environ.pop("BIGSI_HIPHBM_LIBRARY", None)
        else:
            os.environ["BIGSI_HIPHBM_LIBRARY"] = old_env
    return mod


@pytest.fixture(scope="module")
def g14():
    with gzip.open(os.path.join(GOLDEN, "g14_reference_suite.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def row_keys(ids):
    ids = range(ids["from"], ids["from"] + ids["n"]) if isinstance(ids, dict) else ids
    return [b"%d:bitarray" % r for r in ids]


def outcome(fn):
    try:
        return fn()
    except BaseException as e:  # noqa: BLE001 -- the exception type is what the trace holds
        return {"raises": type(e).__name__}


def replay(stub, g14):
    stores, done = {}, {}
    for n, op in enumerate(g14["trace"]):
        kind, name = op[0], op[1]
        st = stores.get(name)
        if st is None:
            st = stores[name] = stub.HipHbmStorage(g14["stores"][name])
        ctx = "op %d %s %r" % (n, kind, op[2] if len(op) > 2 and not isinstance(op[2], list) else "")
        if kind == "set":
            got = outcome(lambda: st.__setitem__(op[2], bytes.fromhex(op[3])))
            assert got == (op[4] if len(op) > 4 else None), ctx
        elif kind == "get":
            got = outcome(lambda: st[op[2]])
            assert (got if isinstance(got, dict) else got.hex()) == op[3], ctx
        elif kind == "mset_rows":
            blob, w = bytes.fromhex(op[4]), op[3]
            keys = row_keys(op[2])
            got = outcome(lambda: st.batch_set(keys, [blob[i * w:(i + 1) * w] for i in range(len(keys))]))
            assert got == (op[5] if len(op) > 5 else None), ctx
        elif kind == "mset":
            got = outcome(lambda: st.batch_set([k.encode() for k in op[2]], [bytes.fromhex(v) for v in op[3]]))
            assert got == (op[4] if len(op) > 4 else None), ctx
        elif kind == "mget_rows":
            got = outcome(lambda: st.batch_get(row_keys(op[2])))
            assert not isinstance(got, dict) and b"".join(got).hex() == op[4] and {len(v) for v in got} == {op[3]}, ctx
        elif kind == "mget":
            got = outcome(lambda: st.batch_get([k.encode() for k in op[2]]))
            assert (got if isinstance(got, dict) else [v.hex() for v in got]) == op[3], ctx
        elif kind == "delete_all":
            assert outcome(st.delete_all) == (op[2] if len(op) > 2 else None), ctx
        elif kind == "lookup":
            got = outcome(lambda: st.lookup_kmers(op[2], op[3]))
            assert (got if "raises" in got else {k: v.hex() for k, v in got.items()}) == op[4], ctx
        elif kind == "search":
            got = outcome(lambda: st.search_batch(op[2], op[3], op[4], op[5]))
            assert (got if isinstance(got, dict) else [list(x) for x in got]) == op[6], "%s %r t=%r score=%r" % (ctx, op[2], op[4], op[5])
        else:
            raise AssertionError("unknown op %r" % kind)
        done[kind] = done.get(kind, 0) + 1
    for st in stores.values():
        st.delete_all()
    return done


def test_g14_report_says_the_references_suites_passed(g14):
    rep = g14["report"]
    assert rep == json.load(open(os.path.join(GOLDEN, "g14_report.json")))
    for tag in ("plain", "fused"):
        ph = rep["phases"][tag]
        names = [t[0] for t in ph["reference_tests"]]
        # 6 + 2 + 4 + 3 + 6 of the reference's tests + the two skipped ones called directly
        assert len(names) == 23 and sum(1 for t in ph["reference_tests"] if t[1] == "passed") == 21
        assert [t[0] for t in ph["reference_tests"] if t[1] == "skipped"] == ["graph/test_end_to_end.py:test_inexact_search", "graph/test_end_to_end.py:test_merge"]
        assert sum(1 for x in names if "called directly" in x) == 2
        assert ph["golden_results_compared_equal"] == 708
    assert not rep["phases"]["plain"]["boundary_calls"].get("search") and rep["phases"]["fused"]["boundary_calls"]["search"] == 585
    assert rep["phases"]["fused"]["boundary_calls"]["lookup"] == 83


def test_the_printed_stub_is_the_recorded_stub(g14):
    """Editing INTEGRATION.md's code blocks without re-running tests/golden/run_reference_suite.py --write fails here."""
    for marker, sha in g14["report"]["integration_md_sha256"].items():
        assert hashlib.sha256(integration_block(marker).encode()).hexdigest() == sha, marker


def test_replay_on_the_cpu_twin(g14):
    assert os.path.exists(TWIN), "libbigsi_cpu.so has not been built (run __graft_entry__.build())"
    done = replay(load_stub(TWIN), g14)
    assert done["search"] == 585 and done["lookup"] == 83 and sum(done.values()) == len(g14["trace"])


@pytest.mark.gpu
def test_replay_on_the_hip_library(g14):
    stub = load_stub(HIP)
    assert stub._PREFIX == "bigsi_hip_"
    done = replay(stub, g14)
    assert done["search"] == 585 and done["lookup"] == 83 and sum(done.values()) == len(g14["trace"])
