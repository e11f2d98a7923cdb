import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unjson(x):
    """Undo make_golden.jsonable's encoding of non-finite floats."""
    if isinstance(x, dict):
        if set(x.keys()) == {"__float__"}:
            return float(x["__float__"])
        return {k: unjson(v) for k, v in x.items()}
    if isinstance(x, list):
        return [unjson(v) for v in x]
    return x


# Score fields the reference leaves unrounded and computes with numpy transcendental functions
# (scoring/score.py:125-132): compared with a relative tolerance of 1e-12 because np.exp may differ in the
# last ulp between the numpy that generated the goldens (1.26) and the one running the tests (2.x).
# Every other field (integers, strings, Python/numpy round()-ed values) is compared exactly.
# pvalue = 1 - np.exp(-evalue) cancels catastrophically for tiny evalues, so a 1-ulp difference in exp()
# near 1.0 (observed: numpy 1.26 returns 1-1.1e-16 for exp(-1.7e-16), numpy 2.2 the correctly rounded
# 1-2.2e-16) moves it by 1.1e-16 absolute: pvalue gets an absolute tolerance of 2 ulp(1.0) = 2.5e-16.
FLOAT_TOL_KEYS = {"evalue": dict(rel=1e-12, abs=0.0), "pvalue": dict(rel=1e-12, abs=2.5e-16)}


def assert_result_equal(got, want, ctx=""):
    assert list(got.keys()) == list(want.keys()), "%s key order: %r vs %r" % (ctx, list(got), list(want))
    for k in want:
        g, w = got[k], want[k]
        if k in FLOAT_TOL_KEYS:
            assert g == pytest.approx(w, **FLOAT_TOL_KEYS[k]), "%s %s: %r vs %r" % (ctx, k, g, w)
        else:
            assert g == w and type(g).__name__.replace("float64", "float") == type(w).__name__.replace("float64", "float"), \
                "%s %s: %r vs %r" % (ctx, k, g, w)


def assert_results_equal(got, want, ctx=""):
    assert len(got) == len(want), "%s: %d results vs %d\n%r\n%r" % (ctx, len(got), len(want), got[:3], want[:3])
    for i, (g, w) in enumerate(zip(got, want)):
        assert_result_equal(g, w, "%s[%d]" % (ctx, i))


def check_search(fn, case, ctx=""):
    """Run fn() and compare with a golden 'out' record ({'results': [...]} or {'raises': name})."""
    out = case["out"]
    if "raises" in out:
        with pytest.raises(BaseException) as ei:
            fn()
        assert type(ei.value).__name__ == out["raises"], "%s raised %r, want %s" % (ctx, ei.value, out["raises"])
    else:
        assert_results_equal(fn(), unjson(out["results"]), ctx)
