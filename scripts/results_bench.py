"""Host-only microbenchmark of bigsi_amd/_results (no device): scored / unscored result dicts per second from synthetic stream arrays.
    python scripts/results_bench.py [hits_per_query] [queries] [positions]"""
import gc
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bigsi_amd.graph import bigsi as front      # noqa: E402
from bigsi_amd.scoring import HIT_SCORE_DTYPE   # noqa: E402

hpq = int(sys.argv[1]) if len(sys.argv) > 1 else 625
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
npos = int(sys.argv[3]) if len(sys.argv) > 3 else 970
rng = np.random.default_rng(0)
ncols = 62500
names = ["s%d" % c for c in range(ncols)]
nk = np.full(nq, npos, np.uint32)
nu = nk.copy()
nh = np.full(nq, hpq)
off = np.zeros(nq + 1, np.int64)
np.cumsum(nh, out=off[1:])
total = int(off[-1])
cols = np.concatenate([np.sort(rng.choice(ncols, hpq, replace=False)) for _ in range(nq)]).astype(np.uint32)
cnts = rng.integers(min(400, npos // 2), npos, size=total).astype(np.uint32)
rec = np.zeros(total, HIT_SCORE_DTYPE)
rec["num_kmers"] = npos
for f in ("score", "min_score", "max_score"):
    rec[f] = np.round(rng.random(total) * 900, 2)
rec["percent_kmers_found"] = np.round(rng.random(total) * 100, 2)
for f in ("max_mismatches", "min_mismatches", "mismatches"):
    rec[f] = rng.integers(0, 40, size=total)
words = (npos + 63) // 64
boff = (np.arange(total + 1, dtype=np.uint64) * np.uint64(words * 8))
bits = rng.integers(0, 256, size=int(boff[-1]), dtype=np.uint8)
for scored in (None, (rec, bits, boff)):
    best = 1e9
    for rep in range(5):
        gc.collect()
        t = time.perf_counter()
        n_d = 0
        for r in front.native_result_lists(nk, nu, off, cols, cnts, False, names, scored, ncols):      # consumed as a stream: a block's dicts are dropped before the next block is made
            n_d += len(r)
        dt = time.perf_counter() - t
        assert n_d == total
        best = min(best, dt)
    print("%s: %d hits, %.3f us per dict, %.2f M dicts/s" % ("scored" if scored else "unscored", total, best / total * 1e6, total / best / 1e6))
