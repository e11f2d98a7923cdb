"""CPU baseline of the query path -- TEST/BENCH INFRASTRUCTURE (run by bench.py's cpu_baseline leg as a subprocess).

Times oracle/bigsi_oracle.c's reference-shaped query (per-k-mer canonicalisation, MurmurHash3 x h, per-row copy + AND,
then AND-all or unpack-to-int32-and-add; graph/index.py:62-80, graph/bigsi.py:35-44,192-195) on the host cores:
  1. one core, for `--seconds`;
  2. a fork pool of `--threads` workers over query sequences, the reference's only parallelism (bulk_search,
     bigsi/__main__.py:273-287), each worker cycling over its share of the queries for `--seconds`.
The index is the same seeded synthetic one as on the GPU at full row width, but only --rows rows so that it fits host
RAM (per-lookup work is identical; rows are served from RAM, which favours the CPU over the reference's BerkeleyDB).
Prints one JSON object."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coracle  # noqa: E402

_T = {}


def _fill(job):
    r0, n = job
    a = _T["args"]
    return r0, coracle.synth_fill(a.seed, 0, r0, n, a.cols, a.and_draws)


def _work(job):
    wid, seconds = job
    a, table, seqs = _T["args"], _T["table"], _T["seqs"]
    mine = seqs[wid::a.threads] or seqs
    done, t0, i = 0, time.time(), 0
    while time.time() - t0 < seconds:
        u, _, _ = coracle.query(table, a.hashes, mine[i % len(mine)], a.k, want_counts=not a.exact, want_and=bool(a.exact))
        done += u
        i += 1
    return done, time.time() - t0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=200_000)
    p.add_argument("--cols", type=int, required=True)
    p.add_argument("--hashes", type=int, required=True)
    p.add_argument("--k", type=int, default=31)
    p.add_argument("--and-draws", type=int, default=2)
    p.add_argument("--seed", type=int, required=True)
    p.add_argument("--batch", type=int, required=True)
    p.add_argument("--qlen", type=int, required=True)
    p.add_argument("--exact", type=int, default=1)
    p.add_argument("--seconds", type=float, default=10.0)
    p.add_argument("--threads", type=int, default=0)
    p.add_argument("--pool-runs", type=int, default=1, help="repetitions of the pool leg (its rate moves from run to run)")
    p.add_argument("--pool-seconds", type=float, default=0.0, help="duration of each pool run (default: --seconds)")
    a = p.parse_args()
    if a.threads <= 0:
        a.threads = max(1, (os.cpu_count() or 2) // 2)      # one worker per physical core (SMT siblings share the FPU/LSU)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)              # same queries as bench.py: rand_seqs(default_rng(1), ...)
    seqs = [lut[r].tobytes().decode("ascii") for r in np.random.default_rng(1).integers(0, 4, size=(a.batch, a.qlen), dtype=np.uint8)]
    _T.update(args=a, seqs=seqs)
    coracle.lib()
    ctx = mp.get_context("fork")
    t0 = time.time()
    step = max(1, a.rows // (4 * a.threads))
    jobs = [(r0, min(step, a.rows - r0)) for r0 in range(0, a.rows, step)]
    table = np.empty((a.rows, (a.cols + 7) // 8), dtype=np.uint8)
    with ctx.Pool(min(a.threads, len(jobs))) as pool:
        for r0, part in pool.imap_unordered(_fill, jobs):
            table[r0:r0 + part.shape[0]] = part
    fill_s = time.time() - t0
    _T["table"] = table
    one_done, one_t = _work((0, a.seconds))
    pool_s = a.pool_seconds or a.seconds
    rates = []
    with ctx.Pool(a.threads) as pool:                        # workers inherit the table copy-on-write
        for _ in range(max(1, a.pool_runs)):
            res = pool.map(_work, [(w, pool_s) for w in range(a.threads)])
            rates.append(sum(r[0] for r in res) / max(r[1] for r in res))
    print(json.dumps({"one_core": {"lookups": one_done, "seconds": one_t, "rate": one_done / one_t},
                      "pool": {"threads": a.threads, "seconds": pool_s, "rates": rates, "rate": float(np.median(rates)),
                               "rate_median": float(np.median(rates)), "rate_best": max(rates)},
                      "rows": a.rows, "cols": a.cols, "fill_seconds": fill_s, "host_threads": os.cpu_count()}))


if __name__ == "__main__":
    main()
