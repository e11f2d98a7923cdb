#!/usr/bin/env python
"""bench.py -- BIGSI query hot path on MI355X: k-mer lookups/s and achieved HBM GB/s of the row-fetch-AND kernel.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole device path (K1 k-merise/dedupe/hash -> K2 row fetch + AND -> K4 threshold +
compaction [-> RCCL all-gather of per-sample result vectors + compaction of the gathered result when N > 1]) over one
batch of synthetic queries that is already resident in HBM.  Default workload = BASELINE.json configs[2], the largest
single-GPU configuration: synthetic 10M-row x 100k-sample index (125 GB), h=4, 256 x 1 kbp queries, threshold 1.0.
Scaling is WEAK: every rank holds a 10M x 100k column shard (index = 10M x N*100k samples) and looks every query up
in its shard; `value` sums the k-mer lookups all ranks performed (each against its own shard) per second, and
config.kmer_lookups_per_s_full_index gives the rate against the whole N-shard index.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
SEED = 20260928


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--cols", type=int, default=100_000, help="sample columns PER GPU shard")
    p.add_argument("--hashes", type=int, default=4)
    p.add_argument("--batch", type=int, default=256)
    p.add_argument("--qlen", type=int, default=1000)
    p.add_argument("--k", type=int, default=31)
    p.add_argument("--threshold", type=float, default=1.0)
    p.add_argument("--and-draws", type=int, default=2, help="bit density of the synthetic index = 2^-draws")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    p.add_argument("--cpu-rows", type=int, default=200_000)
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise the RCCL process group and take the gather path even with one rank (exercises the N>1 code)")
    return p.parse_args()


def make_queries(batch, qlen, rank_independent_seed=1):
    rng = np.random.default_rng(rank_independent_seed)       # every rank sees the same queries
    return ["".join(rng.choice(list("ACGT"), size=qlen)) for _ in range(batch)]


def cpu_baseline(args, seqs, exact):
    """oracle/cpu_baseline.py in a fresh subprocess (its fork pool must not inherit a HIP context): the C oracle
    (reference-shaped port) on ONE core -- the reported `value` -- plus, for context, the reference's own
    process-pool-over-sequences parallelism (bulk_search) on every physical core."""
    import subprocess
    half = max(args.cpu_seconds / 2.0, 1.0)
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--rows", str(min(args.rows, args.cpu_rows)),
           "--cols", str(args.cols), "--hashes", str(args.hashes), "--k", str(args.k), "--and-draws", str(args.and_draws),
           "--seed", str(SEED), "--batch", str(args.batch), "--qlen", str(args.qlen), "--exact", str(int(exact)),
           "--seconds", str(half)]
    r = json.loads(subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1])
    return {"value": r["one_core"]["rate"], "unit": "kmer_lookups/s", "cores": 1, "kind": "port",
            "sample": "%d unique k-mer lookups in %.1f s cycling over the %d bench queries, on a %d-row x %d-sample slice of the same "
                      "synthetic index (full row width, rows reduced to fit host RAM); oracle/bigsi_oracle.c orc_query; rows served from RAM "
                      "instead of BerkeleyDB" % (r["one_core"]["lookups"], r["one_core"]["seconds"], len(seqs), r["rows"], r["cols"]),
            "pool": {"value": r["pool"]["rate"], "cores": r["pool"]["threads"], "host_threads": r["host_threads"],
                     "sample": "fork pool over query sequences (the reference's bulk_search parallelism), %d workers x %.1f s"
                               % (r["pool"]["threads"], r["pool"]["seconds"])}}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    from bigsi_amd import _lib
    from bigsi_amd._lib import check
    from bigsi_amd.parallel import ShardedSearch
    from bigsi_amd.storage import get_storage

    if os.environ.get("BIGSI_BENCH_DEVICE"):          # dry runs: several ranks sharing one GPU
        local_rank = int(os.environ["BIGSI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    exact = args.threshold == 1.0
    # ---------------- index: this rank's column shard, generated on the device
    st = get_storage({"storage-engine": "hip-hbm", "k": args.k, "m": args.rows, "h": args.hashes,
                      "storage-config": {"name": "bench", "device": local_rank, "max_cols": args.cols}})
    st.delete_all()
    for key, v in (("number_of_rows", args.rows), ("number_of_cols", args.cols),
                   ("ksi:bloomfilter_size", args.rows), ("ksi:num_hashes", args.hashes)):
        st.set_integer(key, v)
    t0 = time.time()
    st.fill_synthetic(SEED, rank, args.and_draws)
    fill_s = time.time() - t0
    info = st.res.info()

    # ---------------- queries: uniform ACGT; ~1% of them planted into a few samples of every shard
    seqs = make_queries(args.batch, args.qlen)
    planted = list(range(0, args.batch, 97))[:8]
    for j, qi in enumerate(planted):
        st.insert_kmers((1009 * (j + 1) + 13 * rank) % args.cols, [seqs[qi]], args.k)
    sh = ShardedSearch(st, args.cols, device=dev, force_gather=args.force_dist)   # puts the library on a torch stream
    # consecutive steps alternate two staged copies of the batch (a serving loop's two workspaces): K1 of one overlaps the
    # row-AND kernel of the other on the library's pre stream, and in sharded runs so does the exchange
    batches = [st.new_batch(seqs, args.k) for _ in range(2)]
    batch = batches[0]
    count_bytes = 2 if (args.qlen - args.k + 1) < 65536 else 4
    sh.prepare(batches, exact, count_bytes)
    check(_lib.lib().bigsi_hip_set_profiling(st.handle, 1))

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    warm = _lib.Stats()
    for w in range(args.warmup):
        sh.step(batches, args.threshold)
        if w == 0:            # the first step pays one-off costs (code object load, allocations): keep it out of the K1 / K4 figures
            sync_all()
            check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(warm), 1))
    sync_all()
    check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(warm), 1))      # K1 / K4 durations come from the warmup steps
    # timed region: HIP events around the row-AND kernel only (every event record costs the stream 5-7 us)
    check(_lib.lib().bigsi_hip_set_profiling(st.handle, 2))
    stats = _lib.Stats()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sh.step(batches, args.threshold)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(stats), 1))

    # ---------------- results of the last step, algorithmic bytes, verification
    batch = batches[(args.steps - 1) % len(batches)]      # the batch the last step ran on
    off, colours, counts = sh.fetch(batch)
    nk, nu, mk = batch.unique()
    total_unique = int(nu.sum())
    wv = -(-args.cols // 64)
    uniq_rows = 0
    for i in range(args.batch):
        uniq_rows += np.unique(batch.rows(i, nu[i])).size      # each needed row counted once (reference fetches the union once)
    # result vector the kernel stores: one bit per sample (AND bitmap / thresholded hit mask); counters only where hits are
    out_bytes = args.batch * wv * 8
    alg_bytes = uniq_rows * wv * 8 + out_bytes
    and_ms = stats.and_ms / max(stats.and_launches, 1)
    achieved = alg_bytes / (and_ms * 1e-3) / 1e9

    # PCIe-inclusive rate of the host-buffer boundary (never `value`): sequences in host memory -> batch_create (H2D) ->
    # run -> fetch_hits (D2H), a few repetitions outside the timed region
    pcie_rate = None
    if world == 1 and not args.force_dist:
        # a serving loop: two workspaces, reloaded per batch; while one batch runs, the host uploads the next one and
        # downloads the hit lists of the one before (what BIGSI.search_stream does)
        reps = 8
        ws = [st.new_batch(seqs, args.k) for _ in range(2)]
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(reps):
            cur = ws[i % 2]
            cur.reload(seqs)                       # H2D of the sequences (waits for this workspace's previous batch only)
            cur.run(args.threshold, sparse_counts=True)
            if i:
                ws[(i - 1) % 2].hits()             # D2H of the previous batch's hit lists while `cur` runs
        ws[(reps - 1) % 2].hits()
        pcie_rate = total_unique * reps / (time.perf_counter() - t1)
        for w_ in ws:
            w_.close()

    # HBM traffic of this kernel on this workload, when a PMC pass for it has been committed (PMC counters cannot be
    # collected from inside the timed run; see profiles/)
    traffic, traffic_src = None, None
    wkey = "rows=%d cols=%d hashes=%d batch=%d qlen=%d k=%d threshold=%s draws=%d" % (
        args.rows, args.cols, args.hashes, args.batch, args.qlen, args.k, repr(float(args.threshold)), args.and_draws)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(wkey)
        if ent:
            traffic, traffic_src = ent["traffic_bytes_per_launch"], ent["source"]
    except OSError:
        pass

    verified = None
    if not args.no_verify and rank == 0:
        # planted round trip on every shard + one sampled query against the oracle on this rank's shard
        from oracle.ref_model import SynthOracle
        for j, qi in enumerate(planted):
            hits = set(colours[int(off[qi]):int(off[qi + 1])].tolist())
            for g in range(world):
                assert g * args.cols + (1009 * (j + 1) + 13 * g) % args.cols in hits, "planted query %d missing on shard %d" % (qi, g)
        orc = SynthOracle(SEED, 0, args.rows, args.cols, args.hashes, args.k, args.and_draws)
        for j, qi in enumerate(planted):
            orc.insert_kmers((1009 * (j + 1)) % args.cols, seqs[qi])
        for qi in (planted[0], 1):
            u, cnt = orc.counts(seqs[qi])
            want = np.flatnonzero(cnt >= (u if exact else mk[qi]))
            got = colours[int(off[qi]):int(off[qi + 1])]
            got0 = got[got < args.cols]
            assert u == nu[qi] and np.array_equal(got0, want), "oracle mismatch on query %d" % qi
        verified = "planted round trip on %d shard(s) + 2 queries bit-exact vs oracle" % world

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        per_rank_rate = total_unique / (elapsed / args.steps)
        line = {
            "metric": "kmer_lookups_per_s", "value": per_rank_rate * world, "unit": "kmer_lookups/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: synthetic %d-row x %d-sample index per GPU (%.1f GB HBM), h=%d, %d x %d bp queries, "
                            "k=%d, threshold=%g (%s)" % (args.rows, args.cols, info.index_bytes / 1e9, args.hashes, args.batch, args.qlen,
                                                         args.k, args.threshold, "exact" if exact else "counts"),
                "rows": args.rows, "cols_per_gpu": args.cols, "total_cols": args.cols * world, "hashes": args.hashes,
                "batch": args.batch, "qlen": args.qlen, "unique_kmers_per_batch": total_unique, "hits_last_step": int(off[-1]),
                "kmer_lookups_per_s_full_index": per_rank_rate, "parallelism": "column-shard x%d + RCCL all-gather" % world,
                "index_fill_s": fill_s, "verified": verified,
                "pcie_inclusive_kmer_lookups_per_s": pcie_rate,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_and_exact" if exact else "k_and_count",
                         "alg_bytes_per_launch": alg_bytes, "kernel_ms": and_ms, "launches_timed": int(stats.and_launches),
                         # per step, from warmup steps 2..W: K1 (+ row sort on the exact path); K4's three launches
                         "kmerize_ms": warm.kmerize_ms / (args.warmup - 1) if args.warmup > 1 else None,
                         "compact_ms": warm.compact_ms / (args.warmup - 1) if args.warmup > 1 else None},
        }
        if args.cpu_seconds > 0 and world == 1:        # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(args, seqs, exact)
    for b_ in batches:
        b_.close()
    st.delete_all()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio (block-buffered when stdout is a pipe/file): flush it out first so
        # that the JSON line is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
