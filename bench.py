#!/usr/bin/env python
"""bench.py -- BIGSI query hot path on MI355X: k-mer lookups/s and achieved HBM GB/s of the row-fetch-AND kernel.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c4|c5|northstar] [--threshold T]

With N > 1 and no launcher environment the script starts its own N ranks (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set for them); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is
one of the launcher's ranks.  Either way rank 0 prints ONE JSON line.

A "step" is one pass of the whole device path over one batch of synthetic queries that is already resident in HBM:
K1 k-merise/dedupe/hash -> K2 row fetch + AND (or bit-sliced counting) -> K4 threshold + compaction, and for N > 1 the RCCL
all-gather of the per-sample result vectors, the compaction of the gathered result and (thresholded searches) the all-reduce
of the per-hit counts, issued by libbigsi_hip.so on its own communicator.

Workloads are BASELINE.json's configurations; each names a WHOLE index, which is split by column range over the N GPUs
(strong scaling: total work fixed).  `value` = unique query k-mers looked up in the WHOLE index per second of wall time,
exchange included -- every rank looks every k-mer up in its shard, so the shard-level lookups/s summed over ranks
(config.shard_lookups_per_s_sum) is N times that and is NOT the headline.
    c3 (default)  BASELINE configs[2]: 10M rows x 100k samples (125 GB), h=4, 8192 x 1 kbp queries, threshold 1.0
    c2            BASELINE configs[1]: 1M x 10k, h=3, 1000 x 61-mers, a different batch every step (32 staged batches cycle)
    c4            BASELINE configs[3]: 25M x 500k, h=3 (1.56 TB: needs 8 GPUs), 256 x 1 kbp queries per step
    c5            BASELINE configs[4]: c4 at threshold 0.4 with score=True: every step also brings the hit lists to the host and
                  runs the whole scored path for them -- K5 + K6 on the device (presence bits, run tallies, the rounded score chain),
                  closed-form score fields, presence strings and result dicts on the host -- one batch behind the launches;
                  16 queries x 16 planted samples per shard and batch
    northstar     BASELINE north_star: 10M x 500k, h=3 (625 GB: needs >= 4 GPUs)
`--shard-of P` runs, on fewer GPUs, the first N of the P column shards of the workload (e.g. `--workload c4 --shard-of 8
--gpus 1` is what one GPU of the 8-GPU C4 run does; value is then the rate against that part of the index and says so).
`--scaling weak` instead gives every GPU the workload's whole shape (index = N x the columns).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
HBM_BYTES = 288e9            # per GPU (spec); an index shard may take at most FIT_FRACTION of it
FIT_FRACTION = 0.93
SEED = 20260928

WORKLOADS = {
    "c2": dict(rows=1_000_000, cols=10_000, hashes=3, batch=1000, qlen=61, threshold=1.0, distinct=32, score=False,
               name="BASELINE configs[1]"),
    "c3": dict(rows=10_000_000, cols=100_000, hashes=4, batch=8192, qlen=1000, threshold=1.0, distinct=2, score=False,
               name="BASELINE configs[2]"),
    "c4": dict(rows=25_000_000, cols=500_000, hashes=3, batch=256, qlen=1000, threshold=1.0, distinct=2, score=False,
               name="BASELINE configs[3]"),
    "c5": dict(rows=25_000_000, cols=500_000, hashes=3, batch=256, qlen=1000, threshold=0.4, distinct=2, score=True,
               name="BASELINE configs[4]"),
    "northstar": dict(rows=10_000_000, cols=500_000, hashes=3, batch=256, qlen=1000, threshold=1.0, distinct=2, score=False,
                      name="BASELINE north_star shape"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    p.add_argument("--shard-of", type=int, default=0, help="hold the first --gpus of this many column shards of the workload")
    # overrides of the workload's shape (ad-hoc runs, tests)
    p.add_argument("--rows", type=int)
    p.add_argument("--cols", type=int, help="sample columns of the WHOLE index (strong) / per GPU (weak)")
    p.add_argument("--hashes", type=int)
    p.add_argument("--batch", type=int)
    p.add_argument("--qlen", type=int)
    p.add_argument("--threshold", type=float)
    p.add_argument("--distinct-batches", type=int, help="staged query batches the steps cycle through")
    p.add_argument("--score", type=int, choices=[0, 1])
    p.add_argument("--k", type=int, default=31)
    p.add_argument("--and-draws", type=int, default=2, help="bit density of the synthetic index = 2^-draws")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    p.add_argument("--cpu-rows", type=int, default=200_000)
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--score-queue", default="beside", choices=["ordered", "beside"],
                   help="score=True workloads: K5 + K6 of a batch queued on the index stream behind the next batch's kernels (ordered) or on "
                        "the library's high-priority score stream beside them (beside)")
    p.add_argument("--host-visible", type=int, default=1, choices=[0, 1],
                   help="0: skip the host-visible measurements after the timed region (profiled runs: rocprofv3's per-kernel averages then "
                        "cover launches of the timed shape only)")
    p.add_argument("--also", default="auto", choices=["auto", "all", "none"],
                   help="after the headline, run short legs of the other BASELINE configurations and report them under config.also "
                        "(auto: only for the default single-GPU c3 run)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for several ranks on one GPU)")
    p.add_argument("--one-device", action="store_true", help="every rank uses device 0 (dry runs of the N>1 path on a 1-GPU box; needs --backend gloo)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise the process group and take the exchange path even with one rank (exercises the N>1 code)")
    a = p.parse_args()
    w = dict(WORKLOADS[a.workload])
    a.custom = []
    for key, arg in (("rows", a.rows), ("cols", a.cols), ("hashes", a.hashes), ("batch", a.batch), ("qlen", a.qlen),
                     ("threshold", a.threshold), ("distinct", a.distinct_batches), ("score", a.score)):
        if arg is not None:
            w[key] = type(w[key])(arg)
            a.custom.append(key)
    a.w = w
    return a


def rand_seqs(rng, n, qlen):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[r].tobytes().decode("ascii") for r in rng.integers(0, 4, size=(n, qlen), dtype=np.uint8)]


# ------------------------------------------------------------------------------------------------ self launch
def self_launch(args):
    """--gpus N without a launcher: start N ranks of this script, wait, relay rank 0's JSON line."""
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out0 = tempfile.NamedTemporaryFile("w+", prefix="bigsi_bench_rank0_", suffix=".out", delete=False)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr, stderr=sys.stderr))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:                    # one rank died: the others would wait for it in a collective for ever
                rc = bad[0].returncode
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate() if rc else p.wait()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    rc = rc or max(abs(p.returncode or 0) for p in procs)
    out0.seek(0)
    text = out0.read()
    out0.close()
    os.unlink(out0.name)
    if rc:
        sys.stderr.write(text)
        raise SystemExit("bench.py: a rank failed (exit code %d)" % rc)
    sys.stdout.write(text)
    sys.stdout.flush()


def cpu_baseline(args, w, cols_cpu, exact):
    """The CPU baseline beside the number (never the target), each leg in a fresh subprocess (fork pools must not inherit a HIP
    context), on the GPU run's synthetic index at full row width with the rows reduced to fit host RAM:
      * THROUGH THE PRODUCT BOUNDARY: libbigsi_cpu.so (include/bigsi_cpu.h, the CPU twin of the C ABI) -- reference-shaped on ONE core
        (the reported `value`), the reference's process-pool-over-sequences parallelism (bulk_search) on every physical core, and
        its word-parallel mode on one core ("best CPU");
      * the oracle's C port (oracle/bigsi_oracle.c, test infrastructure), one core and pool, as a cross-check of the twin."""
    rows, batch = min(w["rows"], args.cpu_rows), min(w["batch"], 256)
    common = ["--rows", str(rows), "--cols", str(cols_cpu), "--hashes", str(w["hashes"]), "--k", str(args.k), "--and-draws", str(args.and_draws),
              "--seed", str(SEED), "--batch", str(batch), "--qlen", str(w["qlen"]), "--exact", str(int(exact))]
    sec = max(args.cpu_seconds / 4.0, 1.0)
    t = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cpu_twin_baseline.py")] + common +
                                  ["--threshold", str(w["threshold"]), "--seconds", str(sec), "--pool-runs", "2"],
                                  check=True, capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1])
    r = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py")] + common +
                                  ["--seconds", str(sec), "--pool-runs", "2", "--pool-seconds", str(max(args.cpu_seconds / 6.0, 1.0))],
                                  check=True, capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1])
    what = "%d-row x %d-sample slice of the same synthetic index (full row width of one GPU's shard, rows reduced to fit host RAM), cycling over %d of the bench's queries; rows served from RAM instead of BerkeleyDB" % (rows, cols_cpu, batch)
    return {"value": t["one_core"]["rate"], "unit": "kmer_lookups/s", "cores": 1, "kind": "port",
            "through": "libbigsi_cpu.so: bigsi_cpu_search_batch, the CPU twin of the C ABI (include/bigsi_cpu.h), reference-shaped (per-k-mer string "
                       "canonicalisation, MurmurHash3 x h, one copy per fetched row, byte-wise AND, unpack-to-int32-and-add)",
            "sample": "%d unique k-mer lookups in %.1f s on a %s" % (t["one_core"]["lookups"], t["one_core"]["seconds"], what),
            "pool": {"value": t["pool"]["rate_median"], "best": t["pool"]["rate_best"], "runs": t["pool"]["rates"], "cores": t["pool"]["threads"],
                     "host_threads": t["host_threads"],
                     "sample": "fork pool over query sequences (the reference's bulk_search parallelism), %d workers, median of %d runs of %.1f s"
                               % (t["pool"]["threads"], len(t["pool"]["rates"]), t["pool"]["seconds"])},
            "word_parallel_one_core": {"value": t["word_parallel_one_core"]["rate"], "what": "BIGSI_CPU_WORD_PARALLEL: 64-bit words of the resident rows, no copies"},
            "word_parallel_pool": {"value": t["word_parallel_pool"]["rate"], "cores": t["word_parallel_pool"]["threads"],
                                   "what": "the same in the fork pool: the best this host's CPUs do on the path"},
            "oracle_port": {"value": r["one_core"]["rate"], "pool": r["pool"]["rate_median"], "pool_cores": r["pool"]["threads"],
                            "what": "oracle/bigsi_oracle.c orc_query (test infrastructure), same slice and queries: cross-check of the twin"}}


ALSO_LEGS = [
    # (key, what, arguments): every leg is this script again in a fresh process, >= 1 s of timed steps, verified against the oracle
    ("c3_t04", "configs[2] at threshold 0.4 (bit-sliced counting kernel)", ["--workload", "c3", "--threshold", "0.4", "--steps", "16", "--warmup", "3"]),
    ("c2", "configs[1]: 1M x 10k, 1000 x 61-mers per step", ["--workload", "c2", "--steps", "50000", "--warmup", "200"]),
    ("c2_t04", "configs[1] at threshold 0.4", ["--workload", "c2", "--threshold", "0.4", "--steps", "50000", "--warmup", "200"]),
    ("c4_shard", "configs[3]: one GPU's shard (1 of 8) of 25M x 500k", ["--workload", "c4", "--shard-of", "8", "--steps", "1200", "--warmup", "10"]),
    ("c5_shard", "configs[4]: the same shard at threshold 0.4 with score=True (K5 + K6 + host assembly in the step)",
     ["--workload", "c5", "--shard-of", "8", "--steps", "1000", "--warmup", "10"]),
    ("northstar_shard", "north_star shape 10M x 500k: one GPU's shard (1 of 8)", ["--workload", "northstar", "--shard-of", "8", "--steps", "1200", "--warmup", "10"]),
]


def run_also_legs():
    """The other BASELINE configurations, one fresh process each (the headline's index has been freed), condensed."""
    out = {}
    for key, what, extra in ALSO_LEGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--cpu-seconds", "0", "--also", "none"] + extra
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf, cf = d["roofline"], d["config"]
            out[key] = {"what": what, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                        "timed_s": d["ms_per_step"] * d["steps"] / 1e3,
                        "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "alg_bytes_per_launch",
                                                           "launches_per_step", "step_GBps", "step_frac", "concurrent_launches", "traffic",
                                                           "traffic_measured_in_run", "read_launches_repeated")},
                        "verified": cf.get("verified"), "host_visible": {"stream_kmer_lookups_per_s": ((cf.get("host_visible") or {}).get("stream") or {}).get("kmer_lookups_per_s"),
                                                                   "stream_sequences": ((cf.get("host_visible") or {}).get("stream") or {}).get("sequences"),
                                                                   "one_call_us": {k_: v_ for k_, v_ in ((cf.get("host_visible") or {}).get("one_call_us") or {}).items() if k_ != "entry"},
                                                                   "two_workspace_loop_kmer_lookups_per_s": (cf.get("host_visible") or {}).get("two_workspace_loop_kmer_lookups_per_s"),
                                                                   "stream_scored_kmer_lookups_per_s": ((cf.get("host_visible") or {}).get("stream_scored") or {}).get("kmer_lookups_per_s")},
                        "scored": {k_: v_ for k_, v_ in (cf.get("presence") or {}).items() if k_ != "what"} or None,
                        "clocks_after": (cf.get("clocks") or {}).get("after_timed_region"),
                        "index_gb": cf.get("index_gb_per_gpu"), "wall_s": round(time.time() - t0, 1), "args": " ".join(extra)}
        except Exception as e:  # noqa: BLE001 -- a leg that fails is reported as such, the headline stands
            out[key] = {"what": what, "error": "%s: %s" % (type(e).__name__, str(e)[:300]), "args": " ".join(extra)}
    return out


def smi_snapshot(device):
    """Clocks / power / temperature of the device as rocm-smi reports them (explains box-to-box spread of the bandwidth figures):
    {sclk_mhz, mclk_mhz, fclk_mhz, power_w, temp_junction_c, temp_memory_c}."""
    import re
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        out = {}
        for k, v in card.items():
            kl = k.lower()
            num = re.search(r"-?\d+(\.\d+)?", str(v))
            if not num:
                continue
            val = float(num.group(0))
            for tag, name in (("sclk clock speed", "sclk_mhz"), ("mclk clock speed", "mclk_mhz"), ("fclk clock speed", "fclk_mhz"), ("power", "power_w"),
                              ("sensor junction", "temp_junction_c"), ("sensor memory", "temp_memory_c")):
                if tag in kl and name not in out:
                    out[name] = val
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}


def main():
    args = parse()
    w = args.w
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    # ---------------- this rank's part of the workload
    from bigsi_amd.parallel import ShardedSearch, plan_shards
    parts = args.shard_of or world
    if parts < world:
        raise SystemExit("--shard-of %d is fewer than --gpus %d" % (parts, world))
    if args.scaling == "weak":
        shard_cols, my_cols, total_cols = w["cols"], w["cols"], w["cols"] * world
    else:
        shard_cols, spans = plan_shards(w["cols"], parts)
        my_cols, total_cols = spans[rank][1], sum(n for _, n in spans[:world])
    words = -(-shard_cols // 64)
    stride_bytes = max(16, -(-words // 16) * 16) * 8          # the library's row pitch: 128-byte multiples
    need = w["rows"] * stride_bytes
    if need > HBM_BYTES * FIT_FRACTION:
        raise SystemExit("workload %s on %d GPU(s): a shard of %d rows x %d samples is %.0f GB, more than one MI355X holds (288 GB); "
                         "use more GPUs (or --shard-of P to run some of P shards)" % (args.workload, parts, w["rows"], shard_cols, need / 1e9))
    if my_cols <= 0:
        raise SystemExit("rank %d would hold no columns" % rank)

    import torch
    import torch.distributed as dist
    from bigsi_amd import _lib
    from bigsi_amd._lib import check
    from bigsi_amd.storage import get_storage

    if args.one_device or os.environ.get("BIGSI_BENCH_DEVICE"):          # dry runs: several ranks sharing one GPU
        local_rank = int(os.environ.get("BIGSI_BENCH_DEVICE", "0"))
    if local_rank >= torch.cuda.device_count():
        if torch.cuda.device_count() == 1 and world > 1 and args.backend == "nccl":
            local_rank = 0          # a launcher that shows every rank ONE device of its own (the PCI bus check below tells a shared GPU apart)
        else:
            raise SystemExit("rank %d needs device %d but only %d GPU(s) are visible (--one-device --backend gloo shares one)"
                             % (rank, local_rank, torch.cuda.device_count()))
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

    thr = w["threshold"]
    exact = thr == 1.0
    # ---------------- index: this rank's column shard, generated on the device
    st = get_storage({"storage-engine": "hip-hbm", "k": args.k, "m": w["rows"], "h": w["hashes"],
                      "storage-config": {"name": "bench", "device": local_rank, "max_cols": shard_cols}})
    st.delete_all()
    for key, v in (("number_of_rows", w["rows"]), ("number_of_cols", my_cols),
                   ("ksi:bloomfilter_size", w["rows"]), ("ksi:num_hashes", w["hashes"])):
        st.set_integer(key, v)
    t0 = time.time()
    st.fill_synthetic(SEED, rank, args.and_draws)
    fill_s = time.time() - t0
    info = st.res.info()

    # ---------------- queries: uniform ACGT, the same on every rank; a few of them planted into a sample of every shard
    nb = max(1, w["distinct"])
    all_seqs = [rand_seqs(np.random.default_rng(1 + i), w["batch"], w["qlen"]) for i in range(nb)]
    seqs = all_seqs[0]
    planted = list(range(0, w["batch"], 97))[:8]

    def plant_col(j, g, cols_g):
        return (1009 * (j + 1) + 13 * g) % cols_g

    # thresholded runs plant the first 70 % of a query's k-mers only, so that hits carry partial counts
    n_kmers = w["qlen"] - args.k + 1
    plant_len = w["qlen"] if exact else args.k - 1 + int(np.ceil(0.7 * n_kmers))
    for j, qi in enumerate(planted):
        st.insert_kmers(plant_col(j, rank, my_cols), [seqs[qi][:plant_len]], args.k)

    # score=True workloads need hits to score: the first 16 queries of every staged batch are planted (70 % of their k-mers)
    # into 16 samples of every shard -> 256 hits per shard and batch on top of the verification plants
    def score_plants(g, cols_g):
        return [(bi, qi, (7919 * (16 * qi + t) + 11 + 13 * g + 101 * bi) % cols_g) for bi in range(nb) for qi in range(min(16, w["batch"])) for t in range(16)]

    if w["score"]:
        for bi, qi, c in score_plants(rank, my_cols):
            st.insert_kmers(c, [all_seqs[bi][qi][:plant_len]], args.k)
    if args.one_device and world > 1 and args.backend == "nccl":
        raise SystemExit("--one-device needs --backend gloo: RCCL refuses two ranks on one device")
    if w["score"] and nb < 2:
        raise SystemExit("score=True workloads need at least two staged batches (--distinct-batches)")
    sh = ShardedSearch(st, shard_cols, device=dev, force_gather=args.force_dist, slots=max(2, nb))
    # the steps cycle through `nb` staged batches (a serving loop's workspaces): with >= 2 the exchange of one batch overlaps
    # the row-AND kernel of the next, and with many (c2) the rows of a step are not what the last steps left in the caches
    batches = [st.new_batch(s_, args.k) for s_ in all_seqs]
    count_bytes = 2 if (w["qlen"] - args.k + 1) < 65536 else 4
    sh.prepare(batches, exact, count_bytes)
    check(_lib.lib().bigsi_hip_set_profiling(st.handle, 1))

    # N > 1: before anything is timed, every rank says which physical GPU it drives; the run refuses ranks that share one (unless
    # asked to: --one-device dry runs) and a communicator whose size is not N; whether the GPUs can reach each other directly is reported
    ranks_info = None
    if use_dist:
        props = torch.cuda.get_device_properties(local_rank)
        bus = ("%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)) if hasattr(props, "pci_bus_id") else "device%d" % local_rank
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "device": local_rank, "pci_bus_id": bus, "gpu": props.name,
                                          "hbm_gb": round(props.total_memory / 1e9, 1)})
        shared = len({g["pci_bus_id"] for g in gathered}) != world
        if shared and not (args.one_device or os.environ.get("BIGSI_BENCH_DEVICE")):
            raise SystemExit("bench.py: ranks share a GPU: %r" % [(g["rank"], g["pci_bus_id"]) for g in gathered])
        peers = [g["device"] for g in gathered if g["device"] != local_rank and g["device"] < torch.cuda.device_count()]
        peer_ok = all(torch.cuda.can_device_access_peer(local_rank, d_) for d_ in peers)
        cr_ = sh.comm_ranks()
        if sh.exchange == "rccl":
            if not cr_ or cr_[1] != world:
                raise SystemExit("bench.py: the RCCL communicator reports %r ranks, %d expected" % (cr_, world))
            # (no peer access is reported, not refused: RCCL then stages through host memory -- slower, still a measurement)
        ranks_info = {"ranks": gathered, "distinct_gpus": not shared, "peer_access_from_rank0": peer_ok if peers else None}

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step_no = [0]

    def own_hits(batch):
        """this rank's share of the batch's (global) hit lists: offsets, local colours and counts, for K5 / K6 on the owning GPU"""
        off, colours, counts = sh.fetch(batch)
        owned = (colours.astype(np.int64) // shard_cols) == rank
        csum = np.concatenate([[0], np.cumsum(owned)])
        off_own = csum[off.astype(np.int64)].astype(np.uint64)
        return off_own, (colours[owned].astype(np.int64) - rank * shard_cols).astype(np.uint32), counts[owned]

    # score=True (BASELINE configs[4]: "per-kmer score accumulation (bigsi/scoring)"): every step runs the WHOLE of
    # BIGSI.search(score=True) for the hits whose columns this rank owns, three batches deep:
    #   step k:  launch batch k  |  hit lists of batch k-1 to the host, K5 + K6 for them queued (bigsi_hip_batch_score_hits_begin)  |
    #            results of batch k-2: K6's records and presence bits from pinned staging, closed-form score fields for all hits at
    #            once (scoring.score_columns), presence strings and the reference's result dicts (graph/bigsi.py:105-114 + 232-239)
    #            in its order (count descending, colour ascending).
    # The device part is K5 (presence bits of the hits) + K6 (remove_short_ones / tabulate_score / calculate_score with Python's
    # round(), percent_kmers_found).  --score-queue beside (default): on the library's high-priority score stream, beside batch k's
    # row-AND kernel -- 0.4 ms of mostly waiting for memory, but off the critical path: 1.03 ms per step measured; ordered: on the
    # index stream behind batch k -- 0.1 ms, but every step pays it: 1.14 ms.
    fetched, begun = [None], [None]
    scored = {"results": None, "hits": 0, "batches": 0, "begin_s": 0.0, "finish_s": 0.0, "end_s": 0.0}
    names = ["s%d" % (rank * shard_cols + c) for c in range(my_cols)] if w["score"] else None
    from bigsi_amd.scoring import SCORE_KEYS
    result_keys = ("percent_kmers_found", "num_kmers", "num_kmers_found", "sample_name") + SCORE_KEYS + ("kmer-presence",)

    def score_begin():
        """hit lists of the batch launched one step ago -> host; its K5 + K6 queued behind the batch just launched"""
        if fetched[0] is None:
            return
        b_, fetched[0] = fetched[0], None
        t_a = time.perf_counter()
        off_own, col_own, cnt_own = own_hits(b_)
        nk_, nu_, _ = b_.unique()
        b_.score_hits_begin(off_own, col_own, None if exact else cnt_own, nk_, ordered=args.score_queue == "ordered")
        begun[0] = (b_, off_own, col_own, cnt_own, nk_, nu_)
        scored["begin_s"] += time.perf_counter() - t_a

    def score_finish(job):
        """the scored result dicts of the batch whose K5 + K6 were queued one step ago"""
        if job is None:
            return
        from bigsi_amd.graph.bigsi import scored_rows
        b_, off_own, col_own, cnt_own, nk_, nu_ = job
        t_a = time.perf_counter()
        rec, pbits, boff = b_.score_hits_end()
        scored["end_s"] += time.perf_counter() - t_a
        o64 = off_own.astype(np.int64)
        rows = scored_rows(rec, pbits, boff, np.repeat(nk_[: w["batch"]].astype(np.int64), np.diff(o64)), total_cols)
        results = [[] for _ in range(w["batch"])]
        for i in np.flatnonzero(np.diff(o64)).tolist():
            lo, hi = int(o64[i]), int(o64[i + 1])
            c_, f_ = col_own[lo:hi], cnt_own[lo:hi]
            keep = np.flatnonzero(c_ < my_cols)
            order = keep if exact else keep[np.argsort(-f_[keep].astype(np.int64), kind="stable")]
            u = int(nu_[i])
            results[i] = [dict(zip(result_keys, (rows[lo + t][0], u, u if exact else int(f_[t]), names[int(c_[t])]) + rows[lo + t][1] + (rows[lo + t][2],)))
                          for t in order.tolist()]
        scored["results"], scored["hits"], scored["batches"] = results, scored["hits"] + len(rows), scored["batches"] + 1
        scored["finish_s"] += time.perf_counter() - t_a

    def collect():
        """drain the scoring pipeline (end of the timed region)"""
        job, begun[0] = begun[0], None
        score_finish(job)
        score_begin()
        job, begun[0] = begun[0], None
        score_finish(job)

    def step():
        """One pass of the path over the next staged batch (score=True: plus the scoring stages of the two batches before it)."""
        batch = batches[step_no[0] % len(batches)]
        step_no[0] += 1
        sh.step(batches, thr)
        if w["score"]:
            job, begun[0] = begun[0], None      # queued one step ago, behind the previous launch: done (or nearly) by now
            score_begin()                       # the batch launched one step ago: its K5 + K6 go behind the launch just made
            fetched[0] = batch
            score_finish(job)                   # (with two staged batches this is the object just run again: its results are staged on the host)

    warm = _lib.Stats()
    for i in range(args.warmup):
        step()
        if i == 0:            # the first step pays one-off costs (code object load, allocations): keep it out of the K1 / K4 figures
            sync_all()
            check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(warm), 1))
    collect()
    sync_all()
    check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(warm), 1))      # K1 / K4 durations come from the warmup steps
    # timed region: HIP events around the row-AND kernel only, and -- an event record costs the stream 5-7 us -- for steps of
    # a few tens of microseconds (short reads) only around every 8th of them, so that the region runs at its untimed speed
    short_steps = w["batch"] * max(w["qlen"] - args.k + 1, 1) < (1 << 17)
    check(_lib.lib().bigsi_hip_set_profiling(st.handle, 8 if short_steps and args.steps >= 64 else 2))
    stats = _lib.Stats()

    sync_all()
    clocks_before = smi_snapshot(local_rank) if rank == 0 else None
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    collect()                      # the last batch's share, inside the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    clocks_after = smi_snapshot(local_rank) if rank == 0 else None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(stats), 1))

    # one-launch read kernels bound their waits and mark a launch in which a workgroup gave up (repeated when its hit lists are
    # fetched): fetch every staged batch's last run, so that such a launch -- none has ever been seen -- would be counted
    repeated = 0
    if batches[0].info().one_launch:
        for b_ in batches[:step_no[0]]:        # (a short run may not have reached every staged batch)
            b_.hits()
        rs = _lib.Stats()
        check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(rs), 0))
        repeated = int(rs.read_launches_repeated)

    # ---------------- results of the first staged batch, algorithmic bytes, verification
    # (its last run: step index (k * nb) for the largest such index below warmup + steps)
    batch = batches[0]
    off, colours, counts = sh.fetch(batch)
    nk, nu, mk = batch.unique()
    total_unique = int(nu.sum())
    wv = -(-my_cols // 64)
    uniq_rows = 0
    for i in range(w["batch"]):
        uniq_rows += np.unique(batch.rows(i, nu[i])).size      # each needed row counted once (reference fetches the union once)
    # result vector the kernel stores: one bit per sample (AND bitmap / thresholded hit mask); counters only where hits are
    out_bytes = w["batch"] * wv * 8
    alg_bytes = uniq_rows * wv * 8 + out_bytes                 # SURVEY.md section 8d, this rank's shard
    # a large batch goes out as several row-AND launches of ~1000 workgroups each: per-launch figures, as rocprofv3 reports them
    launches_per_step = max(stats.and_launches_total, 1) / args.steps
    and_ms = stats.and_ms / max(stats.and_launches, 1)
    alg_bytes_launch = alg_bytes / launches_per_step
    achieved = alg_bytes_launch / (and_ms * 1e-3) / 1e9
    per_rank_gbs = [achieved]
    if use_dist:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = achieved
        dist.all_reduce(t)
        per_rank_gbs = [float(x) for x in t.tolist()]

    # score=True (configs[4]): the scoring leg of a step once more on the first staged batch, its device part timed on its own
    # (events), and the assembled results checked against the oracle's restatement of graph/bigsi.py:211-239 + scoring/score.py
    presence = None
    if w["score"]:
        in_region = dict(scored)
        off_own, col_own, cnt_own = own_hits(batch)
        check(_lib.lib().bigsi_hip_set_profiling(st.handle, 1))
        batch.score_hits(off_own, col_own, None if exact else cnt_own, nk)       # (the first synchronous call creates the score stream)
        check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(_lib.Stats()), 1))
        t1 = time.perf_counter()
        rec, pbits, boff = batch.score_hits(off_own, col_own, None if exact else cnt_own, nk)
        call_ms = (time.perf_counter() - t1) * 1e3
        ps = _lib.Stats()
        check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(ps), 1))
        fetched[0] = batch
        collect()
        presence = {"hits_per_batch": int(col_own.size), "presence_bits_bytes": int(boff[-1]), "score_record_bytes": int(rec.nbytes),
                    "kernels_ms": ps.presence_ms, "score_hits_call_ms": call_ms,
                    "alg_bytes": int(ps.presence_bytes), "GBps": ps.presence_bytes / max(ps.presence_ms, 1e-9) / 1e6,
                    "in_timed_region": {"batches_scored": in_region["batches"], "hits_scored": in_region["hits"],
                                        "begin_ms_per_batch": in_region["begin_s"] / max(in_region["batches"], 1) * 1e3,
                                        "finish_ms_per_batch": in_region["finish_s"] / max(in_region["batches"], 1) * 1e3,
                                        "of_which_waiting_for_the_device_ms": in_region["end_s"] / max(in_region["batches"], 1) * 1e3,
                                        "host_us_per_hit": (in_region["begin_s"] + in_region["finish_s"]) / max(in_region["hits"], 1) * 1e6},
                    "score_queue": args.score_queue,
                    "what": "every step also runs the whole of BIGSI.search(score=True), three batches deep: hit lists of the previous batch "
                            "to the host and K5 + K6 for them queued beside / behind the batch just launched (presence bits, remove_short_ones / "
                            "tabulate_score / calculate_score with Python's round(), percent_kmers_found); for the batch before that: "
                            "closed-form score fields, presence strings and the reference's result dicts in its order"}

    # PCIe-inclusive rate of the host-buffer boundary (never `value`): sequences in host memory -> batch_reload (H2D) ->
    # run -> fetch_hits (D2H), a few repetitions outside the timed region
    pcie_rate = None
    if world == 1 and not args.force_dist:
        # a serving loop: two workspaces, reloaded per batch; while one batch runs, the host uploads the next one and
        # downloads the hit lists of the one before (what BIGSI.search_stream does)
        reps = 8 if w["batch"] * w["qlen"] >= (1 << 20) else 200      # (small batches: enough repetitions to time)
        ws = [st.new_batch(seqs, args.k) for _ in range(2)]
        for w_ in ws:                                                  # both workspaces warm
            w_.run(thr, sparse_counts=True)
            w_.hits()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(reps):
            cur = ws[i % 2]
            cur.reload(all_seqs[i % nb])           # H2D of the sequences (waits for this workspace's previous batch only)
            cur.run(thr, sparse_counts=True)
            if i:
                ws[(i - 1) % 2].hits()             # D2H of the previous batch's hit lists while `cur` runs
        ws[(reps - 1) % 2].hits()
        pcie_rate = total_unique * reps / (time.perf_counter() - t1)
        for w_ in ws:
            w_.close()

    # host-visible figures of the C boundary (wall time of the call, host sequences in, host hit lists out; never
    # `value`, which is quoted with the batch resident in HBM):
    #   stream     ONE bigsi_hip_search_stream call over several of the step's batches (the library pipelines its own chunks)
    #   one_call   bigsi_hip_search_batch latency for the step's first batch (if it is a batch of reads) and for a single query
    host_visible = {"two_workspace_loop_kmer_lookups_per_s": pcie_rate}
    if world == 1 and not args.force_dist and args.host_visible:
        lib_ = _lib.lib()
        want = 256 if w["batch"] * w["qlen"] < (1 << 17) else (8 if w["batch"] * w["qlen"] < (1 << 20) else 2)
        if w["score"]:
            want = max(want, 24)              # (six device batches of the library's own size: its pipeline in steady state)
        many = [s_ for i in range(want) for s_ in all_seqs[i % nb]]
        blob, soff = _lib.pack_seqs(many)
        n_many = len(many)
        hnk, hnu, hoff = np.zeros(n_many, np.uint32), np.zeros(n_many, np.uint32), np.zeros(n_many + 1, np.uint64)
        hcap = max(int(off[-1]) * want * 2, 1 << 16)
        hcol, hcnt = np.zeros(hcap, np.uint32), np.zeros(hcap, np.uint32)

        def stream_call():
            t_ = time.perf_counter()
            check(lib_.bigsi_hip_search_stream(st.handle, blob, _lib.ptr(soff), n_many, args.k, float(thr), 0, _lib.ptr(hnk), _lib.ptr(hnu), None,
                                               _lib.ptr(hoff), _lib.ptr(hcol), _lib.ptr(hcnt), hcap))
            return time.perf_counter() - t_
        stream_call()
        times = sorted(stream_call() for _ in range(5))
        assert np.array_equal(hnu[: w["batch"]], nu) and int(hoff[w["batch"]]) == int(off[-1])      # the first batch again, same answers
        host_visible["stream"] = {"kmer_lookups_per_s": float(hnu.sum()) / times[len(times) // 2], "best": float(hnu.sum()) / times[0],
                                  "sequences": n_many, "sequence_bytes": len(blob), "hits": int(hoff[-1]), "call_ms": times[len(times) // 2] * 1e3,
                                  "entry": "bigsi_hip_search_stream (one call; median of 5)"}

        if w["score"]:
            # score=True through the boundary alone: ONE bigsi_hip_search_stream_scored call (sequences in; hit lists, presence
            # bits and score records out; each device batch's K5 + K6 beside the next batch's row-AND)
            from bigsi_amd.scoring import HIT_SCORE_DTYPE
            need = np.zeros(1, np.uint64)
            n_hits_many = int(hoff[-1])
            hboff, hrec = np.zeros(n_hits_many + 1, np.uint64), np.zeros(max(n_hits_many, 1), HIT_SCORE_DTYPE)
            words = (hnk.astype(np.int64) + 63) // 64
            hbits = np.zeros(max(int((words * np.diff(hoff).astype(np.int64)).sum()) * 8, 8), np.uint8)

            def scored_call():
                t_ = time.perf_counter()
                check(lib_.bigsi_hip_search_stream_scored(st.handle, blob, _lib.ptr(soff), n_many, args.k, float(thr), 0, _lib.ptr(hnk), _lib.ptr(hnu),
                                                          None, _lib.ptr(hoff), _lib.ptr(hcol), _lib.ptr(hcnt), n_hits_many, _lib.ptr(hbits), hbits.size,
                                                          _lib.ptr(hboff), _lib.ptr(hrec), _lib.ptr(need)))
                return time.perf_counter() - t_
            scored_call()
            times = sorted(scored_call() for _ in range(5))
            assert int(need[0]) == hbits.size and int(hoff[-1]) == n_hits_many and (hrec["num_kmers"][:n_hits_many] > 0).all()
            host_visible["stream_scored"] = {"kmer_lookups_per_s": float(hnu.sum()) / times[len(times) // 2], "best": float(hnu.sum()) / times[0],
                                             "hits": n_hits_many, "bit_bytes": int(need[0]), "call_ms": times[len(times) // 2] * 1e3,
                                             "entry": "bigsi_hip_search_stream_scored (one call; median of 5)"}

        def one_call(seq_list, reps):
            bl, so = _lib.pack_seqs(seq_list)
            n_ = len(seq_list)
            a_, b_, c_ = np.zeros(n_, np.uint32), np.zeros(n_, np.uint32), np.zeros(n_ + 1, np.uint64)
            ts = []
            for _ in range(reps + 3):
                t_ = time.perf_counter()
                check(lib_.bigsi_hip_search_batch(st.handle, bl, _lib.ptr(so), n_, args.k, float(thr), 0, _lib.ptr(a_), _lib.ptr(b_), None,
                                                  _lib.ptr(c_), _lib.ptr(hcol), _lib.ptr(hcnt), hcap))
                ts.append(time.perf_counter() - t_)
            return float(np.median(ts[3:]) * 1e6)
        host_visible["one_call_us"] = {"single_query": one_call(seqs[1:2], 100), "entry": "bigsi_hip_search_batch (the C call, median of 100)"}
        if w["batch"] * w["qlen"] < (1 << 17):
            host_visible["one_call_us"]["whole_batch_of_%d" % w["batch"]] = one_call(seqs, 100)

    # HBM traffic of this kernel on this workload, when a PMC pass for it has been committed (PMC counters cannot be
    # collected from inside the timed run; see profiles/)
    traffic, traffic_src = None, None
    wkey = "rows=%d cols=%d hashes=%d batch=%d qlen=%d k=%d threshold=%s draws=%d" % (
        w["rows"], my_cols, w["hashes"], w["batch"], w["qlen"], args.k, repr(float(thr)), args.and_draws)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(wkey)
        if ent:
            traffic, traffic_src = ent["traffic_bytes_per_launch"], ent["source"]
    except OSError:
        pass

    verified = None
    if not args.no_verify and rank == 0:
        # planted round trip on every shard + sampled queries against the oracle on this rank's shard
        from oracle.ref_model import SynthOracle
        spans_cols = [my_cols] * world if args.scaling == "weak" else [n for _, n in plan_shards(w["cols"], parts)[1][:world]]
        for j, qi in enumerate(planted):
            hits = set(colours[int(off[qi]):int(off[qi + 1])].tolist())
            for g in range(world):
                assert g * shard_cols + plant_col(j, g, spans_cols[g]) in hits, "planted query %d missing on shard %d" % (qi, g)
        orc = SynthOracle(SEED, 0, w["rows"], my_cols, w["hashes"], args.k, args.and_draws)
        for j, qi in enumerate(planted):
            orc.insert_kmers(plant_col(j, 0, my_cols), seqs[qi][:plant_len])
        if w["score"]:
            for bi, qi, c in score_plants(0, my_cols):
                orc.insert_kmers(c, all_seqs[bi][qi][:plant_len])
        sample = sorted(set([planted[0], 1 % w["batch"], w["batch"] // 2, w["batch"] - 1]))
        for qi in sample:
            u, cnt = orc.counts(seqs[qi])
            want = np.flatnonzero(cnt >= (u if exact else mk[qi]))
            lo, hi = int(off[qi]), int(off[qi + 1])
            sel = colours[lo:hi] < shard_cols                   # rank 0's shard
            assert u == nu[qi] and np.array_equal(colours[lo:hi][sel], want), "oracle mismatch on query %d" % qi
            assert np.array_equal(counts[lo:hi][sel], cnt[want].astype(np.uint32)), "oracle count mismatch on query %d" % qi
        verified = "planted round trip on %d shard(s) + %d queries bit-exact (colours and counts) vs oracle" % (world, len(sample))
        if w["score"]:
            # the result dicts of two planted queries of the first staged batch, whole lists, against the oracle's scorer
            from oracle import coracle
            from oracle.ref_model import Scorer as OracleScorer
            osc, n_checked = OracleScorer(total_cols), 0
            for qi in (0, min(15, w["batch"] - 1)):
                kmers, uniq, rows_q = orc.per_kmer_rows(seqs[qi])
                cnt = coracle.unpack_and_sum(rows_q)[:my_cols]
                u = len(uniq)
                want_cols = [int(c) for c in np.flatnonzero(cnt >= (u if exact else int(np.ceil(u * thr))))]
                if not exact:
                    want_cols.sort(key=lambda c: -int(cnt[c]))
                qbits = np.unpackbits(rows_q, axis=1)
                idx = {km: t for t, km in enumerate(uniq)}
                got = scored["results"][qi]
                assert [r["sample_name"] for r in got] == ["s%d" % c for c in want_cols], "scored hit list of query %d differs from the oracle" % qi
                for r, c in zip(got, want_cols):
                    col = "".join("1" if qbits[idx[km], c] else "0" for km in kmers)
                    want = {"percent_kmers_found": round(100 * float(cnt[c]) / u, 2), "num_kmers": u, "num_kmers_found": int(cnt[c]), "sample_name": "s%d" % c}
                    want.update(osc.score(col))
                    want["kmer-presence"] = col
                    assert list(r) == list(want)
                    for key, v in want.items():
                        ok = abs(r[key] - v) <= 1e-12 * abs(v) + 2.5e-16 if key in ("evalue", "pvalue") else r[key] == v
                        assert ok, "score field %s of query %d sample %d: %r vs oracle %r" % (key, qi, c, r[key], v)
                    n_checked += 1
            assert n_checked >= 16
            verified += "; %d scored result dicts (all 22 keys, order included) equal to the oracle's restatement of BIGSI.score" % n_checked

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rate = total_unique / (elapsed / args.steps)
        cr = sh.comm_ranks()
        whole = args.scaling == "strong" and parts == world
        line = {
            "metric": "kmer_lookups_per_s", "value": rate, "unit": "kmer_lookups/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "%s%s: synthetic %d-row x %d-sample index%s, h=%d, %d x %d bp queries per step (%d staged batches), k=%d, "
                            "threshold=%g (%s)%s"
                            % (w["name"], " (shape overridden: %s)" % ",".join(args.custom) if args.custom else "", w["rows"],
                               w["cols"] if args.scaling == "strong" else total_cols,
                               "" if whole and world == 1 else
                               (" column-sharded over %d GPUs (%d samples each)" % (world, shard_cols) if whole else
                                " of which this run holds %d of %d column shards (%d samples)" % (world, parts, total_cols)
                                if args.scaling == "strong" else " = %d samples per GPU, weak scaling" % shard_cols),
                               w["hashes"], w["batch"], w["qlen"], nb, args.k, thr, "exact" if exact else "counts",
                               ", score=True: K5 + K6 + host assembly of every hit's scored result dict inside the step" if w["score"] else ""),
                "workload_key": args.workload, "rows": w["rows"], "cols_per_gpu": shard_cols, "total_cols": total_cols,
                "index_gb_per_gpu": info.index_bytes / 1e9, "hashes": w["hashes"], "batch": w["batch"], "qlen": w["qlen"],
                "unique_kmers_per_batch": total_unique, "hits_first_batch": int(off[-1]),
                "value_is": "unique query k-mers per second against the %d samples held by this run, exchange included" % total_cols,
                "shard_lookups_per_s_sum": rate * world,
                "aggregate_GBps": sum(per_rank_gbs), "per_rank_GBps": per_rank_gbs,
                "parallelism": "column-shard x%d%s" % (world, "" if not use_dist else
                                                      " + ncclAllGather of 1 bit/sample (library-owned RCCL communicator)"
                                                      if sh.exchange == "rccl" else " + torch.distributed(%s) all-gather" % args.backend),
                "backend": args.backend if use_dist else None, "exchange": sh.exchange if use_dist else None,
                "rccl_ranks": cr[1] if cr else None, "ranks": ranks_info,
                "index_fill_s": fill_s, "index_contiguous": bool(stats.index_contiguous), "verified": verified, "presence": presence,
                "pcie_inclusive_kmer_lookups_per_s": pcie_rate,
                "host_visible": host_visible,
                "clocks": {"before_timed_region": clocks_before, "after_timed_region": clocks_after},
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_run": False,
                         "kernel": "k_reads_fused (K1 + row-AND + K4 in one launch)" if batch.info().one_launch else "k_and_exact" if exact else "k_and_count",
                         "alg_bytes_per_launch": alg_bytes_launch, "kernel_ms": and_ms, "launches_timed": int(stats.and_launches),
                         "launches_per_step": launches_per_step, "alg_bytes_per_step": alg_bytes,
                         # bytes of a step over the step's wall time: what the HBM delivers to the whole pipeline.  For batches of
                         # reads the one-launch kernels of consecutive steps overlap on the library's three read streams (one
                         # step's k-merising and compaction under its neighbours' row fetches), so each kernel's own duration
                         # -- `kernel_ms`, what `achieved` is priced on -- spans its neighbours too and exceeds the step time.
                         "step_GBps": alg_bytes / (ms_per_step * 1e-3) / 1e9, "step_frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "concurrent_launches": 3 if batch.info().one_launch else 1, "read_launches_repeated": repeated,
                         "rank": 0,
                         # per step, from warmup steps 2..W: K1 (+ row sort on the exact path); K4
                         "kmerize_ms": warm.kmerize_ms / (args.warmup - 1) if args.warmup > 1 else None,
                         "compact_ms": warm.compact_ms / (args.warmup - 1) if args.warmup > 1 else None},
        }
        if args.cpu_seconds > 0 and world == 1:        # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(args, w, my_cols, exact)
    for b_ in batches:
        b_.close()
    sh.close()
    st.delete_all()
    if rank == 0 and world == 1 and not use_dist and (args.also == "all" or (args.also == "auto" and args.workload == "c3" and not args.custom
                                                                             and not args.shard_of and args.scaling == "strong")):
        # the other BASELINE configurations, each a short run of this script in a fresh process now that the index is freed
        line["config"]["also"] = run_also_legs()
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
