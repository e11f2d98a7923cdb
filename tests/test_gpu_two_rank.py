"""Two real ranks of bench.py on ONE GPU (gloo backend carrying CUDA tensors; RCCL refuses two ranks on one device): the
rank-dependent halves of the column-shard path -- per-shard synthetic contents and planting, in-place all-gather slots,
gathered compaction with colour = shard * shard_cols + local, per-hit count all-reduce, two alternating batches on the
compute / comm streams -- run with two actual processes and the bench's own verification against the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, launcher_ranks=0, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher_ranks:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr", "127.0.0.1",
                "--master-port", "29541"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert r.stdout.strip().splitlines()[-1] == line                # the JSON is the last line on stdout
    return json.loads(line)


@pytest.mark.gpu
@pytest.mark.parametrize("threshold", ["1.0", "0.4"])
def test_bench_self_launches_two_ranks(threshold):
    """`python bench.py --gpus 2` as typed, no launcher: the script starts its own two ranks (here both on the one GPU of the
    test box, gloo carrying the exchange) and rank 0 prints the line; strong scaling -- the index is split, `value` is the rate
    against the whole of it."""
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--rows", "400000", "--cols", "40000", "--batch", "256",
                "--backend", "gloo", "--one-device", "--threshold", threshold, "--cpu-seconds", "0"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_cols"] == 40000 and d["config"]["cols_per_gpu"] == 20000
    assert d["config"]["hits_first_batch"] == 6                    # 3 planted queries x 2 shards
    assert "2 shard(s)" in d["config"]["verified"]
    assert len(d["config"]["per_rank_GBps"]) == 2 and d["config"]["exchange"] == "torch"
    assert d["value"] == pytest.approx(d["config"]["unique_kmers_per_batch"] / d["ms_per_step"] * 1e3, rel=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("cols", ["129", "257", "1000"])
def test_bench_uneven_shards(cols):
    """Shard widths that straddle a 64-column word (65 + 64), a 128-column pair of words (129 + 128), and three ranks
    (334 + 334 + 332): every rank must size its result vectors from the group's shard width, not from its own column count."""
    world = 3 if cols == "1000" else 2
    d = _bench(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--rows", "50021", "--cols", cols, "--batch", "200",
                "--qlen", "200", "--backend", "gloo", "--one-device", "--threshold", "0.4", "--cpu-seconds", "0"])
    assert d["n_gpus"] == world and d["config"]["total_cols"] == int(cols)
    assert d["config"]["hits_first_batch"] == 3 * world and "%d shard(s)" % world in d["config"]["verified"]


@pytest.mark.gpu
def test_bench_eight_ranks_under_the_launcher():
    """world_size 8 (the node the scaling bench runs on) squeezed onto the one GPU of the test box, started the way the
    driver starts it (torch.distributed.run): eight gather slots, colour offsets of eight shards, the count all-reduce over
    eight ranks, planted hits found on every shard."""
    d = _bench(["--gpus", "8", "--steps", "4", "--warmup", "1", "--rows", "200000", "--cols", "80000", "--batch", "256",
                "--backend", "gloo", "--one-device", "--threshold", "0.4", "--cpu-seconds", "0"], launcher_ranks=8)
    assert d["n_gpus"] == 8 and d["config"]["total_cols"] == 80000
    assert d["config"]["hits_first_batch"] == 24                   # 3 planted queries x 8 shards
    assert "8 shard(s)" in d["config"]["verified"]


@pytest.mark.gpu
def test_bench_gpus_8_runs_the_whole_index_legs(tmp_path):
    """The driver's 8-GPU command, as it launches it, dry-run on the one GPU of the test box (eight gloo ranks sharing it, rows cut to
    200 k so that eight shards of the 500 k-sample indexes fit): after the headline every rank starts its rank of the WHOLE-index
    legs -- configs[3] (25M x 500k exact), configs[4] (0.4 with score=True in the step), the north-star 10M x 500k shape, c3 at 0.4 --
    each verified against the oracle on every shard, and the one line rank 0 prints carries them all inside the driver's 8 KB."""
    details = tmp_path / "full.json"
    d = _bench(["--gpus", "8", "--steps", "4", "--warmup", "2", "--backend", "gloo", "--one-device", "--rows-cap", "200000", "--leg-seconds", "0.02",
                "--cpu-seconds", "0", "--details", str(details)], launcher_ranks=8, timeout=1500)
    assert d["n_gpus"] == 8 and d["config"]["workload_key"] == "c3" and "rows<=200000" in d["config"]["workload"]
    assert "8 shard(s)" in d["config"]["verified"] and "EVERY shard" in d["config"]["verified"]
    also = d["config"]["also"]
    assert list(also) == ["c4", "c5", "northstar", "c3_t04"]
    for key, leg in also.items():
        assert "error" not in leg, (key, leg)
        assert leg["ok"] == 1 and leg["v"] > 0 and leg["k"] in ("k_and_exact", "k_and_count"), (key, leg)
        assert len(leg["gbs"]) == 8
    assert also["c5"]["hits"] > 0 and also["c5"]["k"] == "k_and_count" and also["c4"]["k"] == "k_and_exact"
    assert len(json.dumps(d)) <= 7500
    full = json.load(open(details))
    assert full["config"]["also"] == d["config"]["also"] or set(full["config"]["also"]) == set(also)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_gpus_2_and_4_run_their_legs(world):
    """The driver's 2- and 4-GPU commands as it launches them, dry-run on the one GPU of the test box (gloo ranks sharing it, rows
    capped): the headline over `world` shards and, after it, the legs of that world size -- c3 at 0.4 (2 GPUs); the north-star
    10M x 500k shape as a WHOLE index and c3 at 0.4 (4 GPUs) -- every one verified against the oracle on every shard."""
    d = _bench(["--gpus", str(world), "--steps", "4", "--warmup", "2", "--backend", "gloo", "--one-device", "--rows-cap", "200000", "--leg-seconds", "0.02",
                "--cpu-seconds", "0"], launcher_ranks=world, timeout=1500)
    assert d["n_gpus"] == world and d["config"]["workload_key"] == "c3" and "%d shard(s)" % world in d["config"]["verified"]
    also = d["config"]["also"]
    assert list(also) == (["c3_t04"] if world == 2 else ["northstar", "c3_t04"])
    for key, leg in also.items():
        assert "error" not in leg, (key, leg)
        assert leg["ok"] == 1 and leg["v"] > 0 and len(leg["gbs"]) == world, (key, leg)
    assert len(json.dumps(d)) <= 7500


@pytest.mark.gpu
def test_a_failing_rank_still_leaves_a_diagnosable_line():
    """A multi-rank run in which one rank fails (here: rank 1 is told to use a device that does not exist) ends with a non-zero exit
    code AND a JSON line from rank 0 that names the error of the rank that failed and carries the RCCL log tail and the
    environment -- what the first run on real multi-GPU hardware needs if it goes wrong."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BIGSI_BENCH_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "100000", "--cols", "4000", "--batch", "64",
           "--backend", "gloo", "--one-device", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["value"] is None and d["rc"] == 1 and d["n_gpus"] == 2 and d["metric"] == "kmer_lookups_per_s"
    blame = [d] + d["other_ranks"]
    assert any("BIGSI_BENCH_FAIL_RANK" in (b.get("error") or "") for b in blame), d
    assert "HSA_ENABLE_IPC_MODE_LEGACY" in d["config"]["env"] and isinstance(d["nccl_debug_tail"], list)


@pytest.mark.gpu
def test_bench_default_line_fits_the_drivers_tail():
    """`python bench.py` cut down to seconds (rows capped, short legs): the single-GPU line with its six legs, calibration, host-visible
    figures and CPU baseline stays under 7.5 KB and keeps the keys the judge's checks read."""
    d = _bench(["--steps", "4", "--warmup", "2", "--rows-cap", "400000", "--leg-seconds", "0.05", "--cpu-seconds", "2"], timeout=1500)
    assert len(json.dumps(d)) <= 7500
    assert list(d["config"]["also"]) == ["c3_t04", "c2", "c2_t04", "c4_shard", "c5_shard", "ns_shard", "c5_dense", "c2_dense", "c5_ee", "c3_ee", "ingest"]
    dense = d["config"]["also"]["c5_dense"]
    assert dense["hits"] > 1000 and dense["hps"] > 0 and dense["hvsh"] > 0 and dense["dps"] > 0 and dense["k4"] > 0 and dense["k56"] > 0, dense
    assert d["config"]["also"]["c5_ee"]["ee"] == 1 and d["config"]["also"]["c2_dense"]["hps"] > 0 and d["config"]["also"]["ingest"]["grp_load_GBps"] > 1
    for key, leg in d["config"]["also"].items():
        assert "error" not in leg and leg["ok"] == 1, (key, leg)
        if key == "ingest":
            assert leg["load_GBps"] > 1 and leg["save_GBps"] > 0.5 and leg["gb"] > 0.5, leg
            continue
        assert leg["box"] > 0.5 and leg["us1"] > 0 and (leg["rv"] > 0 if leg["in"] == "h" else leg["hv"] > 0), (key, leg)
        assert leg["in"] == ("h" if key in ("c3_t04", "c4_shard", "ns_shard", "c5_ee", "c3_ee") else "r"), (key, leg)      # gene-length unscored legs: host-visible steps
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert r["box_sorted_GBps"] > 1000 and r["box_random_GBps"] > 1000 and 0.5 < r["frac_of_box"] < 2
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    # the headline's steps take host sequences in and leave host hit lists out; the resident figure stands beside `value`
    assert d["config"]["value_inputs"] == "host" and d["config"]["resident_lookups_per_s"] > 0 and d["config"]["one_call_us"] > 0
    c2 = d["config"]["also"]["c2"]
    assert c2["k"].startswith("k_reads_fused") and c2["f3"] < c2["f"]          # the kernel alone on the device vs three launches overlapping


@pytest.mark.gpu
def test_bench_one_rank_rccl_exchange():
    """--force-dist: the RCCL process group and the library's own communicator with the one rank a one-GPU box allows."""
    d = _bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--rows", "400000", "--cols", "20000", "--batch", "256",
                "--force-dist", "--threshold", "0.4", "--cpu-seconds", "0"])
    assert d["config"]["exchange"] == "rccl" and d["config"]["rccl_ranks"] == 1 and "1 shard(s)" in d["config"]["verified"]
    assert d["config"]["exchange_ms"] > 0          # all-gather + gathered compaction + all-reduce, events on the communicator's stream


@pytest.mark.gpu
def test_bench_named_workloads_refuse_what_does_not_fit():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4", "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "more than one MI355X holds" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("split", [103, 136, 65, 129])
def test_sharded_bigsi_equals_whole_index(tmp_path, split):
    """ShardedBIGSI over two uneven shards (103 + 97, 136 + 64, 65 + 135, 129 + 71 samples: word counts 2+2, 3+1, 2+3, 3+2; two processes) must return, query for query, what the
    reference returned on the WHOLE 200-sample index (G7 goldens): names, counts, order, percentages and -- for score=True --
    presence strings extracted on the rank that owns each hit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import assert_results_equal, load_golden, unjson
    out = tmp_path / "sharded.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "tests", "helpers", "sharded_worker.py"), str(out), str(split)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.load(open(out))
    g = load_golden("g7_random.json")
    assert len(got["searches"]) == len(g["searches"])
    for s, res in zip(g["searches"], got["searches"]):
        if "raises" in s["out"]:
            assert res.get("raises") == s["out"]["raises"]
        else:
            assert "results" in res, res
            assert_results_equal(res["results"], unjson(s["out"]["results"]), "q%d t=%r score=%r" % (s["q"], s["threshold"], s["score"]))
    want = {(s["q"], s["threshold"]): s["out"]["results"] for s in g["searches"] if not s["score"] and "results" in s["out"]}
    for qi in range(10):
        assert_results_equal(got["batch"][qi], unjson(want[(qi, 0.4)]), "batch q%d" % qi)
    # non-ASCII queries (an accented character spliced in): against the oracle on the whole index (pinned on such text by G13)
    from oracle.ref_model import OracleBIGSI, seq_to_kmers
    k, m, h = g["k"], g["m"], g["h"]
    o = OracleBIGSI.build([OracleBIGSI.bloom(seq_to_kmers(a, k) + seq_to_kmers(c, k), m, h) for a, c in g["sample_seqs"]], g["sample_names"], k, m, h)
    wide_q = [q[:45] + "\u00e9" + q[45:] for q in g["queries"][:3]]
    for q, res in zip(wide_q, got["wide"]):
        assert_results_equal(res, o.search(q, 0.3, True), "non-ASCII")
    for q, res in zip([g["queries"][0], wide_q[0], g["queries"][1], wide_q[1], g["queries"][2]], got["mixed"]):
        assert_results_equal(res, o.search(q, 0.3), "stream with non-ASCII")
