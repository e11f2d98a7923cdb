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
    assert d["config"]["shard_lookups_per_s_sum"] == pytest.approx(2 * d["value"])
    assert len(d["config"]["per_rank_GBps"]) == 2 and d["config"]["exchange"] == "torch"
    assert d["value"] == pytest.approx(d["config"]["unique_kmers_per_batch"] / d["ms_per_step"] * 1e3)


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
def test_bench_one_rank_rccl_exchange():
    """--force-dist: the RCCL process group and the library's own communicator with the one rank a one-GPU box allows."""
    d = _bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--rows", "400000", "--cols", "20000", "--batch", "256",
                "--force-dist", "--threshold", "0.4", "--cpu-seconds", "0"])
    assert d["config"]["exchange"] == "rccl" and d["config"]["rccl_ranks"] == 1 and "1 shard(s)" in d["config"]["verified"]


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
