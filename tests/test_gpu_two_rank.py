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


@pytest.mark.gpu
@pytest.mark.parametrize("threshold", ["1.0", "0.4"])
def test_bench_two_ranks_on_one_gpu(threshold):
    env = dict(os.environ, BIGSI_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--rows", "400000",
           "--cols", "20000", "--backend", "gloo", "--threshold", threshold, "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["total_cols"] == 40000
    assert d["config"]["hits_last_step"] == 6                      # 3 planted queries x 2 shards
    assert "2 shard(s)" in d["config"]["verified"]
    assert d["value"] == pytest.approx(2 * d["config"]["kmer_lookups_per_s_full_index"])
    assert r.stdout.strip().splitlines()[-1] == line                # the JSON is the last line on stdout


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """world_size 8 (the node the scaling bench runs on) squeezed onto the one GPU of the test box: eight gather slots, colour
    offsets of eight shards, the count all-reduce over eight ranks, planted hits found on every shard."""
    env = dict(os.environ, BIGSI_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--rows", "200000",
           "--cols", "10000", "--backend", "gloo", "--threshold", "0.4", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["config"]["total_cols"] == 80000
    assert d["config"]["hits_last_step"] == 24                     # 3 planted queries x 8 shards
    assert "8 shard(s)" in d["config"]["verified"]


@pytest.mark.gpu
def test_sharded_bigsi_equals_whole_index(tmp_path):
    """ShardedBIGSI over two uneven shards (103 + 97 samples, two processes) must return, query for query, what the
    reference returned on the WHOLE 200-sample index (G7 goldens): names, counts, order, percentages and -- for score=True --
    presence strings extracted on the rank that owns each hit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import assert_results_equal, load_golden, unjson
    out = tmp_path / "sharded.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "tests", "helpers", "sharded_worker.py"), str(out)]
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
