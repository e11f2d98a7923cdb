"""One rank of tests/test_gpu_two_rank.py::test_sharded_bigsi_equals_whole_index (launched by torch.distributed.run).
Builds this rank's half of the G7 samples as its own hip-hbm index, then answers the G7 queries through ShardedBIGSI;
rank 0 writes the results as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bigsi_amd import BIGSI  # noqa: E402
from bigsi_amd.parallel import ShardedBIGSI  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                                   # both ranks share the one GPU of the test box
dist.init_process_group("gloo", rank=rank, world_size=world)
g = json.load(open(os.path.join(ROOT, "tests", "golden", "g7_random.json")))
n = len(g["sample_names"])
split = int(sys.argv[2]) if len(sys.argv) > 2 else 103                # uneven shards: `split` samples on rank 0, the rest on rank 1
lo, hi = (0, split) if rank == 0 else (split, n)
names = g["sample_names"][lo:hi]
cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "shard%d" % rank, "device": 0}, "k": g["k"], "m": g["m"], "h": g["h"]}
local = BIGSI.build_from_sequences(cfg, {nm: list(g["sample_seqs"][lo + i]) for i, nm in enumerate(names)})
sb = ShardedBIGSI(local, device=torch.device("cuda", 0))
assert sb.num_samples == n and sb.shard_sizes == [split, n - split]
out = []
for s in g["searches"]:
    q = g["queries"][s["q"]]
    try:
        res = {"results": sb.search(q, s["threshold"], s["score"])}
    except BaseException as e:  # noqa: BLE001
        res = {"raises": type(e).__name__, "message": str(e)[:300]}
    out.append(res)
batch = sb.search_batch(g["queries"][:10], 0.4)
# non-ASCII text through the exchange: an accented character spliced into three queries (explicit-k-mer batches)
wide_q = [q[:45] + "\u00e9" + q[45:] for q in g["queries"][:3]]
wide = [sb.search(q, 0.3, True) for q in wide_q]
mixed = [r for _, r in sb.search_stream([g["queries"][0], wide_q[0], g["queries"][1], wide_q[1], g["queries"][2]], 0.3, batch_size=2)]
if rank == 0:
    json.dump({"searches": out, "batch": batch, "wide": wide, "mixed": mixed}, open(sys.argv[1], "w"))
sb.close()
local.delete()
dist.barrier()
dist.destroy_process_group()
