"""The process that OWNS the resident indexes of tests/test_gpu_sharing.py: builds golden G7's index and a synthetic 16+ GB one,
exports both for attach, then obeys one-word commands on stdin (sync / plant / quit), answering each with one line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bigsi_amd                      # noqa: E402
from bigsi_amd.storage import get_storage      # noqa: E402
from conftest import load_golden      # noqa: E402

out_dir, big_rows, big_cols = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = load_golden("g7_random.json")
k, m, h = g["k"], g["m"], g["h"]
cfg = {"storage-engine": "hip-hbm", "k": k, "m": m, "h": h, "storage-config": {"name": "g7", "export": os.path.join(out_dir, "g7.attach")}}
kmers = lambda s: [s[i:i + k] for i in range(len(s) - k + 1)]      # noqa: E731
b = bigsi_amd.BIGSI.build(cfg, [bigsi_amd.BIGSI.bloom(cfg, kmers(a) + kmers(c)) for a, c in g["sample_seqs"]], g["sample_names"])
b.storage.sync()                      # (writes the attach file: storage-config `export`)

# the same G7 index spread over three shards of one process (a device group): exported shard by shard
gcfg = {"storage-engine": "hip-hbm", "k": k, "m": m, "h": h,
        "storage-config": {"name": "g7group", "devices": [0, 0, 0], "max_cols": len(g["sample_names"]), "export": os.path.join(out_dir, "g7group.attach")}}
bg = bigsi_amd.BIGSI.build(gcfg, [bigsi_amd.BIGSI.bloom(cfg, kmers(a) + kmers(c)) for a, c in g["sample_seqs"]], g["sample_names"])
bg.storage.sync()

big = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": big_rows, "h": 3,
                   "storage-config": {"name": "big", "max_cols": big_cols, "export": os.path.join(out_dir, "big.attach")}})
big.delete_all()
for key, v in (("number_of_rows", big_rows), ("number_of_cols", big_cols), ("ksi:bloomfilter_size", big_rows), ("ksi:num_hashes", 3)):
    big.set_integer(key, v)
big.fill_synthetic(4242, 0, 2)
rng = np.random.default_rng(9)
seqs = ["".join(rng.choice(list("ACGT"), size=200)) for _ in range(4)]
big.insert_kmers(12345, [seqs[0]], 31)
big.sync()
print(json.dumps({"ready": True, "pid": os.getpid(), "seqs": seqs, "index_bytes": int(big.res.info().index_bytes)}), flush=True)
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "sync":
        big.sync()
        b.storage.sync()
        print(json.dumps({"ok": "sync"}), flush=True)
    elif cmd == "plant":              # the owner writes: an attached handle's next search sees it
        big.insert_kmers(777, [seqs[1]], 31)
        big.sync()
        print(json.dumps({"ok": "plant"}), flush=True)
    elif cmd == "quit":
        break
big.delete_all()
bg.delete()
b.delete()
