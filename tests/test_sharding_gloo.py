"""The N>1 path on CPU: two processes, gloo backend.  Each rank owns one column shard of a small synthetic index, puts
its per-sample result vectors (computed by the oracle standing in for the device kernels) into its slot of the gather
buffer, and the in-place all-gather + colour globalisation used by bigsi_amd.parallel must reproduce the oracle's answer
on the whole (concatenated) index.  The RCCL/xGMI run itself is the driver's 8-GPU bench; this covers layout, ordering
and colour arithmetic of the exchange step."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

M, SHARD_COLS, H, K, SEED, WORLD = 4001, 150, 3, 31, 99, 2     # 150 columns: shards are not word-aligned


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _queries():
    rng = np.random.default_rng(5)
    return ["".join(rng.choice(list("ACGT"), size=int(n))) for n in (40, 75, 120, 31)]


def _shard_oracle(shard, seqs):
    from oracle.ref_model import SynthOracle
    orc = SynthOracle(SEED, shard, M, SHARD_COLS, H, K, 1)
    orc.insert_kmers(7 + shard, seqs[0])            # one planted sample per shard
    return orc


def _worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import torch
    import torch.distributed as dist
    from bigsi_amd.parallel import ShardGroup, plan_shards
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        sg = ShardGroup()
        assert (sg.rank, sg.world) == (rank, WORLD)
        shard_cols, spans = plan_shards(SHARD_COLS * WORLD, WORLD)
        assert shard_cols == SHARD_COLS and spans[rank] == (rank * SHARD_COLS, SHARD_COLS)
        seqs = _queries()
        orc = _shard_oracle(rank, seqs)
        wv_pad = (-(-SHARD_COLS // 64) + 1) // 2 * 2
        # exact: bitmap rows padded to wv_pad words; counting: uint16 counters, wv_pad*64 per sequence
        bm_stride, ct_stride = wv_pad * 8, wv_pad * 64 * 2
        bm = sg.gather_buffer(len(seqs) * bm_stride, "cpu")
        ct = sg.gather_buffer(len(seqs) * ct_stride, "cpu")
        for i, s in enumerate(seqs):
            u, bitmap = orc.exact_bitmap(s)
            row = np.zeros(bm_stride, np.uint8)
            row[: bitmap.size] = bitmap
            bm[rank, i * bm_stride:(i + 1) * bm_stride] = torch.from_numpy(row)
            _, cnt = orc.counts(s)
            c16 = np.zeros(wv_pad * 64, np.uint16)
            c16[: cnt.size] = cnt
            ct[rank, i * ct_stride:(i + 1) * ct_stride] = torch.from_numpy(c16.view(np.uint8))
        sg.all_gather_in_place(bm)
        sg.all_gather_in_place(ct)
        np.save(os.path.join(out_dir, "bm%d.npy" % rank), bm.numpy())
        np.save(os.path.join(out_dir, "ct%d.npy" % rank), ct.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_globalise(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    seqs = _queries()
    bms = [np.load(tmp_path / ("bm%d.npy" % r)) for r in range(WORLD)]
    cts = [np.load(tmp_path / ("ct%d.npy" % r)) for r in range(WORLD)]
    assert np.array_equal(bms[0], bms[1]) and np.array_equal(cts[0], cts[1])     # every rank holds the whole result
    wv_pad = (-(-SHARD_COLS // 64) + 1) // 2 * 2
    orcs = [_shard_oracle(g, seqs) for g in range(WORLD)]
    for i, s in enumerate(seqs):
        u = len(set(s[j:j + K] for j in range(len(s) - K + 1)))
        # layout [shard][seq][stride]; colour = shard * SHARD_COLS + local column
        exact_hits, count_vec = [], []
        for g in range(WORLD):
            row = bms[0][g, i * wv_pad * 8:(i + 1) * wv_pad * 8]
            bits = np.unpackbits(row)[:SHARD_COLS]
            exact_hits += [g * SHARD_COLS + int(c) for c in np.flatnonzero(bits)]
            c16 = cts[0][g, i * wv_pad * 128:(i + 1) * wv_pad * 128].view(np.uint16)[:SHARD_COLS]
            count_vec.append(c16.astype(np.int64))
        count_vec = np.concatenate(count_vec)
        # the oracle on the concatenated index: per-shard counts side by side
        want = np.concatenate([orcs[g].counts(s)[1] for g in range(WORLD)])
        assert np.array_equal(count_vec, want)
        assert exact_hits == [int(c) for c in np.flatnonzero(want == u)]
        if i == 0:
            assert {7, SHARD_COLS + 8} <= set(exact_hits)
