"""Column-range sharding of one BIGSI index over the GPUs of a node.

Every stage of the query path is independent per sample column -- hashing depends only on the query, AND and
counting are per column -- so rank g (one process per GPU) holds ALL m rows of columns
[g * shard_cols, (g+1) * shard_cols) and runs K1-K3 on its slice for the same query batch.  The only exchange is
one all-gather per batch of ONE BIT PER SAMPLE -- the AND bitmap of an exact search, or the `count >= min_kmers` hit mask
the counting kernel leaves for a thresholded one -- RCCL over xGMI through torch.distributed (backend "nccl" is RCCL on
ROCm).  The local vector is written by the kernels straight into this rank's slot of the gather buffer
(bigsi_hip_batch_set_outputs), so the collective runs in place; compaction to (colour, count) lists then runs on every
rank over the gathered [shard][seq][stride] buffer with colour = shard * shard_cols + local column.  For a thresholded
search each rank fills the counts of the hits that lie in ITS shard from its own counters (zero elsewhere) and one small
fixed-size all-reduce (sum) of the per-hit count array completes the lists: 3.2 MB + 256 KB per rank and batch at C3
instead of the 51 MB of uint16 counters a dense exchange would move.

Row-range sharding is deliberately not offered: a query touches random rows, so every k-mer would need an
AND-reduce across GPUs.
"""
import os

import numpy as np

from . import _lib
from ._lib import check


def plan_shards(total_cols, world_size):
    """(shard_cols, [(first_col, n_cols) per rank]): equal-width shards, the last one possibly short or empty."""
    shard_cols = -(-int(total_cols) // int(world_size))
    spans = []
    for g in range(world_size):
        lo = min(g * shard_cols, total_cols)
        spans.append((lo, min(shard_cols, total_cols - lo)))
    return shard_cols, spans


class ShardGroup(object):
    """torch.distributed plumbing: rank / world, the gather buffer and the in-place all-gather."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def gather_buffer(self, bytes_per_rank, device):
        import torch
        return torch.zeros((self.world, int(bytes_per_rank)), dtype=torch.uint8, device=device)

    def all_gather_in_place(self, buf):
        """buf: [world, bytes]; slot [rank] already holds this rank's result."""
        if not self.dist.is_initialized():
            return buf
        self.dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].clone() if buf.device.type == "cpu" else buf[self.rank],
                                         group=self.group)
        return buf


class ShardedSearch(object):
    """Query batches against this rank's column shard + the collective that makes every rank see the whole result.

    Two torch streams.  `stream` (compute) carries the library's K1-K3 kernels (bigsi_hip_set_stream); `comm` carries the
    RCCL all-gather and the compaction of the gathered buffer (bigsi_hip_batch_set_gather_stream).  With two batches /
    gather buffers used alternately (`slots=2`) the exchange of batch i overlaps the row-AND kernels of batch i+1:
        compute:  run(i) ---------------- run(i+1) ------------- run(i+2) ...
        comm:            gather(i) K4g(i)          gather(i+1) K4g(i+1)
    Events order the two: comm waits for run(i); run(i+2) waits until gather/compaction of batch i released its buffer.
    (torch's default stream has handle 0, which the C ABI reserves for "use the library's private stream", hence the
    explicit side streams.)"""

    def __init__(self, storage, shard_cols, group=None, device=None, force_gather=False, slots=2):
        import torch
        self.torch = torch
        self.force_gather = force_gather      # take the all-gather path even with one rank (testing)
        self.storage = storage
        self.shard_cols = int(shard_cols)
        self.sg = ShardGroup(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(self.device)
        self.comm = torch.cuda.Stream(self.device)
        check(_lib.lib().bigsi_hip_set_stream(storage.handle, self.stream.cuda_stream))
        self.slots = int(slots)
        self._bufs = [None] * self.slots
        self._hitbufs = [None] * self.slots       # (colours, counts) int32 tensors of the gathered hit lists, per slot
        self._slot_of = {}
        self._exact = True
        self._ev_run = [torch.cuda.Event() for _ in range(self.slots)]
        self._ev_free = [None] * self.slots       # recorded on comm when a slot's gather + compaction are done
        self._i = 0

    @property
    def gathering(self):
        return self.sg.world > 1 or self.force_gather

    def close(self):
        check(_lib.lib().bigsi_hip_set_stream(self.storage.handle, None))

    def prepare(self, batches, exact, count_bytes=2):
        """Give each batch (one per slot) its own [world, n_seqs*stride] gather buffer and point the batch's result at
        this rank's slot of it."""
        if not isinstance(batches, (list, tuple)):
            batches = [batches]
        if not self.gathering:
            return
        assert len(batches) <= self.slots
        inf = self.storage.res.info()
        wv_pad = (-(-int(inf.num_cols) // 64) + 1) // 2 * 2
        stride_bytes = wv_pad * 8                   # one bit per sample, exact bitmap or thresholded hit mask
        self._exact = bool(exact)
        self._slot_of = {id(batch): s for s, batch in enumerate(batches)}
        for s, batch in enumerate(batches):
            need = batch.n * stride_bytes
            if self._bufs[s] is None or self._bufs[s].shape[1] != need:      # buffers are kept across calls of one shape
                with self.torch.cuda.stream(self.stream):
                    self._bufs[s] = self.sg.gather_buffer(need, self.device)
            slot = self._bufs[s][self.sg.rank].data_ptr()
            check(_lib.lib().bigsi_hip_batch_set_outputs(batch.b, slot, None))
            check(_lib.lib().bigsi_hip_batch_set_gather_stream(batch.b, self.comm.cuda_stream))
            if exact:
                check(_lib.lib().bigsi_hip_batch_set_gathered_hit_outputs(batch.b, None, None, 0))
            elif self._hitbufs[s] is None:
                self._set_hit_capacity(s, batch, 1 << 16)
            else:
                col, cnt = self._hitbufs[s]
                check(_lib.lib().bigsi_hip_batch_set_gathered_hit_outputs(batch.b, col.data_ptr(), cnt.data_ptr(), col.numel()))
        self.stream.synchronize()

    def _set_hit_capacity(self, s, batch, cap):
        """(Re)allocate slot s's gathered hit list buffers: torch tensors, because the counts get all-reduced."""
        torch = self.torch
        with torch.cuda.stream(self.comm):
            col = torch.zeros(cap, dtype=torch.int32, device=self.device)
            cnt = torch.zeros(cap, dtype=torch.int32, device=self.device)
        self.comm.synchronize()
        self._hitbufs[s] = (col, cnt)
        check(_lib.lib().bigsi_hip_batch_set_gathered_hit_outputs(batch.b, col.data_ptr(), cnt.data_ptr(), cap))

    def _exchange(self, s, batch):
        """On the comm stream: all-gather the bit vectors, compact, and (thresholded) sum the per-hit counts."""
        self.sg.all_gather_in_place(self._bufs[s])
        if self._exact:
            check(_lib.lib().bigsi_hip_batch_compact_gathered(batch.b, self._bufs[s].data_ptr(), self.sg.world, self.shard_cols))
        else:
            check(_lib.lib().bigsi_hip_batch_compact_gathered_masks(batch.b, self._bufs[s].data_ptr(), self.sg.world, self.shard_cols,
                                                                   self.sg.rank))
            if self.sg.dist.is_initialized():
                self.sg.dist.all_reduce(self._hitbufs[s][1], group=self.sg.group)

    def step(self, batches, threshold):
        """Asynchronous.  Alone: K1-K4 of the next batch.  Sharded: K1-K3 on the compute stream, then all-gather of the
        per-sample vectors + K4 over the gathered result on the comm stream."""
        if not isinstance(batches, (list, tuple)):
            batches = [batches]
        s = self._i % len(batches)
        batch = batches[s]
        self._i += 1
        if not self.gathering:
            batch.run(threshold, sparse_counts=True)
            return
        torch = self.torch
        with torch.cuda.stream(self.stream):
            if self._ev_free[s] is not None:
                self.stream.wait_event(self._ev_free[s])        # this slot's previous exchange has released the buffer
            batch.run(threshold, skip_compact=True, sparse_counts=True)
            self._ev_run[s].record(self.stream)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self._ev_run[s])
            self._exchange(s, batch)
            if self._ev_free[s] is None:
                self._ev_free[s] = torch.cuda.Event()
            self._ev_free[s].record(self.comm)

    def fetch(self, batch, slot=None):
        if not self.gathering:
            return batch.hits()
        off = np.zeros(batch.n + 1, np.uint64)
        if not self._exact:
            # caller-owned hit buffers: on overflow grow them and redo this batch's exchange (rare: > capacity hits)
            s = slot if slot is not None else self._slot_of[id(batch)]
            while True:
                cap = self._hitbufs[s][0].numel()
                col = np.zeros(cap, np.uint32)
                cnt = np.zeros(cap, np.uint32)
                rc = _lib.lib().bigsi_hip_batch_fetch_gathered_hits(batch.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
                if rc == _lib.ERR_CAPACITY:
                    self._set_hit_capacity(s, batch, 1 << int(off[-1] - 1).bit_length())
                    with self.torch.cuda.stream(self.comm):
                        self._exchange(s, batch)
                    continue
                check(rc)
                return off, col[: int(off[-1])], cnt[: int(off[-1])]
        cap = 1 << 12
        while True:
            col = np.zeros(cap, np.uint32)
            cnt = np.zeros(cap, np.uint32)
            rc = _lib.lib().bigsi_hip_batch_fetch_gathered_hits(batch.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
            if rc == _lib.ERR_CAPACITY:
                cap = int(off[-1])
                continue
            check(rc)
            return off, col[: int(off[-1])], cnt[: int(off[-1])]


def shard_config(config, rank, world, device=None):
    """`config` for one rank of a sharded index: resident name and snapshot file get a ".shard<r>-of-<w>" suffix, the
    device ordinal is the rank's own.  world == 1 returns the config unchanged apart from the device."""
    cfg = dict(config)
    sc = dict(cfg.get("storage-config") or {})
    if world > 1:
        tag = ".shard%d-of-%d" % (rank, world)
        sc["name"] = str(sc.get("name", "default")) + tag
        if sc.get("filename"):
            sc["filename"] = str(sc["filename"]) + tag
    if device is not None:
        sc["device"] = int(device)
    cfg["storage-config"] = sc
    return cfg


class ShardedBIGSI(object):
    """One BIGSI index whose samples are spread over the ranks of a process group: `search_batch` returns exactly what
    `BIGSI.search` would return on the concatenation of all shards (rank 0's samples first, then rank 1's, ...).

    Every rank calls the same methods with the same arguments (SPMD).  `local` is this rank's ordinary `bigsi_amd.BIGSI`
    over its shard.  Device work and the two collectives per batch are ShardedSearch's; the host side exchanges only
    sample names (once) and, for `score=True`, the presence strings of the hits, each produced on the rank that owns the
    hit's column (k_presence) -- the n x N matrix the reference materialises for scoring (graph/bigsi.py:232-237) never
    exists anywhere."""

    def __init__(self, local, group=None, device=None):
        import torch.distributed as dist
        from .scoring import Scorer
        self.local = local
        self.dist = dist
        self.group = group
        sizes = self._gather(int(local.bitmatrix.num_cols))
        self.shard_cols = max(max(sizes), 1)
        self.shard_sizes = sizes
        names = self._gather([local.colour_to_sample(c) for c in range(local.num_samples)])
        self.names = names                                   # names[shard][local colour]
        self.num_samples = sum(len(n) for n in names)
        self.engine = ShardedSearch(local.storage, self.shard_cols, group=group, device=device, force_gather=True)
        self.scorer = Scorer(self.num_samples)               # DB_SIZE = number of samples of the whole index
        self._batch = None

    # ---- one index spread over the ranks of a torch.distributed.run launch (CLI: --sharded)
    @staticmethod
    def launch(backend=None):
        """(rank, world, torch.device) of this process; joins the default process group from the launcher's environment
        (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) if that has not happened yet.  BIGSI_SHARD_DEVICE overrides the device
        ordinal (several ranks on one GPU, with backend "gloo": RCCL refuses two ranks on one device)."""
        import torch
        import torch.distributed as dist
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        ordinal = int(os.environ.get("BIGSI_SHARD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(ordinal)
        dev = torch.device("cuda", ordinal)
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            backend = backend or os.environ.get("BIGSI_SHARD_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        return rank, world, dev

    @classmethod
    def open(cls, config, backend=None):
        """The shard of `config`'s index that belongs to this rank (see shard_config), wrapped for whole-index queries."""
        from .graph.bigsi import BIGSI
        rank, world, dev = cls.launch(backend)
        return cls(BIGSI(shard_config(config, rank, world, dev.index)), device=dev)

    @classmethod
    def build(cls, config, bloomfilters, samples, backend=None):
        """BIGSI.build (graph/bigsi.py:157-172) with the samples dealt out in contiguous column ranges, rank r keeping
        range r: every rank passes the same full lists (or only its own range filled in -- entries outside a
        rank's range are never touched, so `bloomfilters` may hold None or a file name to load there)."""
        from .bitrow import BitRow
        from .graph.bigsi import BIGSI
        rank, world, dev = cls.launch(backend)
        if len(bloomfilters) != len(samples):
            raise ValueError("There must be the same number of bloomfilters and sample names")
        if len(samples) < world:
            raise ValueError("%d samples cannot be spread over %d shards" % (len(samples), world))
        lo, hi = rank * len(samples) // world, (rank + 1) * len(samples) // world      # sizes differ by at most one, none empty
        n = hi - lo
        cfg = shard_config(config, rank, world, dev.index)
        mine = []
        for b in bloomfilters[lo:lo + n]:
            if isinstance(b, str):
                with open(b, "rb") as f:
                    b = BitRow.frombytes(f.read(), config["m"])
            mine.append(b)
        return cls(BIGSI.build(cfg, mine, list(samples[lo:lo + n])), device=dev)

    def _gather(self, obj):
        if not self.dist.is_initialized():
            return [obj]
        out = [None] * self.dist.get_world_size(self.group)
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def search_batch(self, seqs, threshold=1.0, score=False):
        from .graph.bigsi import BigsiQueryResult
        from .graph.metadata import DELETION_SPECIAL_SAMPLE_NAME
        assert threshold <= 1
        seqs = list(seqs)
        if not seqs:
            return []
        if self._batch is None:
            self._batch = self.local.storage.new_batch(seqs, self.local.kmer_size)
        else:
            self._batch.reload(seqs, self.local.kmer_size)
        batch, sh, exact = self._batch, self.engine, threshold == 1.0
        count_bytes = 2 if max(len(s) for s in seqs) - self.local.kmer_size + 1 < 65536 else 4
        sh.prepare([batch], exact, count_bytes)
        sh.step([batch], threshold)
        off, colours, counts = sh.fetch(batch)
        num_kmers, num_unique, _ = batch.unique()
        rank = sh.sg.rank
        out, wanted = [], []                                 # wanted: (seq index, [local colours on this rank])
        for i in range(len(seqs)):
            u, n = int(num_unique[i]), int(num_kmers[i])
            if u == 0:
                if exact:
                    raise TypeError("reduce() of empty sequence with no initial value")
                raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
            lo, hi = int(off[i]), int(off[i + 1])
            col, cnt = colours[lo:hi].astype(np.int64), counts[lo:hi]
            shard, local_c = col // self.shard_cols, col % self.shard_cols
            if not exact:
                keep = local_c < np.array([len(self.names[s]) for s in shard], dtype=np.int64) if len(shard) else np.zeros(0, bool)
                shard, local_c, cnt = shard[keep], local_c[keep], cnt[keep]
                order = np.argsort(-cnt.astype(np.int64), kind="stable")
                shard, local_c, cnt = shard[order], local_c[order], cnt[order]
            res = [BigsiQueryResult((int(s), int(c)), self.names[int(s)][int(c)], u if exact else int(f), u)
                   for s, c, f in zip(shard, local_c, cnt)]
            if score and res and n == 1:
                raise IndexError("too many indices for array: array is 1-dimensional, but 2 were indexed")
            wanted.append([int(c) for s, c in zip(shard, local_c) if int(s) == rank] if score else [])
            out.append((res, n))
        if score:
            mine = {}
            for i, cols in enumerate(wanted):
                if cols:
                    strings = batch.presence(i, np.array(cols, dtype=np.uint32), out[i][1])
                    mine.update({(i, rank, c): s for c, s in zip(cols, strings)})
            merged = {}
            for part in self._gather(mine):
                merged.update(part)
            for i, (res, n) in enumerate(out):
                for r in res:
                    col = merged[(i, r.colour[0], r.colour[1])]
                    sd = self.scorer.score(col)
                    sd["kmer-presence"] = col
                    r.add_score(sd)
        return [[r.todict() for r in res if r.sample_name != DELETION_SPECIAL_SAMPLE_NAME] for res, _ in out]

    def search(self, seq, threshold=1.0, score=False):
        return self.search_batch([seq], threshold, score)[0]

    def close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None
        self.engine.close()
