"""Column-range sharding of one BIGSI index over the GPUs of a node.

Every stage of the query path is independent per sample column -- hashing depends only on the query, AND and
counting are per column -- so rank g (one process per GPU) holds ALL m rows of columns
[g * shard_cols, (g+1) * shard_cols) and runs K1-K3 on its slice for the same query batch.  The only exchange is
one all-gather per batch of ONE BIT PER SAMPLE -- the AND bitmap of an exact search, or the `count >= min_kmers` hit mask
the counting kernel leaves for a thresholded one -- RCCL over xGMI, issued by libbigsi_hip.so itself on its own
communicator (include/bigsi_hip.h, "column shards"); torch.distributed is the launcher-side plumbing (rendezvous, the
128-byte communicator id, barriers).  The local vector is written by the kernels straight into this rank's slot of the
gather buffer, so the collective runs in place; compaction to (colour, count) lists then runs on every
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
    """Query batches against this rank's column shard + the exchange that makes every rank see the whole result.

    exchange="rccl" (default whenever the process group's backend is RCCL, or there is no process group): the library
    itself issues ncclAllGather / ncclAllReduce on its own communicator and stream (bigsi_hip_comm_init_rank,
    bigsi_hip_batch_run_sharded); torch.distributed only carries the 128-byte communicator id to the other ranks once.
        index stream:  run(i) ---------------- run(i+1) ------------- run(i+2) ...
        comm stream:          gather(i) K4g(i)          gather(i+1) K4g(i+1)
    With two batches used alternately the exchange of batch i overlaps the row-AND kernels of batch i+1; events inside the
    library order the two streams, the host never waits between steps.

    exchange="torch": the same schedule with torch.distributed collectives on two torch streams -- for process groups
    whose backend is not RCCL (gloo: several ranks sharing one GPU in tests; RCCL refuses two ranks on one device).
    (torch's default stream has handle 0, which the C ABI reserves for "use the library's private stream", hence the
    explicit side streams.)"""

    def __init__(self, storage, shard_cols, group=None, device=None, force_gather=False, slots=2, exchange=None):
        import torch
        self.torch = torch
        self.force_gather = force_gather      # take the gather path even with one rank (testing)
        self.storage = storage
        self.shard_cols = int(shard_cols)
        self.sg = ShardGroup(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if exchange is None:
            backend = self.sg.dist.get_backend(group) if self.sg.dist.is_initialized() else "nccl"
            exchange = "rccl" if "nccl" in str(backend) else "torch"
        assert exchange in ("rccl", "torch")
        self.exchange = exchange
        self.comm = None
        self.slots = int(slots)
        self._exact = True
        self._i = 0
        if not self.gathering:
            return
        L = _lib.lib()
        inf = storage.res.info()
        if inf.col_capacity < self.shard_cols:       # every shard is as wide as the widest: one geometry for the exchange
            check(L.bigsi_hip_reserve_cols(storage.handle, self.shard_cols))
        if exchange == "rccl":
            ident = np.zeros(128, np.uint8)
            ok, err = 1, None
            try:
                if self.sg.rank == 0:
                    check(L.bigsi_hip_comm_unique_id(_lib.ptr(ident)))
            except _lib.BigsiHipError as e:
                ok, err = 0, e
            multi = self.sg.dist.is_initialized() and self.sg.world > 1
            if multi:
                box = [ident.tobytes() if ok else None]
                self.sg.dist.broadcast_object_list(box, src=self.sg.dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                ok = int(box[0] is not None)
                if ok:
                    ident = np.frombuffer(box[0], np.uint8).copy()
            if ok:
                out = _lib.C.c_void_p()
                try:
                    check(L.bigsi_hip_comm_init_rank(int(self.device.index), _lib.ptr(ident), self.sg.rank, self.sg.world, _lib.C.byref(out)))
                    self.comm = out
                except _lib.BigsiHipError as e:
                    ok, err = 0, e
            if multi:
                # every rank must take the same route: if the library's communicator did not come up everywhere, all fall
                # back to torch.distributed's collectives (same RCCL underneath, torch's communicator)
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
                self.sg.dist.all_reduce(flag, op=self.sg.dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if not ok:
                if not multi:
                    raise err
                import warnings
                warnings.warn("libbigsi_hip could not bring up its own RCCL communicator on every rank (%s): "
                              "falling back to torch.distributed collectives" % (err,))
                if self.comm is not None:
                    check(L.bigsi_hip_comm_destroy(self.comm))
                    self.comm = None
                self.exchange = exchange = "torch"
        if exchange == "torch":
            self.stream = torch.cuda.Stream(self.device)
            self.comm_stream = torch.cuda.Stream(self.device)
            check(L.bigsi_hip_set_stream(storage.handle, self.stream.cuda_stream))
            self._bufs = [None] * self.slots
            self._hitbufs = [None] * self.slots       # (colours, counts) int32 tensors of the gathered hit lists, per slot
            self._slot_of = {}
            self._ev_run = [torch.cuda.Event() for _ in range(self.slots)]
            self._ev_free = [None] * self.slots       # recorded on comm when a slot's gather + compaction are done

    @property
    def gathering(self):
        return self.sg.world > 1 or self.force_gather

    def comm_ranks(self):
        """(rank, world) as the library's RCCL communicator reports them; None without one."""
        if self.comm is None:
            return None
        r, w = _lib.C.c_int(0), _lib.C.c_int(0)
        check(_lib.lib().bigsi_hip_comm_info(self.comm, _lib.C.byref(r), _lib.C.byref(w)))
        return r.value, w.value

    def close(self):
        if not self.gathering:
            return
        if self.exchange == "torch":
            check(_lib.lib().bigsi_hip_set_stream(self.storage.handle, None))
        elif self.comm is not None:
            for b in list(self.storage.res.batches):
                if b.b is not None:
                    check(_lib.lib().bigsi_hip_batch_set_comm(b.b, None, 0))
            check(_lib.lib().bigsi_hip_comm_destroy(self.comm))
            self.comm = None

    def prepare(self, batches, exact, count_bytes=2):
        """Attach each batch (one per slot) to the exchange: results `shard_cols` wide, written straight into this rank's
        slot of a [world, n_seqs * stride] gather buffer."""
        if not isinstance(batches, (list, tuple)):
            batches = [batches]
        if not self.gathering:
            return
        self._exact = bool(exact)
        L = _lib.lib()
        if self.exchange == "rccl":
            for batch in batches:
                check(L.bigsi_hip_batch_set_comm(batch.b, self.comm, self.shard_cols))
            return
        assert len(batches) <= self.slots
        wv_pad = (-(-self.shard_cols // 64) + 1) // 2 * 2     # from the group's shard width, NOT this rank's own num_cols
        stride_bytes = wv_pad * 8                   # one bit per sample, exact bitmap or thresholded hit mask
        self._slot_of = {id(batch): s for s, batch in enumerate(batches)}
        for s, batch in enumerate(batches):
            check(L.bigsi_hip_batch_set_result_cols(batch.b, self.shard_cols))
            need = batch.n * stride_bytes
            if self._bufs[s] is None or self._bufs[s].shape[1] != need:      # buffers are kept across calls of one shape
                with self.torch.cuda.stream(self.stream):
                    self._bufs[s] = self.sg.gather_buffer(need, self.device)
            slot = self._bufs[s][self.sg.rank].data_ptr()
            check(L.bigsi_hip_batch_set_outputs(batch.b, slot, None))
            check(L.bigsi_hip_batch_set_gather_stream(batch.b, self.comm_stream.cuda_stream))
            if exact:
                check(L.bigsi_hip_batch_set_gathered_hit_outputs(batch.b, None, None, 0))
            elif self._hitbufs[s] is None:
                self._set_hit_capacity(s, batch, 1 << 16)
            else:
                col, cnt = self._hitbufs[s]
                check(L.bigsi_hip_batch_set_gathered_hit_outputs(batch.b, col.data_ptr(), cnt.data_ptr(), col.numel()))
        self.stream.synchronize()

    def _set_hit_capacity(self, s, batch, cap):
        """(Re)allocate slot s's gathered hit list buffers: torch tensors, because the counts get all-reduced."""
        torch = self.torch
        with torch.cuda.stream(self.comm_stream):
            col = torch.zeros(cap, dtype=torch.int32, device=self.device)
            cnt = torch.zeros(cap, dtype=torch.int32, device=self.device)
        self.comm_stream.synchronize()
        self._hitbufs[s] = (col, cnt)
        check(_lib.lib().bigsi_hip_batch_set_gathered_hit_outputs(batch.b, col.data_ptr(), cnt.data_ptr(), cap))

    def _exchange(self, s, batch):
        """torch exchange, on the comm stream: all-gather the bit vectors, compact, and (thresholded) sum the per-hit counts."""
        self.sg.all_gather_in_place(self._bufs[s])
        if self._exact:
            check(_lib.lib().bigsi_hip_batch_compact_gathered(batch.b, self._bufs[s].data_ptr(), self.sg.world, self.shard_cols))
        else:
            check(_lib.lib().bigsi_hip_batch_compact_gathered_masks(batch.b, self._bufs[s].data_ptr(), self.sg.world, self.shard_cols,
                                                                   self.sg.rank))
            if self.sg.dist.is_initialized():
                self.sg.dist.all_reduce(self._hitbufs[s][1], group=self.sg.group)

    def step(self, batches, threshold, early_exit=False):
        """Asynchronous.  Alone: K1-K4 of the next batch.  Sharded: K1-K3, then all-gather of the per-sample vectors + K4
        over the gathered result on the communicator's stream."""
        if not isinstance(batches, (list, tuple)):
            batches = [batches]
        s = self._i % len(batches)
        batch = batches[s]
        self._i += 1
        if not self.gathering:
            batch.run(threshold, sparse_counts=True, early_exit=early_exit)
            return
        if self.exchange == "rccl":
            check(_lib.lib().bigsi_hip_batch_run_sharded(batch.b, float(threshold), _lib.RUN_EARLY_EXIT if early_exit else 0))
            return
        torch = self.torch
        with torch.cuda.stream(self.stream):
            if self._ev_free[s] is not None:
                self.stream.wait_event(self._ev_free[s])        # this slot's previous exchange has released the buffer
            batch.run(threshold, skip_compact=True, sparse_counts=True, early_exit=early_exit)
            self._ev_run[s].record(self.stream)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self._ev_run[s])
            self._exchange(s, batch)
            if self._ev_free[s] is None:
                self._ev_free[s] = torch.cuda.Event()
            self._ev_free[s].record(self.comm_stream)

    def fetch(self, batch, slot=None):
        if not self.gathering:
            return batch.hits()
        off = np.zeros(batch.n + 1, np.uint64)
        if self.exchange == "torch" and not self._exact:
            # caller-owned hit buffers: on overflow grow them and redo this batch's exchange (rare: > capacity hits)
            s = slot if slot is not None else self._slot_of[id(batch)]
            while True:
                cap = self._hitbufs[s][0].numel()
                col = np.zeros(cap, np.uint32)
                cnt = np.zeros(cap, np.uint32)
                rc = _lib.lib().bigsi_hip_batch_fetch_gathered_hits(batch.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
                if rc == _lib.ERR_CAPACITY:
                    self._set_hit_capacity(s, batch, 1 << int(off[-1] - 1).bit_length())
                    with self.torch.cuda.stream(self.comm_stream):
                        self._exchange(s, batch)
                    continue
                check(rc)
                return off, col[: int(off[-1])], cnt[: int(off[-1])]
        # library-owned hit lists (they grow inside the library; with RCCL every rank re-reduces the counts identically)
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
        self._batches = {}                                   # workspace slot -> QueryBatch (two, alternating, in search_stream)

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

    # ---- queries: submit (asynchronous device work + exchange) / collect (fetch, names, scores), two workspaces deep
    def _submit(self, slot, seqs, threshold):
        batch = self._batches.get(slot)
        if batch is None:
            batch = self._batches[slot] = self.local.storage.new_batch(seqs, self.local.kmer_size)
        else:
            batch.reload(seqs, self.local.kmer_size)
        workspaces = [self._batches[s] for s in sorted(self._batches)]
        exact = threshold == 1.0
        count_bytes = 2 if max(len(s) for s in seqs) - self.local.kmer_size + 1 < 65536 else 4
        self.engine.prepare(workspaces, exact, count_bytes)
        self.engine._i = sorted(self._batches).index(slot)       # the engine steps the workspace we just loaded
        self.engine.step(workspaces, threshold)
        return batch

    def _search_wide(self, seqs, threshold, score):
        """Sequences with non-ASCII characters: an explicit-k-mer batch (BIGSI._elements_of) through the same exchange."""
        batch = self.local.storage.new_element_batch([self.local._elements_of(s) for s in seqs])
        try:
            count_bytes = 2 if max(len(s) for s in seqs) - self.local.kmer_size + 1 < 65536 else 4
            self.engine.prepare([batch], threshold == 1.0, count_bytes)
            self.engine._i = 0
            self.engine.step([batch], threshold)
            return self._collect(batch, len(seqs), threshold, score)
        finally:
            batch.close()

    def _collect(self, batch, n_seqs, threshold, score):
        from .graph.bigsi import BigsiQueryResult
        from .graph.metadata import DELETION_SPECIAL_SAMPLE_NAME
        sh, exact = self.engine, threshold == 1.0
        off, colours, counts = sh.fetch(batch)
        num_kmers, num_unique, _ = batch.unique()
        rank = sh.sg.rank
        out = []
        for i in range(n_seqs):
            u, n = int(num_unique[i]), int(num_kmers[i])
            if u == 0:
                if exact:
                    raise TypeError("reduce() of empty sequence with no initial value")
                raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
            lo, hi = int(off[i]), int(off[i + 1])
            col, cnt, pos = colours[lo:hi].astype(np.int64), counts[lo:hi], np.arange(lo, hi)
            shard, local_c = col // self.shard_cols, col % self.shard_cols
            if not exact:
                keep = local_c < np.array([len(self.names[s]) for s in shard], dtype=np.int64) if len(shard) else np.zeros(0, bool)
                shard, local_c, cnt, pos = shard[keep], local_c[keep], cnt[keep], pos[keep]
                order = np.argsort(-cnt.astype(np.int64), kind="stable")
                shard, local_c, cnt, pos = shard[order], local_c[order], cnt[order], pos[order]
            res = [BigsiQueryResult((int(s), int(c)), self.names[int(s)][int(c)], u if exact else int(f), u)
                   for s, c, f in zip(shard, local_c, cnt)]
            if score and res and n == 1:
                raise IndexError("too many indices for array: array is 1-dimensional, but 2 were indexed")
            out.append((res, pos))
        if score:
            # graph/bigsi.py:232-239 for every hit of the batch: ONE K5 + K6 pass on each rank over the hits whose columns it owns
            # (bigsi_hip_batch_score_hits), then one exchange of (hit positions, scored rows)
            owned = (colours.astype(np.int64) // self.shard_cols) == rank
            csum = np.concatenate([[0], np.cumsum(owned)])
            off_own = csum[off.astype(np.int64)].astype(np.uint64)
            col_own = (colours[owned].astype(np.int64) - rank * self.shard_cols).astype(np.uint32)
            cnt_own = None if exact else counts[owned]
            from .graph.bigsi import score_hit_rows
            from .scoring import SCORE_KEYS
            # K6 on the rank that owns the hit's column: presence bits, run tallies and the rounded score chain on the device
            mine = (np.flatnonzero(owned), score_hit_rows(batch, off_own, col_own, cnt_own, num_kmers, n_seqs, self.scorer.DB_SIZE))
            scored = {}
            for where, rows in self._gather(mine):
                scored.update(zip(where.tolist(), rows))
            for res, pos in out:
                for r, t in zip(res, pos.tolist()):
                    _, fields, col = scored[t]
                    sd = dict(zip(SCORE_KEYS, fields))
                    sd["kmer-presence"] = col
                    r.add_score(sd)
        return [[r.todict() for r in res if r.sample_name != DELETION_SPECIAL_SAMPLE_NAME] for res, _ in out]

    def search_batch(self, seqs, threshold=1.0, score=False):
        assert threshold <= 1
        seqs = list(seqs)
        if not seqs:
            return []
        wide = [i for i, s in enumerate(seqs) if not s.isascii()]
        if wide:
            out = [None] * len(seqs)
            rest = [i for i in range(len(seqs)) if seqs[i].isascii()]
            for i, r in zip(wide, self._search_wide([seqs[i] for i in wide], threshold, score)):
                out[i] = r
            if rest:
                for i, r in zip(rest, self.search_batch([seqs[i] for i in rest], threshold, score)):
                    out[i] = r
            return out
        return self._collect(self._submit(0, seqs, threshold), len(seqs), threshold, score)

    def search_stream(self, seqs, threshold=1.0, score=False, batch_size=None, batch_kmers=1 << 19):
        """Generator over (sequence, results), two workspaces deep like BIGSI.search_stream: while the GPUs run (and exchange)
        batch i+1, the host fetches and assembles batch i.  Every rank iterates it in step (SPMD).  The two workspaces are the
        object's own (they are bound to the exchange): finish consuming the stream before calling search() / search_batch()
        on the same object."""
        assert threshold <= 1
        pending, slot, chunk, held = None, 0, [], 0
        k = self.local.kmer_size

        def flush(chunk):
            """Submit `chunk`; yield the results of the batch before it (or, for a chunk holding non-ASCII sequences -- which
            rebinds the exchange -- drain the pipeline and answer the chunk at once)."""
            nonlocal pending, slot
            if not all(s.isascii() for s in chunk):
                if pending is not None:
                    yield from zip(pending[1], self._collect(pending[0], len(pending[1]), threshold, score))
                    pending = None
                yield from zip(chunk, self.search_batch(chunk, threshold, score))
                return
            nxt = (self._submit(slot, chunk, threshold), chunk)
            if pending is not None:
                yield from zip(pending[1], self._collect(pending[0], len(pending[1]), threshold, score))
            pending, slot = nxt, slot ^ 1

        for s in seqs:
            chunk.append(s)
            held += max(len(s) - k + 1, 1)
            if (len(chunk) == batch_size) if batch_size else (held >= batch_kmers):
                yield from flush(chunk)
                chunk, held = [], 0
        if chunk:
            yield from flush(chunk)
        if pending is not None:
            yield from zip(pending[1], self._collect(pending[0], len(pending[1]), threshold, score))

    def search(self, seq, threshold=1.0, score=False):
        return self.search_batch([seq], threshold, score)[0]

    def close(self):
        for batch in self._batches.values():
            batch.close()
        self._batches = {}
        self.engine.close()
