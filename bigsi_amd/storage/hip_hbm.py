"""`hip-hbm`: a BIGSI storage backend whose bit matrix lives in MI355X HBM.

Sits beside berkeleydb / rocksdb / redis behind the reference's storage registry
(bigsi/storage/__init__.py:3-19).  "<row>:bitarray" records go to / come from the device matrix through
libbigsi_hip.so (include/bigsi_hip.h); every other record (the four index integers, sample metadata) stays in a
host dict, as they are a few bytes each.  Besides the plain contract the backend advertises `fused = True`:
`BIGSI.search()` / `lookup()` then hand whole queries to the device (`search_batch`, `lookup_kmers`) instead of
fetching rows one key at a time.

storage-config keys
    name         identity of the resident index; like a BerkeleyDB filename or a Redis server, the data
                 outlives any one storage object (BIGSI.build closes and re-opens its storage,
                 bigsi/graph/bigsi.py:171-172).  default "default"
    device       HIP device ordinal.  default 0
    devices      list of HIP device ordinals: ONE index split by column range over several GPUs of this process
                 (bigsi_hip_group_*: one shard per device, RCCL all-gather of the per-sample result bits inside every
                 query batch).  The column capacity (`max_cols`) is then fixed: shard width = ceil(max_cols / n) rounded
                 up to 64, colour c lives on device c // width.  A device may be repeated (tests on a one-GPU box)
    m, h         rows / hashes, if known before the first row arrives (get_storage() copies them from the
                 top-level config); otherwise rows are buffered until `number_of_rows` is stored
    max_cols     initial column capacity (grown on demand by re-striding on the device).  default 1024
    filename     optional snapshot file written by sync() and loaded when the index is not resident
    export       path of an ATTACH FILE this process keeps up to date (written when the index is opened from its snapshot and by
                 every sync()): the hipIpc handle of the resident matrix + its geometry + the host-side records.  While this
                 process lives, any other process opens the same index from it in milliseconds and without a second copy in HBM
    attach       path of such a file: this storage is a READ-ONLY handle onto the index another process holds resident
                 (`python -m bigsi_amd hold` is that process).  Takes precedence over `filename` when the file exists and its owner
                 is alive; otherwise the index is loaded as usual
"""
import json
import os
import re
import shutil
import struct
import threading
import weakref

import numpy as np

from .. import _lib
from .._lib import BigsiHipError, check
from .contract import BaseStorage

_ROW_KEY = re.compile(rb"^(\d+):bitarray$")
_RESIDENT = {}        # name -> _Resident (process-wide, survives storage objects)
_MAGIC = b"BIGSIHBM1\n"


class _Resident(object):
    """The device index plus the host-side records of one named store."""

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self.device = int(cfg.get("device", 0))
        self.devices = [int(d) for d in cfg["devices"]] if cfg.get("devices") else None   # group mode (bigsi_hip_group_*)
        self.kv = {}                 # bytes -> bytes, everything that is not a row
        self.ix = None               # bigsi_hip_index*
        self.m = None
        self.pending = {}            # row -> bytes, only while m is unknown
        self.written = None          # np.bool_[m]: rows that have been stored (KeyError semantics of a KV store)
        self.uniform_len = None      # byte length every stored row was given (None: as wide as the index is now)
        self.rowlen = None           # np.uint32[m] once rows of differing lengths have been stored
        self.batches = weakref.WeakSet()   # live QueryBatch objects: they hold the index handle and must die before it
        self.attached = False        # the matrix belongs to another process (storage-config `attach`): read-only

    @property
    def is_group(self):
        return self.devices is not None

    def fn(self, name):
        """The C entry point `name` for this index: bigsi_hip_<name>, or bigsi_hip_group_<name> for a multi-GPU group."""
        return getattr(_lib.lib(), ("bigsi_hip_group_" if self.is_group else "bigsi_hip_") + name)

    # ---- device lifecycle
    def _hint(self, key):
        v = self.cfg.get(key)
        return int(v) if v is not None else None

    def open(self, m, n_cols=0, cap=None):
        assert self.ix is None
        h = self._hint("h") or 1
        if b"ksi:num_hashes:int" in self.kv:
            h = int(self.kv[b"ksi:num_hashes:int"])
        cap = max(int(cap or 0), n_cols, self._hint("max_cols") or 1024)
        out = _lib.C.c_void_p()
        if self.is_group:
            devs = (_lib.C.c_int * len(self.devices))(*self.devices)
            check(_lib.lib().bigsi_hip_group_open(int(m), int(n_cols), cap, h, devs, len(self.devices), _lib.C.byref(out)))
        else:
            check(_lib.lib().bigsi_hip_open(int(m), int(n_cols), cap, h, self.device, _lib.C.byref(out)))
        self.ix, self.m = out, int(m)
        self.written = np.zeros(self.m, dtype=bool)

    def ensure_open(self):
        if self.ix is None:
            m = self._hint("m")
            if b"number_of_rows:int" in self.kv:
                m = int(self.kv[b"number_of_rows:int"])
            if m is None:
                return False
            n = int(self.kv.get(b"number_of_cols:int", b"0"))
            self.open(m, n)
        return True

    def info(self):
        inf = _lib.GroupInfo() if self.is_group else _lib.Info()
        check(self.fn("get_info")(self.ix, _lib.C.byref(inf)))
        return inf

    def reserve_cols(self, cols):
        """Grow the column capacity (re-striding on the device); a group's capacity is fixed when it is opened."""
        if self.is_group:
            if cols > self.info().col_capacity:
                raise BigsiHipError(_lib.ERR_CAPACITY, "a multi-GPU index cannot grow beyond the max_cols it was opened with "
                                                       "(%d columns asked, capacity %d)" % (cols, self.info().col_capacity))
            return
        check(_lib.lib().bigsi_hip_reserve_cols(self.ix, int(cols)))

    def free(self):
        for b in list(self.batches):
            b.close()
        if self.ix is not None:
            check(self.fn("close")(self.ix))
        self.ix, self.m, self.written, self.rowlen = None, None, None, None
        self.kv, self.pending, self.attached = {}, {}, False

    # ---- rows
    def put_rows(self, row_ids, blobs):
        """row_ids: sequence of ints; blobs: list of bytes (all of one length) or uint8[n, rb]."""
        if not len(row_ids):
            return
        stored_len = None
        if not self.ensure_open():
            for r, b in zip(row_ids, blobs):
                self.pending[int(r)] = bytes(b)
            return
        if isinstance(blobs, np.ndarray):
            packed = np.ascontiguousarray(blobs, dtype=np.uint8)
            rb = packed.shape[1]
        else:
            lens = {len(b) for b in blobs}
            if len(lens) != 1:        # ragged: one call per length class
                for L in lens:
                    sel = [i for i, b in enumerate(blobs) if len(b) == L]
                    self.put_rows([row_ids[i] for i in sel], [blobs[i] for i in sel])
                return
            rb = lens.pop()
            if rb == 0:               # an empty bitarray: the row exists and is all zero
                packed, rb, stored_len = np.zeros((len(blobs), 1), np.uint8), 1, 0
            else:
                packed = np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(len(blobs), rb)
        ids = np.ascontiguousarray(row_ids, dtype=np.uint64)
        if ids.size and int(ids.max()) >= self.m:
            raise KeyError("row %d outside [0, %d)" % (int(ids.max()), self.m))
        if rb * 8 > self.info().col_capacity:
            self.reserve_cols(rb * 8)
        check(self.fn("set_rows")(self.ix, _lib.ptr(ids), ids.size, _lib.ptr(packed), rb))
        idx = ids.astype(np.int64)
        self.written[idx] = True
        self._note_lengths(idx, stored_len if stored_len is not None else rb)

    def get_rows(self, row_ids, rb=None):
        """uint8[n, rb]; rb defaults to ceil(num_cols/8).  KeyError for rows never stored."""
        if not self.ensure_open():
            try:
                return [self.pending[int(r)] for r in row_ids]
            except KeyError as e:
                raise KeyError("%s:bitarray" % e.args[0])
        ids = np.ascontiguousarray(row_ids, dtype=np.uint64)
        bad = [int(r) for r in ids if int(r) >= self.m or not self.written[int(r)]]
        if bad:
            raise KeyError("%d:bitarray" % bad[0])
        if rb is None and self.rowlen is not None:
            # rows were stored with differing byte lengths: a KV store hands each value back as it was stored
            # (bigsi/storage/base.py:96-99) -- fetch at the widest length, cut each row to its own
            lens = self.rowlen[ids.astype(np.int64)]
            wide = max(int(lens.max()) if lens.size else 1, 1)
            full = np.zeros((ids.size, wide), dtype=np.uint8)
            if ids.size:
                check(self.fn("get_rows")(self.ix, _lib.ptr(ids), ids.size, _lib.ptr(full), wide))
            return [full[i, : int(n)].tobytes() for i, n in enumerate(lens)]
        if rb is None:
            rb = max(int(self.info().row_bytes), 1) if self.uniform_len is None else max(int(self.uniform_len), 1)
            cut = self.uniform_len == 0
        else:
            cut = False
        out = np.zeros((ids.size, rb), dtype=np.uint8)
        if ids.size:
            check(self.fn("get_rows")(self.ix, _lib.ptr(ids), ids.size, _lib.ptr(out), rb))
        return [b"" for _ in range(ids.size)] if cut else out

    def _note_lengths(self, idx, n):
        """Remember the byte length rows were stored with.  While every stored row has ONE length (any index built
        through BitMatrix) that is a single integer; only a store that really mixes lengths pays for a per-row array."""
        if self.rowlen is None and (self.uniform_len is None or self.uniform_len == n):
            self.uniform_len = n
            return
        if self.rowlen is None:
            self.rowlen = np.full(self.m, self.uniform_len, dtype=np.uint32)
        self.rowlen[idx] = n

    def lengths_reset(self):
        """Every row now has the index's current width (a device-side build / insert / merge rewrote them all)."""
        self.rowlen, self.uniform_len = None, None

    def flush_pending(self):
        if self.pending and self.ensure_open():
            items = sorted(self.pending.items())
            self.pending = {}
            self.put_rows([r for r, _ in items], [b for _, b in items])

    # ---- the integers the device needs to hear about
    def on_put(self, key, value):
        if key == b"number_of_rows:int":
            m = int(value)
            if self.ix is not None and m != self.m:
                raise BigsiHipError(_lib.ERR_INVALID, "number_of_rows %d differs from the resident index (%d rows)" % (m, self.m))
            self.kv[key] = value
            self.flush_pending()
            return
        self.kv[key] = value
        if key == b"number_of_cols:int" and self.ensure_open():
            n = int(value)
            if n > self.info().col_capacity:
                self.reserve_cols(n)
            check(self.fn("set_num_cols")(self.ix, n))
        elif key == b"ksi:num_hashes:int" and self.ix is not None:
            check(self.fn("set_num_hashes")(self.ix, int(value)))


class TooManyHits(Exception):
    """search_many_scored(max_bits=...): the presence bits of the hits would exceed the caller's limit (the caller scores in slices)."""


class HipHbmStorage(BaseStorage):
    fused = True      # BIGSI.search/lookup may call search_batch / lookup_kmers
    _search_cap = 1 << 12     # hit entries search_batch brings buffers for (grows to what a call needed)
    _one_ws = None            # search_batch_arrays: the argument arrays of single-query calls and their addresses
    _bits_cap = 1 << 16       # bytes of presence bits search_many_scored brings a buffer for

    def __init__(self, storage_config=None):
        self.storage_config = dict(storage_config or {})
        self.name = self.storage_config.get("name", "default")
        self._one_lock = threading.Lock()
        _lib.lib()    # fail loudly, here, if the HIP library has not been built
        res = _RESIDENT.get(self.name)
        if res is not None and self.storage_config.get("replace"):
            # storage-config {"replace": true}: whatever is resident under this name is dropped first (what building over an
            # existing BerkeleyDB file does in the reference) -- the way to give a long-lived process a different index under
            # the same name without recreating the old config just to delete it
            HipHbmStorage.drop(self.name)
            res = None
        if res is not None and res.ix is None and not res.kv and not res.pending:
            res.__init__(self.storage_config)       # an emptied (deleted) store: the name is free to describe something else
        elif res is not None:
            # one name = one resident index: a second config under the same name must describe the same thing, it is not
            # silently served somebody else's matrix (m / h are only compared when both sides state them)
            for key in ("device", "devices", "filename", "m", "h"):
                a, b = res.cfg.get(key), self.storage_config.get(key)
                if key in ("device",):
                    a, b = int(a or 0), int(b or 0)
                if a != b and (key in ("device", "devices", "filename") or (a is not None and b is not None)):
                    raise BigsiHipError(_lib.ERR_STATE, "a resident hip-hbm index named %r already exists with %s=%r (this config says %r): "
                                                        "give the other index its own storage-config name" % (self.name, key, a, b))
        if res is None:
            res = _RESIDENT[self.name] = _Resident(self.storage_config)
            fn, att = self.storage_config.get("filename"), self.storage_config.get("attach")
            if att and _attach(res, att):
                pass                                   # a handle onto the index another process holds resident
            elif fn and os.path.exists(fn):
                _load_snapshot(res, fn, int(self.storage_config.get("io_threads", 0)))
                if self.storage_config.get("export"):
                    _export_attach(res, self.storage_config["export"])
        self.res = res
        self.storage = self       # reference convention: backend.storage[key] (base.py:13-21); routed below

    def __repr__(self):
        return "hip-hbm Storage"

    # ---- raw primitives: rows to the device, the rest to the host dict
    def _get_raw(self, key):
        m = _ROW_KEY.match(key)
        if m:
            rows = self.res.get_rows([int(m.group(1))])
            return bytes(rows[0]) if not isinstance(rows, list) else rows[0]
        return self.res.kv[key]

    def _put_raw(self, key, value):
        m = _ROW_KEY.match(key)
        if m:
            self.res.put_rows([int(m.group(1))], [bytes(value)])
        else:
            self.res.on_put(key, bytes(value))

    def _get_many_raw(self, keys):
        keys = list(keys)
        ms = [_ROW_KEY.match(k) for k in keys]
        if keys and all(ms):
            rows = self.res.get_rows([int(m.group(1)) for m in ms])
            return [bytes(r) for r in rows]
        return [self._get_raw(k) for k in keys]

    def _put_many_raw(self, keys, values):
        keys, values = list(keys), list(values)
        ms = [_ROW_KEY.match(k) for k in keys]
        if keys and all(ms):
            self.res.put_rows([int(m.group(1)) for m in ms], [bytes(v) for v in values])
        else:
            for k, v in zip(keys, values):
                self._put_raw(k, v)

    # allow backend.storage[key] for code written against the reference's convention
    def __contains__(self, key):
        try:
            self._get_raw(self.convert_key_to_bytes(key))
            return True
        except KeyError:
            return False

    @staticmethod
    def drop(name="default"):
        """Free the resident index called `name` (device memory and host keys; its snapshot file, if any, stays) whatever
        configuration it was created with.  True if there was one."""
        res = _RESIDENT.pop(name, None)
        if res is None:
            return False
        res.free()
        return True

    def delete_all(self):
        """BaseStorage.delete_all (bigsi/storage/base.py:132-133; berkeleydb.py removes the file): the resident index AND its
        snapshot file go, so that a later process does not find the deleted index again."""
        attached = self.res.attached
        self.res.free()
        fn = self.storage_config.get("filename")
        if attached:
            return            # (a handle onto somebody else's index: detaching is all this process may do to it)
        for f in ((fn, fn + ".tmp") if fn else ()):
            if os.path.exists(f):
                os.remove(f)
            for d in _data_dirs(f):
                shutil.rmtree(d, ignore_errors=True)

    def sync(self):
        fn = self.storage_config.get("filename")
        if fn and not self.res.attached:
            _save_snapshot(self.res, fn, int(self.storage_config.get("io_threads", 0)))
        if self.res.ix is not None:
            check(self.res.fn("synchronize")(self.res.ix))
        if self.storage_config.get("export") and not self.res.attached and self.res.ix is not None:
            _export_attach(self.res, self.storage_config["export"])

    def export_attach(self, path):
        """Write the attach file of this resident index to `path` (storage-config `export` does it at every sync()): another
        process then opens the index with storage-config {"attach": path} -- bigsi_hip_open_ipc, milliseconds, no second copy in
        HBM -- for as long as THIS process lives and keeps the index."""
        _export_attach(self.res, path)

    def save_snapshot(self, filename, threads=0):
        """Write the resident index to `filename` in the device layout (what sync() does for storage-config `filename`); returns
        the I/O statistics of the matrix part (_lib.IoStats: bytes, seconds, file_seconds, threads)."""
        return _save_snapshot(self.res, filename, threads)

    @staticmethod
    def load_snapshot(storage_config, filename, threads=0):
        """A storage object whose index is loaded from `filename` now (whatever is resident under the config's name is dropped
        first); returns (storage, I/O statistics)."""
        cfg = dict(storage_config, replace=True)
        cfg.pop("filename", None)
        st = HipHbmStorage(cfg)
        stats = _load_snapshot(st.res, filename, threads)
        return st, stats

    def close(self):
        self.res = None      # the resident index stays (see module docstring)

    # ---- bulk / device-side construction (beyond the plain contract)
    def set_rows_packed(self, row0, packed):
        """rows row0 .. row0+n-1 from a uint8[n, rb] array in one call."""
        self.res.put_rows(np.arange(row0, row0 + packed.shape[0], dtype=np.uint64), packed)

    def get_rows_packed(self, row_ids, rb=None):
        return self.res.get_rows(row_ids, rb)

    def insert_column(self, col, bloom_bytes):
        """BitMatrix.insert_column on the device (bigsi/matrix/bitmatrix.py:67-75)."""
        res = self.res
        if not res.ensure_open():
            raise KeyError("number_of_rows:int")
        if col >= res.info().col_capacity:
            res.reserve_cols(max(col + 1, 2 * int(res.info().col_capacity)))
        buf = np.frombuffer(bytes(bloom_bytes), dtype=np.uint8)
        need = (res.m + 7) // 8
        if buf.size < need:
            buf = np.concatenate([buf, np.zeros(need - buf.size, np.uint8)])
        buf = np.ascontiguousarray(buf)
        if res.is_group:
            check(_lib.lib().bigsi_hip_group_insert_columns(res.ix, int(col), 1, _lib.ptr(buf), buf.size))
        else:
            check(_lib.lib().bigsi_hip_insert_column(res.ix, int(col), _lib.ptr(buf)))
        res.written[:] = True
        res.lengths_reset()
        res.kv[b"number_of_cols:int"] = str(int(res.info().num_cols)).encode()

    def insert_columns(self, col0, blooms):
        """n Bloom filters (uint8[n, >= ceil(m/8)]) -> columns [col0, col0+n): the transpose runs on the device."""
        res = self.res
        if not res.ensure_open():
            raise KeyError("number_of_rows:int")
        blooms = np.ascontiguousarray(blooms, dtype=np.uint8)
        n, need = blooms.shape[0], (res.m + 7) // 8
        if blooms.shape[1] < need:
            blooms = np.concatenate([blooms, np.zeros((n, need - blooms.shape[1]), np.uint8)], axis=1)
        if col0 + n > res.info().col_capacity:
            res.reserve_cols(col0 + n)
        check(res.fn("insert_columns")(res.ix, int(col0), n, _lib.ptr(blooms), blooms.shape[1]))
        res.written[:] = True
        res.lengths_reset()
        res.kv[b"number_of_cols:int"] = str(int(res.info().num_cols)).encode()

    def append_from(self, other):
        """Append every column of another resident hip-hbm index (same device, same m), device to device."""
        if self.res.is_group or other.res.is_group:
            raise BigsiHipError(_lib.ERR_STATE, "merge is not available for multi-GPU (devices=[...]) indexes")
        check(_lib.lib().bigsi_hip_append_index(self.handle, other.handle))
        self.res.written[:] = True
        self.res.lengths_reset()
        self.res.kv[b"number_of_cols:int"] = str(int(self.res.info().num_cols)).encode()

    def get_column(self, col):
        res = self.res
        out = np.zeros((res.m + 7) // 8, dtype=np.uint8)
        check(res.fn("get_column")(res.ix, int(col), _lib.ptr(out)))
        return out.tobytes()

    def fill_synthetic(self, seed, shard=0, and_draws=2):
        if self.res.is_group:       # shard i of the group is filled as (seed, shard i)
            check(_lib.lib().bigsi_hip_group_fill_synthetic(self.res.ix, int(seed), int(and_draws)))
        else:
            check(_lib.lib().bigsi_hip_fill_synthetic(self.res.ix, int(seed), int(shard), int(and_draws)))
        self.res.written[:] = True
        self.res.lengths_reset()

    def insert_kmers(self, col, seqs, k):
        blob, off = _lib.pack_seqs(seqs)
        check(self.res.fn("insert_kmers")(self.res.ix, int(col), blob, _lib.ptr(off), len(seqs), int(k)))

    # ---- fused query path
    @property
    def handle(self):
        if not self.res.ensure_open():
            raise KeyError("number_of_rows:int")
        return self.res.ix

    def lookup_kmers(self, kmers):
        """{kmer: row bytes} for a list of distinct k-mer strings (graph/index.py:42-49); any lengths."""
        from ..utils import canonical
        out = {}
        rb = max(int(self.res.info().row_bytes), 1) if self.res.ensure_open() else 1
        by_len = {}
        wide = [km for km in kmers if isinstance(km, str) and not km.isascii()]
        if wide:
            # non-ASCII k-mers (k characters, hashed as UTF-8): canonical form on the host, hashing + row AND on the device
            data = [canonical(km).encode("utf-8") for km in wide]
            off = np.zeros(len(data) + 1, np.uint64)
            off[1:] = np.cumsum([len(d) for d in data])
            rows = np.zeros((len(data), rb), dtype=np.uint8)
            check(self.res.fn("lookup_raw")(self.handle, b"".join(data), _lib.ptr(off), len(data), _lib.ptr(rows)))
            for km, r in zip(wide, rows):
                out[km] = r.tobytes()
        for km in kmers:
            if km not in out:
                by_len.setdefault(len(km), []).append(km)
        for k, group in by_len.items():
            if k == 0:
                raise ValueError("cannot look up an empty k-mer")
            blob, _ = _lib.pack_seqs(group)
            rows = np.zeros((len(group), rb), dtype=np.uint8)
            check(self.res.fn("lookup")(self.handle, blob, k, len(group), _lib.ptr(rows)))
            for km, r in zip(group, rows):
                out[km] = r.tobytes()
        return out

    def new_batch(self, seqs, k):
        return QueryBatch(self, seqs, k)

    def new_element_batch(self, queries):
        """A batch whose k-mers are given explicitly (bigsi_hip_batch_create_elements): `queries` = [(unique canonical k-mers
        as bytes, first-occurrence order; position -> unique index list)] -- the route of non-ASCII sequences."""
        return ElementBatch(self, queries)

    def search_batch(self, seqs, k, threshold=1.0):
        """The fused query path in one call, the shape INTEGRATION.md binds into the reference's BIGSI.search: for every
        sequence (num_kmers, num_unique_kmers, colours, counts) with colours ascending (exact: counts == num_unique;
        thresholded: every colour with count >= ceil(num_unique * threshold), graph/bigsi.py:179,241-242).  Names,
        ordering by count, percentages and scores stay with the caller."""
        assert threshold <= 1                                   # graph/bigsi.py:176
        seqs = list(seqs)
        if not seqs:
            return []
        nk, nu, off, col, cnt = self.search_batch_arrays(seqs, k, threshold)
        # (most sequences of a bulk search match nothing: they share ONE pair of empty arrays instead of 2 slices each, which is
        # where a 1000-read call spent its time once the C call was down to 0.07 ms)
        o, e_col, e_cnt = off.tolist(), col[:0], cnt[:0]
        return [(k_, u_, col[x:y], cnt[x:y]) if y > x else (k_, u_, e_col, e_cnt) for k_, u_, x, y in zip(nk.tolist(), nu.tolist(), o, o[1:])]

    def search_batch_arrays(self, seqs, k, threshold=1.0, flags=0, borrow=False):
        """The same call with its results as the C ABI leaves them: (num_kmers uint32[n], num_unique uint32[n], hit offsets
        uint64[n + 1], colours uint32[hits], counts uint32[hits]).  BIGSI.search / search_batch take this route for unscored
        queries: one C call (bigsi_hip_search_batch) instead of reload + run + two fetches.  `borrow`: the arrays of a single-query
        call may be views of buffers this object keeps for the next such call (for a caller that is done with them before it calls
        again: BIGSI.search, under its lock)."""
        n = len(seqs)
        if n == 1 and type(seqs[0]) is str and not self.res.is_group and self._one_lock.acquire(False):
            # ONE query -- the latency-bound call: its argument arrays and their addresses are kept between calls (numpy's
            # .ctypes.data is ~1 us per pointer, fresh zeroed outputs another 2: together a third of what the C call itself takes)
            try:
                ws = self._one_ws
                if ws is None or ws[0] != self._search_cap:
                    cap = self._search_cap
                    arrs = (np.zeros(2, np.uint64), np.zeros(1, np.uint32), np.zeros(1, np.uint32), np.zeros(2, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32))
                    ws = self._one_ws = (cap, arrs, tuple(_lib.ptr(a) for a in arrs), _lib.lib().bigsi_hip_search_batch)
                cap, (soff, nk, nu, off, col, cnt), (p_soff, p_nk, p_nu, p_off, p_col, p_cnt), fn = ws
                try:
                    blob = seqs[0].encode("ascii")
                except UnicodeEncodeError:
                    raise ValueError("query sequences must be ASCII for the hip-hbm backend")
                soff[1] = len(blob)
                rc = fn(self.handle, blob, p_soff, 1, int(k), float(threshold), int(flags), p_nk, p_nu, None, p_off, p_col, p_cnt, cap)
                if rc == _lib.OK:
                    h = int(off[1])
                    if borrow:
                        return nk, nu, off, col[:h], cnt[:h]
                    return nk.copy(), nu.copy(), off.copy(), col[:h].copy(), cnt[:h].copy()
                if rc != _lib.ERR_CAPACITY or int(off[1]) <= cap:
                    check(rc)
                self._search_cap = int(off[1])            # more hits than the buffers hold: the general route below brings that many
            finally:
                self._one_lock.release()
        # the C ABI's one-call entry point (bigsi_hip_search_batch / bigsi_hip_group_search_batch): the index keeps the
        # workspace, so a call is two uploads, the kernels and three downloads -- no device allocation
        blob, soff = _lib.pack_seqs(seqs)
        fn = getattr(_lib.lib(), "bigsi_hip_group_search_batch" if self.res.is_group else "bigsi_hip_search_batch")
        nk, nu = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        off = np.zeros(n + 1, np.uint64)
        cap = self._search_cap
        while True:
            col, cnt = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            rc = fn(self.handle, blob, _lib.ptr(soff), n, int(k), float(threshold), int(flags), _lib.ptr(nk), _lib.ptr(nu), None,
                    _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
            if rc != _lib.ERR_CAPACITY or int(off[-1]) <= cap:      # (any other CAPACITY error -- a result wider than the row stride -- is not ours to retry)
                break
            cap = self._search_cap = int(off[-1])             # offsets are filled in: bring that much next time
        check(rc)
        return nk, nu, off, col[: int(off[-1])], cnt[: int(off[-1])]


    def search_many(self, seqs, k, threshold=1.0):
        """bigsi_hip_search_stream: any number of sequences in ONE call -- the library cuts them into device batches and keeps four
        in flight (upload / kernels / export overlap).  Returns (num_kmers, num_unique, hit_offsets, colours, counts): sequence i
        matched colours[hit_offsets[i]:hit_offsets[i+1]] (ascending) with that many of its unique k-mers.  Single index only."""
        assert threshold <= 1
        seqs = seqs if isinstance(seqs, (list, tuple)) else list(seqs)
        n = len(seqs)
        nk, nu = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        off = np.zeros(n + 1, np.uint64)
        if n == 0:
            return nk, nu, off, np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        if self.res.is_group:
            raise BigsiHipError(_lib.ERR_STATE, "search_many is not available on a multi-GPU index")
        blob, soff = _lib.pack_seqs(seqs)
        return self.search_many_packed(blob, soff, k, threshold)

    def search_many_packed(self, blob, soff, k, threshold=1.0):
        """search_many for sequences that are already packed: `blob` (bytes or a uint8 array) holds sequence i at
        soff[i]:soff[i+1] (uint64, n + 1 entries) -- what bigsi_hip_fasta_pack leaves."""
        import ctypes as C
        assert threshold <= 1
        n = len(soff) - 1
        nk, nu = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        off = np.zeros(n + 1, np.uint64)
        if n == 0:
            return nk, nu, off, np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        if self.res.is_group:
            raise BigsiHipError(_lib.ERR_STATE, "search_many is not available on a multi-GPU index")
        text = blob if isinstance(blob, bytes) else C.cast(_lib.ptr(blob), C.c_char_p)
        cap = max(self._search_cap, 1 << 12)
        while True:
            col, cnt = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            rc = _lib.lib().bigsi_hip_search_stream(self.handle, text, _lib.ptr(soff), n, int(k), float(threshold), 0, _lib.ptr(nk), _lib.ptr(nu), None,
                                                    _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
            if rc != _lib.ERR_CAPACITY or int(off[-1]) <= cap:
                break
            cap = self._search_cap = int(off[-1])
        check(rc)
        total = int(off[-1])
        return nk, nu, off, col[:total], cnt[:total]

    def search_many_scored(self, seqs, k, threshold=1.0, max_bits=None, packed=None):
        """bigsi_hip_search_stream_scored: search_many plus, per hit, the presence bits and the score record of score=True
        (bigsi/scoring/score.py:96-121), K5 + K6 of one device batch running beside the row-AND kernels of the next.  Returns
        (num_kmers, num_unique, hit_offsets, colours, counts, bits, bit_offsets, scores): hit t's presence string is
        the slice of scoring.unpack_presence(bits, bit_offsets) that starts at character 8 * bit_offsets[t], num_kmers of its sequence long; scores is a HIT_SCORE_DTYPE array.
        `packed` = (blob, offsets) of the sequences if the caller has packed them already (_lib.pack_seqs; `seqs` is then not read)."""
        from bigsi_amd.scoring import HIT_SCORE_DTYPE
        assert threshold <= 1
        if packed is None:
            seqs = seqs if isinstance(seqs, (list, tuple)) else list(seqs)
        n = len(seqs) if packed is None else len(packed[1]) - 1
        nk, nu = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        off = np.zeros(n + 1, np.uint64)
        if n == 0:
            return nk, nu, off, np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, HIT_SCORE_DTYPE)
        if self.res.is_group:
            raise BigsiHipError(_lib.ERR_STATE, "search_many_scored is not available on a multi-GPU index")
        blob, soff = packed if packed is not None else _lib.pack_seqs(seqs)
        if not isinstance(blob, bytes):
            import ctypes as C
            blob = C.cast(_lib.ptr(blob), C.c_char_p)          # (a uint8 array, e.g. what bigsi_hip_fasta_pack left)
        cap, bcap = max(self._search_cap, 1 << 12), max(self._bits_cap, 1 << 16)
        need = np.zeros(1, np.uint64)
        while True:
            col, cnt = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            bits, boff, rec = np.zeros(bcap, np.uint8), np.zeros(cap + 1, np.uint64), np.zeros(cap, HIT_SCORE_DTYPE)
            rc = _lib.lib().bigsi_hip_search_stream_scored(self.handle, blob, _lib.ptr(soff), n, int(k), float(threshold), 0, _lib.ptr(nk), _lib.ptr(nu),
                                                           None, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap, _lib.ptr(bits), bcap,
                                                           _lib.ptr(boff), _lib.ptr(rec), _lib.ptr(need))
            if rc != _lib.ERR_CAPACITY or (int(off[-1]) <= cap and int(need[0]) <= bcap):
                break
            if max_bits is not None and int(need[0]) > max_bits:
                raise TooManyHits(int(need[0]))
            cap = self._search_cap = max(cap, int(off[-1]))
            bcap = self._bits_cap = max(bcap, int(need[0]))
        check(rc)
        total = int(off[-1])
        return nk, nu, off, col[:total], cnt[:total], bits[:int(need[0])], boff[:total + 1], rec[:total]



class QueryBatch(object):
    """A batch of query sequences staged on the device (bigsi_hip_batch)."""

    def __init__(self, storage, seqs, k):
        self.storage = storage
        self.n = len(seqs)
        self.group = storage.res.is_group          # a batch over all shards of a multi-GPU index (bigsi_hip_group_batch)
        blob, off = _lib.pack_seqs(seqs)
        self._off = off
        out = _lib.C.c_void_p()
        check(self._fn("create")(storage.handle, blob, _lib.ptr(off), self.n, int(k), _lib.C.byref(out)))
        self.b = out
        self.k = int(k)
        storage.res.batches.add(self)

    def _fn(self, name):
        return getattr(_lib.lib(), ("bigsi_hip_group_batch_" if self.group else "bigsi_hip_batch_") + name)

    def _single_only(self, what):
        if self.group:
            raise BigsiHipError(_lib.ERR_STATE, "%s is not available on a multi-GPU batch (use the shard's own index)" % what)

    def reload(self, seqs, k=None):
        """Stage a different set of sequences in this batch object, keeping its device buffers (bigsi_hip_batch_reload)."""
        blob, off = _lib.pack_seqs(seqs)
        self.k = int(k) if k is not None else self.k
        check(self._fn("reload")(self.b, blob, _lib.ptr(off), len(seqs), self.k))
        self.n, self._off = len(seqs), off
        return self

    def close(self):
        if self.b is not None:
            # (a finaliser may run this while another thread of the process is inside a call on the same index handle: the library's
            # destroy waits for that thread -- every other entry point refuses a second thread)
            check(self._fn("destroy")(self.b))
            self.b = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def run(self, threshold, force_counts=False, skip_compact=False, k1_global=False, sparse_counts=False, early_exit=False,
            weak_fingerprint=False, one_stream=False):
        flags = ((_lib.RUN_ONE_STREAM if one_stream else 0) | (_lib.RUN_FORCE_COUNTS if force_counts else 0) | (_lib.RUN_SKIP_COMPACT if skip_compact else 0) |
                 (_lib.RUN_K1_GLOBAL if k1_global else 0) | (_lib.RUN_SPARSE_COUNTS if sparse_counts else 0) |
                 (_lib.RUN_EARLY_EXIT if early_exit else 0) | (_lib.RUN_WEAK_FINGERPRINT if weak_fingerprint else 0))
        if self.group:
            flags &= ~(_lib.RUN_SKIP_COMPACT | _lib.RUN_SPARSE_COUNTS)      # the group run sets what it needs itself
        check(self._fn("run")(self.b, float(threshold), flags))

    def info(self):
        self._single_only("info()")
        inf = _lib.BatchInfo()
        check(_lib.lib().bigsi_hip_batch_get_info(self.b, _lib.C.byref(inf)))
        return inf

    def unique(self):
        nk = np.zeros(self.n, np.uint32)
        nu = np.zeros(self.n, np.uint32)
        mk = np.zeros(self.n, np.uint32)
        check(self._fn("fetch_unique")(self.b, _lib.ptr(nk), _lib.ptr(nu), _lib.ptr(mk)))
        return nk, nu, mk

    def hits(self):
        """(offsets uint64[n+1], colours uint32[], counts uint32[]), ascending colour inside each sequence."""
        off = np.zeros(self.n + 1, np.uint64)
        cap = 1 << 12
        while True:
            col = np.zeros(cap, np.uint32)
            cnt = np.zeros(cap, np.uint32)
            rc = self._fn("fetch_hits")(self.b, _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), cap)
            if rc == _lib.ERR_CAPACITY and int(off[-1]) > cap:
                cap = int(off[-1])
                continue
            check(rc)
            total = int(off[-1])
            return off, col[:total], cnt[:total]

    def counts(self, i):
        self._single_only("counts()")
        out = np.zeros(int(self.storage.res.info().num_cols), np.uint32)
        check(_lib.lib().bigsi_hip_batch_fetch_counts(self.b, i, _lib.ptr(out)))
        return out

    def bitmap(self, i):
        self._single_only("bitmap()")
        out = np.zeros(max(int(self.storage.res.info().row_bytes), 1), np.uint8)
        check(_lib.lib().bigsi_hip_batch_fetch_bitmap(self.b, i, _lib.ptr(out)))
        return out

    def rows(self, i, num_unique):
        self._single_only("rows()")
        h = int(self.storage.res.info().num_hashes)
        out = np.zeros((max(int(num_unique), 1), h), np.uint64)
        check(_lib.lib().bigsi_hip_batch_fetch_rows(self.b, i, _lib.ptr(out), out.size))
        return out[:int(num_unique)]

    def lookup(self, i, num_unique):
        self._single_only("lookup()")
        rb = max(int(self.storage.res.info().row_bytes), 1)
        first = np.zeros(max(int(num_unique), 1), np.uint32)
        rows = np.zeros((max(int(num_unique), 1), rb), np.uint8)
        check(_lib.lib().bigsi_hip_batch_lookup(self.b, i, _lib.ptr(first), _lib.ptr(rows), rows.shape[0]))
        return first[:int(num_unique)], rows[:int(num_unique)]

    def presence(self, i, colours, num_kmers):
        """['0101..', ...]: one presence string (length num_kmers) per colour (graph/bigsi.py:232-237)."""
        colours = np.ascontiguousarray(colours, dtype=np.uint32)
        if colours.size == 0 or num_kmers == 0:
            return ["" for _ in colours]
        out = np.zeros((colours.size, int(num_kmers)), np.uint8)
        check(self._fn("presence")(self.b, i, _lib.ptr(colours), colours.size, _lib.ptr(out)))
        return [r.tobytes().decode("ascii") for r in out]


    def presence_hits(self, off, colours, num_kmers):
        """Presence strings for the hits of EVERY sequence in one device pass (bigsi_hip_batch_presence_hits): `off`, `colours` as
        hits() returns them (any colour order inside a sequence).  Returns (blob uint8[], starts int64[n_hits], lengths
        int64[n_hits]): the string of hit t is blob[starts[t] : starts[t] + lengths[t]] (starts are 16-byte aligned)."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        colours = np.ascontiguousarray(colours, dtype=np.uint32)
        n_hits = int(off[self.n] - off[0])
        lens = np.repeat(np.asarray(num_kmers[: self.n], dtype=np.int64), np.diff(off[: self.n + 1]).astype(np.int64))
        blob = np.zeros(max(int(((lens + 15) // 16 * 16).sum()), 16), np.uint8)
        soff = np.zeros(n_hits + 1, np.uint64)
        check(self._fn("presence_hits")(self.b, _lib.ptr(off), _lib.ptr(colours) if n_hits else None, _lib.ptr(blob), blob.size, _lib.ptr(soff)))
        return blob, soff[:-1].astype(np.int64), lens


    def score_hits(self, off, colours, counts, num_kmers):
        """BIGSI.score for the hits of EVERY sequence on the device (K6, bigsi_hip_batch_score_hits): `off`, `colours`, `counts` as
        hits() returns them (counts None: an exact search).  Returns (records, bits, bit_offsets): records[t]
        (scoring.HIT_SCORE_DTYPE) holds what calculate_score computes for hit t plus its percent_kmers_found; hit t's presence
        string is bits[bit_offsets[t]:] read as a bitarray of num_kmers positions (scoring.unpack_presence)."""
        from ..scoring import HIT_SCORE_DTYPE
        off = np.ascontiguousarray(off, dtype=np.uint64)
        colours = np.ascontiguousarray(colours, dtype=np.uint32)
        counts = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
        n_hits = int(off[self.n] - off[0])
        words = (np.asarray(num_kmers[: self.n], dtype=np.int64) + 63) // 64
        need = int((words * np.diff(off[: self.n + 1]).astype(np.int64)).sum()) * 8
        bits = np.zeros(max(need, 8), np.uint8)
        boff = np.zeros(n_hits + 1, np.uint64)
        rec = np.zeros(max(n_hits, 1), HIT_SCORE_DTYPE)
        check(self._fn("score_hits")(self.b, _lib.ptr(off), _lib.ptr(colours) if n_hits else None, _lib.ptr(counts) if n_hits else None,
                                     _lib.ptr(bits), bits.size, _lib.ptr(boff), _lib.ptr(rec)))
        return rec[:n_hits], bits, boff


    def score_hits_begin(self, off, colours, counts, num_kmers, ordered=False):
        """First half of score_hits (bigsi_hip_batch_score_hits_begin): queues K5 + K6 for the given hit lists and returns at once.
        ordered=True puts the kernels on the index's stream behind the runs already issued (BIGSI_SCORE_ORDERED: a throughput
        loop three batches deep), otherwise on the library's high-priority score stream.  Single index only."""
        self._single_only("score_hits_begin()")
        off = np.ascontiguousarray(off, dtype=np.uint64)
        colours = np.ascontiguousarray(colours, dtype=np.uint32)
        counts = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
        n_hits = int(off[self.n] - off[0])
        boff = np.zeros(n_hits + 1, np.uint64)
        check(_lib.lib().bigsi_hip_batch_score_hits_begin(self.b, _lib.ptr(off), _lib.ptr(colours) if n_hits else None,
                                                          _lib.ptr(counts) if n_hits else None, _lib.SCORE_ORDERED if ordered else 0, _lib.ptr(boff)))
        self._score_job = (n_hits, boff)

    def score_hits_end(self):
        """Second half: waits for the request and returns (records, bits, bit_offsets) as score_hits does."""
        from ..scoring import HIT_SCORE_DTYPE
        n_hits, boff = self._score_job
        self._score_job = None
        bits = np.zeros(max(int(boff[-1]), 8), np.uint8)
        rec = np.zeros(max(n_hits, 1), HIT_SCORE_DTYPE)
        check(_lib.lib().bigsi_hip_batch_score_hits_end(self.b, _lib.ptr(bits), bits.size, _lib.ptr(rec)))
        return rec[:n_hits], bits, boff


class ElementBatch(QueryBatch):
    """QueryBatch over explicit k-mers: same run / unique / hits / presence interface, no reload."""

    def __init__(self, storage, queries):      # noqa: D107 - does not call QueryBatch.__init__ (different constructor on the C side)
        self.storage, self.n, self.group, self.k = storage, len(queries), storage.res.is_group, 0
        elems = [e for uniq, _ in queries for e in uniq]
        eoff = np.zeros(len(elems) + 1, np.uint64)
        eoff[1:] = np.cumsum([len(e) for e in elems])
        seoff = np.zeros(self.n + 1, np.uint64)
        seoff[1:] = np.cumsum([len(uniq) for uniq, _ in queries])
        spoff = np.zeros(self.n + 1, np.uint64)
        spoff[1:] = np.cumsum([len(pu) for _, pu in queries])
        pu = np.ascontiguousarray([j for _, p in queries for j in p], dtype=np.uint32)
        out = _lib.C.c_void_p()
        check(self._fn("create_elements")(storage.handle, b"".join(elems), _lib.ptr(eoff), _lib.ptr(seoff),
                                          _lib.ptr(pu) if pu.size else None, _lib.ptr(spoff), self.n, _lib.C.byref(out)))
        self.b = out
        storage.res.batches.add(self)

    def reload(self, seqs, k=None):
        raise BigsiHipError(_lib.ERR_STATE, "a batch of explicit k-mers cannot be reloaded")


# ------------------------------------------------------------------------------- snapshots (sync / reopen)
# Version 2 (written): the DEVICE layout -- [magic | header length | JSON header (host records, geometry) | written bitmap, raw |
# per-row lengths, raw, if any | zero padding to a 4096-byte boundary | m rows of row_stride_bytes each], so that save and load are
# bigsi_hip_save_rows_file / bigsi_hip_load_rows_file: threads on the file, two pinned buffers, asynchronous copies, no kernel
# and no Python in the data path.  Version 1 (rows at ceil(num_cols / 8) bytes, hex bitmap in the header, 64 MB synchronous
# blocks) is still read.
_MAGIC2 = b"BIGSIHBM2\n"
_ALIGN = 4096
_STRIPE_FROM = 256 << 20      # matrices from this size on are saved as striped part files
_ATTACH_FORMAT = "bigsi-hip-attach-1"


def _export_attach(res, path):
    """The attach file (JSON): hipIpc handle of the matrix, geometry, the host-side records (index integers, sample metadata),
    which rows have been written.  Written to a temporary name and renamed: a reader never sees half a file."""
    if res.attached:
        raise BigsiHipError(_lib.ERR_STATE, "an attached index cannot be re-exported (only its owner can)")
    if not res.ensure_open():
        raise BigsiHipError(_lib.ERR_STATE, "nothing resident to export")
    inf = res.info()
    n_sh = int(inf.n_shards) if res.is_group else 1
    handle = (_lib.C.c_uint8 * (64 * n_sh))()
    check(res.fn("export_ipc")(res.ix, handle))          # (a group: one hipIpc handle per shard, in shard order)
    doc = {"format": _ATTACH_FORMAT, "handle": bytes(handle).hex(), "pid": os.getpid(), "device": res.device, "devices": res.devices,
           "m": int(inf.num_rows), "num_cols": int(inf.num_cols), "col_capacity": int(inf.col_capacity), "num_hashes": int(inf.num_hashes),
           "kv": {k.decode("latin-1"): v.decode("latin-1") for k, v in res.kv.items()}, "uniform_len": res.uniform_len,
           "written": None if res.written.all() else np.packbits(res.written).tobytes().hex(),
           "rowlen": None if res.rowlen is None else res.rowlen.tobytes().hex()}
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(doc, f)
    os.replace(tmp, path)


def _attach(res, path):
    """Open `res` as a read-only handle onto the index described by the attach file; False (and nothing changed) when there is
    no such file or its owner is gone -- the caller then loads the index the usual way."""
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return False
    if doc.get("format") != _ATTACH_FORMAT:
        raise BigsiHipError(_lib.ERR_INVALID, "%s is not a hip-hbm attach file" % path)
    try:
        os.kill(int(doc["pid"]), 0)
    except ProcessLookupError:
        return False                                  # a stale file: the owner has exited and its allocation with it
    except PermissionError:
        pass                                          # alive, another user's
    if int(doc["pid"]) == os.getpid():
        raise BigsiHipError(_lib.ERR_STATE, "%s was exported by this very process: use the resident index (same storage-config name)" % path)
    raw = bytes.fromhex(doc["handle"])
    handle = (_lib.C.c_uint8 * len(raw)).from_buffer_copy(raw)
    out = _lib.C.c_void_p()
    device = int(res.cfg.get("device", doc.get("device", 0)))
    try:
        _attach_open(res, path, doc, raw, handle, out, device)
    except BigsiHipError as e:
        if e.code != _lib.ERR_HIP:
            raise
        # the handle did not open -- typically a file whose owner died and whose pid now belongs to another process: as for a stale
        # file, the index loads the usual way
        import warnings
        warnings.warn("%s: could not attach (%s); loading the index instead" % (path, e))
        return False
    res.ix, res.m, res.device, res.attached = out, int(doc["m"]), device, True
    res.kv = {k.encode("latin-1"): v.encode("latin-1") for k, v in doc["kv"].items()}
    res.written = np.ones(res.m, dtype=bool) if not doc.get("written") else np.unpackbits(np.frombuffer(bytes.fromhex(doc["written"]), np.uint8))[: res.m].astype(bool)
    res.uniform_len = doc.get("uniform_len")
    res.rowlen = np.frombuffer(bytes.fromhex(doc["rowlen"]), np.uint32).copy() if doc.get("rowlen") else None
    return True


def _attach_open(res, path, doc, raw, handle, out, device):
    if doc.get("devices"):
        # a multi-GPU index: one handle per shard; this process names its own devices (default: the owner's) in the same shard order
        devs = [int(d) for d in (res.cfg.get("devices") or doc["devices"])]
        if len(devs) * 64 != len(raw):
            raise BigsiHipError(_lib.ERR_INVALID, "%s holds %d shard handle(s), storage-config `devices` names %d device(s)" % (path, len(raw) // 64, len(devs)))
        res.devices = devs
        arr = (_lib.C.c_int * len(devs))(*devs)
        check(_lib.lib().bigsi_hip_group_open_ipc(handle, int(doc["m"]), int(doc["num_cols"]), int(doc["col_capacity"]), int(doc["num_hashes"]), arr, len(devs), _lib.C.byref(out)))
    else:
        if res.is_group:
            raise BigsiHipError(_lib.ERR_STATE, "%s describes a single-GPU index; storage-config `devices` does not apply" % path)
        check(_lib.lib().bigsi_hip_open_ipc(handle, int(doc["m"]), int(doc["num_cols"]), int(doc["col_capacity"]), int(doc["num_hashes"]), device, _lib.C.byref(out)))


def _save_snapshot(res, fn, threads=0):
    """Returns the I/O statistics of the matrix part (None for an index without a matrix)."""
    header = {"kv": {k.decode("latin-1"): v.decode("latin-1") for k, v in res.kv.items()}, "m": res.m, "stride": 0}
    tmp = fn + ".tmp"
    stats = None
    with open(tmp, "wb") as f:
        extra = b""
        if res.ix is not None:
            inf = res.info()
            if res.is_group:
                # a multi-GPU index: whole rows in the reference's row format at a pitch of 8-byte multiples (shard i owns bytes
                # [i * shard_cols / 8, ...) of every row; bigsi_hip_group_save_rows_file gathers them with one 2-D copy per shard)
                pitch = max(int(inf.row_bytes), int(res.uniform_len or 0), int(res.rowlen.max()) if res.rowlen is not None else 0, 1)
                pitch = min(-(-pitch // 8) * 8, int(inf.col_capacity) // 8)
            else:
                pitch = int(inf.row_stride_bytes)
            header.update(stride=pitch, num_cols=int(inf.num_cols), uniform_len=res.uniform_len,
                          written_bytes=(res.m + 7) // 8, rowlen_bytes=0 if res.rowlen is None else int(res.rowlen.nbytes),
                          all_written=bool(res.written.all()))
            if not header["all_written"]:
                extra += np.packbits(res.written).tobytes()
            else:
                header["written_bytes"] = 0
            if res.rowlen is not None:
                extra += res.rowlen.tobytes()
            # a large matrix goes into a DIRECTORY of striped part files beside the header file (`fn`.d/): writes to one file are
            # serialised by its inode -- 3.6-5 GB/s however many threads -- and 16 files take the PCIe rate (csrc: BigsiRowsFile)
            header["striped"] = res.m * header["stride"] >= _STRIPE_FROM
        hb = json.dumps(header).encode("utf-8")
        head = _MAGIC2 + struct.pack("<Q", len(hb)) + hb + extra
        data_off = -(-len(head) // _ALIGN) * _ALIGN
        f.write(head + b"\0" * (data_off - len(head)))
    if res.ix is not None:
        stats = _lib.IoStats()
        if header["striped"]:
            # The part files go into a directory of their OWN that the header names (`data_dir`), alternating between two names, and
            # the header is replaced -- atomically -- only when they are complete: a crash at any point leaves the previous header
            # beside the previous, untouched directory (round-5 advisor: renaming directories around one fixed name left a window in
            # which the old header sat next to the new part files).  The directory not named by the header is garbage and is removed.
            data_dir = next(d for d in _data_dirs(fn) if os.path.basename(d) != _current_data_dir(fn))
            shutil.rmtree(data_dir, ignore_errors=True)
            check(res.fn("save_rows_file")(res.ix, (data_dir + "/").encode(), 0, 0, res.m, header["stride"], int(threads), _lib.C.byref(stats)))
            header["data_dir"] = os.path.basename(data_dir)
            hb = json.dumps(header).encode("utf-8")
            with open(tmp, "wb") as f:
                f.write(_MAGIC2 + struct.pack("<Q", len(hb)) + hb + extra)
        else:
            check(res.fn("save_rows_file")(res.ix, tmp.encode(), data_off, 0, res.m, header["stride"], int(threads), _lib.C.byref(stats)))
    os.replace(tmp, fn)
    for d in _data_dirs(fn):
        if os.path.basename(d) != header.get("data_dir"):
            shutil.rmtree(d, ignore_errors=True)
    return stats


def _data_dirs(fn):
    """The two names a striped snapshot's part-file directory alternates between."""
    return [fn + ".d", fn + ".1.d"]


def _current_data_dir(fn):
    """basename of the directory the snapshot header at `fn` names, or None (no snapshot, not striped, unreadable)"""
    try:
        with open(fn, "rb") as f:
            if f.read(len(_MAGIC2)) != _MAGIC2:
                return None
            (n,) = struct.unpack("<Q", f.read(8))
            header = json.loads(f.read(n).decode("utf-8"))
        return header.get("data_dir", os.path.basename(fn) + ".d") if header.get("striped") else None
    except (OSError, ValueError, struct.error):
        return None


def _load_snapshot(res, fn, threads=0):
    with open(fn, "rb") as f:
        magic = f.read(len(_MAGIC))
        if magic == _MAGIC2:
            (n,) = struct.unpack("<Q", f.read(8))
            header = json.loads(f.read(n).decode("utf-8"))
            res.kv = {k.encode("latin-1"): v.encode("latin-1") for k, v in header["kv"].items()}
            if not (header.get("m") and header.get("stride")):
                return None
            m, stride = int(header["m"]), int(header["stride"])
            wb, lb = int(header.get("written_bytes", 0)), int(header.get("rowlen_bytes", 0))
            written = np.unpackbits(np.frombuffer(f.read(wb), np.uint8))[:m].astype(bool) if wb else None
            rowlen = np.frombuffer(f.read(lb), np.uint32).copy() if lb else None
            data_off = -(-f.tell() // _ALIGN) * _ALIGN
            # (a single-GPU snapshot loads into a group and the other way round: the file holds whole rows either way, at the
            # writer's pitch; whatever the pitch pads beyond the columns is zero)
            res.open(m, int(header.get("num_cols", 0)), cap=max(stride * 8 if not res.is_group else 0, int(header.get("num_cols", 0))))
            have = int(res.info().col_capacity) // 8 if res.is_group else int(res.info().row_stride_bytes)
            if have < (-(-int(header.get("num_cols", 0)) // 8) if res.is_group else stride):
                raise BigsiHipError(_lib.ERR_CAPACITY, "%s holds rows of %d bytes, the index was opened with room for %d" % (fn, stride, have))
            stats = _lib.IoStats()
            if header.get("striped"):
                data_dir = os.path.join(os.path.dirname(fn), header.get("data_dir", os.path.basename(fn) + ".d"))
                check(res.fn("load_rows_file")(res.ix, (data_dir + "/").encode(), 0, 0, m, stride, int(threads), _lib.C.byref(stats)))
            else:
                check(res.fn("load_rows_file")(res.ix, fn.encode(), data_off, 0, m, stride, int(threads), _lib.C.byref(stats)))
            res.written = written if written is not None else np.ones(m, dtype=bool)
            res.uniform_len, res.rowlen = header.get("uniform_len"), rowlen
            return stats
        if magic != _MAGIC:
            raise BigsiHipError(_lib.ERR_INVALID, "%s is not a hip-hbm snapshot" % fn)
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n).decode("utf-8"))
        res.kv = {k.encode("latin-1"): v.encode("latin-1") for k, v in header["kv"].items()}
        if header.get("m") and header.get("rb"):
            m, rb = int(header["m"]), int(header["rb"])
            n_cols = int(res.kv.get(b"number_of_cols:int", b"0"))
            res.open(m, n_cols, cap=rb * 8)
            # (an old file still loads at the new rate: rows of rb bytes through the scatter kernel, or 2-D copies into a group)
            check(res.fn("load_rows_file")(res.ix, fn.encode(), f.tell(), 0, m, rb, int(threads), None))
            res.written = np.unpackbits(np.frombuffer(bytes.fromhex(header["written"]), np.uint8))[:m].astype(bool)
            res.uniform_len = header.get("uniform_len")
            if header.get("rowlen"):
                res.rowlen = np.frombuffer(bytes.fromhex(header["rowlen"]), np.uint32).copy()
    return None
