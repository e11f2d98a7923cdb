"""Copy an existing BIGSI index into the hip-hbm backend through the storage contract.

`src` is any storage object that speaks the reference's contract (bigsi/storage/base.py:9-151) -- e.g. the reference's
own BerkeleyDBStorage / RocksDBStorage / RedisStorage opened in an environment where that package is installed; nothing
here imports it.  Everything an index consists of is copied: the four index integers (bitmatrix.py:3-4, index.py:10-11),
all m rows ("<row>:bitarray" records are already in the device's byte format, so they are moved verbatim, a block at a
time) and the sample metadata (metadata.py:82-112)."""
import numpy as np

try:          # host-side C++ helpers (bigsi_amd/_results.cpp, built by bigsi_amd/pyext_build.sh); the loops below are their definition
    from . import _results as _ext
except ImportError:
    _ext = None

INDEX_INTS = ("number_of_rows", "number_of_cols", "ksi:bloomfilter_size", "ksi:num_hashes")


class RowUploader(object):
    """Two-deep upload of row blocks: put(ids, block) hands the block to a worker thread (the copy to the device holds no GIL) and
    returns once the block before it is on the device, so that the caller assembles block i + 1 -- reading and parsing records of the
    source store, which is Python work -- while block i travels.  Used by migrate_index and bdb.import_index."""

    def __init__(self, res, overlap=True):
        from concurrent.futures import ThreadPoolExecutor
        self.res, self.pending = res, None
        self.pool = ThreadPoolExecutor(1) if overlap else None

    def put(self, ids, block):
        if self.pool is None:
            self.res.put_rows(ids, block)
            return
        if self.pending is not None:
            self.pending.result()
        self.pending = self.pool.submit(self.res.put_rows, ids, block)

    def close(self):
        try:
            if self.pending is not None:
                self.pending.result()
        finally:
            self.pending = None
            if self.pool is not None:
                self.pool.shutdown()


def migrate_index(src, dst, block_rows=None, overlap=True):
    """Returns (num_rows, num_cols, num_samples)."""
    m = src.get_integer("number_of_rows")
    n = src.get_integer("number_of_cols")
    dst.delete_all()
    dst.set_integer("ksi:bloomfilter_size", src.get_integer("ksi:bloomfilter_size"))
    dst.set_integer("ksi:num_hashes", src.get_integer("ksi:num_hashes"))
    dst.set_integer("number_of_rows", m)
    rb = (n + 7) // 8
    step = block_rows or max(1, (256 << 20) // max(rb, 1))
    up = RowUploader(dst.res, overlap) if hasattr(dst, "res") else None
    try:
        _copy_rows(src, dst, up, m, n, rb, step)
    finally:
        if up is not None:
            up.close()
    dst.set_integer("number_of_cols", n)
    try:
        ns = src.get_integer("metadata:colour_count")
    except KeyError:
        ns = 0
    for c in range(ns):
        name = src.get_string("metadata:%d" % c)
        dst.set_string("metadata:%d" % c, name)
        try:
            dst.set_integer("metadata:%s" % name, src.get_integer("metadata:%s" % name))
        except KeyError:
            pass
    if ns:
        dst.set_integer("metadata:colour_count", ns)
    dst.sync()
    return m, n, ns


def _copy_rows(src, dst, up, m, n, rb, step):
    blocks = [None, None]          # two blocks used alternately (the uploader holds one): a fresh 256 MB array per block is 65 k page faults
    for bi, r0 in enumerate(range(0, m, step)):
        ids = list(range(r0, min(m, r0 + step)))
        keys = [src.convert_key_to_bytes(src.convert_to_bitarray_key(i)) if hasattr(src, "convert_key_to_bytes")
                else ("%d:bitarray" % i).encode() for i in ids]
        raws = src.batch_get(keys)
        mask = (0xFF << (8 - n % 8)) & 0xFF if n % 8 else 0xFF      # columns beyond number_of_cols are not part of the index
        if _ext is not None and isinstance(raws, list) and all(type(r_) in (bytes, bytearray) for r_ in raws[:4]):
            # the rows of the block copied (cut / zero-extended to rb bytes) by a few threads below Python (bigsi_amd/_results.cpp)
            if blocks[bi & 1] is None or blocks[bi & 1].shape[0] < len(ids):
                blocks[bi & 1] = np.empty((len(ids), max(rb, 1)), dtype=np.uint8)
            block = blocks[bi & 1][: len(ids)]
            _ext.pack_rows(raws, block, mask)
        else:
            block = np.zeros((len(ids), max(rb, 1)), dtype=np.uint8)
            for j, raw in enumerate(raws):       # rows may be stored shorter or longer than ceil(n/8): pad / trim
                a = np.frombuffer(bytes(raw), dtype=np.uint8)[:rb]
                block[j, : a.size] = a
            if n % 8:
                block[:, rb - 1] &= mask
        if up is not None:
            up.put(np.arange(r0, r0 + len(ids), dtype=np.uint64), block)
        else:
            dst.set_rows_packed(r0, block)
