"""Copy an existing BIGSI index into the hip-hbm backend through the storage contract.

`src` is any storage object that speaks the reference's contract (bigsi/storage/base.py:9-151) -- e.g. the reference's
own BerkeleyDBStorage / RocksDBStorage / RedisStorage opened in an environment where that package is installed; nothing
here imports it.  Everything an index consists of is copied: the four index integers (bitmatrix.py:3-4, index.py:10-11),
all m rows ("<row>:bitarray" records are already in the device's byte format, so they are moved verbatim, a block at a
time) and the sample metadata (metadata.py:82-112)."""
import numpy as np

INDEX_INTS = ("number_of_rows", "number_of_cols", "ksi:bloomfilter_size", "ksi:num_hashes")


def migrate_index(src, dst, block_rows=None):
    """Returns (num_rows, num_cols, num_samples)."""
    m = src.get_integer("number_of_rows")
    n = src.get_integer("number_of_cols")
    dst.delete_all()
    dst.set_integer("ksi:bloomfilter_size", src.get_integer("ksi:bloomfilter_size"))
    dst.set_integer("ksi:num_hashes", src.get_integer("ksi:num_hashes"))
    dst.set_integer("number_of_rows", m)
    rb = (n + 7) // 8
    step = block_rows or max(1, (64 << 20) // max(rb, 1))
    for r0 in range(0, m, step):
        ids = list(range(r0, min(m, r0 + step)))
        keys = [src.convert_key_to_bytes(src.convert_to_bitarray_key(i)) if hasattr(src, "convert_key_to_bytes")
                else ("%d:bitarray" % i).encode() for i in ids]
        raws = src.batch_get(keys)
        block = np.zeros((len(ids), max(rb, 1)), dtype=np.uint8)
        for j, raw in enumerate(raws):       # rows may be stored shorter or longer than ceil(n/8): pad / trim
            a = np.frombuffer(bytes(raw), dtype=np.uint8)[:rb]
            block[j, : a.size] = a
        if n % 8:
            block[:, rb - 1] &= (0xFF << (8 - n % 8)) & 0xFF      # columns beyond number_of_cols are not part of the index
        dst.set_rows_packed(r0, block)
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
