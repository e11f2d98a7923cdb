"""K-mer signature index over a storage backend (bigsi/graph/index.py:1-80).

`lookup()` is the reference's per-k-mer query: unique k-mers -> canonical -> h hashes -> rows -> AND.  With a fused
backend (hip-hbm) the whole of it runs on the device (bigsi_hip_lookup: k_kmerize + k_lookup); the storage only
returns the AND-ed rows."""
import numpy as np

from ..bitrow import BitRow, row_bytes_of
from ..matrix import BitMatrix

BLOOMFILTER_SIZE_KEY = "ksi:bloomfilter_size"
NUM_HASH_FUNCTS_KEY = "ksi:num_hashes"


def _require_fused(storage):
    if not getattr(storage, "fused", False):
        raise TypeError("bigsi_amd runs the query path on the device and needs a fused storage backend "
                        "(storage-engine: hip-hbm); got %r.  Use the reference package for host-side backends." % (storage,))


class KmerSignatureIndex(object):
    def __init__(self, storage):
        _require_fused(storage)
        self.storage = storage
        self.bitmatrix = BitMatrix(storage)
        self.bloomfilter_size = storage.get_integer(BLOOMFILTER_SIZE_KEY)
        self.num_hashes = storage.get_integer(NUM_HASH_FUNCTS_KEY)

    @classmethod
    def create(cls, storage, bloomfilters, bloomfilter_size, num_hashes, lowmem=False):
        _require_fused(storage)
        blooms = [getattr(bf, "bitarray", bf) for bf in bloomfilters]      # BloomFilter objects or bit rows
        storage.set_integer(BLOOMFILTER_SIZE_KEY, bloomfilter_size)
        storage.set_integer(NUM_HASH_FUNCTS_KEY, num_hashes)
        storage.set_integer("number_of_rows", bloomfilter_size)
        storage.set_integer("number_of_cols", 0)
        # transpose on the device, a slab of filters at a time (the reference materialises an N x m bool array,
        # matrix/transpose.py:37-40)
        # `lowmem` (the reference's low_mem_build, matrix/transpose.py:14-30: transpose in row chunks to bound host memory):
        # here the bound is the host staging slab, 16 MB of filters at a time instead of 128 MB
        nb = (int(bloomfilter_size) + 7) // 8
        slab = max(64, (((16 if lowmem else 128) << 20) // max(nb, 1)) // 64 * 64)
        for c0 in range(0, len(blooms), slab):
            part = blooms[c0:c0 + slab]
            arr = np.zeros((len(part), nb), dtype=np.uint8)
            for i, bf in enumerate(part):
                data = np.frombuffer(row_bytes_of(bf)[0], dtype=np.uint8)[:nb]
                arr[i, : data.size] = data
            storage.insert_columns(c0, arr)
        if not blooms:
            storage.res.ensure_open()
            storage.res.written[:] = True
        storage.set_integer("number_of_cols", len(blooms))
        storage.sync()
        return cls(storage)

    def lookup(self, kmers, remove_trailing_zeros=True):
        if isinstance(kmers, str):
            kmers = [kmers]
        uniq = list(dict.fromkeys(kmers))           # set(kmers): unique query strings, non-canonical keys
        raw = self.storage.lookup_kmers(uniq)
        nbits = self.bitmatrix.num_cols if remove_trailing_zeros else None
        return {km: BitRow.frombytes(raw[km], nbits) for km in uniq}

    def insert_bloom(self, bloomfilter, column_index):
        self.bitmatrix.insert_column(bloomfilter, column_index)

    def merge_indexes(self, ksi):
        """Append the other index's columns to every row (graph/index.py:54-60): one device-to-device kernel when both
        live on the same GPU, else in row blocks through the host."""
        n1, n2 = self.bitmatrix.num_cols, ksi.bitmatrix.num_cols
        same_gpu = getattr(ksi.storage, "fused", False) and ksi.storage.res.device == self.storage.res.device
        if same_gpu:
            self.storage.append_from(ksi.storage)
        else:
            m = self.bloomfilter_size
            rb1, rb2 = (n1 + 7) // 8, (n2 + 7) // 8
            step = max(1, (32 << 20) // max(rb1 + rb2, 1))
            for r0 in range(0, m, step):
                ids = np.arange(r0, min(m, r0 + step), dtype=np.uint64)
                a = np.unpackbits(self.storage.get_rows_packed(ids, max(rb1, 1)), axis=1)[:, :n1]
                b = np.unpackbits(np.stack([np.frombuffer(r.tobytes(), np.uint8) for r in ksi.bitmatrix.get_rows(ids, False)]), axis=1)[:, :n2]
                self.storage.set_rows_packed(r0, np.packbits(np.concatenate([a, b], axis=1), axis=1))
        self.bitmatrix.set_num_cols(n1 + n2)
