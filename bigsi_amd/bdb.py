"""Read-only access to BerkeleyDB *hash* files -- the on-disk format of the reference's default backend
(`db.DB().open(filename, None, db.DB_HASH, db.DB_CREATE)`, bigsi/storage/berkeleydb.py:12-19) -- in pure Python, so an
existing index can be loaded into HBM on a machine that has no `bsddb3` / libdb headers (index ingest).

Only what a sequential dump needs is implemented: the metadata page, hash pages, and overflow chains for values larger
than a page (every row of an index with more than a few thousand samples).  Bucket arithmetic is not needed: every
key/data pair lives on exactly one hash page, so scanning all pages visits each pair once.  Layout per Oracle Berkeley
DB's public `db_page.h` (4.x - 6.x, hash version 7-10): 26-byte page header, `inp[]` item offsets growing up, items growing
down from the page end; item type byte H_KEYDATA (1) = inline bytes, H_OFFPAGE (3) = {pgno, total length} of an overflow
chain.  Duplicate sets (H_DUPLICATE / H_OFFDUP) do not occur in BIGSI stores and are rejected.
"""
import struct

HASH_MAGIC = 0x061561
P_HASH_UNSORTED, P_OVERFLOW, P_HASHMETA, P_HASH = 2, 7, 8, 13
H_KEYDATA, H_DUPLICATE, H_OFFPAGE, H_OFFDUP = 1, 2, 3, 4
_HDR = 26


class BdbFormatError(ValueError):
    pass


class BdbHashFile(object):
    def __init__(self, path):
        self.f = open(path, "rb")
        head = self.f.read(512)
        if len(head) < 72:
            raise BdbFormatError("%s: too short for a BerkeleyDB file" % path)
        if struct.unpack_from("<I", head, 12)[0] == HASH_MAGIC:
            self.e = "<"
        elif struct.unpack_from(">I", head, 12)[0] == HASH_MAGIC:
            self.e = ">"
        else:
            raise BdbFormatError("%s: not a BerkeleyDB hash file (magic %s)" % (path, head[12:16].hex()))
        self.version, self.pagesize = struct.unpack_from(self.e + "II", head, 16)
        if head[24] != 0:
            raise BdbFormatError("%s: encrypted databases are not supported" % path)
        if head[25] != P_HASHMETA:
            raise BdbFormatError("%s: page 0 is not a hash metadata page" % path)
        self.last_pgno = struct.unpack_from(self.e + "I", head, 32)[0]
        self.f.seek(0, 2)
        self.n_pages = self.f.tell() // self.pagesize

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _page(self, pgno):
        self.f.seek(pgno * self.pagesize)
        p = self.f.read(self.pagesize)
        if len(p) != self.pagesize:
            raise BdbFormatError("page %d beyond end of file" % pgno)
        return p

    def _overflow(self, pgno, tlen):
        """Bytes of an overflow chain: each page carries hf_offset bytes after the header, next_pgno links the chain."""
        out = bytearray()
        while pgno != 0 and len(out) < tlen:
            p = self._page(pgno)
            if p[25] != P_OVERFLOW:
                raise BdbFormatError("page %d: expected an overflow page, found type %d" % (pgno, p[25]))
            nxt, used = struct.unpack_from(self.e + "I", p, 16)[0], struct.unpack_from(self.e + "H", p, 22)[0]
            out += p[_HDR:_HDR + used]
            pgno = nxt
        if len(out) < tlen:
            raise BdbFormatError("overflow chain ends after %d of %d bytes" % (len(out), tlen))
        return bytes(out[:tlen])

    def _item(self, p, offs, i):
        end = self.pagesize if i == 0 else offs[i - 1]
        start = offs[i]
        kind = p[start]
        if kind == H_KEYDATA:
            return p[start + 1:end]
        if kind == H_OFFPAGE:
            pgno, tlen = struct.unpack_from(self.e + "II", p, start + 4)
            return self._overflow(pgno, tlen)
        raise BdbFormatError("hash item type %d (duplicates) is not supported" % kind)

    def items(self, want_key=None):
        """Generator of (key, value) over the whole file, in page order.  `want_key(key) -> bool` lets a caller skip the
        (possibly large, overflow-resident) values of keys it does not care about."""
        for pgno in range(1, self.n_pages):
            p = self._page(pgno)
            if p[25] not in (P_HASH, P_HASH_UNSORTED):
                continue
            n = struct.unpack_from(self.e + "H", p, 20)[0]
            if n == 0:
                continue
            offs = struct.unpack_from(self.e + "%dH" % n, p, _HDR)
            for i in range(0, n - 1, 2):
                key = self._item(p, offs, i)
                if want_key is not None and not want_key(key):
                    continue
                yield key, self._item(p, offs, i + 1)

    def __iter__(self):
        return self.items()


def read_all(path):
    with BdbHashFile(path) as db:
        return dict(db.items())


def small_records(path, threads=0):
    """({key: value} of every record that is not a "<row>:bitarray" row, number of row records, longest row) of a BerkeleyDB hash
    file, through the library's threaded page scan (bigsi_hip_bdb_small_records: no Python statement per record)."""
    import ctypes as C

    import numpy as np

    from . import _lib
    need, rows, widest = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    L = _lib.lib()
    buf = np.zeros(1 << 20, np.uint8)          # (an index's small records are a few KB: one scan of the file, a second only if they are not)
    rc = L.bigsi_hip_bdb_small_records(path.encode(), _lib.ptr(buf), buf.size, C.byref(need), C.byref(rows), C.byref(widest), int(threads))
    if rc == _lib.ERR_CAPACITY:
        buf = np.zeros(int(need.value), np.uint8)
        rc = L.bigsi_hip_bdb_small_records(path.encode(), _lib.ptr(buf), buf.size, C.byref(need), C.byref(rows), C.byref(widest), int(threads))
    _lib.check(rc)
    raw, out, at = buf.tobytes(), {}, 0
    while at < need.value:
        kl, vl = struct.unpack_from("<II", raw, at)
        out[raw[at + 8:at + 8 + kl]] = raw[at + 8 + kl:at + 8 + kl + vl]
        at += 8 + kl + vl
    return out, int(rows.value), int(widest.value)


def import_index(path, dst, block_bytes=64 << 20, native=True, threads=0, timings=None):
    """Load a v0.3-format BIGSI BerkeleyDB store (keys "<row>:bitarray", "<name>:int", "<name>:string",
    bigsi/storage/base.py:29-36) into a hip-hbm storage.  Returns (num_rows, num_cols).

    native (default): below Python -- the small records through bigsi_hip_bdb_small_records, the rows through
    bigsi_hip_load_rows_file, which recognises the file, locates every row record with one threaded scan of the hash pages and
    feeds the pinned double buffer of the file <-> HBM pipeline from wherever the rows lie.  A store that does not hold all m
    rows, or native=False, takes the Python page walk below (the definition of the format here, and the test oracle of the
    native reader): two sequential scans, small records first, then the rows in blocks."""
    import re

    import numpy as np
    row_key = re.compile(rb"^(\d+):bitarray$")
    dst.delete_all()
    if native and hasattr(dst, "res") and not isinstance(path, bytes):
        import time

        from . import _lib
        t0 = time.perf_counter()
        small, n_row_records, widest = small_records(path, threads)
        t1 = time.perf_counter()
        try:
            m = int(small[b"number_of_rows:int"])
            n = int(small[b"number_of_cols:int"])
        except KeyError as e:
            raise BdbFormatError("%s holds no %s record: not a BIGSI v0.3 index" % (path, e.args[0].decode()))
        if n_row_records == m:
            for k in (b"ksi:bloomfilter_size:int", b"ksi:num_hashes:int", b"number_of_rows:int"):
                if k in small:
                    dst[k] = small[k]
            rb = max((n + 7) // 8, 1)
            dst.set_integer("number_of_cols", n)              # (opens the matrix at its width; the rows below are cut / zero-extended to rb bytes)
            res = dst.res
            res.ensure_open()
            t2 = time.perf_counter()
            io = _lib.IoStats()
            _lib.check(res.fn("load_rows_file")(res.ix, path.encode(), 0, 0, m, rb, int(threads), _lib.C.byref(io)))
            if timings is not None:       # (scripts/import_bench.py: where an import's time goes)
                timings.update(scan_s=t1 - t0, open_s=t2 - t1, load_s=time.perf_counter() - t2, load_file_s=io.file_seconds, threads=int(io.threads))
            # (pad bits of a row's last byte are taken as stored -- zero in anything bitarray.tobytes() wrote; the kernels mask
            # columns beyond number_of_cols out of every result anyway)
            res.written[:] = True
            res.lengths_reset()
            res.uniform_len = rb
            for k, v in small.items():
                if k not in (b"number_of_rows:int", b"number_of_cols:int", b"ksi:bloomfilter_size:int", b"ksi:num_hashes:int"):
                    dst[k] = v
            dst.sync()
            return m, n
        dst.delete_all()
    with BdbHashFile(path) as db:
        small = {k: v for k, v in db.items(want_key=lambda k: not row_key.match(k))}
        try:
            m = int(small[b"number_of_rows:int"])
            n = int(small[b"number_of_cols:int"])
        except KeyError as e:
            raise BdbFormatError("%s holds no %s record: not a BIGSI v0.3 index" % (path, e.args[0].decode()))
        for k in (b"ksi:bloomfilter_size:int", b"ksi:num_hashes:int", b"number_of_rows:int"):
            if k in small:
                dst[k] = small[k]
        rb = max((n + 7) // 8, 1)
        per = max(1, block_bytes // rb)
        ids, blobs = [], []
        from .migrate import RowUploader
        up = RowUploader(dst.res)          # block i travels to the device while the pages of block i + 1 are parsed

        def flush():
            if ids:
                block = np.zeros((len(ids), rb), dtype=np.uint8)
                for j, raw in enumerate(blobs):
                    a = np.frombuffer(raw, dtype=np.uint8)[:rb]
                    block[j, : a.size] = a
                if n % 8:
                    block[:, rb - 1] &= (0xFF << (8 - n % 8)) & 0xFF
                up.put(np.array(ids, dtype=np.uint64), block)
                del ids[:], blobs[:]

        for k, v in db.items(want_key=lambda k: bool(row_key.match(k))):
            r = int(row_key.match(k).group(1))
            if r < m:
                ids.append(r)
                blobs.append(v)
                if len(ids) >= per:
                    flush()
        try:
            flush()
        finally:
            up.close()
        dst.set_integer("number_of_cols", n)
        for k, v in small.items():
            if k not in (b"number_of_rows:int", b"number_of_cols:int", b"ksi:bloomfilter_size:int", b"ksi:num_hashes:int"):
                dst[k] = v
    dst.sync()
    return m, n


def import_v01_index(directory, dst):
    """Load a legacy v0.1 index -- the two-file layout of the reference's `example-data/test-bigsi/` that
    scripts/convert_v01_to_v03.py:23-71 converts: `graph` maps 4-byte big-endian row ids to row bytes, `metadata` holds
    bloom_filter_size / kmer_size / num_hashes / num_colours as 4-byte big-endian integers and "colour<i>" -> sample name.
    Returns (num_rows, num_cols, kmer_size)."""
    import os

    import numpy as np
    meta = read_all(os.path.join(directory, "metadata"))
    be = lambda key: int.from_bytes(meta[key], "big")       # noqa: E731
    m, k, h, n = be(b"bloom_filter_size"), be(b"kmer_size"), be(b"num_hashes"), be(b"num_colours")
    dst.delete_all()
    dst.set_integer("ksi:bloomfilter_size", m)
    dst.set_integer("ksi:num_hashes", h)
    dst.set_integer("number_of_rows", m)
    rb = max((n + 7) // 8, 1)
    block = np.zeros((m, rb), dtype=np.uint8)
    with BdbHashFile(os.path.join(directory, "graph")) as db:
        for key, v in db.items():
            r = int.from_bytes(key, "big")
            if len(key) == 4 and r < m:
                a = np.frombuffer(v, dtype=np.uint8)[:rb]
                block[r, : a.size] = a
    if n % 8:
        block[:, rb - 1] &= (0xFF << (8 - n % 8)) & 0xFF
    dst.set_rows_packed(0, block)
    dst.set_integer("number_of_cols", n)
    for c in range(n):
        name = meta[b"colour%d" % c].decode("utf-8")
        if "DELETE" in name:                      # convert_v01_to_v03.py:55-59
            dst.set_string("metadata:%d" % c, "D3L3T3D")
        else:
            dst.set_string("metadata:%d" % c, name)
            dst.set_integer("metadata:%s" % name, c)
    dst.set_integer("metadata:colour_count", n)
    dst.sync()
    return m, n, k
