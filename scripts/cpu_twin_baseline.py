"""CPU baseline THROUGH the product boundary: libbigsi_cpu.so (include/bigsi_cpu.h), the CPU twin of the C ABI, timed on the host
cores by bench.py's cpu_baseline leg (a subprocess: its fork pool must not inherit a HIP context).

  1. one core, the reference's shape (per-k-mer canonicalisation on strings, MurmurHash3 x h, one copy per fetched row, byte-wise
     AND, unpack-to-int32-and-add: bigsi_cpu_search_batch) for --seconds;
  2. the same in a fork pool of one worker per physical core over query sequences -- the reference's only parallelism
     (bulk_search, bigsi/__main__.py:273-287);
  3. BIGSI_CPU_WORD_PARALLEL (64-bit words of the resident rows, no copies) on one core and in the same fork pool: the "best CPU" lines.
The index is the GPU run's synthetic index (same generator, same seed) at full row width but only --rows rows, so that it fits
host RAM; per-lookup work is identical.  Rows are served from RAM, which favours the CPU over the reference's BerkeleyDB.
Prints one JSON object."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_T = {}


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _work(job):
    wid, seconds, flags = job
    a, L, ix, packs = _T["args"], _T["lib"], _T["ix"], _T["packs"]
    mine = packs[wid::a.threads] or packs
    nk, nu = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
    ho = np.zeros(2, np.uint64)
    col, cnt = np.zeros(1 << 16, np.uint32), np.zeros(1 << 16, np.uint32)
    done, t0, i = 0, time.time(), 0
    thr = 1.0 if a.exact else a.threshold
    while time.time() - t0 < seconds:
        blob, off = mine[i % len(mine)]
        rc = L.bigsi_cpu_search_batch(ix, blob, ptr(off), C.c_uint32(1), C.c_uint32(a.k), C.c_double(thr), C.c_uint32(flags), ptr(nk), ptr(nu), None,
                                      ptr(ho), ptr(col), ptr(cnt), C.c_uint64(col.size))
        assert rc in (0, -5), L.bigsi_cpu_last_error()
        done += int(nu[0])
        i += 1
    return done, time.time() - t0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=200_000)
    p.add_argument("--cols", type=int, required=True)
    p.add_argument("--hashes", type=int, required=True)
    p.add_argument("--k", type=int, default=31)
    p.add_argument("--and-draws", type=int, default=2)
    p.add_argument("--seed", type=int, required=True)
    p.add_argument("--batch", type=int, required=True)
    p.add_argument("--qlen", type=int, required=True)
    p.add_argument("--exact", type=int, default=1)
    p.add_argument("--threshold", type=float, default=0.4)
    p.add_argument("--seconds", type=float, default=3.0)
    p.add_argument("--threads", type=int, default=0)
    p.add_argument("--pool-runs", type=int, default=2)
    p.add_argument("--bdb", type=int, default=1, help="also time the searches with the rows served from a BerkeleyDB hash file of the same index (0 = skip)")
    a = p.parse_args()
    if a.threads <= 0:
        a.threads = max(1, (os.cpu_count() or 2) // 2)
    L = C.CDLL(os.path.join(ROOT, "bigsi_amd", "libbigsi_cpu.so"))
    L.bigsi_cpu_last_error.restype = C.c_char_p
    ix = C.c_void_p()
    t0 = time.time()
    assert L.bigsi_cpu_open(C.c_uint64(a.rows), C.c_uint64(a.cols), C.c_uint64(a.cols), C.c_uint32(a.hashes), 0, C.byref(ix)) == 0, L.bigsi_cpu_last_error()
    assert L.bigsi_cpu_fill_synthetic(ix, C.c_uint64(a.seed), C.c_uint64(0), C.c_uint32(a.and_draws)) == 0
    fill_s = time.time() - t0
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)              # same queries as bench.py: rand_seqs(default_rng(1), ...)
    seqs = [lut[r].tobytes() for r in np.random.default_rng(1).integers(0, 4, size=(a.batch, a.qlen), dtype=np.uint8)]
    packs = [(s, np.array([0, len(s)], np.uint64)) for s in seqs]
    _T.update(args=a, lib=L, ix=ix, packs=packs)
    one_done, one_t = _work((0, a.seconds, 0))
    wp_done, wp_t = _work((0, max(a.seconds / 2, 1.0), 1 << 16))
    rates = []
    with mp.get_context("fork").Pool(a.threads) as pool:     # workers inherit the index copy-on-write
        for _ in range(max(1, a.pool_runs)):
            res = pool.map(_work, [(w, a.seconds, 0) for w in range(a.threads)])
            rates.append(sum(r[0] for r in res) / max(r[1] for r in res))
    with mp.get_context("fork").Pool(a.threads) as pool:     # the "best CPU" line: word-parallel mode on every physical core
        res = pool.map(_work, [(w, a.seconds, 1 << 16) for w in range(a.threads)])
        wp_pool = sum(r[0] for r in res) / max(r[1] for r in res)
    # the same searches with the rows served from a BerkeleyDB hash FILE -- the reference's own store (bigsi/storage/berkeleydb.py:6-19),
    # written here by libdb itself (dbm.ndbm) into tmpfs, read by the twin's own page walk (no bsddb3 / libdb headers on these hosts:
    # a table from one scan of the hash pages stands in for libdb's bucket lookup): bigsi_cpu_open_bdb
    bdb = None
    if a.bdb:
        try:
            import dbm.ndbm as ndbm
            import shutil
            import tempfile
            if getattr(ndbm, "library", "") == "Berkeley DB":
                d = tempfile.mkdtemp(prefix="bigsi_cpu_bdb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
                try:
                    t0 = time.time()
                    rb = (a.cols + 7) // 8
                    db = ndbm.open(os.path.join(d, "store"), "n")
                    for key, v in (("number_of_rows:int", a.rows), ("number_of_cols:int", a.cols), ("ksi:bloomfilter_size:int", a.rows), ("ksi:num_hashes:int", a.hashes)):
                        db[key] = str(v)
                    step = max(1, (64 << 20) // rb)
                    for r0 in range(0, a.rows, step):
                        ids = np.arange(r0, min(a.rows, r0 + step), dtype=np.uint64)
                        blk = np.zeros((ids.size, rb), np.uint8)
                        assert L.bigsi_cpu_get_rows(ix, ptr(ids), C.c_uint64(ids.size), ptr(blk), C.c_uint64(rb)) == 0
                        for j, r in enumerate(ids.tolist()):
                            db["%d:bitarray" % r] = blk[j].tobytes()
                    db.close()
                    write_s = time.time() - t0
                    fn = os.path.join(d, "store.db")
                    bix = C.c_void_p()
                    t0 = time.time()
                    assert L.bigsi_cpu_open_bdb(fn.encode(), C.c_uint32(0), C.byref(bix)) == 0, L.bigsi_cpu_last_error()
                    open_s = time.time() - t0
                    ram_ix = _T["ix"]
                    _T["ix"] = bix
                    b_done, b_t = _work((0, max(a.seconds / 2, 1.0), 0))
                    with mp.get_context("fork").Pool(a.threads) as pool:
                        res = pool.map(_work, [(w, max(a.seconds / 2, 1.0), 0) for w in range(a.threads)])
                        b_pool = sum(r[0] for r in res) / max(r[1] for r in res)
                    _T["ix"] = ram_ix
                    L.bigsi_cpu_close(bix)
                    bdb = {"one_core": {"lookups": b_done, "seconds": b_t, "rate": b_done / b_t}, "pool": {"threads": a.threads, "rate": b_pool},
                           "file_gb": os.path.getsize(fn) / 1e9, "write_seconds": write_s, "open_seconds": open_s, "dir": d,
                           "what": "rows read from a BerkeleyDB hash file (written by libdb through dbm.ndbm into tmpfs; read by the twin's page walk: bigsi_cpu_open_bdb)"}
                finally:
                    shutil.rmtree(d, ignore_errors=True)
        except ImportError:
            pass
    print(json.dumps({"one_core": {"lookups": one_done, "seconds": one_t, "rate": one_done / one_t}, "bdb_file": bdb,
                      "word_parallel_pool": {"threads": a.threads, "rate": wp_pool},
                      "word_parallel_one_core": {"lookups": wp_done, "seconds": wp_t, "rate": wp_done / wp_t},
                      "pool": {"threads": a.threads, "seconds": a.seconds, "rates": rates, "rate_median": float(np.median(rates)), "rate_best": max(rates)},
                      "rows": a.rows, "cols": a.cols, "fill_seconds": fill_s, "host_threads": os.cpu_count()}))


if __name__ == "__main__":
    main()
