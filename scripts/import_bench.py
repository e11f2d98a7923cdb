"""migrate_index from an in-memory source store (rows as bytes objects, as a KV store's batch_get hands them over) into the hip-hbm
backend: what the importer's own side costs -- assembling row blocks in Python and the copy to the device -- with the two
overlapped (RowUploader) and one after the other.  One JSON line.   python scripts/import_bench.py [--gb 4] [--cols 500000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class MemorySource(object):
    """the part of the storage contract migrate_index reads (bigsi/storage/base.py:29-36, 58-59)"""

    def __init__(self, m, n, h, rows):
        self.ints = {"number_of_rows": m, "number_of_cols": n, "ksi:bloomfilter_size": m, "ksi:num_hashes": h}
        self.rows = rows

    def get_integer(self, key):
        return self.ints[key]

    def batch_get(self, keys):
        return [self.rows[int(k.split(b":")[0])] for k in keys]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gb", type=float, default=4.0)
    p.add_argument("--cols", type=int, default=500_000)
    p.add_argument("--bdb-gb", type=float, default=2.0, help="rows written into a BerkeleyDB hash file (libdb through dbm.ndbm) and imported from it")
    a = p.parse_args()
    from bigsi_amd.migrate import migrate_index
    from bigsi_amd.storage import get_storage
    rb = (a.cols + 7) // 8
    m = int(a.gb * 1e9 // rb)
    rng = np.random.default_rng(1)
    base = [rng.integers(0, 256, size=rb, dtype=np.uint8).tobytes() for _ in range(64)]
    rows = [base[i % 64] for i in range(m)]
    src = MemorySource(m, a.cols, 3, rows)
    out = {"what": "migrate_index of %d rows x %d columns (%.1f GB) from bytes objects in memory" % (m, a.cols, m * rb / 1e9)}
    for overlap in (False, True, False, True):
        dst = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": m, "h": 3, "storage-config": {"name": "import_bench", "max_cols": a.cols}})
        dst.delete_all()
        t0 = time.perf_counter()
        migrate_index(src, dst, overlap=overlap)
        dt = time.perf_counter() - t0
        got = np.asarray(dst.get_rows_packed(np.array([0, 1, m - 1], np.uint64)))
        assert got[0].tobytes()[:rb] == rows[0] and got[2].tobytes()[:rb] == rows[m - 1]
        out.setdefault("overlapped_GBps" if overlap else "sequential_GBps", []).append(round(m * rb / dt / 1e9, 2))
        dst.delete_all()
    # a BerkeleyDB hash store of the same rows, written by libdb itself (dbm.ndbm), imported below Python (bigsi_hip_load_rows_file on the file)
    # and through the Python page walk
    try:
        import dbm.ndbm as ndbm
        import tempfile
        from bigsi_amd import bdb
        if getattr(ndbm, "library", "") == "Berkeley DB":
            d = tempfile.mkdtemp(prefix="bigsi_import_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            mb = min(m, int(a.bdb_gb * 1e9 // rb))
            db = ndbm.open(os.path.join(d, "store"), "n")
            for key, v in (("number_of_rows:int", mb), ("number_of_cols:int", a.cols), ("ksi:bloomfilter_size:int", mb), ("ksi:num_hashes:int", 3)):
                db[key] = str(v)
            for i in range(mb):
                db["%d:bitarray" % i] = rows[i]
            db.close()
            fn = os.path.join(d, "store.db")
            out["bdb_file_gb"] = round(os.path.getsize(fn) / 1e9, 2)
            for native in (True, False):
                dst = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": mb, "h": 3, "storage-config": {"name": "import_bench_bdb", "max_cols": a.cols}})
                t0 = time.perf_counter()
                tm = {}
                bdb.import_index(fn, dst, native=native, timings=tm)
                dt = time.perf_counter() - t0
                if native:
                    out["bdb_native_phases_s"] = {k_: round(v_, 3) for k_, v_ in tm.items()}
                got = np.asarray(dst.get_rows_packed(np.array([0, 1, mb - 1], np.uint64)))
                assert got[0].tobytes()[:rb] == rows[0] and got[2].tobytes()[:rb] == rows[mb - 1]
                out["bdb_native_GBps" if native else "bdb_python_GBps"] = round(mb * rb / dt / 1e9, 2)
                dst.delete_all()
            import shutil
            shutil.rmtree(d, ignore_errors=True)
    except ImportError:
        pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
