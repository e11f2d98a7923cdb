"""Host-visible rate of the build: Bloom filters in HOST memory -> columns of the matrix (bigsi_hip_insert_columns: staging at a 128-byte
pitch + k_transpose_regs), against the device-resident transpose and the PCIe rate.    python scripts/build_bench.py [rows cols]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bigsi_amd import _lib                                    # noqa: E402
from bigsi_amd.storage import get_storage                     # noqa: E402


def main():
    m, ncols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4_000_000, 16384)
    nb = (m + 7) // 8
    rng = np.random.default_rng(1)
    blooms = rng.integers(0, 256, size=(ncols, nb), dtype=np.uint8)
    st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": m, "h": 3, "storage-config": {"name": "bb", "device": 0, "max_cols": ncols}})
    st.delete_all()
    for key, v in (("number_of_rows", m), ("number_of_cols", 0), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", 3)):
        st.set_integer(key, v)
    st.insert_columns(0, blooms[:512])          # warm
    out = {"m": m, "cols": ncols, "filter_GB": round(blooms.nbytes / 1e9, 3)}
    for rep in range(3):
        t = time.perf_counter()
        st.insert_columns(0, blooms)
        dt = time.perf_counter() - t
        out["rep%d_s" % rep] = round(dt, 4)
        out["rep%d_GBps_filters_in" % rep] = round(blooms.nbytes / dt / 1e9, 2)
    for r in (0, 1, 511, m // 2 + 3, m - 1):
        bits = (blooms[:, r >> 3] >> (7 - (r & 7))) & 1
        assert np.array_equal(st.get_rows_packed([r], (ncols + 7) // 8)[0], np.packbits(bits)), r
    st.delete_all()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
