"""N Bloom filters of m bits -> m rows of N bits (bigsi/matrix/transpose.py:14-50), as packed bytes.

The reference's helper behind BitMatrix.create.  Here it is the device transpose (bigsi_hip_insert_columns) run on a
scratch index: the filters go up as columns, the rows come back in the storage row format.  Index construction itself
(KmerSignatureIndex.create) transposes straight into the resident matrix and never materialises the rows on the host;
this function exists for callers that want the rows, as the reference's did."""
import numpy as np

from .. import _lib
from .._lib import check
from ..bitrow import BitRow, row_bytes_of


def transpose_packed(bloomfilters, num_rows=None, device=0):
    """uint8[m, ceil(N/8)] in the storage row format."""
    bloomfilters = list(bloomfilters)
    n = len(bloomfilters)
    pairs = [row_bytes_of(bf) for bf in bloomfilters]
    m = num_rows if num_rows is not None else (pairs[0][1] if pairs else 0)
    if n == 0 or m == 0:
        return np.zeros((m, 0), np.uint8)
    nb = (m + 7) // 8
    arr = np.zeros((n, nb), dtype=np.uint8)
    for i, (data, nbits) in enumerate(pairs):
        a = np.frombuffer(data, dtype=np.uint8)[:nb]
        arr[i, : a.size] = a
        if nbits < m and nbits % 8:                       # a filter shorter than m: bits beyond its length are zero
            arr[i, nbits // 8] &= (0xFF << (8 - nbits % 8)) & 0xFF
    L = _lib.lib()
    ix = _lib.C.c_void_p()
    check(L.bigsi_hip_open(m, 0, n, 1, int(device), _lib.C.byref(ix)))
    try:
        check(L.bigsi_hip_insert_columns(ix, 0, n, _lib.ptr(arr), nb))
        rb = (n + 7) // 8
        out = np.zeros((m, rb), dtype=np.uint8)
        ids = np.arange(m, dtype=np.uint64)
        check(L.bigsi_hip_get_rows(ix, _lib.ptr(ids), m, _lib.ptr(out), rb))
    finally:
        check(L.bigsi_hip_close(ix))
    return out


def transpose(bitarrays, lowmem=False):
    """Generator of BitRow rows, the reference's return shape.  `lowmem` (host-memory chunking in the reference) has nothing
    to bound here: the only host copy is the result itself."""
    bitarrays = list(bitarrays)
    packed = transpose_packed(bitarrays)
    n = len(bitarrays)
    for r in packed:
        yield BitRow.frombytes(r.tobytes(), n)
