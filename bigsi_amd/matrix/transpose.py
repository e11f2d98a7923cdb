"""N Bloom filters of m bits -> m rows of N bits (bigsi/matrix/transpose.py:14-50), as packed bytes.

Index construction only; host numpy for now: unpack to an m x N bit array, re-pack by row."""
import numpy as np

from ..bitrow import BitRow, row_bytes_of


def transpose_packed(bloomfilters, num_rows=None):
    """uint8[m, ceil(N/8)] in the storage row format."""
    cols = []
    for bf in bloomfilters:
        data, nbits = row_bytes_of(bf)
        bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))[:nbits]
        cols.append(bits)
    m = num_rows if num_rows is not None else (len(cols[0]) if cols else 0)
    mat = np.zeros((m, len(cols)), dtype=np.uint8)
    for j, c in enumerate(cols):
        mat[: min(m, len(c)), j] = c[:m]
    return np.packbits(mat, axis=1) if len(cols) else np.zeros((m, 0), np.uint8)


def transpose(bitarrays, lowmem=False):
    """Generator of BitRow rows, the reference's return shape (lowmem is accepted and ignored)."""
    bitarrays = list(bitarrays)
    packed = transpose_packed(bitarrays)
    n = len(bitarrays)
    for r in packed:
        yield BitRow.frombytes(r.tobytes(), n)
