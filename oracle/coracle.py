"""ctypes binding of oracle/bigsi_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "bigsi_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None
_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_mmh3_hash.restype = C.c_int32
        L.orc_mmh3_hash.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32]
        L.orc_row_of.restype = C.c_uint64
        L.orc_row_of.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_reverse_comp.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        L.orc_canonical.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        L.orc_kmer_rows.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        L.orc_seq_rows.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        L.orc_unique_kmers.restype = C.c_uint32
        L.orc_unique_kmers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_and_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_lookup.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_unpack_and_sum.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_and_all.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_query.restype = C.c_uint32
        L.orc_query.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint64, C.c_uint64,
                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_synth_word.restype = C.c_uint64
        L.orc_synth_word.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_valid_mask.restype = C.c_uint64
        L.orc_valid_mask.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_synth_row.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_synth_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def mmh3_hash(s, seed=0):
    b = _b(s)
    return int(lib().orc_mmh3_hash(b, len(b), seed & 0xFFFFFFFF))


def row_of(canon, seed, m):
    b = _b(canon)
    return int(lib().orc_row_of(b, len(b), seed, m))


def reverse_comp(s):
    b = _b(s)
    out = C.create_string_buffer(len(b))
    lib().orc_reverse_comp(b, len(b), out)
    return out.raw.decode("utf-8")


def canonical(s):
    b = _b(s)
    out = C.create_string_buffer(len(b))
    lib().orc_canonical(b, len(b), out)
    return out.raw.decode("utf-8")


def kmer_rows(kmer, h, m):
    b = _b(kmer)
    out = np.zeros(h, dtype=np.uint64)
    lib().orc_kmer_rows(b, len(b), h, m, out.ctypes.data)
    return [int(x) for x in out]


def seq_rows(seq, k, h, m):
    """uint64[n, h]: the h rows of every k-mer position of seq (one C call)."""
    b = _b(seq)
    n = max(len(b) - k + 1, 0)
    out = np.zeros((n, h), dtype=np.uint64)
    if n:
        lib().orc_seq_rows(b, len(b), k, h, m, out.ctypes.data)
    return out


def unique_kmers(seq, k):
    b = _b(seq)
    n = max(len(b) - k + 1, 0)
    first = np.zeros(max(n, 1), dtype=np.uint32)
    p2u = np.zeros(max(n, 1), dtype=np.uint32)
    u = lib().orc_unique_kmers(b, len(b), k, first.ctypes.data, p2u.ctypes.data)
    return first[:u].copy(), p2u[:n].copy()


def lookup(index, h, kmers, k):
    """index: uint8[m, rb]; kmers: list of str of length k.  Returns uint8[u, rb]."""
    index = np.ascontiguousarray(index, dtype=np.uint8)
    m, rb = index.shape
    blob = b"".join(_b(x) for x in kmers)
    out = np.zeros((len(kmers), rb), dtype=np.uint8)
    lib().orc_lookup(index.ctypes.data, m, rb, h, blob, len(kmers), k, out.ctypes.data)
    return out


def unpack_and_sum(rows):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    u, rb = rows.shape
    counts = np.zeros(8 * rb, dtype=np.int32)
    lib().orc_unpack_and_sum(rows.ctypes.data, u, rb, counts.ctypes.data)
    return counts


def and_all(rows):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    u, rb = rows.shape
    out = np.zeros(rb, dtype=np.uint8)
    lib().orc_and_all(rows.ctypes.data, u, rb, out.ctypes.data)
    return out


def query(index, h, seq, k, want_counts=True, want_and=True):
    """Whole reference-shaped query on an in-RAM index.  Returns (u, counts int32[8*rb] | None, and_all uint8[rb] | None)."""
    index = np.ascontiguousarray(index, dtype=np.uint8)
    m, rb = index.shape
    b = _b(seq)
    n = max(len(b) - k + 1, 1)
    scratch = np.empty((n, rb), dtype=np.uint8)
    counts = np.zeros(8 * rb, dtype=np.int32) if want_counts else None
    aa = np.zeros(rb, dtype=np.uint8) if want_and else None
    u = lib().orc_query(index.ctypes.data, m, rb, h, b, len(b), k, scratch.ctypes.data,
                        counts.ctypes.data if want_counts else None, aa.ctypes.data if want_and else None)
    return int(u), counts, aa


def synth_word(seed, shard, row, word, and_draws):
    return int(lib().orc_synth_word(seed, shard, row, word, and_draws))


def synth_row(seed, shard, row, n_cols, and_draws):
    out = np.zeros((n_cols + 7) // 8, dtype=np.uint8)
    lib().orc_synth_row(seed, shard, row, n_cols, and_draws, out.ctypes.data)
    return out


def synth_fill(seed, shard, row0, n_rows, n_cols, and_draws):
    out = np.zeros((n_rows, (n_cols + 7) // 8), dtype=np.uint8)
    lib().orc_synth_fill(seed, shard, row0, n_rows, n_cols, and_draws, out.ctypes.data)
    return out
