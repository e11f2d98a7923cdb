"""ctypes binding of libbigsi_hip.so (include/bigsi_hip.h).

The HIP library IS the product path: there is no CPU fallback.  If the shared object is missing or no
GPU is visible, importing symbols still works (so `-m "not gpu"` tests can check the ABI) but any call
that needs a device raises BigsiHipError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BIGSI_HIP_LIB selects another build of the same ABI (kernel A/B experiments); default: the in-tree library
LIB_PATH = os.environ.get("BIGSI_HIP_LIB") or os.path.join(_HERE, "libbigsi_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_RANGE, ERR_CAPACITY, ERR_STATE = 0, -1, -2, -3, -4, -5, -6
RUN_FORCE_COUNTS = 1
RUN_SKIP_COMPACT = 2
RUN_K1_GLOBAL = 4
RUN_SPARSE_COUNTS = 8
RUN_NO_SORT = 16
RUN_EARLY_EXIT = 32
RUN_WEAK_FINGERPRINT = 64
RUN_ONE_STREAM = 256
BLOOM_RAW = 1
SCORE_ORDERED = 1


class BigsiHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbigsi_hip error %d: %s" % (code, msg))
        self.code = code


class Info(C.Structure):
    _fields_ = [("num_rows", C.c_uint64), ("num_cols", C.c_uint64), ("col_capacity", C.c_uint64),
                ("row_bytes", C.c_uint64), ("row_stride_bytes", C.c_uint64), ("index_bytes", C.c_uint64),
                ("num_hashes", C.c_uint32), ("device", C.c_int32)]


class BatchInfo(C.Structure):
    _fields_ = [("n_seqs", C.c_uint32), ("k", C.c_uint32), ("exact", C.c_uint32), ("count_bytes", C.c_uint32),
                ("total_kmers", C.c_uint64), ("total_unique", C.c_uint64), ("total_hits", C.c_uint64),
                ("bitmap_stride_bytes", C.c_uint64), ("counts_stride", C.c_uint64),
                ("d_bitmaps", C.c_void_p), ("d_counts", C.c_void_p), ("d_num_unique", C.c_void_p),
                ("one_launch", C.c_uint32), ("reserved", C.c_uint32)]


class GroupInfo(C.Structure):
    _fields_ = [("num_rows", C.c_uint64), ("num_cols", C.c_uint64), ("col_capacity", C.c_uint64), ("shard_cols", C.c_uint64),
                ("row_bytes", C.c_uint64), ("index_bytes", C.c_uint64), ("num_hashes", C.c_uint32), ("n_shards", C.c_uint32),
                ("rccl", C.c_uint32)]


class IoStats(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("seconds", C.c_double), ("file_seconds", C.c_double), ("threads", C.c_uint32), ("direct", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("and_launches", C.c_uint64), ("and_ms", C.c_double),
                ("kmerize_launches", C.c_uint64), ("kmerize_ms", C.c_double),
                ("compact_launches", C.c_uint64), ("compact_ms", C.c_double),
                ("presence_launches", C.c_uint64), ("presence_ms", C.c_double), ("presence_bytes", C.c_uint64),
                ("transpose_launches", C.c_uint64), ("transpose_ms", C.c_double),
                ("and_launches_total", C.c_uint64), ("read_launches_repeated", C.c_uint64), ("index_contiguous", C.c_uint64),
                ("exchange_launches", C.c_uint64), ("exchange_ms", C.c_double)]


_P = C.c_void_p
_u64, _u32, _i32, _dbl = C.c_uint64, C.c_uint32, C.c_int, C.c_double

# every symbol include/bigsi_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bigsi_hip_last_error": (C.c_char_p, []),
    "bigsi_hip_device_count": (_i32, [C.POINTER(C.c_int)]),
    "bigsi_hip_open": (_i32, [_u64, _u64, _u64, _u32, _i32, C.POINTER(_P)]),
    "bigsi_hip_close": (_i32, [_P]),
    "bigsi_hip_export_ipc": (_i32, [_P, _P]),
    "bigsi_hip_open_ipc": (_i32, [_P, _u64, _u64, _u64, _u32, _i32, C.POINTER(_P)]),
    "bigsi_hip_open_view": (_i32, [_P, C.POINTER(_P)]),
    "bigsi_hip_get_info": (_i32, [_P, C.POINTER(Info)]),
    "bigsi_hip_set_num_cols": (_i32, [_P, _u64]),
    "bigsi_hip_set_num_hashes": (_i32, [_P, _u32]),
    "bigsi_hip_reserve_cols": (_i32, [_P, _u64]),
    "bigsi_hip_set_stream": (_i32, [_P, _P]),
    "bigsi_hip_synchronize": (_i32, [_P]),
    "bigsi_hip_set_rows": (_i32, [_P, _P, _u64, _P, _u64]),
    "bigsi_hip_get_rows": (_i32, [_P, _P, _u64, _P, _u64]),
    "bigsi_hip_clear": (_i32, [_P]),
    "bigsi_hip_load_rows_file": (_i32, [_P, C.c_char_p, _u64, _u64, _u64, _u64, _u32, C.POINTER(IoStats)]),
    "bigsi_hip_save_rows_file": (_i32, [_P, C.c_char_p, _u64, _u64, _u64, _u64, _u32, C.POINTER(IoStats)]),
    "bigsi_hip_bdb_small_records": (_i32, [C.c_char_p, _P, _u64, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), _u32]),
    "bigsi_hip_insert_column": (_i32, [_P, _u64, _P]),
    "bigsi_hip_get_column": (_i32, [_P, _u64, _P]),
    "bigsi_hip_insert_columns": (_i32, [_P, _u64, _u64, _P, _u64]),
    "bigsi_hip_insert_columns_device": (_i32, [_P, _u64, _u64, _P, _u64]),
    "bigsi_hip_append_index": (_i32, [_P, _P]),
    "bigsi_hip_insert_kmers": (_i32, [_P, _u64, C.c_char_p, _P, _u32, _u32]),
    "bigsi_hip_fill_synthetic": (_i32, [_P, _u64, _u64, _u32]),
    "bigsi_hip_bloom": (_i32, [_i32, C.c_char_p, _u64, _u32, _u64, _u32, _u32, _P]),
    "bigsi_hip_lookup": (_i32, [_P, C.c_char_p, _u32, _u64, _P]),
    "bigsi_hip_batch_create": (_i32, [_P, C.c_char_p, _P, _u32, _u32, C.POINTER(_P)]),
    "bigsi_hip_batch_create_elements": (_i32, [_P, C.c_char_p, _P, _P, _P, _P, _u32, C.POINTER(_P)]),
    "bigsi_hip_lookup_raw": (_i32, [_P, C.c_char_p, _P, _u64, _P]),
    "bigsi_hip_batch_destroy": (_i32, [_P]),
    "bigsi_hip_batch_reload": (_i32, [_P, C.c_char_p, _P, _u32, _u32]),
    "bigsi_hip_batch_run": (_i32, [_P, _dbl, _u32]),
    "bigsi_hip_batch_get_info": (_i32, [_P, C.POINTER(BatchInfo)]),
    "bigsi_hip_batch_set_outputs": (_i32, [_P, _P, _P]),
    "bigsi_hip_batch_fetch_unique": (_i32, [_P, _P, _P, _P]),
    "bigsi_hip_batch_fetch_hits": (_i32, [_P, _P, _P, _P, _u64]),
    "bigsi_hip_batch_fetch_counts": (_i32, [_P, _u32, _P]),
    "bigsi_hip_batch_fetch_bitmap": (_i32, [_P, _u32, _P]),
    "bigsi_hip_batch_fetch_rows": (_i32, [_P, _u32, _P, _u64]),
    "bigsi_hip_batch_lookup": (_i32, [_P, _u32, _P, _P, _u64]),
    "bigsi_hip_batch_presence": (_i32, [_P, _u32, _P, _u32, _P]),
    "bigsi_hip_batch_presence_hits": (_i32, [_P, _P, _P, _P, _u64, _P]),
    "bigsi_hip_group_batch_presence_hits": (_i32, [_P, _P, _P, _P, _u64, _P]),
    "bigsi_hip_batch_score_hits": (_i32, [_P, _P, _P, _P, _P, _u64, _P, _P]),
    "bigsi_hip_group_batch_score_hits": (_i32, [_P, _P, _P, _P, _P, _u64, _P, _P]),
    "bigsi_hip_batch_score_hits_begin": (_i32, [_P, _P, _P, _P, _u32, _P]),
    "bigsi_hip_batch_score_hits_end": (_i32, [_P, _P, _u64, _P]),
    "bigsi_hip_score_presence": (_i32, [_i32, _P, _P, _P, _P, _P, _u64, _P]),
    "bigsi_hip_batch_set_gather_stream": (_i32, [_P, _P]),
    "bigsi_hip_batch_compact_gathered": (_i32, [_P, _P, _u32, _u64]),
    "bigsi_hip_batch_compact_gathered_masks": (_i32, [_P, _P, _u32, _u64, _u32]),
    "bigsi_hip_batch_set_gathered_hit_outputs": (_i32, [_P, _P, _P, _u64]),
    "bigsi_hip_batch_fetch_gathered_hits": (_i32, [_P, _P, _P, _P, _u64]),
    "bigsi_hip_batch_set_result_cols": (_i32, [_P, _u64]),
    "bigsi_hip_search_batch": (_i32, [_P, C.c_char_p, _P, _u32, _u32, _dbl, _u32, _P, _P, _P, _P, _P, _P, _u64]),
    "bigsi_hip_search_stream": (_i32, [_P, C.c_char_p, _P, _u64, _u32, _dbl, _u32, _P, _P, _P, _P, _P, _P, _u64]),
    "bigsi_hip_search_stream_scored": (_i32, [_P, C.c_char_p, _P, _u64, _u32, _dbl, _u32, _P, _P, _P, _P, _P, _P, _u64, _P, _u64, _P, _P, _P]),
    "bigsi_hip_comm_unique_id": (_i32, [_P]),
    "bigsi_hip_comm_init_rank": (_i32, [_i32, _P, _i32, _i32, C.POINTER(_P)]),
    "bigsi_hip_comm_destroy": (_i32, [_P]),
    "bigsi_hip_comm_info": (_i32, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bigsi_hip_batch_set_comm": (_i32, [_P, _P, _u64]),
    "bigsi_hip_batch_run_sharded": (_i32, [_P, _dbl, _u32]),
    "bigsi_hip_group_open": (_i32, [_u64, _u64, _u64, _u32, C.POINTER(C.c_int), _i32, C.POINTER(_P)]),
    "bigsi_hip_group_close": (_i32, [_P]),
    "bigsi_hip_group_export_ipc": (_i32, [_P, _P]),
    "bigsi_hip_group_open_ipc": (_i32, [_P, _u64, _u64, _u64, _u32, C.POINTER(C.c_int), _i32, C.POINTER(_P)]),
    "bigsi_hip_group_load_rows_file": (_i32, [_P, C.c_char_p, _u64, _u64, _u64, _u64, _u32, C.POINTER(IoStats)]),
    "bigsi_hip_group_save_rows_file": (_i32, [_P, C.c_char_p, _u64, _u64, _u64, _u64, _u32, C.POINTER(IoStats)]),
    "bigsi_hip_group_get_info": (_i32, [_P, C.POINTER(GroupInfo)]),
    "bigsi_hip_group_shard": (_i32, [_P, _u32, C.POINTER(_P)]),
    "bigsi_hip_group_set_num_cols": (_i32, [_P, _u64]),
    "bigsi_hip_group_set_num_hashes": (_i32, [_P, _u32]),
    "bigsi_hip_group_synchronize": (_i32, [_P]),
    "bigsi_hip_group_clear": (_i32, [_P]),
    "bigsi_hip_group_set_rows": (_i32, [_P, _P, _u64, _P, _u64]),
    "bigsi_hip_group_get_rows": (_i32, [_P, _P, _u64, _P, _u64]),
    "bigsi_hip_group_insert_columns": (_i32, [_P, _u64, _u64, _P, _u64]),
    "bigsi_hip_group_get_column": (_i32, [_P, _u64, _P]),
    "bigsi_hip_group_insert_kmers": (_i32, [_P, _u64, C.c_char_p, _P, _u32, _u32]),
    "bigsi_hip_group_fill_synthetic": (_i32, [_P, _u64, _u32]),
    "bigsi_hip_group_lookup": (_i32, [_P, C.c_char_p, _u32, _u64, _P]),
    "bigsi_hip_group_batch_create": (_i32, [_P, C.c_char_p, _P, _u32, _u32, C.POINTER(_P)]),
    "bigsi_hip_group_batch_create_elements": (_i32, [_P, C.c_char_p, _P, _P, _P, _P, _u32, C.POINTER(_P)]),
    "bigsi_hip_group_lookup_raw": (_i32, [_P, C.c_char_p, _P, _u64, _P]),
    "bigsi_hip_group_batch_reload": (_i32, [_P, C.c_char_p, _P, _u32, _u32]),
    "bigsi_hip_group_batch_destroy": (_i32, [_P]),
    "bigsi_hip_group_batch_run": (_i32, [_P, _dbl, _u32]),
    "bigsi_hip_group_batch_fetch_unique": (_i32, [_P, _P, _P, _P]),
    "bigsi_hip_group_batch_fetch_hits": (_i32, [_P, _P, _P, _P, _u64]),
    "bigsi_hip_group_batch_presence": (_i32, [_P, _u32, _P, _u32, _P]),
    "bigsi_hip_group_search_batch": (_i32, [_P, C.c_char_p, _P, _u32, _u32, _dbl, _u32, _P, _P, _P, _P, _P, _P, _u64]),
    "bigsi_hip_set_profiling": (_i32, [_P, _i32]),
    "bigsi_hip_stats": (_i32, [_P, C.POINTER(Stats), _i32]),
    "bigsi_hip_probe_rows": (_i32, [_P, _u32, _u32, _u32, _u32, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bigsi_hip_fasta_pack": (_i32, [_P, _u64, _P, _P, _u64, C.POINTER(_u64)]),
    "bigsi_hip_format_results": (_i32, [_i32, _P, _P, _u64, C.c_char_p, C.c_char_p, _i32, _P, _P, _P, _P, _P, _P, _P, _u64, _u32,
                                         C.POINTER(_P), C.POINTER(_u64)]),
    "bigsi_hip_format_results_scored": (_i32, [_i32, _P, _P, _u64, C.c_char_p, C.c_char_p, _i32, _P, _P, _P, _P, _P, _P, _P, _u64, _P, _u32,
                                                C.POINTER(_P), C.POINTER(_u64)]),
    "bigsi_hip_free_text": (None, [_P]),
}

_lib = None


def lib():
    """The loaded library; raises if libbigsi_hip.so has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BigsiHipError(ERR_STATE, "%s is missing: build it with bigsi_amd/csrc/build.sh "
                                           "(or python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
        # One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64; if this library pulls
        # in /opt/rocm's copy first, a later `import torch` finds "No HIP GPUs".  Loading torch first (when it is
        # installed) makes both share torch's copy, which is what the multi-GPU path (torch.distributed/RCCL) needs
        # anyway.  BIGSI_HIP_NO_TORCH=1 skips this for torch-free deployments.
        if not os.environ.get("BIGSI_HIP_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise BigsiHipError(rc, (lib().bigsi_hip_last_error() or b"").decode("utf-8", "replace"))
    return rc


def device_count():
    n = C.c_int(0)
    check(lib().bigsi_hip_device_count(C.byref(n)))
    return n.value


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def fasta_pack(data):
    """bigsi_hip_fasta_pack over the bytes of a FASTA file: (uint8 blob, uint64 offsets[n+1]) of its sequences, or None when the
    text is not plain ASCII (the caller's Python route then reads the file)."""
    n = C.c_uint64(0)
    rc = lib().bigsi_hip_fasta_pack(data, len(data), None, None, 0, C.byref(n))          # the sizing call
    if rc == ERR_INVALID:
        return None
    check(rc)
    blob, off = np.empty(max(len(data), 1), np.uint8), np.empty(n.value + 1, np.uint64)
    check(lib().bigsi_hip_fasta_pack(data, len(data), ptr(blob), ptr(off), n.value, C.byref(n)))
    return blob[:int(off[n.value])], off


class ScoredText(C.Structure):
    """bigsi_hip_scored_text (include/bigsi_hip.h): what bigsi_hip_format_results_scored needs per hit beside the hit lists"""
    _fields_ = [("scores", C.c_void_p), ("bits", C.c_void_p), ("bit_offsets", C.c_void_p), ("evalue", C.c_void_p), ("pvalue", C.c_void_p),
                ("log_evalue", C.c_void_p), ("log_pvalue", C.c_void_p), ("k", C.c_uint32), ("reserved", C.c_uint32)]


def format_results(fmt, blob, soff, threshold, citation_json, nu, off, col, cnt, names, name_off, deleted, threads=0, scored=None):
    """bigsi_hip_format_results(_scored) -> str (the reference's bulk_search text); BigsiHipError(ERR_STATE) where the reference raises
    instead of answering.  scored = (records, bits, bit_offsets, evalue, pvalue, log_evalue, log_pvalue, k): per-hit arrays of a
    score=True search (K6's records and presence bits, the closed-form columns of scoring.score_columns)."""
    import json
    n = len(soff) - 1
    sc = None
    if scored is not None:
        keep = [np.ascontiguousarray(a) for a in scored[:7]]          # (alive until the calls below have returned)
        sc = ScoredText(*([ptr(a) for a in keep] + [int(scored[7]), 0]))
    args = (int(fmt), ptr(blob) if not isinstance(blob, bytes) else blob, ptr(soff), n, json.dumps(threshold).encode(), citation_json.encode(),
            1 if threshold == 1.0 else 0, ptr(nu), ptr(off), ptr(col), ptr(cnt), names, ptr(name_off), ptr(deleted), len(name_off) - 1,
            C.addressof(sc) if sc is not None else None, int(threads))
    try:
        from . import _results          # (the CPython extension: it can hand out a str whose body the library fills in place)
    except ImportError:
        _results = None
    if _results is not None and scored is None:          # (scored text is formatted once, into blocks: its sizing call would format it all)
        # the sizing call (a zero-byte buffer of the caller's), then the text written straight into a new str: a bulk search of a
        # million reads is 200 MB of JSON, whose malloc + decode + free took three times as long as formatting it
        probe = C.create_string_buffer(1)
        text, size = C.c_void_p(C.addressof(probe)), C.c_uint64(0)
        rc = lib().bigsi_hip_format_results_scored(*args, C.byref(text), C.byref(size))
        if rc != ERR_CAPACITY:
            check(rc)
            return ""                    # (an empty text fits a zero-byte buffer)
        s, address = _results.ascii_str(size.value)
        text, cap = C.c_void_p(address), C.c_uint64(size.value)
        check(lib().bigsi_hip_format_results_scored(*args, C.byref(text), C.byref(cap)))
        return s
    text, size = C.c_void_p(), C.c_uint64(0)
    check(lib().bigsi_hip_format_results_scored(*args, C.byref(text), C.byref(size)))
    try:
        return str(memoryview((C.c_char * size.value).from_address(text.value)), "ascii") if size.value else ""
    finally:
        lib().bigsi_hip_free_text(text)


def pack_seqs(seqs):
    """list of str/bytes -> (blob bytes, uint64 offsets[n+1]).  Sequences must be ASCII: the reference hashes the
    UTF-8 bytes of k *characters* (mmh3.hash(str)); for ASCII that is k bytes, which is what the kernels window.  Callers
    route non-ASCII text through the element batches instead (BIGSI._elements_of, bigsi_hip_batch_create_elements)."""
    if not isinstance(seqs, (list, tuple)):
        seqs = list(seqs)
    if seqs and all(type(s) is str for s in seqs):
        # fast path (bulk reads): one join + one encode; for ASCII text the character lengths are the byte lengths
        text = "".join(seqs)
        try:
            blob = text.encode("ascii")
        except UnicodeEncodeError:
            raise ValueError("query sequences must be ASCII for the hip-hbm backend")
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(np.fromiter(map(len, seqs), dtype=np.uint64, count=len(seqs)), out=off[1:])
        return blob, off
    enc = []
    for s in seqs:
        if isinstance(s, str):
            try:
                s = s.encode("ascii")
            except UnicodeEncodeError:
                raise ValueError("query sequences must be ASCII for the hip-hbm backend")
        enc.append(bytes(s))
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        off[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    return b"".join(enc), off
