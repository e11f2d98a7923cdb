/*
 * bigsi_hip.h -- C ABI of libbigsi_hip.so: the BIGSI query hot path on an HBM-resident
 * bit-sliced signature matrix (AMD MI355X / gfx950).
 *
 * This is the drop-in boundary.  The reference (Phelimb/BIGSI, pure Python) reaches its storage
 * through the `bigsi.storage` plugin contract (bigsi/storage/base.py:9-151, registry
 * bigsi/storage/__init__.py:3-19); a `hip-hbm` backend binds the entry points below with ctypes
 * (INTEGRATION.md shows the stub).  Every entry point cites the reference interface it replaces;
 * paths are relative to the reference root.
 *
 * Conventions
 *   - every function returns 0 (BIGSI_OK) or a negative BIGSI_ERR_* code; the message of the last
 *     failure on the calling thread is bigsi_hip_last_error().
 *   - the caller owns all host buffers; no call retains a host pointer after it returns.
 *   - one index / batch handle may be used from one host thread at a time -- ENFORCED: a call on a handle (or on a batch of
 *     it) that another thread is inside fails with BIGSI_ERR_STATE, it does not race (bigsi_hip_batch_destroy alone WAITS for
 *     that thread: finalisers run it from any thread, and a refused destroy would be a leak).  Distinct handles are independent, and
 *     bigsi_hip_open_view gives every thread of a serving host its own handle onto the one resident matrix.  HIP contexts do not survive fork(): open after forking (bulk_search,
 *     bigsi/__main__.py:273-287, forks one worker per chunk).
 *   - LAYERS.  A binder needs only what its host does:
 *       CORE       index lifecycle + storage contract + bigsi_hip_lookup + bigsi_hip_search_batch (the whole of
 *                  BIGSI.search for a batch in ONE call) + bigsi_hip_batch_presence_hits: enough for any host.
 *       BATCHES    bigsi_hip_batch_*: staged workspaces and asynchronous runs for serving loops.
 *       MULTI-GPU  bigsi_hip_comm_* / batch_set_comm / batch_run_sharded (one process per GPU): the RCCL exchange is issued by
 *                  the library.  One process driving N GPUs: bigsi_hip_group_* in include/bigsi_hip_group.h.
 *       SHARING    export_ipc / open_ipc (another process) and open_view (another thread): read-only handles onto ONE resident matrix.
 *       MEASUREMENT  fill_synthetic, set_profiling, stats (calibration probe and device-resident filters: bigsi_hip_testing.h).
 *     Not advertised here (exported all the same, declared in include/bigsi_hip_testing.h): hooks for hosts that bring their own
 *     collective (torch.distributed over gloo on a one-GPU test box) and the BIGSI_RUN_* flags that force an A/B route.
 *     The library reads no environment variables (tuning knobs exist only in builds made with -DBIGSI_HIP_TUNING).
 *   - ROW FORMAT: a row is the reference's `bitarray.tobytes()` (bigsi/storage/base.py:85-99):
 *     ceil(num_cols/8) bytes, column c at byte c/8 under mask 0x80 >> (c%8), zero pad bits.
 *     The device stores exactly these bytes, zero-extended to a 128-byte-multiple row stride.
 */
#ifndef BIGSI_HIP_H
#define BIGSI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIGSI_OK 0
#define BIGSI_ERR_INVALID (-1)  /* bad argument                                   */
#define BIGSI_ERR_HIP (-2)      /* HIP runtime failure (message carries hipGetErrorString) */
#define BIGSI_ERR_NOMEM (-3)    /* host or device allocation failed                */
#define BIGSI_ERR_RANGE (-4)    /* row / column / sequence index out of range      */
#define BIGSI_ERR_CAPACITY (-5) /* caller's output buffer is too small             */
#define BIGSI_ERR_STATE (-6)    /* call made in the wrong order (e.g. fetch before run) */

typedef struct bigsi_hip_index bigsi_hip_index; /* one device-resident m x N bit matrix (one column shard) */
typedef struct bigsi_hip_batch bigsi_hip_batch; /* one batch of query sequences staged on the device       */

const char *bigsi_hip_last_error(void);
int bigsi_hip_device_count(int *out);

/* ================================================================== CORE: index lifecycle
 * Replaces opening a KV store and reading its four integers: BerkeleyDBStorage.__init__
 * (bigsi/storage/berkeleydb.py:6-19), BitMatrix.__init__ (bigsi/matrix/bitmatrix.py:14-17),
 * KmerSignatureIndex.__init__ (bigsi/graph/index.py:21-25).  Rows start out all-zero. */
int bigsi_hip_open(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                   int device, bigsi_hip_index **out);
int bigsi_hip_close(bigsi_hip_index *ix); /* BaseStorage.close, bigsi/storage/base.py:149-151 */
/* Destroy every batch created on an index before closing it: batches hold the index handle. */

/* ------------------------------------------------------------------ one resident matrix, several handles
 * The reference's store is a file any process opens (per request, per pool worker: bigsi/__main__.py:75-80, 204-205,
 * bigsi/storage/berkeleydb.py:12-19).  A resident index is one process's HBM allocation; these give OTHER handles onto it,
 * without a second copy and without re-ingesting it:
 *   export_ipc  the owner's matrix as BIGSI_IPC_HANDLE_BYTES opaque bytes (hipIpc), to be carried by any channel (a file ...).
 *   open_ipc    in ANOTHER process: map that allocation; num_rows / num_cols / col_capacity / num_hashes are the owner's
 *               (bigsi_hip_get_info there).  Milliseconds, whatever the index size.
 *   open_view   in the OWNER's process: a second handle (own streams, own workspaces) for another host thread.
 * Attached handles and views are READ-ONLY (calls that write the matrix fail with BIGSI_ERR_STATE) and are closed with
 * bigsi_hip_close.  The owner must outlive them; while views exist it refuses bigsi_hip_close and bigsi_hip_reserve_cols
 * (attached processes it cannot see: do not re-stride or close an exported index while others use it).  Writes the owner makes
 * (set_rows, insert) are seen by the other handles' later calls once the owner's call has returned; the COLUMN COUNT of an
 * attached handle or view is the one it was opened with (bigsi_hip_set_num_cols on that handle, or a fresh attach, to see
 * samples the owner has appended since). */
#define BIGSI_IPC_HANDLE_BYTES 64
int bigsi_hip_export_ipc(bigsi_hip_index *ix, uint8_t *handle /* BIGSI_IPC_HANDLE_BYTES */);
int bigsi_hip_open_ipc(const uint8_t *handle, uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                       int device, bigsi_hip_index **out);
int bigsi_hip_open_view(bigsi_hip_index *owner, bigsi_hip_index **out);

typedef struct {
    uint64_t num_rows;         /* m  = ksi:bloomfilter_size = number_of_rows */
    uint64_t num_cols;         /* N  = number_of_cols                        */
    uint64_t col_capacity;     /* columns the current row stride can hold    */
    uint64_t row_bytes;        /* ceil(num_cols / 8): bytes of one row in the reference format */
    uint64_t row_stride_bytes; /* device row pitch (multiple of 128)         */
    uint64_t index_bytes;      /* device bytes held by the matrix            */
    uint32_t num_hashes;       /* h  = ksi:num_hashes                        */
    int32_t device;
} bigsi_hip_info;
int bigsi_hip_get_info(const bigsi_hip_index *ix, bigsi_hip_info *out);

/* BitMatrix.set_num_cols (bigsi/matrix/bitmatrix.py:45-47).  Fails with BIGSI_ERR_CAPACITY beyond col_capacity. */
int bigsi_hip_set_num_cols(bigsi_hip_index *ix, uint64_t num_cols);
/* storage.set_integer("ksi:num_hashes", h) (bigsi/graph/index.py:11,33). */
int bigsi_hip_set_num_hashes(bigsi_hip_index *ix, uint32_t num_hashes);
/* Grow the row stride on the device so that insert / merge can append columns
 * (bigsi/matrix/bitmatrix.py:67-75, bigsi/graph/index.py:54-60). */
int bigsi_hip_reserve_cols(bigsi_hip_index *ix, uint64_t col_capacity);
/* STREAMS.  On its private stream(s) the library orders everything itself: bigsi_hip_batch_run is asynchronous; the fetch /
 * presence calls of a batch wait for THAT batch's kernels only; batches of reads (the one-launch kernel: k = 31, < 64
 * k-mers per query, hit lists only) are issued round-robin on three internal streams so that consecutive batches overlap;
 * every call that changes the index, bigsi_hip_stats and bigsi_hip_synchronize wait for all of them; the kernels of scored
 * searches (bigsi_hip_batch_score_hits*, presence_hits) run on a fourth, high-priority stream beside whatever the others are
 * doing, and a batch that is run again is ordered behind its own pending score request on the device.  (A caller-owned stream: bigsi_hip_testing.h.) */
int bigsi_hip_synchronize(bigsi_hip_index *ix);

/* ================================================================== CORE: storage contract
 * BaseStorage.set_bitarrays / get_bitarrays / batch_set / batch_get over "<row>:bitarray" keys
 * (bigsi/storage/base.py:43-59, 85-109): n rows of row_bytes bytes each, packed, in row_ids order.
 * row_bytes may be anything up to the stride; bytes beyond it are zeroed (set) / not returned (get). */
int bigsi_hip_set_rows(bigsi_hip_index *ix, const uint64_t *row_ids, uint64_t n, const uint8_t *bytes, uint64_t row_bytes);
int bigsi_hip_get_rows(bigsi_hip_index *ix, const uint64_t *row_ids, uint64_t n, uint8_t *out, uint64_t row_bytes);
int bigsi_hip_clear(bigsi_hip_index *ix); /* delete_all (bigsi/storage/base.py:132-133): all rows zero */

/* Bulk ingest: n_rows consecutive rows [row0, row0 + n_rows), row_bytes bytes each, packed back to back in a file from file_offset on,
 * straight between the file and HBM.  Replaces, for a whole index, the record-at-a-time traffic of opening / filling a KV store
 * (BerkeleyDBStorage, bigsi/storage/berkeleydb.py:6-19; KmerSignatureIndex.create -> BitMatrix.create -> set_rows one batch_set per
 * row block, bigsi/graph/index.py:27-40, bigsi/matrix/bitmatrix.py:19-25; the build command's loop, bigsi/cmds/build.py:43-73).
 * `threads` host threads (0 = a quarter of the cores, at most 16) pread / pwrite slices of a 256 MB pinned buffer while the other
 * buffer is in flight over PCIe (hipMemcpyAsync); with row_bytes == row_stride_bytes (the hip-hbm snapshot layout) the bytes
 * go to their place without a kernel, otherwise through one scatter / gather kernel per buffer.  Synchronous for the caller. */
typedef struct {
    uint64_t bytes;        /* n_rows * row_bytes                                     */
    double seconds;        /* wall time of the call (open .. last byte on the device / fsync) */
    double file_seconds;   /* of which inside pread / pwrite (the file system's share; the rest overlaps or is PCIe) */
    uint32_t threads;      /* host threads used for the file                          */
    uint32_t direct;       /* 1: row_bytes == the device pitch, no kernel on the way  */
} bigsi_hip_io_stats;
/* `path` may also be
 *   - a DIRECTORY (the path ends in '/'): the same rows striped over 16 part files (stripes of ~4 MB; `layout` in the directory
 *     records the geometry).  Writes to ONE file are serialised by its inode -- 16 threads put 3.6-5 GB/s into one fresh tmpfs
 *     file and 63 GB/s into 16 -- so a save that is to run at the PCIe rate needs several files; file_offset must be 0;
 *   - (load only, file_offset 0) a BerkeleyDB HASH file, the reference's default store (bigsi/storage/berkeleydb.py:6-19): the rows
 *     are its "<row>:bitarray" records (bigsi/storage/base.py:29-36), found by one threaded scan of its hash pages and gathered
 *     from wherever they lie (inline, or an overflow chain of pages), zero-extended or cut to row_bytes; rows the store does not
 *     hold read as zeros.  bigsi_hip_bdb_small_records hands over everything else the store holds. */
int bigsi_hip_load_rows_file(bigsi_hip_index *ix, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                             uint32_t threads, bigsi_hip_io_stats *stats /* may be NULL */);
/* The records of a BerkeleyDB hash file that are NOT rows -- the index integers ("number_of_rows:int", ...) and the sample
 * metadata, a few bytes each -- packed as [u32 key_len][u32 value_len][key][value]... in key order; *needed = bytes of that
 * (out == NULL: the sizing call), *n_rows / *max_row_bytes = how many "<row>:bitarray" records the file holds and the longest.
 * No device involved.  Read without libdb (layout per Berkeley DB's public db_page.h, hash versions 7-10). */
int bigsi_hip_bdb_small_records(const char *path, uint8_t *out, uint64_t capacity, uint64_t *needed, uint64_t *n_rows, uint64_t *max_row_bytes,
                                uint32_t threads);
int bigsi_hip_save_rows_file(bigsi_hip_index *ix, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                             uint32_t threads, bigsi_hip_io_stats *stats /* may be NULL */);

/* BitMatrix.insert_column (bigsi/matrix/bitmatrix.py:67-75) / BaseStorage.set_bits (bigsi/storage/base.py:111-122):
 * bloom = one sample's Bloom filter, ceil(num_rows/8) bytes in the row format above (bit r = row r);
 * writes bit `col` of every row.  col may equal num_cols (append; num_cols grows by one). */
int bigsi_hip_insert_column(bigsi_hip_index *ix, uint64_t col, const uint8_t *bloom);
/* transpose + BitMatrix.create (bigsi/matrix/transpose.py:33-43, bigsi/graph/index.py:27-40), on the device: n Bloom
 * filters (filter i at blooms + i*bloom_stride_bytes, ceil(num_rows/8) bytes each) become columns [col0, col0+n).
 * col0 <= num_cols; num_cols grows to col0+n if that is larger.  Needs col0+n <= col_capacity. */
int bigsi_hip_insert_columns(bigsi_hip_index *ix, uint64_t col0, uint64_t n, const uint8_t *blooms, uint64_t bloom_stride_bytes);
/* KmerSignatureIndex.merge_indexes (bigsi/graph/index.py:54-60): append all columns of src after dst's, device to
 * device (same device, same num_rows); dst's capacity grows as needed. */
int bigsi_hip_append_index(bigsi_hip_index *dst, const bigsi_hip_index *src);
/* BitMatrix.get_column (bigsi/matrix/bitmatrix.py:50-61): out = ceil(num_rows/8) bytes. */
int bigsi_hip_get_column(bigsi_hip_index *ix, uint64_t col, uint8_t *out);

/* Bloom-add every k-mer of every sequence to sample `col` directly on the transposed matrix: OR of
 * BloomFilter.update (bigsi/bloom/bloomfilter.py:25-32) with canonical k-mers (bigsi/graph/bigsi.py:151). */
int bigsi_hip_insert_kmers(bigsi_hip_index *ix, uint64_t col, const char *seqs, const uint64_t *offsets,
                           uint32_t n_seqs, uint32_t k);

/* Seeded synthetic contents, generated on the device (never crosses PCIe): word w of row r =
 * AND of `and_draws` draws of a counter-based hash of (seed, shard, r, w); bit density 2^-and_draws.
 * The CPU mirror is oracle/bigsi_oracle.c: orc_synth_word.  Not in the reference (benchmark input). */
int bigsi_hip_fill_synthetic(bigsi_hip_index *ix, uint64_t seed, uint64_t shard, uint32_t and_draws);

/* BIGSI.bloom (bigsi/graph/bigsi.py:150-155) + BloomFilter (bigsi/bloom/bloomfilter.py:16-32, zero-initialised):
 * u k-mers of k ASCII bytes each -> Bloom filter of m bits, out = ceil(m/8) bytes.  Needs no index.
 * BIGSI_BLOOM_RAW hashes the elements as given (BloomFilter.add / generate_hashes, bloomfilter.py:9-27);
 * without it they are canonicalised first, as BIGSI.bloom does (graph/bigsi.py:151). */
#define BIGSI_BLOOM_RAW 1u
int bigsi_hip_bloom(int device, const char *kmers, uint64_t u, uint32_t k, uint64_t m, uint32_t h, uint32_t flags, uint8_t *out);

/* ================================================================== BATCHES (and CORE: bigsi_hip_lookup): fused query path
 * One batch = n_seqs query sequences (ASCII, concatenated; sequence i = seqs[offsets[i] .. offsets[i+1])).
 * create  uploads them and sizes the device workspace.
 * run     launches, asynchronously on the index's stream, the whole of BIGSI.search up to the hit list
 *         (bigsi/graph/bigsi.py:174-230 with bigsi/graph/index.py:42-80, bigsi/utils/fncts.py:24-65,
 *         bigsi/bloom/bloomfilter.py:5-13, bigsi/matrix/bitmatrix.py:30-37, bigsi/storage/base.py:96-109):
 *           K1 k-merise, dedupe query k-mers, canonicalise, MurmurHash3 -> h row ids per unique k-mer;
 *           K2 fetch rows, AND across h, then AND across k-mers (threshold == 1.0, exact_filter) or
 *              per-sample counts in bit-sliced counters (threshold < 1, inexact_filter);
 *           K4 min_kmers = ceil(num_unique * threshold) in IEEE double (graph/bigsi.py:179), keep samples with
 *              count >= min_kmers, compact to (colour, count) lists in ascending colour order.
 * fetch_* synchronise and copy results out.  A batch can be run any number of times. */
/* KmerSignatureIndex.lookup for an explicit k-mer list (bigsi/graph/index.py:42-49): u k-mers of k ASCII bytes,
 * out_rows = u rows of row_bytes bytes (AND of each k-mer's h rows), in input order.  One-shot, synchronous. */
int bigsi_hip_lookup(bigsi_hip_index *ix, const char *kmers, uint32_t k, uint64_t u, uint8_t *out_rows);
/* The same for u elements of any byte lengths, hashed as they are (already canonical): the non-ASCII twin of bigsi_hip_lookup. */
int bigsi_hip_lookup_raw(bigsi_hip_index *ix, const char *blob, const uint64_t *elem_offsets, uint64_t u, uint8_t *out_rows);

#define BIGSI_RUN_FORCE_COUNTS 1u /* use the counting path even when threshold == 1.0 */
#define BIGSI_RUN_SKIP_COMPACT 2u /* stop after K2/K3: the caller compacts a gathered buffer instead (multi-GPU) */
#define BIGSI_RUN_SPARSE_COUNTS 8u /* counting path: store per-sample counters only where a sample reaches min_kmers
                                      (hit lists are complete; fetch_counts is unavailable for that run) */
#define BIGSI_RUN_EARLY_EXIT 32u /* stop fetching a query's rows for a column segment once its result is settled: exact, the
                                    running AND is all zero; thresholded (with SPARSE_COUNTS), no sample of the segment can reach
                                    min_kmers with the k-mers that are left.  Same hit lists, fewer bytes than the reference
                                    reads -- hence opt-in (config key `early_exit` of the host shim) */
int bigsi_hip_batch_create(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs,
                           uint32_t k, bigsi_hip_batch **out);
/* A batch whose k-mers are given explicitly ("elements"), for sequences that are not byte strings of k-byte windows: the
 * reference k-merises CHARACTERS (bigsi/utils/fncts.py:63-65), reverse-complements them one by one (:12,38-39), compares
 * Python strings (:51-54) and hashes the UTF-8 bytes (bigsi/bloom/bloomfilter.py:5-6), so the k-mers of a non-ASCII query have
 * different byte lengths.  The host lists, per sequence, its unique k-mers in first-occurrence order, already canonical
 * (element e = blob[elem_offsets[e] .. elem_offsets[e+1]); sequence i owns elements [seq_elem_offsets[i], seq_elem_offsets[i+1])),
 * and for every k-mer position of every sequence the index of its unique k-mer (pos_unique, sequence i owns
 * [seq_pos_offsets[i], seq_pos_offsets[i+1])).  Hashing, row fetch, AND, counts, threshold, compaction and presence run on the
 * device as for any batch; bigsi_hip_batch_reload is not available for such a batch. */
int bigsi_hip_batch_create_elements(bigsi_hip_index *ix, const char *blob, const uint64_t *elem_offsets, const uint64_t *seq_elem_offsets,
                                    const uint32_t *pos_unique, const uint64_t *seq_pos_offsets, uint32_t n_seqs, bigsi_hip_batch **out);
int bigsi_hip_batch_destroy(bigsi_hip_batch *b);
/* Load a different set of sequences into an existing batch object: device buffers are kept and only grow, so a serving
 * loop pays allocation once.  Output / stream settings of the batch are kept; results of earlier runs are discarded.
 * create and reload copy the sequences on a library-internal upload stream and return when the copy is complete; reload
 * waits for the earlier work of THIS batch only, so it can be called while another batch's kernels are still running
 * (two alternating batch objects = upload of one overlaps the row-AND kernel of the other). */
int bigsi_hip_batch_reload(bigsi_hip_batch *b, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k);
int bigsi_hip_batch_run(bigsi_hip_batch *b, double threshold, uint32_t flags);

typedef struct {
    uint32_t n_seqs;
    uint32_t k;
    uint32_t exact;        /* 1: last run took the exact (AND) path, 0: counting path */
    uint32_t count_bytes;  /* bytes per per-sample counter of the last counting run (2 or 4) */
    uint64_t total_kmers;  /* sum over sequences of len - k + 1                  */
    uint64_t total_unique; /* sum of unique query k-mers (after run)             */
    uint64_t total_hits;   /* hits of the last run                               */
    uint64_t bitmap_stride_bytes; /* pitch of one sequence in the exact result buffer  */
    uint64_t counts_stride;       /* counters per sequence in the counts buffer        */
    void *d_bitmaps;       /* device: n_seqs x bitmap_stride_bytes, row format, exact runs   */
    void *d_counts;        /* device: n_seqs x counts_stride counters, counting runs         */
    void *d_num_unique;    /* device: uint32[n_seqs]                                         */
    uint32_t one_launch;   /* 1: the last run was the one-launch read kernel (K1 + K2 + K4 fused) */
    uint32_t reserved;
} bigsi_hip_batch_info;
int bigsi_hip_batch_get_info(bigsi_hip_batch *b, bigsi_hip_batch_info *out); /* synchronises */

/* per sequence: number of k-mers n (with duplicates), unique query k-mers u = len(set(kmers))
 * (graph/bigsi.py:177-179) and min_kmers.  Any pointer may be NULL. */
int bigsi_hip_batch_fetch_unique(bigsi_hip_batch *b, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers);
/* hits of sequence i are [hit_offsets[i], hit_offsets[i+1]) of colours/counts, ascending colour.
 * BIGSI_ERR_CAPACITY (with hit_offsets filled) if capacity < hit_offsets[n_seqs]. */
int bigsi_hip_batch_fetch_hits(bigsi_hip_batch *b, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity);
/* counting runs: per-sample k-mer counts of one sequence (unpack_and_sum, graph/bigsi.py:35-44), out[num_cols]. */
int bigsi_hip_batch_fetch_counts(bigsi_hip_batch *b, uint32_t seq, uint32_t *out);
/* exact runs: AND of all fetched rows of one sequence (graph/bigsi.py:192-195), out[row_bytes], row format. */
int bigsi_hip_batch_fetch_bitmap(bigsi_hip_batch *b, uint32_t seq, uint8_t *out);
/* row ids K1 produced for one sequence: num_unique x num_hashes, k-mers in first-occurrence order, seeds 0..h-1. */
int bigsi_hip_batch_fetch_rows(bigsi_hip_batch *b, uint32_t seq, uint64_t *rows, uint64_t capacity);
/* KmerSignatureIndex.lookup for one sequence (graph/index.py:42-49): for each unique k-mer (first-occurrence
 * order) its position in the sequence and the AND of its h rows (row_bytes bytes each, row format). */
int bigsi_hip_batch_lookup(bigsi_hip_batch *b, uint32_t seq, uint32_t *first_pos, uint8_t *out_rows, uint64_t capacity_rows);
/* BIGSI.score's presence strings (graph/bigsi.py:232-237): for each of n_colours colours, n ASCII '0'/'1'
 * characters, one per k-mer position of the sequence in order (duplicates included).  out[n_colours * n]. */
int bigsi_hip_batch_presence(bigsi_hip_batch *b, uint32_t seq, const uint32_t *colours, uint32_t n_colours, uint8_t *out);
/* The same for the hits of EVERY sequence of the batch in one pass (score=True on a thresholded search with thousands of hits
 * per query): colours of sequence i = colours[hit_offsets[i] .. hit_offsets[i+1]) in any order (the layout fetch_hits
 * returns).  The string of hit t starts at out[string_offsets[t]] -- always a multiple of 16 -- and is num_kmers(sequence
 * of t) characters long (the bytes up to the next string are padding); string_offsets gets hit_offsets[n_seqs] + 1
 * entries, the last one the bytes needed.  BIGSI_ERR_CAPACITY (string_offsets filled) if out_capacity is smaller. */
int bigsi_hip_batch_presence_hits(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, uint8_t *out,
                                  uint64_t out_capacity, uint64_t *string_offsets);

/* CORE: BIGSI.score on the device (K6).  Replaces, for every hit of the batch, BIGSI.score's column -> string -> Scorer.score
 * chain (bigsi/graph/bigsi.py:232-239, bigsi/scoring/score.py:7-107): remove_short_ones, tabulate_score and
 * Scorer.calculate_score (its three scores with Python's round(x, 2) after every gap, the SNP totals, math.ceil / floor) run on
 * the device in IEEE doubles, bit-equal to CPython; so does BigsiQueryResult's percent_kmers_found (graph/bigsi.py:97-99).
 * The caller derives the remaining Scorer.score fields in closed form (score.py:104-121: nident / pident from the mismatch
 * counts, evalue / pvalue / log_* from `score` through exp / log10 -- kept on the host because no two libm agree bit for bit). */
typedef struct {
    double score, min_score, max_score;  /* score.py:86-88                                         */
    double percent_kmers_found;          /* round(100 * float(found) / num_unique, 2)               */
    int64_t max_mismatches, min_mismatches, mismatches; /* score.py:89-93                           */
    uint32_t num_kmers;                  /* n: length of the hit's presence string                 */
    uint32_t reserved;
} bigsi_hip_hit_score;
/* hit_offsets / colours as for bigsi_hip_batch_presence_hits; counts[t] = k-mers hit t found (what fetch_hits returned; NULL:
 * every hit found all unique k-mers of its sequence, i.e. an exact search).  scores[t] is the record of hit t.  The presence
 * string itself comes back as BITS: hit t's n positions start at bits[bit_offsets[t]] (a multiple of 8 bytes), position p in
 * byte p / 8 under mask 0x80 >> (p % 8) -- bitarray(presence_string).tobytes(), zero-padded to whole 8-byte words;
 * bit_offsets gets hit_offsets[n_seqs] - hit_offsets[0] + 1 entries, the last one the bytes needed.
 * BIGSI_ERR_CAPACITY (bit_offsets filled) when bits_capacity is smaller. */
int bigsi_hip_batch_score_hits(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts,
                               uint8_t *bits, uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores);
/* The same in two halves, for serving loops: _begin takes the hit lists, fills bit_offsets (so the caller can size `bits`:
 * bit_offsets[n_hits] bytes) and queues the device work; _end waits for it and copies the results out.  Between the two the
 * caller is free to do anything else -- including running the batch again: the results are staged in host memory the
 * batch owns (not after bigsi_hip_batch_reload / destroy).  One request per batch at a time.
 * By default the kernels run on the library's high-priority score stream BESIDE whatever the index stream is doing (next to a
 * row-AND kernel they spend most of their time waiting for memory: 0.4 ms instead of 0.1 ms for a few hundred hits -- which
 * does not matter to a loop three batches deep, that collects the results a step later: 1.03 ms per step at BASELINE configs[4]).
 * BIGSI_SCORE_ORDERED queues them on the index's stream BEHIND the runs already issued there: a tenth of the device time, but
 * on that stream's critical path (1.14 ms per step in the same loop); the choice when the device has nothing else to do. */
#define BIGSI_SCORE_ORDERED 1u
int bigsi_hip_batch_score_hits_begin(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts,
                                     uint32_t flags, uint64_t *bit_offsets);
int bigsi_hip_batch_score_hits_end(bigsi_hip_batch *b, uint8_t *bits, uint64_t bits_capacity, bigsi_hip_hit_score *scores);
/* Scorer.score (bigsi/scoring/score.py:96-121, the part listed above) for n presence strings the caller holds as bits in the
 * layout bigsi_hip_batch_score_hits returns (string t: num_kmers[t] positions at bits + bit_offsets[t], multiple of 8).  found /
 * unique feed percent_kmers_found and may be NULL.  Needs no index. */
int bigsi_hip_score_presence(int device, const uint8_t *bits, const uint64_t *bit_offsets, const uint32_t *num_kmers,
                             const uint32_t *found, const uint32_t *unique, uint64_t n, bigsi_hip_hit_score *scores);

/* MULTI-GPU: the hit lists of a sharded run (bigsi_hip_batch_run_sharded below), identical on every rank: synchronises and copies out,
 * same format as fetch_hits with global colours = shard * shard_cols + local column. */
int bigsi_hip_batch_fetch_gathered_hits(bigsi_hip_batch *b, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity);

/* ================================================================== CORE: one-call search
 * The whole of BIGSI.search for a batch of sequences in ONE call (what a non-Python binder of this boundary needs):
 * load + run + fetch_unique + fetch_hits on a workspace the index keeps from call to call (its device buffers are allocated
 * once and only grow).  Outputs as in fetch_unique / fetch_hits; any of num_kmers / num_unique / min_kmers may be NULL.
 * BIGSI_ERR_CAPACITY (hit_offsets filled) when hit_capacity is too small. */
int bigsi_hip_search_batch(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                           double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                           uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity);

/* BIGSI.search for ANY number of sequences in one call -- what bulk_search (bigsi/__main__.py:261-314) does with a fork pool and one
 * BIGSI.search per query.  The library cuts the input into device batches (about 2^20 k-mer positions each), keeps four
 * workspaces in flight -- while one batch runs, the next is staged and uploaded and the results of the one before are exported and
 * copied out: pinned staging, asynchronous copies, one wait per batch -- and writes each sequence's results at its place in the
 * caller's arrays.  Outputs as bigsi_hip_search_batch with n_seqs-long arrays; hit_offsets (n_seqs + 1 entries) index colours /
 * counts globally.  BIGSI_ERR_CAPACITY (hit_offsets complete, lists filled as far as they fit) when hit_capacity is too small. */
int bigsi_hip_search_stream(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                            double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                            uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity);

/* BIGSI.search(..., score=True) (bigsi/graph/bigsi.py:174-190, 232-239 with bigsi/scoring/score.py:96-116) for any number of sequences
 * in ONE call: bigsi_hip_search_stream plus, for every hit t (global index into colours), its presence bits at
 * bits + bit_offsets[t] and its score record scores[t] (layout and fields of bigsi_hip_batch_score_hits).  The scoring kernels
 * of one device batch run beside the row-AND kernels of the next.  bit_offsets: hit_capacity + 1 entries; scores:
 * hit_capacity entries; *bits_needed (may be NULL) gets the bytes all presence bits take.  (A sequence without k-mers has, as in
 * bigsi_hip_search_stream, no hits at threshold 1 and every column with count 0 below it; those hits get all-zero records.)
 * BIGSI_ERR_CAPACITY when hit_capacity or bits_capacity is too small: hit_offsets and *bits_needed are complete then, so one
 * retry with hit_offsets[n_seqs] entries and *bits_needed bytes succeeds (bits = NULL, bits_capacity = 0 is a sizing call). */
int bigsi_hip_search_stream_scored(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                                   double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                   uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity, uint8_t *bits,
                                   uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores, uint64_t *bits_needed);

/* ================================================================== MULTI-GPU: column shards, the exchange (RCCL over xGMI)
 * An index too wide for one GPU is split by COLUMN RANGE: shard g holds all num_rows rows of columns
 * [g * shard_cols, (g+1) * shard_cols).  Every shard runs K1-K3 on the same queries; the only exchange is one
 * ncclAllGather per batch of ONE BIT PER SAMPLE (the AND bitmap, or the count >= min_kmers mask of a thresholded search)
 * followed by the compaction of the gathered [shard][seq][stride] buffer on every rank, and -- thresholded searches only --
 * one fixed-size ncclAllReduce (sum) of the per-hit count array, each rank having filled in the hits of its own shard.
 * The reference has no counterpart: its only parallelism is bulk_search's fork pool (bigsi/__main__.py:273-287).
 * librccl.so.1 is loaded with dlopen at the first call (a process that already holds torch's copy shares it).
 *
 * One process per GPU: rank 0 calls bigsi_hip_comm_unique_id and passes the 128 bytes to the other ranks out of band
 * (torch.distributed store, MPI, a file); every rank then calls bigsi_hip_comm_init_rank for its device. */
#define BIGSI_HIP_COMM_ID_BYTES 128
typedef struct bigsi_hip_comm bigsi_hip_comm;
int bigsi_hip_comm_unique_id(uint8_t *id /* [BIGSI_HIP_COMM_ID_BYTES] */);
int bigsi_hip_comm_init_rank(int device, const uint8_t *id, int rank, int world, bigsi_hip_comm **out);
int bigsi_hip_comm_destroy(bigsi_hip_comm *c);
int bigsi_hip_comm_info(const bigsi_hip_comm *c, int *rank, int *world); /* as the communicator reports them (ncclCommCount) */
/* Attach a batch to its rank's communicator: results are produced `shard_cols` wide (see set_result_cols; the index's
 * capacity grows to it if needed) straight into this rank's slot of a library-owned gather buffer.  NULL detaches. */
int bigsi_hip_batch_set_comm(bigsi_hip_batch *b, bigsi_hip_comm *c, uint64_t shard_cols);
/* K1-K3 on the index's stream, then -- queued behind them on the communicator's own stream, no host-side wait -- all-gather,
 * gathered compaction and count all-reduce.  With two batches used alternately the exchange of one overlaps the row-AND
 * kernel of the other.  Results: bigsi_hip_batch_fetch_unique / bigsi_hip_batch_fetch_gathered_hits (global colours =
 * shard * shard_cols + local column), identical on every rank. */
int bigsi_hip_batch_run_sharded(bigsi_hip_batch *b, double threshold, uint32_t flags);

/* One process driving several GPUs (storage-config {"devices": [0, 1, ...]}): include/bigsi_hip_group.h -- a group owns one
 * column shard per device and issues the same exchange for all of them (ncclCommInitAll, ncclGroupStart/End). */

/* ================================================================== MEASUREMENT */
typedef struct {
    uint64_t and_launches;   /* row-fetch-AND kernel launches timed since the last reset   */
    double and_ms;           /* their summed duration (HIP events on the launch stream)     */
    uint64_t kmerize_launches;
    double kmerize_ms;
    uint64_t compact_launches;
    double compact_ms;
    uint64_t presence_launches; /* bigsi_hip_batch_presence_hits calls: K5 (both kernels) */
    double presence_ms;
    uint64_t presence_bytes;    /* algorithmic bytes of those calls: unique k-mers x h x 8 x distinct hit words + string bytes */
    uint64_t transpose_launches; /* bigsi_hip_insert_columns_device calls (the build transpose, filters resident) */
    double transpose_ms;
    uint64_t and_launches_total; /* row-AND launches since the last reset, timed or not (and_launches counts the timed ones) */
    uint64_t read_launches_repeated; /* always 0 since round 4 (rounds 2-3: read launches repeated after a bounded wait between
                                        workgroups ran out; no workgroup of any kernel waits for another any more); kept for the layout */
    uint64_t index_contiguous;       /* 1: the matrix got physically contiguous device memory (hipDeviceMallocContiguous: largest
                                        page-table fragments), 0: the ordinary allocation it falls back to */
    uint64_t exchange_launches;      /* bigsi_hip_batch_run_sharded calls timed (profiling level 1) */
    double exchange_ms;              /* their all-gather + gathered compaction + count all-reduce, by events on the communicator's stream */
} bigsi_hip_stats_t;
/* record HIP events around the kernels of batch_run: 0 off, 1 around K1 / K2 / K4 each, 2 around the row-AND kernel only,
 * n > 2 around the row-AND kernel of every n-th run (an event record costs the stream 5-7 us, which is a fifth of a
 * step for a batch of short reads: sampled, the timed region runs at its untimed speed) */
int bigsi_hip_set_profiling(bigsi_hip_index *ix, int on);
int bigsi_hip_stats(bigsi_hip_index *ix, bigsi_hip_stats_t *out, int reset); /* synchronises */

/* FRONT-END TEXT (host only: FASTA in, the reference's JSON / CSV out, over the arrays of the streaming searches): declared in
 * include/bigsi_hip_text.h, exported by the same library. */

#ifdef __cplusplus
}
#endif
#endif /* BIGSI_HIP_H */
