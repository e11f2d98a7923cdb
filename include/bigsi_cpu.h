/*
 * bigsi_cpu.h -- libbigsi_cpu.so: the CPU twin of the CORE layer of include/bigsi_hip.h ("a bigsi_cpu_*
 * twin with identical signatures is the CPU restatement / baseline").
 *
 * Same entry points, same argument meaning, same error codes, same row format, same results -- computed on the host, the way
 * the reference computes them (bigsi/graph/index.py:42-80, bigsi/graph/bigsi.py:35-56,174-242): per k-mer canonicalisation on
 * strings, MurmurHash3 x h, one copy per fetched row, byte-wise AND, unpack-to-int32-and-add.  It exists so that
 *   - a host WITHOUT a GPU can bind the same boundary (it links this library explicitly; the hip-hbm backend never loads it:
 *     there is no fallback, bigsi_amd raises when libbigsi_hip.so is missing);
 *   - the CPU baseline of bench.py is produced THROUGH the product boundary, not through test infrastructure;
 *   - a host written against bigsi_hip.h can be built against either library unchanged (-DBIGSI_USE_CPU_TWIN, below).
 * Written from the reference's behaviour (citations at each entry point in bigsi_hip.h); shares no code with oracle/.
 * Handles are not interchangeable between the two libraries.  `device` arguments are ignored (the one "device" is the host).
 */
#ifndef BIGSI_CPU_H
#define BIGSI_CPU_H

#include "bigsi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bigsi_cpu_index bigsi_cpu_index;

/* search flag of this library only: operate on 64-bit words of the resident rows (no per-row copies, counters touched only at
 * set bits) instead of the reference's shape -- the "best CPU" line of the baseline.  Results are identical. */
#define BIGSI_CPU_WORD_PARALLEL (1u << 16)

const char *bigsi_cpu_last_error(void);
int bigsi_cpu_device_count(int *out); /* always 1: the host */
int bigsi_cpu_open(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes, int device, bigsi_cpu_index **out);
int bigsi_cpu_close(bigsi_cpu_index *ix);
/* An index whose rows STAY in the reference's own store -- a BerkeleyDB hash file (bigsi/storage/berkeleydb.py:6-19) with the
 * "<row>:bitarray" records and index integers of a v0.3 index (bigsi/storage/base.py:29-36): every row a search needs is read from
 * the file when it is needed (an overflow chain of pages for wide rows), as the reference's `storage[key]` does; a table built by one
 * scan of the hash pages stands in for libdb's bucket lookup (no libdb on these hosts).  Read-only: reference-shaped search_batch /
 * search_stream, lookup, get_rows, presence.  What bench.py's cpu_baseline reports as its BerkeleyDB-file variant. */
int bigsi_cpu_open_bdb(const char *path, uint32_t threads, bigsi_cpu_index **out);
int bigsi_cpu_get_info(const bigsi_cpu_index *ix, bigsi_hip_info *out);
int bigsi_cpu_set_num_cols(bigsi_cpu_index *ix, uint64_t num_cols);
int bigsi_cpu_set_num_hashes(bigsi_cpu_index *ix, uint32_t num_hashes);
int bigsi_cpu_reserve_cols(bigsi_cpu_index *ix, uint64_t col_capacity);
int bigsi_cpu_synchronize(bigsi_cpu_index *ix);
int bigsi_cpu_set_rows(bigsi_cpu_index *ix, const uint64_t *row_ids, uint64_t n, const uint8_t *bytes, uint64_t row_bytes);
int bigsi_cpu_get_rows(bigsi_cpu_index *ix, const uint64_t *row_ids, uint64_t n, uint8_t *out, uint64_t row_bytes);
int bigsi_cpu_clear(bigsi_cpu_index *ix);
int bigsi_cpu_insert_column(bigsi_cpu_index *ix, uint64_t col, const uint8_t *bloom);
int bigsi_cpu_insert_columns(bigsi_cpu_index *ix, uint64_t col0, uint64_t n, const uint8_t *blooms, uint64_t bloom_stride_bytes);
int bigsi_cpu_get_column(bigsi_cpu_index *ix, uint64_t col, uint8_t *out);
int bigsi_cpu_insert_kmers(bigsi_cpu_index *ix, uint64_t col, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k);
int bigsi_cpu_fill_synthetic(bigsi_cpu_index *ix, uint64_t seed, uint64_t shard, uint32_t and_draws);
int bigsi_cpu_bloom(int device, const char *kmers, uint64_t u, uint32_t k, uint64_t m, uint32_t h, uint32_t flags, uint8_t *out);
int bigsi_cpu_lookup(bigsi_cpu_index *ix, const char *kmers, uint32_t k, uint64_t u, uint8_t *out_rows);
int bigsi_cpu_search_batch(bigsi_cpu_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                           double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                           uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity);
int bigsi_cpu_search_stream(bigsi_cpu_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                            double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                            uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity);
int bigsi_cpu_search_stream_scored(bigsi_cpu_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                                   double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                   uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity, uint8_t *bits,
                                   uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores, uint64_t *bits_needed);
int bigsi_cpu_score_presence(int device, const uint8_t *bits, const uint64_t *bit_offsets, const uint32_t *num_kmers,
                             const uint32_t *found, const uint32_t *unique, uint64_t n, bigsi_hip_hit_score *scores);
/* BIGSI.score's presence strings (bigsi/graph/bigsi.py:232-237) of ONE sequence for n_colours samples: n = len - k + 1 ASCII
 * '0'/'1' characters per colour, k-mer positions in order (duplicates included).  (The HIP library offers this per batch:
 * bigsi_hip_batch_presence.) */
int bigsi_cpu_presence(bigsi_cpu_index *ix, const char *seq, uint64_t len, uint32_t k, const uint32_t *colours, uint32_t n_colours, uint8_t *out);

#ifdef __cplusplus
}
#endif

/* A host written against bigsi_hip.h, built against the twin: compile it with
 *     -DBIGSI_USE_CPU_TWIN -include bigsi_cpu.h      and link   -lbigsi_cpu
 * (tests/c_host/search_host.c is built both ways; its output is the same). */
#ifdef BIGSI_USE_CPU_TWIN
#define bigsi_hip_index bigsi_cpu_index
#define bigsi_hip_last_error bigsi_cpu_last_error
#define bigsi_hip_device_count bigsi_cpu_device_count
#define bigsi_hip_open bigsi_cpu_open
#define bigsi_hip_close bigsi_cpu_close
#define bigsi_hip_get_info bigsi_cpu_get_info
#define bigsi_hip_set_num_cols bigsi_cpu_set_num_cols
#define bigsi_hip_set_num_hashes bigsi_cpu_set_num_hashes
#define bigsi_hip_reserve_cols bigsi_cpu_reserve_cols
#define bigsi_hip_synchronize bigsi_cpu_synchronize
#define bigsi_hip_set_rows bigsi_cpu_set_rows
#define bigsi_hip_get_rows bigsi_cpu_get_rows
#define bigsi_hip_clear bigsi_cpu_clear
#define bigsi_hip_insert_column bigsi_cpu_insert_column
#define bigsi_hip_insert_columns bigsi_cpu_insert_columns
#define bigsi_hip_get_column bigsi_cpu_get_column
#define bigsi_hip_insert_kmers bigsi_cpu_insert_kmers
#define bigsi_hip_fill_synthetic bigsi_cpu_fill_synthetic
#define bigsi_hip_bloom bigsi_cpu_bloom
#define bigsi_hip_lookup bigsi_cpu_lookup
#define bigsi_hip_search_batch bigsi_cpu_search_batch
#define bigsi_hip_search_stream bigsi_cpu_search_stream
#define bigsi_hip_search_stream_scored bigsi_cpu_search_stream_scored
#define bigsi_hip_score_presence bigsi_cpu_score_presence
#endif

#endif /* BIGSI_CPU_H */
