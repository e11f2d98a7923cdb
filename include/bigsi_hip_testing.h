/*
 * bigsi_hip_testing.h -- entry points and flags of libbigsi_hip.so that are exported but NOT part of the advertised boundary
 * (include/bigsi_hip.h): what the test-suite and the A/B scripts need to drive the library from outside.
 *
 *   - EXCHANGE-BY-CALLER: a host that brings its own collective instead of the library's RCCL communicator -- here
 *     torch.distributed over gloo, so that several ranks can share the one GPU of a test box (RCCL refuses two ranks on one
 *     device): caller-owned streams and result buffers, compaction of a buffer the caller gathered.
 *   - BIGSI_RUN_* flags that force a route (same results by construction; tests/test_gpu_parity.py proves it).
 * A production binder needs none of this.
 */
#ifndef BIGSI_HIP_TESTING_H
#define BIGSI_HIP_TESTING_H

#include "bigsi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test / A-B flags: same results, different route (tests/test_gpu_parity.py, scripts/ab_*.py); not for production callers */
#define BIGSI_RUN_K1_GLOBAL 4u   /* take the multi-launch K1 (global-memory dedupe table) whatever the query lengths */
#define BIGSI_RUN_NO_SORT 16u    /* stream each query's rows in hash order instead of address order */
/* (128u was BIGSI_RUN_NO_WAITING in rounds 2-3: the read kernel's bounded wait between workgroups; the kernel no longer waits) */
#define BIGSI_RUN_ONE_STREAM 256u /* one-launch read path: this run goes to the index stream instead of the next of the three read
                                    streams, so that consecutive launches do NOT overlap -- a kernel's own duration is then what it
                                    takes alone on the device (bench.py: roofline.frac of read workloads) */
#define BIGSI_RUN_WEAK_FINGERPRINT 64u /* one-launch read path: 1-bit k-mer fingerprints, so that the dedupe takes its exact
                                          pairwise route (otherwise reached only on a 2^-32 fingerprint collision) */

/* Run this index's kernels and copies on a caller-owned hipStream_t (e.g. torch's current stream, so
 * that RCCL collectives issued by the caller are ordered after them).  NULL restores the private stream.
 * With a caller-owned stream set, every kernel of the index goes to that one stream (see STREAMS in bigsi_hip.h). */
int bigsi_hip_set_stream(bigsi_hip_index *ix, void *hip_stream);

/* Write the per-sample result of later runs into caller-owned device memory (e.g. this rank's slot of an
 * RCCL all-gather buffer) instead of the batch's own buffers.  Either may be NULL (= keep own buffer). */
int bigsi_hip_batch_set_outputs(bigsi_hip_batch *b, void *d_bitmaps, void *d_counts);
/* Width, in columns, of the per-sample result vectors of later runs (default 0 = the index's num_cols).  The shards of one
 * index all set the group's shard width here, so that uneven shards still exchange buffers of ONE geometry (strides, word
 * counts); needs cols <= col_capacity.  Columns beyond the shard's own num_cols read as zero. */
int bigsi_hip_batch_set_result_cols(bigsi_hip_batch *b, uint64_t cols);

/* EXCHANGE-BY-CALLER.  Multi-GPU assembly: compact hits from result buffers gathered from n_shards column shards (device pointer,
 * layout [shard][seq][stride] with this batch's strides; colour = shard * shard_cols + local column).
 * compact_gathered* are asynchronous, also for the host: they are queued (on the gather stream, below) behind this batch's
 * run through an event, never by waiting for it -- call them after the RCCL all-gather, which the caller issues on the
 * same stream; fetch_gathered_hits synchronises and copies out, same format as fetch_hits. */
/* Run this batch's gathered compaction (and the copies of fetch_gathered_hits) on a caller-owned hipStream_t -- typically
 * the stream the collective is issued under, so that all-gather + compaction of one batch overlap the row-AND kernels of
 * the next batch on the index's stream.  NULL = the index's stream. */
int bigsi_hip_batch_set_gather_stream(bigsi_hip_batch *b, void *hip_stream);
int bigsi_hip_batch_compact_gathered(bigsi_hip_batch *b, const void *d_gathered, uint32_t n_shards, uint64_t shard_cols);
/* Thresholded search over column shards without moving per-sample counters: the counting kernel also leaves each
 * shard's hit mask (1 bit per sample: count >= min_kmers) in the bitmap output (bigsi_hip_batch_set_outputs), which is what
 * gets all-gathered.  compact_gathered_masks compacts the gathered masks -- identically on every rank -- and fills each
 * hit's count from THIS rank's counters when the hit lies in shard `own_shard`, 0 otherwise; the caller then sums the
 * count arrays of all ranks (one fixed-size all-reduce over the buffer given to set_gathered_hit_outputs). */
int bigsi_hip_batch_compact_gathered_masks(bigsi_hip_batch *b, const void *d_gathered_masks, uint32_t n_shards, uint64_t shard_cols,
                                           uint32_t own_shard);
/* Put the gathered hit lists (colours, counts: uint32[capacity] each) into caller-owned device memory.  With caller-owned
 * buffers fetch_gathered_hits reports BIGSI_ERR_CAPACITY instead of growing them. */
int bigsi_hip_batch_set_gathered_hit_outputs(bigsi_hip_batch *b, void *d_colours, void *d_counts, uint64_t capacity);

/* MEASUREMENT (bench.py, scripts/measure.py).
 * bigsi_hip_insert_columns with the filters already in device memory (no staging copy: what prices the transpose kernel alone).
 * A 16-byte aligned pointer and pitch take the tiled transpose; anything else the column-at-a-time route. */
int bigsi_hip_insert_columns_device(bigsi_hip_index *ix, uint64_t col0, uint64_t n, const void *d_blooms, uint64_t bloom_stride_bytes);
/* Same-box calibration: achieved GB/s of bare row streams over this index's matrix -- a kernel with no BIGSI code, the load
 * pattern of the row-AND kernels (one wavefront per 1 KiB column segment, 16 B per lane, 8 loads in flight) -- over n_queries
 * lists of rows_per_query rows, uniform random (sorted = 0: what the counting kernel sees) or ascending (1: the exact kernel's
 * address-ordered lists); launches of `wgs` workgroups (0 = the library's own launch size); median of `reps` passes.
 * Lets a bench line state its fraction of what THIS box delivers (boxes differ by several per cent at identical clocks). */
int bigsi_hip_probe_rows(bigsi_hip_index *ix, uint32_t rows_per_query, uint32_t n_queries, uint32_t sorted, uint32_t wgs, uint32_t reps,
                         double *gbps, double *launch_ms);

#ifdef __cplusplus
}
#endif
#endif /* BIGSI_HIP_TESTING_H */
