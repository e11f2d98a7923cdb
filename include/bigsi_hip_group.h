/*
 * bigsi_hip_group.h -- device groups of libbigsi_hip.so: ONE process driving several GPUs.
 *
 * Part of the MULTI-GPU layer of the C ABI (include/bigsi_hip.h): the column shards of one index, one per device, behind a
 * single handle.  The reference has no counterpart (its only parallelism is bulk_search's fork pool,
 * bigsi/__main__.py:273-287); this is what the `hip-hbm` backend opens for storage-config {"devices": [0, 1, ...]}, so that one
 * get_storage() call (bigsi/storage/__init__.py:3-19) reaches every GPU of the node.  Every entry point has the meaning of its
 * single-index namesake in bigsi_hip.h (cited there line by line); colours are global (shard * shard_cols + local column).
 */
#ifndef BIGSI_HIP_GROUP_H
#define BIGSI_HIP_GROUP_H

#include "bigsi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One process driving several GPUs: a group owns one column shard per device and a communicator per shard
 * (ncclCommInitAll); every call below fans out to the devices, collectives are issued inside ncclGroupStart/End.
 * This is what the `hip-hbm` backend opens for storage-config {"devices": [0, 1, ...]}: one get_storage() call
 * (bigsi/storage/__init__.py:3-19) reaches every GPU of the node.  shard_cols = ceil(col_capacity / n_dev) rounded up to
 * 64 columns, fixed for the life of the group; colour c lives on shard c / shard_cols.  A device may be listed more than
 * once (testing on a one-GPU box): its shards then exchange through shared device memory instead of RCCL. */
typedef struct bigsi_hip_group bigsi_hip_group;
typedef struct bigsi_hip_group_batch bigsi_hip_group_batch;
int bigsi_hip_group_open(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                         const int *device_ids, int n_dev, bigsi_hip_group **out);
int bigsi_hip_group_close(bigsi_hip_group *g);
typedef struct {
    uint64_t num_rows, num_cols, col_capacity, shard_cols, row_bytes, index_bytes; /* whole index; index_bytes summed over devices */
    uint32_t num_hashes, n_shards;
    uint32_t rccl; /* 1: shards exchange through RCCL; 0: shared device memory (repeated device ids) */
} bigsi_hip_group_info;
int bigsi_hip_group_get_info(const bigsi_hip_group *g, bigsi_hip_group_info *out);
/* the shard on device_ids[i], for the single-index entry points above (bulk fills, profiling, statistics) */
int bigsi_hip_group_shard(bigsi_hip_group *g, uint32_t i, bigsi_hip_index **out);
int bigsi_hip_group_set_num_cols(bigsi_hip_group *g, uint64_t num_cols);
int bigsi_hip_group_set_num_hashes(bigsi_hip_group *g, uint32_t num_hashes);
int bigsi_hip_group_synchronize(bigsi_hip_group *g);
int bigsi_hip_group_clear(bigsi_hip_group *g);
/* storage contract over whole rows (row_bytes = bytes of a row of the WHOLE index, as bigsi_hip_set_rows / get_rows) */
int bigsi_hip_group_set_rows(bigsi_hip_group *g, const uint64_t *row_ids, uint64_t n, const uint8_t *bytes, uint64_t row_bytes);
int bigsi_hip_group_get_rows(bigsi_hip_group *g, const uint64_t *row_ids, uint64_t n, uint8_t *out, uint64_t row_bytes);
int bigsi_hip_group_insert_columns(bigsi_hip_group *g, uint64_t col0, uint64_t n, const uint8_t *blooms, uint64_t bloom_stride_bytes);
int bigsi_hip_group_get_column(bigsi_hip_group *g, uint64_t col, uint8_t *out);
int bigsi_hip_group_insert_kmers(bigsi_hip_group *g, uint64_t col, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k);
int bigsi_hip_group_fill_synthetic(bigsi_hip_group *g, uint64_t seed, uint32_t and_draws); /* shard i = fill_synthetic(seed, i) */
/* bigsi_hip_export_ipc / bigsi_hip_open_ipc for a group: n_shards handles of BIGSI_IPC_HANDLE_BYTES bytes each; the attaching process
 * names its own devices (same shard order, same col_capacity / shard geometry as the owner: bigsi_hip_group_get_info there) and gets a
 * read-only group with communicator, streams and workspaces of its own. */
int bigsi_hip_group_export_ipc(bigsi_hip_group *g, uint8_t *handles /* n_shards x BIGSI_IPC_HANDLE_BYTES */);
int bigsi_hip_group_open_ipc(const uint8_t *handles, uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                             const int *device_ids, int n_dev, bigsi_hip_group **out);
/* bigsi_hip_load_rows_file / bigsi_hip_save_rows_file for a group (KmerSignatureIndex.create, bigsi/graph/index.py:27-40; the
 * store a BerkeleyDBStorage opens, bigsi/storage/berkeleydb.py:6-19): rows [row0, row0 + n_rows) as WHOLE rows of row_bytes bytes
 * each, in the reference's row format, at file_offset of `path`.  Shard i owns bytes [i * shard_cols / 8, +shard_cols / 8) of every
 * row: each 256 MB chunk of the file goes out as one two-dimensional copy per shard from a pinned buffer every device reads, on the
 * shards' own streams (all PCIe links busy at once), while host threads read the next chunk.  A load may name rows longer than the
 * group's capacity (the padded pitch of a single-GPU snapshot of the same index): the excess is skipped.  Bytes of a shard's rows
 * beyond row_bytes are left as they are (zero in an index that was just opened). */
int bigsi_hip_group_load_rows_file(bigsi_hip_group *g, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                                   uint32_t threads, bigsi_hip_io_stats *stats /* may be NULL */);
int bigsi_hip_group_save_rows_file(bigsi_hip_group *g, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                                   uint32_t threads, bigsi_hip_io_stats *stats /* may be NULL */);
int bigsi_hip_group_lookup(bigsi_hip_group *g, const char *kmers, uint32_t k, uint64_t u, uint8_t *out_rows);
int bigsi_hip_group_lookup_raw(bigsi_hip_group *g, const char *blob, const uint64_t *elem_offsets, uint64_t u, uint8_t *out_rows);
/* fused query path over all shards; same meaning as the bigsi_hip_batch_* calls, colours are global */
int bigsi_hip_group_batch_create(bigsi_hip_group *g, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                                 bigsi_hip_group_batch **out);
int bigsi_hip_group_batch_create_elements(bigsi_hip_group *g, const char *blob, const uint64_t *elem_offsets,
                                          const uint64_t *seq_elem_offsets, const uint32_t *pos_unique,
                                          const uint64_t *seq_pos_offsets, uint32_t n_seqs, bigsi_hip_group_batch **out);
int bigsi_hip_group_batch_reload(bigsi_hip_group_batch *gb, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k);
int bigsi_hip_group_batch_destroy(bigsi_hip_group_batch *gb);
int bigsi_hip_group_batch_run(bigsi_hip_group_batch *gb, double threshold, uint32_t flags); /* asynchronous */
int bigsi_hip_group_batch_fetch_unique(bigsi_hip_group_batch *gb, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers);
int bigsi_hip_group_batch_fetch_hits(bigsi_hip_group_batch *gb, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity);
int bigsi_hip_group_batch_presence(bigsi_hip_group_batch *gb, uint32_t seq, const uint32_t *colours, uint32_t n_colours, uint8_t *out);
int bigsi_hip_group_batch_presence_hits(bigsi_hip_group_batch *gb, const uint64_t *hit_offsets, const uint32_t *colours, uint8_t *out,
                                        uint64_t out_capacity, uint64_t *string_offsets);
int bigsi_hip_group_batch_score_hits(bigsi_hip_group_batch *gb, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts,
                                     uint8_t *bits, uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores);
int bigsi_hip_group_search_batch(bigsi_hip_group *g, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                                 double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                 uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity);

#ifdef __cplusplus
}
#endif
#endif /* BIGSI_HIP_GROUP_H */
