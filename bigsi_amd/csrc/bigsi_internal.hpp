// bigsi_internal.hpp -- host-side structures shared by the translation units of libbigsi_hip.so
// (bigsi_hip.hip: single-shard C ABI; bigsi_shard.hip: RCCL exchange, device groups, one-call search).
#pragma once
#include "bigsi_hip.h"
#include "bigsi_hip_group.h"
#include "bigsi_hip_testing.h"
#include "bigsi_hip_text.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------ errors
int bigsi_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail bigsi_fail

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? BIGSI_ERR_NOMEM : BIGSI_ERR_HIP, "%s:%d %s: %s", __FILE__, \
                        __LINE__, #expr, hipGetErrorString(e_));                                       \
    } while (0)

#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_ != BIGSI_OK) \
            return rc_;      \
    } while (0)

static inline uint64_t round_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
static inline uint64_t ceil_div(uint64_t x, uint64_t a) { return (x + a - 1) / a; }

// Tuning knobs (A/B measurements: scripts/ab_*.py) are read from the environment only in builds made with
// -DBIGSI_HIP_TUNING (csrc/build.sh tuning); the product library takes the defaults and never calls getenv.
static inline int env_int(const char *name, int dflt)
{
#ifdef BIGSI_HIP_TUNING
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// tuning builds: host timestamps inside bigsi_hip_search_batch (scripts/call_breakdown.py), summed per phase; nothing in the product
#ifdef BIGSI_HIP_TUNING
#include <time.h>
extern uint64_t g_call_trace[16];      // [i] = ns spent up to mark i since the mark before, summed over calls; [15] = calls
extern uint64_t g_call_last;
static inline uint64_t call_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }
#define CALL_MARK(i) do { const uint64_t n_ = call_now(); if ((i) == 0) g_call_trace[15]++; else g_call_trace[i] += n_ - g_call_last; g_call_last = n_; } while (0)
#else
#define CALL_MARK(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------ device buffer with growth
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool view = false;          // points into another DevBuf (point()): never allocates, never frees
    int reserve(size_t bytes)
    {
        if (view) return fail(BIGSI_ERR_STATE, "internal: reserve() on a buffer view");
        if (bytes <= cap) return BIGSI_OK;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 256);
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return BIGSI_OK;
    }
    void point(void *q, size_t bytes)
    {
        p = q;
        cap = bytes;
        view = true;
    }
    void release()
    {
        if (p && !view) { hipError_t e = hipFree(p); (void)e; }
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    uint32_t launches = 1;      // kernel launches bracketed by the pair (large batches go out as several K2 launches)
};

constexpr int kReadStreams = 4;            // streams created; kReadStreamsUsed of them take launches (tuning builds: BIGSI_HIP_READ_STREAMS)
constexpr int kReadStreamsUsed = 3;

struct bigsi_hip_index {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    // the sequences of a batch are uploaded here, so that loading one batch does not wait for the kernels of another
    // (and, with BIGSI_HIP_K1_OVERLAP=1 only, K1 and the row sort run here too: see k1_stream)
    hipStream_t pre_stream = nullptr;
    hipStream_t rd_stream[kReadStreams] = {};      // k_reads_fused launches alternate over these (created at the first one)
    uint32_t rd_next = 0;
    hipStream_t sc_stream = nullptr;               // K5 / K6 of scored searches: highest priority (score_stream)
    bool rd_pending = false;      // something may still be running on them
    bool sc_pending = false;      // a K5 / K6 request may still be queued on sc_stream (it reads d_index): quiesce_index
    hipEvent_t main_ev = nullptr; // end of the last batch run on the index stream (mark_main)
    uint64_t fused_repeats = 0;   // (rounds 2-3: read launches repeated after a bounded wait ran out; nothing waits any more: stays 0)
    struct bigsi_hip_batch *search_ws = nullptr;      // bigsi_hip_search_batch's workspace, created at its first call
    struct bigsi_hip_batch *stream_ws[6] = {};        // bigsi_hip_search_stream's workspaces (four in use; up to six in tuning builds)
    uint64_t m = 0, n_cols = 0, cap_cols = 0, stride_words = 0;
    uint32_t h = 0;
    uint64_t *d_index = nullptr;
    bool contiguous = false;      // d_index is physically contiguous memory (index_malloc)
    // A handle that does not own its matrix: attached to another process's index over hipIpc (bigsi_hip_open_ipc: closed with
    // hipIpcCloseMemHandle) or a second handle of this process onto an open index (bigsi_hip_open_view: nothing to free).  Such a
    // handle is READ-ONLY: every entry point that writes the matrix fails with BIGSI_ERR_STATE (bigsi_writable).
    enum Attach { kOwner = 0, kIpc = 1, kView = 2 } attach = kOwner;
    // one handle = one host thread at a time (include/bigsi_hip.h): enforced, not just documented -- every entry point that takes
    // the handle (or a batch of it) holds this for the duration of the call; a second thread gets BIGSI_ERR_STATE instead of a
    // race.  Re-entrant for the owning thread (entry points call each other).
    bigsi_hip_index *view_of = nullptr;      // kView: the owner
    std::atomic<int> views{0};               // live views of this (owning) handle
    std::atomic<std::thread::id> busy_owner{};
    uint32_t busy_depth = 0;
    DevBuf stage, stage_ids;
    // bigsi_hip_set_rows: two pinned host buffers + device twins, filled by a few host threads while the other one is in flight
    void *pin_rows[2] = {nullptr, nullptr};
    DevBuf dev_rows[2], dev_ids[2];
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    // profiling
    int profiling = 0;            // 0 off, 1 every kernel group of a run, 2 the row-AND kernel only
    uint32_t prof_every = 1, prof_tick = 0;      // level 2 sampled: every prof_every-th run is timed
    uint64_t and_total = 0;       // row-AND launches since the last stats reset, timed or not
    std::vector<EventPair> ev_and, ev_km, ev_cp, ev_pr, ev_tr, ev_ex, ev_free;
    uint64_t presence_bytes = 0;   // algorithmic bytes of the timed presence_hits calls
    uint64_t wv() const { return ceil_div(n_cols, 64); }
    uint64_t rb() const { return ceil_div(n_cols, 8); }
};

struct BusyGuard {
    bigsi_hip_index *ix;
    bool ok = true;
    // wait = true: a teardown call (bigsi_hip_batch_destroy -- finalisers run it from whichever thread drops the last reference) waits for
    // the thread that is inside the handle instead of being refused: every call is finite, and a refused destroy is a leaked batch
    explicit BusyGuard(const bigsi_hip_index *cix, bool wait = false) : ix(const_cast<bigsi_hip_index *>(cix))
    {
        if (!ix) return;
        const std::thread::id me = std::this_thread::get_id();
        for (;;) {
            std::thread::id cur{};
            if (ix->busy_owner.compare_exchange_strong(cur, me, std::memory_order_acquire)) { ix->busy_depth = 1; return; }
            if (cur == me) { ix->busy_depth++; return; }
            if (!wait) break;
            std::this_thread::yield();
        }
        ok = false;
        ix = nullptr;
    }
    void release()
    {
        if (ix && --ix->busy_depth == 0) ix->busy_owner.store(std::thread::id{}, std::memory_order_release);
        ix = nullptr;
    }
    ~BusyGuard() { release(); }
    BusyGuard(const BusyGuard &) = delete;
    BusyGuard &operator=(const BusyGuard &) = delete;
};
#define BIGSI_ENTER(ixp)                                                                                                          \
    BusyGuard busy_guard_(ixp);                                                                                                   \
    if (!busy_guard_.ok)                                                                                                          \
        return fail(BIGSI_ERR_STATE, "this index handle is in use by another host thread (one handle = one thread at a time; "  \
                                     "bigsi_hip_open_view gives every thread a handle of its own onto the same matrix)")
int bigsi_writable(const bigsi_hip_index *ix);      // BIGSI_OK, or BIGSI_ERR_STATE for an attached / view handle

struct bigsi_hip_comm;

// ------------------------------------------------------------------------------ batches
struct HitBufs {
    DevBuf chunk_hits, chunk_off, hit_off, hit_col, hit_cnt, overflow;
    // k_reads_fused: per query where its hits start in col / cnt and how many they are (no order between queries), and the two
    // allocation counters its launches use alternately
    DevBuf q_start, q_cnt, alloc;
    uint32_t gen = 0;           // launches so far: the last one used allocation counter gen & 1
    uint64_t cap = 0;   // hits the col/cnt buffers can hold
    uint32_t *xcol = nullptr, *xcnt = nullptr;   // caller-owned hit buffers (e.g. torch tensors that are then all-reduced)
    uint64_t xcap = 0;
    uint32_t *col() const { return xcol ? xcol : hit_col.as<uint32_t>(); }
    uint32_t *cnt() const { return xcnt ? xcnt : hit_cnt.as<uint32_t>(); }
    uint64_t capacity() const { return xcol ? xcap : cap; }
    void release()
    {
        chunk_hits.release(); chunk_off.release(); hit_off.release(); hit_col.release(); hit_cnt.release(); overflow.release();
        q_start.release(); q_cnt.release(); alloc.release();
    }
};

// one K5 / K6 request of a batch between its _begin and its _end (bigsi_hip.hip: presence_begin / presence_end): host vectors
// and pinned staging are kept from call to call
struct PresJob {
    bool pending = false, packed = false, device_work = false;
    uint64_t n_hits = 0, str = 0;        // hits, bytes of strings / presence bits
    size_t o_scores = 0;                 // where the score records start in the staged output
    hipEvent_t done = nullptr;
    void *h_in = nullptr, *h_out = nullptr;      // pinned
    size_t h_in_cap = 0, h_out_cap = 0;
    std::vector<uint32_t> hit_seq, hit_q, perm, order;
    std::vector<uint64_t> hit_pos0;
};

struct bigsi_hip_batch {
    bigsi_hip_index *ix = nullptr;
    uint32_t n_seqs = 0, k = 0;
    std::vector<uint64_t> seq_off, pos_off, tab_off;
    uint64_t total_pos = 0, max_pos = 0, max_len = 0;
    DevBuf seqs, d_seq_off, d_pos_off, d_tab_off, tab, first_pos, pos_unique, tmp, rows, num_kmers, num_unique, min_kmers;
    // few, larger copies (a host-side call is mostly the latency of its copies): num_kmers | num_unique | min_kmers are views
    // into `uniq` (ONE download); in a sequence batch d_seq_off | d_pos_off | d_tab_off | seqs are views into `upload`
    // (one or two uploads)
    DevBuf uniq, upload;
    std::vector<uint8_t> h_upload;
    std::vector<uint32_t> h_uniq;
    DevBuf pos_query, hsh, rep;   // per k-mer position: owning sequence, dedupe hash, class representative
    DevBuf rows_sorted;           // the row ids K2 streams: each query's list in address order (k_sort_rows)
    DevBuf bitmaps, counts, scratch;
    DevBuf planes;                // sliced counting runs: every slice's bit-sliced partial counts (k_count_combine)
    uint64_t run_serial = 0, marks_of_run = ~0ull;   // K1 runs of this batch; the run whose piece marks pres_desc holds
    const void *marks_at = nullptr;                  // ... and where (pres_desc may have been reallocated since)
    // deferred loads and exported results (the one-call and streaming entry points): pinned staging the batch owns
    void *pin_up = nullptr, *pin_out = nullptr;
    size_t pin_up_cap = 0, pin_out_cap = 0, pin_up_bytes = 0;
    bool upload_deferred = false;                    // pin_up holds tables + sequences that the next run uploads on its stream
    bool one_call = false;                           // the index's bigsi_hip_search_batch workspace: nothing else ever touches it
    bool exported_inline = false;   // the last run's read kernel wrote the export block itself (one read in a one-call search)
    bool zero_copy = false;                          // this load's tables + sequences are read by K1 straight from pin_up (no upload)
    bool idle = false;                               // nothing of this batch is in flight (its last export was collected)
    bool done_stale = false;                         // the last run did not record `done` (one-call route): wait on its stream instead
    hipEvent_t exp_done = nullptr;                   // end of the export kernel (only when the flag below is not used)
    uint64_t *pin_flag = nullptr;                    // coherent pinned word the export kernel's last workgroup writes exp_serial to
    uint64_t exp_serial = 0;                         // serial of the last export queued (0: none)
    bool exp_flagged = false;                        // the last export signals through pin_flag (the host spins on it)
    hipStream_t exp_stream = nullptr;                // the stream it was queued on
    DevBuf exp_count;                                // the export kernel's finished-workgroups counter
    uint32_t exp_spec = 0;                           // hits the export carried along speculatively
    PresJob job;                                     // the K5 / K6 request in flight, its host vectors and pinned staging
    DevBuf pres_in, pres_bits, pres_out, pres_desc;   // K5 at scale (presence_hits): host-built pair lists, presence bits, strings, piece marks
    void *ext_bitmaps = nullptr, *ext_counts = nullptr;
    HitBufs hits, ghits;
    // state of the last run
    bool ran = false, exact = false, compacted = false, sparse_counts = false;
    bool pos_query_loaded = false;    // pos_query holds this load's position -> sequence map
    hipStream_t run_stream = nullptr;   // the stream `done` was last recorded on
    bool weak_fp = false;         // BIGSI_RUN_WEAK_FINGERPRINT of the last one-launch run (a re-launch after a regrow repeats it)
    bool fused_run = false;           // the last run was the one-launch read kernel (k_reads_fused)
    bool elements = false;            // k-mers were given explicitly (bigsi_hip_batch_create_elements): K1 = k_rows_raw
    DevBuf elem_seq_off;              // elements: first element of every sequence
    uint32_t count_bytes = 2;
    double threshold = 1.0;
    uint64_t wv = 0, wv_pad = 0;   // valid / padded words per row at run time
    uint32_t run_h = 0;            // num_hashes the row ids of the last K1 were produced with
    hipEvent_t done = nullptr;     // recorded at the end of every run: fetches wait on it, not on the whole stream, so the
                                   // results of one batch can be read while the next batch's kernels are queued behind it
    hipEvent_t k1_done = nullptr;  // only when K1 runs on the pre stream: recorded after K1 (+ row sort), the index stream waits on it before K2
    hipEvent_t g_done = nullptr;   // recorded on the gather stream after a gathered compaction (it reads K1's per-query arrays)
    bool dirty = false;            // a run was started and its `done` event has not been recorded (error path): full syncs needed
    hipStream_t gstream = nullptr; // stream of the gathered compaction (null: the index's stream)
    const void *g_src = nullptr;   // last gathered buffer handed to compact_gathered
    uint32_t g_shards = 0;
    uint64_t g_shard_cols = 0;
    uint32_t g_own = 0;
    bool g_masks = false;          // the gathered buffer holds hit masks of a counting run (counts come from this rank's counters)
    std::vector<uint32_t> h_num_unique, h_num_kmers, h_min_kmers;
    bool host_counts_valid = false;
    // column-shard exchange (bigsi_shard.hip)
    uint64_t result_cols = 0;      // > 0: width of the per-sample result vectors (the group's shard_cols), so that every shard of
                                   // an index produces vectors of ONE geometry whatever its own num_cols; 0: the index's num_cols
    bigsi_hip_comm *comm = nullptr; // attached communicator (bigsi_hip_batch_set_comm)
    uint64_t shard_cols = 0;
    DevBuf gbuf;                   // [world][n_seqs][wv_pad] gathered bit vectors; this rank's slot is what K2 writes
    void *gbuf_ext = nullptr;      // loopback groups: the group's shared buffer instead of gbuf
};


// internal entry points of bigsi_hip.hip used by bigsi_shard.hip
// one-call / streaming searches: stage a load without copying (created on first use: *pb may be NULL), queue the export of a
// run's results into pinned memory, wait for it and hand the results out
int bigsi_batch_stage(bigsi_hip_index *ix, bigsi_hip_batch **pb, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k);
int bigsi_batch_run(bigsi_hip_batch *b, double threshold, uint32_t flags, bool one_call);
int bigsi_batch_export(bigsi_hip_batch *b);
// queries a row-AND launch of a large exact batch takes on this index (bigsi_batch_run's launch rule: a whole number of workgroups per
// CU, ~1600 live wavefronts): bigsi_hip_search_stream cuts its device batches at multiples of it, so that none ends in a part launch
uint32_t bigsi_exact_launch_queries(const bigsi_hip_index *ix);
int bigsi_batch_collect(bigsi_hip_batch *b, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers, uint64_t *hit_offsets,
                        uint32_t *colours, uint32_t *counts, uint64_t capacity);
int bigsi_use_device(const bigsi_hip_index *ix);
// pread / pwrite of [off, off + len) with up to `threads` host threads (bigsi_hip.hip); 0 or errno
int bigsi_file_io(int fd, bool write, uint8_t *buf, uint64_t off, uint64_t len, unsigned threads);
constexpr uint64_t kBigsiIoChunkBytes = 256ull << 20;      // one pinned buffer of the file <-> HBM pipelines
// The file side of load_rows_file / save_rows_file (single index and group).  `path` is a FILE (rows back to back from file_offset
// on) or, when it ends in '/', a DIRECTORY holding the same rows striped RAID-0 fashion over kParts part files: writes to one
// file are serialised by its inode lock and its page-cache tree -- 16 threads put 3.6-5.3 GB/s into ONE fresh tmpfs file
// (pwrite or mmap alike) and 63 GB/s into 16 files (scripts/probe/tmpfs_write_probe.py, mmap_write_probe.cpp) -- so a save that is
// to run at the PCIe rate needs several inodes.  Stripe s (rows [s * S, (s + 1) * S) of the range) lives in part s % P at
// row offset (s / P) * S; `layout` in the directory records P, S, row_bytes and the row count, and a load reads it back.
#include "bigsi_bdb.hpp"      // BigsiBdb: BerkeleyDB hash files without libdb (shared with libbigsi_cpu.so)

struct BigsiRowsFile {
    bool striped = false;
    bool bdb = false;                      // `path` is a BerkeleyDB hash file: rows are its "<row>:bitarray" records (loads only)
    BigsiBdb db;
    std::vector<BigsiBdb::Loc> loc;        // bdb: where row row0 + i lives
    uint64_t bdb_row0 = 0;
    int fd = -1;
    std::vector<int> fds;
    uint64_t parts = 0, stripe_rows = 0, row_bytes = 0, file_offset = 0;
    int open_(const char *path, bool save, uint64_t file_offset_, uint64_t row_bytes_, uint64_t n_rows, uint64_t row0 = 0, unsigned threads = 1);     // BIGSI_OK or an error (message set)
    uint64_t chunk_rows() const;                                           // rows per pinned buffer
    int io(bool write, uint8_t *buf, uint64_t rel_row, uint64_t n, unsigned threads) const;      // rows [rel_row, +n) of the range; 0 or errno
    int sync_all() const;
    void close_();
};
// profiling events (bigsi_hip_set_profiling): a pair around a group of launches on `st` (null: the index stream), collected in `dst`
int bigsi_ev_begin(bigsi_hip_index *ix, EventPair *p, hipStream_t st = nullptr, bool row_and = false);
int bigsi_ev_end(bigsi_hip_index *ix, EventPair *p, std::vector<EventPair> &dst, hipStream_t st = nullptr, uint32_t launches = 1);
// write pass of the gathered compaction again after the hit buffers grew (fetch_gathered_hits); defined in bigsi_shard.hip
int bigsi_reduce_gathered_counts(bigsi_hip_batch *b);
