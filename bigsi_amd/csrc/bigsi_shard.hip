// bigsi_shard.hip -- column shards of one index: the RCCL exchange, device groups, and the one-call search
// (include/bigsi_hip.h, sections "one-call search" and "column shards").  Host code only, on top of the single-shard entry
// points of bigsi_hip.hip; the one kernel here sums per-hit count arrays of shards that share a device.
#include "bigsi_internal.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is loaded at run time

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <set>

// ------------------------------------------------------------------------------ RCCL, loaded on first use
// dlopen by SONAME: a process that already holds a librccl.so.1 (torch bundles one) gets that copy, anything else finds
// /opt/rocm's through the library's RUNPATH.  Linking it at build time would pull a second HIP runtime in under torch.
namespace {
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char err[256] = "";
};

RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(api.err, sizeof api.err, "%s", dlerror());
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(h, name);
        if (!p) {
            ok = false;
            snprintf(api.err, sizeof api.err, "librccl has no symbol %s", name);
        }
        return p;
    };
#define RCCL_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(sym(name))
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(CommCount, "ncclCommCount");
    RCCL_SYM(CommUserRank, "ncclCommUserRank");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    if (!ok) {
        dlclose(h);
        return nullptr;
    }
    api.handle = h;
    return &api;
}

int need_rccl(RcclApi **out)
{
    RcclApi *r = rccl();
    if (!r) return fail(BIGSI_ERR_STATE, "RCCL is not available (dlopen of librccl.so.1 failed or it lacks a symbol)");
    *out = r;
    return BIGSI_OK;
}
}   // namespace

#define NCCL_TRY(api, expr)                                                                                                 \
    do {                                                                                                                    \
        ncclResult_t r_ = (expr);                                                                                           \
        if (r_ != ncclSuccess)                                                                                              \
            return fail(BIGSI_ERR_HIP, "%s:%d %s: %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(r_));            \
    } while (0)

// ------------------------------------------------------------------------------ one-call search
// The workspace stages tables + sequences in pinned memory; the run uploads them on its own stream, ahead of its first kernel;
// a small kernel exports what the caller gets back into pinned memory; the host waits ONCE.  (Round 2: two synchronous uploads
// and three synchronous downloads -- 77-84 us for one read, 138 us for 1000 reads of 61 bp, of which the kernels were 10-25.)
extern "C" int bigsi_hip_search_batch(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                                      double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                      uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity)
{
    BIGSI_ENTER(ix);
    if (!hit_offsets) return fail(BIGSI_ERR_INVALID, "hit_offsets is NULL");
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    // the index keeps ONE workspace for this entry point (a caller in a loop pays its ~20 device allocations once; the
    // workspace goes with the index, or right away if a call fails)
    CALL_MARK(0);
    if (ix->search_ws) ix->search_ws->one_call = true;
    TRY(bigsi_batch_stage(ix, &ix->search_ws, seqs, offsets, n_seqs, k));
    CALL_MARK(1);
    bigsi_hip_batch *b = ix->search_ws;
    b->one_call = true;      // (a small input is then read by K1 straight from the pinned staging, and the run records no event)
    int rc = bigsi_batch_run(b, threshold, (flags & ~BIGSI_RUN_SKIP_COMPACT) | BIGSI_RUN_SPARSE_COUNTS, true);
    CALL_MARK(2);
    if (rc == BIGSI_OK) rc = bigsi_batch_export(b);
    CALL_MARK(3);
    if (rc == BIGSI_OK) rc = bigsi_batch_collect(b, num_kmers, num_unique, min_kmers, hit_offsets, colours, counts, hit_capacity);
    CALL_MARK(5);
    if (rc != BIGSI_OK && rc != BIGSI_ERR_CAPACITY) {      // (a too small hit buffer is the caller's to retry: offsets are filled in)
        ix->search_ws = nullptr;
        bigsi_hip_batch_destroy(b);      // leaves the thread's error message of the failed call above in place
    }
    return rc;
}

// BIGSI.search for ANY number of sequences in one call (bulk_search, bigsi/__main__.py:261-314, pays per query what this pays
// per call): the library cuts the input into device batches of about 2^20 k-mer positions, keeps three workspaces in flight --
// while one batch runs, the next is staged and uploaded and the results of the one before are exported and copied out -- and
// writes every sequence's results at its place in the caller's arrays (hit_offsets are global: n_seqs + 1 entries).
// BIGSI_ERR_CAPACITY (hit_offsets complete, colours / counts filled as far as they fit) when hit_capacity is too small.
//
// With `so` (score=True): a chunk's hit lists are collected one chunk after its launch instead of two, its presence bits and
// score records are requested at once (K5 + K6 on the score stream, beside the row-AND kernels of the chunk just launched)
// and copied out one chunk later, just before its workspace is staged again.
namespace {
struct ScoredOut {
    uint8_t *bits;
    uint64_t bits_capacity;
    uint64_t *bit_offsets;            // hit_capacity + 1 entries
    bigsi_hip_hit_score *scores;      // hit_capacity entries
    uint64_t *bits_needed;
};

int search_stream_impl(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, double threshold,
                       uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers, uint64_t *hit_offsets,
                       uint32_t *colours, uint32_t *counts, uint64_t hit_capacity, const ScoredOut *so)
{
    if (!ix || !offsets || !hit_offsets) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    if (so && (!so->bit_offsets || !so->bits_needed)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    hit_offsets[0] = 0;
    if (so) { so->bit_offsets[0] = 0; *so->bits_needed = 0; }
    if (n_seqs == 0) return BIGSI_OK;
    constexpr int kMaxSlots = 6;
    // device batches in flight (1 M reads of 61 bp, host-visible, interleaved: 2 -> 1.14, 3 -> 1.47-1.50, 4 -> 1.51-1.53, 6 -> 1.49-1.51 G
    // lookups/s; gene-length and scored streams +-0)
    static const int kSlots = std::min(std::max(env_int("BIGSI_HIP_STREAM_SLOTS", 4), 2), kMaxSlots);
    // a batch = at most 2^20 k-mer positions (gene-length queries: ~1000 of 1 kbp) and at most kChunkSeqs sequences (reads).  With the
    // read kernel that ordered its hit lists inside the launch (rounds 2-3) 2^14 reads per launch measured best; the wait-free kernel
    // of round 4 prefers smaller launches, more of them in flight: host-visible over 1 M reads of 61 bp 1.04 / 1.41 / 1.51 / 1.45 G
    // lookups/s at 1024 / 2048 / 4096 / 16384 reads per batch, over 64 k reads 1.01 / 1.36 / 1.39 / 1.27 G (interleaved, twice each)
    // thresholded searches of gene-length queries take four times that: the counting kernel is ONE launch per chunk (chunked it measured
    // -4 ... 0 %, bigsi_batch_run), and every launch pays its tail -- 8192 x 1 kbp at 0.4 on 10 M x 100 k: 8 launches of ~1080 queries
    // 0.774 of peak by the kernel's clock, one launch of 8192 0.80 (round 6); the counters of a chunk are 2 bytes x samples per query
    const uint64_t kChunkPositions = (threshold == 1.0 || so) ? 1ull << 20 : 1ull << 22;      // (scored: K5 + K6 of a chunk run beside the next chunk's row-AND -- more, smaller chunks)
    static const int chunk_seqs_env = env_int("BIGSI_HIP_STREAM_SEQS", 0);
    const uint64_t kChunkSeqs = chunk_seqs_env > 0 ? (uint64_t)chunk_seqs_env : 4096;
    const uint32_t launch_q = bigsi_exact_launch_queries(ix);
    struct Chunk { uint64_t first; uint32_t n; uint64_t hit0, bit0; bool scoring; };
    Chunk inflight[kMaxSlots] = {};
    bool busy[kMaxSlots] = {};       // launched, hit lists not collected yet
    uint64_t total = 0;              // hits so far (global offset of the next chunk's first hit)
    uint64_t bits_total = 0;         // bytes of presence bits so far
    bool overflow = false, bits_overflow = false;
    std::vector<uint64_t> rel, rel_bits;
    auto collect = [&](int s) -> int {
        Chunk &c = inflight[s];
        busy[s] = false;
        rel.resize(c.n + 1ull);
        const uint64_t room = total < hit_capacity ? hit_capacity - total : 0;
        uint32_t *nk = num_kmers ? num_kmers + c.first : nullptr;
        int rc = bigsi_batch_collect(ix->stream_ws[s], nk, num_unique ? num_unique + c.first : nullptr,
                                     min_kmers ? min_kmers + c.first : nullptr, rel.data(), colours ? colours + total : nullptr,
                                     counts ? counts + total : nullptr, overflow ? 0 : room);
        if (rc == BIGSI_ERR_CAPACITY) { overflow = true; rc = BIGSI_OK; }
        if (rc != BIGSI_OK) return rc;
        for (uint32_t i = 1; i <= c.n; i++) hit_offsets[c.first + i] = total + rel[i];
        c.hit0 = total;
        c.bit0 = bits_total;
        c.scoring = false;
        if (so) {
            if (!overflow && rel[c.n]) {
                // the request: this chunk's hit lists as they lie in the caller's arrays
                rel_bits.resize(rel[c.n] + 1);
                rc = bigsi_hip_batch_score_hits_begin(ix->stream_ws[s], rel.data(), colours + total, counts ? counts + total : nullptr, 0, rel_bits.data());
                if (rc != BIGSI_OK) return rc;
                c.scoring = true;
                for (uint64_t t = 0; t <= rel[c.n]; t++) so->bit_offsets[total + t] = bits_total + rel_bits[t];
                bits_total += rel_bits[rel[c.n]];
            } else if (rel[c.n]) {
                // the lists did not fit: only the bytes their strings would need (whole 8-byte words per string)
                uint32_t tmp = 0;
                for (uint32_t i = 0; i < c.n; i++) {
                    const uint64_t len = offsets[c.first + i + 1] - offsets[c.first + i];
                    tmp = len >= k ? (uint32_t)(len - k + 1) : 0;
                    bits_total += (rel[i + 1] - rel[i]) * (((uint64_t)tmp + 63) / 64 * 8);
                }
            }
        }
        total += rel[c.n];
        return BIGSI_OK;
    };
    auto finish_score = [&](int s) -> int {
        Chunk &c = inflight[s];
        if (!c.scoring) return BIGSI_OK;
        c.scoring = false;
        const uint64_t need = so->bit_offsets[c.hit0 + (hit_offsets[c.first + c.n] - c.hit0)] - c.bit0;
        if (c.bit0 + need > so->bits_capacity || !so->bits) {
            // no room for this chunk's bits: finish the request into nothing (the batch has to be free for its next run)
            bits_overflow = true;
            std::vector<uint8_t> sink_bits(need + 8);
            std::vector<bigsi_hip_hit_score> sink(hit_offsets[c.first + c.n] - c.hit0);
            return bigsi_hip_batch_score_hits_end(ix->stream_ws[s], sink_bits.data(), need, sink.data());
        }
        return bigsi_hip_batch_score_hits_end(ix->stream_ws[s], so->bits + c.bit0, need, so->scores + c.hit0);
    };
    int rc = BIGSI_OK;
    uint64_t next = 0;
    int slot = 0;
    while (next < n_seqs && rc == BIGSI_OK) {
        // the next chunk: up to kChunkPositions k-mer positions, at least one sequence
        uint64_t end = next, pos = 0;
        while (end < n_seqs && end - next < kChunkSeqs) {
            if (offsets[end + 1] < offsets[end]) { rc = fail(BIGSI_ERR_INVALID, "offsets must be non-decreasing"); break; }
            const uint64_t len = offsets[end + 1] - offsets[end], n = len >= k ? len - k + 1 : 0;
            if (end > next && pos + n > kChunkPositions) break;
            pos += std::max<uint64_t>(n, 1);
            end++;
        }
        if (rc != BIGSI_OK) break;
        // an exact chunk of gene-length queries goes out as several row-AND launches of launch_q queries each: end it on a multiple of
        // that, so that no chunk closes with a part launch (10 M x 100 k, 8192 x 1 kbp per call: 8 chunks x 8 launches of 128 queries
        // instead of 8 x (8 + a launch of 57); round 6)
        if (threshold == 1.0 && end < n_seqs && end - next >= 2ull * launch_q) end = next + (end - next) / launch_q * launch_q;
        if (busy[slot]) rc = collect(slot);                    // this workspace's previous chunk (three chunks ago)
        if (rc == BIGSI_OK && so) rc = finish_score(slot);
        if (rc != BIGSI_OK) break;
        rc = bigsi_batch_stage(ix, &ix->stream_ws[slot], seqs, offsets + next, (uint32_t)(end - next), k);
        if (rc == BIGSI_OK) rc = bigsi_hip_batch_run(ix->stream_ws[slot], threshold, (flags & ~BIGSI_RUN_SKIP_COMPACT) | BIGSI_RUN_SPARSE_COUNTS);
        if (rc == BIGSI_OK) rc = bigsi_batch_export(ix->stream_ws[slot]);
        if (rc != BIGSI_OK) break;
        inflight[slot] = Chunk{next, (uint32_t)(end - next), 0, 0, false};
        busy[slot] = true;
        next = end;
        if (so) {
            // the chunk before this one: hit lists out, scores requested (they run beside the chunk just launched)
            const int prev = (slot + kSlots - 1) % kSlots;
            if (busy[prev]) rc = collect(prev);
        }
        slot = (slot + 1) % kSlots;
    }
    // drain in submission order
    for (int i = 0; i < kSlots && rc == BIGSI_OK; i++) {
        const int s2 = (slot + i) % kSlots;
        if (busy[s2]) rc = collect(s2);
    }
    for (int i = 0; i < kSlots && rc == BIGSI_OK && so; i++) rc = finish_score((slot + i) % kSlots);
    if (rc != BIGSI_OK) {
        for (auto &w : ix->stream_ws)
            if (w) { bigsi_hip_batch_destroy(w); w = nullptr; }
        return rc;
    }
    if (so) *so->bits_needed = bits_total;
    if (overflow) return fail(BIGSI_ERR_CAPACITY, "hit buffers hold %llu entries, %llu needed", (unsigned long long)hit_capacity, (unsigned long long)total);
    if (bits_overflow) return fail(BIGSI_ERR_CAPACITY, "bit buffer holds %llu bytes, %llu needed", (unsigned long long)so->bits_capacity, (unsigned long long)bits_total);
    return BIGSI_OK;
}
}   // namespace

extern "C" int bigsi_hip_search_stream(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                                       double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                       uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity)
{
    BIGSI_ENTER(ix);
    return search_stream_impl(ix, seqs, offsets, n_seqs, k, threshold, flags, num_kmers, num_unique, min_kmers, hit_offsets, colours, counts,
                              hit_capacity, nullptr);
}

// BIGSI.search(..., score=True) for any number of sequences in one call: as above, plus for every hit t (global index into
// colours) its presence bits at bits + bit_offsets[t] (the layout of bigsi_hip_batch_score_hits) and its score record.
extern "C" int bigsi_hip_search_stream_scored(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                                              double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique,
                                              uint32_t *min_kmers, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts,
                                              uint64_t hit_capacity, uint8_t *bits, uint64_t bits_capacity, uint64_t *bit_offsets,
                                              bigsi_hip_hit_score *scores, uint64_t *bits_needed)
{
    BIGSI_ENTER(ix);
    if (hit_capacity && (!colours || !scores)) return fail(BIGSI_ERR_INVALID, "colours / scores is NULL");
    uint64_t needed = 0;
    const ScoredOut so{bits, bits_capacity, bit_offsets, scores, bits_needed ? bits_needed : &needed};
    return search_stream_impl(ix, seqs, offsets, n_seqs, k, threshold, flags, num_kmers, num_unique, min_kmers, hit_offsets, colours, counts,
                              hit_capacity, &so);
}

// ------------------------------------------------------------------------------ communicators
struct bigsi_hip_comm {
    ncclComm_t comm = nullptr;          // null: member of a group whose shards share a device (no RCCL)
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;       // collectives and gathered compaction run here
    bigsi_hip_group *group = nullptr;   // non-null: collectives are issued by the group for all its members at once
};

static int comm_alloc(int device, int rank, int world, bigsi_hip_comm **out)
{
    HIP_TRY(hipSetDevice(device));
    bigsi_hip_comm *c = new (std::nothrow) bigsi_hip_comm();
    if (!c) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    c->rank = rank;
    c->world = world;
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(BIGSI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_comm_unique_id(uint8_t *id)
{
    if (!id) return fail(BIGSI_ERR_INVALID, "id is NULL");
    RcclApi *api = nullptr;
    TRY(need_rccl(&api));
    static_assert(sizeof(ncclUniqueId) == BIGSI_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCL_TRY(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return BIGSI_OK;
}

extern "C" int bigsi_hip_comm_init_rank(int device, const uint8_t *id, int rank, int world, bigsi_hip_comm **out)
{
    if (!out || !id) return fail(BIGSI_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(BIGSI_ERR_INVALID, "rank %d not in [0,%d)", rank, world);
    RcclApi *api = nullptr;
    TRY(need_rccl(&api));
    bigsi_hip_comm *c = nullptr;
    TRY(comm_alloc(device, rank, world, &c));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = api->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        hipError_t e = hipStreamDestroy(c->stream); (void)e;
        delete c;
        return fail(BIGSI_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, api->GetErrorString(r));
    }
    *out = c;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_comm_destroy(bigsi_hip_comm *c)
{
    if (!c) return BIGSI_OK;
    hipError_t e = hipSetDevice(c->device);
    if (c->stream) e = hipStreamSynchronize(c->stream);
    if (c->comm) {
        RcclApi *api = rccl();
        if (api) api->CommDestroy(c->comm);
    }
    if (c->stream) e = hipStreamDestroy(c->stream);
    (void)e;
    delete c;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_comm_info(const bigsi_hip_comm *c, int *rank, int *world)
{
    if (!c) return fail(BIGSI_ERR_INVALID, "NULL communicator");
    int r = c->rank, w = c->world;
    if (c->comm) {
        RcclApi *api = nullptr;
        TRY(need_rccl(&api));
        NCCL_TRY(api, api->CommCount(c->comm, &w));
        NCCL_TRY(api, api->CommUserRank(c->comm, &r));
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ a batch on one shard of a sharded index
extern "C" int bigsi_hip_batch_set_comm(bigsi_hip_batch *b, bigsi_hip_comm *c, uint64_t shard_cols)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!c) {
        if (b->comm) {      // an exchange of this batch may still be in flight on the communicator's stream
            TRY(bigsi_use_device(b->ix));
            if (b->g_done) HIP_TRY(hipEventSynchronize(b->g_done));
            HIP_TRY(hipStreamSynchronize(b->comm->stream));
        }
        b->comm = nullptr;
        b->shard_cols = b->result_cols = 0;
        b->ext_bitmaps = nullptr;
        b->gstream = nullptr;
        return BIGSI_OK;
    }
    bigsi_hip_index *ix = b->ix;
    if (c->device != ix->device) return fail(BIGSI_ERR_INVALID, "communicator is on device %d, the index on device %d", c->device, ix->device);
    if (shard_cols == 0 || shard_cols < ix->n_cols)
        return fail(BIGSI_ERR_INVALID, "shard_cols %llu is smaller than this shard's %llu columns", (unsigned long long)shard_cols,
                    (unsigned long long)ix->n_cols);
    if ((uint64_t)c->world * shard_cols > 0xFFFFFFFFull) return fail(BIGSI_ERR_INVALID, "more than 2^32-1 colours in total");
    TRY(bigsi_hip_reserve_cols(ix, shard_cols));          // no-op when the stride already covers it
    TRY(bigsi_hip_batch_set_result_cols(b, shard_cols));
    b->comm = c;
    b->shard_cols = shard_cols;
    b->gstream = c->stream;
    return BIGSI_OK;
}

// bytes of one shard's slot of the gather buffer for the batch's current load
static uint64_t slot_bytes(const bigsi_hip_batch *b)
{
    const uint64_t wv = ceil_div(std::max(b->ix->n_cols, b->result_cols), 64);
    return (uint64_t)b->n_seqs * round_up(wv, 2) * 8;
}

// point K2's output at this rank's slot (growing the library-owned buffer first if the load needs more)
static int prepare_slot(bigsi_hip_batch *b, void *shared /* loopback groups: one buffer for all members */)
{
    const bigsi_hip_comm *c = b->comm;
    const uint64_t per = slot_bytes(b);
    uint8_t *g = (uint8_t *)shared;
    if (!g) {
        if (b->gbuf.cap < per * c->world) {
            // the previous exchange of this batch may still be reading the old buffer
            if (b->done) HIP_TRY(hipEventSynchronize(b->done));
            if (b->g_done) HIP_TRY(hipEventSynchronize(b->g_done));
            TRY(b->gbuf.reserve(per * c->world));
        }
        g = b->gbuf.as<uint8_t>();
    }
    b->gbuf_ext = shared;
    b->ext_bitmaps = g + per * c->rank;
    return BIGSI_OK;
}

static uint8_t *gather_base(const bigsi_hip_batch *b) { return b->gbuf_ext ? (uint8_t *)b->gbuf_ext : b->gbuf.as<uint8_t>(); }

// after the collective: compaction of the gathered vectors on the communicator's stream (already queued behind the run)
static int compact_after_gather(bigsi_hip_batch *b)
{
    const bigsi_hip_comm *c = b->comm;
    if (b->exact) return bigsi_hip_batch_compact_gathered(b, gather_base(b), (uint32_t)c->world, b->shard_cols);
    return bigsi_hip_batch_compact_gathered_masks(b, gather_base(b), (uint32_t)c->world, b->shard_cols, (uint32_t)c->rank);
}

// thresholded searches: every rank filled in the counts of the hits of its own shard, zero elsewhere -> sum over ranks.
// Fixed size (the capacity of the hit buffers, identical on every rank: it only ever grows with the identical totals).
int bigsi_reduce_gathered_counts(bigsi_hip_batch *b)
{
    bigsi_hip_comm *c = b->comm;
    if (!c || c->group || !c->comm) return BIGSI_OK;      // groups reduce all their members at once (group_reduce_counts)
    RcclApi *api = nullptr;
    TRY(need_rccl(&api));
    NCCL_TRY(api, api->AllReduce(b->ghits.cnt(), b->ghits.cnt(), b->ghits.capacity(), ncclUint32, ncclSum, c->comm, c->stream));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_run_sharded(bigsi_hip_batch *b, double threshold, uint32_t flags)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    bigsi_hip_comm *c = b->comm;
    if (!c) return fail(BIGSI_ERR_STATE, "no communicator attached (bigsi_hip_batch_set_comm)");
    if (c->group) return fail(BIGSI_ERR_STATE, "this batch belongs to a device group: use bigsi_hip_group_batch_run");
    TRY(bigsi_use_device(b->ix));
    TRY(prepare_slot(b, nullptr));
    TRY(bigsi_hip_batch_run(b, threshold, flags | BIGSI_RUN_SKIP_COMPACT | BIGSI_RUN_SPARSE_COUNTS));
    RcclApi *api = nullptr;
    TRY(need_rccl(&api));
    HIP_TRY(hipStreamWaitEvent(c->stream, b->done, 0));
    // the exchange as the communicator's stream sees it (bigsi_hip_stats: exchange_ms): all-gather + gathered compaction +
    // count all-reduce, from the moment this batch's kernels are over
    EventPair xe{};
    TRY(bigsi_ev_begin(b->ix, &xe, c->stream));
    // in place: this rank's slot is where its kernels wrote
    NCCL_TRY(api, api->AllGather(b->ext_bitmaps, gather_base(b), slot_bytes(b), ncclUint8, c->comm, c->stream));
    TRY(compact_after_gather(b));
    if (!b->exact) TRY(bigsi_reduce_gathered_counts(b));
    TRY(bigsi_ev_end(b->ix, &xe, b->ix->ev_ex, c->stream));
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ device groups (one process, several GPUs)
struct bigsi_hip_group {
    std::vector<bigsi_hip_index *> ix;
    std::vector<bigsi_hip_comm *> comm;
    uint64_t m = 0, n_cols = 0, cap_cols = 0, shard_cols = 0;
    uint32_t h = 0;
    bool rccl = false;
    bigsi_hip_group_batch *search_ws = nullptr;      // bigsi_hip_group_search_batch's workspace
    uint32_t n() const { return (uint32_t)ix.size(); }
    // columns of the whole index that live on shard i, for a given total
    uint64_t cols_of(uint32_t i, uint64_t total) const
    {
        const uint64_t lo = (uint64_t)i * shard_cols;
        return total <= lo ? 0 : std::min(shard_cols, total - lo);
    }
};

struct bigsi_hip_group_batch {
    bigsi_hip_group *g = nullptr;
    std::vector<bigsi_hip_batch *> b;
    DevBuf shared;                       // loopback groups: the one gather buffer all members write their slot of
    bool ran = false;
    uint64_t reduced_cap = 0;            // hit-buffer capacity the last count reduction covered
};

__global__ void k_add_counts(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

static int group_open_impl(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes, const int *device_ids, int n_dev,
                           const uint8_t *ipc_handles, bigsi_hip_group **out);

extern "C" int bigsi_hip_group_open(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                                    const int *device_ids, int n_dev, bigsi_hip_group **out)
{
    return group_open_impl(num_rows, num_cols, col_capacity, num_hashes, device_ids, n_dev, nullptr, out);
}

// A group onto matrices ANOTHER process holds resident (bigsi_hip_export_ipc / bigsi_hip_open_ipc, shard by shard): read-only, no copy.
extern "C" int bigsi_hip_group_export_ipc(bigsi_hip_group *g, uint8_t *handles)
{
    if (!g || !handles) return fail(BIGSI_ERR_INVALID, "NULL argument");
    for (uint32_t i = 0; i < g->n(); i++) TRY(bigsi_hip_export_ipc(g->ix[i], handles + (size_t)i * BIGSI_IPC_HANDLE_BYTES));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_open_ipc(const uint8_t *handles, uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                                        const int *device_ids, int n_dev, bigsi_hip_group **out)
{
    if (!handles) return fail(BIGSI_ERR_INVALID, "handles is NULL");
    return group_open_impl(num_rows, num_cols, col_capacity, num_hashes, device_ids, n_dev, handles, out);
}

static int group_open_impl(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes, const int *device_ids, int n_dev,
                           const uint8_t *ipc_handles, bigsi_hip_group **out)
{
    if (!out || !device_ids) return fail(BIGSI_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 64) return fail(BIGSI_ERR_INVALID, "n_dev %d not in [1,64]", n_dev);
    if (col_capacity < num_cols) col_capacity = num_cols;
    if (col_capacity == 0) col_capacity = 64 * (uint64_t)n_dev;
    bigsi_hip_group *g = new (std::nothrow) bigsi_hip_group();
    if (!g) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    g->m = num_rows;
    g->h = num_hashes;
    g->shard_cols = round_up(ceil_div(col_capacity, (uint64_t)n_dev), 64);
    g->cap_cols = g->shard_cols * (uint64_t)n_dev;
    g->n_cols = num_cols;
    if (g->cap_cols > 0xFFFFFFFFull) {
        delete g;
        return fail(BIGSI_ERR_INVALID, "more than 2^32-1 colours in total");
    }
    std::set<int> distinct(device_ids, device_ids + n_dev);
    g->rccl = (int)distinct.size() == n_dev;
    int rc = BIGSI_OK;
    for (int i = 0; i < n_dev && rc == BIGSI_OK; i++) {
        bigsi_hip_index *ix = nullptr;
        if (ipc_handles) rc = bigsi_hip_open_ipc(ipc_handles + (size_t)i * BIGSI_IPC_HANDLE_BYTES, num_rows, g->cols_of((uint32_t)i, num_cols), g->shard_cols, num_hashes, device_ids[i], &ix);
        else rc = bigsi_hip_open(num_rows, g->cols_of((uint32_t)i, num_cols), g->shard_cols, num_hashes, device_ids[i], &ix);
        if (rc == BIGSI_OK) g->ix.push_back(ix);
    }
    for (int i = 0; i < n_dev && rc == BIGSI_OK; i++) {
        bigsi_hip_comm *c = nullptr;
        rc = comm_alloc(device_ids[i], i, n_dev, &c);
        if (rc == BIGSI_OK) {
            c->group = g;
            g->comm.push_back(c);
        }
    }
    if (rc == BIGSI_OK && g->rccl) {
        RcclApi *api = nullptr;
        rc = need_rccl(&api);
        if (rc == BIGSI_OK) {
            std::vector<ncclComm_t> comms(n_dev);
            ncclResult_t r = api->CommInitAll(comms.data(), n_dev, device_ids);
            if (r != ncclSuccess) rc = fail(BIGSI_ERR_HIP, "ncclCommInitAll over %d devices: %s", n_dev, api->GetErrorString(r));
            else
                for (int i = 0; i < n_dev; i++) g->comm[i]->comm = comms[i];
        }
    }
    if (rc != BIGSI_OK) {
        char keep[1024];
        snprintf(keep, sizeof keep, "%s", bigsi_hip_last_error());
        bigsi_hip_group_close(g);
        return fail(rc, "%s", keep);
    }
    *out = g;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_close(bigsi_hip_group *g)
{
    if (!g) return BIGSI_OK;
    if (g->search_ws) bigsi_hip_group_batch_destroy(g->search_ws);
    for (auto *c : g->comm) bigsi_hip_comm_destroy(c);
    for (auto *ix : g->ix) bigsi_hip_close(ix);
    delete g;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_get_info(const bigsi_hip_group *g, bigsi_hip_group_info *out)
{
    if (!g || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    memset(out, 0, sizeof *out);
    out->num_rows = g->m;
    out->num_cols = g->n_cols;
    out->col_capacity = g->cap_cols;
    out->shard_cols = g->shard_cols;
    out->row_bytes = ceil_div(g->n_cols, 8);
    out->num_hashes = g->h;
    out->n_shards = g->n();
    out->rccl = g->rccl ? 1 : 0;
    for (auto *ix : g->ix) {
        bigsi_hip_info inf;
        TRY(bigsi_hip_get_info(ix, &inf));
        out->index_bytes += inf.index_bytes;
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_shard(bigsi_hip_group *g, uint32_t i, bigsi_hip_index **out)
{
    if (!g || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (i >= g->n()) return fail(BIGSI_ERR_RANGE, "shard %u not in [0,%u)", i, g->n());
    *out = g->ix[i];
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_set_num_cols(bigsi_hip_group *g, uint64_t num_cols)
{
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    if (num_cols > g->cap_cols)
        return fail(BIGSI_ERR_CAPACITY, "num_cols %llu exceeds the group's col_capacity %llu (fixed when the group was opened)",
                    (unsigned long long)num_cols, (unsigned long long)g->cap_cols);
    for (uint32_t i = 0; i < g->n(); i++) TRY(bigsi_hip_set_num_cols(g->ix[i], g->cols_of(i, num_cols)));
    g->n_cols = num_cols;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_set_num_hashes(bigsi_hip_group *g, uint32_t num_hashes)
{
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    for (auto *ix : g->ix) TRY(bigsi_hip_set_num_hashes(ix, num_hashes));
    g->h = num_hashes;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_synchronize(bigsi_hip_group *g)
{
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    for (uint32_t i = 0; i < g->n(); i++) {
        TRY(bigsi_hip_synchronize(g->ix[i]));
        HIP_TRY(hipStreamSynchronize(g->comm[i]->stream));
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_clear(bigsi_hip_group *g)
{
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    for (auto *ix : g->ix) TRY(bigsi_hip_clear(ix));
    return BIGSI_OK;
}

// whole rows <-> per-shard rows: shard_cols is a multiple of 64, so shard i's bytes are [i * shard_cols / 8, ...) of a row
extern "C" int bigsi_hip_group_set_rows(bigsi_hip_group *g, const uint64_t *row_ids, uint64_t n, const uint8_t *bytes, uint64_t row_bytes)
{
    if (!g || (n && (!row_ids || !bytes))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (row_bytes == 0 || row_bytes > g->cap_cols / 8)
        return fail(BIGSI_ERR_CAPACITY, "row_bytes %llu not in [1, %llu]", (unsigned long long)row_bytes, (unsigned long long)(g->cap_cols / 8));
    const uint64_t sb = g->shard_cols / 8;
    std::vector<uint8_t> part;
    for (uint32_t i = 0; i < g->n(); i++) {
        const uint64_t lo = (uint64_t)i * sb;
        // shards beyond the end of the given bytes are zeroed: bytes past row_bytes read as zero (bigsi_hip_set_rows)
        const uint64_t w = row_bytes > lo ? std::min(sb, row_bytes - lo) : 0;
        const uint64_t pw = std::max<uint64_t>(w, 1);
        part.assign(n * pw, 0);
        if (w)
            for (uint64_t r = 0; r < n; r++) memcpy(part.data() + r * pw, bytes + r * row_bytes + lo, w);
        TRY(bigsi_hip_set_rows(g->ix[i], row_ids, n, part.data(), pw));
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_get_rows(bigsi_hip_group *g, const uint64_t *row_ids, uint64_t n, uint8_t *out, uint64_t row_bytes)
{
    if (!g || (n && (!row_ids || !out))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (row_bytes == 0) return fail(BIGSI_ERR_INVALID, "row_bytes is 0");
    const uint64_t sb = g->shard_cols / 8;
    memset(out, 0, n * row_bytes);
    std::vector<uint8_t> part;
    for (uint32_t i = 0; i < g->n(); i++) {
        const uint64_t lo = (uint64_t)i * sb;
        if (row_bytes <= lo) break;
        const uint64_t w = std::min(sb, row_bytes - lo);
        part.resize(n * w);
        TRY(bigsi_hip_get_rows(g->ix[i], row_ids, n, part.data(), w));
        for (uint64_t r = 0; r < n; r++) memcpy(out + r * row_bytes + lo, part.data() + r * w, w);
    }
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ a row range between ONE file and all shards
// KmerSignatureIndex.create / BerkeleyDBStorage (bigsi/graph/index.py:27-40, bigsi/storage/berkeleydb.py:6-19) for an index that is
// spread over several GPUs: the file holds WHOLE rows (row_bytes each, the reference's row format: what a snapshot, a converted
// v0.3 store or the single-GPU snapshot of the same index holds); shard i owns bytes [i * shard_cols / 8, +shard_cols / 8) of every
// row.  One pinned double buffer for the group (portable: every device's DMA engine reads it); host threads on the file while the
// other buffer is in flight; every chunk goes out as ONE two-dimensional copy per shard (source pitch = row_bytes, destination
// pitch = the shard's row stride) on that shard's own stream -- the rows are cut at the shard boundaries by the copy engines, not
// by the CPU, and all devices' PCIe links carry their share of a chunk at the same time.
static int group_rows_file(bigsi_hip_group *g, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes, uint32_t threads,
                           bool save, bigsi_hip_io_stats *st)
{
    if (!g || !path) return fail(BIGSI_ERR_INVALID, "NULL argument");
    const uint64_t sb = g->shard_cols / 8;
    // (a LOAD may name rows longer than the group holds -- the 128-byte-multiple pitch of a single-GPU snapshot of the same index:
    // the bytes past the last shard are padding and are skipped)
    if (row_bytes == 0 || (save && row_bytes > sb * g->n()))
        return fail(BIGSI_ERR_CAPACITY, "row_bytes %llu not in [1, %llu] (the group's column capacity)", (unsigned long long)row_bytes, (unsigned long long)(sb * g->n()));
    if (row0 > g->m || n_rows > g->m - row0) return fail(BIGSI_ERR_RANGE, "rows [%llu, +%llu) outside [0, %llu)", (unsigned long long)row0, (unsigned long long)n_rows, (unsigned long long)g->m);
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 4));
    const uint32_t used = (uint32_t)std::min<uint64_t>(ceil_div(row_bytes, sb), g->n());          // shards that hold bytes of these rows
    for (uint32_t i = 0; i < g->n(); i++) {
        if (!save) TRY(bigsi_writable(g->ix[i]));
        TRY(bigsi_hip_synchronize(g->ix[i]));
    }
    BigsiRowsFile rf;
    TRY(rf.open_(path, save, file_offset, row_bytes, n_rows, row0, threads));
    const uint64_t per = rf.chunk_rows();
    void *pin[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev[2];
    const auto t_begin = std::chrono::steady_clock::now();
    double io_s = 0;
    auto wait_slot = [&](int slot) -> int {
        for (uint32_t i = 0; i < used; i++) {
            HIP_TRY(hipSetDevice(g->ix[i]->device));
            HIP_TRY(hipEventSynchronize(ev[slot][i]));
        }
        return BIGSI_OK;
    };
    auto copies = [&](int slot, uint64_t r0, uint64_t cn) -> int {      // chunk [r0, r0 + cn) between pin[slot] and every shard
        for (uint32_t i = 0; i < used; i++) {
            bigsi_hip_index *ix = g->ix[i];
            const uint64_t lo = (uint64_t)i * sb, w = std::min(sb, row_bytes - lo), stride = ix->stride_words * 8;
            uint8_t *dev = reinterpret_cast<uint8_t *>(ix->d_index) + r0 * stride, *host = static_cast<uint8_t *>(pin[slot]) + lo;
            HIP_TRY(hipSetDevice(ix->device));
            if (save) HIP_TRY(hipMemcpy2DAsync(host, row_bytes, dev, stride, w, cn, hipMemcpyDeviceToHost, ix->stream));
            else HIP_TRY(hipMemcpy2DAsync(dev, stride, host, row_bytes, w, cn, hipMemcpyHostToDevice, ix->stream));
            HIP_TRY(hipEventRecord(ev[slot][i], ix->stream));
        }
        return BIGSI_OK;
    };
    auto body = [&]() -> int {
        for (int s = 0; s < 2; s++) {
            HIP_TRY(hipHostMalloc(&pin[s], std::min(per, std::max<uint64_t>(n_rows, 1)) * row_bytes, hipHostMallocPortable));
            ev[s].assign(used, nullptr);
            for (uint32_t i = 0; i < used; i++) {
                HIP_TRY(hipSetDevice(g->ix[i]->device));
                HIP_TRY(hipEventCreateWithFlags(&ev[s][i], hipEventDisableTiming));
            }
        }
        const uint64_t n_chunks = ceil_div(n_rows, per);
        // load:  read(c) | copies(c) in flight while read(c + 1) runs;   save:  copies(c + 1) in flight while write(c) runs
        for (uint64_t c = 0; c < n_chunks + (save ? 1 : 0); c++) {
            const int slot = (int)(c & 1);
            const uint64_t r0 = row0 + c * per, cn = c < n_chunks ? std::min(per, row0 + n_rows - r0) : 0;
            if (save) {
                if (c < n_chunks) TRY(copies(slot, r0, cn));
                if (c > 0) {
                    const int ps = (int)((c - 1) & 1);
                    const uint64_t pr0 = row0 + (c - 1) * per, pn = std::min(per, row0 + n_rows - pr0);
                    TRY(wait_slot(ps));
                    const auto t0 = std::chrono::steady_clock::now();
                    const int e = rf.io(true, static_cast<uint8_t *>(pin[ps]), pr0 - row0, pn, threads);
                    io_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (e) return fail(BIGSI_ERR_INVALID, "writing %s: %s", path, strerror(e));
                }
            } else {
                TRY(wait_slot(slot));                                    // the copies that last read this buffer
                const auto t0 = std::chrono::steady_clock::now();
                const int e = rf.io(false, static_cast<uint8_t *>(pin[slot]), r0 - row0, cn, threads);
                io_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (e) return fail(BIGSI_ERR_INVALID, "reading %s: %s", path, e == ENODATA ? "file too short" : strerror(e));
                TRY(copies(slot, r0, cn));
            }
        }
        for (int s = 0; s < 2; s++) TRY(wait_slot(s));
        return BIGSI_OK;
    };
    int rc = body();
    if (save && rc == BIGSI_OK) { const int e_ = rf.sync_all(); if (e_) rc = fail(BIGSI_ERR_INVALID, "fsync %s: %s", path, strerror(e_)); }
    char keep[1024] = "";
    if (rc != BIGSI_OK) snprintf(keep, sizeof keep, "%s", bigsi_hip_last_error());
    hipError_t e = hipSuccess;
    for (uint32_t i = 0; i < g->n(); i++) {
        e = hipSetDevice(g->ix[i]->device);
        e = hipStreamSynchronize(g->ix[i]->stream);
    }
    for (int s = 0; s < 2; s++) {
        for (uint32_t i = 0; i < (uint32_t)ev[s].size(); i++)
            if (ev[s][i]) { e = hipSetDevice(g->ix[i]->device); e = hipEventDestroy(ev[s][i]); }
        if (pin[s]) e = hipHostFree(pin[s]);
    }
    (void)e;
    rf.close_();
    if (rc != BIGSI_OK) return fail(rc, "%s", keep);
    if (st) {
        st->bytes = n_rows * row_bytes;
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        st->file_seconds = io_s;
        st->threads = threads;
        st->direct = 1u;
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_load_rows_file(bigsi_hip_group *g, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                                              uint32_t threads, bigsi_hip_io_stats *stats)
{
    return group_rows_file(g, path, file_offset, row0, n_rows, row_bytes, threads, false, stats);
}

extern "C" int bigsi_hip_group_save_rows_file(bigsi_hip_group *g, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                                              uint32_t threads, bigsi_hip_io_stats *stats)
{
    return group_rows_file(g, path, file_offset, row0, n_rows, row_bytes, threads, true, stats);
}

extern "C" int bigsi_hip_group_insert_columns(bigsi_hip_group *g, uint64_t col0, uint64_t n, const uint8_t *blooms, uint64_t bloom_stride_bytes)
{
    if (!g || (n && !blooms)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (col0 > g->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu beyond num_cols %llu", (unsigned long long)col0, (unsigned long long)g->n_cols);
    if (col0 + n > g->cap_cols)
        return fail(BIGSI_ERR_CAPACITY, "columns [%llu,%llu) beyond the group's col_capacity %llu", (unsigned long long)col0,
                    (unsigned long long)(col0 + n), (unsigned long long)g->cap_cols);
    const uint64_t total = std::max(g->n_cols, col0 + n);
    for (uint64_t c = col0; c < col0 + n;) {
        const uint32_t i = (uint32_t)(c / g->shard_cols);
        const uint64_t local = c - (uint64_t)i * g->shard_cols;
        const uint64_t cnt = std::min(col0 + n - c, g->shard_cols - local);
        // the shard's own num_cols must reach `local` before it can take columns there (earlier shards are full by then)
        bigsi_hip_info inf;
        TRY(bigsi_hip_get_info(g->ix[i], &inf));
        if (inf.num_cols < local) TRY(bigsi_hip_set_num_cols(g->ix[i], local));
        TRY(bigsi_hip_insert_columns(g->ix[i], local, cnt, blooms + (c - col0) * bloom_stride_bytes, bloom_stride_bytes));
        c += cnt;
    }
    return bigsi_hip_group_set_num_cols(g, total);
}

static int locate_col(bigsi_hip_group *g, uint64_t col, uint32_t *shard, uint64_t *local)
{
    if (col >= g->cap_cols) return fail(BIGSI_ERR_RANGE, "column %llu beyond col_capacity %llu", (unsigned long long)col, (unsigned long long)g->cap_cols);
    *shard = (uint32_t)(col / g->shard_cols);
    *local = col - (uint64_t)*shard * g->shard_cols;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_get_column(bigsi_hip_group *g, uint64_t col, uint8_t *out)
{
    if (!g || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    uint32_t i;
    uint64_t local;
    TRY(locate_col(g, col, &i, &local));
    return bigsi_hip_get_column(g->ix[i], local, out);
}

extern "C" int bigsi_hip_group_insert_kmers(bigsi_hip_group *g, uint64_t col, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    if (col >= g->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu >= num_cols %llu", (unsigned long long)col, (unsigned long long)g->n_cols);
    uint32_t i;
    uint64_t local;
    TRY(locate_col(g, col, &i, &local));
    return bigsi_hip_insert_kmers(g->ix[i], local, seqs, offsets, n_seqs, k);
}

extern "C" int bigsi_hip_group_fill_synthetic(bigsi_hip_group *g, uint64_t seed, uint32_t and_draws)
{
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    for (uint32_t i = 0; i < g->n(); i++) TRY(bigsi_hip_fill_synthetic(g->ix[i], seed, i, and_draws));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_lookup(bigsi_hip_group *g, const char *kmers, uint32_t k, uint64_t u, uint8_t *out_rows)
{
    if (!g || (u && (!kmers || !out_rows))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (g->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    const uint64_t rb = ceil_div(g->n_cols, 8), sb = g->shard_cols / 8;
    std::vector<uint8_t> part;
    for (uint32_t i = 0; i < g->n(); i++) {
        const uint64_t nc = g->cols_of(i, g->n_cols);
        if (!nc) break;
        const uint64_t w = ceil_div(nc, 8);
        part.resize(u * w);
        TRY(bigsi_hip_lookup(g->ix[i], kmers, k, u, part.data()));
        for (uint64_t r = 0; r < u; r++) memcpy(out_rows + r * rb + (uint64_t)i * sb, part.data() + r * w, w);
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_lookup_raw(bigsi_hip_group *g, const char *blob, const uint64_t *elem_offsets, uint64_t u, uint8_t *out_rows)
{
    if (!g || (u && (!blob || !elem_offsets || !out_rows))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (g->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    const uint64_t rb = ceil_div(g->n_cols, 8), sb = g->shard_cols / 8;
    std::vector<uint8_t> part;
    for (uint32_t i = 0; i < g->n(); i++) {
        const uint64_t nc = g->cols_of(i, g->n_cols);
        if (!nc) break;
        const uint64_t w = ceil_div(nc, 8);
        part.resize(u * w);
        TRY(bigsi_hip_lookup_raw(g->ix[i], blob, elem_offsets, u, part.data()));
        for (uint64_t r = 0; r < u; r++) memcpy(out_rows + r * rb + (uint64_t)i * sb, part.data() + r * w, w);
    }
    return BIGSI_OK;
}

// ---- fused query path over all shards
// one member batch per shard (made by `make`), each bound to its shard's communicator
template <typename Make> static int group_batch_new(bigsi_hip_group *g, bigsi_hip_group_batch **out, Make make)
{
    if (!g || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    *out = nullptr;
    bigsi_hip_group_batch *gb = new (std::nothrow) bigsi_hip_group_batch();
    if (!gb) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    gb->g = g;
    int rc = BIGSI_OK;
    for (uint32_t i = 0; i < g->n() && rc == BIGSI_OK; i++) {
        bigsi_hip_batch *b = nullptr;
        rc = make(g->ix[i], &b);
        if (rc == BIGSI_OK) {
            gb->b.push_back(b);
            rc = bigsi_hip_batch_set_comm(b, g->comm[i], g->shard_cols);
        }
    }
    if (rc != BIGSI_OK) {
        char keep[1024];
        snprintf(keep, sizeof keep, "%s", bigsi_hip_last_error());
        bigsi_hip_group_batch_destroy(gb);
        return fail(rc, "%s", keep);
    }
    *out = gb;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_batch_create(bigsi_hip_group *g, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                                            bigsi_hip_group_batch **out)
{
    return group_batch_new(g, out, [&](bigsi_hip_index *ix, bigsi_hip_batch **b) { return bigsi_hip_batch_create(ix, seqs, offsets, n_seqs, k, b); });
}

extern "C" int bigsi_hip_group_batch_create_elements(bigsi_hip_group *g, const char *blob, const uint64_t *elem_offsets,
                                                     const uint64_t *seq_elem_offsets, const uint32_t *pos_unique,
                                                     const uint64_t *seq_pos_offsets, uint32_t n_seqs, bigsi_hip_group_batch **out)
{
    return group_batch_new(g, out, [&](bigsi_hip_index *ix, bigsi_hip_batch **b) {
        return bigsi_hip_batch_create_elements(ix, blob, elem_offsets, seq_elem_offsets, pos_unique, seq_pos_offsets, n_seqs, b);
    });
}

extern "C" int bigsi_hip_group_batch_reload(bigsi_hip_group_batch *gb, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    gb->ran = false;
    for (auto *b : gb->b) TRY(bigsi_hip_batch_reload(b, seqs, offsets, n_seqs, k));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_batch_destroy(bigsi_hip_group_batch *gb)
{
    if (!gb) return BIGSI_OK;
    for (auto *b : gb->b) bigsi_hip_batch_destroy(b);
    if (gb->shared.p && !gb->g->ix.empty()) {
        hipError_t e = hipSetDevice(gb->g->ix[0]->device); (void)e;
    }
    gb->shared.release();
    delete gb;
    return BIGSI_OK;
}

// sum of the members' per-hit count arrays, over `cap` entries: RCCL all-reduce, or -- shards sharing a device -- an add
// kernel into every member's array in turn (each ends up with the sum, as after an all-reduce)
static int group_reduce_counts(bigsi_hip_group_batch *gb)
{
    bigsi_hip_group *g = gb->g;
    const uint64_t cap = gb->b[0]->ghits.capacity();
    for (auto *b : gb->b)
        if (b->ghits.capacity() != cap) return fail(BIGSI_ERR_STATE, "hit buffers of the shards have diverged");
    gb->reduced_cap = cap;
    if (g->rccl) {
        RcclApi *api = nullptr;
        TRY(need_rccl(&api));
        NCCL_TRY(api, api->GroupStart());
        for (uint32_t i = 0; i < g->n(); i++) {
            bigsi_hip_batch *b = gb->b[i];
            HIP_TRY(hipSetDevice(g->ix[i]->device));
            ncclResult_t r = api->AllReduce(b->ghits.cnt(), b->ghits.cnt(), cap, ncclUint32, ncclSum, g->comm[i]->comm, g->comm[i]->stream);
            if (r != ncclSuccess) {
                api->GroupEnd();
                return fail(BIGSI_ERR_HIP, "ncclAllReduce on shard %u: %s", i, api->GetErrorString(r));
            }
        }
        NCCL_TRY(api, api->GroupEnd());
        return BIGSI_OK;
    }
    // one device: member 0 collects (the host only ever reads member 0's lists)
    HIP_TRY(hipSetDevice(g->ix[0]->device));
    hipStream_t s0 = g->comm[0]->stream;
    for (uint32_t i = 1; i < g->n(); i++) {
        HIP_TRY(hipStreamWaitEvent(s0, gb->b[i]->g_done, 0));
        hipLaunchKernelGGL(k_add_counts, dim3(256), dim3(256), 0, s0, gb->b[0]->ghits.cnt(), gb->b[i]->ghits.cnt(), cap);
        HIP_TRY(hipGetLastError());
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_batch_run(bigsi_hip_group_batch *gb, double threshold, uint32_t flags)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    bigsi_hip_group *g = gb->g;
    gb->ran = false;
    void *shared = nullptr;
    if (!g->rccl) {
        // every member's slot lives in ONE buffer on the shared device: the "all-gather" is the kernels' own stores
        const uint64_t need = slot_bytes(gb->b[0]) * g->n();
        HIP_TRY(hipSetDevice(g->ix[0]->device));
        if (gb->shared.cap < need) {
            for (auto *b : gb->b) {
                if (b->done) HIP_TRY(hipEventSynchronize(b->done));
                if (b->g_done) HIP_TRY(hipEventSynchronize(b->g_done));
            }
            TRY(gb->shared.reserve(need));
        }
        shared = gb->shared.p;
    }
    // K1-K3 on every device, asynchronously
    for (uint32_t i = 0; i < g->n(); i++) {
        bigsi_hip_batch *b = gb->b[i];
        TRY(bigsi_use_device(b->ix));
        if (shared)      // a member's K2 overwrites its slot: every member's previous compaction of the shared buffer must be over
            for (auto *o : gb->b)
                if (o != b && o->g_done) HIP_TRY(hipStreamWaitEvent(b->ix->stream, o->g_done, 0));
        TRY(prepare_slot(b, shared));
        TRY(bigsi_hip_batch_run(b, threshold, flags | BIGSI_RUN_SKIP_COMPACT | BIGSI_RUN_SPARSE_COUNTS));
    }
    // the exchange
    if (g->rccl) {
        RcclApi *api = nullptr;
        TRY(need_rccl(&api));
        for (uint32_t i = 0; i < g->n(); i++) {
            HIP_TRY(hipSetDevice(g->ix[i]->device));
            HIP_TRY(hipStreamWaitEvent(g->comm[i]->stream, gb->b[i]->done, 0));
        }
        NCCL_TRY(api, api->GroupStart());
        for (uint32_t i = 0; i < g->n(); i++) {
            bigsi_hip_batch *b = gb->b[i];
            HIP_TRY(hipSetDevice(g->ix[i]->device));
            ncclResult_t r = api->AllGather(b->ext_bitmaps, gather_base(b), slot_bytes(b), ncclUint8, g->comm[i]->comm, g->comm[i]->stream);
            if (r != ncclSuccess) {
                api->GroupEnd();
                return fail(BIGSI_ERR_HIP, "ncclAllGather on shard %u: %s", i, api->GetErrorString(r));
            }
        }
        NCCL_TRY(api, api->GroupEnd());
    } else {
        HIP_TRY(hipSetDevice(g->ix[0]->device));
        for (uint32_t i = 0; i < g->n(); i++)
            for (auto *o : gb->b) HIP_TRY(hipStreamWaitEvent(g->comm[i]->stream, o->done, 0));     // every slot written
    }
    for (uint32_t i = 0; i < g->n(); i++) {
        TRY(bigsi_use_device(gb->b[i]->ix));
        TRY(compact_after_gather(gb->b[i]));
    }
    if (!gb->b[0]->exact) TRY(group_reduce_counts(gb));
    gb->ran = true;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_batch_fetch_unique(bigsi_hip_group_batch *gb, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!gb->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_group_batch_run has not completed for this batch");
    return bigsi_hip_batch_fetch_unique(gb->b[0], num_kmers, num_unique, min_kmers);     // K1 is the same on every shard
}

extern "C" int bigsi_hip_group_batch_fetch_hits(bigsi_hip_group_batch *gb, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!gb->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_group_batch_run has not completed for this batch");
    if (!hit_offsets) return fail(BIGSI_ERR_INVALID, "hit_offsets is NULL");
    bigsi_hip_group *g = gb->g;
    if (!gb->b[0]->exact) {
        // a fetch with unbounded caller capacity makes every member grow + rewrite its lists if they overflowed (all members
        // see the same totals); the counts then have to be summed again over the larger arrays
        for (uint32_t i = 0; i < g->n(); i++) {
            TRY(bigsi_use_device(gb->b[i]->ix));
            TRY(bigsi_hip_batch_fetch_gathered_hits(gb->b[i], hit_offsets, nullptr, nullptr, ~0ull));
        }
        if (gb->b[0]->ghits.capacity() != gb->reduced_cap) {
            if (!g->rccl)      // the rewritten arrays hold own-shard counts only; member 0's add must see them complete
                for (uint32_t i = 1; i < g->n(); i++) {
                    HIP_TRY(hipSetDevice(g->ix[i]->device));
                    HIP_TRY(hipEventRecord(gb->b[i]->g_done, g->comm[i]->stream));
                }
            TRY(group_reduce_counts(gb));
        }
    }
    TRY(bigsi_use_device(gb->b[0]->ix));
    return bigsi_hip_batch_fetch_gathered_hits(gb->b[0], hit_offsets, colours, counts, capacity);
}

extern "C" int bigsi_hip_group_batch_presence(bigsi_hip_group_batch *gb, uint32_t seq, const uint32_t *colours, uint32_t n_colours, uint8_t *out)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!gb->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_group_batch_run has not completed for this batch");
    if (n_colours == 0) return BIGSI_OK;
    if (!colours || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    bigsi_hip_group *g = gb->g;
    for (uint32_t j = 0; j < n_colours; j++)
        if (colours[j] >= g->n_cols) return fail(BIGSI_ERR_RANGE, "colour %u >= num_cols", colours[j]);
    std::vector<uint32_t> nk(gb->b[0]->n_seqs);
    TRY(bigsi_hip_batch_fetch_unique(gb->b[0], nk.data(), nullptr, nullptr));
    if (seq >= nk.size()) return fail(BIGSI_ERR_RANGE, "sequence %u out of range", seq);
    const uint32_t n = nk[seq];
    if (n == 0) return BIGSI_OK;
    // each colour's string is produced on the device that owns the column (K5 there), then placed in the caller's order
    std::vector<uint32_t> local, where;
    std::vector<uint8_t> part;
    for (uint32_t i = 0; i < g->n(); i++) {
        local.clear();
        where.clear();
        for (uint32_t j = 0; j < n_colours; j++)
            if (colours[j] / g->shard_cols == i) {
                local.push_back((uint32_t)(colours[j] - (uint64_t)i * g->shard_cols));
                where.push_back(j);
            }
        if (local.empty()) continue;
        part.resize((size_t)local.size() * n);
        TRY(bigsi_hip_batch_presence(gb->b[i], seq, local.data(), (uint32_t)local.size(), part.data()));
        for (size_t t = 0; t < local.size(); t++) memcpy(out + (size_t)where[t] * n, part.data() + t * n, n);
    }
    return BIGSI_OK;
}

// all hits of the batch: each shard produces the strings of the hits it owns (K5 there, one pass per shard), the host puts
// them in the caller's order
extern "C" int bigsi_hip_group_batch_presence_hits(bigsi_hip_group_batch *gb, const uint64_t *hit_offsets, const uint32_t *colours, uint8_t *out,
                                                   uint64_t out_capacity, uint64_t *string_offsets)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!gb->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_group_batch_run has not completed for this batch");
    if (!hit_offsets || !string_offsets) return fail(BIGSI_ERR_INVALID, "NULL argument");
    bigsi_hip_group *g = gb->g;
    const uint32_t nq = gb->b[0]->n_seqs;
    const uint64_t h0 = hit_offsets[0], n_hits = hit_offsets[nq] - h0;
    if (n_hits && !colours) return fail(BIGSI_ERR_INVALID, "colours is NULL");
    std::vector<uint32_t> nk(nq);
    TRY(bigsi_hip_batch_fetch_unique(gb->b[0], nk.data(), nullptr, nullptr));
    uint64_t str = 0;
    for (uint32_t q = 0; q < nq; q++)
        for (uint64_t t = hit_offsets[q] - h0; t < hit_offsets[q + 1] - h0; t++) {
            if (colours[h0 + t] >= g->n_cols) return fail(BIGSI_ERR_RANGE, "colour %u >= num_cols", colours[h0 + t]);
            string_offsets[t] = str;
            str += round_up(nk[q], 16);
        }
    string_offsets[n_hits] = str;
    if (str > out_capacity) return fail(BIGSI_ERR_CAPACITY, "string buffer holds %llu bytes, %llu needed", (unsigned long long)out_capacity, (unsigned long long)str);
    if (n_hits == 0 || str == 0) return BIGSI_OK;
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    std::vector<uint64_t> off(nq + 1), soff;
    std::vector<uint32_t> local, where;
    std::vector<uint8_t> part;
    for (uint32_t i = 0; i < g->n(); i++) {
        local.clear();
        where.clear();
        for (uint32_t q = 0; q < nq; q++) {
            off[q] = local.size();
            for (uint64_t t = hit_offsets[q] - h0; t < hit_offsets[q + 1] - h0; t++)
                if (colours[h0 + t] / g->shard_cols == i) {
                    local.push_back((uint32_t)(colours[h0 + t] - (uint64_t)i * g->shard_cols));
                    where.push_back((uint32_t)t);
                }
        }
        off[nq] = local.size();
        if (local.empty()) continue;
        uint64_t need = 0;
        for (uint32_t t : where) need += string_offsets[t + 1] - string_offsets[t];      // padded lengths: the shard pads the same way
        part.resize(need);
        soff.resize(local.size() + 1);
        TRY(bigsi_use_device(gb->b[i]->ix));
        TRY(bigsi_hip_batch_presence_hits(gb->b[i], off.data(), local.data(), part.data(), need, soff.data()));
        for (size_t r = 0; r < where.size(); r++) memcpy(out + string_offsets[where[r]], part.data() + soff[r], soff[r + 1] - soff[r]);
    }
    return BIGSI_OK;
}

// K6 over a group: every shard scores the hits whose columns it owns (packed presence bits + score records), the host puts them
// in the caller's order
extern "C" int bigsi_hip_group_batch_score_hits(bigsi_hip_group_batch *gb, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts,
                                                uint8_t *bits, uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores)
{
    if (!gb) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!gb->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_group_batch_run has not completed for this batch");
    if (!hit_offsets || !bit_offsets || !scores) return fail(BIGSI_ERR_INVALID, "NULL argument");
    bigsi_hip_group *g = gb->g;
    const uint32_t nq = gb->b[0]->n_seqs;
    const uint64_t h0 = hit_offsets[0], n_hits = hit_offsets[nq] - h0;
    if (n_hits && !colours) return fail(BIGSI_ERR_INVALID, "colours is NULL");
    std::vector<uint32_t> nk(nq);
    TRY(bigsi_hip_batch_fetch_unique(gb->b[0], nk.data(), nullptr, nullptr));
    uint64_t need_all = 0;
    for (uint32_t q = 0; q < nq; q++)
        for (uint64_t t = hit_offsets[q] - h0; t < hit_offsets[q + 1] - h0; t++) {
            if (colours[h0 + t] >= g->n_cols) return fail(BIGSI_ERR_RANGE, "colour %u >= num_cols", colours[h0 + t]);
            bit_offsets[t] = need_all;
            need_all += round_up(nk[q], 64) / 8;
        }
    bit_offsets[n_hits] = need_all;
    if (need_all > bits_capacity) return fail(BIGSI_ERR_CAPACITY, "bit buffer holds %llu bytes, %llu needed", (unsigned long long)bits_capacity, (unsigned long long)need_all);
    if (n_hits == 0) return BIGSI_OK;
    if (need_all && !bits) return fail(BIGSI_ERR_INVALID, "bits is NULL");
    std::vector<uint64_t> off(nq + 1), soff;
    std::vector<uint32_t> local, lcnt, where;
    std::vector<uint8_t> part;
    std::vector<bigsi_hip_hit_score> prec;
    for (uint32_t i = 0; i < g->n(); i++) {
        local.clear();
        lcnt.clear();
        where.clear();
        for (uint32_t q = 0; q < nq; q++) {
            off[q] = local.size();
            for (uint64_t t = hit_offsets[q] - h0; t < hit_offsets[q + 1] - h0; t++)
                if (colours[h0 + t] / g->shard_cols == i) {
                    local.push_back((uint32_t)(colours[h0 + t] - (uint64_t)i * g->shard_cols));
                    if (counts) lcnt.push_back(counts[h0 + t]);
                    where.push_back((uint32_t)t);
                }
        }
        off[nq] = local.size();
        if (local.empty()) continue;
        uint64_t need = 0;
        for (uint32_t t : where) need += bit_offsets[t + 1] - bit_offsets[t];
        part.resize(std::max<uint64_t>(need, 8));
        soff.resize(local.size() + 1);
        prec.resize(local.size());
        TRY(bigsi_use_device(gb->b[i]->ix));
        TRY(bigsi_hip_batch_score_hits(gb->b[i], off.data(), local.data(), counts ? lcnt.data() : nullptr, part.data(), part.size(), soff.data(), prec.data()));
        for (size_t r = 0; r < where.size(); r++) {
            memcpy(bits + bit_offsets[where[r]], part.data() + soff[r], soff[r + 1] - soff[r]);
            scores[where[r]] = prec[r];
        }
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_group_search_batch(bigsi_hip_group *g, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                                            double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                            uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity)
{
    if (!hit_offsets) return fail(BIGSI_ERR_INVALID, "hit_offsets is NULL");
    if (!g) return fail(BIGSI_ERR_INVALID, "NULL group");
    if (!g->search_ws) TRY(bigsi_hip_group_batch_create(g, seqs, offsets, n_seqs, k, &g->search_ws));      // kept, as for one index
    else TRY(bigsi_hip_group_batch_reload(g->search_ws, seqs, offsets, n_seqs, k));
    bigsi_hip_group_batch *gb = g->search_ws;
    int rc = bigsi_hip_group_batch_run(gb, threshold, flags);
    if (rc == BIGSI_OK) rc = bigsi_hip_group_batch_fetch_unique(gb, num_kmers, num_unique, min_kmers);
    if (rc == BIGSI_OK) rc = bigsi_hip_group_batch_fetch_hits(gb, hit_offsets, colours, counts, hit_capacity);
    if (rc != BIGSI_OK && rc != BIGSI_ERR_CAPACITY) {
        g->search_ws = nullptr;
        bigsi_hip_group_batch_destroy(gb);
    }
    return rc;
}
