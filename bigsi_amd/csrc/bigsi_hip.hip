// bigsi_hip.hip -- host side of libbigsi_hip.so: the C ABI declared in include/bigsi_hip.h.
// Plain HIP runtime (hipMalloc / streams / events); no torch types anywhere in this library.
#include "bigsi_internal.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bigsi_kernels.hpp"

using namespace bigsi;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

int bigsi_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *bigsi_hip_last_error(void) { return g_err; }

extern "C" int bigsi_hip_device_count(int *out)
{
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    HIP_TRY(hipGetDeviceCount(out));
    return BIGSI_OK;
}

int bigsi_use_device(const bigsi_hip_index *ix)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == ix->device) return BIGSI_OK;      // (a thread-local read: every entry point comes through here)
    HIP_TRY(hipSetDevice(ix->device));
    return BIGSI_OK;
}
#define use_device bigsi_use_device

// The one-launch read kernel (k_reads_fused) of successive batches goes out on alternating library streams, so that one
// batch's k-merising and hit compaction -- phases with the HBM idle -- overlap the row fetches of its neighbours (BASELINE
// configs[1]: 31.6 -> 25 us per step).  Ordering: a batch's next run waits for its own `done` event; everything that
// changes the index, reads the profiling events or hands the stream back waits for the read streams (quiesce_reads).
static int read_stream(bigsi_hip_index *ix, hipStream_t *out)
{
    *out = ix->stream;
    if (ix->stream != ix->own_stream) return BIGSI_OK;       // the caller's own stream (bigsi_hip_set_stream): everything stays on it
    static const int n_streams = std::min(env_int("BIGSI_HIP_READ_STREAMS", kReadStreamsUsed), kReadStreams);      // A/B: 1 = no overlap
    if (n_streams <= 1) return BIGSI_OK;
    if (!ix->rd_stream[0])
        for (auto &st : ix->rd_stream) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out = ix->rd_stream[ix->rd_next++ % (uint32_t)n_streams];
    ix->rd_pending = true;
    return BIGSI_OK;
}

// end of a batch run on the index stream: what later read-kernel launches wait for.  Only once read streams exist (an index
// that serves gene-length queries alone never pays for the record).
static int mark_main(bigsi_hip_index *ix)
{
    if (!ix->rd_stream[0] || ix->stream != ix->own_stream) return BIGSI_OK;
    if (!ix->main_ev) HIP_TRY(hipEventCreateWithFlags(&ix->main_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ix->main_ev, ix->stream));
    return BIGSI_OK;
}

// K5 / K6 (scored searches) run on a stream of their own with the HIGHEST priority: they are a few small kernels whose result
// the host waits for while the NEXT batch's row-AND kernel fills the device, and at equal priority their workgroups queue up
// behind that kernel's (bench workload c5: 0.37 ms of waiting for 0.05 ms of work).  A request may stay queued here after its
// _begin returns (sc_pending): calls that change the index wait for it (quiesce_index).
static int score_stream(bigsi_hip_index *ix, hipStream_t *out)
{
    *out = ix->stream;
    if (ix->stream != ix->own_stream) return BIGSI_OK;       // the caller's own stream: everything stays on it
    if (!ix->sc_stream) {
        // (tuning builds: BIGSI_HIP_SCORE_CUS = n > 0 confines the stream to n compute units instead -- a CU-masked stream has no priority)
        static const int score_cus = env_int("BIGSI_HIP_SCORE_CUS", 0);
        if (score_cus > 0) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            static const int stride = std::max(env_int("BIGSI_HIP_SCORE_CU_STRIDE", 1), 1);
            for (int i = 0, c = 0; i < score_cus && c < 256; i++, c += stride) mask[c / 32] |= 1u << (c % 32);
            HIP_TRY(hipExtStreamCreateWithCUMask(&ix->sc_stream, 8, mask));
        } else {
            int least = 0, greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_TRY(hipStreamCreateWithPriority(&ix->sc_stream, hipStreamNonBlocking, greatest));
        }
    }
    *out = ix->sc_stream;
    return BIGSI_OK;
}

static int quiesce_reads(bigsi_hip_index *ix)
{
    if (!ix->rd_pending) return BIGSI_OK;
    for (auto st : ix->rd_stream)
        if (st) HIP_TRY(hipStreamSynchronize(st));
    ix->rd_pending = false;
    return BIGSI_OK;
}

// Everything that reads the matrix on a stream other than the index stream is over: the read streams AND the score stream
// (a K5 request between its _begin and its _end -- bigsi_hip_batch_score_hits_begin, the scored stream's chunks -- is
// k_presence_bits reading d_index there).  Every entry point that changes the index, hands the stream back or reads the
// profiling events calls this; batch runs call quiesce_reads only (scoring beside the next batch's row-AND is the point).
static int quiesce_index(bigsi_hip_index *ix)
{
    TRY(quiesce_reads(ix));
    if (ix->sc_pending && ix->sc_stream) HIP_TRY(hipStreamSynchronize(ix->sc_stream));
    ix->sc_pending = false;
    return BIGSI_OK;
}

static uint64_t stride_for(uint64_t cols)
{
    static const int align_words = env_int("BIGSI_HIP_ROW_ALIGN_WORDS", 16);      // A/B (tuning builds): 256 = rows at a 2 KB pitch
    return std::max<uint64_t>(16, round_up(ceil_div(cols, 64), (uint64_t)std::max(align_words, 16)));
}

// The matrix is an ordinary hipMalloc.  Tuning builds can ask for PHYSICALLY CONTIGUOUS memory instead (BIGSI_HIP_CONTIGUOUS=1:
// hipDeviceMallocContiguous, largest page-table fragments): the bare-kernel probe measured +4 % on random 12.5 KB rows with it and
// the counting kernel +0.5 %, but in a process that opens and closes several indexes one after the other it CORRUPTS another
// buffer -- counters of the last queries of a batch read back as zeros, deterministically, in
// test_one_launch_read_path_equals_three_launch_path[4-32768] (bisected to this flag alone; the driver appears to move or clear
// memory under a running process when it makes room for a contiguous block).  Not shipped.
static hipError_t index_malloc(uint64_t **out, size_t bytes, bool *contiguous)
{
    static const int contig = env_int("BIGSI_HIP_CONTIGUOUS", 0);
    *contiguous = false;
    if (contig) {
        hipError_t e = hipExtMallocWithFlags((void **)out, bytes, hipDeviceMallocContiguous);
        if (e == hipSuccess) { *contiguous = true; return e; }
        (void)hipGetLastError();          // clear the sticky error: fall back
        *out = nullptr;
    }
    return hipMalloc((void **)out, bytes);
}

// ------------------------------------------------------------------------------ lifecycle
int bigsi_writable(const bigsi_hip_index *ix)
{
    if (ix && ix->attach != bigsi_hip_index::kOwner)
        return fail(BIGSI_ERR_STATE, "this handle does not own its matrix (%s): it is read-only", ix->attach == bigsi_hip_index::kIpc ? "attached over hipIpc" : "a view");
    return BIGSI_OK;
}

// the matrix of a new handle: allocated and zeroed here (ipc == nullptr && view == nullptr), another process's allocation mapped
// into this one (ipc), or another handle's pointer (view)
static int open_impl(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes, int device, const hipIpcMemHandle_t *ipc,
                     const bigsi_hip_index *view, bigsi_hip_index **out)
{
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_rows == 0) return fail(BIGSI_ERR_INVALID, "num_rows must be > 0");
    if (num_hashes == 0) return fail(BIGSI_ERR_INVALID, "num_hashes must be > 0");
    if (col_capacity < num_cols) col_capacity = num_cols;
    if (col_capacity == 0) col_capacity = 64;
    if (col_capacity > 0xFFFFFFFFull) return fail(BIGSI_ERR_INVALID, "col_capacity %llu exceeds 2^32-1 colours per shard", (unsigned long long)col_capacity);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BIGSI_ERR_INVALID, "device %d not in [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    bigsi_hip_index *ix = new (std::nothrow) bigsi_hip_index();
    if (!ix) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    ix->device = device;
    ix->m = num_rows;
    ix->n_cols = num_cols;
    ix->h = num_hashes;
    ix->stride_words = stride_for(col_capacity);
    ix->cap_cols = ix->stride_words * 64;
    hipError_t e = hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ix; return fail(BIGSI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    ix->stream = ix->own_stream;
    {
        // lowest priority: whatever runs here must not delay the workgroups of a row-AND kernel on the index stream
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        static const int pre_prio = env_int("BIGSI_HIP_PRE_PRIORITY", 1);
        if (e == hipSuccess && pre_prio) e = hipStreamCreateWithPriority(&ix->pre_stream, hipStreamNonBlocking, least);
        else e = hipStreamCreateWithFlags(&ix->pre_stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
        hipError_t e2 = hipStreamDestroy(ix->own_stream); (void)e2;
        delete ix;
        return fail(BIGSI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    const size_t bytes = (size_t)ix->m * ix->stride_words * 8;
    if (view) {
        ix->d_index = view->d_index;
        ix->attach = bigsi_hip_index::kView;
        ix->view_of = const_cast<bigsi_hip_index *>(view);
        ix->view_of->views++;
        *out = ix;
        return BIGSI_OK;
    }
    if (ipc) {
        void *p = nullptr;
        e = hipIpcOpenMemHandle(&p, *ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();          // (the runtime's last-error slot is sticky: the next launch check of this thread must not inherit it)
            hipError_t e2 = hipStreamDestroy(ix->own_stream); (void)e2;
            e2 = hipStreamDestroy(ix->pre_stream);
            delete ix;
            return fail(BIGSI_ERR_HIP, "hipIpcOpenMemHandle: %s (the owner must be another live process on this device; HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose "
                                       "driver only does dmabuf IPC)", hipGetErrorString(e));
        }
        ix->d_index = static_cast<uint64_t *>(p);
        ix->attach = bigsi_hip_index::kIpc;
        *out = ix;
        return BIGSI_OK;
    }
    e = index_malloc(&ix->d_index, bytes, &ix->contiguous);
    if (e != hipSuccess) {
        hipError_t e2 = hipStreamDestroy(ix->own_stream); (void)e2;
        e2 = hipStreamDestroy(ix->pre_stream);
        delete ix;
        return fail(BIGSI_ERR_NOMEM, "hipMalloc of %zu index bytes failed: %s", bytes, hipGetErrorString(e));
    }
    e = hipMemsetAsync(ix->d_index, 0, bytes, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
        hipError_t e2 = hipFree(ix->d_index); (void)e2;
        e2 = hipStreamDestroy(ix->own_stream);
        e2 = hipStreamDestroy(ix->pre_stream);
        delete ix;
        return fail(BIGSI_ERR_HIP, "zeroing the index failed: %s", hipGetErrorString(e));
    }
    *out = ix;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_open(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes, int device,
                              bigsi_hip_index **out)
{
    return open_impl(num_rows, num_cols, col_capacity, num_hashes, device, nullptr, nullptr, out);
}

// ------------------------------------------------------------------------------ an index shared between processes / threads
// The reference opens its store once per request and once per pool worker (bigsi/__main__.py:75-80, 204-205;
// storage/berkeleydb.py:12-19): the data lives in a file, any process can open it.  Here the data lives in ONE process's HBM
// allocation; these three give other handles onto it without a second copy:
//   export_ipc   the owner's matrix as a 64-byte hipIpc handle (any byte channel carries it: a file, a socket)
//   open_ipc     another PROCESS maps that allocation: a read-only handle with streams / workspaces of its own.  The caller passes
//                the owner's geometry (rows, columns, column capacity, hashes: bigsi_hip_get_info there).  The owner must stay
//                alive and must not close / re-stride (reserve_cols) the index while handles are attached.
//   open_view    another THREAD of the owner's process: the same, without the mapping (hipIpc does not open a handle in the
//                process that made it).  The owner handle must outlive its views.
static_assert(sizeof(hipIpcMemHandle_t) == BIGSI_IPC_HANDLE_BYTES, "hipIpc handle size");

extern "C" int bigsi_hip_export_ipc(bigsi_hip_index *ix, uint8_t *handle)
{
    if (!ix || !handle) return fail(BIGSI_ERR_INVALID, "NULL argument");
    BIGSI_ENTER(ix);
    if (ix->attach != bigsi_hip_index::kOwner) return fail(BIGSI_ERR_STATE, "only the handle that owns the matrix can export it");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    HIP_TRY(hipStreamSynchronize(ix->stream));       // whatever filled the matrix is visible to the process that maps it next
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, ix->d_index));
    memcpy(handle, &h, sizeof h);
    return BIGSI_OK;
}

extern "C" int bigsi_hip_open_ipc(const uint8_t *handle, uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes,
                                  int device, bigsi_hip_index **out)
{
    if (!handle) return fail(BIGSI_ERR_INVALID, "handle is NULL");
    if (num_cols > col_capacity) return fail(BIGSI_ERR_INVALID, "num_cols %llu exceeds the owner's column capacity %llu", (unsigned long long)num_cols, (unsigned long long)col_capacity);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    return open_impl(num_rows, num_cols, col_capacity, num_hashes, device, &h, nullptr, out);
}

extern "C" int bigsi_hip_open_view(bigsi_hip_index *owner, bigsi_hip_index **out)
{
    if (!owner || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    BIGSI_ENTER(owner);
    TRY(use_device(owner));
    TRY(quiesce_index(owner));
    HIP_TRY(hipStreamSynchronize(owner->stream));    // (streams of different handles are not ordered with each other)
    TRY(open_impl(owner->m, owner->n_cols, owner->cap_cols, owner->h, owner->device, nullptr, owner, out));
    return BIGSI_OK;
}

static void recycle_events(bigsi_hip_index *ix)
{
    for (auto *v : {&ix->ev_and, &ix->ev_km, &ix->ev_cp, &ix->ev_pr, &ix->ev_tr, &ix->ev_ex}) {
        for (auto &p : *v) ix->ev_free.push_back(p);
        v->clear();
    }
}

extern "C" int bigsi_hip_close(bigsi_hip_index *ix)
{
    if (!ix) return BIGSI_OK;
    {
        BusyGuard g(ix);
        if (!g.ok) return fail(BIGSI_ERR_STATE, "bigsi_hip_close: the handle is in use by another host thread");
        g.release();      // (the guard must not outlive the handle; whoever closes a handle owns it)
    }
    if (ix->views.load() > 0) return fail(BIGSI_ERR_STATE, "bigsi_hip_close: %d view(s) of this index are still open (close them first)", ix->views.load());
    if (ix->search_ws) {
        bigsi_hip_batch_destroy(ix->search_ws);
        ix->search_ws = nullptr;
    }
    for (auto &w : ix->stream_ws)
        if (w) { bigsi_hip_batch_destroy(w); w = nullptr; }
    hipError_t e = hipSetDevice(ix->device);
    e = hipStreamSynchronize(ix->stream);
    if (ix->pre_stream) e = hipStreamSynchronize(ix->pre_stream);
    for (auto st : ix->rd_stream)
        if (st) { e = hipStreamSynchronize(st); e = hipStreamDestroy(st); }
    if (ix->sc_stream) { e = hipStreamSynchronize(ix->sc_stream); e = hipStreamDestroy(ix->sc_stream); }
    if (ix->main_ev) e = hipEventDestroy(ix->main_ev);
    recycle_events(ix);
    for (auto &p : ix->ev_free) { e = hipEventDestroy(p.a); e = hipEventDestroy(p.b); }
    ix->stage.release();
    ix->stage_ids.release();
    for (int s_ = 0; s_ < 2; s_++) {
        if (ix->pin_rows[s_]) e = hipHostFree(ix->pin_rows[s_]);
        if (ix->pin_ev[s_]) e = hipEventDestroy(ix->pin_ev[s_]);
        ix->dev_rows[s_].release();
        ix->dev_ids[s_].release();
    }
    if (ix->view_of) ix->view_of->views--;
    if (ix->d_index && ix->attach == bigsi_hip_index::kOwner) e = hipFree(ix->d_index);
    else if (ix->d_index && ix->attach == bigsi_hip_index::kIpc) e = hipIpcCloseMemHandle(ix->d_index);
    if (ix->own_stream) e = hipStreamDestroy(ix->own_stream);
    if (ix->pre_stream) e = hipStreamDestroy(ix->pre_stream);
    (void)e;
    delete ix;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_get_info(const bigsi_hip_index *ix, bigsi_hip_info *out)
{
    if (!ix || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    out->num_rows = ix->m;
    out->num_cols = ix->n_cols;
    out->col_capacity = ix->cap_cols;
    out->row_bytes = ix->rb();
    out->row_stride_bytes = ix->stride_words * 8;
    out->index_bytes = ix->m * ix->stride_words * 8;
    out->num_hashes = ix->h;
    out->device = ix->device;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_set_num_cols(bigsi_hip_index *ix, uint64_t num_cols)
{
    BIGSI_ENTER(ix);
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (num_cols > ix->cap_cols)
        return fail(BIGSI_ERR_CAPACITY, "num_cols %llu exceeds col_capacity %llu (call bigsi_hip_reserve_cols)",
                    (unsigned long long)num_cols, (unsigned long long)ix->cap_cols);
    ix->n_cols = num_cols;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_set_num_hashes(bigsi_hip_index *ix, uint32_t num_hashes)
{
    BIGSI_ENTER(ix);
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (num_hashes == 0) return fail(BIGSI_ERR_INVALID, "num_hashes must be > 0");
    ix->h = num_hashes;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_reserve_cols(bigsi_hip_index *ix, uint64_t col_capacity)
{
    BIGSI_ENTER(ix);
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (col_capacity <= ix->cap_cols) return BIGSI_OK;
    TRY(bigsi_writable(ix));          // (only a call that would really re-stride the matrix needs to own it)
    if (col_capacity > 0xFFFFFFFFull) return fail(BIGSI_ERR_INVALID, "col_capacity exceeds 2^32-1");
    if (ix->views.load() > 0) return fail(BIGSI_ERR_STATE, "bigsi_hip_reserve_cols: re-striding would move the matrix under %d open view(s)", ix->views.load());
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    const uint64_t ns = stride_for(col_capacity);
    uint64_t *nd = nullptr;
    bool nd_contig = false;
    HIP_TRY(index_malloc(&nd, (size_t)ix->m * ns * 8, &nd_contig));
    const uint64_t total = ix->m * ns;
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(total, kBlock), 256 * 8 * 4);
    hipLaunchKernelGGL(k_restride, dim3(grid), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, nd, ns, ix->m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
    HIP_TRY(hipFree(ix->d_index));
    ix->d_index = nd;
    ix->contiguous = nd_contig;
    ix->stride_words = ns;
    ix->cap_cols = ns * 64;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_set_stream(bigsi_hip_index *ix, void *hip_stream)
{
    BIGSI_ENTER(ix);
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->stream = hip_stream ? (hipStream_t)hip_stream : ix->own_stream;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_synchronize(bigsi_hip_index *ix)
{
    BIGSI_ENTER(ix);
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    HIP_TRY(hipStreamSynchronize(ix->pre_stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ storage contract
static const uint64_t kStageBytes = 64ull << 20;
static const uint64_t kZeroCopyBytes = 256ull << 10;     // inputs of a one-call search up to this size are read by K1 from pinned memory

extern "C" int bigsi_hip_set_rows(bigsi_hip_index *ix, const uint64_t *row_ids, uint64_t n, const uint8_t *bytes, uint64_t row_bytes)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    if (!ix || (n && (!row_ids || !bytes))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (row_bytes == 0 || row_bytes > ix->stride_words * 8)
        return fail(BIGSI_ERR_CAPACITY, "row_bytes %llu not in [1, %llu] (row stride)", (unsigned long long)row_bytes,
                    (unsigned long long)(ix->stride_words * 8));
    for (uint64_t i = 0; i < n; i++)
        if (row_ids[i] >= ix->m) return fail(BIGSI_ERR_RANGE, "row %llu out of range [0,%llu)", (unsigned long long)row_ids[i], (unsigned long long)ix->m);
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    const uint64_t kMaxChunkRows = 1ull << 20;      // (the pinned buffers keep room for this many row ids behind the row bytes)
    const uint64_t per = std::min(kMaxChunkRows, std::max<uint64_t>(1, kStageBytes / row_bytes));
    if (n * row_bytes < (4ull << 20) || row_bytes > kStageBytes) {
        // a few rows (the storage contract's own calls): one staged copy.  Rows wider than a pinned buffer of the route below (64 MB:
        // more than 512 M columns) come this way too, one row at a time -- that route sizes its buffers for kStageBytes and would
        // overrun them (round-5 advisor)
        const uint64_t rows_at_once = row_bytes > kStageBytes ? 1 : n;
        for (uint64_t i0 = 0; i0 < n; i0 += rows_at_once) {
            const uint64_t c = std::min(rows_at_once, n - i0);
            TRY(ix->stage.reserve(c * row_bytes));
            TRY(ix->stage_ids.reserve(c * 8));
            HIP_TRY(hipMemcpyAsync(ix->stage.p, bytes + i0 * row_bytes, c * row_bytes, hipMemcpyHostToDevice, ix->stream));
            HIP_TRY(hipMemcpyAsync(ix->stage_ids.p, row_ids + i0, c * 8, hipMemcpyHostToDevice, ix->stream));
            hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)c), dim3(kBlock), 0, ix->stream, (uint8_t *)ix->d_index,
                               ix->stride_words * 8, ix->m, ix->stage_ids.as<uint64_t>(), ix->stage.as<uint8_t>(), row_bytes);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(ix->stream));
        }
        return BIGSI_OK;
    }
    // blocks of rows (importers: migrate_index, bdb.import_index): the caller's pageable memory goes into one of two pinned
    // buffers on a few host threads while the other buffer's copy and scatter kernel are in flight (a hipMemcpy from pageable
    // memory stages through the runtime's own small buffers on one thread: 7-8 GB/s)
    for (int s_ = 0; s_ < 2; s_++) {
        if (!ix->pin_rows[s_]) HIP_TRY(hipHostMalloc(&ix->pin_rows[s_], kStageBytes + kMaxChunkRows * 8, hipHostMallocDefault));
        if (!ix->pin_ev[s_]) HIP_TRY(hipEventCreateWithFlags(&ix->pin_ev[s_], hipEventDisableTiming));
        TRY(ix->dev_rows[s_].reserve(kStageBytes));
        TRY(ix->dev_ids[s_].reserve(per * 8));
    }
    const unsigned T = std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 8));
    uint64_t chunk = 0;
    for (uint64_t i0 = 0; i0 < n; i0 += per, chunk++) {
        const int s_ = (int)(chunk & 1);
        const uint64_t c = std::min(per, n - i0), nb = c * row_bytes;
        HIP_TRY(hipEventSynchronize(ix->pin_ev[s_]));          // (never recorded: returns at once) the copy that last read this buffer
        uint8_t *pin = static_cast<uint8_t *>(ix->pin_rows[s_]);
        const uint8_t *src = bytes + i0 * row_bytes;
        {
            std::vector<std::thread> pool;
            const uint64_t part = round_up(ceil_div(nb, T), 1 << 16);
            for (unsigned t = 1; t < T; t++) {
                const uint64_t a = std::min<uint64_t>((uint64_t)t * part, nb), b = std::min<uint64_t>((uint64_t)(t + 1) * part, nb);
                if (a < b) pool.emplace_back([=]() { memcpy(pin + a, src + a, b - a); });
            }
            memcpy(pin, src, std::min(part, nb));
            for (auto &th : pool) th.join();
        }
        memcpy(pin + kStageBytes, row_ids + i0, c * 8);
        HIP_TRY(hipMemcpyAsync(ix->dev_rows[s_].p, pin, nb, hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipMemcpyAsync(ix->dev_ids[s_].p, pin + kStageBytes, c * 8, hipMemcpyHostToDevice, ix->stream));
        hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)c), dim3(kBlock), 0, ix->stream, (uint8_t *)ix->d_index,
                           ix->stride_words * 8, ix->m, ix->dev_ids[s_].as<uint64_t>(), ix->dev_rows[s_].as<uint8_t>(), row_bytes);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ix->pin_ev[s_], ix->stream));
    }
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_get_rows(bigsi_hip_index *ix, const uint64_t *row_ids, uint64_t n, uint8_t *out, uint64_t row_bytes)
{
    BIGSI_ENTER(ix);
    if (!ix || (n && (!row_ids || !out))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (row_bytes == 0) return fail(BIGSI_ERR_INVALID, "row_bytes is 0");
    for (uint64_t i = 0; i < n; i++)
        if (row_ids[i] >= ix->m) return fail(BIGSI_ERR_RANGE, "row %llu out of range [0,%llu)", (unsigned long long)row_ids[i], (unsigned long long)ix->m);
    TRY(use_device(ix));
    const uint64_t per = std::max<uint64_t>(1, kStageBytes / row_bytes);
    for (uint64_t i0 = 0; i0 < n; i0 += per) {
        const uint64_t c = std::min(per, n - i0);
        TRY(ix->stage.reserve(c * row_bytes));
        TRY(ix->stage_ids.reserve(c * 8));
        HIP_TRY(hipMemcpyAsync(ix->stage_ids.p, row_ids + i0, c * 8, hipMemcpyHostToDevice, ix->stream));
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)c), dim3(kBlock), 0, ix->stream, (const uint8_t *)ix->d_index,
                           ix->stride_words * 8, ix->m, ix->stage_ids.as<uint64_t>(), ix->stage.as<uint8_t>(), row_bytes);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out + i0 * row_bytes, ix->stage.p, c * row_bytes, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ bulk ingest: file <-> HBM
// The reference loads an index by opening a BerkeleyDB file (bigsi/storage/berkeleydb.py:6-19) and pays a page lookup + two copies
// per row at query time; here the rows go to HBM once, at the rate the file system and PCIe allow: the file range is read by
// `threads` host threads (pread, one slice each) into one of two pinned buffers while the other one is in flight to the device
// (hipMemcpyAsync; rows whose file length is not the device pitch take one scatter kernel on the way).
namespace {
struct IoJob {
    int fd = -1;
    bool write = false;
    std::atomic<int> err{0};
};

}   // namespace
// read / write [off, off + len) of the file into / from buf with up to `threads` threads; returns 0 or errno
int bigsi_file_io(int fd, bool write, uint8_t *buf, uint64_t off, uint64_t len, unsigned threads)
{
    threads = std::max(1u, std::min<unsigned>(threads, (unsigned)ceil_div(std::max<uint64_t>(len, 1), 4ull << 20)));
    std::atomic<int> err{0};
    auto work = [&](uint64_t a, uint64_t b) {
        while (a < b && !err.load()) {
            const ssize_t r = write ? pwrite(fd, buf + (a - off), (size_t)std::min<uint64_t>(b - a, 1ull << 30), (off_t)a)
                                    : pread(fd, buf + (a - off), (size_t)std::min<uint64_t>(b - a, 1ull << 30), (off_t)a);
            if (r < 0) { if (errno == EINTR) continue; err.store(errno); return; }
            if (r == 0) { err.store(write ? EIO : ENODATA); return; }      // short file
            a += (uint64_t)r;
        }
    };
    if (threads == 1) { work(off, off + len); return err.load(); }
    std::vector<std::thread> pool;
    const uint64_t per = round_up(ceil_div(len, threads), 1ull << 20);
    for (unsigned t = 0; t < threads; t++) {
        const uint64_t a = off + std::min<uint64_t>((uint64_t)t * per, len), b = off + std::min<uint64_t>((uint64_t)(t + 1) * per, len);
        if (a < b) pool.emplace_back(work, a, b);
    }
    for (auto &th : pool) th.join();
    return err.load();
}
// ---- BerkeleyDB hash files: csrc/bigsi_bdb.hpp (shared with the CPU twin)
static inline bool bdb_row_key(const uint8_t *key, uint32_t len, uint64_t *row) { return bigsi_bdb_row_key(key, len, row); }

// bigsi_hip_bdb_small_records meets every row record on its way; an import calls bigsi_hip_load_rows_file on the same file next,
// which would scan it again: the row locations of the LAST file scanned are kept (one entry, identified by device / inode / size /
// mtime) and handed to the loader, which consumes them.
struct BdbScanCache {
    std::mutex mu;
    bool valid = false;
    dev_t dev = 0; ino_t ino = 0; off_t size = 0; struct timespec mtime {};
    std::vector<std::pair<uint64_t, BigsiBdb::Loc>> rows;      // (row id, where)
};
static BdbScanCache g_bdb_cache;

// The records of a BerkeleyDB hash file that are NOT rows -- the index integers and the sample metadata, a few bytes each --
// packed as [u32 key_len][u32 value_len][key][value]...; "<row>:bitarray" records are counted and measured only.
extern "C" int bigsi_hip_bdb_small_records(const char *path, uint8_t *out, uint64_t capacity, uint64_t *needed, uint64_t *n_rows, uint64_t *max_row_bytes,
                                           uint32_t threads)
{
    if (!path || !needed) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 4));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(BIGSI_ERR_INVALID, "%s: %s", path, strerror(errno));
    BigsiBdb db;
    int rc = db.open_fd(fd) ? fail(BIGSI_ERR_INVALID, "%s: %s", path, db.error.c_str()) : BIGSI_OK;
    struct Small { std::string key; BigsiBdb::Loc loc; };
    std::vector<Small> small;
    std::mutex mu;
    std::atomic<uint64_t> rows{0}, widest{0};
    std::vector<std::vector<std::pair<uint64_t, BigsiBdb::Loc>>> found(threads);      // row locations, per scanning thread
    // (the scan is independent 4 MB reads + a few hundred bytes of parsing per page: it takes more threads than the file <-> HBM pipeline)
    threads = std::max(threads, std::min(48u, std::max(1u, std::thread::hardware_concurrency() / 4)));
    found.resize(threads);
    if (rc == BIGSI_OK && db.scan(threads, [&](unsigned tid, const uint8_t *key, uint32_t klen, const BigsiBdb::Loc &l) {
            uint64_t r;
            if (bdb_row_key(key, klen, &r)) {
                rows++;
                uint64_t w = widest.load();
                while (l.len > w && !widest.compare_exchange_weak(w, l.len)) {}
                found[tid].emplace_back(r, l);
                return;
            }
            std::lock_guard<std::mutex> g(mu);
            small.push_back(Small{std::string(reinterpret_cast<const char *>(key), klen), l});
        }))
        rc = fail(BIGSI_ERR_INVALID, "%s: %s", path, db.error.c_str());
    if (rc == BIGSI_OK) {
        struct stat sb;
        std::lock_guard<std::mutex> g(g_bdb_cache.mu);
        g_bdb_cache.valid = false;
        g_bdb_cache.rows.clear();
        if (fstat(fd, &sb) == 0) {
            for (auto &v : found) { g_bdb_cache.rows.insert(g_bdb_cache.rows.end(), v.begin(), v.end()); std::vector<std::pair<uint64_t, BigsiBdb::Loc>>().swap(v); }
            g_bdb_cache.dev = sb.st_dev; g_bdb_cache.ino = sb.st_ino; g_bdb_cache.size = sb.st_size; g_bdb_cache.mtime = sb.st_mtim;
            g_bdb_cache.valid = true;
        }
    }
    uint64_t need = 0;
    if (rc == BIGSI_OK) {
        std::sort(small.begin(), small.end(), [](const Small &a, const Small &b) { return a.key < b.key; });      // (threads meet the pages in any order)
        for (auto &s_ : small) need += 8 + s_.key.size() + s_.loc.len;
        *needed = need;
        if (n_rows) *n_rows = rows.load();
        if (max_row_bytes) *max_row_bytes = widest.load();
        if (out && capacity < need) rc = fail(BIGSI_ERR_CAPACITY, "buffer holds %llu bytes, %llu needed", (unsigned long long)capacity, (unsigned long long)need);
        else if (out) {
            std::vector<uint8_t> page;
            uint8_t *q = out;
            for (auto &s_ : small) {
                const uint32_t kl = (uint32_t)s_.key.size(), vl = s_.loc.len;
                memcpy(q, &kl, 4); memcpy(q + 4, &vl, 4); memcpy(q + 8, s_.key.data(), kl);
                const int e = db.read_value(s_.loc, q + 8 + kl, vl, page);
                if (e) { rc = fail(BIGSI_ERR_INVALID, "%s: reading the value of %s: %s", path, s_.key.c_str(), e == EILSEQ ? "corrupt overflow chain" : strerror(e)); break; }
                q += 8 + kl + vl;
            }
        }
    }
    close(fd);
    return rc;
}

// ---- BigsiRowsFile (bigsi_internal.hpp): one file, or a directory of striped part files
static const uint64_t kStripeParts = 16, kStripeBytes = 4ull << 20;

int BigsiRowsFile::open_(const char *path, bool save, uint64_t file_offset_, uint64_t row_bytes_, uint64_t n_rows, uint64_t row0, unsigned threads)
{
    file_offset = file_offset_;
    row_bytes = row_bytes_;
    const size_t len = strlen(path);
    striped = len > 0 && path[len - 1] == '/';
    if (!striped) {
        fd = open(path, save ? (O_WRONLY | O_CREAT) : O_RDONLY, 0644);
        if (fd < 0) return fail(BIGSI_ERR_INVALID, "%s: %s", path, strerror(errno));
        if (!save && file_offset == 0 && BigsiBdb::is_bdb(fd)) {
            // a v0.3 BerkeleyDB store: one scan of its hash pages locates the "<row>:bitarray" records of the range; io() then
            // gathers rows from wherever they are (inline, or an overflow chain of pages)
            bdb = true;
            bdb_row0 = row0;
            int rc = db.open_fd(fd) ? fail(BIGSI_ERR_INVALID, "%s", db.error.c_str()) : BIGSI_OK;
            if (rc == BIGSI_OK) {
                loc.assign(n_rows, BigsiBdb::Loc{});
                bool cached = false;
                {
                    // the scan bigsi_hip_bdb_small_records has just made of this very file (same inode, size and mtime), if any
                    struct stat sb;
                    std::lock_guard<std::mutex> g(g_bdb_cache.mu);
                    if (g_bdb_cache.valid && fstat(fd, &sb) == 0 && sb.st_dev == g_bdb_cache.dev && sb.st_ino == g_bdb_cache.ino && sb.st_size == g_bdb_cache.size &&
                        sb.st_mtim.tv_sec == g_bdb_cache.mtime.tv_sec && sb.st_mtim.tv_nsec == g_bdb_cache.mtime.tv_nsec) {
                        for (auto &e : g_bdb_cache.rows)
                            if (e.first >= row0 && e.first - row0 < n_rows) loc[e.first - row0] = e.second;
                        cached = true;
                        if (row0 == 0 && n_rows >= g_bdb_cache.rows.size()) { g_bdb_cache.valid = false; std::vector<std::pair<uint64_t, BigsiBdb::Loc>>().swap(g_bdb_cache.rows); }      // consumed
                    }
                }
                if (!cached && db.scan(std::max(threads, std::min(48u, std::max(1u, std::thread::hardware_concurrency() / 4))), [&](unsigned, const uint8_t *key, uint32_t klen, const BigsiBdb::Loc &l) {
                        uint64_t r;
                        if (bdb_row_key(key, klen, &r) && r >= row0 && r - row0 < n_rows) loc[r - row0] = l;      // (one writer per row: a key occurs once)
                    }))
                    rc = fail(BIGSI_ERR_INVALID, "%s", db.error.c_str());
            }
            if (rc != BIGSI_OK) { char keep[512]; snprintf(keep, sizeof keep, "%s", bigsi_hip_last_error()); close_(); return fail(rc, "%s: %s", path, keep); }
        }
        return BIGSI_OK;
    }
    const std::string dir(path);
    const std::string lay = dir + "layout";
    if (save) {
        if (mkdir(dir.substr(0, len - 1).c_str(), 0755) != 0 && errno != EEXIST) return fail(BIGSI_ERR_INVALID, "mkdir %s: %s", path, strerror(errno));
        parts = kStripeParts;
        stripe_rows = std::max<uint64_t>(1, kStripeBytes / row_bytes);
        FILE *f = fopen(lay.c_str(), "w");
        if (!f) return fail(BIGSI_ERR_INVALID, "%s: %s", lay.c_str(), strerror(errno));
        fprintf(f, "bigsi-rows-striped 1\nparts %llu\nstripe_rows %llu\nrow_bytes %llu\nrows %llu\nfile_offset %llu\n", (unsigned long long)parts,
                (unsigned long long)stripe_rows, (unsigned long long)row_bytes, (unsigned long long)n_rows, (unsigned long long)file_offset);
        fclose(f);
    } else {
        FILE *f = fopen(lay.c_str(), "r");
        if (!f) return fail(BIGSI_ERR_INVALID, "%s: %s", lay.c_str(), strerror(errno));
        unsigned long long ver = 0, p_ = 0, s_ = 0, rb_ = 0, rows_ = 0, fo_ = 0;
        const int got = fscanf(f, "bigsi-rows-striped %llu parts %llu stripe_rows %llu row_bytes %llu rows %llu file_offset %llu", &ver, &p_, &s_, &rb_, &rows_, &fo_);
        fclose(f);
        if (got != 6 || ver != 1 || p_ == 0 || p_ > 4096 || s_ == 0) return fail(BIGSI_ERR_INVALID, "%s is not a striped rows layout", lay.c_str());
        if (rb_ != row_bytes || fo_ != file_offset || rows_ < n_rows)
            return fail(BIGSI_ERR_INVALID, "%s holds %llu rows of %llu bytes (offset %llu); asked for %llu rows of %llu bytes (offset %llu)", lay.c_str(), rows_, rb_, fo_,
                        (unsigned long long)n_rows, (unsigned long long)row_bytes, (unsigned long long)file_offset);
        parts = p_;
        stripe_rows = s_;
    }
    for (uint64_t p = 0; p < parts; p++) {
        char name[32];
        snprintf(name, sizeof name, "part.%03llu", (unsigned long long)p);
        // (a part file of an earlier, longer save must not keep its tail: the parts of a save that starts at offset 0 are cut to size)
        const int f = open((dir + name).c_str(), save ? (O_WRONLY | O_CREAT | (file_offset == 0 ? O_TRUNC : 0)) : O_RDONLY, 0644);
        if (f < 0) { const int e = errno; close_(); return fail(BIGSI_ERR_INVALID, "%s%s: %s", path, name, strerror(e)); }
        fds.push_back(f);
    }
    return BIGSI_OK;
}

uint64_t BigsiRowsFile::chunk_rows() const
{
    uint64_t per = std::max<uint64_t>(1, kBigsiIoChunkBytes / row_bytes);
    if (striped && per > stripe_rows * parts) per -= per % (stripe_rows * parts);      // whole rounds over the parts: every part file gets the same share of a chunk
    return per;
}

int BigsiRowsFile::io(bool write, uint8_t *buf, uint64_t rel_row, uint64_t n, unsigned threads) const
{
    if (bdb) {
        if (write) return EROFS;
        // rows of the store, zero-extended (or cut) to row_bytes; a row the store does not hold reads as zeros
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, ceil_div(n, 64)));
        std::atomic<int> err{0};
        std::atomic<uint64_t> next{0};
        auto work = [&]() {
            std::vector<uint8_t> page;
            for (;;) {
                const uint64_t a = next.fetch_add(64);
                if (a >= n || err.load()) return;
                for (uint64_t i = a; i < std::min(n, a + 64); i++) {
                    const BigsiBdb::Loc &l = loc[rel_row + i];
                    uint8_t *dst = buf + i * row_bytes;
                    const uint32_t take = l.kind ? (uint32_t)std::min<uint64_t>(l.len, row_bytes) : 0u;
                    if (take) { const int e = db.read_value(l, dst, take, page); if (e) { err.store(e); return; } }
                    if (take < row_bytes) memset(dst + take, 0, row_bytes - take);
                }
            }
        };
        if (T == 1) work();
        else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; t++) pool.emplace_back(work);
            for (auto &th : pool) th.join();
        }
        return err.load();
    }
    if (!striped) return bigsi_file_io(fd, write, buf, file_offset + rel_row * row_bytes, n * row_bytes, threads);
    // a thread owns the part files p == t (mod T): no two threads inside one inode
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(threads, parts), ceil_div(n, stripe_rows)));
    std::atomic<int> err{0};
    auto work = [&](unsigned t) {
        const uint64_t s0 = rel_row / stripe_rows, s1 = ceil_div(rel_row + n, stripe_rows);
        for (uint64_t s = s0; s < s1 && !err.load(); s++) {
            const uint64_t p = s % parts;
            if (p % T != t) continue;
            const uint64_t a = std::max(rel_row, s * stripe_rows), b = std::min(rel_row + n, (s + 1) * stripe_rows);
            uint64_t off = file_offset + ((s / parts) * stripe_rows + (a - s * stripe_rows)) * row_bytes, left = (b - a) * row_bytes;
            uint8_t *q = buf + (a - rel_row) * row_bytes;
            while (left && !err.load()) {
                const ssize_t r = write ? pwrite(fds[p], q, (size_t)left, (off_t)off) : pread(fds[p], q, (size_t)left, (off_t)off);
                if (r < 0) { if (errno == EINTR) continue; err.store(errno); break; }
                if (r == 0) { err.store(write ? EIO : ENODATA); break; }
                q += r; off += (uint64_t)r; left -= (uint64_t)r;
            }
        }
    };
    if (T == 1) { work(0); return err.load(); }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; t++) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
    return err.load();
}

int BigsiRowsFile::sync_all() const
{
    int e = 0;
    if (fd >= 0 && fsync(fd) != 0 && errno != EINVAL && errno != EROFS) e = errno;
    for (int f : fds)
        if (fsync(f) != 0 && errno != EINVAL && errno != EROFS) e = errno;
    return e;
}

void BigsiRowsFile::close_()
{
    if (fd >= 0) close(fd);
    for (int f : fds) close(f);
    fd = -1;
    fds.clear();
}

namespace {

int rows_file(bigsi_hip_index *ix, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes, uint32_t threads,
              bool save, bigsi_hip_io_stats *st)
{
    if (!ix || !path) return fail(BIGSI_ERR_INVALID, "NULL argument");
    const uint64_t stride = ix->stride_words * 8;
    if (row_bytes == 0 || (!save && row_bytes > stride))
        return fail(BIGSI_ERR_CAPACITY, "row_bytes %llu not in [1, %llu] (row stride)", (unsigned long long)row_bytes, (unsigned long long)stride);
    if (row0 > ix->m || n_rows > ix->m - row0) return fail(BIGSI_ERR_RANGE, "rows [%llu, +%llu) outside [0, %llu)", (unsigned long long)row0, (unsigned long long)n_rows, (unsigned long long)ix->m);
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 4));
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    BigsiRowsFile rf;
    TRY(rf.open_(path, save, file_offset, row_bytes, n_rows, row0, threads));
    const bool direct = row_bytes == stride;                         // the file holds the device layout: no kernel on the way
    const uint64_t per = rf.chunk_rows();
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    DevBuf dev[2];
    int rc = BIGSI_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    double io_s = 0;
    auto body = [&]() -> int {
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipHostMalloc(&pin[i], std::min(per, std::max<uint64_t>(n_rows, 1)) * row_bytes, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            if (!direct) TRY(dev[i].reserve(std::min(per, std::max<uint64_t>(n_rows, 1)) * row_bytes));
        }
        uint8_t *base = reinterpret_cast<uint8_t *>(ix->d_index);
        const uint64_t n_chunks = ceil_div(n_rows, per);
        // load:  read(c) | H2D(c) in flight while read(c+1) runs;   save:  D2H(c+1) in flight while write(c) runs
        for (uint64_t c = 0; c < n_chunks + (save ? 1 : 0); c++) {
            const int slot = (int)(c & 1);
            const uint64_t r0 = row0 + c * per, cn = c < n_chunks ? std::min(per, row0 + n_rows - r0) : 0;
            if (save) {
                if (c < n_chunks) {                                      // queue the download of chunk c
                    if (direct) HIP_TRY(hipMemcpyAsync(pin[slot], base + r0 * stride, cn * row_bytes, hipMemcpyDeviceToHost, ix->stream));
                    else {
                        hipLaunchKernelGGL(k_gather_run, dim3((unsigned)cn), dim3(kBlock), 0, ix->stream, base, stride, r0, dev[slot].as<uint8_t>(), row_bytes);
                        HIP_TRY(hipGetLastError());
                        HIP_TRY(hipMemcpyAsync(pin[slot], dev[slot].p, cn * row_bytes, hipMemcpyDeviceToHost, ix->stream));
                    }
                    HIP_TRY(hipEventRecord(ev[slot], ix->stream));
                }
                if (c > 0) {                                             // write chunk c - 1 while it comes down
                    const int ps = (int)((c - 1) & 1);
                    const uint64_t pr0 = row0 + (c - 1) * per, pn = std::min(per, row0 + n_rows - pr0);
                    HIP_TRY(hipEventSynchronize(ev[ps]));
                    const auto t0 = std::chrono::steady_clock::now();
                    const int e = rf.io(true, static_cast<uint8_t *>(pin[ps]), pr0 - row0, pn, threads);
                    io_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (e) return fail(BIGSI_ERR_INVALID, "writing %s: %s", path, strerror(e));
                }
            } else {
                HIP_TRY(hipEventSynchronize(ev[slot]));                  // (never recorded: returns at once) the copy that last used this buffer
                const auto t0 = std::chrono::steady_clock::now();
                const int e = rf.io(false, static_cast<uint8_t *>(pin[slot]), r0 - row0, cn, threads);
                io_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (e) return fail(BIGSI_ERR_INVALID, "reading %s: %s", path, e == ENODATA ? "file too short" : strerror(e));
                if (direct) HIP_TRY(hipMemcpyAsync(base + r0 * stride, pin[slot], cn * row_bytes, hipMemcpyHostToDevice, ix->stream));
                else {
                    HIP_TRY(hipMemcpyAsync(dev[slot].p, pin[slot], cn * row_bytes, hipMemcpyHostToDevice, ix->stream));
                    hipLaunchKernelGGL(k_scatter_run, dim3((unsigned)cn), dim3(kBlock), 0, ix->stream, base, stride, r0, dev[slot].as<uint8_t>(), row_bytes);
                    HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(ev[slot], ix->stream));
            }
        }
        HIP_TRY(hipStreamSynchronize(ix->stream));
        return BIGSI_OK;
    };
    rc = body();
    if (save && rc == BIGSI_OK) { const int e_ = rf.sync_all(); if (e_) rc = fail(BIGSI_ERR_INVALID, "fsync %s: %s", path, strerror(e_)); }
    char keep[1024] = "";
    if (rc != BIGSI_OK) snprintf(keep, sizeof keep, "%s", bigsi_hip_last_error());      // (the clean-up below must not overwrite the message)
    hipError_t e = hipStreamSynchronize(ix->stream);
    for (int i = 0; i < 2; i++) {
        if (pin[i]) e = hipHostFree(pin[i]);
        if (ev[i]) e = hipEventDestroy(ev[i]);
        dev[i].release();
    }
    (void)e;
    rf.close_();
    if (rc != BIGSI_OK) return fail(rc, "%s", keep);
    if (st) {
        st->bytes = n_rows * row_bytes;
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        st->file_seconds = io_s;
        st->threads = threads;
        st->direct = direct ? 1u : 0u;
    }
    return BIGSI_OK;
}
}   // namespace

extern "C" int bigsi_hip_load_rows_file(bigsi_hip_index *ix, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                                        uint32_t threads, bigsi_hip_io_stats *stats)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    return rows_file(ix, path, file_offset, row0, n_rows, row_bytes, threads, false, stats);
}

extern "C" int bigsi_hip_save_rows_file(bigsi_hip_index *ix, const char *path, uint64_t file_offset, uint64_t row0, uint64_t n_rows, uint64_t row_bytes,
                                        uint32_t threads, bigsi_hip_io_stats *stats)
{
    BIGSI_ENTER(ix);
    return rows_file(ix, path, file_offset, row0, n_rows, row_bytes, threads, true, stats);
}

extern "C" int bigsi_hip_clear(bigsi_hip_index *ix)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    HIP_TRY(hipMemsetAsync(ix->d_index, 0, (size_t)ix->m * ix->stride_words * 8, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_insert_column(bigsi_hip_index *ix, uint64_t col, const uint8_t *bloom)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    if (!ix || !bloom) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (col > ix->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu beyond num_cols %llu", (unsigned long long)col, (unsigned long long)ix->n_cols);
    if (col >= ix->cap_cols) return fail(BIGSI_ERR_CAPACITY, "column %llu beyond col_capacity %llu", (unsigned long long)col, (unsigned long long)ix->cap_cols);
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    const uint64_t nb = ceil_div(ix->m, 8);
    TRY(ix->stage.reserve(nb));
    HIP_TRY(hipMemcpyAsync(ix->stage.p, bloom, nb, hipMemcpyHostToDevice, ix->stream));
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(ix->m, kBlock), 256 * 8);
    hipLaunchKernelGGL(k_insert_column, dim3(grid), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, ix->m, col, ix->stage.as<uint8_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
    if (col == ix->n_cols) ix->n_cols++;
    return BIGSI_OK;
}

#define ev_begin bigsi_ev_begin
#define ev_end bigsi_ev_end

// n filters already on the device -> columns [col0, col0 + n): whole 64-column words through the tiled transpose
// (k_transpose_regs), the ragged head (up to the next multiple of 128 columns) and tail through k_insert_columns
static int transpose_device(bigsi_hip_index *ix, uint64_t col0, uint64_t n, const uint8_t *d_blooms, uint64_t bstride)
{
    const uint64_t nb = ceil_div(ix->m, 8), end = col0 + n;
    auto slow = [&](uint64_t ca, uint64_t cnt) -> int {
        if (!cnt) return BIGSI_OK;
        const uint64_t items = ix->m * (((ca + cnt - 1) >> 6) - (ca >> 6) + 1);
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(items, kBlock), 256 * 32);
        hipLaunchKernelGGL(k_insert_columns, dim3(grid), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, ix->m, ca, cnt,
                           d_blooms + (ca - col0) * bstride, bstride);
        HIP_TRY(hipGetLastError());
        return BIGSI_OK;
    };
    const uint64_t c_lo = std::min(end, round_up(col0, 128));
    uint64_t n_words = (end - c_lo) / 64, c_hi = c_lo + n_words * 64;
    static const int tiled = env_int("BIGSI_HIP_TRANSPOSE_TILED", 1);
    static const int tr_rg = env_int("BIGSI_HIP_TR_RG", 4), tr_cg = env_int("BIGSI_HIP_TR_CG", 1);      // XCD groups: A/B in profiles/r06_transpose_regs_ab.txt
    // k_transpose_regs<2>: 1024 rows x 1024 columns per workgroup -- whole 128-byte lines of the filters in (two consecutive loads per
    // lane), 128-byte runs of the rows out.  (RT = 1, 512 rows: 4.4-5.0 TB/s against 4.9-5.4, its half lines shared with another workgroup.)
    static const int tr_double = env_int("BIGSI_HIP_TR_DOUBLE", 1);
    bool regs = true;
    uint64_t rt = tr_double ? 2 : 1, ct = 2;
#ifdef BIGSI_HIP_TUNING
    static const int tr_regs = env_int("BIGSI_HIP_TR_REGS", 1), tr_wide = env_int("BIGSI_HIP_TR_WIDE", 1);      // 0: k_transpose_tiles<RT, CT> of rounds 2-6
    regs = tr_regs != 0;
    if (!regs) ct = tr_wide ? 2 : 1;
    static const int tr_cw = env_int("BIGSI_HIP_TR_CW", 1);      // 2: 2048-column tiles (256-byte row runs), 1024 threads, one workgroup per CU
    if (regs && tr_cw == 2) ct = 4;
#endif
    // APPENDING (nothing valid at or beyond `end`: the usual build) the tiled kernel also takes the ragged tail: it writes whole 128-byte
    // lines, zeros for the columns that have no filter -- what those bits of the row hold anyway -- instead of leaving up to 63 columns to
    // the column-at-a-time kernel and a partly written line to the memory (round 6: 1 M x 100 000 against 1 M x 98 304: -5 %)
    if (regs && end >= ix->n_cols && end > c_lo) {
        n_words = std::min<uint64_t>(round_up(ceil_div(end - c_lo, 64), 16), ix->stride_words - c_lo / 64);
        c_hi = end;
    }
    // supertiles of 1024 tiles: 32 wide, narrower (and higher) when the matrix has fewer tile columns than that
    const uint64_t tiles_c = ceil_div(n_words, 8 * ct);
    static const int tr_supw = env_int("BIGSI_HIP_TR_SUPW", kTransposeSuper);
    uint32_t sup_w = (uint32_t)tr_supw;
    while (sup_w > 1 && sup_w / 2 >= tiles_c) sup_w /= 2;
    const uint32_t sup_h = (uint32_t)(kTransposeSuper * kTransposeSuper) / sup_w, cg_eff = std::min<uint32_t>((uint32_t)tr_cg, sup_w);
    const uint64_t sup_blocks = ceil_div(ceil_div(ix->m, kTransposeTile * rt), sup_h) * ceil_div(tiles_c, sup_w) * (uint64_t)(kTransposeSuper * kTransposeSuper);
    if (!tiled || n_words == 0 || bstride % 16 || ((uintptr_t)d_blooms & 15u) || sup_blocks > 0x7FFFFFFFull) return slow(col0, n);
    TRY(slow(col0, c_lo - col0));
#define BIGSI_TR_ARGS                                                                                                          \
    dim3((unsigned)sup_blocks), dim3(kBlock * (unsigned)ct), 0, ix->stream, ix->d_index, ix->stride_words, ix->m, c_lo / 64, n_words,  \
        d_blooms + (c_lo - col0) * bstride, end - c_lo, bstride, nb, (uint32_t)tr_rg, cg_eff, sup_w
#define COMMA ,
#ifdef BIGSI_HIP_TUNING
    if (regs && ct == 4 && rt == 2) hipLaunchKernelGGL((k_transpose_regs<2 COMMA 2>), BIGSI_TR_ARGS);
    else if (regs && ct == 4) hipLaunchKernelGGL((k_transpose_regs<1 COMMA 2>), BIGSI_TR_ARGS);
    else
#endif
    if (regs && rt == 2) hipLaunchKernelGGL((k_transpose_regs<2>), BIGSI_TR_ARGS);
    else if (regs) hipLaunchKernelGGL((k_transpose_regs<1>), BIGSI_TR_ARGS);
#ifdef BIGSI_HIP_TUNING
    else {
        static const int skip = env_int("BIGSI_HIP_TR_SKIP", 0);
        static bool set = false;
        if (!set) { const uint32_t v = (uint32_t)skip; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tr_skip), &v, 4)); set = true; }
        if (rt == 2 && ct == 2) hipLaunchKernelGGL((k_transpose_tiles<2 COMMA 2>), BIGSI_TR_ARGS);
        else if (rt == 2) hipLaunchKernelGGL((k_transpose_tiles<2 COMMA 1>), BIGSI_TR_ARGS);
        else if (ct == 2) hipLaunchKernelGGL((k_transpose_tiles<1 COMMA 2>), BIGSI_TR_ARGS);
        else hipLaunchKernelGGL((k_transpose_tiles<1 COMMA 1>), BIGSI_TR_ARGS);
    }
#endif
#undef COMMA
#undef BIGSI_TR_ARGS
    HIP_TRY(hipGetLastError());
    return slow(c_hi, end - c_hi);
}

static int check_insert_columns(bigsi_hip_index *ix, uint64_t col0, uint64_t n, const void *blooms, uint64_t bloom_stride_bytes)
{
    if (!ix || (n && !blooms)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    const uint64_t nb = ceil_div(ix->m, 8);
    if (n && bloom_stride_bytes < nb) return fail(BIGSI_ERR_INVALID, "bloom_stride_bytes %llu < ceil(num_rows/8) = %llu", (unsigned long long)bloom_stride_bytes, (unsigned long long)nb);
    if (col0 > ix->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu beyond num_cols %llu", (unsigned long long)col0, (unsigned long long)ix->n_cols);
    if (col0 + n > ix->cap_cols) return fail(BIGSI_ERR_CAPACITY, "columns [%llu,%llu) beyond col_capacity %llu", (unsigned long long)col0, (unsigned long long)(col0 + n), (unsigned long long)ix->cap_cols);
    return BIGSI_OK;
}

extern "C" int bigsi_hip_insert_columns(bigsi_hip_index *ix, uint64_t col0, uint64_t n, const uint8_t *blooms, uint64_t bloom_stride_bytes)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    TRY(check_insert_columns(ix, col0, n, blooms, bloom_stride_bytes));
    if (n == 0) return BIGSI_OK;
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    // filters are staged a slab at a time at a 128-byte pitch -- the transpose reads a filter in 128-byte runs, and a run that
    // straddles two L2 lines is fetched twice by neighbouring tiles (round 6, TCC_EA0_RDREQ_128B: 1.32 x the filter bytes at a
    // 16-byte pitch, 1.0 x at this one) --: at least 512 of them when 2 GB allow it (one transpose tile is 512 columns wide),
    // otherwise about 256 MB worth
    // (a pitch that is a multiple of 4 KB puts the same line of all the filters of a tile on few memory channels: 2 M-row filters at a
    //  256 KB pitch transpose at 4.5-4.8 TB/s against 5.0-5.3 one line further apart)
    const uint64_t nb = ceil_div(ix->m, 8), pitch = round_up(nb, 128) + (round_up(nb, 128) % 4096 == 0 ? 128 : 0);
    const uint64_t stage_bytes = std::min<uint64_t>(std::max<uint64_t>(512 * pitch, 256ull << 20), 2048ull << 20);
    uint64_t per = std::max<uint64_t>(1, stage_bytes / pitch);
    if (per >= 128) per = per / 128 * 128;
    for (uint64_t c0 = 0; c0 < n; c0 += per) {
        const uint64_t cn = std::min(per, n - c0);
        TRY(ix->stage.reserve(cn * pitch));
        HIP_TRY(hipMemcpy2DAsync(ix->stage.p, pitch, blooms + c0 * bloom_stride_bytes, bloom_stride_bytes, nb, cn, hipMemcpyHostToDevice, ix->stream));
        TRY(transpose_device(ix, col0 + c0, cn, ix->stage.as<uint8_t>(), pitch));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    if (col0 + n > ix->n_cols) ix->n_cols = col0 + n;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_insert_columns_device(bigsi_hip_index *ix, uint64_t col0, uint64_t n, const void *d_blooms, uint64_t bloom_stride_bytes)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    TRY(check_insert_columns(ix, col0, n, d_blooms, bloom_stride_bytes));
    if (n == 0) return BIGSI_OK;
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    EventPair ep{};
    TRY(ev_begin(ix, &ep));
    TRY(transpose_device(ix, col0, n, (const uint8_t *)d_blooms, bloom_stride_bytes));
    TRY(ev_end(ix, &ep, ix->ev_tr));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    if (col0 + n > ix->n_cols) ix->n_cols = col0 + n;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_append_index(bigsi_hip_index *dst, const bigsi_hip_index *src)
{
    BIGSI_ENTER(dst);
    TRY(bigsi_writable(dst));
    if (!dst || !src) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (dst == src) return fail(BIGSI_ERR_INVALID, "cannot append an index to itself");
    if (dst->m != src->m) return fail(BIGSI_ERR_INVALID, "row counts differ (%llu vs %llu)", (unsigned long long)dst->m, (unsigned long long)src->m);
    if (dst->device != src->device) return fail(BIGSI_ERR_INVALID, "both indexes must live on the same device");
    if (src->n_cols == 0) return BIGSI_OK;
    TRY(bigsi_hip_reserve_cols(dst, dst->n_cols + src->n_cols));
    TRY(use_device(dst));
    TRY(quiesce_index(dst));
    HIP_TRY(hipStreamSynchronize(src->stream));
    const uint64_t per_row = ceil_div(dst->n_cols + src->n_cols, 64) - (dst->n_cols >> 6);
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(dst->m * per_row, kBlock), 256 * 64);
    hipLaunchKernelGGL(k_append_columns, dim3(grid), dim3(kBlock), 0, dst->stream, dst->d_index, dst->stride_words, dst->n_cols,
                       src->d_index, src->stride_words, src->n_cols, dst->m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(dst->stream));
    dst->n_cols += src->n_cols;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_get_column(bigsi_hip_index *ix, uint64_t col, uint8_t *out)
{
    BIGSI_ENTER(ix);
    if (!ix || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (col >= ix->cap_cols) return fail(BIGSI_ERR_RANGE, "column %llu beyond col_capacity", (unsigned long long)col);
    TRY(use_device(ix));
    const uint64_t nb = ceil_div(ix->m, 8);
    TRY(ix->stage.reserve(nb));
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(nb, kBlock), 256 * 8);
    hipLaunchKernelGGL(k_get_column, dim3(grid), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, ix->m, col, ix->stage.as<uint8_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ix->stage.p, nb, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

static int check_offsets(const uint64_t *offsets, uint32_t n_seqs)
{
    for (uint32_t i = 0; i < n_seqs; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(BIGSI_ERR_INVALID, "offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] > 0xFFFFFFF0ull) return fail(BIGSI_ERR_INVALID, "sequence %u longer than 2^32-16 bytes", i);
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_insert_kmers(bigsi_hip_index *ix, uint64_t col, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    if (!ix || !offsets || (n_seqs && !seqs && offsets[n_seqs] > offsets[0])) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    if (col >= ix->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu >= num_cols %llu", (unsigned long long)col, (unsigned long long)ix->n_cols);
    if (n_seqs == 0) return BIGSI_OK;
    TRY(check_offsets(offsets, n_seqs));
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    const uint64_t base = offsets[0], nbytes = offsets[n_seqs] - base;
    std::vector<uint64_t> rel(n_seqs + 1);
    for (uint32_t i = 0; i <= n_seqs; i++) rel[i] = offsets[i] - base;
    TRY(ix->stage.reserve(std::max<uint64_t>(nbytes, 1)));
    TRY(ix->stage_ids.reserve((n_seqs + 1) * 8ull));
    if (nbytes) HIP_TRY(hipMemcpyAsync(ix->stage.p, seqs + base, nbytes, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(hipMemcpyAsync(ix->stage_ids.p, rel.data(), (n_seqs + 1) * 8ull, hipMemcpyHostToDevice, ix->stream));
    hipLaunchKernelGGL(k_insert_kmers, dim3(n_seqs), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, ix->m, ix->h, col,
                       ix->stage.as<char>(), ix->stage_ids.as<uint64_t>(), k);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));   // rel[] and the staging buffers go out of scope
    return BIGSI_OK;
}

extern "C" int bigsi_hip_fill_synthetic(bigsi_hip_index *ix, uint64_t seed, uint64_t shard, uint32_t and_draws)
{
    BIGSI_ENTER(ix);
    TRY(bigsi_writable(ix));
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (and_draws == 0 || and_draws > 8) return fail(BIGSI_ERR_INVALID, "and_draws must be in [1,8]");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    hipLaunchKernelGGL(k_fill_synth, dim3(256 * 16), dim3(kBlock), 0, ix->stream, ix->d_index, ix->m, ix->stride_words, ix->n_cols, seed, shard, and_draws);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_bloom(int device, const char *kmers, uint64_t u, uint32_t k, uint64_t m, uint32_t h, uint32_t flags, uint8_t *out)
{
    if (!out || (u && !kmers)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0 || m == 0 || h == 0) return fail(BIGSI_ERR_INVALID, "k, m and h must be > 0");
    HIP_TRY(hipSetDevice(device));
    const uint64_t nwords = ceil_div(m, 32), nb = ceil_div(m, 8);
    uint32_t *d_bits = nullptr;
    char *d_km = nullptr;
    HIP_TRY(hipMalloc((void **)&d_bits, nwords * 4));
    hipError_t e = hipMemset(d_bits, 0, nwords * 4);
    if (e == hipSuccess && u) {
        e = hipMalloc((void **)&d_km, u * k);
        if (e == hipSuccess) e = hipMemcpy(d_km, kmers, u * k, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(u, kBlock), 256 * 8);
            hipLaunchKernelGGL(k_bloom, dim3(grid), dim3(kBlock), 0, 0, d_bits, m, h, d_km, u, k, (flags & BIGSI_BLOOM_RAW) != 0);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_bits, nb, hipMemcpyDeviceToHost);   // little-endian words == byte order
    hipError_t e2 = hipFree(d_bits); (void)e2;
    if (d_km) e2 = hipFree(d_km);
    if (e != hipSuccess) return fail(BIGSI_ERR_HIP, "bigsi_hip_bloom: %s", hipGetErrorString(e));
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ profiling events
int bigsi_ev_begin(bigsi_hip_index *ix, EventPair *p, hipStream_t st, bool row_and)
{
    if (!st) st = ix->stream;
    *p = EventPair{};
    if (!ix->profiling || (ix->profiling == 2 && !row_and)) return BIGSI_OK;
    if (ix->prof_every > 1 && row_and && (ix->prof_tick++ % ix->prof_every) != 0) return BIGSI_OK;      // sampled: this run goes untimed
    if (ix->ev_free.empty()) {
        EventPair n;
        HIP_TRY(hipEventCreate(&n.a));
        HIP_TRY(hipEventCreate(&n.b));
        ix->ev_free.push_back(n);
    }
    *p = ix->ev_free.back();
    ix->ev_free.pop_back();
    HIP_TRY(hipEventRecord(p->a, st));
    return BIGSI_OK;
}

int bigsi_ev_end(bigsi_hip_index *ix, EventPair *p, std::vector<EventPair> &dst, hipStream_t st, uint32_t launches)
{
    if (&dst == &ix->ev_and) ix->and_total += launches;
    if (!p->a) return BIGSI_OK;          // not being timed
    if (!st) st = ix->stream;
    HIP_TRY(hipEventRecord(p->b, st));
    p->launches = launches;
    dst.push_back(*p);
    return BIGSI_OK;
}

extern "C" int bigsi_hip_set_profiling(bigsi_hip_index *ix, int on)
{
    BIGSI_ENTER(ix);
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    ix->profiling = on >= 2 ? 2 : (on != 0);
    ix->prof_every = on > 2 ? (uint32_t)on : 1u;
    ix->prof_tick = 0;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_stats(bigsi_hip_index *ix, bigsi_hip_stats_t *out, int reset)
{
    BIGSI_ENTER(ix);
    if (!ix || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    HIP_TRY(hipStreamSynchronize(ix->pre_stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    auto sum = [&](std::vector<EventPair> &v, uint64_t *n, double *ms) -> int {
        *n = 0;
        *ms = 0;
        for (auto &p : v) {
            *n += p.launches;
            float t = 0;
            HIP_TRY(hipEventSynchronize(p.b));      // (pairs recorded on a communicator's or a caller's stream)
            HIP_TRY(hipEventElapsedTime(&t, p.a, p.b));
            *ms += t;
        }
        return BIGSI_OK;
    };
    TRY(sum(ix->ev_and, &out->and_launches, &out->and_ms));
    TRY(sum(ix->ev_km, &out->kmerize_launches, &out->kmerize_ms));
    TRY(sum(ix->ev_cp, &out->compact_launches, &out->compact_ms));
    TRY(sum(ix->ev_pr, &out->presence_launches, &out->presence_ms));
    TRY(sum(ix->ev_tr, &out->transpose_launches, &out->transpose_ms));
    TRY(sum(ix->ev_ex, &out->exchange_launches, &out->exchange_ms));
    out->presence_bytes = ix->presence_bytes;
    out->index_contiguous = ix->contiguous ? 1 : 0;
    out->and_launches_total = ix->and_total;
    out->read_launches_repeated = ix->fused_repeats;
    if (reset) {
        recycle_events(ix);
        ix->presence_bytes = 0;
        ix->and_total = 0;
        ix->fused_repeats = 0;
    }
    return BIGSI_OK;
}

// MEASUREMENT: bare row streams over this index's own matrix (k_probe_rows): GB/s of `n_queries` lists of `rows_per_query`
// rows each, uniform random (sorted = 0) or ascending (1), in launches of `wgs` four-wavefront workgroups (0: as the library
// sizes its own row-AND launches -- the smallest multiple of 256 workgroups with >= 1600 live wavefronts); median launch of
// `reps` passes after one warm pass, HIP events on the index stream.
extern "C" int bigsi_hip_probe_rows(bigsi_hip_index *ix, uint32_t rows_per_query, uint32_t n_queries, uint32_t sorted, uint32_t wgs, uint32_t reps,
                                    double *gbps, double *launch_ms)
{
    BIGSI_ENTER(ix);
    if (!ix || !gbps) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (rows_per_query == 0 || n_queries == 0 || reps == 0) return fail(BIGSI_ERR_INVALID, "rows_per_query, n_queries and reps must be > 0");
    if (ix->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    TRY(use_device(ix));
    TRY(quiesce_index(ix));
    const uint32_t wv = (uint32_t)ix->wv(), segs = (uint32_t)ceil_div(wv, 64 * kVec), live = (uint32_t)ceil_div(wv, 64 * kVec);
    if (wgs == 0) wgs = (uint32_t)round_up(ceil_div(1600ull * ceil_div(segs, 4), live), 256);
    // as many LIVE wavefronts per launch as a row-AND launch of `wgs` workgroups has (there a query's last workgroup is partly empty)
    const uint32_t q_per_launch = std::max<uint32_t>(wgs / (uint32_t)ceil_div(segs, 4), 1);
    if (n_queries < 2 * q_per_launch) n_queries = 2 * q_per_launch;
    DevBuf ids, out;
    int rc = ids.reserve((size_t)n_queries * rows_per_query * 8);
    if (rc == BIGSI_OK) rc = out.reserve((size_t)q_per_launch * segs * 64 * 16);
    std::vector<float> ms;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_probe_ids, dim3(1024), dim3(kBlock), 0, ix->stream, ids.as<uint64_t>(), (uint64_t)n_queries, rows_per_query, ix->m, sorted, 0x5EEDull + sorted);
        HIP_TRY(hipGetLastError());
        for (uint32_t rep = 0; rep <= reps; rep++)
            for (uint32_t q0 = 0; q0 + q_per_launch <= n_queries; q0 += q_per_launch) {
                const uint32_t waves = q_per_launch * segs, blocks = (uint32_t)ceil_div(waves, 4);
                HIP_TRY(hipEventRecord(e0, ix->stream));
                hipLaunchKernelGGL(k_probe_rows, dim3(blocks), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, wv, ids.as<uint64_t>(), rows_per_query, segs,
                                   q0, q0 + q_per_launch, reinterpret_cast<u64x2 *>(out.p));
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(e1, ix->stream));
                HIP_TRY(hipEventSynchronize(e1));
                float t = 0;
                HIP_TRY(hipEventElapsedTime(&t, e0, e1));
                if (rep) ms.push_back(t);
            }
        return BIGSI_OK;
    };
    if (rc == BIGSI_OK) rc = body();
    hipError_t e = hipSuccess;
    if (e0) e = hipEventDestroy(e0);
    if (e1) e = hipEventDestroy(e1);
    (void)e;
    ids.release();
    out.release();
    if (rc != BIGSI_OK) return rc;
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    *gbps = (double)q_per_launch * rows_per_query * wv * 8.0 / (med * 1e-3) / 1e9;
    if (launch_ms) *launch_ms = med;
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ batches
static uint64_t pow2_at_least(uint64_t x)
{
    uint64_t p = 2;
    while (p < x) p <<= 1;
    return p;
}

// (re)load a batch object with sequences: host-side prefix arrays, grow-only device buffers, H2D copies
// num_kmers | num_unique | min_kmers of the batch's n_seqs sequences, side by side in b->uniq (host_counts: one download)
static void point_uniq(bigsi_hip_batch *b)
{
    uint32_t *u = b->uniq.as<uint32_t>();
    b->num_kmers.point(u, b->n_seqs * 4ull);
    b->num_unique.point(u + b->n_seqs, b->n_seqs * 4ull);
    b->min_kmers.point(u + 2ull * b->n_seqs, b->n_seqs * 4ull);
}

static int pinned_reserve(void **p, size_t *cap, size_t bytes);
static int export_wait(bigsi_hip_batch *b);

// `deferred`: the offset tables and the sequences are staged in pinned memory the batch owns and go up at the start of the next
// run, on the stream that run uses -- no copy, no synchronisation here (the one-call and streaming entry points: a call is then
// one asynchronous upload, the kernels, one export kernel and ONE wait).  Otherwise (create / reload) the copy is made now on
// the upload stream and waited for.
static int batch_load(bigsi_hip_batch *b, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k, bool deferred = false)
{
    bigsi_hip_index *ix = b->ix;
    b->upload_deferred = false;
    b->zero_copy = false;
    const uint64_t base = offsets[0], nbytes = offsets[n_seqs] - base;
    b->n_seqs = n_seqs;
    b->k = k;
    b->ran = b->compacted = b->host_counts_valid = false;
    b->g_src = nullptr;
    b->max_pos = b->max_len = 0;
    b->seq_off.resize(n_seqs + 1);
    b->pos_off.resize(n_seqs + 1);
    b->tab_off.resize(n_seqs + 1);
    b->pos_off[0] = b->tab_off[0] = 0;
    for (uint32_t i = 0; i <= n_seqs; i++) b->seq_off[i] = offsets[i] - base;
    for (uint32_t i = 0; i < n_seqs; i++) {
        const uint64_t len = b->seq_off[i + 1] - b->seq_off[i];
        const uint64_t n = len >= k ? len - k + 1 : 0;
        b->pos_off[i + 1] = b->pos_off[i] + n;
        b->tab_off[i + 1] = b->tab_off[i] + pow2_at_least(2 * n);
        b->max_pos = std::max(b->max_pos, n);
        b->max_len = std::max(b->max_len, len);
    }
    b->total_pos = b->pos_off[n_seqs];
    const uint64_t T = std::max<uint64_t>(b->total_pos, 1);
    int rc = BIGSI_OK;
    auto R = [&](DevBuf &d, size_t bytes) { if (rc == BIGSI_OK) rc = d.reserve(bytes); };
    // upload arena: [seq_off | pos_off | tab_off | sequence bytes]; the three offset tables go up as ONE copy, together with the
    // sequences when those are short (a batch of reads: one copy instead of four), else the sequences straight from the caller
    const size_t ob = (n_seqs + 1) * 8ull;
    R(b->upload, 3 * ob + std::max<uint64_t>(nbytes, 1));
    R(b->first_pos, T * 4);
    R(b->pos_unique, T * 4);
    R(b->tmp, T * 4);
    R(b->rep, T * 4);
    R(b->rows, T * ix->h * 8);
    R(b->uniq, 3ull * n_seqs * 4);
    if (rc == BIGSI_OK) {
        uint8_t *u = b->upload.as<uint8_t>();
        b->d_seq_off.point(u, ob);
        b->d_pos_off.point(u + ob, ob);
        b->d_tab_off.point(u + 2 * ob, ob);
        b->seqs.point(u + 3 * ob, std::max<uint64_t>(nbytes, 1));
        point_uniq(b);
    }
    auto H2D = [&](void *dst, const void *src, size_t bytes) {
        if (rc == BIGSI_OK && bytes) {
            hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ix->pre_stream);
            if (e != hipSuccess) rc = fail(BIGSI_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
        }
    };
    if (deferred && rc == BIGSI_OK) {
        const size_t bytes = 3 * ob + nbytes;
        rc = pinned_reserve(&b->pin_up, &b->pin_up_cap, bytes);
        if (rc != BIGSI_OK) return rc;
        uint8_t *h = static_cast<uint8_t *>(b->pin_up);
        memcpy(h, b->seq_off.data(), ob);
        memcpy(h + ob, b->pos_off.data(), ob);
        memcpy(h + 2 * ob, b->tab_off.data(), ob);
        if (nbytes) memcpy(h + 3 * ob, seqs + base, nbytes);
        b->pin_up_bytes = bytes;
        b->upload_deferred = true;
        b->pos_query_loaded = false;
        return BIGSI_OK;
    }
    const bool packed = nbytes <= (64u << 10);
    b->h_upload.resize(3 * ob + (packed ? nbytes : 0));
    memcpy(b->h_upload.data(), b->seq_off.data(), ob);
    memcpy(b->h_upload.data() + ob, b->pos_off.data(), ob);
    memcpy(b->h_upload.data() + 2 * ob, b->tab_off.data(), ob);
    if (packed && nbytes) memcpy(b->h_upload.data() + 3 * ob, seqs + base, nbytes);
    H2D(b->upload.p, b->h_upload.data(), b->h_upload.size());
    if (!packed) H2D(b->seqs.p, seqs + base, nbytes);
    b->pos_query_loaded = false;
    if (rc == BIGSI_OK) {
        hipError_t e = hipStreamSynchronize(ix->pre_stream);  // the host vectors above are read by the copies
        if (e != hipSuccess) rc = fail(BIGSI_ERR_HIP, "sync: %s", hipGetErrorString(e));
    }
    return rc;
}

// host-side wait for everything queued on behalf of this batch (its K1 on the pre stream, K2-K4 on the index stream up to
// `done`, gathered compaction on the gather stream)
static int batch_quiesce(bigsi_hip_batch *b)
{
    if (b->idle) return BIGSI_OK;      // its last export was collected and nothing has been queued since
    if (b->done_stale && !b->dirty && b->run_stream) HIP_TRY(hipStreamSynchronize(b->run_stream));
    if (b->dirty) {
        HIP_TRY(hipStreamSynchronize(b->ix->pre_stream));
        HIP_TRY(hipStreamSynchronize(b->ix->stream));
        TRY(quiesce_reads(b->ix));
        b->dirty = false;
    } else if (b->done && !b->done_stale) {
        HIP_TRY(hipEventSynchronize(b->done));
    }
    if (b->g_done) HIP_TRY(hipEventSynchronize(b->g_done));
    if (b->job.done && b->job.device_work) HIP_TRY(hipEventSynchronize(b->job.done));      // a K5 / K6 request still reading this batch's arrays
    return BIGSI_OK;
}

static int check_batch_args(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    if (!ix || !offsets) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (n_seqs == 0) return fail(BIGSI_ERR_INVALID, "a batch needs at least one sequence");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    TRY(check_offsets(offsets, n_seqs));
    if (offsets[n_seqs] - offsets[0] && !seqs) return fail(BIGSI_ERR_INVALID, "seqs is NULL");
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_create(bigsi_hip_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k, bigsi_hip_batch **out)
{
    BIGSI_ENTER(ix);
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    TRY(check_batch_args(ix, seqs, offsets, n_seqs, k));
    TRY(use_device(ix));
    bigsi_hip_batch *b = new (std::nothrow) bigsi_hip_batch();
    if (!b) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    b->ix = ix;
    int rc = batch_load(b, seqs, offsets, n_seqs, k);
    if (rc != BIGSI_OK) { bigsi_hip_batch_destroy(b); return rc; }
    *out = b;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_create_elements(bigsi_hip_index *ix, const char *blob, const uint64_t *elem_offsets, const uint64_t *seq_elem_offsets,
                                               const uint32_t *pos_unique, const uint64_t *seq_pos_offsets, uint32_t n_seqs, bigsi_hip_batch **out)
{
    BIGSI_ENTER(ix);
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!ix || !elem_offsets || !seq_elem_offsets || !seq_pos_offsets) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (n_seqs == 0) return fail(BIGSI_ERR_INVALID, "a batch needs at least one sequence");
    const uint64_t e_base = seq_elem_offsets[0], n_elems = seq_elem_offsets[n_seqs] - e_base;
    const uint64_t p_base = seq_pos_offsets[0], n_pos = seq_pos_offsets[n_seqs] - p_base;
    if (n_pos && !pos_unique) return fail(BIGSI_ERR_INVALID, "pos_unique is NULL");
    if (n_pos > 0xFFFFFFF0ull) return fail(BIGSI_ERR_INVALID, "too many k-mer positions");
    const uint64_t b_base = elem_offsets[e_base], n_bytes = elem_offsets[e_base + n_elems] - b_base;
    if (n_bytes && !blob) return fail(BIGSI_ERR_INVALID, "blob is NULL");
    TRY(use_device(ix));
    bigsi_hip_batch *b = new (std::nothrow) bigsi_hip_batch();
    if (!b) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    b->ix = ix;
    b->n_seqs = n_seqs;
    b->k = 0;
    b->elements = true;
    b->pos_off.resize(n_seqs + 1);
    b->seq_off.resize(n_elems + 1);              // element byte offsets (relative)
    std::vector<uint64_t> eso(n_seqs + 1);
    std::vector<uint32_t> first(std::max<uint64_t>(n_pos, 1)), rep(std::max<uint64_t>(n_pos, 1));
    int rc = BIGSI_OK;
    for (uint64_t e = 0; e <= n_elems && rc == BIGSI_OK; e++) {
        if (e && elem_offsets[e_base + e] < elem_offsets[e_base + e - 1]) rc = fail(BIGSI_ERR_INVALID, "elem_offsets must be non-decreasing");
        b->seq_off[e] = elem_offsets[e_base + e] - b_base;
    }
    for (uint32_t i = 0; i <= n_seqs; i++) {
        b->pos_off[i] = seq_pos_offsets[i] - p_base;
        eso[i] = seq_elem_offsets[i] - e_base;
    }
    for (uint32_t i = 0; i < n_seqs && rc == BIGSI_OK; i++) {
        if (b->pos_off[i + 1] < b->pos_off[i] || eso[i + 1] < eso[i]) { rc = fail(BIGSI_ERR_INVALID, "offsets must be non-decreasing"); break; }
        const uint64_t n = b->pos_off[i + 1] - b->pos_off[i], u = eso[i + 1] - eso[i];
        if (u > n) { rc = fail(BIGSI_ERR_INVALID, "sequence %u lists %llu unique k-mers for %llu positions", i, (unsigned long long)u, (unsigned long long)n); break; }
        const uint32_t *pu = pos_unique + p_base + b->pos_off[i];
        uint32_t *fp = first.data() + b->pos_off[i], *rp = rep.data() + b->pos_off[i];
        for (uint64_t j = 0; j < u; j++) fp[j] = 0xFFFFFFFFu;
        for (uint64_t t = 0; t < n && rc == BIGSI_OK; t++) {
            if (pu[t] >= u) rc = fail(BIGSI_ERR_RANGE, "pos_unique[%llu] of sequence %u is %u, beyond its %llu unique k-mers", (unsigned long long)t, i, pu[t], (unsigned long long)u);
            else if (fp[pu[t]] == 0xFFFFFFFFu) fp[pu[t]] = (uint32_t)t;
        }
        for (uint64_t j = 0; j < u && rc == BIGSI_OK; j++)
            if (fp[j] == 0xFFFFFFFFu) rc = fail(BIGSI_ERR_INVALID, "unique k-mer %llu of sequence %u occurs at no position", (unsigned long long)j, i);
        for (uint64_t t = 0; t < n && rc == BIGSI_OK; t++) rp[t] = fp[pu[t]];
        b->max_pos = std::max(b->max_pos, n);
    }
    b->total_pos = n_pos;
    const uint64_t T = std::max<uint64_t>(n_pos, 1);
    auto R = [&](DevBuf &d, size_t bytes) { if (rc == BIGSI_OK) rc = d.reserve(bytes); };
    R(b->seqs, std::max<uint64_t>(n_bytes, 1));
    R(b->d_seq_off, (n_elems + 1) * 8);
    R(b->elem_seq_off, (n_seqs + 1) * 8ull);
    R(b->d_pos_off, (n_seqs + 1) * 8ull);
    R(b->first_pos, T * 4);
    R(b->pos_unique, T * 4);
    R(b->rep, T * 4);
    R(b->rows, T * ix->h * 8);
    R(b->uniq, 3ull * n_seqs * 4);
    if (rc == BIGSI_OK) point_uniq(b);
    auto H2D = [&](void *dst, const void *src, size_t bytes) {
        if (rc == BIGSI_OK && bytes) {
            hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ix->pre_stream);
            if (e != hipSuccess) rc = fail(BIGSI_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
        }
    };
    H2D(b->seqs.p, blob ? blob + b_base : nullptr, n_bytes);
    H2D(b->d_seq_off.p, b->seq_off.data(), (n_elems + 1) * 8);
    H2D(b->elem_seq_off.p, eso.data(), (n_seqs + 1) * 8ull);
    H2D(b->d_pos_off.p, b->pos_off.data(), (n_seqs + 1) * 8ull);
    H2D(b->first_pos.p, first.data(), n_pos * 4);
    H2D(b->pos_unique.p, pos_unique ? pos_unique + p_base : nullptr, n_pos * 4);
    H2D(b->rep.p, rep.data(), n_pos * 4);
    if (rc == BIGSI_OK) {
        hipError_t e = hipStreamSynchronize(ix->pre_stream);
        if (e != hipSuccess) rc = fail(BIGSI_ERR_HIP, "sync: %s", hipGetErrorString(e));
    }
    if (rc != BIGSI_OK) {
        char keep[1024];
        snprintf(keep, sizeof keep, "%s", bigsi_hip_last_error());
        bigsi_hip_batch_destroy(b);
        return fail(rc, "%s", keep);
    }
    *out = b;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_reload(bigsi_hip_batch *b, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (b->elements) return fail(BIGSI_ERR_STATE, "a batch of explicit k-mers cannot be reloaded: create a new one");
    TRY(check_batch_args(b->ix, seqs, offsets, n_seqs, k));
    TRY(use_device(b->ix));
    // only THIS batch's earlier work has to be over before its buffers are rewritten: other batches may still be running
    TRY(batch_quiesce(b));
    return batch_load(b, seqs, offsets, n_seqs, k);
}

extern "C" int bigsi_hip_batch_destroy(bigsi_hip_batch *b)
{
    BusyGuard busy_guard_(b ? b->ix : nullptr, /*wait=*/true);
    if (!b) return BIGSI_OK;
    hipError_t e = hipSetDevice(b->ix->device);
    e = hipStreamSynchronize(b->ix->pre_stream);
    e = hipStreamSynchronize(b->ix->stream);
    if (b->done) e = hipEventSynchronize(b->done);           // a run on one of the read streams
    if (b->exp_serial && b->exp_stream) e = hipStreamSynchronize(b->exp_stream);      // an export still writing into pin_out
    if (b->gstream && b->g_done) e = hipEventSynchronize(b->g_done);
    (void)e;
    b->planes.release();
    for (DevBuf *d : {&b->uniq, &b->upload, &b->pres_desc, &b->elem_seq_off, &b->pres_in, &b->pres_bits, &b->pres_out, &b->rows_sorted, &b->pos_query, &b->hsh, &b->rep, &b->seqs, &b->d_seq_off, &b->d_pos_off, &b->d_tab_off, &b->tab, &b->first_pos, &b->pos_unique, &b->tmp, &b->rows,
                      &b->num_kmers, &b->num_unique, &b->min_kmers, &b->bitmaps, &b->counts, &b->scratch})
        d->release();
    b->hits.release();
    b->ghits.release();
    b->gbuf.release();
    if (b->job.done) { e = hipEventSynchronize(b->job.done); e = hipEventDestroy(b->job.done); (void)e; }
    if (b->pin_up) { e = hipHostFree(b->pin_up); (void)e; }
    if (b->pin_out) { e = hipHostFree(b->pin_out); (void)e; }
    if (b->pin_flag) { e = hipHostFree(b->pin_flag); (void)e; }
    b->exp_count.release();
    if (b->exp_done) { e = hipEventDestroy(b->exp_done); (void)e; }
    if (b->job.h_in) { e = hipHostFree(b->job.h_in); (void)e; }
    if (b->job.h_out) { e = hipHostFree(b->job.h_out); (void)e; }
    if (b->done) { e = hipEventDestroy(b->done); (void)e; }
    if (b->k1_done) { e = hipEventDestroy(b->k1_done); (void)e; }
    if (b->g_done) { e = hipEventDestroy(b->g_done); (void)e; }
    delete b;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_set_result_cols(bigsi_hip_batch *b, uint64_t cols)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (cols > b->ix->cap_cols)
        return fail(BIGSI_ERR_CAPACITY, "result width %llu exceeds col_capacity %llu (call bigsi_hip_reserve_cols)", (unsigned long long)cols,
                    (unsigned long long)b->ix->cap_cols);
    b->result_cols = cols;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_set_outputs(bigsi_hip_batch *b, void *d_bitmaps, void *d_counts)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    b->ext_bitmaps = d_bitmaps;
    b->ext_counts = d_counts;
    return BIGSI_OK;
}

// -------- K2 dispatch
// one launch of the counting kernel over the queries [q0, q1) of the batch
struct CountLaunch {
    const uint64_t *k2_rows;
    unsigned block;
    uint32_t tiles;             // column tiles per query
    void *out;
    uint64_t out_stride;
    uint64_t *hit_bitmap;
    uint32_t sparse, slices;
    bool deep;                  // software-pipelined row loads (small grids; h = 3 or 4 only)
    uint32_t early_exit;        // BIGSI_RUN_EARLY_EXIT on a hits-only, one-slice run
    uint64_t *partial;          // slices > 1: bit-sliced partial counts of every slice (k_count_combine adds them up)
    uint32_t planes_out;
    bool half;                  // half the row loads in flight per lane (k_and_count<..., 3>)
    int vec;                    // 64-column words per lane: 2, or 1 (h = 3 / 4, one slice, not pipelined): `tiles` is computed for it
};

template <int P, typename CountT>
static void launch_count_wide(bigsi_hip_batch *b, const CountLaunch &c, uint32_t q0, uint32_t q1)
{
    bigsi_hip_index *ix = b->ix;
    // (sliced launches map workgroups to queries in plain order -- map_block -- and need no padding to 8 queries: a single sliced query
    // used to launch 8 x its workgroups, seven eighths of them leaving at once)
    const unsigned grid = (unsigned)((c.slices > 1 ? (uint64_t)(q1 - q0) : ceil_div(q1 - q0, 8) * 8) * (uint64_t)c.tiles * c.slices);
#define BIGSI_COUNT_ARGS                                                                                                      \
    dim3(grid), dim3(c.block), 0, ix->stream, ix->d_index, ix->stride_words, (uint32_t)b->wv, c.k2_rows, b->d_pos_off.as<uint64_t>(), \
        b->num_unique.as<uint32_t>(), ix->h, q0, q1, c.tiles, (CountT *)c.out, c.out_stride, b->min_kmers.as<uint32_t>(), ix->n_cols,  \
        c.hit_bitmap, b->wv_pad, c.sparse, c.slices, c.early_exit, c.partial, c.planes_out
#define COMMA ,
#define BIGSI_LAUNCH_COUNT(H) hipLaunchKernelGGL((k_and_count<P, H, CountT>), BIGSI_COUNT_ARGS)
#ifdef BIGSI_HIP_TUNING      // (the one-word-per-lane form: an A/B variant, not in the product library -- DESIGN.md section 7)
#define BIGSI_LAUNCH_COUNT_VEC1(H)                                                                                          \
    if (c.vec == 1) hipLaunchKernelGGL((k_and_count<P COMMA H COMMA CountT COMMA 1 COMMA 1>), BIGSI_COUNT_ARGS);          \
    else if (c.half) hipLaunchKernelGGL((k_and_count<P COMMA H COMMA CountT COMMA 3>), BIGSI_COUNT_ARGS);                  \
    else
#else
#define BIGSI_LAUNCH_COUNT_VEC1(H)
#endif
#define BIGSI_LAUNCH_COUNT_DEEP(H)                                                                        \
    if (c.deep) hipLaunchKernelGGL((k_and_count<P COMMA H COMMA CountT COMMA 2>), BIGSI_COUNT_ARGS);       \
    else BIGSI_LAUNCH_COUNT_VEC1(H) hipLaunchKernelGGL((k_and_count<P, H, CountT>), BIGSI_COUNT_ARGS)
    switch (ix->h) {
    case 1: BIGSI_LAUNCH_COUNT(1); break;
    case 2: BIGSI_LAUNCH_COUNT(2); break;
    case 3: BIGSI_LAUNCH_COUNT_DEEP(3); break;
    case 4: BIGSI_LAUNCH_COUNT_DEEP(4); break;
    case 5: BIGSI_LAUNCH_COUNT(5); break;
    default: BIGSI_LAUNCH_COUNT(0); break;
    }
#undef BIGSI_LAUNCH_COUNT
#undef BIGSI_LAUNCH_COUNT_DEEP
#undef BIGSI_LAUNCH_COUNT_VEC1
#undef BIGSI_COUNT_ARGS
#undef COMMA
}

static void launch_count(bigsi_hip_batch *b, int P, const CountLaunch &c, uint32_t q0, uint32_t q1)
{
    switch (P) {
    case 6: launch_count_wide<6, uint16_t>(b, c, q0, q1); break;
    case 10: launch_count_wide<10, uint16_t>(b, c, q0, q1); break;
    case 12: launch_count_wide<12, uint16_t>(b, c, q0, q1); break;
    case 16: launch_count_wide<16, uint16_t>(b, c, q0, q1); break;
    default: launch_count_wide<32, uint32_t>(b, c, q0, q1); break;
    }
}

static int compact(bigsi_hip_batch *b, HitBufs &hb, const void *src, uint32_t n_shards, uint64_t shard_cols, bool write_only);

// Reads against a narrow index: K1 + K2 + K4 in one launch (k_reads_fused) when the batch qualifies.
static bool reads_fusable(const bigsi_hip_batch *b, uint32_t flags)
{
    static const int fuse = env_int("BIGSI_HIP_FUSE_READS", 1);
    const bigsi_hip_index *ix = b->ix;
    const bool exact = b->exact;
    return fuse && b->k == 31 && b->total_pos > 0 && b->max_pos <= 63 && b->n_seqs <= kReadsMaxSeqs && b->wv <= (uint64_t)kBlock * kVec &&
           ix->h >= 2 && ix->h <= 4 && !b->ext_bitmaps && !b->ext_counts && b->result_cols == 0 &&
           !(flags & (BIGSI_RUN_SKIP_COMPACT | BIGSI_RUN_K1_GLOBAL | BIGSI_RUN_EARLY_EXIT | BIGSI_RUN_NO_SORT)) &&
           (exact || (flags & BIGSI_RUN_SPARSE_COUNTS));     // the fused kernel keeps counters in registers: hits only
}

#ifdef BIGSI_HIP_TUNING
uint64_t g_call_trace[16];
uint64_t g_call_last;
// tuning builds only: host time per phase of bigsi_hip_search_batch since the last reset (ns sums: stage, run, export, wait,
// collect; out[15] = calls)
extern "C" int bigsi_hip_debug_call_trace(uint64_t *out, int reset)
{
    if (out) memcpy(out, g_call_trace, sizeof g_call_trace);
    if (reset) memset(g_call_trace, 0, sizeof g_call_trace);
    return BIGSI_OK;
}
// tuning builds only (not declared in include/bigsi_hip.h): the phase timestamps of the last k_reads_fused launch, 8 per workgroup
extern "C" int bigsi_hip_debug_phases(bigsi_hip_index *ix, uint64_t *out, uint32_t n_groups)
{
    BIGSI_ENTER(ix);
    HIP_TRY(hipStreamSynchronize(ix->stream));
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), std::min<uint32_t>(n_groups, 1024u) * 64ull));
    return BIGSI_OK;
}
#endif

// Where K1 reads a load's tables and sequences.  Normally the device copies (create / reload / a flushed deferred load).  A one-call
// search whose input is small reads them straight from the pinned staging (zero copy: scripts/probe/latency_probe.hip prices a 64 KB
// hipMemcpyAsync ahead of a kernel at 23 us against 9.5 us of direct reads) and writes the device copy of pos_off -- all the later
// kernels need of them -- on its way.
struct K1Src {
    const char *seqs;
    const uint64_t *seq_off, *pos_off;
    uint64_t *pos_off_out;
    uint32_t one_len;          // > 0: the batch is one sequence of this length, read in place (no offset tables to fetch)
    // one_len > 0: the same bytes as the host sees them -- short enough, they travel in the kernel arguments instead (SeqArg)
    template <int N>
    SeqArg<N> by_value() const
    {
        SeqArg<N> a;
        memset(a.w, 0, sizeof a.w);
        memcpy(a.w, seqs, one_len);
        return a;
    }
};

static K1Src k1_src(const bigsi_hip_batch *b)
{
    if (b->zero_copy) {
        const uint8_t *h = static_cast<const uint8_t *>(b->pin_up);
        const size_t ob = (b->n_seqs + 1) * 8ull;
        return K1Src{reinterpret_cast<const char *>(h + 3 * ob), reinterpret_cast<const uint64_t *>(h), reinterpret_cast<const uint64_t *>(h + ob),
                     b->d_pos_off.as<uint64_t>(), b->n_seqs == 1 && b->max_len ? (uint32_t)b->max_len : 0u};
    }
    return K1Src{b->seqs.as<char>(), b->d_seq_off.as<uint64_t>(), b->d_pos_off.as<uint64_t>(), nullptr, 0u};
}

static int export_prepare(bigsi_hip_batch *b, hipStream_t st);

// `inline_export`: a one-call search of ONE read -- the kernel's only workgroup writes the caller's block and raises the flag itself
static int launch_reads_fused(bigsi_hip_batch *b, hipStream_t st = nullptr, bool inline_export = false)
{
    const uint32_t fp_mask = b->weak_fp ? 1u : ~0u;
    if (!st) st = b->ix->stream;
    bigsi_hip_index *ix = b->ix;
    HitBufs &hb = b->hits;
    TRY(b->bitmaps.reserve((size_t)b->n_seqs * b->wv_pad * 8));
    TRY(hb.hit_off.reserve((b->n_seqs + 2) * 8ull));          // (query-ordered offsets: filled by whoever orders the lists)
    if (hb.cap == 0 && !hb.xcol) {
        const uint64_t want = 1u << 16;
        TRY(hb.hit_col.reserve(want * 4));
        TRY(hb.hit_cnt.reserve(want * 4));
        hb.cap = want;
    }
    // per query: where its hits start in the hit buffers and how many they are; two allocation counters used alternately (a
    // launch zeroes the one its successor will use: launches of one batch never overlap)
    TRY(hb.q_start.reserve((size_t)b->n_seqs * 8));
    TRY(hb.q_cnt.reserve((size_t)b->n_seqs * 4));
    if (!hb.alloc.p) {
        TRY(hb.alloc.reserve(256));
        HIP_TRY(hipMemsetAsync(hb.alloc.p, 0, 256, st));
        hb.gen = 0;
    }
    // (the generation advances only once the launch is queued: a launch that fails -- export_prepare, hipGetLastError -- leaves the
    // counters as the last successful launch left them, i.e. this slot still zeroed for the retry)
    const uint32_t gen_next = hb.gen + 1;
    b->exported_inline = false;
    if (inline_export) {
        TRY(export_prepare(b, st));
        if (b->exp_flagged) b->exported_inline = true;
        else b->exp_serial--;          // (tuning builds with the event route: the export kernel as usual)
    }
    const K1Src src = k1_src(b);
    // one read read in place: its bytes go in the kernel arguments, not over the host link (SeqArg, bigsi_kernels.hpp)
    static const int arg_env = env_int("BIGSI_HIP_SEQ_BY_ARG", 1);
    const bool by_arg = arg_env && src.one_len && src.one_len <= kSeqArgRead;
    SeqArg<kSeqArgRead> read_arg;
    if (by_arg) read_arg = src.by_value<kSeqArgRead>();
    else memset(read_arg.w, 0, sizeof read_arg.w);
#define BIGSI_READS_ARGS                                                                                                          \
    dim3(b->n_seqs), dim3(kBlock), 0, st, ix->d_index, ix->stride_words, (uint32_t)b->wv, ix->n_cols, ix->m, b->threshold,              \
        by_arg ? (const char *)nullptr : src.seqs, src.seq_off, src.pos_off, b->n_seqs, b->first_pos.as<uint32_t>(),                        \
        b->pos_unique.as<uint32_t>(), b->rep.as<uint32_t>(), b->rows.as<uint64_t>(), b->num_kmers.as<uint32_t>(),                      \
        b->num_unique.as<uint32_t>(), b->min_kmers.as<uint32_t>(), b->bitmaps.as<uint64_t>(), b->wv_pad, hb.q_start.as<uint64_t>(),   \
        hb.q_cnt.as<uint32_t>(), hb.alloc.as<unsigned long long>(), gen_next & 1u, hb.col(), hb.cnt(), hb.capacity(), fp_mask,          \
        src.pos_off_out, src.one_len, b->exported_inline ? static_cast<uint64_t *>(b->pin_out) : nullptr, b->exp_spec,                          \
        (volatile uint64_t *)b->pin_flag, b->exp_serial, read_arg
#define COMMA ,
    // rows of at most kBlock words, counting: one word per lane (8-byte loads: half the registers -- 8 wavefronts per SIMD instead
    // of 4 -- and twice the lanes that hold columns).  Interleaved A/B at C2: counting 1377 -> 1523 M lookups/s; the exact kernel
    // (6 -> 8 wavefronts per SIMD) 1563 -> 1534 M: stays at two words per lane
    static const int vec1_exact = env_int("BIGSI_HIP_READS_VEC1_EXACT", 0), vec1_count = env_int("BIGSI_HIP_READS_VEC1_COUNT", 1);
    const bool narrow = b->wv <= (uint64_t)kBlock;
#ifdef BIGSI_HIP_TUNING
    static const int reads_unr = env_int("BIGSI_HIP_READS_UNROLL", 16);
#define BIGSI_READS_UNR(H)                                                                                                          \
    if (b->exact && reads_unr == 8) hipLaunchKernelGGL((k_reads_fused<H COMMA true COMMA kVec COMMA 8>), BIGSI_READS_ARGS);           \
    else if (b->exact && reads_unr == 12) hipLaunchKernelGGL((k_reads_fused<H COMMA true COMMA kVec COMMA 12>), BIGSI_READS_ARGS);    \
    else if (b->exact && reads_unr == 24) hipLaunchKernelGGL((k_reads_fused<H COMMA true COMMA kVec COMMA 24>), BIGSI_READS_ARGS);    \
    else
#else
#define BIGSI_READS_UNR(H)
#endif
    // a handful of reads (a latency-bound call): the exact route splits each query's rows over the two halves of its workgroup
    static const int split_max = env_int("BIGSI_HIP_READS_SPLIT_MAX", 64);
    const bool split = b->wv <= (uint64_t)kBlock && (int)b->n_seqs <= split_max;
#define BIGSI_READS(H)                                                                              \
    BIGSI_READS_UNR(H)                                                                              \
    if (split && b->exact) hipLaunchKernelGGL((k_reads_fused<H COMMA true COMMA kVec COMMA 16 COMMA true>), BIGSI_READS_ARGS);     \
    else if (split) hipLaunchKernelGGL((k_reads_fused<H COMMA false COMMA kVec COMMA 16 COMMA true>), BIGSI_READS_ARGS);          \
    else                                                                                            \
    if (b->exact && narrow && vec1_exact) hipLaunchKernelGGL((k_reads_fused<H COMMA true COMMA 1>), BIGSI_READS_ARGS);   \
    else if (b->exact) hipLaunchKernelGGL((k_reads_fused<H COMMA true>), BIGSI_READS_ARGS);                \
    else if (narrow && vec1_count) hipLaunchKernelGGL((k_reads_fused<H COMMA false COMMA 1>), BIGSI_READS_ARGS);       \
    else hipLaunchKernelGGL((k_reads_fused<H COMMA false>), BIGSI_READS_ARGS)
    switch (ix->h) {
    case 2: BIGSI_READS(2); break;
    case 3: BIGSI_READS(3); break;
    default: BIGSI_READS(4); break;
    }
#undef BIGSI_READS
#undef BIGSI_READS_UNR
#undef BIGSI_READS_ARGS
#undef COMMA
    HIP_TRY(hipGetLastError());
    hb.gen = gen_next;
    return BIGSI_OK;
}

// the stream K1 and the row sort run on.  Default: the index stream itself.  BIGSI_HIP_K1_OVERLAP=1 moves them to the pre
// stream so that they overlap the row-AND kernel of the batch before; measured a LOSS at C3 (exact: K2 1.93 -> 2.12-2.22 ms,
// the late-placed workgroups break the lock-step sweep of k_and_exact; counts: +-0), kept for A/B runs only
static hipStream_t k1_stream(const bigsi_hip_index *ix)
{
    static const int overlap = env_int("BIGSI_HIP_K1_OVERLAP", 0);
    return overlap ? ix->pre_stream : ix->stream;
}

// see bigsi_internal.hpp; the same arithmetic as the launch rule in bigsi_batch_run below (256-thread workgroups, one slice)
uint32_t bigsi_exact_launch_queries(const bigsi_hip_index *ix)
{
    const uint64_t wv = ceil_div(ix->n_cols, 64);
    if (wv == 0) return 8;
    const uint64_t tiles = ceil_div(wv, (uint64_t)256 * kVec), waves_per_q = ceil_div(wv, (uint64_t)64 * kVec);
    const uint64_t kb = round_up(ceil_div((uint64_t)1600 * tiles, waves_per_q), 256);
    return (uint32_t)std::max<uint64_t>(8, (kb / tiles) / 8 * 8);
}

enum K1Route { K1_ELEMENTS, K1_WAVE, K1_LDS, K1_GLOBAL };
struct K1Plan {
    K1Route route;
    uint32_t hs_cap = 0, sq_bytes = 0, tab_mult = 4, tab_cap = 2;
    size_t lds = 0;
};

static K1Plan k1_plan(const bigsi_hip_batch *b, bool force_global)
{
    K1Plan p;
    static const int k1_global = env_int("BIGSI_HIP_K1_GLOBAL", 0);
    static const int k1_wave = env_int("BIGSI_HIP_K1_WAVE", 1);
    if (b->elements) { p.route = K1_ELEMENTS; return p; }
    if (!force_global && !k1_global && k1_wave && b->max_pos <= 64) { p.route = K1_WAVE; return p; }
    // dedupe table of the LDS route: 4 slots per position when that fits the LDS window (shorter probe chains), else 2
    p.hs_cap = (uint32_t)round_up(std::max<uint64_t>(b->max_pos, 1), 4);
    p.sq_bytes = (uint32_t)round_up(b->max_len + 16, 16);
    // (a handful of queries -- a latency-bound call -- have the LDS to themselves: 8 slots per position, insert phase of one 1 kbp
    // query 2.04 / 1.08 / 0.80 us at 2 / 4 / 8)
    static const int tab_mult_env = env_int("BIGSI_HIP_K1_TABMULT", 0);      // A/B: 2 = half the LDS per workgroup, longer probe chains
    p.tab_mult = tab_mult_env == 2 ? 2u : tab_mult_env == 4 ? 4u : (tab_mult_env == 8 || b->n_seqs <= 32) ? 8u : 4u;
    for (;; p.tab_mult /= 2) {
        p.tab_cap = 2;
        while (p.tab_cap < p.tab_mult * b->max_pos && p.tab_cap < (1u << 30)) p.tab_cap <<= 1;
        p.lds = (size_t)(p.tab_cap + p.tab_cap / 32 + 4) * 4 + 64 + (size_t)p.hs_cap * 4 + 2 * p.sq_bytes;      // table (+ sort pad) | scan | fingerprints | sequence | its complement
        if (p.lds <= 60 * 1024 || p.tab_mult == 2) break;
    }
    // fused single-launch K1 (dedupe table + sequence in LDS) when every query fits the default 64 KiB dynamic-LDS window
    p.route = (!force_global && !k1_global && b->max_pos <= kLdsMaxPos && p.lds <= 60 * 1024) ? K1_LDS : K1_GLOBAL;
    return p;
}

// K1 for the whole batch (h may have changed since create: the rows buffer is sized for it here)
struct Preset {              // result words K1 sets for the sliced row-AND launches of a small batch (see k_kmerize_lds)
    uint64_t *p = nullptr;
    uint64_t words = 0, value = 0;
    bool done = false;       // the K1 route taken did it (the others leave it to a memset)
};

static int run_kmerize(bigsi_hip_batch *b, double threshold, bool force_global = false, bool want_sorted = false, bool *sorted = nullptr,
                       Preset *preset = nullptr)
{
    uint64_t *ps_p = preset ? preset->p : nullptr;
    const uint64_t ps_words = preset ? preset->words : 0, ps_value = preset ? preset->value : 0;
    bigsi_hip_index *ix = b->ix;
    EventPair ep{};
    hipStream_t ks = k1_stream(ix);
    // K1 rewrites arrays the previous run of THIS batch may still be reading (K2/K4 on the index stream, a gathered
    // compaction on the gather stream); other batches' kernels are not waited for -- that is the overlap
    if (ks != ix->stream) {
        if (b->dirty) {
            HIP_TRY(hipStreamSynchronize(ix->stream));
            b->dirty = false;
        } else if (b->done) {
            HIP_TRY(hipStreamWaitEvent(ks, b->done, 0));
        }
    }
    // (on the index stream itself the callers' own ordering applies, as for every other entry point)
    if (b->g_done && b->gstream && b->gstream != ks) HIP_TRY(hipStreamWaitEvent(ks, b->g_done, 0));
    TRY(b->rows.reserve(std::max<uint64_t>(b->total_pos, 1) * ix->h * 8));
    const K1Plan plan = k1_plan(b, force_global);
    const K1Src src = k1_src(b);
    if (plan.route == K1_ELEMENTS) {       // explicit k-mers: only the hashing is left of K1
        TRY(ev_begin(ix, &ep, ks));
        hipLaunchKernelGGL(k_rows_raw, dim3(b->n_seqs), dim3(kBlock), 0, ks, b->seqs.as<char>(), b->d_seq_off.as<uint64_t>(), b->elem_seq_off.as<uint64_t>(),
                           b->d_pos_off.as<uint64_t>(), ix->h, ix->m, threshold, b->rows.as<uint64_t>(), b->num_kmers.as<uint32_t>(),
                           b->num_unique.as<uint32_t>(), b->min_kmers.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        TRY(ev_end(ix, &ep, ix->ev_km, ks));
        b->run_h = ix->h;
        return BIGSI_OK;
    }
    if (plan.route == K1_WAVE) {
        // probe / read-length queries: one wavefront per query, no LDS, no atomics
        TRY(ev_begin(ix, &ep, ks));
        const unsigned grid = (unsigned)ceil_div(b->n_seqs, kBlock / 64);
#define BIGSI_K1_WAVE(KF)                                                                                                      \
    hipLaunchKernelGGL((k_kmerize_wave<KF>), dim3(grid), dim3(kBlock), 0, ks, src.seqs, src.seq_off,                               \
                       src.pos_off, b->k, ix->h, ix->m, threshold, b->n_seqs, b->first_pos.as<uint32_t>(),                               \
                       b->pos_unique.as<uint32_t>(), b->rep.as<uint32_t>(), b->rows.as<uint64_t>(), b->num_kmers.as<uint32_t>(),           \
                       b->num_unique.as<uint32_t>(), b->min_kmers.as<uint32_t>(), ps_p, ps_words, ps_value, src.pos_off_out)
        if (b->k == 31) BIGSI_K1_WAVE(31);
        else BIGSI_K1_WAVE(0);
#undef BIGSI_K1_WAVE
        if (preset) preset->done = ps_p != nullptr;
        HIP_TRY(hipGetLastError());
        TRY(ev_end(ix, &ep, ix->ev_km, ks));
        b->run_h = ix->h;
        return BIGSI_OK;
    }
    const uint32_t hs_cap = plan.hs_cap, sq_bytes = plan.sq_bytes, tab_mult = plan.tab_mult, tab_cap = plan.tab_cap;
    const size_t lds = plan.lds;
    if (plan.route == K1_LDS) {
        // one thread per position for small batches (latency); once there are several queries per CU anyway, smaller
        // workgroups that loop over the positions let more queries overlap their barrier-separated phases
        static const int k1_block_env = env_int("BIGSI_HIP_K1_BLOCK", 0);
        const uint32_t block_cap = k1_block_env > 0 ? (uint32_t)k1_block_env : (b->n_seqs >= 1024 ? 256u : 1024u);
        uint32_t block = 64;
        while (block < b->max_pos && block < block_cap) block <<= 1;
        if (want_sorted) {
            TRY(b->rows_sorted.reserve(std::max<uint64_t>(b->total_pos, 1) * ix->h * 8));
            if (sorted) *sorted = true;
        }
        TRY(ev_begin(ix, &ep, ks));
        // a handful of gene-length queries (a latency-bound call): the hashing of a query's unique k-mers is spread over several
        // workgroups (k_kmerize_lds, `parts`)
        static const int parts_env = env_int("BIGSI_HIP_K1_PARTS", 8);
        const uint32_t parts = (parts_env > 1 && !(parts_env & (parts_env - 1)) && !want_sorted && b->max_pos <= block && b->max_pos >= 256 && (uint64_t)b->n_seqs * parts_env <= 256) ? (uint32_t)parts_env : 1u;
#define BIGSI_K1_LDS_(KF, ARGB, SARG)                                                                                           \
    hipLaunchKernelGGL((k_kmerize_lds<KF, ARGB>), dim3(b->n_seqs * parts), dim3(block), lds, ks, src.seqs, src.seq_off,           \
                       src.pos_off, b->k, ix->h, ix->m, threshold, tab_cap, tab_mult, hs_cap, sq_bytes, b->first_pos.as<uint32_t>(), b->tmp.as<uint32_t>(), \
                       b->pos_unique.as<uint32_t>(), b->rep.as<uint32_t>(), b->rows.as<uint64_t>(), b->num_kmers.as<uint32_t>(),      \
                       b->num_unique.as<uint32_t>(), b->min_kmers.as<uint32_t>(), want_sorted ? b->rows_sorted.as<uint64_t>() : (uint64_t *)nullptr, \
                       ps_p, ps_words, ps_value, src.pos_off_out, src.one_len, parts, SARG)
        // one sequence read in place (a one-call search): short enough, its bytes go in the kernel arguments (SeqArg)
        static const int arg_env = env_int("BIGSI_HIP_SEQ_BY_ARG", 1);
#define BIGSI_K1_LDS(KF)                                                                                   \
    if (arg_env && src.one_len && src.one_len <= kSeqArgSmall) BIGSI_K1_LDS_(KF, kSeqArgSmall, src.by_value<kSeqArgSmall>());       \
    else if (arg_env && src.one_len && src.one_len <= kSeqArgLarge) BIGSI_K1_LDS_(KF, kSeqArgLarge, src.by_value<kSeqArgLarge>());  \
    else BIGSI_K1_LDS_(KF, 0, SeqArg<0>{})
        if (b->k == 31) BIGSI_K1_LDS(31);
        else BIGSI_K1_LDS(0);
#undef BIGSI_K1_LDS
#undef BIGSI_K1_LDS_
        if (preset) preset->done = ps_p != nullptr;
        HIP_TRY(hipGetLastError());
        TRY(ev_end(ix, &ep, ix->ev_km, ks));
        b->run_h = ix->h;
        return BIGSI_OK;
    }
    {   // the multi-launch route's scratch (allocated lazily: batches of short queries never need it)
        const uint64_t Tn = std::max<uint64_t>(b->total_pos, 1);
        TRY(b->tab.reserve(b->tab_off[b->n_seqs] * 4));
        TRY(b->hsh.reserve(Tn * 4));
        if (b->pos_query.cap < Tn * 4 || !b->pos_query_loaded) {
            TRY(b->pos_query.reserve(Tn * 4));
            std::vector<uint32_t> pq(b->total_pos);
            for (uint32_t i = 0; i < b->n_seqs; i++) std::fill(pq.begin() + b->pos_off[i], pq.begin() + b->pos_off[i + 1], i);
            if (b->total_pos) HIP_TRY(hipMemcpy(b->pos_query.p, pq.data(), b->total_pos * 4, hipMemcpyHostToDevice));
            b->pos_query_loaded = true;
        }
    }
    HIP_TRY(hipMemsetAsync(b->tab.p, 0xFF, b->tab_off[b->n_seqs] * 4, ks));
    TRY(ev_begin(ix, &ep, ks));
    const uint64_t T = b->total_pos;
    const unsigned pgrid = (unsigned)ceil_div(std::max<uint64_t>(T, 1), kBlock);
#define BIGSI_K1_INSERT(KF)                                                                                                   \
    hipLaunchKernelGGL((k_kmer_insert<KF>), dim3(pgrid), dim3(kBlock), 0, ks, b->seqs.as<char>(), b->d_seq_off.as<uint64_t>(), \
                       b->d_pos_off.as<uint64_t>(), b->pos_query.as<uint32_t>(), b->d_tab_off.as<uint64_t>(), b->tab.as<uint32_t>(), \
                       b->k, T, b->hsh.as<uint32_t>())
#define BIGSI_K1_ROWS(KF)                                                                                                     \
    hipLaunchKernelGGL((k_kmer_rows<KF>), dim3(pgrid), dim3(kBlock), 0, ks, b->seqs.as<char>(), b->d_seq_off.as<uint64_t>(), \
                       b->d_pos_off.as<uint64_t>(), b->pos_query.as<uint32_t>(), b->rep.as<uint32_t>(), b->tmp.as<uint32_t>(), b->k, \
                       ix->h, ix->m, T, b->rows.as<uint64_t>())
    if (T) {
        if (b->k == 31) BIGSI_K1_INSERT(31);
        else BIGSI_K1_INSERT(0);
        hipLaunchKernelGGL(k_kmer_resolve, dim3(pgrid), dim3(kBlock), 0, ks, b->seqs.as<char>(), b->d_seq_off.as<uint64_t>(),
                           b->d_pos_off.as<uint64_t>(), b->pos_query.as<uint32_t>(), b->d_tab_off.as<uint64_t>(), b->tab.as<uint32_t>(),
                           b->k, T, b->hsh.as<uint32_t>(), b->rep.as<uint32_t>());
    }
    hipLaunchKernelGGL(k_kmer_rank, dim3(b->n_seqs), dim3(kBlock), 0, ks, b->d_seq_off.as<uint64_t>(), b->d_pos_off.as<uint64_t>(),
                       b->rep.as<uint32_t>(), b->k, threshold, b->first_pos.as<uint32_t>(), b->tmp.as<uint32_t>(), b->pos_unique.as<uint32_t>(),
                       b->num_kmers.as<uint32_t>(), b->num_unique.as<uint32_t>(), b->min_kmers.as<uint32_t>());
    if (T) {
        if (b->k == 31) BIGSI_K1_ROWS(31);
        else BIGSI_K1_ROWS(0);
    }
#undef BIGSI_K1_INSERT
#undef BIGSI_K1_ROWS
    HIP_TRY(hipGetLastError());
    TRY(ev_end(ix, &ep, ix->ev_km, ks));
    b->run_h = ix->h;
    return BIGSI_OK;
}

// K1's outputs become visible to the index stream (K2, K4, lookups): it waits for the pre stream's work of this batch
static int k1_publish(bigsi_hip_batch *b)
{
    bigsi_hip_index *ix = b->ix;
    if (k1_stream(ix) == ix->stream) return BIGSI_OK;       // same stream: already ordered
    if (!b->k1_done) HIP_TRY(hipEventCreateWithFlags(&b->k1_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->k1_done, k1_stream(ix)));
    HIP_TRY(hipStreamWaitEvent(ix->stream, b->k1_done, 0));
    return BIGSI_OK;
}

// a deferred load (batch_load): the staged tables and sequences go up now, ahead of this run's first kernel on its stream
static int flush_upload(bigsi_hip_batch *b, hipStream_t st, bool k1_reads_host = false)
{
    if (!b->upload_deferred) return BIGSI_OK;
    // a one-call search with a small input: this run's K1 (one of the single-launch routes) reads pin_up itself
    static const int zc_env = env_int("BIGSI_HIP_ZERO_COPY", 1);
    b->zero_copy = zc_env && k1_reads_host && b->one_call && b->pin_up_bytes <= kZeroCopyBytes;
    // ... unless many workgroups would each fetch their own few bytes over the link: then one small kernel copies the staging
    // with wide loads first (k_stage_in), and K1 reads the device arrays
    static const int stage_min = env_int("BIGSI_HIP_STAGE_KERNEL_MIN", 8 << 10);
    if (b->zero_copy && b->n_seqs > 1 && b->pin_up_bytes >= (size_t)stage_min) {
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(b->pin_up_bytes, (uint64_t)kBlock * 16), 64);
        hipLaunchKernelGGL(k_stage_in, dim3(grid), dim3(kBlock), 0, st, static_cast<const uint8_t *>(b->pin_up), b->upload.as<uint8_t>(), (uint64_t)b->pin_up_bytes);
        HIP_TRY(hipGetLastError());
        b->zero_copy = false;
    } else if (!b->zero_copy) HIP_TRY(hipMemcpyAsync(b->upload.p, b->pin_up, b->pin_up_bytes, hipMemcpyHostToDevice, st));
    b->upload_deferred = false;
    return BIGSI_OK;
}

// `one_call`: the caller is bigsi_hip_search_batch, which waits for this run through the export's flag on the same stream: the
// completion event is not recorded (one HIP call less on a path whose device work is a few microseconds); everything that would
// wait for it waits for the stream instead (done_stale).
int bigsi_batch_run(bigsi_hip_batch *b, double threshold, uint32_t flags, bool one_call)
{
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!(threshold <= 1.0)) return fail(BIGSI_ERR_INVALID, "threshold must be <= 1 (bigsi/graph/bigsi.py:176), got %g", threshold);
    bigsi_hip_index *ix = b->ix;
    // (a shard of a wider index may be empty: its result vectors, result_cols wide, are then all zero)
    if (ix->n_cols == 0 && b->result_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    TRY(use_device(ix));
    const bool was_idle = b->idle;      // nothing of this batch in flight (collected since its last run): no waits to queue
    b->idle = false;
    b->ran = false;
    b->host_counts_valid = false;
    b->run_serial++;
    b->threshold = threshold;
    b->exact = (threshold == 1.0) && !(flags & BIGSI_RUN_FORCE_COUNTS);
    // result vectors as wide as the index, or as the shard width agreed by a group of column shards (uneven shards then
    // still exchange buffers of one geometry; words beyond this shard's num_cols are zero through valid_mask)
    b->wv = ceil_div(std::max(ix->n_cols, b->result_cols), 64);
    if (b->wv > ix->stride_words)
        return fail(BIGSI_ERR_CAPACITY, "result width %llu columns exceeds the row stride (call bigsi_hip_reserve_cols)", (unsigned long long)b->result_cols);
    b->wv_pad = round_up(b->wv, 2);

    b->fused_run = false;
    b->exported_inline = false;
    if (reads_fusable(b, flags)) {
        // (K1 rewrites arrays the previous run of this batch may still be reading, on a read stream or the gather stream)
        TRY(b->rows.reserve(std::max<uint64_t>(b->total_pos, 1) * ix->h * 8));
        EventPair fe{};
        b->dirty = true;
        hipStream_t st = ix->stream;
        if (!(flags & BIGSI_RUN_ONE_STREAM)) TRY(read_stream(ix, &st));
        else if (ix->rd_pending) TRY(quiesce_reads(ix));      // (alone on the device: nothing of the read streams beside it)
        // (the compaction kernel of a gene-length batch waits between workgroups too: such a run is over before read kernels start)
        if (ix->main_ev && st != ix->stream) HIP_TRY(hipStreamWaitEvent(st, ix->main_ev, 0));
        if (!was_idle) {
            if (b->done_stale && b->run_stream && b->run_stream != st) HIP_TRY(hipStreamSynchronize(b->run_stream));
            else if (b->done && b->run_stream != st) HIP_TRY(hipStreamWaitEvent(st, b->done, 0));      // this batch's previous run
            if (b->g_done && b->gstream && b->gstream != st) HIP_TRY(hipStreamWaitEvent(st, b->g_done, 0));
            // a K5 / K6 request of this batch still in flight on another stream reads what K1 is about to rewrite
            if (b->job.done && b->job.device_work) HIP_TRY(hipStreamWaitEvent(st, b->job.done, 0));
        }
        TRY(flush_upload(b, st, true));
        TRY(ev_begin(ix, &fe, st, true));
        b->weak_fp = (flags & BIGSI_RUN_WEAK_FINGERPRINT) != 0;
        static const int inline_env = env_int("BIGSI_HIP_INLINE_EXPORT", 1);
        TRY(launch_reads_fused(b, st, one_call && inline_env && b->n_seqs == 1));
        TRY(ev_end(ix, &fe, ix->ev_and, st));
        b->run_h = ix->h;
        b->fused_run = true;
        b->count_bytes = 2;
        b->sparse_counts = !b->exact;
        b->compacted = true;
        b->done_stale = one_call;
        if (!one_call) {
            if (!b->done) HIP_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(b->done, st));
        }
        b->run_stream = st;
        b->ran = true;
        b->dirty = false;
        return BIGSI_OK;
    }

    // read kernels of other batches still in flight on the read streams: over before this run's kernels start (its hit
    // compaction waits between workgroups, as they do; each kind has the device to its own launches)
    TRY(quiesce_reads(ix));
    b->run_stream = ix->stream;
    // K1e: address-ordered copy of the row lists for K2 (BIGSI_HIP_SORT_ROWS=0 streams them in hash order instead)
    static const int sort_rows = env_int("BIGSI_HIP_SORT_ROWS", 1);
    static const int sort_min_rows = env_int("BIGSI_HIP_SORT_MIN_ROWS", 1024);
    // (not for the few queries of a latency-bound call either: their row lists are cut into slices over many workgroups -- see
    // `slices` below -- and the ordering buys nothing, it only lengthens the chain of kernels: 10 us of a 65 us single query)
    // caller-owned result buffers (a shard's slot of a gather buffer): a bitmap can be preset and sliced like the batch's own
    // (the counting path then cuts its hit mask from the slices' summed partial counts, k_count_combine); caller-owned counters are
    // written in place, without presets
    const bool sliceable = !b->ext_counts;
    const bool few = (uint64_t)b->n_seqs * ceil_div(b->wv, 64 * kVec) < 1024 && sliceable;
    const bool want_sorted = sort_rows && b->exact && !few && !(flags & BIGSI_RUN_NO_SORT) && b->total_pos && b->max_pos * ix->h >= (uint64_t)sort_min_rows;
    // small batches: cut every query's row list into slices so that ~2k wavefronts are in flight (see map_block)
    static const int slices_env = env_int("BIGSI_HIP_SLICES", 0);
    uint32_t slices = 1;
    {
        const uint64_t waves = (uint64_t)b->n_seqs * ceil_div(b->wv, 64 * kVec);
        if (slices_env > 0) slices = (uint32_t)slices_env;
        // (one 1 kbp query on 100 k samples, its slices spread over all XCDs (map_block): exact 35 / 14.7 / 16.6 / 22.9 us at
        // 16 / 64 / 128 / 256 slices, counting 59 / 31 / 30 / 32 us; beyond that the atomics that combine the slices show)
        // (counting, round 4: a slice of ~10 k-mers leaves 4 bit-sliced planes instead of 5 for k_count_combine to add up -- one
        // 1 kbp query on 100 k samples at 0.4, the whole call: 60 slices 50.3 us, 96: 47.0, 128: 49.0; exact: 60 -> 36.3, 96 -> 35.8, 128 -> 41)
        else if (waves < 1024 && b->exact) slices = (uint32_t)std::min<uint64_t>({64, ceil_div(2048, std::max<uint64_t>(waves, 1)), std::max<uint64_t>(b->max_pos / 16, 1)});
        else if (waves < 1024) slices = (uint32_t)std::min<uint64_t>({96, ceil_div(2048, std::max<uint64_t>(waves, 1)), std::max<uint64_t>(b->max_pos / 10, 1)});
        if (!sliceable) slices = 1;
        slices = std::max<uint32_t>(slices, 1);
    }
    // planes needed for the largest possible count = max k-mers of any sequence in the batch
    const uint64_t maxu = b->max_pos;
    const int P = maxu < (1ull << 6) ? 6 : maxu < (1ull << 10) ? 10 : maxu < (1ull << 12) ? 12 : maxu < (1ull << 16) ? 16 : 32;
    // the sliced launches combine into preset result words (all ones for the AND, zero counters): K1 sets them on its way
    Preset preset;
    if (slices > 1 && b->exact) {
        uint64_t *out = (uint64_t *)b->ext_bitmaps;
        if (!out) { TRY(b->bitmaps.reserve((size_t)b->n_seqs * b->wv_pad * 8)); out = b->bitmaps.as<uint64_t>(); }
        preset.p = out; preset.words = b->wv_pad; preset.value = ~0ull;
    }
    // K1 (its LDS route emits the sorted list itself; the other routes leave that to k_sort_rows below)
    EventPair ep{};
    bool sorted_by_k1 = false;
    if (!was_idle && b->job.done && b->job.device_work) HIP_TRY(hipStreamWaitEvent(k1_stream(ix), b->job.done, 0));      // (as on the read path above)
    if (!was_idle && b->done_stale && b->run_stream && b->run_stream != ix->stream) HIP_TRY(hipStreamSynchronize(b->run_stream));      // (a read run of this workspace)
    {
        const K1Route route = k1_plan(b, (flags & BIGSI_RUN_K1_GLOBAL) != 0).route;
        TRY(flush_upload(b, k1_stream(ix), route == K1_WAVE || route == K1_LDS));
    }
    TRY(run_kmerize(b, threshold, (flags & BIGSI_RUN_K1_GLOBAL) != 0, want_sorted, &sorted_by_k1, &preset));
    b->dirty = true;        // until `done` is recorded at the end
    const uint64_t *k2_rows = sorted_by_k1 ? b->rows_sorted.as<uint64_t>() : b->rows.as<uint64_t>();
    // exact path only: there every row can move freely (+4.7 % C3, +7.6 % C4-shard, interleaved A/B); on the counting path a
    // k-mer's h rows must stay together and ordering k-mers by their first row measured 1.00x
    // and only for long row lists (>= 1024 rows per query): for read-length queries (C2: 93 rows) the extra launch costs more
    // than the ordering gains (0.100 vs 0.083 ms per step measured)
    if (want_sorted && !sorted_by_k1) {
        TRY(b->rows_sorted.reserve(std::max<uint64_t>(b->total_pos, 1) * ix->h * 8));
        uint32_t shift = 0;
        while (((ix->m - 1) >> shift) >= (uint64_t)kSortBuckets) shift++;
        TRY(ev_begin(ix, &ep, k1_stream(ix)));
        // 1024 threads per query for long row lists (a 1 kbp query at h=4 has 3880 rows), 256 otherwise
        if (b->max_pos * ix->h >= 2048)
            hipLaunchKernelGGL((k_sort_rows<1024>), dim3(b->n_seqs), dim3(1024), 0, k1_stream(ix), b->rows.as<uint64_t>(), b->rows_sorted.as<uint64_t>(),
                               b->d_pos_off.as<uint64_t>(), b->num_unique.as<uint32_t>(), ix->h, 1u, shift);
        else
            hipLaunchKernelGGL((k_sort_rows<kBlock>), dim3(b->n_seqs), dim3(kBlock), 0, k1_stream(ix), b->rows.as<uint64_t>(), b->rows_sorted.as<uint64_t>(),
                               b->d_pos_off.as<uint64_t>(), b->num_unique.as<uint32_t>(), ix->h, 1u, shift);
        HIP_TRY(hipGetLastError());
        TRY(ev_end(ix, &ep, ix->ev_km, k1_stream(ix)));
        k2_rows = b->rows_sorted.as<uint64_t>();
    }
    TRY(k1_publish(b));
    // K2
    static const int and_block_env = [] { int v = env_int("BIGSI_HIP_AND_BLOCK", 256); return (v >= 64 && v <= 1024 && v % 64 == 0) ? v : 256; }();
    // the counting kernels are compiled for at most 256 threads per workgroup (register budget of the plane arrays)
    // one wavefront per workgroup for batches of a few thousand wavefronts (80 ... 300 gene-length queries on 100 k samples):
    // they are a single launch, a CU's share of it is what bounds it, and 4-wavefront workgroups leave the CUs unevenly loaded
    // (320 / 576 / 800 workgroups on 256 CUs: 0.69 / 0.68 / 0.71 of peak against 0.81 / 0.76 / 0.76 with 64 threads); the large
    // launches, sized in whole workgroups per CU, keep 256 threads (0.85 against 0.79)
    // (exact batches large enough for the launch rule below to cut them -- from 256 such queries on -- are not "mid")
    const uint64_t all_waves = (uint64_t)b->n_seqs * ceil_div(b->wv, 64 * kVec);
    bool mid = and_block_env == 256 && all_waves >= 1024 && all_waves < 4096;
    if (mid) {
        const uint64_t t256 = ceil_div(b->wv, 256 * kVec), wq = ceil_div(b->wv, 64 * kVec);
        const uint64_t blocks256 = ceil_div(b->n_seqs, 8) * 8 * t256, kb256 = round_up(ceil_div(1600 * t256, wq), 256);
        // ... nor are batches that already are a whole number of 4-wavefront workgroups per CU (256 queries on a 62.5 k-sample
        // shard: 512 workgroups, 0.80-0.82 either way)
        if (blocks256 % 256 == 0 || (b->exact && blocks256 >= 2 * kb256)) mid = false;
    }
    // (a sliced exact launch -- a latency-bound call -- in workgroups of two wavefronts: the pieces spread more evenly over the CUs and the
    // stragglers end sooner; one 1 kbp query on 100 k samples, the call: 256 -> 41.9, 128 -> 41.2, 64 -> 41.4 us; counting: no difference)
    static const int and_block_set = env_int("BIGSI_HIP_AND_BLOCK", 0);
    const int and_block = mid ? 64 : (slices > 1 && b->exact && !and_block_set) ? 128 : b->exact ? and_block_env : std::min(and_block_env, 256);
    // row loads a lane keeps in flight: 8, or 4 when 8 would put more bytes in flight on the chip (queries of the launch x row bytes x
    // loads) than the memory system schedules well -- the optimum measured at 8-13 MB.  Interleaved A/B: 256 queries per launch on
    // 62.5 k-sample shards (7.8 KB rows: 16 MB at 8 loads): 4 -> +3.3 % (C4 shard 263 -> 272 M lookups/s) / +2.2 % (north-star shard),
    // 6 -> +1.5 %, 2 -> -17 %; unchunked C3 launches of 160-248 queries (16-25 MB): 4 -> +2 ... +9 %.  At 12.8 MB 8 stays: C3's 128-query
    // launches (4: -5 %) and C3 split over 2 / 4 / 8 GPUs -- 256 x 6.3 KB, 512 x 3.1 KB, 1024 x 1.6 KB rows per launch (4: -7 / -10 /
    // -7 %).  BIGSI_HIP_AND_UNROLL (tuning builds) forces one.
    static const int and_unroll_env = env_int("BIGSI_HIP_AND_UNROLL", 0);
    const uint32_t tiles = (uint32_t)ceil_div(b->wv, (uint64_t)and_block * kVec);
    // large exact batches go out as several launches, each a whole number of workgroups per CU (launches of 384 or 640
    // workgroups measured 0.72-0.78 of peak, 512 / 768 / 1024: 0.82-0.85) with about 1600-2000 LIVE wavefronts: all co-resident,
    // sweeping the address-ordered row lists together, and no more bytes in flight than the memory system schedules well --
    // 10 M x 100 k (13 live wavefronts per query in 4 workgroups): 512 workgroups per launch 0.853 of peak, 1024: 0.819, 256:
    // 0.68; a 12.5 k-sample shard (2 live wavefronts per workgroup): 1024 workgroups 0.773, 512: 0.581.  Queries per launch
    // a multiple of 8 (the blockIdx -> XCD map).  The counting kernel measured -4 ... 0 % chunked and stays one launch.
    static const int k2_blocks = env_int("BIGSI_HIP_K2_BLOCKS", 0);          // > 0: workgroups per launch, fixed
    static const int k2_waves = env_int("BIGSI_HIP_K2_WAVES", 1600);         // live wavefronts a launch should reach at least
    const uint64_t blocks_per_q = (uint64_t)tiles * slices;
    uint32_t chunk_q = b->n_seqs;
    {
        const uint64_t total_blocks = ceil_div(b->n_seqs, 8) * 8 * blocks_per_q;
        if (total_blocks > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "batch too large for one launch (%llu workgroups)", (unsigned long long)total_blocks);
        uint64_t kb = k2_blocks > 0 ? (uint64_t)k2_blocks : 0;
        if (!kb && slices == 1) {
            const uint64_t waves_per_q = ceil_div(b->wv, 64 * kVec);      // wavefronts of a query that hold columns
            kb = round_up(ceil_div((uint64_t)k2_waves * blocks_per_q, waves_per_q), 256);
        }
        if (kb > 0 && b->exact && total_blocks >= 2 * kb)
            chunk_q = (uint32_t)std::max<uint64_t>(8, (kb / blocks_per_q) / 8 * 8);
    }
    uint32_t n_launches = 0;
    if (b->exact) {
        uint64_t *out = (uint64_t *)b->ext_bitmaps;
        if (!out) { TRY(b->bitmaps.reserve((size_t)b->n_seqs * b->wv_pad * 8)); out = b->bitmaps.as<uint64_t>(); }
        if (slices > 1 && !preset.done) HIP_TRY(hipMemsetAsync(out, 0xFF, (size_t)b->n_seqs * b->wv_pad * 8, ix->stream));
        TRY(ev_begin(ix, &ep, nullptr, true));
        static const int and_nt = env_int("BIGSI_HIP_AND_NT", 1);     // 0: plain loads (A/B against non-temporal)
        for (uint32_t q0 = 0; q0 < b->n_seqs; q0 += chunk_q, n_launches++) {
            const uint32_t q1 = std::min<uint64_t>((uint64_t)q0 + chunk_q, b->n_seqs);
            unsigned grid = (unsigned)((slices > 1 ? (uint64_t)(q1 - q0) : ceil_div(q1 - q0, 8) * 8) * blocks_per_q), l_block = (unsigned)and_block;
            uint32_t l_tiles = tiles, l_slices = slices;
            if (chunk_q < b->n_seqs && q1 - q0 < chunk_q && and_block == 256) {
                // the last launch of a batch that is not a multiple of the launch size is a batch of its own kind: with a few
                // thousand wavefronts one-wavefront workgroups (see `mid` above), with fewer the sliced launch of a small batch
                const uint64_t waves_r = (uint64_t)(q1 - q0) * ceil_div(b->wv, 64 * kVec);
                if (waves_r < 1024) {
                    l_slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({64, ceil_div(2048, std::max<uint64_t>(waves_r, 1)), std::max<uint64_t>(b->max_pos / 16, 1)}));
                    if (l_slices > 1) HIP_TRY(hipMemsetAsync(out + (uint64_t)q0 * b->wv_pad, 0xFF, (size_t)(q1 - q0) * b->wv_pad * 8, ix->stream));
                    grid = (unsigned)((l_slices > 1 ? (uint64_t)(q1 - q0) : ceil_div(q1 - q0, 8) * 8) * (uint64_t)tiles * l_slices);
                } else if (grid % 256 != 0) {
                    l_block = 64;
                    l_tiles = (uint32_t)ceil_div(b->wv, 64 * kVec);
                    grid = (unsigned)(ceil_div(q1 - q0, 8) * 8 * (uint64_t)l_tiles);
                }
            }
#define COMMA ,
#define BIGSI_LAUNCH_EXACT(U)                                                                                                  \
    hipLaunchKernelGGL((k_and_exact<U>), dim3(grid), dim3(l_block), 0, ix->stream, ix->d_index, ix->stride_words, (uint32_t)b->wv, \
                       ix->n_cols, k2_rows, b->d_pos_off.as<uint64_t>(), b->num_unique.as<uint32_t>(), ix->h, q0,    \
                       q1, l_tiles, out, b->wv_pad, l_slices, (flags & BIGSI_RUN_EARLY_EXIT) ? 1u : 0u)
            const uint64_t in_flight_at_8 = (uint64_t)(q1 - q0) * b->wv * 8 * 8;
            const int and_unroll = and_unroll_env ? and_unroll_env : (in_flight_at_8 > (29ull << 19) /* 14.5 MB */ && l_slices == 1 ? 4 : 8);
            if (and_unroll == 4) BIGSI_LAUNCH_EXACT(4);
#ifdef BIGSI_HIP_TUNING
            else if (and_unroll == 2) BIGSI_LAUNCH_EXACT(2);
            else if (and_unroll == 6) BIGSI_LAUNCH_EXACT(6);
            else if (and_unroll == 16) BIGSI_LAUNCH_EXACT(16);
#endif
            else if (!and_nt) BIGSI_LAUNCH_EXACT(8 COMMA false);
            else BIGSI_LAUNCH_EXACT(8);
#undef BIGSI_LAUNCH_EXACT
#undef COMMA
        }
        HIP_TRY(hipGetLastError());
        TRY(ev_end(ix, &ep, ix->ev_and, nullptr, n_launches));
    } else {
        b->count_bytes = P <= 16 ? 2 : 4;
        const uint64_t cstride = b->wv_pad * 64;
        void *out = b->ext_counts;
        if (!out) { TRY(b->counts.reserve((size_t)b->n_seqs * cstride * b->count_bytes)); out = b->counts.p; }
        // the kernel also leaves the thresholded hit bitmap (count >= min_kmers), which is what K4 compacts on a single GPU
        TRY(b->bitmaps.reserve((size_t)b->n_seqs * b->wv_pad * 8));
        uint64_t *hb = b->ext_bitmaps ? (uint64_t *)b->ext_bitmaps : b->bitmaps.as<uint64_t>();
        b->sparse_counts = (flags & BIGSI_RUN_SPARSE_COUNTS) && !b->ext_counts;
        const uint32_t sparse = b->sparse_counts ? 1u : 0u;
        // a sliced (small) batch: every slice leaves its partial counts bit-sliced in scratch memory -- as many planes as a slice's
        // k-mers need -- and k_count_combine adds them up, thresholds and expands (no presets, no atomics)
        uint32_t planes_out = 0;
        uint64_t *partial = nullptr;
        if (slices > 1) {
            const uint64_t per_slice = ceil_div(std::max<uint64_t>(b->max_pos, 1), slices);
            while (planes_out < (uint32_t)P && (per_slice >> planes_out) != 0) planes_out++;
            TRY(b->planes.reserve((size_t)b->n_seqs * slices * planes_out * b->wv_pad * 8));
            partial = b->planes.as<uint64_t>();
        }
        TRY(ev_begin(ix, &ep, nullptr, true));
        // fewer than ~3 wavefronts per SIMD in the whole grid (e.g. 128 gene-length queries): the software-pipelined loop,
        // whose wavefronts load the next k-mers' rows while adding the current ones (5.6 -> 6.3 TB/s at 128 x 2-4 kbp; with a
        // full grid other wavefronts already cover the ALU phase and it measured -2 ... +0 %)
        static const int deep_env = env_int("BIGSI_HIP_COUNT_DEEP", -1);
        const uint64_t grid_waves = (uint64_t)b->n_seqs * tiles * (and_block / 64);
        // (only with >= 12 planes, i.e. queries of >= 1024 k-mers: at 10 planes the ALU phase is short and it measured -2 %)
        const bool deep = deep_env >= 0 ? deep_env != 0 : (slices == 1 && P >= 12 && grid_waves < 3 * 1024);
        const uint32_t early = ((flags & BIGSI_RUN_EARLY_EXIT) && sparse && slices == 1) ? 1u : 0u;
        // one word per lane (8-byte loads, half the plane registers: 8 instead of 4-5 wavefronts per SIMD) -- tuning builds only:
        // interleaved A/B +-0 at C3 (h = 4), -3.5 % on the north-star shard, -2 % on the C5 shard (h = 3): these kernels are at the
        // memory system's random-row rate, not short of wavefronts (unlike the read kernel, where the same change is +10 %)
        static const int count_vec1 = env_int("BIGSI_HIP_COUNT_VEC1", 0);
        const int vec = (count_vec1 && slices == 1 && !(deep && !early) && (ix->h == 3 || ix->h == 4) && P <= 16) ? 1 : 2;
        const uint32_t ctiles = vec == 1 ? (uint32_t)ceil_div(b->wv, (uint64_t)and_block) : tiles;
        if (ceil_div(b->n_seqs, 8) * 8 * (uint64_t)ctiles * slices > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "batch too large for one launch");
        // half the row loads in flight per lane (tuning builds; what gives the EXACT kernel +2-3 % on 62.5 k-sample shards): -5 % on the
        // north-star shard at 0.4, +-0 on the C5 shard and at C3 -- the counting kernel keeps 8-12
        static const int count_half = env_int("BIGSI_HIP_COUNT_HALF", 0);
        const bool half = count_half == 1 && slices == 1 && !(deep && !early) && (ix->h == 3 || ix->h == 4);
        const CountLaunch cl{k2_rows, (unsigned)and_block, ctiles, out, cstride, hb, sparse, slices, deep && !early, early, partial, planes_out, half, vec};
        for (uint32_t q0 = 0; q0 < b->n_seqs; q0 += chunk_q, n_launches++)
            launch_count(b, P, cl, q0, (uint32_t)std::min<uint64_t>((uint64_t)q0 + chunk_q, b->n_seqs));
        if (slices > 1) {        // the slices' partial counts -> totals, hit mask, counters
            const unsigned grid = (unsigned)(b->n_seqs * ceil_div(b->wv, kBlock / 8));      // 32 words per workgroup, 8 slice groups per word
#define BIGSI_COMBINE(PP, T)                                                                                                          \
    hipLaunchKernelGGL((k_count_combine<PP, T>), dim3(grid), dim3(kBlock), 0, ix->stream, partial, slices, planes_out, b->wv_pad, (uint32_t)b->wv, \
                       b->n_seqs, b->num_unique.as<uint32_t>(), b->min_kmers.as<uint32_t>(), ix->n_cols, hb, (T *)out, cstride, sparse)
            switch (P) {
            case 6: BIGSI_COMBINE(6, uint16_t); break;
            case 10: BIGSI_COMBINE(10, uint16_t); break;
            case 12: BIGSI_COMBINE(12, uint16_t); break;
            case 16: BIGSI_COMBINE(16, uint16_t); break;
            default: BIGSI_COMBINE(32, uint32_t); break;
            }
#undef BIGSI_COMBINE
        }
        HIP_TRY(hipGetLastError());
        TRY(ev_end(ix, &ep, ix->ev_and, nullptr, n_launches));
    }

    b->compacted = !(flags & BIGSI_RUN_SKIP_COMPACT);
    if (!b->done) HIP_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    b->done_stale = false;
    if (!b->compacted) {
        HIP_TRY(hipEventRecord(b->done, ix->stream));
        TRY(mark_main(ix));
        b->ran = true;
        b->dirty = false;
        return BIGSI_OK;
    }
    // K4 on this shard's own result
    TRY(ev_begin(ix, &ep));
    const void *src = b->exact ? (b->ext_bitmaps ? b->ext_bitmaps : b->bitmaps.p) : (b->ext_counts ? b->ext_counts : b->counts.p);
    TRY(compact(b, b->hits, src, 1, ix->n_cols, false));
    TRY(ev_end(ix, &ep, ix->ev_cp));
    b->done_stale = one_call;
    if (!one_call) HIP_TRY(hipEventRecord(b->done, ix->stream));
    TRY(mark_main(ix));
    b->ran = true;
    b->dirty = false;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_run(bigsi_hip_batch *b, double threshold, uint32_t flags)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    return bigsi_batch_run(b, threshold, flags, false);
}

// three compaction passes over [shard][seq][stride]; write_only re-runs just the write pass (after growing buffers).
// `from_counts`: src is a counter buffer gathered from several shards -> threshold while compacting (k_hits_count);
// otherwise src is a hit bitmap (the exact AND, or the counting kernel's fused count >= min_kmers mask) and the per-hit
// count comes from `counters` (null on the exact path: every hit has count == num_unique).
static int compact_ex(bigsi_hip_batch *b, HitBufs &hb, const void *src, bool from_counts, const void *counters,
                      uint32_t n_shards, uint64_t shard_cols, bool write_only, hipStream_t st, uint32_t own_shard = kAllShards)
{
    const uint32_t chunks = !from_counts ? (uint32_t)ceil_div(b->wv, kBlock) : (uint32_t)ceil_div(b->wv_pad * 64, kChunkCols);
    const uint64_t per_seq = (uint64_t)n_shards * chunks, nchunks = per_seq * b->n_seqs;
    if (nchunks > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "too many compaction chunks");
    TRY(hb.hit_off.reserve((b->n_seqs + 1) * 8ull));
    if (hb.cap == 0 && !hb.xcol) {
        const uint64_t want = 1u << 16;
        TRY(hb.hit_col.reserve(want * 4));
        TRY(hb.hit_cnt.reserve(want * 4));
        hb.cap = want;
    }
    if (!from_counts) {
        // bit vectors (the AND bitmap / the fused count >= min_kmers mask): group totals, then prefix + ordered write
        // (k_hits_totals, k_hits_write: no waiting between workgroups).  `write_only` (the lists overflowed and were grown)
        // runs the write pass again: the totals are still there.
        static const int k4_groups = env_int("BIGSI_HIP_K4_GROUPS", (int)kHitsMaxGroups);
        const uint64_t max_groups = (uint64_t)std::min<int>(std::max(k4_groups, 1), (int)kHitsMaxGroups);
        // a handful of items (one or two gene-length queries: a latency-bound call) go to ONE workgroup, which needs nobody's totals
        const uint32_t ipb = nchunks <= 16 ? (uint32_t)nchunks : (uint32_t)ceil_div(nchunks, max_groups);
        const uint64_t ngroups = ceil_div(nchunks, ipb);
        TRY(hb.chunk_hits.reserve(kHitsMaxGroups * 4));
        if (!write_only && ngroups > 1)
            hipLaunchKernelGGL(k_hits_totals, dim3((unsigned)ngroups), dim3(kBlock), 0, st, (const uint64_t *)src, b->wv_pad, (uint32_t)b->wv, b->n_seqs,
                               n_shards, chunks, ipb, hb.chunk_hits.as<uint32_t>());
        // a one-call search whose compaction is ONE workgroup (a gene-length query or two): that workgroup exports the results itself
        static const int k4_export = env_int("BIGSI_HIP_K4_EXPORT", 1);
        const bool inline_export = k4_export && b->one_call && &hb == &b->hits && !write_only && ngroups == 1 && n_shards == 1 && chunks <= 16 && st == b->ix->stream;
        if (inline_export) {
            TRY(export_prepare(b, st));
            if (b->exp_flagged) b->exported_inline = true;
            else b->exp_serial--;          // (tuning builds with the event route: the export kernel as usual)
        }
        hipLaunchKernelGGL(k_hits_write, dim3((unsigned)ngroups), dim3(kBlock), 0, st, (const uint64_t *)src, b->wv_pad, (uint32_t)b->wv, b->n_seqs,
                           n_shards, chunks, shard_cols, b->num_unique.as<uint32_t>(), ipb, hb.chunk_hits.as<uint32_t>(),
                           hb.hit_off.as<uint64_t>(), hb.col(), hb.cnt(), hb.capacity(), counters, b->count_bytes, b->wv_pad * 64, own_shard,
                           b->exported_inline && inline_export ? static_cast<uint64_t *>(b->pin_out) : nullptr, b->exp_spec, b->uniq.as<uint32_t>(),
                           (volatile uint64_t *)b->pin_flag, b->exp_serial);
        HIP_TRY(hipGetLastError());
        return BIGSI_OK;
    }
    // gathered dense counters / row-sliced local counters: threshold while compacting, three passes
    TRY(hb.chunk_hits.reserve(nchunks * 4));
    TRY(hb.chunk_off.reserve(nchunks * 8));
    TRY(hb.overflow.reserve(4));
    const unsigned grid = (unsigned)nchunks;
    HIP_TRY(hipMemsetAsync(hb.overflow.p, 0, 4, st));
#define BIGSI_HITS_COMMON                                                                                                        \
    b->n_seqs, n_shards, chunks, shard_cols, b->min_kmers.as<uint32_t>(), hb.chunk_hits.as<uint32_t>(), hb.chunk_off.as<uint64_t>(), \
        hb.col(), hb.cnt(), hb.capacity(), hb.overflow.as<uint32_t>()
    for (int pass = write_only ? 1 : 0; pass < 2; pass++) {
        if (b->count_bytes == 2) {
            const uint16_t *c16 = (const uint16_t *)src;
            if (pass == 0) hipLaunchKernelGGL((k_hits_count<uint16_t, false>), dim3(grid), dim3(kBlock), 0, st, c16, b->wv_pad * 64, (uint32_t)b->wv, BIGSI_HITS_COMMON);
            else hipLaunchKernelGGL((k_hits_count<uint16_t, true>), dim3(grid), dim3(kBlock), 0, st, c16, b->wv_pad * 64, (uint32_t)b->wv, BIGSI_HITS_COMMON);
        } else {
            const uint32_t *c32 = (const uint32_t *)src;
            if (pass == 0) hipLaunchKernelGGL((k_hits_count<uint32_t, false>), dim3(grid), dim3(kBlock), 0, st, c32, b->wv_pad * 64, (uint32_t)b->wv, BIGSI_HITS_COMMON);
            else hipLaunchKernelGGL((k_hits_count<uint32_t, true>), dim3(grid), dim3(kBlock), 0, st, c32, b->wv_pad * 64, (uint32_t)b->wv, BIGSI_HITS_COMMON);
        }
        HIP_TRY(hipGetLastError());
        if (pass == 0) {
            hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(kBlock), 0, st, hb.chunk_hits.as<uint32_t>(), nchunks, (uint32_t)per_seq,
                               b->n_seqs, hb.chunk_off.as<uint64_t>(), hb.hit_off.as<uint64_t>());
            HIP_TRY(hipGetLastError());
        }
    }
#undef BIGSI_HITS_COMMON
    return BIGSI_OK;
}

// this shard's own result (n_shards == 1): always a bitmap; gathered buffers: bitmaps (exact) or counters (counting)
static int compact(bigsi_hip_batch *b, HitBufs &hb, const void *src, uint32_t n_shards, uint64_t shard_cols, bool write_only)
{
    if (&hb == &b->hits) {
        const void *bm = b->ext_bitmaps ? b->ext_bitmaps : b->bitmaps.p;
        const void *counters = b->exact ? nullptr : (b->ext_counts ? b->ext_counts : b->counts.p);
        return compact_ex(b, hb, bm, false, counters, 1, shard_cols, write_only, b->ix->stream);
    }
    hipStream_t gst = b->gstream ? b->gstream : b->ix->stream;
    if (!b->exact && b->g_masks)
        return compact_ex(b, hb, src, false, b->ext_counts ? b->ext_counts : b->counts.p, n_shards, shard_cols, write_only, gst, b->g_own);
    return compact_ex(b, hb, src, !b->exact, nullptr, n_shards, shard_cols, write_only, gst);
}

// A one-launch read run leaves every query's hits where its workgroup allocated them (k_reads_fused: no order between queries).
// fetch_hits brings them to the host in QUERY order: the totals decide whether the lists fit (grow + launch again if not), the
// per-query (start, count) pairs give the offsets, one download of the used part of the buffers and one pass over the queries
// put every list at its place.  The batch's `done` event has been waited for.
static int fetch_read_hits(bigsi_hip_batch *b, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity)
{
    HitBufs &hb = b->hits;
    hipStream_t st = b->run_stream ? b->run_stream : b->ix->stream;
    const uint32_t n = b->n_seqs;
    unsigned long long total = 0;
    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipMemcpy(&total, hb.alloc.as<unsigned long long>() + (hb.gen & 1u), 8, hipMemcpyDeviceToHost));
        if (total <= hb.capacity()) break;
        if (hb.xcol) {
            if (hit_offsets) memset(hit_offsets, 0, (n + 1) * 8ull), hit_offsets[n] = total;
            return fail(BIGSI_ERR_CAPACITY, "caller-owned hit buffers hold %llu entries, %llu needed", (unsigned long long)hb.xcap, total);
        }
        if (attempt) return fail(BIGSI_ERR_HIP, "internal: a read run overflowed hit buffers sized for its own total");
        TRY(hb.hit_col.reserve(total * 4));          // counters lived in registers: the whole pass again, on the stream it ran on
        TRY(hb.hit_cnt.reserve(total * 4));
        hb.cap = total;
        TRY(launch_reads_fused(b, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    std::vector<uint32_t> qc(n);
    std::vector<uint64_t> off(n + 1), qs;
    HIP_TRY(hipMemcpy(qc.data(), hb.q_cnt.p, n * 4ull, hipMemcpyDeviceToHost));
    off[0] = 0;
    for (uint32_t q = 0; q < n; q++) off[q + 1] = off[q] + qc[q];
    if (off[n] != total) return fail(BIGSI_ERR_HIP, "internal: hit counts of a read run do not add up (%llu vs %llu)", (unsigned long long)off[n], total);
    if (hit_offsets) memcpy(hit_offsets, off.data(), (n + 1) * 8ull);
    if (total > capacity)
        return fail(BIGSI_ERR_CAPACITY, "hit buffers hold %llu entries, %llu needed", (unsigned long long)capacity, total);
    if (!total || (!colours && !counts)) return BIGSI_OK;
    qs.resize(n);
    HIP_TRY(hipMemcpy(qs.data(), hb.q_start.p, n * 8ull, hipMemcpyDeviceToHost));
    std::vector<uint32_t> ucol(colours ? total : 0), ucnt(counts ? total : 0);
    if (colours) HIP_TRY(hipMemcpy(ucol.data(), hb.col(), total * 4, hipMemcpyDeviceToHost));
    if (counts) HIP_TRY(hipMemcpy(ucnt.data(), hb.cnt(), total * 4, hipMemcpyDeviceToHost));
    for (uint32_t q = 0; q < n; q++) {
        if (!qc[q]) continue;
        if (colours) memcpy(colours + off[q], ucol.data() + qs[q], qc[q] * 4ull);
        if (counts) memcpy(counts + off[q], ucnt.data() + qs[q], qc[q] * 4ull);
    }
    return BIGSI_OK;
}

// synchronise, make sure the hit lists fit (grow + rewrite if the write pass overflowed), copy them out
static int fetch_hits_from(bigsi_hip_batch *b, HitBufs &hb, const void *src, uint32_t n_shards, uint64_t shard_cols,
                           uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity)
{
    if (&hb == &b->hits && b->fused_run) return fetch_read_hits(b, hit_offsets, colours, counts, capacity);
    hipStream_t st = (&hb == &b->ghits && b->gstream) ? b->gstream : b->ix->stream;
    std::vector<uint64_t> off(b->n_seqs + 2);
    // local hit lists were produced before b->done (already waited for); gathered ones on the gather stream
    if (&hb == &b->ghits || !b->compacted) HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(off.data(), hb.hit_off.p, (b->n_seqs + 1) * 8ull, hipMemcpyDeviceToHost));
    const uint64_t total = off[b->n_seqs];
    if (hb.xcol && total > hb.xcap) {
        if (hit_offsets) memcpy(hit_offsets, off.data(), (b->n_seqs + 1) * 8ull);
        return fail(BIGSI_ERR_CAPACITY, "caller-owned hit buffers hold %llu entries, %llu needed", (unsigned long long)hb.xcap, (unsigned long long)total);
    }
    if (!hb.xcol && total > hb.cap) {
        TRY(hb.hit_col.reserve(total * 4));
        TRY(hb.hit_cnt.reserve(total * 4));
        hb.cap = total;
        TRY(compact(b, hb, src, n_shards, shard_cols, true));
        if (&hb == &b->ghits && b->comm && !b->exact) TRY(bigsi_reduce_gathered_counts(b));   // every rank takes this branch: totals are identical
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (hit_offsets) memcpy(hit_offsets, off.data(), (b->n_seqs + 1) * 8ull);
    if (total > capacity)
        return fail(BIGSI_ERR_CAPACITY, "hit buffers hold %llu entries, %llu needed", (unsigned long long)capacity, (unsigned long long)total);
    if (total && colours) HIP_TRY(hipMemcpy(colours, hb.col(), total * 4, hipMemcpyDeviceToHost));
    if (total && counts) HIP_TRY(hipMemcpy(counts, hb.cnt(), total * 4, hipMemcpyDeviceToHost));
    return BIGSI_OK;
}

static int need_run(bigsi_hip_batch *b)
{
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!b->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_batch_run has not completed for this batch");
    TRY(use_device(b->ix));
    if (b->idle) return BIGSI_OK;
    if (b->done_stale && b->run_stream) HIP_TRY(hipStreamSynchronize(b->run_stream));
    else if (b->done) HIP_TRY(hipEventSynchronize(b->done));      // this batch's kernels; later batches may still be running
    return BIGSI_OK;
}

static int host_counts(bigsi_hip_batch *b)
{
    if (b->host_counts_valid) return BIGSI_OK;
    const size_t n = b->n_seqs;
    b->h_uniq.resize(3 * n);
    HIP_TRY(hipMemcpy(b->h_uniq.data(), b->uniq.p, 3 * n * 4, hipMemcpyDeviceToHost));       // (point_uniq)
    b->h_num_kmers.assign(b->h_uniq.begin(), b->h_uniq.begin() + n);
    b->h_num_unique.assign(b->h_uniq.begin() + n, b->h_uniq.begin() + 2 * n);
    b->h_min_kmers.assign(b->h_uniq.begin() + 2 * n, b->h_uniq.end());
    b->host_counts_valid = true;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_get_info(bigsi_hip_batch *b, bigsi_hip_batch_info *out)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    memset(out, 0, sizeof *out);
    out->n_seqs = b->n_seqs;
    out->k = b->k;
    out->total_kmers = b->total_pos;
    if (!b->ran) return BIGSI_OK;
    TRY(need_run(b));          // waits for this batch's completion event
    TRY(host_counts(b));
    out->exact = b->exact ? 1 : 0;
    out->count_bytes = b->count_bytes;
    for (uint32_t v : b->h_num_unique) out->total_unique += v;
    uint64_t total = 0;
    if (b->fused_run) HIP_TRY(hipMemcpy(&total, b->hits.alloc.as<unsigned long long>() + (b->hits.gen & 1u), 8, hipMemcpyDeviceToHost));
    else if (b->compacted) HIP_TRY(hipMemcpy(&total, b->hits.hit_off.as<uint64_t>() + b->n_seqs, 8, hipMemcpyDeviceToHost));
    out->total_hits = total;
    out->bitmap_stride_bytes = b->wv_pad * 8;
    out->counts_stride = b->wv_pad * 64;
    out->d_bitmaps = b->ext_bitmaps ? b->ext_bitmaps : b->bitmaps.p;
    out->d_counts = b->ext_counts ? b->ext_counts : b->counts.p;
    out->d_num_unique = b->num_unique.p;
    out->one_launch = b->fused_run ? 1 : 0;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_fetch_unique(bigsi_hip_batch *b, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    TRY(host_counts(b));
    if (num_kmers) memcpy(num_kmers, b->h_num_kmers.data(), b->n_seqs * 4ull);
    if (num_unique) memcpy(num_unique, b->h_num_unique.data(), b->n_seqs * 4ull);
    if (min_kmers) memcpy(min_kmers, b->h_min_kmers.data(), b->n_seqs * 4ull);
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_fetch_hits(bigsi_hip_batch *b, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    const void *src = b->exact ? (b->ext_bitmaps ? b->ext_bitmaps : b->bitmaps.p) : (b->ext_counts ? b->ext_counts : b->counts.p);
    if (!b->compacted) {   // the run skipped K4: do it now
        TRY(compact(b, b->hits, src, 1, b->ix->n_cols, false));
        HIP_TRY(hipStreamSynchronize(b->ix->stream));
        b->compacted = true;
    }
    return fetch_hits_from(b, b->hits, src, 1, b->ix->n_cols, hit_offsets, colours, counts, capacity);
}

// gathered compaction is queued behind the batch's run through its `done` event: no host-side wait, so the caller can go on
// to launch the next batch while this one's row-AND kernel is still running
static int gather_begin(bigsi_hip_batch *b, hipStream_t st)
{
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if (!b->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_batch_run has not completed for this batch");
    TRY(use_device(b->ix));
    if (b->done) HIP_TRY(hipStreamWaitEvent(st, b->done, 0));
    return BIGSI_OK;
}

static int gather_end(bigsi_hip_batch *b, hipStream_t st)
{
    if (!b->g_done) HIP_TRY(hipEventCreateWithFlags(&b->g_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->g_done, st));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_compact_gathered(bigsi_hip_batch *b, const void *d_gathered, uint32_t n_shards, uint64_t shard_cols)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    hipStream_t gst = b->gstream ? b->gstream : b->ix->stream;
    TRY(gather_begin(b, gst));
    if (!d_gathered || n_shards == 0) return fail(BIGSI_ERR_INVALID, "bad gathered buffer");
    if ((uint64_t)n_shards * shard_cols > 0xFFFFFFFFull) return fail(BIGSI_ERR_INVALID, "more than 2^32-1 colours in total");
    b->g_src = d_gathered;
    b->g_shards = n_shards;
    b->g_shard_cols = shard_cols;
    b->g_masks = false;
    EventPair ep{};
    TRY(ev_begin(b->ix, &ep, b->gstream));
    TRY(compact(b, b->ghits, d_gathered, n_shards, shard_cols, false));
    TRY(ev_end(b->ix, &ep, b->ix->ev_cp, b->gstream));
    TRY(gather_end(b, gst));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_compact_gathered_masks(bigsi_hip_batch *b, const void *d_gathered_masks, uint32_t n_shards, uint64_t shard_cols,
                                                      uint32_t own_shard)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    TRY(gather_begin(b, b->gstream ? b->gstream : b->ix->stream));
    if (!d_gathered_masks || n_shards == 0 || own_shard >= n_shards) return fail(BIGSI_ERR_INVALID, "bad gathered buffer / shard");
    if ((uint64_t)n_shards * shard_cols > 0xFFFFFFFFull) return fail(BIGSI_ERR_INVALID, "more than 2^32-1 colours in total");
    if (b->exact) return bigsi_hip_batch_compact_gathered(b, d_gathered_masks, n_shards, shard_cols);
    b->g_src = d_gathered_masks;
    b->g_shards = n_shards;
    b->g_shard_cols = shard_cols;
    b->g_own = own_shard;
    b->g_masks = true;
    hipStream_t st = b->gstream ? b->gstream : b->ix->stream;
    EventPair ep{};
    TRY(ev_begin(b->ix, &ep, st));
    const void *counters = b->ext_counts ? b->ext_counts : b->counts.p;
    TRY(compact_ex(b, b->ghits, d_gathered_masks, false, counters, n_shards, shard_cols, false, st, own_shard));
    TRY(ev_end(b->ix, &ep, b->ix->ev_cp, st));
    TRY(gather_end(b, st));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_set_gathered_hit_outputs(bigsi_hip_batch *b, void *d_colours, void *d_counts, uint64_t capacity)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    if ((d_colours == nullptr) != (d_counts == nullptr)) return fail(BIGSI_ERR_INVALID, "give both buffers or neither");
    b->ghits.xcol = (uint32_t *)d_colours;
    b->ghits.xcnt = (uint32_t *)d_counts;
    b->ghits.xcap = d_colours ? capacity : 0;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_set_gather_stream(bigsi_hip_batch *b, void *hip_stream)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    b->gstream = (hipStream_t)hip_stream;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_fetch_gathered_hits(bigsi_hip_batch *b, uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t capacity)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    if (!b->g_src) return fail(BIGSI_ERR_STATE, "bigsi_hip_batch_compact_gathered has not been called");
    return fetch_hits_from(b, b->ghits, b->g_src, b->g_shards, b->g_shard_cols, hit_offsets, colours, counts, capacity);
}

extern "C" int bigsi_hip_batch_fetch_counts(bigsi_hip_batch *b, uint32_t seq, uint32_t *out)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    if (b->exact) return fail(BIGSI_ERR_STATE, "the last run took the exact path; re-run with BIGSI_RUN_FORCE_COUNTS or threshold < 1");
    if (b->sparse_counts) return fail(BIGSI_ERR_STATE, "the last run used BIGSI_RUN_SPARSE_COUNTS: only counters of hits were stored");
    if (seq >= b->n_seqs) return fail(BIGSI_ERR_RANGE, "sequence %u out of range", seq);
    const uint64_t n = b->ix->n_cols, cstride = b->wv_pad * 64;
    const uint8_t *src = (const uint8_t *)(b->ext_counts ? b->ext_counts : b->counts.p) + (uint64_t)seq * cstride * b->count_bytes;
    if (b->count_bytes == 4) {
        HIP_TRY(hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> tmp(n);
        HIP_TRY(hipMemcpy(tmp.data(), src, n * 2, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) out[i] = tmp[i];
    }
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_fetch_bitmap(bigsi_hip_batch *b, uint32_t seq, uint8_t *out)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    if (!b->exact) return fail(BIGSI_ERR_STATE, "the last run took the counting path");
    if (seq >= b->n_seqs) return fail(BIGSI_ERR_RANGE, "sequence %u out of range", seq);
    const uint8_t *src = (const uint8_t *)(b->ext_bitmaps ? b->ext_bitmaps : b->bitmaps.p) + (uint64_t)seq * b->wv_pad * 8;
    HIP_TRY(hipMemcpy(out, src, b->ix->rb(), hipMemcpyDeviceToHost));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_fetch_rows(bigsi_hip_batch *b, uint32_t seq, uint64_t *rows, uint64_t capacity)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    if (seq >= b->n_seqs) return fail(BIGSI_ERR_RANGE, "sequence %u out of range", seq);
    TRY(host_counts(b));
    const uint64_t n = (uint64_t)b->h_num_unique[seq] * b->ix->h;
    if (n > capacity) return fail(BIGSI_ERR_CAPACITY, "rows buffer holds %llu, %llu needed", (unsigned long long)capacity, (unsigned long long)n);
    if (n) HIP_TRY(hipMemcpy(rows, b->rows.as<uint64_t>() + b->pos_off[seq] * b->ix->h, n * 8, hipMemcpyDeviceToHost));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_lookup(bigsi_hip_batch *b, uint32_t seq, uint32_t *first_pos, uint8_t *out_rows, uint64_t capacity_rows)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    if (seq >= b->n_seqs) return fail(BIGSI_ERR_RANGE, "sequence %u out of range", seq);
    TRY(host_counts(b));
    bigsi_hip_index *ix = b->ix;
    const uint32_t u = b->h_num_unique[seq];
    if (u > capacity_rows) return fail(BIGSI_ERR_CAPACITY, "lookup buffer holds %llu rows, %u needed", (unsigned long long)capacity_rows, u);
    if (u == 0) return BIGSI_OK;
    const uint64_t wv = ix->wv(), rb = ix->rb();
    if (first_pos) HIP_TRY(hipMemcpy(first_pos, b->first_pos.as<uint32_t>() + b->pos_off[seq], u * 4ull, hipMemcpyDeviceToHost));
    if (!out_rows) return BIGSI_OK;
    TRY(b->scratch.reserve((size_t)u * wv * 8));
    const uint64_t wblocks = ceil_div(wv, kBlock);
    if (wblocks * u > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "lookup too large for one launch");
    hipLaunchKernelGGL(k_lookup, dim3((unsigned)(wblocks * u)), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, (uint32_t)wv,
                       b->rows.as<uint64_t>() + b->pos_off[seq] * ix->h, ix->h, u, b->scratch.as<uint64_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(out_rows, rb, b->scratch.p, wv * 8, rb, u, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_presence(bigsi_hip_batch *b, uint32_t seq, const uint32_t *colours, uint32_t n_colours, uint8_t *out)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(need_run(b));
    if (seq >= b->n_seqs) return fail(BIGSI_ERR_RANGE, "sequence %u out of range", seq);
    if (n_colours == 0) return BIGSI_OK;
    if (!colours || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    bigsi_hip_index *ix = b->ix;
    for (uint32_t i = 0; i < n_colours; i++)
        if (colours[i] >= ix->n_cols) return fail(BIGSI_ERR_RANGE, "colour %u >= num_cols", colours[i]);
    TRY(host_counts(b));
    const uint32_t n = b->h_num_kmers[seq];
    if (n == 0) return BIGSI_OK;
    const size_t cbytes = round_up((size_t)n_colours * 4, 256);
    TRY(b->scratch.reserve(cbytes + (size_t)n_colours * n));
    uint8_t *d_out = b->scratch.as<uint8_t>() + cbytes;
    HIP_TRY(hipMemcpyAsync(b->scratch.p, colours, n_colours * 4ull, hipMemcpyHostToDevice, ix->stream));
    if (ceil_div(n, kBlock) * n_colours > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "presence request too large for one launch");
    hipLaunchKernelGGL(k_presence, dim3((unsigned)(ceil_div(n, kBlock) * n_colours)), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words,
                       b->rows.as<uint64_t>() + b->pos_off[seq] * ix->h, b->pos_unique.as<uint32_t>() + b->pos_off[seq], ix->h, n,
                       b->scratch.as<uint32_t>(), n_colours, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n_colours * n, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return BIGSI_OK;
}

// K5 at scale: the presence strings of all hits of the batch (see k_presence_bits).  The host sorts each sequence's hits by
// colour and groups them into 128-column word pairs (a few microseconds per thousand hits); the device does the rest.
// K5 / K6 for the hits of a batch, in two halves so that a serving loop can overlap them with other work:
//   presence_begin  host: string / bit offsets, per-sequence colour order, word pairs -> pinned staging; device (asynchronous):
//                   upload, k_presence_bits, then either the ASCII strings (k_presence_expand*, K5) or the packed presence bits
//                   + score records (k_presence_score, K6), download into pinned staging, completion event.
//   presence_end    waits for that event and copies the staged results into the caller's buffers.
// The kernels go to the library's score stream (highest priority: the host is waiting for them while the next batch's row-AND
// kernel fills the device) or, BIGSI_SCORE_ORDERED, to the index stream behind whatever is already queued there: alone on
// the device they take a tenth of the time they take beside a row-AND kernel, and a loop three batches deep never waits for
// them (bench workload c5).
static_assert(sizeof(bigsi_hip_hit_score) == sizeof(bigsi_score::HitScore) && sizeof(bigsi_hip_hit_score) == 64, "score record layout");

static int pinned_reserve(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return BIGSI_OK;
    if (*p) { hipError_t e = hipHostFree(*p); (void)e; *p = nullptr; *cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    HIP_TRY(hipHostMalloc(p, want, hipHostMallocCoherent | hipHostMallocMapped));      // (kernels write results / read small inputs here directly)
    *cap = want;
    return BIGSI_OK;
}

static int presence_begin(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts, bool packed,
                          bool ordered, uint64_t out_capacity, bool check_capacity, uint64_t *string_offsets)
{
    TRY(need_run(b));
    if (!hit_offsets || !string_offsets) return fail(BIGSI_ERR_INVALID, "NULL argument");
    PresJob &job = b->job;
    if (job.pending) return fail(BIGSI_ERR_STATE, "a score / presence request of this batch is still pending (call the matching _end)");
    bigsi_hip_index *ix = b->ix;
    const uint32_t nq = b->n_seqs;
    const uint64_t n_hits = hit_offsets[nq] - hit_offsets[0], h0 = hit_offsets[0];
    if (n_hits && !colours) return fail(BIGSI_ERR_INVALID, "colours is NULL");
    if (n_hits > 0xFFFFFFF0ull) return fail(BIGSI_ERR_INVALID, "too many hits for one call");
    TRY(host_counts(b));
    hipStream_t ps = ix->stream;
    if (!ordered) TRY(score_stream(ix, &ps));
    // host side: string offsets, per-sequence colour order, word pairs
    std::vector<uint32_t> &hit_seq = job.hit_seq, &hit_q = job.hit_q, &perm = job.perm, &order = job.order;      // hit_seq: k-mers of the hit's sequence (its string length); hit_q: which sequence
    std::vector<uint64_t> &hit_pos0 = job.hit_pos0;
    std::vector<PresencePair> pairs;
    std::vector<PresenceWave> waves;          // (k_presence_bits: a wavefront = up to 64 pairs of one query)
    std::vector<PresenceWave> few;            // (k_presence_bits_sparse: queries with few pairs, one entry each: lanes = k-mers there)
    // (tuning builds: queries with at most this many pairs go to k_presence_bits_sparse.  Interleaved A/B on the C5 shard, 259 hits in
    // 256 queries per batch: 245.5 -> 239.5 M lookups/s -- the kernel that wastes no lanes issues its requests faster and takes more
    // from the row-AND kernel it runs beside (0.974 -> 0.997 ms) than the one-live-lane form that trickles them.  Off.)
    static const int sparse_max = env_int("BIGSI_HIP_K5_SPARSE_MAX", 0);
    hit_seq.resize(n_hits); hit_q.resize(n_hits); perm.resize(n_hits); hit_pos0.resize(n_hits);
    pairs.clear();
    for (uint64_t t = 0; t < n_hits; t++) perm[t] = (uint32_t)t;
    uint64_t str = 0, alg = 0;
    uint32_t max_u = 0, max_n = 0;
    for (uint32_t q = 0; q < nq; q++) {
        if (hit_offsets[q + 1] < hit_offsets[q]) return fail(BIGSI_ERR_INVALID, "hit_offsets must be non-decreasing");
        const uint64_t lo = hit_offsets[q] - h0, hi = hit_offsets[q + 1] - h0;
        for (uint64_t t = lo; t < hi; t++) {
            if (colours[h0 + t] >= ix->n_cols) return fail(BIGSI_ERR_RANGE, "colour %u >= num_cols", colours[h0 + t]);
            hit_seq[t] = b->h_num_kmers[q];
            hit_q[t] = q;
            hit_pos0[t] = b->pos_off[q];
            string_offsets[t] = str;
            // every string starts on a 16-byte boundary (16-character stores); packed: whole 8-byte words
            str += packed ? round_up(b->h_num_kmers[q], 64) / 8 : round_up(b->h_num_kmers[q], 16);
        }
        if (hi == lo || b->h_num_kmers[q] == 0) continue;
        const size_t first_pair = pairs.size();
        // the hit lists fetch_hits returns are in colour order already: sort only what is not
        bool sorted = true;
        for (uint64_t t = lo + 1; t < hi && sorted; t++) sorted = colours[h0 + t - 1] < colours[h0 + t];
        if (!sorted) {
            order.resize(hi - lo);
            for (uint64_t t = lo; t < hi; t++) order[t - lo] = (uint32_t)t;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return colours[h0 + x] < colours[h0 + y]; });
        }
        uint64_t words = 0, last_word = ~0ull;
        for (uint64_t r = 0; r < hi - lo; r++) {
            const uint32_t src = sorted ? (uint32_t)(lo + r) : order[r];
            const uint32_t c = colours[h0 + src];
            if (r && c == colours[h0 + (sorted ? (uint32_t)(lo + r - 1) : order[r - 1])])
                return fail(BIGSI_ERR_INVALID, "colour %u listed twice for sequence %u", c, q);
            const uint32_t wp = c >> 7;
            if (pairs.size() == first_pair || pairs.back().wpair != wp) pairs.push_back(PresencePair{wp, (uint32_t)(lo + r), 0ull, 0ull, q, 0u});
            const uint64_t bit = 1ull << bit_of_col(c & 63u);
            if (c & 64u) pairs.back().mask_hi |= bit;
            else pairs.back().mask_lo |= bit;
            perm[lo + r] = src;
            if ((uint64_t)(c >> 6) != last_word) { words++; last_word = c >> 6; }
        }
        if (sparse_max > 0 && pairs.size() - first_pair <= (size_t)sparse_max) few.push_back(PresenceWave{(uint32_t)first_pair, (uint32_t)(pairs.size() - first_pair)});
        else for (size_t f = first_pair; f < pairs.size(); f += 64) waves.push_back(PresenceWave{(uint32_t)f, (uint32_t)std::min<size_t>(64, pairs.size() - f)});
        max_u = std::max(max_u, b->h_num_unique[q]);
        max_n = std::max(max_n, b->h_num_kmers[q]);
        alg += (uint64_t)b->h_num_unique[q] * b->run_h * words * 8 +
               (hi - lo) * (packed ? round_up(b->h_num_kmers[q], 64) / 8 + sizeof(bigsi_hip_hit_score) : (uint64_t)b->h_num_kmers[q]);
    }
    string_offsets[n_hits] = str;
    if (check_capacity && str > out_capacity)
        return fail(BIGSI_ERR_CAPACITY, "%s buffer holds %llu bytes, %llu needed", packed ? "bit" : "string", (unsigned long long)out_capacity, (unsigned long long)str);
    job.packed = packed;
    job.n_hits = n_hits;
    job.str = str;
    job.o_scores = round_up(str, 256);
    job.device_work = false;
    job.pending = true;
    if (n_hits == 0 || str == 0) return BIGSI_OK;          // (hits of sequences without k-mers: nothing to extract, all-zero records)
    if (b->run_h != ix->h) { job.pending = false; return fail(BIGSI_ERR_STATE, "num_hashes changed since the batch was run"); }
    job.pending = false;                                   // (set again once everything is enqueued)
    // device input: [str_off | hit_seq | perm | hit_pos0 | pairs | hit_q | found | unique] in one upload from pinned memory
    const size_t o_str = 0, o_seq = round_up(o_str + (n_hits + 1) * 8, 256);
    const size_t o_perm = round_up(o_seq + n_hits * 4, 256), o_pos0 = round_up(o_perm + n_hits * 4, 256), o_pairs = round_up(o_pos0 + n_hits * 8, 256);
    const size_t o_waves = round_up(o_pairs + pairs.size() * sizeof(PresencePair), 256);
    const size_t o_few = o_waves + waves.size() * sizeof(PresenceWave);
    const size_t o_q = round_up(o_few + few.size() * sizeof(PresenceWave), 256), o_hoff = round_up(o_q + n_hits * 4, 256);
    // (K6) per rank: k-mers the hit found, unique k-mers of its sequence
    const size_t o_found = round_up(o_hoff + (nq + 1) * 8ull, 256), o_uniq = round_up(o_found + (packed ? n_hits * 4 : 0), 256);
    const size_t in_bytes = packed ? o_uniq + n_hits * 4 : o_hoff + (nq + 1) * 8ull;
    TRY(pinned_reserve(&job.h_in, &job.h_in_cap, in_bytes));
    uint8_t *stage = static_cast<uint8_t *>(job.h_in);
    // the device works in rank order (each sequence's hits by colour): string offset of the hit with that rank; the other per-hit
    // values (k-mers, map start, sequence) are the same for all hits of a sequence, whose hits keep their index range
    for (uint64_t r = 0; r < n_hits; r++) reinterpret_cast<uint64_t *>(stage + o_str)[r] = string_offsets[perm[r]];
    memcpy(stage + o_seq, hit_seq.data(), n_hits * 4);
    memcpy(stage + o_perm, perm.data(), n_hits * 4);
    memcpy(stage + o_pos0, hit_pos0.data(), n_hits * 8);
    memcpy(stage + o_pairs, pairs.data(), pairs.size() * sizeof(PresencePair));
    memcpy(stage + o_waves, waves.data(), waves.size() * sizeof(PresenceWave));
    memcpy(stage + o_few, few.data(), few.size() * sizeof(PresenceWave));
    memcpy(stage + o_q, hit_q.data(), n_hits * 4);
    for (uint32_t q = 0; q <= nq; q++) reinterpret_cast<uint64_t *>(stage + o_hoff)[q] = hit_offsets[q] - h0;
    if (packed)
        for (uint64_t r = 0; r < n_hits; r++) {
            const uint32_t u = b->h_num_unique[hit_q[r]];
            reinterpret_cast<uint32_t *>(stage + o_uniq)[r] = u;
            reinterpret_cast<uint32_t *>(stage + o_found)[r] = counts ? counts[h0 + perm[r]] : u;
        }
    const uint32_t n_chunks = (uint32_t)ceil_div(std::max<uint32_t>(max_u, 1), 16);
    const size_t out_bytes = packed ? job.o_scores + n_hits * sizeof(bigsi_hip_hit_score) : str;
    TRY(pinned_reserve(&job.h_out, &job.h_out_cap, out_bytes));
    TRY(b->pres_in.reserve(in_bytes));
    TRY(b->pres_bits.reserve((size_t)round_up(n_hits, 32) * n_chunks * 2 + 16));
    TRY(b->pres_out.reserve(out_bytes));
    const uint64_t max_pieces = (b->total_pos >> 4) + nq + 1;      // marks | list of unmarked pieces | their number
    TRY(b->pres_desc.reserve(round_up(max_pieces * 4, 256) + max_pieces * 8 + 256));
    uint32_t *d_marks = b->pres_desc.as<uint32_t>();
    uint2 *d_listed = reinterpret_cast<uint2 *>(b->pres_desc.as<uint8_t>() + round_up(max_pieces * 4, 256));
    uint32_t *d_listed_n = reinterpret_cast<uint32_t *>(b->pres_desc.as<uint8_t>() + round_up(max_pieces * 4, 256) + max_pieces * 8);
    if (!job.done) HIP_TRY(hipEventCreateWithFlags(&job.done, hipEventDisableTiming));
    b->idle = false;
    if (ps == ix->sc_stream) ix->sc_pending = true;      // from here on something of this index may be queued there
    HIP_TRY(hipMemcpyAsync(b->pres_in.p, stage, in_bytes, hipMemcpyHostToDevice, ps));
    const uint8_t *din = b->pres_in.as<uint8_t>();
    EventPair ep{};
    TRY(ev_begin(ix, &ep, ps));
    if (b->marks_of_run != b->run_serial || b->marks_at != b->pres_desc.p) {
        // the piece marks depend on K1's position -> unique k-mer map alone: once per run of the batch, not once per call
        HIP_TRY(hipMemsetAsync(d_listed_n, 0, 4, ps));
        hipLaunchKernelGGL(k_presence_pieces, dim3(nq), dim3(kBlock), 0, ps, b->pos_unique.as<uint32_t>(), b->d_pos_off.as<uint64_t>(),
                           b->num_kmers.as<uint32_t>(), d_marks, d_listed_n, d_listed);
        b->marks_of_run = b->run_serial;
        b->marks_at = b->pres_desc.p;
    }
    const dim3 grid_a((unsigned)ceil_div(std::max<uint64_t>(waves.size(), 1), kBlock / 64), (unsigned)ceil_div(std::max<uint32_t>(max_u, 1), 16), 1);
    static const int k5_waves = env_int("BIGSI_HIP_K5_WAVES", 2);
#define BIGSI_PRESENCE_ARGS                                                                                                        \
    grid_a, dim3(kBlock), 0, ps, ix->d_index, ix->stride_words, b->rows.as<uint64_t>(), b->d_pos_off.as<uint64_t>(),            \
        b->num_unique.as<uint32_t>(), ix->h, (uint32_t)waves.size(), (const PresenceWave *)(din + o_waves), (const PresencePair *)(din + o_pairs),   \
        b->pres_bits.as<uint16_t>(), n_chunks
#define COMMA ,
#define BIGSI_PRESENCE(H)                                                                          \
    if (k5_waves == 4) hipLaunchKernelGGL((k_presence_bits<H COMMA 4>), BIGSI_PRESENCE_ARGS);        \
    else hipLaunchKernelGGL((k_presence_bits<H COMMA 2>), BIGSI_PRESENCE_ARGS)
    if (!waves.empty()) switch (ix->h) {
    case 1: BIGSI_PRESENCE(1); break;
    case 2: BIGSI_PRESENCE(2); break;
    case 3: BIGSI_PRESENCE(3); break;
    case 4: BIGSI_PRESENCE(4); break;
    case 5: BIGSI_PRESENCE(5); break;
    default: BIGSI_PRESENCE(0); break;
    }
#undef BIGSI_PRESENCE
#undef BIGSI_PRESENCE_ARGS
#undef COMMA
#ifdef BIGSI_HIP_TUNING
    if (!few.empty()) {
        // queries with few pairs: lanes = unique k-mers (k_presence_bits_sparse)
        const dim3 grid_s((unsigned)few.size(), (unsigned)ceil_div(std::max<uint32_t>(max_u, 1), kBlock), 1);
#define BIGSI_PRESENCE_SPARSE(H)                                                                                                        \
    hipLaunchKernelGGL((k_presence_bits_sparse<H>), grid_s, dim3(kBlock), 0, ps, ix->d_index, ix->stride_words, b->rows.as<uint64_t>(),   \
                       b->d_pos_off.as<uint64_t>(), b->num_unique.as<uint32_t>(), ix->h, (const PresenceWave *)(din + o_few),             \
                       (const PresencePair *)(din + o_pairs), b->pres_bits.as<uint16_t>(), n_chunks)
        switch (ix->h) {
        case 1: BIGSI_PRESENCE_SPARSE(1); break;
        case 2: BIGSI_PRESENCE_SPARSE(2); break;
        case 3: BIGSI_PRESENCE_SPARSE(3); break;
        case 4: BIGSI_PRESENCE_SPARSE(4); break;
        case 5: BIGSI_PRESENCE_SPARSE(5); break;
        default: BIGSI_PRESENCE_SPARSE(0); break;
        }
#undef BIGSI_PRESENCE_SPARSE
    }
#endif
    HIP_TRY(hipGetLastError());
    if (packed) {
        // K6: position-ordered bits of every hit and its score record (written at the hit's place in the caller's order)
        hipLaunchKernelGGL(k_presence_score, dim3((unsigned)ceil_div(n_hits, kBlock)), dim3(kBlock), 0, ps, b->pres_bits.as<uint16_t>(), n_chunks, n_hits,
                           (const uint32_t *)(din + o_seq), (const uint64_t *)(din + o_pos0), (const uint32_t *)(din + o_q), d_marks,
                           b->pos_unique.as<uint32_t>(), (const uint64_t *)(din + o_str), b->pres_out.as<uint8_t>(),
                           (const uint32_t *)(din + o_found), (const uint32_t *)(din + o_uniq), (const uint32_t *)(din + o_perm),
                           reinterpret_cast<bigsi_score::HitScore *>(b->pres_out.as<uint8_t>() + job.o_scores));
    } else {
        const uint32_t pieces = (uint32_t)pow2_at_least(ceil_div(std::max<uint32_t>(max_n, 1), 16));      // power of two: shifts, not divisions
        const uint64_t groups = ceil_div(n_hits, kPresenceHits * kPresenceRounds), blocks = ceil_div(groups * pieces, kBlock);
        if (blocks > 0x7FFFFFFFull || groups > 0x7FFFFFFFull)
            return fail(BIGSI_ERR_INVALID, "presence request too large for one launch");
#define BIGSI_EXPAND_ARGS                                                                                                          \
    dim3((unsigned)blocks), dim3(kBlock), 0, ps, b->pres_bits.as<uint16_t>(), n_chunks, n_hits, pieces,                          \
        (const uint32_t *)(din + o_seq), (const uint64_t *)(din + o_pos0), (const uint64_t *)(din + o_str), b->pres_out.as<uint8_t>(),    \
        (const uint32_t *)(din + o_q), d_marks
        if (pieces >= 64) hipLaunchKernelGGL(k_presence_expand<true>, BIGSI_EXPAND_ARGS);
        else hipLaunchKernelGGL(k_presence_expand<false>, BIGSI_EXPAND_ARGS);
        hipLaunchKernelGGL(k_presence_expand_listed, dim3(1024), dim3(kBlock), 0, ps, b->pres_bits.as<uint16_t>(), n_chunks, d_listed_n, d_listed,
                           (const uint64_t *)(din + o_hoff), b->d_pos_off.as<uint64_t>(), b->num_kmers.as<uint32_t>(), b->pos_unique.as<uint32_t>(),
                           (const uint64_t *)(din + o_str), b->pres_out.as<uint8_t>());
#undef BIGSI_EXPAND_ARGS
    }
    HIP_TRY(hipGetLastError());
    TRY(ev_end(ix, &ep, ix->ev_pr, ps));
    if (ep.a) ix->presence_bytes += alg;
    // one download: [strings or bits | (score records)] are contiguous in pres_out
    HIP_TRY(hipMemcpyAsync(job.h_out, b->pres_out.p, out_bytes, hipMemcpyDeviceToHost, ps));
    HIP_TRY(hipEventRecord(job.done, ps));
    if (ps == ix->sc_stream) ix->sc_pending = true;
    job.device_work = true;
    job.pending = true;
    return BIGSI_OK;
}

static int presence_end(bigsi_hip_batch *b, uint8_t *out, uint64_t out_capacity, bigsi_hip_hit_score *scores)
{
    if (!b) return fail(BIGSI_ERR_INVALID, "NULL batch");
    PresJob &job = b->job;
    if (!job.pending) return fail(BIGSI_ERR_STATE, "no score / presence request of this batch is pending");
    TRY(use_device(b->ix));
    // arguments first: a call that fails on them leaves the request pending (its results stay staged in job.h_out), so the
    // caller can repeat _end with a larger buffer
    if (job.packed && !scores) return fail(BIGSI_ERR_INVALID, "scores is NULL");
    if (job.str > out_capacity)
        return fail(BIGSI_ERR_CAPACITY, "%s buffer holds %llu bytes, %llu needed", job.packed ? "bit" : "string", (unsigned long long)out_capacity, (unsigned long long)job.str);
    if (job.n_hits && job.device_work && !out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    if (job.device_work) HIP_TRY(hipEventSynchronize(job.done));
    job.pending = false;
    if (job.n_hits == 0) return BIGSI_OK;
    if (!job.device_work) {
        if (job.packed) memset(scores, 0, job.n_hits * sizeof(bigsi_hip_hit_score));
        return BIGSI_OK;
    }
    memcpy(out, job.h_out, job.str);
    if (job.packed) memcpy(scores, static_cast<const uint8_t *>(job.h_out) + job.o_scores, job.n_hits * sizeof(bigsi_hip_hit_score));
    return BIGSI_OK;
}

extern "C" int bigsi_hip_batch_presence_hits(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, uint8_t *out,
                                             uint64_t out_capacity, uint64_t *string_offsets)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    TRY(presence_begin(b, hit_offsets, colours, nullptr, false, false, out_capacity, true, string_offsets));
    return presence_end(b, out, out_capacity, nullptr);
}

extern "C" int bigsi_hip_batch_score_hits(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts,
                                          uint8_t *bits, uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!scores) return fail(BIGSI_ERR_INVALID, "scores is NULL");
    TRY(presence_begin(b, hit_offsets, colours, counts, true, false, bits_capacity, true, bit_offsets));
    return presence_end(b, bits, bits_capacity, scores);
}

extern "C" int bigsi_hip_batch_score_hits_begin(bigsi_hip_batch *b, const uint64_t *hit_offsets, const uint32_t *colours, const uint32_t *counts,
                                                uint32_t flags, uint64_t *bit_offsets)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    return presence_begin(b, hit_offsets, colours, counts, true, (flags & BIGSI_SCORE_ORDERED) != 0, 0, false, bit_offsets);
}

extern "C" int bigsi_hip_batch_score_hits_end(bigsi_hip_batch *b, uint8_t *bits, uint64_t bits_capacity, bigsi_hip_hit_score *scores)
{
    BIGSI_ENTER(b ? b->ix : nullptr);
    if (!scores) return fail(BIGSI_ERR_INVALID, "scores is NULL");
    return presence_end(b, bits, bits_capacity, scores);
}

// Scorer.score for presence strings the caller already holds as bits: no index involved (one upload, K6's scoring kernel,
// one download).
extern "C" int bigsi_hip_score_presence(int device, const uint8_t *bits, const uint64_t *bit_offsets, const uint32_t *num_kmers,
                                        const uint32_t *found, const uint32_t *unique, uint64_t n, bigsi_hip_hit_score *scores)
{
    if (n == 0) return BIGSI_OK;
    if (!bits || !bit_offsets || !num_kmers || !scores) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (n > 0x7FFFFFFFull * kBlock) return fail(BIGSI_ERR_INVALID, "too many strings for one call");
    uint64_t end = 0;
    for (uint64_t t = 0; t < n; t++) {
        if (bit_offsets[t] & 7u) return fail(BIGSI_ERR_INVALID, "bit_offsets[%llu] is not a multiple of 8", (unsigned long long)t);
        end = std::max(end, bit_offsets[t] + round_up(num_kmers[t], 64) / 8);
    }
    HIP_TRY(hipSetDevice(device));
    const size_t o_off = round_up(end, 256), o_n = round_up(o_off + n * 8, 256), o_f = round_up(o_n + n * 4, 256), o_u = round_up(o_f + n * 4, 256);
    const size_t o_out = round_up(o_u + n * 4, 256), total = o_out + n * sizeof(bigsi_hip_hit_score);
    std::vector<uint8_t> stage(o_out, 0);
    memcpy(stage.data(), bits, end);
    memcpy(stage.data() + o_off, bit_offsets, n * 8);
    memcpy(stage.data() + o_n, num_kmers, n * 4);
    if (found) memcpy(stage.data() + o_f, found, n * 4);
    if (unique) memcpy(stage.data() + o_u, unique, n * 4);
    uint8_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, total));
    hipError_t e = hipMemcpy(d, stage.data(), o_out, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_score_packed, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, 0, d, (const uint64_t *)(d + o_off), n,
                           (const uint32_t *)(d + o_n), found ? (const uint32_t *)(d + o_f) : nullptr, unique ? (const uint32_t *)(d + o_u) : nullptr,
                           (const uint32_t *)nullptr, reinterpret_cast<bigsi_score::HitScore *>(d + o_out));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(scores, d + o_out, n * sizeof(bigsi_hip_hit_score), hipMemcpyDeviceToHost);
    hipError_t e2 = hipFree(d); (void)e2;
    if (e != hipSuccess) return fail(BIGSI_ERR_HIP, "bigsi_hip_score_presence: %s", hipGetErrorString(e));
    return BIGSI_OK;
}

// ------------------------------------------------------------------------------ one-call / streaming searches: stage, export, collect
int bigsi_batch_stage(bigsi_hip_index *ix, bigsi_hip_batch **pb, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    TRY(check_batch_args(ix, seqs, offsets, n_seqs, k));
    TRY(use_device(ix));
    bigsi_hip_batch *b = *pb;
    if (!b) {
        b = new (std::nothrow) bigsi_hip_batch();
        if (!b) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
        b->ix = ix;
        *pb = b;
    } else {
        if (b->elements) return fail(BIGSI_ERR_STATE, "a batch of explicit k-mers cannot be reloaded: create a new one");
        TRY(batch_quiesce(b));                                         // this batch's earlier run ...
        if (b->exp_serial) TRY(export_wait(b));                        // ... and the export of its results
    }
    return batch_load(b, seqs, offsets, n_seqs, k, true);
}

// the pinned block an export writes, the flag its last workgroup raises, and the export's serial number
static int export_prepare(bigsi_hip_batch *b, hipStream_t st)
{
    HitBufs &hb = b->hits;
    const uint32_t n = b->n_seqs;
    // (room in the pinned block for the hit lists the export carries along: 16 per sequence -- a read stream whose reads match a
    // handful of samples each then never takes the slower fetch route; the block is only written as far as there are hits)
    const uint64_t spec = std::min<uint64_t>(std::max<uint64_t>(1024, 16ull * n), std::min<uint64_t>(hb.capacity(), 1u << 20));
    const size_t o_uniq = (n + 2ull) * 8, o_col = o_uniq + ((3ull * n + 1) & ~1ull) * 4, bytes = o_col + 8 * spec;
    TRY(pinned_reserve(&b->pin_out, &b->pin_out_cap, bytes));
    // completion: the kernel's last workgroup writes the export's serial into a pinned word the host spins on (no event record, no
    // hipEventSynchronize: a 1 kbp query's call is ~57 us, of which an event wait is several).  BIGSI_HIP_EXPORT_FLAG=0 (tuning
    // builds): the event, as before.
    static const int use_flag = env_int("BIGSI_HIP_EXPORT_FLAG", 1);
    if (use_flag && !b->pin_flag) {
        HIP_TRY(hipHostMalloc((void **)&b->pin_flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
        *b->pin_flag = 0;
        TRY(b->exp_count.reserve(256));
        HIP_TRY(hipMemsetAsync(b->exp_count.p, 0, 256, st));
    }
    if (!use_flag && !b->exp_done) HIP_TRY(hipEventCreateWithFlags(&b->exp_done, hipEventDisableTiming));
    b->exp_spec = (uint32_t)spec;
    b->exp_serial++;
    b->exp_flagged = use_flag != 0;
    b->exp_stream = st;
    return BIGSI_OK;
}

int bigsi_batch_export(bigsi_hip_batch *b)
{
    if (!b || !b->ran) return fail(BIGSI_ERR_STATE, "bigsi_hip_batch_run has not completed for this batch");
    if (!b->compacted) return fail(BIGSI_ERR_STATE, "internal: export of a run without hit lists");
    if (b->exported_inline) return BIGSI_OK;          // (one read: its kernel wrote the block and raises the flag)
    HitBufs &hb = b->hits;
    hipStream_t st = b->run_stream ? b->run_stream : b->ix->stream;
    const uint32_t n = b->n_seqs;
    TRY(export_prepare(b, st));
    const uint64_t spec = b->exp_spec;
    const bool use_flag = b->exp_flagged;
    if (b->fused_run) {
        // a read run: its export also puts the hit lists in query order (k_export_reads); workgroups own ranges of queries
        // (queries per workgroup, A/B at 1000 reads in one call: 1024 -> 51.2 us, 256 -> 47.7, 64 -> 47.8)
        static const int per_wg = env_int("BIGSI_HIP_EXPORT_READS_PER_WG", kBlock);
        const unsigned rgrid = (unsigned)std::min<uint64_t>(ceil_div(n, (uint64_t)std::max(per_wg, 1)), 64);
        hipLaunchKernelGGL(k_export_reads, dim3(std::max(rgrid, 1u)), dim3(kBlock), 0, st, hb.q_start.as<uint64_t>(), hb.q_cnt.as<uint32_t>(), n,
                           b->uniq.as<uint32_t>(), hb.col(), hb.cnt(), hb.capacity(), (uint32_t)spec, static_cast<uint64_t *>(b->pin_out), b->exp_count.as<uint32_t>(),
                           (volatile uint64_t *)(use_flag ? b->pin_flag : nullptr), b->exp_serial);
        HIP_TRY(hipGetLastError());
        if (!use_flag) HIP_TRY(hipEventRecord(b->exp_done, st));
        return BIGSI_OK;
    }
    // the hits it carries along speculatively are few: one workgroup unless the batch is large (one workgroup needs no counter)
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(std::max<uint64_t>(3ull * n, spec), 4 * kBlock), 64);
    hipLaunchKernelGGL(k_export_results, dim3(std::max(grid, 1u)), dim3(kBlock), 0, st, hb.hit_off.as<uint64_t>(), n, b->fused_run ? 1u : 0u, b->uniq.as<uint32_t>(),
                       hb.col(), hb.cnt(), (uint32_t)spec, static_cast<uint64_t *>(b->pin_out), b->exp_count.as<uint32_t>(),
                       (volatile uint64_t *)(use_flag ? b->pin_flag : nullptr), b->exp_serial);
    HIP_TRY(hipGetLastError());
    if (!use_flag) HIP_TRY(hipEventRecord(b->exp_done, st));
    return BIGSI_OK;
}

// host-side wait for the last export of the batch: spin on the pinned flag for a while (a latency-bound call is over in tens of
// microseconds), then block on the stream (a throughput batch takes milliseconds: no point burning a core)
static int export_wait(bigsi_hip_batch *b)
{
    if (!b->exp_serial) return fail(BIGSI_ERR_STATE, "internal: nothing exported");
    if (!b->exp_flagged) {
        HIP_TRY(hipEventSynchronize(b->exp_done));
        return BIGSI_OK;
    }
    volatile uint64_t *f = b->pin_flag;
    const uint64_t want = b->exp_serial;
    for (uint32_t spins = 0; *f != want; spins++) {
        __builtin_ia32_pause();
        if (spins >= 20000) {                    // ~0.5 ms
            HIP_TRY(hipStreamSynchronize(b->exp_stream));
            if (*f != want) return fail(BIGSI_ERR_HIP, "internal: export finished without raising its flag");
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return BIGSI_OK;
}

// outputs as bigsi_hip_batch_fetch_unique + fetch_hits (hit_offsets relative to this batch, n_seqs + 1 entries)
int bigsi_batch_collect(bigsi_hip_batch *b, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers, uint64_t *hit_offsets,
                        uint32_t *colours, uint32_t *counts, uint64_t capacity)
{
    if (!b) return fail(BIGSI_ERR_STATE, "internal: nothing exported");
    TRY(use_device(b->ix));
    TRY(export_wait(b));
    CALL_MARK(4);
    HitBufs &hb = b->hits;
    const uint32_t n = b->n_seqs;
    const uint64_t *off = static_cast<const uint64_t *>(b->pin_out);
    const uint64_t total = off[n];
    if (total > hb.capacity() || (b->fused_run && total > b->exp_spec)) {
        // rare: hit lists that outgrew the device buffers (the general route regrows them and repeats the launch), or a read
        // run with more hits than its export carried along (their query order is made on the host)
        TRY(bigsi_hip_batch_fetch_unique(b, num_kmers, num_unique, min_kmers));
        const int rc2 = bigsi_hip_batch_fetch_hits(b, hit_offsets, colours, counts, capacity);
        b->idle = !b->job.pending;
        return rc2;
    }
    b->idle = !b->job.pending;          // the export ran behind everything the run queued on its stream
    const uint32_t *u32 = reinterpret_cast<const uint32_t *>(off + n + 2);
    b->h_uniq.assign(u32, u32 + 3ull * n);
    b->h_num_kmers.assign(u32, u32 + n);
    b->h_num_unique.assign(u32 + n, u32 + 2ull * n);
    b->h_min_kmers.assign(u32 + 2ull * n, u32 + 3ull * n);
    b->host_counts_valid = true;
    if (num_kmers) memcpy(num_kmers, u32, n * 4ull);
    if (num_unique) memcpy(num_unique, u32 + n, n * 4ull);
    if (min_kmers) memcpy(min_kmers, u32 + 2ull * n, n * 4ull);
    if (hit_offsets) memcpy(hit_offsets, off, (n + 1ull) * 8);
    if (total > capacity)
        return fail(BIGSI_ERR_CAPACITY, "hit buffers hold %llu entries, %llu needed", (unsigned long long)capacity, (unsigned long long)total);
    const uint32_t *pcol = u32 + ((3ull * n + 1) & ~1ull), *pcnt = pcol + b->exp_spec;
    const uint64_t m = std::min<uint64_t>(total, b->exp_spec);
    if (m && colours) memcpy(colours, pcol, m * 4);
    if (m && counts) memcpy(counts, pcnt, m * 4);
    if (total > m) {          // more hits than the export carried along: the rest by plain copies
        if (colours) HIP_TRY(hipMemcpy(colours + m, hb.col() + m, (total - m) * 4, hipMemcpyDeviceToHost));
        if (counts) HIP_TRY(hipMemcpy(counts + m, hb.cnt() + m, (total - m) * 4, hipMemcpyDeviceToHost));
    }
    return BIGSI_OK;
}

// one-shot KmerSignatureIndex.lookup for an explicit k-mer list: every k-mer is its own one-window sequence, so K1's
// row ids land contiguously (u x h) and one k_lookup launch covers them all.
extern "C" int bigsi_hip_lookup(bigsi_hip_index *ix, const char *kmers, uint32_t k, uint64_t u, uint8_t *out_rows)
{
    BIGSI_ENTER(ix);
    if (!ix || (u && (!kmers || !out_rows))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    if (u == 0) return BIGSI_OK;
    if (u > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "too many k-mers for one call");
    if (ix->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    std::vector<uint64_t> off(u + 1);
    for (uint64_t i = 0; i <= u; i++) off[i] = i * k;
    bigsi_hip_batch *b = nullptr;
    TRY(bigsi_hip_batch_create(ix, kmers, off.data(), (uint32_t)u, k, &b));
    int rc = run_kmerize(b, 1.0);
    if (rc == BIGSI_OK) rc = k1_publish(b);
    const uint64_t wv = ix->wv(), rb = ix->rb();
    if (rc == BIGSI_OK) rc = b->scratch.reserve((size_t)u * wv * 8);
    if (rc == BIGSI_OK) {
        const uint64_t wblocks = ceil_div(wv, kBlock);
        if (wblocks * u > 0x7FFFFFFFull) rc = fail(BIGSI_ERR_INVALID, "lookup too large for one launch");
        else {
            hipLaunchKernelGGL(k_lookup, dim3((unsigned)(wblocks * u)), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, (uint32_t)wv,
                               b->rows.as<uint64_t>(), ix->h, (uint32_t)u, b->scratch.as<uint64_t>());
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpy2DAsync(out_rows, rb, b->scratch.p, wv * 8, rb, u, hipMemcpyDeviceToHost, ix->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
            if (e != hipSuccess) rc = fail(BIGSI_ERR_HIP, "bigsi_hip_lookup: %s", hipGetErrorString(e));
        }
    }
    bigsi_hip_batch_destroy(b);
    return rc;
}

// KmerSignatureIndex.lookup for explicit, already canonical elements of any byte lengths (non-ASCII k-mers): every element is
// its own one-position sequence of an element batch, k_rows_raw hashes them, one k_lookup launch ANDs their rows.
extern "C" int bigsi_hip_lookup_raw(bigsi_hip_index *ix, const char *blob, const uint64_t *elem_offsets, uint64_t u, uint8_t *out_rows)
{
    BIGSI_ENTER(ix);
    if (!ix || (u && (!elem_offsets || !out_rows))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (u == 0) return BIGSI_OK;
    if (u > 0x7FFFFFFFull) return fail(BIGSI_ERR_INVALID, "too many k-mers for one call");
    if (ix->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    std::vector<uint64_t> one(u + 1);
    std::vector<uint32_t> zero(u, 0u);
    for (uint64_t i = 0; i <= u; i++) one[i] = i;
    bigsi_hip_batch *b = nullptr;
    TRY(bigsi_hip_batch_create_elements(ix, blob, elem_offsets, one.data(), zero.data(), one.data(), (uint32_t)u, &b));
    int rc = run_kmerize(b, 1.0);
    if (rc == BIGSI_OK) rc = k1_publish(b);
    const uint64_t wv = ix->wv(), rb = ix->rb();
    if (rc == BIGSI_OK) rc = b->scratch.reserve((size_t)u * wv * 8);
    if (rc == BIGSI_OK) {
        const uint64_t wblocks = ceil_div(wv, kBlock);
        if (wblocks * u > 0x7FFFFFFFull) rc = fail(BIGSI_ERR_INVALID, "lookup too large for one launch");
        else {
            hipLaunchKernelGGL(k_lookup, dim3((unsigned)(wblocks * u)), dim3(kBlock), 0, ix->stream, ix->d_index, ix->stride_words, (uint32_t)wv,
                               b->rows.as<uint64_t>(), ix->h, (uint32_t)u, b->scratch.as<uint64_t>());
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpy2DAsync(out_rows, rb, b->scratch.p, wv * 8, rb, u, hipMemcpyDeviceToHost, ix->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
            if (e != hipSuccess) rc = fail(BIGSI_ERR_HIP, "bigsi_hip_lookup_raw: %s", hipGetErrorString(e));
        }
    }
    bigsi_hip_batch_destroy(b);
    return rc;
}
