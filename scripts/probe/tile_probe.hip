// tile_probe.hip -- what does the MEMORY SYSTEM give the transpose's access pattern as a function of the RUN LENGTHS, with no
// transpose in the way?  (measurement tool under scripts/, not part of the product; built by scripts/probe/build.sh)
//
// The build's transpose (k_transpose_tiles) reads n filters of m bits (pitch_in bytes apart) and writes m rows of n bits
// (pitch_out bytes apart).  A workgroup that owns a tile of (8 L_in rows) x (8 L_out columns) reads 8 L_out runs of L_in bytes
// and writes 8 L_in runs of L_out bytes.  This probe moves exactly those runs -- 16 bytes per lane, 8 loads in flight per
// lane, non-temporal, whatever was loaded is stored: the traffic pattern without the bit work or the LDS -- for any
// (L_in, L_out), also shapes no on-chip buffer could hold, to see what longer runs WOULD buy:
//     tile_probe <m_bits> <n_cols> [super]      prints a table of GB/s (in + out) over L_in x L_out, plus read-only / write-only rows
// Tiles are visited in supertiles of <super> x <super> tiles (row-major inside; narrower and taller when the matrix has fewer
// tile columns), as the kernel does.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
constexpr int kThreads = 512;

// mode: 0 = read + write, 1 = read only, 2 = write only
template <int MODE, int kInFlight>
__global__ __launch_bounds__(kThreads) void k_move_runs(const uint8_t *__restrict__ in, uint64_t pitch_in, uint8_t *__restrict__ out, uint64_t pitch_out,
                                                        uint64_t m_bytes /* bytes of a filter */, uint64_t n_bytes /* bytes of a row */, uint32_t l_in,
                                                        uint32_t l_out, uint32_t tiles_r, uint32_t tiles_c, uint32_t super, u64x2 *__restrict__ sink)
{
    // block -> tile, supertiles of sup_w x sup_h tiles (sup_w = min(super, tiles_c): no blocks wasted beyond a narrow matrix's edge)
    const uint32_t sup_w = super < tiles_c ? super : tiles_c;
    uint32_t sup_h = super * super / sup_w;
    if (sup_h > tiles_r) sup_h = tiles_r;
    const uint32_t sup_c = (tiles_c + sup_w - 1) / sup_w;
    const uint64_t per_sup = (uint64_t)sup_w * sup_h;
    const uint64_t s = blockIdx.x / per_sup;
    const uint32_t within = (uint32_t)(blockIdx.x % per_sup);
    const uint64_t tile_r = (s / sup_c) * sup_h + within / sup_w, tile_c = (s % sup_c) * sup_w + within % sup_w;
    if (tile_r >= tiles_r || tile_c >= tiles_c) return;
    const uint64_t byte0 = tile_r * l_in;                 // offset inside every filter
    const uint64_t col0 = tile_c * 8ull * l_out;          // first filter
    const uint64_t row0 = tile_r * 8ull * l_in;           // first row
    const uint64_t obyte0 = tile_c * (uint64_t)l_out;     // offset inside every row
    const uint32_t pin = l_in / 16, pout = l_out / 16;    // 16-byte pieces per run
    const uint64_t pieces = (uint64_t)l_in * l_out * 8 / 16;
    const uint64_t n_cols = n_bytes * 8, m_rows = m_bytes * 8;
    u64x2 acc = {0ull, 0ull};
    for (uint64_t base = 0; base < pieces; base += (uint64_t)kThreads * kInFlight) {
        u64x2 v[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
            const uint64_t p = base + (uint64_t)u * kThreads + threadIdx.x;
            const uint64_t run = p / pin, part = p % pin;
            const bool ok = p < pieces && col0 + run < n_cols && byte0 + part * 16 + 16 <= m_bytes;
            if (MODE != 2) v[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + (col0 + run) * pitch_in + byte0 + part * 16)) : u64x2{0ull, 0ull};
            else v[u] = u64x2{p, base};
        }
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
            const uint64_t p = base + (uint64_t)u * kThreads + threadIdx.x;
            const uint64_t run = p / pout, part = p % pout;
            const bool ok = p < pieces && row0 + run < m_rows && obyte0 + part * 16 + 16 <= n_bytes;
            if (MODE != 1) {
                if (ok) __builtin_nontemporal_store(v[u], reinterpret_cast<u64x2 *>(out + (row0 + run) * pitch_out + obyte0 + part * 16));
            } else acc ^= v[u];
        }
    }
    if (MODE == 1 && acc.x == 0x1234567ull) sink[0] = acc;      // (never true in practice: keeps the loads alive)
}

// the kernel's PHASE STRUCTURE on its (128, 128) tile, still without bit work: a workgroup of 512 threads loads the 1024 x 1024-bit tile
// (16 pieces per lane, all in flight), passes it through 64 KB of LDS in two halves (write, barrier, read back, store: the barriers of
// k_transpose_tiles<2,2>), two workgroups per CU.  PERSIST: workgroups loop over tiles and issue the NEXT tile's loads before the
// last half's stores (software pipelining).
template <bool PERSIST>
__global__ __launch_bounds__(kThreads) void k_phased(const uint8_t *__restrict__ in, uint64_t pitch_in, uint8_t *__restrict__ out, uint64_t pitch_out,
                                                     uint64_t m_bytes, uint64_t n_bytes, uint32_t tiles_r, uint32_t tiles_c, uint32_t super, uint64_t n_tiles_padded)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[73728];          // (the kernel's 2 x 36 KB: two workgroups per CU)
    const uint32_t sup_w = super < tiles_c ? super : tiles_c;
    uint32_t sup_h = super * super / sup_w;
    if (sup_h > tiles_r) sup_h = tiles_r;
    const uint32_t sup_c = (tiles_c + sup_w - 1) / sup_w;
    const uint64_t per_sup = (uint64_t)sup_w * sup_h;
    const uint64_t n_cols = n_bytes * 8, m_rows = m_bytes * 8;
    u64x2 ld[16];
    auto tile_of = [&](uint64_t b, uint64_t &tile_r, uint64_t &tile_c) {
        const uint64_t sidx = b / per_sup;
        const uint32_t within = (uint32_t)(b % per_sup);
        tile_r = (sidx / sup_c) * sup_h + within / sup_w;
        tile_c = (sidx % sup_c) * sup_w + within % sup_w;
        return b < n_tiles_padded && tile_r < tiles_r && tile_c < tiles_c;
    };
    auto load_tile = [&](uint64_t tile_r, uint64_t tile_c) {
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const uint32_t p = it * kThreads + threadIdx.x, run = p >> 3, part = p & 7u;
            const uint64_t col = tile_c * 1024 + run, off = tile_r * 128 + part * 16;
            const bool ok = col < n_cols && off + 16 <= m_bytes;
            ld[it] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + col * pitch_in + off)) : u64x2{0ull, 0ull};
        }
    };
    uint64_t b = blockIdx.x, tile_r, tile_c;
    bool have = tile_of(b, tile_r, tile_c);
    if (have) load_tile(tile_r, tile_c);
    while (b < n_tiles_padded) {
        uint64_t nr = 0, nc = 0;
        const uint64_t nb = b + gridDim.x;
        const bool next = PERSIST && tile_of(nb, nr, nc);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const uint32_t p = it * kThreads + threadIdx.x, run = p >> 3, part = p & 7u;
                if ((int)(part >> 2) != half) continue;
                *reinterpret_cast<u64x2 *>(lds + (run * 4 + (part & 3u)) * 16) = ld[it];
            }
            __syncthreads();
            if (half == 1 && next) load_tile(nr, nc);            // the registers are free: the next tile's loads overlap this half's stores
            if (have) {
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const uint32_t q = it * kThreads + threadIdx.x, row = q >> 3, part = q & 7u;
                    const uint64_t r = tile_r * 1024 + half * 512 + row, off = tile_c * 128 + part * 16;
                    const u64x2 v = *reinterpret_cast<const u64x2 *>(lds + q * 16);
                    if (r < m_rows && off + 16 <= n_bytes) __builtin_nontemporal_store(v, reinterpret_cast<u64x2 *>(out + r * pitch_out + off));
                }
            }
        }
        if (!PERSIST) break;
        __syncthreads();
        b = nb;
        have = next;
        tile_r = nr;
        tile_c = nc;
        if (!next) {                       // (tiles beyond the matrix edge inside the last supertiles: keep walking)
            bool any = false;
            while (b < n_tiles_padded && !(any = tile_of(b, tile_r, tile_c))) b += gridDim.x;
            if (!any) break;
            have = true;
            load_tile(tile_r, tile_c);
        }
    }
}

// calibration: a straight copy, 16 bytes per lane, U pieces in flight per lane, each workgroup a contiguous block of the buffer
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u64x2 *__restrict__ in, u64x2 *__restrict__ out, uint64_t n)
{
    const uint64_t base = (uint64_t)blockIdx.x * 256 * U + threadIdx.x;
    u64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        if (base + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(in + base + u * 256) : in[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; u++)
        if (base + u * 256 < n) {
            if (NT) __builtin_nontemporal_store(v[u], out + base + u * 256);
            else out[base + u * 256] = v[u];
        }
}

__global__ void k_fill(uint64_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull;
}

int main(int argc, char **argv)
{
    const uint64_t m_bits = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t n_cols = argc > 2 ? strtoull(argv[2], nullptr, 10) : 8192ull;
    const uint32_t super = argc > 3 ? (uint32_t)atoi(argv[3]) : 32u;
    const uint64_t m_bytes = (m_bits + 7) / 8 / 16 * 16, n_bytes = (n_cols + 7) / 8 / 16 * 16;
    const uint64_t align_in = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1ull;      // filter pitch rounded up to this (1: packed, as rounds 3-5 measured)
    const uint64_t pitch_in = (m_bytes + align_in - 1) / align_in * align_in, pitch_out = (n_bytes + 127) / 128 * 128;
    uint8_t *in = nullptr, *out = nullptr;
    u64x2 *sink = nullptr;
    CK(hipMalloc(&in, pitch_in * n_cols + 4096));
    CK(hipMalloc(&out, pitch_out * m_bytes * 8 + 4096));
    CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint64_t *>(in), pitch_in * n_cols / 8);
    CK(hipMemset(out, 0, pitch_out * m_bytes * 8));
    CK(hipDeviceSynchronize());
    printf("matrix %llu rows x %llu columns: filters %.2f GB (pitch %llu), rows %.2f GB (pitch %llu); supertiles %u x %u; GB/s are in + out\n",
           (unsigned long long)m_bytes * 8, (unsigned long long)n_cols, pitch_in * n_cols / 1e9, (unsigned long long)pitch_in,
           pitch_out * m_bytes * 8 / 1e9, (unsigned long long)pitch_out, super, super);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    {   // calibration
        const uint64_t n16 = std::min(pitch_in * n_cols, pitch_out * m_bytes * 8) / 16;
        auto time_copy = [&](auto launch, const char *name) {
            std::vector<float> ms;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, 0));
                launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t = 0;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("straight copy %-28s %7.0f GB/s in + out\n", name, 2.0 * n16 * 16 / (ms[1] * 1e-3) / 1e9);
        };
        time_copy([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3((uint32_t)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const u64x2 *)in, (u64x2 *)out, n16); }, "4 in flight, non-temporal");
        time_copy([&] { hipLaunchKernelGGL((k_copy<8, true>), dim3((uint32_t)((n16 + 2047) / 2048)), dim3(256), 0, 0, (const u64x2 *)in, (u64x2 *)out, n16); }, "8 in flight, non-temporal");
        time_copy([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3((uint32_t)((n16 + 2047) / 2048)), dim3(256), 0, 0, (const u64x2 *)in, (u64x2 *)out, n16); }, "8 in flight, plain");
        time_copy([&] { hipLaunchKernelGGL((k_copy<1, false>), dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, 0, (const u64x2 *)in, (u64x2 *)out, n16); }, "1 in flight, plain");
        time_copy([&] { CK(hipMemcpyAsync(out, in, n16 * 16, hipMemcpyDeviceToDevice, 0)); }, "hipMemcpyAsync D2D");
    }
    {   // the kernel's phase structure on (128, 128), one tile per workgroup / persistent and pipelined
        const uint32_t tiles_r = (uint32_t)((m_bytes + 127) / 128), tiles_c = (uint32_t)((n_bytes + 127) / 128);
        const uint32_t sup_w = std::min(super, tiles_c), sup_h = std::min(super * super / sup_w, tiles_r);
        const uint64_t blocks = (uint64_t)((tiles_r + sup_h - 1) / sup_h) * ((tiles_c + sup_w - 1) / sup_w) * sup_w * sup_h;
        auto run = [&](bool persist, uint32_t grid, const char *name) {
            std::vector<float> ms;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, 0));
                if (persist) hipLaunchKernelGGL((k_phased<true>), dim3(grid), dim3(kThreads), 0, 0, in, pitch_in, out, pitch_out, m_bytes, n_bytes, tiles_r, tiles_c, super, blocks);
                else hipLaunchKernelGGL((k_phased<false>), dim3(grid), dim3(kThreads), 0, 0, in, pitch_in, out, pitch_out, m_bytes, n_bytes, tiles_r, tiles_c, super, blocks);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t = 0;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("phased (128, 128) through LDS, %-42s %7.0f GB/s in + out\n", name, 2.0 * m_bytes * n_cols / (ms[1] * 1e-3) / 1e9);
        };
        run(false, (uint32_t)blocks, "one tile per workgroup (the kernel's form)");
        run(true, 512, "persistent, 512 workgroups, pipelined");
        run(true, 1024, "persistent, 1024 workgroups, pipelined");
        run(true, 2048, "persistent, 2048 workgroups, pipelined");
    }
    {   // pieces in flight per lane (loads issued before the first store), read + write
        printf("read + write by pieces in flight per lane\n  (L_in, L_out)        1       2       4       8      16\n");
        const uint32_t shapes[][2] = {{128, 128}, {256, 128}, {128, 256}, {256, 256}, {1024, 1024}};
        for (auto &sh : shapes) {
            const uint32_t li = sh[0], lo = sh[1];
            if (li > m_bytes || lo > n_bytes) continue;
            const uint32_t tiles_r = (uint32_t)((m_bytes + li - 1) / li), tiles_c = (uint32_t)((n_bytes + lo - 1) / lo);
            const uint32_t sup_w = std::min(super, tiles_c), sup_h = std::min(super * super / sup_w, tiles_r);
            const uint64_t blocks = (uint64_t)((tiles_r + sup_h - 1) / sup_h) * ((tiles_c + sup_w - 1) / sup_w) * sup_w * sup_h;
            printf("  (%4u, %4u)   ", li, lo);
            for (int u = 1; u <= 16; u *= 2) {
                std::vector<float> ms;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0, 0));
#define LAUNCH_U(U) hipLaunchKernelGGL((k_move_runs<0, U>), dim3((uint32_t)blocks), dim3(kThreads), 0, 0, in, pitch_in, out, pitch_out, m_bytes, n_bytes, li, lo, tiles_r, tiles_c, super, sink)
                    if (u == 1) LAUNCH_U(1);
                    else if (u == 2) LAUNCH_U(2);
                    else if (u == 4) LAUNCH_U(4);
                    else if (u == 8) LAUNCH_U(8);
                    else LAUNCH_U(16);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float t = 0;
                    CK(hipEventElapsedTime(&t, e0, e1));
                    ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                printf(" %7.0f", 2.0 * m_bytes * n_cols / (ms[1] * 1e-3) / 1e9);
                fflush(stdout);
            }
            printf("\n");
        }
    }
    const uint32_t ls[] = {64, 128, 256, 512, 1024, 2048, 4096};
    for (int mode = 0; mode < 3; mode++) {
        printf("%s\n  L_in \\ L_out", mode == 0 ? "read + write" : mode == 1 ? "read only (GB/s of reads)" : "write only (GB/s of writes)");
        for (uint32_t lo : ls) printf(" %7u", lo);
        printf("\n");
        for (uint32_t li : ls) {
            if (li > m_bytes) continue;
            printf("  %12u", li);
            for (uint32_t lo : ls) {
                if (lo > n_bytes) { printf(" %7s", "-"); continue; }
                const uint32_t tiles_r = (uint32_t)((m_bytes + li - 1) / li), tiles_c = (uint32_t)((n_bytes + lo - 1) / lo);
                const uint32_t sup_w = std::min(super, tiles_c), sup_h = std::min(super * super / sup_w, tiles_r);
                const uint64_t blocks = (uint64_t)((tiles_r + sup_h - 1) / sup_h) * ((tiles_c + sup_w - 1) / sup_w) * sup_w * sup_h;
                if (blocks > 0x7FFFFFFFull) { printf(" %7s", "big"); continue; }
                std::vector<float> ms;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0, 0));
#define LAUNCH(M) hipLaunchKernelGGL((k_move_runs<M, 8>), dim3((uint32_t)blocks), dim3(kThreads), 0, 0, in, pitch_in, out, pitch_out, m_bytes, n_bytes, li, lo, tiles_r, tiles_c, super, sink)
                    if (mode == 0) LAUNCH(0);
                    else if (mode == 1) LAUNCH(1);
                    else LAUNCH(2);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float t = 0;
                    CK(hipEventElapsedTime(&t, e0, e1));
                    ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                const double bytes = (double)m_bytes * n_cols * (mode == 0 ? 2.0 : 1.0);
                printf(" %7.0f", bytes / (ms[1] * 1e-3) / 1e9);
                fflush(stdout);
            }
            printf("\n");
        }
    }
    return 0;
}
