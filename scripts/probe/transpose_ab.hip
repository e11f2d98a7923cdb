// transpose_ab.hip -- A/B of the build transpose's ingredients against its own skeleton (measurement tool, not part of the product).
// tile_probe's "phased" mover -- the kernel's loads, barriers, 2 x 36 KB of LDS and stores, without bit work -- runs 20 % faster than
// k_transpose_tiles<2,2> on the same box.  This file holds a copy of that kernel (from bigsi_amd/csrc/bigsi_kernels.hpp, which it
// includes for the butterflies) with its ingredients behind switches, to see which of them costs what:
//   VAR & 1   tiles row-major inside the supertile instead of XCD groups
//   VAR & 2   no phase 2 (no butterflies, no LDS traffic of theirs)
//   VAR & 4   16-byte LDS accesses in phases 1 and 3 (pitch 80 instead of 72)
// usage: transpose_ab <m_bits> <n_cols> [filter pitch alignment, default 128]
#include "../../bigsi_amd/csrc/bigsi_kernels.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace bigsi {
template <int RT, int CT, int VAR>
__global__ __launch_bounds__(kBlock * CT) void k_tr_var(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t w_first /* first column word written; even */,
    uint64_t n_words /* whole 64-column words to write */, const uint8_t *__restrict__ blooms /* filter of column 64 * w_first */,
    uint64_t bstride /* bytes between filters; multiple of 16 */, uint64_t nb /* valid bytes of a filter: ceil(m / 8) */,
    uint32_t rg, uint32_t cg /* tiles per XCD group along rows / columns: powers of two, rg * cg <= 128, cg <= sup_w */,
    uint32_t sup_w /* supertile width in tiles: a power of two <= 32 */)
{
    // ONE buffer per 512 x 512 tile: line L holds, before the transpose, the 64 row-bytes of column L and, after it, the 64
    // column-bytes of row L.  Block (cw, rc) of 64 x 64 bits sits at lines [64 cw, +64), bytes [8 rc, +8) and its transpose
    // belongs at lines [64 rc, +64), bytes [8 cw, +8) -- the place of block (rc, cw) -- so blocks are transposed in mirrored
    // pairs, each written where the other was read (37 KB of LDS per tile instead of 74).
    constexpr int kPitch = (VAR & 4) ? 80 : kTransposePitch;      // VAR & 4: 16-byte LDS accesses in phases 1 and 3 (pitch 80)
    __shared__ __attribute__((aligned(16))) uint8_t tiles[CT][kTransposeTile * kPitch];
    constexpr uint32_t kWordsPerBlock = 8 * CT;
    // workgroup -> tile in SUPERTILES of 1024 tiles, sup_w wide (32, or fewer when the matrix has fewer tile columns: no
    // workgroups wasted on tiles beyond its edge) and 1024 / sup_w high: the ~1000 workgroups resident at any time then read
    // long runs of each filter and write long runs of each row, instead of short pieces strided by a whole row or filter
    const uint64_t tiles_c = (n_words + kWordsPerBlock - 1) / kWordsPerBlock, sup_c = (tiles_c + sup_w - 1) / sup_w;
    const uint32_t sup_h = (uint32_t)(kTransposeSuper * kTransposeSuper) / sup_w;
    const uint64_t sup = blockIdx.x / (kTransposeSuper * kTransposeSuper);
    const uint32_t within = blockIdx.x % (kTransposeSuper * kTransposeSuper);
    // inside a supertile: groups of rg x cg neighbouring tiles go to the SAME XCD (block b runs on XCD b % 8), one right after
    // the other: row-neighbours share the 128-byte lines of the filters, column-neighbours those of the rows, and with
    // consecutive blocks they landed in different L2s (FETCH_SIZE showed every filter line read about twice)
    const uint32_t gsz = rg * cg, xcd = within & 7u, sl = within >> 3, t = sl % gsz, g = (sl / gsz) * 8u + xcd;
    const uint32_t gpr = sup_w / cg;                // groups per supertile row of groups
    const uint64_t tile_r = (VAR & 1) ? (sup / sup_c) * sup_h + within / sup_w : (sup / sup_c) * sup_h + (g / gpr) * rg + t % rg;      // VAR & 1: row-major inside the supertile
    const uint64_t tile_c = (VAR & 1) ? (sup % sup_c) * sup_w + within % sup_w : (sup % sup_c) * sup_w + (g % gpr) * cg + t / rg;
    if (tile_r * kTransposeTile * RT >= m || tile_c >= tiles_c) return;
    const uint64_t byte0 = tile_r * (kTransposeTile / 8) * RT;
    const uint64_t w0 = tile_c * kWordsPerBlock;
    const uint32_t words_here = (uint32_t)(n_words - w0 < kWordsPerBlock ? n_words - w0 : kWordsPerBlock);
    // this thread's 512-column tile (of the workgroup's CT) and its place among that tile's 256 threads
    const uint32_t ct = threadIdx.x / kBlock, tid = threadIdx.x % kBlock;
    uint8_t *tile = tiles[ct];
    const uint32_t cols_here = words_here > 8 * ct ? min(words_here - 8 * ct, 8u) * 64u : 0u;
    // phase 1: 64 bytes of each column's filter -> tile[col][0..64) (columns beyond the last word: zeros).  A 16-byte load
    // that starts inside the filter's pitch is always in bounds (pitch and offsets are multiples of 16); bytes past
    // ceil(m / 8), like bits past m inside the last byte, belong to rows >= m, which phase 3 never stores.
    // (RT = 2: 8 lanes x 16 bytes = one whole 128-byte line per filter and wave instruction; a thread's parts all have the
    // same index, so its loads all belong to the same one of the two stacked tiles)
    constexpr int kParts = 4 * RT, kLoads = kTransposeTile * kParts / kBlock;
    u64x2 ld[kLoads];
#pragma unroll
    for (int it = 0; it < kLoads; it++) {
        const uint32_t item = it * kBlock + tid, col = item / kParts, part = item % kParts;
        const uint64_t off = byte0 + part * 16;
        const bool ok = col < cols_here && off + 16 <= bstride && off < nb;
        const u64x2 *src = reinterpret_cast<const u64x2 *>(blooms + ((w0 + 8 * ct) * 64 + (ok ? col : 0)) * bstride + (ok ? off : 0));
        ld[it] = ok ? __builtin_nontemporal_load(src) : u64x2{0ull, 0ull};      // (plain loads / stores measured the same)
    }
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t mycol = bit_of_col(lane);
#pragma unroll
    for (int half = 0; half < RT; half++) {
    const uint64_t r0 = (tile_r * RT + half) * kTransposeTile;
    if (half) __syncthreads();                          // phase 3 of the first tile has read the buffer
#pragma unroll
    for (int it = 0; it < kLoads; it++) {
        const uint32_t item = it * kBlock + tid, col = item / kParts, part = item % kParts;
        if ((int)(part >> 2) != half) continue;
        uint64_t *d = reinterpret_cast<uint64_t *>(tile + col * kPitch + (part & 3u) * 16);
        if (VAR & 4) *reinterpret_cast<u64x2 *>(d) = ld[it];
        else {
            d[0] = ld[it].x;
            d[1] = ld[it].y;
        }
    }
    __syncthreads();
    // phase 2: the 36 unordered pairs {(cw, rc), (rc, cw)} of the 8 x 8 blocks, nine per wavefront
    if (!(VAR & 2))                                      // VAR & 2: no phase 2 (tiles moved, not transposed)
    for (uint32_t pi = wave; pi < 36; pi += kBlock / 64) {
        // pi -> (x, y) with x <= y: row y of the lower triangle starts at y (y + 1) / 2
        uint32_t y = 0;
        while ((y + 1) * (y + 2) / 2 <= pi) y++;
        const uint32_t x = pi - y * (y + 1) / 2;
        uint8_t *pa = tile + (x * 64) * kPitch + y * 8;      // block (cw = x, rc = y): lines of column word x
        uint8_t *pb = tile + (y * 64) * kPitch + x * 8;      // block (cw = y, rc = x)
        uint64_t va = *reinterpret_cast<const uint64_t *>(pa + mycol * kPitch);
        uint64_t vb = x != y ? *reinterpret_cast<const uint64_t *>(pb + mycol * kPitch) : 0ull;
        va = transpose64_lanes(by_column(va), lane);
        if (x != y) vb = transpose64_lanes(by_column(vb), lane);
        // every lane of the wavefront has read both blocks before any lane overwrites them (the butterflies in between are
        // wavefront-wide exchanges), and no other wavefront touches this pair
        *reinterpret_cast<uint64_t *>(pb + lane * kPitch) = va;      // rows of chunk y, column word x
        if (x != y) *reinterpret_cast<uint64_t *>(pa + lane * kPitch) = vb;
    }
    __syncthreads();
    // phase 3: tiles[..][row][0 .. 64) -> the rows' words [w_first + w0, +words_here): 4 * CT lanes per row, 16 bytes each
#pragma unroll
    for (int it = 0; it < kTransposeTile * 4 / kBlock; it++) {
        const uint32_t item = it * (kBlock * CT) + threadIdx.x, row = item / (4 * CT), part = item % (4 * CT);
        const uint64_t r = r0 + row;
        if (r >= m || part * 2 >= words_here) continue;
        const uint64_t *sp = reinterpret_cast<const uint64_t *>(tiles[part >> 2] + row * kPitch + (part & 3u) * 16);
        uint64_t *dst = index + r * stride_words + w_first + w0 + part * 2;
        if (part * 2 + 1 < words_here) __builtin_nontemporal_store((VAR & 4) ? *reinterpret_cast<const u64x2 *>(sp) : u64x2{sp[0], sp[1]}, reinterpret_cast<u64x2 *>(dst));
        else dst[0] = sp[0];
    }
    }
}
// The same kernel (RT = CT = 2), PERSISTENT and software-pipelined: a workgroup walks tiles b, b + gridDim.x, ... and issues the next
// tile's loads once the current tile's second half is in LDS -- before that half's butterflies and stores -- so the memory system has
// this workgroup's loads in flight while its vector ALUs transpose.  (167 VGPRs: one workgroup per CU; capped at 128 with
// __launch_bounds__(512, 4) it spills 152 bytes per lane and is slower still.)
struct TrTile { uint64_t tile_r, w0; uint32_t words_here; };

template <int VAR>
__global__ __launch_bounds__(kBlock * 2) void k_tr_pipe(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t w_first, uint64_t n_words, const uint8_t *__restrict__ blooms,
    uint64_t bstride, uint64_t nb, uint32_t rg, uint32_t cg, uint32_t sup_w, uint64_t n_blocks)
{
    constexpr int RT = 2, CT = 2, kPitch = kTransposePitch;
    __shared__ __attribute__((aligned(16))) uint8_t tiles[CT][kTransposeTile * kPitch];
    constexpr uint32_t kWordsPerBlock = 8 * CT;
    const uint64_t tiles_c = (n_words + kWordsPerBlock - 1) / kWordsPerBlock, sup_c = (tiles_c + sup_w - 1) / sup_w;
    const uint32_t sup_h = (uint32_t)(kTransposeSuper * kTransposeSuper) / sup_w;
    const uint32_t gsz = rg * cg, gpr = sup_w / cg;
    auto tile_of = [&](uint64_t b, TrTile &o) -> bool {
        const uint64_t sup = b / (kTransposeSuper * kTransposeSuper);
        const uint32_t within = (uint32_t)(b % (kTransposeSuper * kTransposeSuper));
        const uint32_t xcd = within & 7u, sl = within >> 3, t = sl % gsz, g = (sl / gsz) * 8u + xcd;
        const uint64_t tile_r = (sup / sup_c) * sup_h + (g / gpr) * rg + t % rg;
        const uint64_t tile_c = (sup % sup_c) * sup_w + (g % gpr) * cg + t / rg;
        if (tile_r * kTransposeTile * RT >= m || tile_c >= tiles_c) return false;
        o.tile_r = tile_r;
        o.w0 = tile_c * kWordsPerBlock;
        o.words_here = (uint32_t)(n_words - o.w0 < kWordsPerBlock ? n_words - o.w0 : kWordsPerBlock);
        return true;
    };
    auto next_valid = [&](uint64_t b, TrTile &o) -> uint64_t {      // first block >= b (in steps of the grid) that has a tile
        while (b < n_blocks && !tile_of(b, o)) b += gridDim.x;
        return b;
    };
    const uint32_t ct = threadIdx.x / kBlock, tid = threadIdx.x % kBlock;
    uint8_t *tile = tiles[ct];
    constexpr int kParts = 4 * RT, kLoads = kTransposeTile * kParts / kBlock;
    u64x2 ld[kLoads];
    auto load_tile = [&](const TrTile &tt) {
        const uint64_t byte0 = tt.tile_r * (kTransposeTile / 8) * RT;
        const uint32_t cols_here = tt.words_here > 8 * ct ? min(tt.words_here - 8 * ct, 8u) * 64u : 0u;
#pragma unroll
        for (int it = 0; it < kLoads; it++) {
            const uint32_t item = it * kBlock + tid, col = item / kParts, part = item % kParts;
            const uint64_t off = byte0 + part * 16;
            const bool ok = col < cols_here && off + 16 <= bstride && off < nb;
            const u64x2 *src = reinterpret_cast<const u64x2 *>(blooms + ((tt.w0 + 8 * ct) * 64 + (ok ? col : 0)) * bstride + (ok ? off : 0));
            ld[it] = ok ? __builtin_nontemporal_load(src) : u64x2{0ull, 0ull};
        }
    };
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t mycol = bit_of_col(lane);
    TrTile cur, nxt;
    uint64_t b = next_valid(blockIdx.x, cur);
    if (b >= n_blocks) return;
    load_tile(cur);
    for (;;) {
        const uint64_t b_next = next_valid(b + gridDim.x, nxt);
        const bool have_next = b_next < n_blocks;
#pragma unroll
        for (int half = 0; half < RT; half++) {
            const uint64_t r0 = (cur.tile_r * RT + half) * kTransposeTile;
            if (half) __syncthreads();
#pragma unroll
            for (int it = 0; it < kLoads; it++) {
                const uint32_t item = it * kBlock + tid, col = item / kParts, part = item % kParts;
                if ((int)(part >> 2) != half) continue;
                uint64_t *d = reinterpret_cast<uint64_t *>(tile + col * kPitch + (part & 3u) * 16);
                d[0] = ld[it].x;
                d[1] = ld[it].y;
            }
            __syncthreads();
            if (half == RT - 1 && have_next && !(VAR & 1)) load_tile(nxt);      // (VAR & 1: persistent but not pipelined)
            if (!(VAR & 2))
            for (uint32_t pi = wave; pi < 36; pi += kBlock / 64) {
                uint32_t y = 0;
                while ((y + 1) * (y + 2) / 2 <= pi) y++;
                const uint32_t x = pi - y * (y + 1) / 2;
                uint8_t *pa = tile + (x * 64) * kPitch + y * 8;
                uint8_t *pb = tile + (y * 64) * kPitch + x * 8;
                uint64_t va = *reinterpret_cast<const uint64_t *>(pa + mycol * kPitch);
                uint64_t vb = x != y ? *reinterpret_cast<const uint64_t *>(pb + mycol * kPitch) : 0ull;
                va = transpose64_lanes(by_column(va), lane);
                if (x != y) vb = transpose64_lanes(by_column(vb), lane);
                *reinterpret_cast<uint64_t *>(pb + lane * kPitch) = va;
                if (x != y) *reinterpret_cast<uint64_t *>(pa + lane * kPitch) = vb;
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < kTransposeTile * 4 / kBlock; it++) {
                const uint32_t item = it * (kBlock * CT) + threadIdx.x, row = item / (4 * CT), part = item % (4 * CT);
                const uint64_t r = r0 + row;
                if (r >= m || part * 2 >= cur.words_here) continue;
                const uint64_t *sp = reinterpret_cast<const uint64_t *>(tiles[part >> 2] + row * kPitch + (part & 3u) * 16);
                uint64_t *dst = index + r * stride_words + w_first + cur.w0 + part * 2;
                if (part * 2 + 1 < cur.words_here) __builtin_nontemporal_store(u64x2{sp[0], sp[1]}, reinterpret_cast<u64x2 *>(dst));
                else dst[0] = sp[0];
            }
        }
        if (!have_next) break;
        __syncthreads();                                   // phase 3 has read the buffers
        if (VAR & 1) load_tile(nxt);
        cur = nxt;
        b = b_next;
    }
}

}   // namespace bigsi

__global__ void k_fill(uint64_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull;
}

int main(int argc, char **argv)
{
    using namespace bigsi;
    const uint64_t m = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t n_cols = argc > 2 ? strtoull(argv[2], nullptr, 10) / 128 * 128 : 8192ull;
    const uint64_t align = argc > 3 ? strtoull(argv[3], nullptr, 10) : 128ull;      // filter pitch rounded up to this (16: the packed pitch rounds 3-5 measured at)
    const uint64_t nb = (m + 7) / 8, bstride = (nb + align - 1) / align * align, n_words = n_cols / 64, stride_words = (n_words + 15) / 16 * 16;
    uint8_t *blooms = nullptr;
    uint64_t *index = nullptr;
    CK(hipMalloc(&blooms, bstride * n_cols + 4096));
    CK(hipMalloc(&index, stride_words * 8 * m + 4096));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint64_t *>(blooms), bstride * n_cols / 8);
    CK(hipMemset(index, 0, stride_words * 8 * m));
    CK(hipDeviceSynchronize());
    const uint64_t tiles_c = (n_words + 15) / 16;
    uint32_t sup_w = kTransposeSuper;
    while (sup_w > 1 && sup_w / 2 >= tiles_c) sup_w /= 2;
    const uint32_t sup_h = (uint32_t)(kTransposeSuper * kTransposeSuper) / sup_w;
    const uint64_t blocks = ((m + 1023) / 1024 + sup_h - 1) / sup_h * ((tiles_c + sup_w - 1) / sup_w) * (uint64_t)(kTransposeSuper * kTransposeSuper);
    printf("matrix %llu rows x %llu columns (filter pitch %llu, row pitch %llu): %llu workgroups; GB/s in + out\n", (unsigned long long)m,
           (unsigned long long)n_cols, (unsigned long long)bstride, (unsigned long long)stride_words * 8, (unsigned long long)blocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](auto launch, const char *name) {
        std::vector<float> ms;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("  %-78s %7.0f\n", name, 2.0 * nb * n_cols / (ms[1] * 1e-3) / 1e9);
        fflush(stdout);
    };
#define ARGS(rg, cg) dim3((unsigned)blocks), dim3(512), 0, 0, index, stride_words, m, 0ull, n_words, blooms, bstride, nb, (uint32_t)(rg), std::min<uint32_t>((cg), sup_w), sup_w
// (the library's own kernel takes the number of filters as well: k_transpose_regs' argument, unused by k_transpose_tiles)
#define ARGS_LIB(rg, cg) dim3((unsigned)blocks), dim3(512), 0, 0, index, stride_words, m, 0ull, n_words, blooms, n_words * 64, bstride, nb, (uint32_t)(rg), std::min<uint32_t>((cg), sup_w), sup_w
    run([&] { hipLaunchKernelGGL((k_transpose_tiles<2, 2>), ARGS_LIB(4, 1)); }, "k_transpose_tiles<2,2> as shipped (XCD groups 4 x 1)");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 0>), ARGS(4, 1)); }, "the copy, no switch");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 1>), ARGS(4, 1)); }, "1: tiles row-major inside the supertile");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 2>), ARGS(4, 1)); }, "2: no phase 2");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 3>), ARGS(4, 1)); }, "1 + 2");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 4>), ARGS(4, 1)); }, "4: 16-byte LDS accesses in phases 1 and 3");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 6>), ARGS(4, 1)); }, "2 + 4");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 7>), ARGS(4, 1)); }, "1 + 2 + 4 (the skeleton)");
    run([&] { hipLaunchKernelGGL((k_tr_var<2, 2, 5>), ARGS(4, 1)); }, "1 + 4");
#define PARGS(grid) dim3((unsigned)std::min<uint64_t>(blocks, (grid))), dim3(512), 0, 0, index, stride_words, m, 0ull, n_words, blooms, bstride, nb, 4u, std::min<uint32_t>(1u, sup_w), sup_w, blocks
    run([&] { hipLaunchKernelGGL((k_tr_pipe<0>), PARGS(512)); }, "persistent + pipelined, 512 workgroups");
    run([&] { hipLaunchKernelGGL((k_tr_pipe<0>), PARGS(1024)); }, "persistent + pipelined, 1024 workgroups");
    run([&] { hipLaunchKernelGGL((k_tr_pipe<0>), PARGS(2048)); }, "persistent + pipelined, 2048 workgroups");
    run([&] { hipLaunchKernelGGL((k_tr_pipe<0>), PARGS(8192)); }, "persistent + pipelined, 8192 workgroups");
    run([&] { hipLaunchKernelGGL((k_tr_pipe<1>), PARGS(512)); }, "persistent, not pipelined, 512 workgroups");
    run([&] { hipLaunchKernelGGL((k_tr_pipe<2>), PARGS(512)); }, "persistent + pipelined, no phase 2, 512 workgroups");
    // did the pipelined kernel transpose?  (compare with the shipped kernel's output)
    {
        const uint64_t words = stride_words * m;
        uint64_t *ref = nullptr;
        CK(hipMalloc(&ref, words * 8));
        hipLaunchKernelGGL((k_transpose_tiles<2, 2>), ARGS_LIB(4, 1));
        CK(hipMemcpy(ref, index, words * 8, hipMemcpyDeviceToDevice));
        CK(hipMemset(index, 0, words * 8));
        hipLaunchKernelGGL((k_tr_pipe<0>), PARGS(512));
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> a(1 << 20), c(1 << 20);
        bool same = true;
        for (uint64_t off = 0; off < words && same; off += words / 37 + 1) {
            const uint64_t n = std::min<uint64_t>(1 << 20, words - off);
            CK(hipMemcpy(a.data(), ref + off, n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(c.data(), index + off, n * 8, hipMemcpyDeviceToHost));
            same = std::equal(a.begin(), a.begin() + n, c.begin());
        }
        printf("  pipelined output %s the shipped kernel's (37 windows of 8 MB)\n", same ? "equals" : "DIFFERS FROM");
    }
    return 0;
}
