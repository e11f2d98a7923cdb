// row_probe.hip -- what does the MEMORY SYSTEM give the row-AND access pattern, with no BIGSI code in the way?
// (measurement tool under scripts/, not part of the product; built by scripts/probe/build.sh, run on the GPU box)
//
// One buffer of <gb> GB is treated as a matrix of rows of <row_bytes> bytes (pitch rounded up to 128).  A "query" is a list of
// <rows_per_query> row ids; one wavefront streams one 1 KiB column segment of every row of a query (lane = 16 bytes,
// non-temporal loads, 8 rows in flight per lane, AND-reduce, one 16-byte store at the end) -- the load pattern of k_and_exact.
// Launches of <wgs_per_launch> workgroups of 4 wavefronts, as the library sizes them.  Row ids:
//   seq      query q takes rows_per_query CONSECUTIVE rows (a different range per query): pure streaming
//   random   uniform over the matrix, in draw order
//   sorted   the same ids sorted per query (what K1e's bucket sort gives k_and_exact)
//   banded   random ids drawn from a window of <band_mb> MB that advances with the launch (locality without order)
// Allocation: hipMalloc, --vmm: hipMemCreate + hipMemMap in chunks of the recommended granularity (page-table fragment size), or
// --contig: hipExtMallocWithFlags(hipDeviceMallocContiguous).
// Prints GB/s = bytes of all rows streamed / median launch time (hipEvents around each launch).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_stream_rows(const uint8_t *__restrict__ base, uint64_t pitch, const uint64_t *__restrict__ rows,
                                                     uint32_t rows_per_query, uint32_t segs, uint32_t q0, uint32_t n_queries, u64x2 *__restrict__ out)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t q = q0 + wave / segs, seg = wave % segs;
    if (q >= n_queries) return;
    const uint64_t *r = rows + (uint64_t)q * rows_per_query;
    const uint64_t off = (uint64_t)seg * 1024 + lane * 16;
    if (off + 16 > pitch) return;
    u64x2 acc = {~0ull, ~0ull};
    uint32_t i = 0;
    for (; i + 8 <= rows_per_query; i += 8) {
        u64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(base + r[i + j] * pitch + off));
#pragma unroll
        for (int j = 0; j < 8; j++) acc &= v[j];
    }
    for (; i < rows_per_query; i++) acc &= __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(base + r[i] * pitch + off));
    out[(uint64_t)q * segs * 64 + seg * 64 + lane] = acc;
}

// sliced (round 6): the traversal a counting kernel with an address-ordered row list would need -- a WORKGROUP owns one 128-byte column
// slice of a query (the per-k-mer partial ANDs of a slice, 128 bytes each, would fit its LDS: 970 x 128 B = 124 KB, which is what limits
// the CU to one workgroup: the dynamic LDS below stands for it) and its 16 wavefronts walk the query's SORTED rows, 8 rows per wave
// instruction (8 lanes x 16 bytes = one line per row), 8 instructions in flight.  The slices of a query are neighbouring workgroups.
__global__ __launch_bounds__(1024) void k_stream_slices(const uint8_t *__restrict__ base, uint64_t pitch, const uint64_t *__restrict__ rows,
                                                        uint32_t rows_per_query, uint32_t slices, uint32_t q0, uint32_t n_queries, u64x2 *__restrict__ out)
{
    extern __shared__ uint8_t state[];
    const uint32_t q = q0 + blockIdx.x / slices, sl = blockIdx.x % slices;
    if (q >= n_queries) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> 3, part = lane & 7;
    const uint64_t *r = rows + (uint64_t)q * rows_per_query;
    const uint64_t off = (uint64_t)sl * 128 + part * 16;
    u64x2 acc = {~0ull, ~0ull};
    for (uint32_t i = wave * 64; i < rows_per_query; i += 16 * 64) {      // a wavefront takes 64 consecutive rows of every 1024
        u64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t x = i + j * 8 + sub;
            v[j] = x < rows_per_query ? __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(base + r[x] * pitch + off)) : u64x2{~0ull, ~0ull};
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc &= v[j];
    }
    if (threadIdx.x == 0) state[0] = (uint8_t)acc.x;
    out[(uint64_t)blockIdx.x * 1024 + threadIdx.x] = acc;
}

__global__ void k_fill(uint64_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull;
}

int main(int argc, char **argv)
{
    double gb = 16;
    uint64_t row_bytes = 12500, rows_per_query = 3880, band_mb = 4096;
    uint32_t n_queries = 1024, wgs = 512;
    bool vmm = false, contig = false;
    uint32_t hk = 4;                                   // kfirst: rows per k-mer
    std::string modes = "seq,random,sorted,banded";
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() { return std::string(argv[++i]); };
        if (a == "--gb") gb = atof(val().c_str());
        else if (a == "--row-bytes") row_bytes = strtoull(val().c_str(), nullptr, 10);
        else if (a == "--rows-per-query") rows_per_query = strtoull(val().c_str(), nullptr, 10);
        else if (a == "--queries") n_queries = atoi(val().c_str());
        else if (a == "--wgs") wgs = atoi(val().c_str());
        else if (a == "--band-mb") band_mb = strtoull(val().c_str(), nullptr, 10);
        else if (a == "--h") hk = atoi(val().c_str());
        else if (a == "--modes") modes = val();
        else if (a == "--vmm") vmm = true;
        else if (a == "--contig") contig = true;
    }
    const uint64_t pitch = (row_bytes + 127) / 128 * 128, n_rows = (uint64_t)(gb * 1e9) / pitch, bytes = n_rows * pitch;
    const uint32_t segs = (uint32_t)((pitch + 1023) / 1024);
    uint8_t *d = nullptr;
    size_t gran_min = 0, gran_rec = 0;
    if (vmm) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
        const size_t chunk = std::max<size_t>(gran_rec, (size_t)1 << 30), total = (bytes + chunk - 1) / chunk * chunk;
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, total, chunk, nullptr, 0));
        for (size_t o = 0; o < total; o += chunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((uint8_t *)va + o, chunk, 0, h, 0));
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        d = (uint8_t *)va;
    } else if (contig) {
        CK(hipExtMallocWithFlags((void **)&d, bytes, hipDeviceMallocContiguous));      // physically contiguous: the largest page-table fragments
    } else {
        CK(hipMalloc((void **)&d, bytes));
    }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)d, bytes / 8);
    CK(hipDeviceSynchronize());
    printf("matrix %.1f GB: %llu rows x %llu B (pitch %llu, %u segments); %u queries x %llu rows; launches of %u workgroups; alloc %s",
           bytes / 1e9, (unsigned long long)n_rows, (unsigned long long)row_bytes, (unsigned long long)pitch, segs, n_queries,
           (unsigned long long)rows_per_query, wgs, vmm ? "vmm" : contig ? "hipExtMallocWithFlags(contiguous)" : "hipMalloc");
    if (vmm) printf(" (granularity min %zu rec %zu)", gran_min, gran_rec);
    printf("\n");
    uint64_t *d_rows;
    u64x2 *d_out;
    CK(hipMalloc((void **)&d_rows, (size_t)n_queries * rows_per_query * 8));
    CK(hipMalloc((void **)&d_out, (size_t)n_queries * std::max<size_t>(segs * 64, (pitch / 128) * 1024) * 16));
    CK(hipFuncSetAttribute((const void *)k_stream_slices, hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024));
    const uint32_t q_per_launch = std::max<uint32_t>(wgs * 4 / segs, 1);
    std::mt19937_64 rng(1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // kfirst (round 6): the counting kernel's constraint -- the h rows of a k-mer are fetched together -- with the k-mers of a query
    // ordered by the address of their FIRST row: 1/h of the fetches sweep the matrix in step across the co-resident queries, the
    // other (h - 1)/h stay uniform.  kpage: the same with the first rows only bucketed (256 MB buckets), draw order inside a bucket
    for (const char *mode : {"seq", "random", "sorted", "banded", "kfirst", "kpage", "sliced"}) {
        if (("," + modes + ",").find(std::string(",") + mode + ",") == std::string::npos) continue;
        std::vector<uint64_t> ids((size_t)n_queries * rows_per_query);
        for (uint32_t q = 0; q < n_queries; q++) {
            uint64_t *r = ids.data() + (size_t)q * rows_per_query;
            if (!strcmp(mode, "seq")) {
                const uint64_t start = rng() % (n_rows - rows_per_query);
                for (uint64_t i = 0; i < rows_per_query; i++) r[i] = start + i;
            } else if (!strcmp(mode, "banded")) {
                const uint64_t band_rows = std::min<uint64_t>(std::max<uint64_t>(band_mb * (1ull << 20) / pitch, rows_per_query), n_rows);
                const uint64_t start = (uint64_t)((double)(q / q_per_launch) * q_per_launch / n_queries * (n_rows - band_rows));
                for (uint64_t i = 0; i < rows_per_query; i++) r[i] = start + rng() % band_rows;
            } else {
                for (uint64_t i = 0; i < rows_per_query; i++) r[i] = rng() % n_rows;
                if (!strcmp(mode, "sorted") || !strcmp(mode, "sliced")) std::sort(r, r + rows_per_query);
                if (!strcmp(mode, "kfirst") || !strcmp(mode, "kpage")) {
                    const uint64_t nk = rows_per_query / hk, bucket_rows = !strcmp(mode, "kpage") ? std::max<uint64_t>((256ull << 20) / pitch, 1) : 1;
                    std::vector<uint64_t> order(nk), tmp(r, r + nk * hk);
                    for (uint64_t i = 0; i < nk; i++) order[i] = i;
                    std::stable_sort(order.begin(), order.end(), [&](uint64_t a_, uint64_t b_) { return tmp[a_ * hk] / bucket_rows < tmp[b_ * hk] / bucket_rows; });
                    for (uint64_t i = 0; i < nk; i++)
                        for (uint32_t s_ = 0; s_ < hk; s_++) r[i * hk + s_] = tmp[order[i] * hk + s_];
                }
            }
        }
        CK(hipMemcpy(d_rows, ids.data(), ids.size() * 8, hipMemcpyHostToDevice));
        std::vector<float> ms;
        for (int rep = 0; rep < 3; rep++)
            for (uint32_t q0 = 0; q0 < n_queries; q0 += q_per_launch) {
                const uint32_t nq = std::min(q_per_launch, n_queries - q0);
                const uint32_t waves = nq * segs, blocks = (waves + 3) / 4;
                CK(hipEventRecord(e0, 0));
                if (!strcmp(mode, "sliced"))
                    hipLaunchKernelGGL(k_stream_slices, dim3(nq * (uint32_t)(pitch / 128)), dim3(1024), 124 * 1024, 0, d, pitch, d_rows, (uint32_t)rows_per_query,
                                       (uint32_t)(pitch / 128), q0, q0 + nq, d_out);
                else
                hipLaunchKernelGGL(k_stream_rows, dim3(blocks), dim3(256), 0, 0, d, pitch, d_rows, (uint32_t)rows_per_query, segs, q0, q0 + nq, d_out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (rep && nq == q_per_launch) ms.push_back(t);
            }
        std::sort(ms.begin(), ms.end());
        const double med = ms[ms.size() / 2], by = (double)q_per_launch * rows_per_query * (!strcmp(mode, "sliced") ? (double)pitch : segs * 1024.0 > pitch ? pitch : segs * 1024.0);
        printf("  %-7s median launch %.3f ms (%zu launches of %u queries)  %.0f GB/s  = %.3f of 8 TB/s\n", mode, med, ms.size(), q_per_launch, by / med / 1e6, by / med / 1e6 / 8000);
    }
    return 0;
}
