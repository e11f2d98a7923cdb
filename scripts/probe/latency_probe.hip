// latency_probe.hip -- what one small call costs on this box, primitive by primitive (measurement tool, not part of the product;
// built by scripts/probe/build.sh, run on the GPU box).  Each line: median / p10 / p90 microseconds of 2000 host-timed repetitions.
//   A  empty kernel, hipStreamSynchronize
//   B  empty kernel, event record + hipEventSynchronize
//   C  kernel that writes a flag into coherent pinned memory, host spins on the flag (no event, no synchronize)
//   D  H2D copy of 1 KB from pinned memory (hipMemcpyAsync) + kernel reading it + flag
//   E  kernel reading the same 1 KB straight from pinned memory (zero copy) + flag
//   F  two dependent kernels + flag          G  three dependent kernels + flag
//   H  hipGraphLaunch of [copy 1 KB, kernel, kernel] + flag
//   I  D with a 64 KB input                  J  E with a 64 KB input (64 workgroups reading 1 KB each)
//   K  kernel busy for ~10 us (dependent loads) + flag: the floor of a call whose device work is 10 us
//   M  E with the input passed BY VALUE in the kernel arguments (1 KB / 3.9 KB of kernarg) instead of through pinned memory
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ void k_flag(volatile uint64_t *flag, uint64_t v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; }
}
// sum `n` bytes of `src` (device or pinned host memory) into out[block]; the LAST workgroup to finish raises the flag
__global__ void k_read_flag(const uint8_t *src, uint32_t n_per_block, uint32_t *out, uint32_t *counter, volatile uint64_t *flag, uint64_t v)
{
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < n_per_block; i += blockDim.x) acc += src[(size_t)blockIdx.x * n_per_block + i];
    atomicAdd(&s, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = s;
        __threadfence_system();
        if (atomicAdd(counter, 1u) == gridDim.x - 1) { *counter = 0; __threadfence_system(); *flag = v; }
    }
}
template <int N>
struct Blob { uint8_t b[N]; };
template <int N>
__global__ void k_arg_flag(const Blob<N> in, uint32_t n, uint32_t *out, volatile uint64_t *flag, uint64_t v)
{
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc += in.b[i];
    atomicAdd(&s, acc);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = s; __threadfence_system(); *flag = v; }
}
__global__ void k_chase(const uint32_t *next, uint32_t steps, uint32_t *out, volatile uint64_t *flag, uint64_t v)
{
    uint32_t p = 0;
    for (uint32_t i = 0; i < steps; i++) p = next[p];
    *out = p;
    __threadfence_system();
    if (flag) *flag = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void spin(volatile uint64_t *flag, uint64_t v) { while (*flag != v) _mm_pause(); }
static void report(const char *name, std::vector<double> &t)
{
    std::sort(t.begin(), t.end());
    printf("%-78s median %6.1f  p10 %6.1f  p90 %6.1f us\n", name, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10]);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint64_t *flag;
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *flag = 0;
    uint8_t *h_in, *d_in;
    CK(hipHostMalloc((void **)&h_in, 64 << 10, hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < (64 << 10); i++) h_in[i] = (uint8_t)i;
    CK(hipMalloc((void **)&d_in, 64 << 10));
    uint32_t *d_out, *d_counter, *d_next;
    CK(hipMalloc((void **)&d_out, 4096));
    CK(hipMalloc((void **)&d_counter, 4));
    CK(hipMemset(d_counter, 0, 4));
    const uint32_t n_next = 1 << 20;
    {
        std::vector<uint32_t> nx(n_next);
        for (uint32_t i = 0; i < n_next; i++) nx[i] = (uint32_t)(((uint64_t)i * 2654435761u + 12345u) % n_next);
        CK(hipMalloc((void **)&d_next, n_next * 4));
        CK(hipMemcpy(d_next, nx.data(), n_next * 4, hipMemcpyHostToDevice));
    }
    const int reps = 2000, warm = 50;
    uint64_t serial = 0;
    std::vector<double> t;
    auto run = [&](const char *name, auto &&body) {
        t.clear();
        for (int i = 0; i < reps + warm; i++) {
            const double t0 = now_us();
            body();
            if (i >= warm) t.push_back(now_us() - t0);
        }
        report(name, t);
    };
    run("A empty kernel + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); CK(hipStreamSynchronize(st)); });
    run("B empty kernel + event record + hipEventSynchronize", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); CK(hipEventRecord(ev, st)); CK(hipEventSynchronize(ev)); });
    run("C kernel writes a flag into coherent pinned memory, host spins", [&] { ++serial; hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, serial); spin(flag, serial); });
    run("D hipMemcpyAsync 1 KB pinned->device + kernel reading it + flag", [&] {
        ++serial;
        CK(hipMemcpyAsync(d_in, h_in, 1024, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_read_flag, dim3(1), dim3(256), 0, st, d_in, 1024u, d_out, d_counter, flag, serial);
        spin(flag, serial);
    });
    run("E kernel reads the 1 KB straight from pinned memory (zero copy) + flag", [&] {
        ++serial;
        hipLaunchKernelGGL(k_read_flag, dim3(1), dim3(256), 0, st, h_in, 1024u, d_out, d_counter, flag, serial);
        spin(flag, serial);
    });
    run("F two dependent kernels + flag", [&] {
        ++serial;
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, serial);
        spin(flag, serial);
    });
    run("G three dependent kernels + flag", [&] {
        ++serial;
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, serial);
        spin(flag, serial);
    });
    {   // H: a captured graph [copy, kernel, kernel]; the flag value is fixed at capture, so the kernel increments it instead
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        CK(hipMemcpyAsync(d_in, h_in, 1024, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_read_flag, dim3(1), dim3(256), 0, st, d_in, 1024u, d_out, d_counter, (volatile uint64_t *)(flag + 1), 1ull);
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, d_next, 1u, d_out, (volatile uint64_t *)nullptr, 0ull);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        run("H hipGraphLaunch of [copy 1 KB, kernel, kernel] + hipStreamSynchronize", [&] { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); });
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    run("I hipMemcpyAsync 64 KB pinned->device + 64 workgroups reading it + flag", [&] {
        ++serial;
        CK(hipMemcpyAsync(d_in, h_in, 64 << 10, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_read_flag, dim3(64), dim3(256), 0, st, d_in, 1024u, d_out, d_counter, flag, serial);
        spin(flag, serial);
    });
    run("J 64 workgroups read 64 KB straight from pinned memory (zero copy) + flag", [&] {
        ++serial;
        hipLaunchKernelGGL(k_read_flag, dim3(64), dim3(256), 0, st, h_in, 1024u, d_out, d_counter, flag, serial);
        spin(flag, serial);
    });
    for (uint32_t steps : {8u, 16u, 32u}) {
        char name[128];
        snprintf(name, sizeof name, "K kernel of %u dependent HBM loads + flag", steps);
        run(name, [&] { ++serial; hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, d_next, steps, d_out, flag, serial); spin(flag, serial); });
    }
    {
        static Blob<1024> b1;
        static Blob<3968> b4;
        for (int i = 0; i < 1024; i++) b1.b[i] = (uint8_t)i;
        for (int i = 0; i < 3968; i++) b4.b[i] = (uint8_t)i;
        run("M 1 KB passed by value in the kernel arguments + flag", [&] { ++serial; hipLaunchKernelGGL(k_arg_flag<1024>, dim3(1), dim3(256), 0, st, b1, 1024u, d_out, flag, serial); spin(flag, serial); });
        uint32_t got = 0;
        CK(hipMemcpy(&got, d_out, 4, hipMemcpyDeviceToHost));
        uint32_t want = 0;
        for (int i = 0; i < 1024; i++) want += (uint8_t)i;
        if (got != want) { fprintf(stderr, "M: sum %u != %u\n", got, want); return 1; }
        run("M 3.9 KB passed by value in the kernel arguments + flag", [&] { ++serial; hipLaunchKernelGGL(k_arg_flag<3968>, dim3(1), dim3(256), 0, st, b4, 3968u, d_out, flag, serial); spin(flag, serial); });
    }
    run("L K(16) + event record + hipEventSynchronize instead of the flag", [&] {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, d_next, 16u, d_out, (volatile uint64_t *)nullptr, 0ull);
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
    });
    return 0;
}
