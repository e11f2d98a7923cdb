// tr_probe.hip -- what do gfx950's LDS transpose reads return?  (measurement tool under scripts/, not part of the product)
//
// ds_read_b64_tr_b8 / ds_read_b64_tr_b4 deliver 64 bits per lane gathered from the 8-byte words that the lanes of a 16-lane
// group address.  The build's transpose (k_transpose_tiles) uses the 8-bit form to move bytes between lanes on their way out of
// the LDS; this probe prints, for every lane and every result nibble, WHICH lane's word and which nibble of it the result came
// from, for a few address patterns -- the layout is read off the hardware instead of assumed.
//     tr_probe            prints the source map of both forms
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i lds_v2i;

// pass k: nibble at nibble address n of the LDS holds (n >> 4k) & 15; three passes give 12 bits of source nibble address
template <int FORM>
__global__ void k_tr(const uint32_t *__restrict__ lane_addr /* byte address per lane */, uint32_t pass, uint64_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (uint32_t b = threadIdx.x; b < 8192; b += blockDim.x) {
        const uint32_t n0 = 2 * b, n1 = 2 * b + 1;
        lds[b] = (uint8_t)(((n0 >> (4 * pass)) & 15u) | (((n1 >> (4 * pass)) & 15u) << 4));
    }
    __syncthreads();
    lds_v2i *p = (lds_v2i *)(lds + lane_addr[threadIdx.x]);
    v2i r;
    if (FORM == 8) r = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p);
    else r = __builtin_amdgcn_ds_read_tr4_b64_v2i32(p);
    out[threadIdx.x] = ((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x;
}

template <int FORM> static void run(const char *name, const std::vector<uint32_t> &addr)
{
    uint32_t *d_addr;
    uint64_t *d_out;
    CK(hipMalloc(&d_addr, 64 * 4));
    CK(hipMalloc(&d_out, 64 * 8));
    CK(hipMemcpy(d_addr, addr.data(), 64 * 4, hipMemcpyHostToDevice));
    uint64_t res[3][64];
    for (uint32_t pass = 0; pass < 3; pass++) {
        hipLaunchKernelGGL(k_tr<FORM>, dim3(1), dim3(64), 0, 0, d_addr, pass, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(res[pass], d_out, 64 * 8, hipMemcpyDeviceToHost));
    }
    printf("== ds_read_b64_tr_b%d, %s: result nibble -> (source lane, nibble of that lane's word); '?' = not inside any lane's word\n", FORM, name);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d (addr %5u):", l, addr[l]);
        for (int nb = 0; nb < 16; nb++) {
            uint32_t n = 0;
            for (int pass = 0; pass < 3; pass++) n |= (uint32_t)((res[pass][l] >> (4 * nb)) & 15u) << (4 * pass);
            int src = -1;
            for (int s = 0; s < 64; s++)
                if (n >= 2 * addr[s] && n < 2 * addr[s] + 16) { src = s; break; }
            if (src < 0) printf(" ?%u", n);
            else printf(" %2d.%x", src, n - 2 * addr[src]);
            if (nb % 2 == 1) printf(" ");
        }
        printf("\n");
        if (l == 15) printf("   (lanes 16-63: same pattern per 16-lane group? shown in full)\n");
    }
    CK(hipFree(d_addr));
    CK(hipFree(d_out));
}

int main()
{
    std::vector<uint32_t> ident(64), strided(64), rev(64);
    for (int l = 0; l < 64; l++) {
        ident[l] = 8 * l;
        strided[l] = 72 * l;
        rev[l] = 8 * (63 - l);
    }
    run<8>("addr = 8 * lane", ident);
    run<8>("addr = 72 * lane", strided);
    run<8>("addr = 8 * (63 - lane)", rev);
    run<4>("addr = 8 * lane", ident);
    run<4>("addr = 72 * lane", strided);
    return 0;
}
