// bigsi_kernels.hpp -- CDNA4 (gfx950) device code of the BIGSI query hot path.
//
// Data layout in HBM: the index is m rows x stride_words uint64 (stride a multiple of 16 words = 128 B).
// A row holds exactly the reference's row bytes (bitarray.tobytes(), bigsi/storage/base.py:85-99), so a
// little-endian uint64 load of word w covers columns [64w, 64w+64) with column c at bit
// ((c>>3)&7)*8 + 7-(c&7).  AND / popcount / bit-sliced addition are agnostic to that permutation; it is
// only applied where set bits become colour ids (compaction, counter expansion).
//
// Kernels: K1 k_kmer_insert/resolve/rank/rows, K2/K3a k_and_exact, K2/K3b k_and_count,
// K4 k_chunk_hits_* / k_scan_chunks / k_write_hits_*, K5 k_presence, plus lookup / storage / build helpers.
// All bitwise, HBM-bound work: no MFMA.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bigsi_score.hpp"

namespace bigsi {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kBlock = 256;      // 4 wavefronts
constexpr int kVec = 2;          // uint64 words per lane per row load (16 B/lane, 1 KiB per wave instruction)

// ------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// bit position, inside the little-endian uint64 of a row, of column (8*byte + j)
__host__ __device__ __forceinline__ uint32_t bit_of_col(uint32_t c) { return ((c >> 3) & 7u) * 8u + 7u - (c & 7u); }

// column-order view of a word in row bit order: reverse the bits inside every byte
__device__ __forceinline__ uint64_t by_column(uint64_t x)
{
    const uint32_t lo = __builtin_bswap32(__builtin_bitreverse32((uint32_t)x)), hi = __builtin_bswap32(__builtin_bitreverse32((uint32_t)(x >> 32)));
    return ((uint64_t)hi << 32) | lo;
}

// mask of the valid column bits of word w for an index of n_cols columns (pad bits stay zero)
__host__ __device__ __forceinline__ uint64_t valid_mask(uint64_t w, uint64_t n_cols)
{
    uint64_t c0 = w * 64;
    if (c0 + 64 <= n_cols) return ~0ull;
    if (c0 >= n_cols) return 0ull;
    uint64_t mask = 0;
    for (int b = 0; b < 8; b++) {
        uint64_t cb = c0 + 8 * (uint64_t)b;
        uint64_t n = cb >= n_cols ? 0 : (n_cols - cb >= 8 ? 8 : n_cols - cb);
        mask |= (uint64_t)((0xFFu << (8 - n)) & 0xFFu) << (8 * b);
    }
    return mask;
}

// reverse_comp's per-base map (bigsi/utils/fncts.py:12,38-39): A<->T, C<->G, anything else unchanged
__device__ __forceinline__ uint8_t complement(uint8_t c)
{
    // branchless: 'A' ^ 'T' == 0x15 and 'C' ^ 'G' == 0x04 (a chain of ?: compiles to divergent exec-mask branches, and this
    // runs 2k times per k-mer)
    const uint32_t x = c;
    const uint32_t at = (uint32_t)(x == 'A') | (uint32_t)(x == 'T');
    const uint32_t cg = (uint32_t)(x == 'C') | (uint32_t)(x == 'G');
    return (uint8_t)(x ^ (at * 0x15u) ^ (cg * 0x04u));
}

// byte j of the canonical form of the k-mer at s[0..k): forward strand or reverse complement
struct KmerView {
    const char *s;
    uint32_t k;
    bool rc;
    __device__ __forceinline__ uint8_t operator[](uint32_t j) const
    {
        return rc ? complement((uint8_t)s[k - 1 - j]) : (uint8_t)s[j];
    }
};

// canonical (bigsi/utils/fncts.py:51-54): lexicographic min of k-mer and reverse complement -> use rc?
__device__ __forceinline__ bool use_revcomp(const char *s, uint32_t k)
{
    for (uint32_t j = 0; j < k; j++) {
        uint8_t f = (uint8_t)s[j], r = complement((uint8_t)s[k - 1 - j]);
        if (f != r) return r < f;
    }
    return false;
}

// mmh3.hash(kmer, seed) (bigsi/bloom/bloomfilter.py:6): MurmurHash3_x86_32, Austin Appleby's public-domain
// algorithm, over the k bytes of the view.
__device__ __forceinline__ uint32_t murmur3_32(const KmerView &v, uint32_t seed)
{
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
    const uint32_t len = v.k, nblocks = len >> 2;
    for (uint32_t i = 0; i < nblocks; i++) {
        uint32_t k1 = (uint32_t)v[4 * i] | ((uint32_t)v[4 * i + 1] << 8) | ((uint32_t)v[4 * i + 2] << 16) |
                      ((uint32_t)v[4 * i + 3] << 24);
        k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
        h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5u + 0xe6546b64u;
    }
    uint32_t k1 = 0;
    const uint32_t t = nblocks * 4, rem = len & 3u;
    if (rem == 3) k1 ^= (uint32_t)v[t + 2] << 16;
    if (rem >= 2) k1 ^= (uint32_t)v[t + 1] << 8;
    if (rem >= 1) { k1 ^= (uint32_t)v[t]; k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1; }
    h1 ^= len;
    h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
    return h1;
}

// _hash (bigsi/bloom/bloomfilter.py:5-6): signed 32-bit hash, Python floor-mod by m -> row in [0, m)
__device__ __forceinline__ uint64_t row_of_hash(uint32_t h, uint64_t m)
{
    const int32_t sh = (int32_t)h;
    const uint32_t mag = sh >= 0 ? (uint32_t)sh : 0u - (uint32_t)sh;      // |hash| <= 2^31 fits 32 bits
    if (m <= 0xFFFFFFFFull) {                                              // (wave-uniform) 32-bit division: ~4x cheaper than 64-bit
        const uint32_t a = mag % (uint32_t)m;
        return sh >= 0 || a == 0 ? (uint64_t)a : m - a;
    }
    const uint64_t a = (uint64_t)mag % m;
    return sh >= 0 || a == 0 ? a : m - a;
}

// block-wide exclusive scan of one uint32 per thread (blockDim.x a multiple of 64, up to 1024); returns prefix,
// *total = block sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total, uint32_t *lds /* >= blockDim.x/64 entries */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // inclusive scan of the wavefront on the DPP path (shifts within rows of 16 lanes, then the row broadcasts): seven VALU adds instead
    // of six round trips through the LDS crossbar (__shfl_up = ds_bpermute) -- it is on the chain of every latency-bound call
    uint32_t incl = v;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);      // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);      // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);      // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);      // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    __syncthreads();   // protect lds reuse across calls
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t w = 0; w < nw; w++) {
        uint32_t x = lds[w];
        if (w < wave) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

// the same for a FLAG per thread (0 / 1): the wavefront's part is a ballot and two mbcnt instead of six cross-lane shuffles (1.0 us
// of a single query's K1 was the scan above).  No barrier after the LDS reads: the caller reuses `lds` only behind one of its own.
__device__ __forceinline__ uint32_t block_exclusive_scan_flag(bool flag, uint32_t *total, uint32_t *lds /* >= blockDim.x/64 entries */)
{
    const uint64_t bal = __ballot(flag);
    const uint32_t in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63u) == 0) lds[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nw; w++) {
        const uint32_t x = lds[w];
        base += w < wave ? x : 0u;
        tot += x;
    }
    *total = tot;
    return base + in_wave;
}

// ------------------------------------------------------------------------------ K1: k-merise + dedupe + hash
// Restates, per query sequence:
//   seq_to_kmers (bigsi/utils/fncts.py:63-65)          every window of k bytes, no validation;
//   set(kmers) (bigsi/graph/index.py:45, graph/bigsi.py:179)  unique *query strings* (a k-mer and its reverse
//                                                       complement are two), kept in first-occurrence order;
//   canonical + generate_hashes (fncts.py:51-54, bloom/bloomfilter.py:5-13, graph/index.py:62-70)
//                                                       h row ids per unique k-mer, seeds 0..h-1;
//   min_kmers = ceil(u * threshold) (graph/bigsi.py:179) in IEEE double, as Python evaluates it.
// Four launches, the three string-heavy ones with ONE THREAD PER K-MER POSITION of the whole batch (a 256 x 1 kbp batch
// is 248k positions: the chip is full, where one workgroup per query left 252 of 256 CUs with 4 waves each):
//   K1a k_kmer_insert   open-addressing table of positions per query (global scratch), keyed by string equality; the
//                       slot converges to the smallest position of its class (deterministic whatever the atomics' order)
//   K1b k_kmer_resolve  every position reads its class representative
//   K1c k_kmer_rank     one workgroup per query: ordered compaction of representatives -> unique index, u, min_kmers
//   K1d k_kmer_rows     representatives: canonical form, MurmurHash3 with seeds 0..h-1, floor-mod m -> row ids
struct PosRef {
    uint32_t q, i;
    const char *s;
};
__device__ __forceinline__ PosRef locate(uint64_t p, const uint32_t *__restrict__ pos_query, const uint64_t *__restrict__ pos_off,
                                         const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off)
{
    const uint32_t q = pos_query[p];
    return PosRef{q, (uint32_t)(p - pos_off[q]), seqs + seq_off[q]};
}

__device__ __forceinline__ uint32_t fnv1a(const char *s, uint32_t k)
{
    uint32_t h = 2166136261u;
    for (uint32_t j = 0; j < k; j++) h = (h ^ (uint8_t)s[j]) * 16777619u;
    return h ^ (h >> 15);
}

// Compile-time k (KF = 31, the reference's default k, bigsi/constants.py:13): the window is loaded once into registers
// with KF independent byte loads and everything after that (dedupe hash, canonical choice, MurmurHash3 x h) is
// straight-line register code.  KF = 0 is the generic run-time-k path.
template <int KF>
struct RegKmer {
    uint32_t f[KF > 0 ? KF : 1];      // one register per byte (a uint8_t array would live in scratch)
    __device__ __forceinline__ void load(const char *s)
    {
#pragma unroll
        for (int j = 0; j < KF; j++) f[j] = (uint8_t)s[j];
    }
    __device__ __forceinline__ uint32_t fnv() const
    {
        uint32_t h = 2166136261u;
#pragma unroll
        for (int j = 0; j < KF; j++) h = (h ^ f[j]) * 16777619u;
        return h ^ (h >> 15);
    }
    // canonical form packed into little-endian words (utils/fncts.py:51-54)
    __device__ __forceinline__ void canonical_words(uint32_t (&w)[(KF + 3) / 4 > 0 ? (KF + 3) / 4 : 1]) const
    {
        // lexicographic compare of the k-mer with its reverse complement, first difference decides; select-only code
        uint32_t rc = 0, decided = 0;
#pragma unroll
        for (int j = 0; j < KF; j++) {
            const uint32_t a = f[j], b = complement((uint8_t)f[KF - 1 - j]);
            const uint32_t take = (decided ^ 1u) & (uint32_t)(a != b);
            rc = take ? (uint32_t)(b < a) : rc;
            decided |= take;
        }
#pragma unroll
        for (int i = 0; i < (KF + 3) / 4; i++) w[i] = 0;
#pragma unroll
        for (int j = 0; j < KF; j++) {
            const uint32_t c = rc ? (uint32_t)complement((uint8_t)f[KF - 1 - j]) : f[j];
            w[j >> 2] |= (uint32_t)c << (8 * (j & 3));
        }
    }
};

// MurmurHash3_x86_32 over KF bytes already packed in words (tail bytes in the low bits of the last word)
template <int KF>
__device__ __forceinline__ uint32_t murmur3_words(const uint32_t *w, uint32_t seed)
{
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
#pragma unroll
    for (int i = 0; i < KF / 4; i++) {
        uint32_t k1 = w[i];
        k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
        h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5u + 0xe6546b64u;
    }
    if (KF & 3) {
        uint32_t k1 = w[KF / 4];
        k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint32_t)KF;
    h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
    return h1;
}

__device__ __forceinline__ bool kmer_equal(const char *a, const char *b, uint32_t k)
{
    for (uint32_t j = 0; j < k; j++)
        if (a[j] != b[j]) return false;
    return true;
}


// ---- 31-mers on packed words (k_reads_fused, k_kmerize_lds<31>).  The sequence is staged in LDS as 32-bit words (`sw`), its
// byte-wise complement (utils/fncts.py:12) likewise, 4 pad bytes in front (`cw`): a k-mer is eight words cut out with
// v_alignbyte, its reverse complement the complement string read backwards, and the lexicographic comparison of
// utils/fncts.py:51-54 an eight-word big-endian compare -- about 75 vector operations where the byte-wise form takes 450.
__device__ __forceinline__ void kmer31_words(const uint32_t *sw, uint32_t p, uint32_t (&wf)[8])
{
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = sw[(p >> 2) + i];
#pragma unroll
    for (int i = 0; i < 8; i++) wf[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], p & 3u);
    wf[7] &= 0x00ffffffu;                                   // little-endian words, the last one holds 3 bytes
}

// equal k-mers have equal fingerprints; a match is always verified on the bytes / words
__device__ __forceinline__ uint32_t kmer31_fingerprint(const uint32_t (&wf)[8])
{
    uint32_t fp = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) fp = (fp ^ wf[i]) * 0x9E3779B1u;
    return fp ^ (fp >> 15);
}

// canonical form (the smaller of k-mer and reverse complement) and MurmurHash3_x86_32's per-word mix of it, which does
// not depend on the seed: k1[0..6] full blocks, k1[7] the 3-byte tail
__device__ __forceinline__ void kmer31_canonical_premix(const uint32_t (&wf)[8], const uint32_t *cw, uint32_t p, uint32_t (&k1)[8])
{
    // x[i] = complement bytes p+27-4i .. p+30-4i as one little-endian word = bytes 4i .. 4i+3 of the reverse complement with
    // the FIRST in the top byte (the last word: 3 bytes + a zero)
    uint32_t e[9], x[8];
    const uint32_t base = ((p + 31u) >> 2) - 7u, sh = (p + 3u) & 3u;
#pragma unroll
    for (int i = 0; i < 9; i++) e[i] = cw[base + i];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = __builtin_amdgcn_alignbyte(e[8 - i], e[7 - i], sh);
    x[7] &= 0xffffff00u;
    bool rc = false, decided = false;                      // reverse complement < k-mer ?
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t f = __builtin_bswap32(wf[i]);
        const bool ne = f != x[i];
        rc = (!decided && ne) ? x[i] < f : rc;
        decided = decided || ne;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t w = rc ? __builtin_bswap32(x[i]) : wf[i];
        w *= 0xcc9e2d51u; w = rotl32(w, 15); w *= 0x1b873593u;
        k1[i] = w;
    }
}

__device__ __forceinline__ uint32_t murmur3_31_finish(const uint32_t (&k1)[8], uint32_t seed)
{
    uint32_t h1 = seed;
#pragma unroll
    for (int i = 0; i < 7; i++) { h1 ^= k1[i]; h1 = rotl32(h1, 13); h1 = h1 * 5u + 0xe6546b64u; }
    h1 ^= k1[7];                                           // the 3 tail bytes
    h1 ^= 31u;
    h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
    return h1;
}

template <int KF>
__global__ __launch_bounds__(kBlock) void k_kmer_insert(
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off,
    const uint32_t *__restrict__ pos_query, const uint64_t *__restrict__ tab_off, uint32_t *__restrict__ tab,
    uint32_t k, uint64_t total_pos, uint32_t *__restrict__ hsh)
{
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= total_pos) return;
    const PosRef r = locate(p, pos_query, pos_off, seqs, seq_off);
    uint32_t *t = tab + tab_off[r.q];
    const uint32_t mask = (uint32_t)(tab_off[r.q + 1] - tab_off[r.q]) - 1u;   // table size: power of two >= 2n
    uint32_t hv;
    if (KF > 0) {
        RegKmer<KF> km;
        km.load(r.s + r.i);
        hv = km.fnv();
    } else {
        hv = fnv1a(r.s + r.i, k);
    }
    hsh[p] = hv;
    uint32_t slot = hv & mask;
    for (;;) {
        const uint32_t cur = atomicCAS(&t[slot], kEmpty, r.i);
        if (cur == kEmpty) break;
        if (kmer_equal(r.s + cur, r.s + r.i, k)) { atomicMin(&t[slot], r.i); break; }
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(kBlock) void k_kmer_resolve(
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off,
    const uint32_t *__restrict__ pos_query, const uint64_t *__restrict__ tab_off, const uint32_t *__restrict__ tab,
    uint32_t k, uint64_t total_pos, const uint32_t *__restrict__ hsh, uint32_t *__restrict__ rep)
{
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= total_pos) return;
    const PosRef r = locate(p, pos_query, pos_off, seqs, seq_off);
    const uint32_t *t = tab + tab_off[r.q];
    const uint32_t mask = (uint32_t)(tab_off[r.q + 1] - tab_off[r.q]) - 1u;
    uint32_t slot = hsh[p] & mask;
    uint32_t c;
    for (;;) {
        c = t[slot];      // the table is final: written by the previous launch
        if (c == r.i || kmer_equal(r.s + c, r.s + r.i, k)) break;
        slot = (slot + 1) & mask;
    }
    rep[p] = c;
}

__global__ __launch_bounds__(kBlock) void k_kmer_rank(
    const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ rep,
    uint32_t k, double threshold, uint32_t *__restrict__ first_pos, uint32_t *__restrict__ uidx, uint32_t *__restrict__ pos_unique,
    uint32_t *__restrict__ num_kmers, uint32_t *__restrict__ num_unique, uint32_t *__restrict__ min_kmers)
{
    __shared__ uint32_t lds[16];
    const uint32_t q = blockIdx.x;
    const uint64_t len = seq_off[q + 1] - seq_off[q];
    const uint32_t n = len >= k ? (uint32_t)(len - k + 1) : 0u;
    const uint64_t P = pos_off[q];
    const uint32_t *rp = rep + P;
    uint32_t *fp = first_pos + P, *ux = uidx + P, *pu = pos_unique + P;
    uint32_t u = 0;
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t flag = (i < n && rp[i] == i) ? 1u : 0u;
        uint32_t tot;
        const uint32_t pre = block_exclusive_scan(flag, &tot, lds);
        if (flag) { fp[u + pre] = i; ux[i] = u + pre; }
        u += tot;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) pu[i] = ux[rp[i]];   // position -> its unique k-mer (graph/bigsi.py:233)
    if (threadIdx.x == 0) {
        num_kmers[q] = n;
        num_unique[q] = u;
        const double mk = ceil((double)u * threshold);      // one IEEE multiply, as Python's int * float
        min_kmers[q] = mk > 0.0 ? (uint32_t)mk : 0u;          // counts are >= 0, so a negative bound behaves like 0
    }
}

template <int KF>
__global__ __launch_bounds__(kBlock) void k_kmer_rows(
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off,
    const uint32_t *__restrict__ pos_query, const uint32_t *__restrict__ rep, const uint32_t *__restrict__ uidx,
    uint32_t k, uint32_t h, uint64_t m, uint64_t total_pos, uint64_t *__restrict__ rows)
{
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= total_pos) return;
    const PosRef r = locate(p, pos_query, pos_off, seqs, seq_off);
    if (rep[p] != r.i) return;        // duplicates of an earlier window contribute nothing
    const char *km = r.s + r.i;
    uint64_t *dst = rows + (pos_off[r.q] + uidx[p]) * h;
    if (KF > 0) {
        RegKmer<KF> reg;
        reg.load(km);
        uint32_t w[(KF + 3) / 4 > 0 ? (KF + 3) / 4 : 1];
        reg.canonical_words(w);
        for (uint32_t sd = 0; sd < h; sd++) dst[sd] = row_of_hash(murmur3_words<KF>(w, sd), m);
    } else {
        const KmerView v{km, k, use_revcomp(km, k)};
        for (uint32_t sd = 0; sd < h; sd++) dst[sd] = row_of_hash(murmur3_32(v, sd), m);
    }
}

#ifdef BIGSI_HIP_TUNING
// tuning builds only: per-workgroup timestamps of the phases of k_reads_fused / k_kmerize_lds (100 MHz wall clock), read by bigsi_hip_debug_phases
__device__ uint64_t g_phase[1024 * 8];
#define BIGSI_PHASE(i) do { if (threadIdx.x == 0) g_phase[(blockIdx.x & 1023u) * 8 + (i)] = wall_clock64(); } while (0)
#define BIGSI_PHASE_AT(group, i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_phase[(group) * 8 + (i)] = wall_clock64(); } while (0)
#define BIGSI_PHASE_IF(cond, group, i) do { if (threadIdx.x == 0 && (cond)) g_phase[(group) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define BIGSI_PHASE(i) do { } while (0)
#define BIGSI_PHASE_AT(group, i) do { } while (0)
#define BIGSI_PHASE_IF(cond, group, i) do { } while (0)
#endif
// K1 fused: ONE launch for batches whose longest query has at most kLdsMaxPos k-mer positions (a 4 kbp query; reads
// and gene-length queries).  One workgroup per query; the sequence and the dedupe table live in LDS (ds_cmpst / ds_min
// instead of L2 atomics), and the workgroup goes insert -> resolve -> ordered compaction -> hash without leaving the CU.
// Same results as the four-kernel path above, which remains the route for longer queries.
constexpr uint32_t kLdsMaxPos = 4096;

// The bytes of ONE query passed BY VALUE in the kernel arguments (a one-call search of a single sequence).  The runtime writes a
// launch's arguments into device-visible memory together with the packet, so the kernel's first loads are local instead of a round
// trip over the host link: scripts/probe/latency_probe.hip, 1 KB read by one workgroup + flag: 15.1 us from pinned memory (E),
// 9.9 us from the arguments (M); every KB of arguments costs the launch about 1 us, hence two sizes.
template <int N>
struct SeqArg {
    uint32_t w[N / 4];
};
template <>
struct SeqArg<0> {
};
constexpr uint32_t kSeqArgSmall = 1024, kSeqArgLarge = 3072, kSeqArgRead = 96;

template <int KF>
__device__ __forceinline__ uint32_t dedupe_hash(const char *km, uint32_t k)
{
    if (KF > 0) {
        RegKmer<KF> r;
        r.load(km);
        return r.fnv();
    }
    return fnv1a(km, k);
}

// the h rows of the k-mer at position i of a query staged in LDS (sq: its bytes; sc: KF == 31, their complements behind 4 pad bytes)
template <int KF>
// (seeds [sd0, sd1) only: dst[sd] for those)
__device__ __forceinline__ void kmer_rows(const char *sq, const char *sc, uint32_t i, uint32_t k, uint32_t sd0, uint32_t sd1, uint64_t m, uint64_t *dst)
{
    if (KF == 31) {
        uint32_t wf[8], k1[8];
        kmer31_words(reinterpret_cast<const uint32_t *>(sq), i, wf);
        kmer31_canonical_premix(wf, reinterpret_cast<const uint32_t *>(sc), i, k1);
        for (uint32_t sd = sd0; sd < sd1; sd++) dst[sd] = row_of_hash(murmur3_31_finish(k1, sd), m);
    } else if (KF > 0) {
        RegKmer<KF> reg;
        reg.load(sq + i);
        uint32_t w[(KF + 3) / 4 > 0 ? (KF + 3) / 4 : 1];
        reg.canonical_words(w);
        for (uint32_t sd = sd0; sd < sd1; sd++) dst[sd] = row_of_hash(murmur3_words<KF>(w, sd), m);
    } else {
        const KmerView v{sq + i, k, use_revcomp(sq + i, k)};
        for (uint32_t sd = sd0; sd < sd1; sd++) dst[sd] = row_of_hash(murmur3_32(v, sd), m);
    }
}

template <int KF, int ARGB = 0 /* > 0: the one sequence of the batch is `sarg` (ARGB bytes of kernel arguments), not seqs */>
__global__ __launch_bounds__(1024) void k_kmerize_lds(
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off,
    uint32_t k, uint32_t h, uint64_t m, double threshold, uint32_t tab_cap, uint32_t tab_mult, uint32_t hs_cap, uint32_t sq_bytes /* multiple of 16 */, uint32_t *__restrict__ first_pos,
    uint32_t *__restrict__ uidx, uint32_t *__restrict__ pos_unique, uint32_t *__restrict__ rep_out, uint64_t *__restrict__ rows,
    uint32_t *__restrict__ num_kmers, uint32_t *__restrict__ num_unique, uint32_t *__restrict__ min_kmers,
    uint64_t *__restrict__ rows_sorted /* non-null: also emit the query's row list in address order (what k_sort_rows does) */,
    uint64_t *__restrict__ preset /* non-null: the query's `preset_words` result words are set to preset_value here -- what the sliced
                                     (latency-bound) row-AND launches combine into with atomics; saves a memset launch per call */,
    uint64_t preset_words, uint64_t preset_value,
    uint64_t *__restrict__ pos_off_out /* non-null: seqs / seq_off / pos_off are read straight from pinned host memory (a one-call
                                          search: no upload); the device copy of pos_off the later kernels read is written here */,
    uint32_t one_len /* > 0: the batch is ONE sequence of this length (its offset tables need not be read: a PCIe round trip less) */,
    uint32_t parts /* workgroups per query, a power of two (gridDim.x = queries * parts).  > 1 -- a handful of queries whose positions fit one pass of
                      the workgroup (n <= blockDim.x), no sorted copy: every part dedupes the whole query (same table, same ranks: the
                      representative of a k-mer is its smallest position whatever the order of the inserts), then hashes and writes
                      only its share of the unique k-mers.  One CU takes 4 us to hash the ~970 k-mers of a 1 kbp query for 4 seeds
                      (scripts/ab_k1_phases.py): that part of a latency-bound call is ALU work, and this spreads it over `parts` CUs */,
    const SeqArg<ARGB> sarg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t q = blockIdx.x / parts, part = blockIdx.x - q * parts, n_queries = gridDim.x / parts;
    if (pos_off_out && threadIdx.x == 0 && part == 0) {
        if (one_len) {
            pos_off_out[0] = 0;
            pos_off_out[1] = one_len >= k ? one_len - k + 1 : 0u;
        } else {
            pos_off_out[q] = pos_off[q];
            if (q + 1 == n_queries) pos_off_out[n_queries] = pos_off[n_queries];
        }
    }
    if (preset && part == 0) {
        uint64_t *pq = preset + (uint64_t)q * preset_words;
        for (uint64_t i = threadIdx.x; i < preset_words; i += blockDim.x) pq[i] = preset_value;
    }
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    uint32_t *scan = tab + tab_cap + tab_cap / 32 + 4;   // 16 entries (the table is followed by the pad words of the row sort)
    uint32_t *hs = scan + 16;                            // hs[i]: 32-bit hash of the k-mer at position i (hs_cap entries)
    char *sq = reinterpret_cast<char *>(hs + hs_cap);    // the query's bytes
    char *sc = sq + sq_bytes;                            // KF == 31: their complements, 4 pad bytes in front (kmer31_canonical_premix)
    const char *s = one_len ? seqs : seqs + seq_off[q];
    const uint32_t len = one_len ? one_len : (uint32_t)(seq_off[q + 1] - seq_off[q]);
    const uint32_t n = len >= k ? len - k + 1 : 0u;
    const uint64_t P = one_len ? 0ull : pos_off[q];
    uint32_t tsize = 2;
    while (tsize < tab_mult * n) tsize <<= 1;            // load factor <= 1/tab_mult: short probe chains
    const uint32_t mask = tsize - 1;
    BIGSI_PHASE(0);
    for (uint32_t i = threadIdx.x; i < tsize; i += blockDim.x) tab[i] = kEmpty;
    if constexpr (ARGB > 0) {                            // (one_len <= ARGB, zero-padded to a word by the host; sq / sc hold sq_bytes >= that)
        for (uint32_t j = threadIdx.x; j < (len + 3) / 4; j += blockDim.x) {
            const uint32_t w = sarg.w[j];
            reinterpret_cast<uint32_t *>(sq)[j] = w;
            if (KF == 31)
                reinterpret_cast<uint32_t *>(sc + 4)[j] = (uint32_t)complement((uint8_t)w) | (uint32_t)complement((uint8_t)(w >> 8)) << 8 |
                                                          (uint32_t)complement((uint8_t)(w >> 16)) << 16 | (uint32_t)complement((uint8_t)(w >> 24)) << 24;
        }
    } else
    for (uint32_t base = 0; base < len; base += 4 * blockDim.x) {       // four loads in flight per thread and pass
        char c[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t i = base + e * blockDim.x + threadIdx.x;
            c[e] = i < len ? s[i] : (char)0;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t i = base + e * blockDim.x + threadIdx.x;
            if (i >= len) continue;
            sq[i] = c[e];
            if (KF == 31) sc[4 + i] = (char)complement((uint8_t)c[e]);
        }
    }
    __syncthreads();
    BIGSI_PHASE(1);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        if (KF == 31) {
            uint32_t wf[8];
            kmer31_words(reinterpret_cast<const uint32_t *>(sq), i, wf);
            hs[i] = kmer31_fingerprint(wf);
        } else {
            hs[i] = dedupe_hash<KF>(sq + i, k);
        }
    }
    __syncthreads();
    BIGSI_PHASE(2);
    // two positions hold the same k-mer iff their bytes are equal; the stored hashes settle almost every comparison with
    // one LDS word instead of a divergent byte loop
    uint32_t my_slot = 0;                                // where this thread's FIRST position ended up
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t hv = hs[i];
        uint32_t slot = hv & mask;
        for (;;) {
            const uint32_t cur = atomicCAS(&tab[slot], kEmpty, i);
            if (cur == kEmpty) break;
            if (hs[cur] == hv && kmer_equal(sq + cur, sq + i, k)) { atomicMin(&tab[slot], i); break; }
            slot = (slot + 1) & mask;
        }
        if (i == threadIdx.x) my_slot = slot;
    }
    __syncthreads();
    BIGSI_PHASE(3);
    uint32_t *fp = first_pos + P, *ux = uidx + P, *pu = pos_unique + P, *rp = rep_out + P;
    uint64_t *qrows = rows + P * h;
    // (one scan over threads that each take several CONSECUTIVE positions -- fewer barriers -- measured slower: 0.37 against
    // 0.29 ms per 8192 queries; the strided positions keep the LDS reads and the global writes of a wavefront together)
    uint32_t u = 0;
    if (parts > 1) {
        // (n <= blockDim.x: thread i holds position i.)  Ranks as below; then the list of first positions and the rank of every
        // representative go to LDS (the fingerprints and the table are no longer needed), and this part takes its share of both the
        // per-position outputs and the unique k-mers -- whose hashing is repacked onto the first threads of the workgroup.
        const uint32_t i = threadIdx.x;
        const uint32_t c = i < n ? tab[my_slot] : kEmpty;
        const uint32_t flag = (i < n && c == i) ? 1u : 0u;
        const uint32_t pre = block_exclusive_scan_flag(flag != 0, &u, scan);       // (its barrier: every thread has read its table slot)
        BIGSI_PHASE(4);
        if (flag) {
            hs[pre] = i;
            tab[i] = pre;
        }
        __syncthreads();
        const uint32_t pshift = 31u - (uint32_t)__clz((int)parts);         // (parts is a power of two: no division on the chain)
        const uint32_t per_i = (n + parts - 1) >> pshift, i0 = part * per_i, i1 = min(n, i0 + per_i);
        if (i >= i0 && i < i1) {
            rp[i] = c;
            pu[i] = tab[c];
            if (flag) ux[i] = pre;
        }
        const uint32_t per_j = (u + parts - 1) >> pshift, j0 = min(u, part * per_j), j1 = min(u, j0 + per_j);
        BIGSI_PHASE(5);
        for (uint32_t t = threadIdx.x; t < (j1 - j0) * h; t += blockDim.x) {      // a thread per (unique k-mer, seed): the shortest chain
            const uint32_t tj = t / h, j = j0 + tj, sd = t - tj * h, pos = hs[j];
            if (sd == 0) fp[j] = pos;
            kmer_rows<KF>(sq, sc, pos, k, sd, sd + 1, m, qrows + (uint64_t)j * h);
        }
        if (threadIdx.x == 0 && part == 0) {
            num_kmers[q] = n;
            num_unique[q] = u;
            const double mk = ceil((double)u * threshold);
            min_kmers[q] = mk > 0.0 ? (uint32_t)mk : 0u;
        }
        BIGSI_PHASE(6);
        return;
    }
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        uint32_t c = kEmpty;
        if (i < n && base == 0) {
            c = tab[my_slot];           // the slot holds the smallest position with this k-mer once all inserts are done
        } else if (i < n) {
            const uint32_t hv = hs[i];
            uint32_t slot = hv & mask;
            for (;;) {
                c = tab[slot];
                if (c == i || (hs[c] == hv && kmer_equal(sq + c, sq + i, k))) break;
                slot = (slot + 1) & mask;
            }
        }
        if (i < n) {
            rp[i] = c;
        }
        const uint32_t flag = (i < n && c == i) ? 1u : 0u;
        uint32_t tot;
        const uint32_t pre = block_exclusive_scan(flag, &tot, scan);
        if (flag) {
            const uint32_t j = u + pre;
            fp[j] = i;
            ux[i] = j;
            kmer_rows<KF>(sq, sc, i, k, 0, h, m, qrows + (uint64_t)j * h);
        }
        u += tot;
    }
    __syncthreads();
    BIGSI_PHASE(4);
    if (rows_sorted) {
        // K1e fused: counting sort of the u*h row ids by their top bits, the dedupe table's LDS reused as the histogram
        // (tsize >= 2n buckets: about one row per bucket at h <= 4).  Order inside a bucket depends on atomics; K2's result does not.
        uint32_t shift = 0;
        while (((m - 1) >> shift) >= (uint64_t)tsize) shift++;
        const uint32_t R = u * h;
        uint64_t *qsorted = rows_sorted + P * h;
        // (phase timestamps: this sort was 23 of a workgroup's 52 us -- two passes of dependent read-backs of the row ids
        // from L2, and 16-way bank conflicts where a thread walks its 16 consecutive buckets.  Now a thread reads its row
        // ids once, all loads in flight together, and keeps them in registers for the second pass; bucket b lives at word
        // b + b / 32, which spreads the threads' bucket runs over the banks.)
        auto at = [](uint32_t b) { return b + (b >> 5); };
        for (uint32_t i = threadIdx.x; i < tsize + (tsize >> 5) + 1; i += blockDim.x) tab[i] = 0;
        __syncthreads();
        constexpr int kSortRegs = 16;
        const bool in_regs = R <= (uint32_t)kSortRegs * blockDim.x;
        uint64_t mine[kSortRegs];
        if (in_regs) {
#pragma unroll
            for (int t = 0; t < kSortRegs; t++) {
                const uint32_t r = threadIdx.x + (uint32_t)t * blockDim.x;
                mine[t] = r < R ? qrows[r] : ~0ull;
            }
#pragma unroll
            for (int t = 0; t < kSortRegs; t++)
                if (mine[t] != ~0ull) atomicAdd(&tab[at((uint32_t)(mine[t] >> shift))], 1u);
        } else {
            for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) atomicAdd(&tab[at((uint32_t)(qrows[r] >> shift))], 1u);
        }
        __syncthreads();
        const uint32_t per = (tsize + blockDim.x - 1) / blockDim.x, b0 = threadIdx.x * per;
        uint32_t sum = 0, tot;
        for (uint32_t j = 0; j < per; j++) sum += b0 + j < tsize ? tab[at(b0 + j)] : 0u;
        uint32_t run = block_exclusive_scan(sum, &tot, scan);
        for (uint32_t j = 0; j < per && b0 + j < tsize; j++) {
            const uint32_t v = tab[at(b0 + j)];
            tab[at(b0 + j)] = run;
            run += v;
        }
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int t = 0; t < kSortRegs; t++)
                if (mine[t] != ~0ull) qsorted[atomicAdd(&tab[at((uint32_t)(mine[t] >> shift))], 1u)] = mine[t];
        } else {
            for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) {
                const uint64_t row = qrows[r];
                qsorted[atomicAdd(&tab[at((uint32_t)(row >> shift))], 1u)] = row;
            }
        }
    }
    BIGSI_PHASE(5);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) pu[i] = ux[rp[i]];
    BIGSI_PHASE(6);
    if (threadIdx.x == 0) {
        num_kmers[q] = n;
        num_unique[q] = u;
        const double mk = ceil((double)u * threshold);
        min_kmers[q] = mk > 0.0 ? (uint32_t)mk : 0u;
    }
}

// K1 for batches whose k-mers arrive as explicit byte strings ("elements", bigsi_hip_batch_create_elements): the host has
// k-merised, deduplicated and canonicalised (a non-ASCII query: the reference windows k CHARACTERS and hashes their UTF-8 bytes,
// utils/fncts.py:38-65, bloom/bloomfilter.py:5-6); what is left of K1 is MurmurHash3 over each element's bytes with seeds
// 0..h-1, the floor-mod, and min_kmers.
__global__ __launch_bounds__(kBlock) void k_rows_raw(
    const char *__restrict__ blob, const uint64_t *__restrict__ elem_off, const uint64_t *__restrict__ seq_elem_off,
    const uint64_t *__restrict__ pos_off, uint32_t h, uint64_t m, double threshold, uint64_t *__restrict__ rows,
    uint32_t *__restrict__ num_kmers, uint32_t *__restrict__ num_unique, uint32_t *__restrict__ min_kmers)
{
    const uint32_t q = blockIdx.x;
    const uint64_t e0 = seq_elem_off[q], P0 = pos_off[q];
    const uint32_t u = (uint32_t)(seq_elem_off[q + 1] - e0);
    for (uint32_t j = threadIdx.x; j < u; j += kBlock) {
        const uint64_t a = elem_off[e0 + j];
        const KmerView v{blob + a, (uint32_t)(elem_off[e0 + j + 1] - a), false};
        for (uint32_t sd = 0; sd < h; sd++) rows[(P0 + j) * h + sd] = row_of_hash(murmur3_32(v, sd), m);
    }
    if (threadIdx.x == 0) {
        num_kmers[q] = (uint32_t)(pos_off[q + 1] - P0);
        num_unique[q] = u;
        const double mk = ceil((double)u * threshold);
        min_kmers[q] = mk > 0.0 ? (uint32_t)mk : 0u;
    }
}

// K1 for probe / read-length queries (at most 64 k-mer positions, e.g. the 61-mers of BASELINE config 2): ONE WAVEFRONT
// PER QUERY, four queries per workgroup, lane i = position i.  Duplicates are found by broadcasting each lane's dedupe
// hash in turn (64 shuffles, string compare only on a hash match), first occurrences are ranked with ballot + popcount:
// no LDS, no atomics, no barriers.  Same outputs as the other two routes.
template <int KF>
__global__ __launch_bounds__(kBlock) void k_kmerize_wave(
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off,
    uint32_t k, uint32_t h, uint64_t m, double threshold, uint32_t n_seqs, uint32_t *__restrict__ first_pos,
    uint32_t *__restrict__ pos_unique, uint32_t *__restrict__ rep_out, uint64_t *__restrict__ rows,
    uint32_t *__restrict__ num_kmers, uint32_t *__restrict__ num_unique, uint32_t *__restrict__ min_kmers,
    uint64_t *__restrict__ preset, uint64_t preset_words, uint64_t preset_value, uint64_t *__restrict__ pos_off_out /* as k_kmerize_lds */)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= n_seqs) return;                       // whole wavefront leaves together
    if (pos_off_out && lane == 0) {
        pos_off_out[q] = pos_off[q];
        if (q + 1 == n_seqs) pos_off_out[n_seqs] = pos_off[n_seqs];
    }
    if (preset) {
        uint64_t *pq = preset + (uint64_t)q * preset_words;
        for (uint64_t i = lane; i < preset_words; i += 64) pq[i] = preset_value;
    }
    const char *s = seqs + seq_off[q];
    const uint32_t len = (uint32_t)(seq_off[q + 1] - seq_off[q]);
    const uint32_t n = len >= k ? len - k + 1 : 0u;      // <= 64 by the launch condition
    const uint64_t P = pos_off[q];
    const bool live = lane < n;
    RegKmer<KF> reg;                               // the k-mer's bytes, loaded once for the fingerprint and the hashes
    uint32_t fp = 0;
    if (KF > 0) {
        if (live) {
            reg.load(s + lane);
            fp = reg.fnv();
        }
    } else if (live) {
        fp = fnv1a(s + lane, k);
    }
    uint32_t rep = lane;
    for (uint32_t j = 0; j + 1 < n; j++) {         // wave-uniform trip count and lane index: a v_readlane, no LDS round trip
        const uint32_t fj = (uint32_t)__builtin_amdgcn_readlane((int)fp, (int)j);
        if (live && lane > j && rep == lane && fp == fj && kmer_equal(s + j, s + lane, k)) rep = j;
    }
    const bool first = live && rep == lane;
    const unsigned long long mask = __ballot(first);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t u = (uint32_t)__popcll(mask);
    if (live) {
        rep_out[P + lane] = rep;
        pos_unique[P + lane] = (uint32_t)__popcll(mask & (rep ? (~0ull >> (64 - rep)) : 0ull));   // rank of its representative
    }
    if (first) {
        const uint32_t j = (uint32_t)__popcll(mask & below);
        first_pos[P + j] = lane;
        uint64_t *dst = rows + (P + j) * h;
        const char *km = s + lane;
        if (KF > 0) {
            uint32_t w[(KF + 3) / 4 > 0 ? (KF + 3) / 4 : 1];
            reg.canonical_words(w);
            for (uint32_t sd = 0; sd < h; sd++) dst[sd] = row_of_hash(murmur3_words<KF>(w, sd), m);
        } else {
            const KmerView v{km, k, use_revcomp(km, k)};
            for (uint32_t sd = 0; sd < h; sd++) dst[sd] = row_of_hash(murmur3_32(v, sd), m);
        }
    }
    if (lane == 0) {
        num_kmers[q] = n;
        num_unique[q] = u;
        const double mk = ceil((double)u * threshold);
        min_kmers[q] = mk > 0.0 ? (uint32_t)mk : 0u;
    }
}

// ------------------------------------------------------------------------------ K1e: visit rows in address order
// AND (and +1-per-k-mer) are order-free, so each query's row list is bucket-sorted by row id before K2 streams it: all
// resident workgroups then sweep the index from low to high addresses together instead of scattering over 125 GB: with
// launches small enough to be co-resident (bigsi_hip_batch_run) the row-AND kernel runs at 0.857 of peak against 0.79 for
// rows in hash order, whatever the launch size (DRAM page / TLB locality).
// kSortBuckets buckets over [0, m) (about two per row of a 1 kbp query, i.e. nearly a full sort): LDS histogram -> scan -> scatter.  `group` = 1 sorts rows individually (exact path),
// `group` = h keeps each k-mer's h rows together and sorts k-mers by their first row (counting path).
// The order inside a bucket depends on atomics; the results of K2 do not.
constexpr int kSortBuckets = 8192;
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sort_rows(
    const uint64_t *__restrict__ rows, uint64_t *__restrict__ sorted, const uint64_t *__restrict__ pos_off,
    const uint32_t *__restrict__ num_unique, uint32_t h, uint32_t group, uint32_t shift)
{
    __shared__ uint32_t hist[kSortBuckets];
    __shared__ uint32_t lds[16];
    constexpr int kPer = kSortBuckets / BLOCK;      // consecutive buckets scanned by one thread
    const uint32_t q = blockIdx.x;
    const uint64_t n_items = (uint64_t)num_unique[q] * h / group;
    const uint64_t *src = rows + pos_off[q] * h;
    uint64_t *dst = sorted + pos_off[q] * h;
    for (uint32_t i = threadIdx.x; i < kSortBuckets; i += BLOCK) hist[i] = 0;
    __syncthreads();
    for (uint64_t i = threadIdx.x; i < n_items; i += BLOCK) {
        const uint64_t b = src[i * group] >> shift;
        atomicAdd(&hist[b < kSortBuckets ? b : kSortBuckets - 1], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the histogram
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) { v[j] = hist[threadIdx.x * kPer + j]; sum += v[j]; }
        uint32_t tot;
        uint32_t run = block_exclusive_scan(sum, &tot, lds);
#pragma unroll
        for (int j = 0; j < kPer; j++) { hist[threadIdx.x * kPer + j] = run; run += v[j]; }
    }
    __syncthreads();
    for (uint64_t i = threadIdx.x; i < n_items; i += BLOCK) {
        const uint64_t b = src[i * group] >> shift;
        const uint32_t pos = atomicAdd(&hist[b < kSortBuckets ? b : kSortBuckets - 1], 1u);
        for (uint32_t s = 0; s < group; s++) dst[(uint64_t)pos * group + s] = src[i * group + s];
    }
}

// ------------------------------------------------------------------------------ K2 work decomposition
// One wavefront streams one 128-word (1 KiB) column segment of every row a query needs: lane l holds words
// [seg*128 + 2l, +2) -> each row read is ONE coalesced 1 KiB wave instruction (global_load_dwordx4), each
// needed byte of a row is read exactly once, and row ids arrive through scalar loads (they are wave-uniform).
// blockIdx -> (query, tile) is XCD-aware: hardware places block b on XCD b % 8, so all tiles of a query get the
// same residue and run on one XCD, adjacent in dispatch order (their row-id lists and the address translations
// of a row's page are shared in that XCD's L2).
// Small batches do not fill the chip that way (one 1 kbp query on a 100k-sample index is 13 wavefronts), so a launch may
// also cut every query's row list into `slices` pieces handled by different workgroups, which combine through atomics
// (AND for the exact bitmap, ADD for counters).  slices == 1 is the plain, atomic-free path used by large batches.
struct TileMap {
    uint32_t q, tile, slice;
    bool valid;
};
// A launch covers the queries [q0, n_seqs): large batches are cut into launches of about a thousand workgroups, all
// co-resident, which then sweep the (address-ordered) row lists together -- a grid several times the chip's residency
// desynchronises that sweep and measured 2.5 % slower at C3.
__device__ __forceinline__ TileMap map_block(uint32_t b, uint32_t q0, uint32_t n_seqs, uint32_t tiles, uint32_t slices = 1)
{
    const uint32_t per_q = tiles * slices;
    if (slices > 1) {
        // a small batch (few queries, sliced): consecutive workgroups -- different XCDs -- take the slices of one query.  With the
        // map below a query lives on ONE XCD, which for a single query is an eighth of the chip and of its bandwidth: one 1 kbp
        // query against 100 k samples took 44 us (1.1 TB/s) at any number of slices.
        const uint32_t ql = b / per_q, rem = b - ql * per_q;
        const uint32_t q = q0 + ql;
        return TileMap{q, rem / slices, rem % slices, q < n_seqs};
    }
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t ql = slot / per_q, rem = slot - ql * per_q;
    const uint32_t q = q0 + ql * 8u + xcd;
    return TileMap{q, rem / slices, rem % slices, q < n_seqs};
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <bool NT = true>
__device__ __forceinline__ u64x2 load_row_seg(const uint64_t *__restrict__ index, uint64_t row, uint64_t stride_words, uint32_t w0)
{
    const u64x2 *p = reinterpret_cast<const u64x2 *>(index + row * stride_words + w0);
    if (NT) return __builtin_nontemporal_load(p);   // streamed once: keep it out of the way of the row-id lists in L2
    return *p;
}

// VEC consecutive 64-column words of a row, streamed: 2 = one 16-byte load (every row kernel's unit), 1 = one 8-byte load
// (k_reads_fused on rows of <= kBlock words: half the registers per lane, twice the lanes that hold columns)
template <int VEC>
struct RowWords;
template <>
struct RowWords<2> {
    u64x2 v;
    static __device__ __forceinline__ RowWords load(const uint64_t *__restrict__ index, uint64_t row, uint64_t stride_words, uint32_t w0) { return RowWords{load_row_seg(index, row, stride_words, w0)}; }
    static __device__ __forceinline__ RowWords fill(uint64_t x) { return RowWords{u64x2{x, x}}; }
    __device__ __forceinline__ RowWords &operator&=(const RowWords &o) { v &= o.v; return *this; }
    __device__ __forceinline__ uint64_t word(int e) const { return e ? v.y : v.x; }
};
template <>
struct RowWords<1> {
    uint64_t v;
    static __device__ __forceinline__ RowWords load(const uint64_t *__restrict__ index, uint64_t row, uint64_t stride_words, uint32_t w0) { return RowWords{__builtin_nontemporal_load(index + row * stride_words + w0)}; }
    static __device__ __forceinline__ RowWords fill(uint64_t x) { return RowWords{x}; }
    __device__ __forceinline__ RowWords &operator&=(const RowWords &o) { v &= o.v; return *this; }
    __device__ __forceinline__ uint64_t word(int) const { return v; }
};

// ------------------------------------------------------------------------------ K2 + K3a: exact
// AND of every row of every unique k-mer of the query (graph/index.py:75-80 then graph/bigsi.py:192-195):
// out[q][w] for w < wv.  Sequences without k-mers produce an all-zero bitmap (the host shim raises for them).
template <int UNROLL, bool NT = true>
__global__ __launch_bounds__(1024) void k_and_exact(
    const uint64_t *__restrict__ index, uint64_t stride_words, uint32_t wv, uint64_t n_cols,
    const uint64_t *__restrict__ rows, const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ num_unique,
    uint32_t h, uint32_t q0, uint32_t n_seqs, uint32_t tiles, uint64_t *__restrict__ out, uint64_t out_stride_words,
    uint32_t slices /* > 1: `out` was preset to all ones and slices combine with atomicAnd */,
    uint32_t early_exit /* 1: a wavefront stops fetching once its 8192-column segment of the running AND is all zero */)
{
    const TileMap tm = map_block(blockIdx.x, q0, n_seqs, tiles, slices);
    if (!tm.valid) return;
    BIGSI_PHASE_IF(blockIdx.x == 0, 1001, 0);
    BIGSI_PHASE_IF(blockIdx.x == gridDim.x - 4, 1002, 0);
    const uint32_t w0 = (tm.tile * blockDim.x + threadIdx.x) * kVec;
    if (w0 >= wv) return;
    const uint64_t Rall = (uint64_t)num_unique[tm.q] * h;
    BIGSI_PHASE_IF(blockIdx.x == 0, 1001, 1);
    const uint64_t per = (Rall + slices - 1) / slices;
    uint64_t r = (uint64_t)tm.slice * per;
    const uint64_t R = r + per < Rall ? r + per : Rall;
    if (slices > 1 && r >= R && Rall != 0) return;        // nothing in this slice
    const uint64_t *qrows = rows + pos_off[tm.q] * h;
    u64x2 acc = {~0ull, ~0ull};
    for (; r + UNROLL <= R; r += UNROLL) {
        u64x2 v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) v[j] = load_row_seg<NT>(index, qrows[r + j], stride_words, w0);
#pragma unroll
        for (int j = 0; j < UNROLL; j++) acc &= v[j];
        // opt-in (BIGSI_RUN_EARLY_EXIT): no further row can set a bit again, so the result is unchanged; the bytes of the
        // remaining rows are simply not read (NOT used by bench.py, whose algorithmic bytes assume every row is fetched)
        if (early_exit && __ballot((acc.x | acc.y) != 0) == 0) { r = R; break; }
    }
    for (; r < R; r++) acc &= load_row_seg<NT>(index, qrows[r], stride_words, w0);
    if (Rall == 0) acc = u64x2{0ull, 0ull};
    acc.x &= valid_mask(w0, n_cols);
    acc.y &= valid_mask(w0 + 1, n_cols);
    BIGSI_PHASE_IF(blockIdx.x == 0 && (acc.x | 1ull), 1001, 2);
    BIGSI_PHASE_IF(blockIdx.x == gridDim.x - 4 && (acc.x | 1ull), 1002, 2);
    uint64_t *o = out + (uint64_t)tm.q * out_stride_words + w0;
    if (slices > 1) {
        atomicAnd((unsigned long long *)o, (unsigned long long)acc.x);
        if (w0 + 1 < out_stride_words) atomicAnd((unsigned long long *)(o + 1), (unsigned long long)acc.y);
    } else {
        o[0] = acc.x;
        if (w0 + 1 < out_stride_words) o[1] = acc.y;
    }
}

// ------------------------------------------------------------------------------ K2 + K3b: per-sample counts
// For every unique k-mer: AND of its h rows (graph/index.py:75-80), then +1 into the counters of the samples whose
// bit survived (unpack_and_sum, graph/bigsi.py:35-44).  Counters are bit-sliced: P planes of uint64 per word, a
// ripple-carry add of the AND-ed word costs 3 bit-ops per plane, far below what the HBM stream leaves the VALU
// (see DESIGN.md).  Planes are expanded to integers once per (query, segment) and stored as CountT.
template <int P, int H, typename CountT, int KMX = 1 /* 2: software-pipelined loads, for grids too small to fill the SIMDs
    with wavefronts (128 x 4 kbp queries = 2 wavefronts per SIMD: 5.6 -> 6.3 TB/s) */,
          int VEC = kVec /* 64-column words per lane: 2 (16-byte loads), or 1 (8-byte loads: half the plane registers per lane) */>
__global__ __launch_bounds__(kBlock) void k_and_count(
    const uint64_t *__restrict__ index, uint64_t stride_words, uint32_t wv,
    const uint64_t *__restrict__ rows, const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ num_unique,
    uint32_t h_rt, uint32_t q0, uint32_t n_seqs, uint32_t tiles, CountT *__restrict__ out, uint64_t out_stride /* counters per query */,
    const uint32_t *__restrict__ min_kmers, uint64_t n_cols, uint64_t *__restrict__ hit_bitmap /* [seq][bm_stride] or null */,
    uint64_t bm_stride, uint32_t sparse /* 1: store counters only for words that contain a hit */,
    uint32_t slices /* > 1: a small batch, every query's k-mers cut into slices handled by different workgroups */,
    uint32_t early_exit /* 1 (only with sparse, one slice): a wavefront stops fetching once none of its 8192 columns can reach min_kmers */,
    uint64_t *__restrict__ partial /* slices > 1: where a slice leaves the low `planes_out` planes of its partial counts, bit-sliced as
                                      they are -- [query][slice][plane][bm_stride words]; k_count_combine adds the slices up */,
    uint32_t planes_out)
{
    const TileMap tm = map_block(blockIdx.x, q0, n_seqs, tiles, slices);
    if (!tm.valid) return;
    const uint32_t w0 = (tm.tile * blockDim.x + threadIdx.x) * VEC;
    if (w0 >= wv) return;
    const uint32_t h = H > 0 ? (uint32_t)H : h_rt;
    const uint32_t uall = num_unique[tm.q];
    const uint32_t per = (uall + slices - 1) / slices;
    const uint32_t j0 = tm.slice * per;
    const uint32_t u = j0 + per < uall ? j0 + per : uall;      // this slice covers unique k-mers [j0, u)
    if (slices > 1 && j0 >= u) return;
    const uint64_t *qrows = rows + pos_off[tm.q] * h;
    uint64_t pl[VEC][P];
#pragma unroll
    for (int v = 0; v < VEC; v++)
#pragma unroll
        for (int p = 0; p < P; p++) pl[v][p] = 0;

    auto add = [&](const RowWords<VEC> &a) {
        uint64_t c[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) c[e] = a.word(e);
#pragma unroll
        for (int p = 0; p < P; p++) {
#pragma unroll
            for (int e = 0; e < VEC; e++) {          // (the words' carry chains interleaved: independent instructions side by side)
                const uint64_t t = pl[e][p] & c[e];
                pl[e][p] ^= c[e];
                c[e] = t;
            }
        }
    };

    // opt-in (BIGSI_RUN_EARLY_EXIT): with `left` k-mers still to come, a column whose count is below min_kmers - left cannot
    // become a hit any more; when that holds for every column of the wavefront's segment the remaining rows need not be read
    // (same hit lists, fewer bytes than the reference reads -- hence not the default).  One bit-sliced comparison per 32 k-mers.
    const uint32_t thr_exit = early_exit ? min_kmers[tm.q] : 0u;
    auto hopeless = [&](uint32_t left) -> bool {
        if (thr_exit <= left) return false;
        const uint32_t need = thr_exit - left;
        uint64_t any = 0;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            uint64_t gt = 0, eq = ~0ull;
            if (P < 32 && (need >> (P & 31)) != 0) eq = 0;
#pragma unroll
            for (int p = P - 1; p >= 0; p--) {
                if ((need >> p) & 1u) eq &= pl[v][p];
                else { gt |= eq & pl[v][p]; eq &= ~pl[v][p]; }
            }
            any |= gt | eq;
        }
        return __ballot(any != 0) == 0ull;
    };

    uint32_t j = j0;
    if (H > 0) {
        // KM k-mers per iteration so that 8-12 independent row loads are in flight per lane whatever h is (h=3: 12 loads
        // measured +1.9 % over 6; h=4: 8 vs 16 no difference)
        constexpr int KM0 = H == 1 ? 8 : H <= 3 ? 4 : 2;
        constexpr int KM = KMX == 3 ? (KM0 > 1 ? KM0 / 2 : 1) : KM0;      // KMX == 3: half the loads in flight per lane (launches with > ~1900 live wavefronts)
        if (KMX == 2) {
            // software pipeline: the loads of the NEXT KM k-mers are issued before the bit-sliced adds of the current ones, so
            // a wavefront keeps the memory system busy through its own ALU phase (matters when few wavefronts share a SIMD)
            RowWords<VEC> cur[KM * (H > 0 ? H : 1)], nxt[KM * (H > 0 ? H : 1)];
            if (j + KM <= u) {
#pragma unroll
                for (int s = 0; s < KM * H; s++) cur[s] = RowWords<VEC>::load(index, qrows[(uint64_t)j * H + s], stride_words, w0);
            }
            for (; j + KM <= u; j += KM) {
                const bool more = j + 2 * KM <= u;       // wave-uniform
                if (more) {
#pragma unroll
                    for (int s = 0; s < KM * H; s++) nxt[s] = RowWords<VEC>::load(index, qrows[(uint64_t)(j + KM) * H + s], stride_words, w0);
                }
#pragma unroll
                for (int g = 0; g < KM; g++) {
                    RowWords<VEC> a = cur[g * H];
#pragma unroll
                    for (int s = 1; s < H; s++) a &= cur[g * H + s];
                    add(a);
                }
                if (more) {
#pragma unroll
                    for (int s = 0; s < KM * H; s++) cur[s] = nxt[s];
                }
            }
        } else {
            for (; j + KM <= u; j += KM) {
                if (early_exit && ((j - j0) & 31u) < (uint32_t)KM && j > j0 && hopeless(u - j)) { j = u; break; }
                RowWords<VEC> v[KM * (H > 0 ? H : 1)];
#pragma unroll
                for (int s = 0; s < KM * H; s++) v[s] = RowWords<VEC>::load(index, qrows[(uint64_t)j * H + s], stride_words, w0);
#pragma unroll
                for (int g = 0; g < KM; g++) {
                    RowWords<VEC> a = v[g * H];
#pragma unroll
                    for (int s = 1; s < H; s++) a &= v[g * H + s];
                    add(a);
                }
            }
        }
    }
    for (; j < u; j++) {     // tail k-mers, and every k-mer when h is a run-time value (h > 5): rows in groups of four loads
        const uint64_t *kr = qrows + (uint64_t)j * h;
        RowWords<VEC> a = RowWords<VEC>::fill(~0ull);
        uint32_t s = 0;
        for (; s + 4 <= h; s += 4) {
            RowWords<VEC> l0 = RowWords<VEC>::load(index, kr[s], stride_words, w0), l1 = RowWords<VEC>::load(index, kr[s + 1], stride_words, w0);
            const RowWords<VEC> l2 = RowWords<VEC>::load(index, kr[s + 2], stride_words, w0), l3 = RowWords<VEC>::load(index, kr[s + 3], stride_words, w0);
            l0 &= l2; l1 &= l3; l0 &= l1;
            a &= l0;
        }
        for (; s < h; s++) a &= RowWords<VEC>::load(index, kr[s], stride_words, w0);
        add(a);
    }

    if (slices > 1) {
        // a slice of a small batch: its partial counts stay bit-sliced -- a slice of s k-mers needs ceil(log2(s + 1)) planes, 5 for the
        // 16 k-mers of a 1 kbp query's slice -- and go to scratch memory as coalesced 16-byte stores; k_count_combine adds the slices of
        // a word with bit-sliced adders, thresholds and expands.  (Round 3 expanded every slice's planes to integers and added them
        // with packed atomics: 31 us for one 1 kbp query on 100 k samples, against 14 us for the exact AND of the same rows.)
        uint64_t *o = partial + ((uint64_t)tm.q * slices + tm.slice) * planes_out * bm_stride + w0;
#pragma unroll
        for (int p = 0; p < P; p++)
            if ((uint32_t)p < planes_out) {
                if (VEC == 2) *reinterpret_cast<u64x2 *>(o + (uint64_t)p * bm_stride) = u64x2{pl[0][p], pl[VEC - 1][p]};
                else o[(uint64_t)p * bm_stride] = pl[0][p];
            }
        return;
    }
    // threshold in bit-sliced form (graph/bigsi.py:241-242: count >= min_kmers): MSB-first comparator over the planes,
    // ~2 bit-ops per plane per word; the threshold is wave-uniform so its bit tests are scalar branches
    uint64_t ge[VEC];
    {
        const uint32_t thr = min_kmers[tm.q];
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            uint64_t gt = 0, eq = ~0ull;
            if (P < 32 && (thr >> (P & 31)) != 0) eq = 0;     // threshold above any representable count
#pragma unroll
            for (int p = P - 1; p >= 0; p--) {
                if ((thr >> p) & 1u) eq &= pl[v][p];
                else { gt |= eq & pl[v][p]; eq &= ~pl[v][p]; }
            }
            ge[v] = (gt | eq) & valid_mask((uint64_t)w0 + v, n_cols);
            if (hit_bitmap && (uint64_t)w0 + v < bm_stride) hit_bitmap[(uint64_t)tm.q * bm_stride + w0 + v] = ge[v];
        }
    }

    // expand: column 8b+jj of word w sits at bit 8b+7-jj; 8 consecutive counters per store
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const uint64_t cbase = ((uint64_t)w0 + v) * 64;
        if (cbase >= out_stride) break;
        if (sparse && ge[v] == 0) continue;
        CountT *o = out + (uint64_t)tm.q * out_stride + cbase;
        // 64 columns x P planes: fully unrolled up to 16 planes; for the 32-plane (> 65535 k-mers) variant the byte loop
        // stays rolled (run-time shift amounts, plane indices still compile-time) to keep the code and registers bounded
        constexpr int kByteUnroll = P > 16 ? 1 : 8;
#pragma unroll kByteUnroll
        for (int b = 0; b < 8; b++) {
            CountT c[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const int bit = 8 * b + 7 - jj;
                uint32_t x = 0;
#pragma unroll
                for (int p = 0; p < P; p++) x |= (uint32_t)((pl[v][p] >> bit) & 1ull) << p;
                c[jj] = (CountT)x;
            }
            if (sizeof(CountT) == 2) {
                uint4 pk;
                pk.x = (uint32_t)c[0] | ((uint32_t)c[1] << 16); pk.y = (uint32_t)c[2] | ((uint32_t)c[3] << 16);
                pk.z = (uint32_t)c[4] | ((uint32_t)c[5] << 16); pk.w = (uint32_t)c[6] | ((uint32_t)c[7] << 16);
                *reinterpret_cast<uint4 *>(o + 8 * b) = pk;
            } else {
                uint4 lo{(uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2], (uint32_t)c[3]};
                uint4 hi{(uint32_t)c[4], (uint32_t)c[5], (uint32_t)c[6], (uint32_t)c[7]};
                *reinterpret_cast<uint4 *>(o + 8 * b) = lo;
                *reinterpret_cast<uint4 *>(o + 8 * b + 4) = hi;
            }
        }
    }
}


// The slices of a small counting batch added up.  A workgroup owns 32 consecutive 64-column words of one query; its 256 threads are
// 8 groups of 32 (thread = word x group), group g adding the slices g, g + 8, ... of its word with a bit-sliced ripple-carry adder
// (~5 bit-ops per plane), four slices' planes loaded at a time so that a thread has 4 x planes_in loads in flight (one thread per
// word walking all 60 slices of a 1 kbp query was 38 us of dependent round trips).  The eight partial sums meet in LDS; the word is
// thresholded (count >= min_kmers, graph/bigsi.py:241-242) into the hit mask that K4 compacts and column shards exchange, and its
// counters are expanded -- all of them, or (sparse) only those of words that hold a hit -- every group taking one byte (8 columns).
template <int P, typename CountT>
__global__ __launch_bounds__(kBlock) void k_count_combine(
    const uint64_t *__restrict__ partial, uint32_t slices, uint32_t planes_in, uint64_t bm_stride, uint32_t wv, uint32_t n_seqs,
    const uint32_t *__restrict__ num_unique, const uint32_t *__restrict__ min_kmers, uint64_t n_cols, uint64_t *__restrict__ hit_bitmap,
    CountT *__restrict__ out, uint64_t out_stride, uint32_t sparse)
{
    constexpr int G = 8, W = kBlock / G;                 // groups, words per workgroup
    __shared__ uint64_t red[G][P][W];
    const uint32_t wl = threadIdx.x & (W - 1), g = threadIdx.x / W;
    const uint32_t per_q = (wv + W - 1) / W, q = blockIdx.x / per_q, w = (blockIdx.x - q * per_q) * W + wl;
    if (q >= n_seqs) return;                             // (whole workgroup)
    const bool live_w = w < wv;
    // (slices that hold no k-mer of this query wrote nothing: the same rule as k_and_count's early return)
    const uint32_t uall = num_unique[q], per = (uall + slices - 1) / slices;
    const uint32_t live = per ? (uall + per - 1) / per : 0u;
    uint64_t acc[P];
#pragma unroll
    for (int p = 0; p < P; p++) acc[p] = 0;
    const uint64_t slice_words = (uint64_t)planes_in * bm_stride;
    const uint64_t *base = partial + (uint64_t)q * slices * slice_words + w;
    constexpr int KB = 12, PB = 4;
    if (live_w && planes_in <= (uint32_t)PB && live <= (uint32_t)KB * G) {
        // the usual shape of a latency-bound call (<= 96 slices of <= 15 k-mers each: 4 planes): ALL of the group's slices in one
        // batch of loads -- one round trip where the loop below makes three
        uint64_t x[KB][PB];
#pragma unroll
        for (int k = 0; k < KB; k++)
#pragma unroll
            for (int p = 0; p < PB; p++)
                x[k][p] = (g + k * G < live && (uint32_t)p < planes_in) ? base[(uint64_t)(g + k * G) * slice_words + (uint64_t)p * bm_stride] : 0ull;
#pragma unroll
        for (int k = 0; k < KB; k++) {
            uint64_t carry = 0;
#pragma unroll
            for (int p = 0; p < P; p++) {
                const uint64_t a = acc[p];
                if (p < PB) {
                    const uint64_t t = a ^ x[k][p < PB ? p : 0];
                    acc[p] = t ^ carry;
                    carry = (a & x[k][p < PB ? p : 0]) | (carry & t);
                } else {
                    acc[p] = a ^ carry;
                    carry &= a;
                }
            }
        }
    } else if (live_w) {
        for (uint32_t s0 = g; s0 < live; s0 += 4 * G) {
            uint64_t x[4][P];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int p = 0; p < P; p++)
                    x[k][p] = (s0 + k * G < live && (uint32_t)p < planes_in) ? base[(uint64_t)(s0 + k * G) * slice_words + (uint64_t)p * bm_stride] : 0ull;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint64_t carry = 0;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const uint64_t a = acc[p], t = a ^ x[k][p];
                    acc[p] = t ^ carry;
                    carry = (a & x[k][p]) | (carry & t);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++) red[g][p][wl] = acc[p];
    __syncthreads();
    if (g == 0) {                                        // the eight partial sums of a word -> its total, into red[0]
#pragma unroll 1
        for (int k = 1; k < G; k++) {
            uint64_t carry = 0;
#pragma unroll
            for (int p = 0; p < P; p++) {
                const uint64_t x = red[k][p][wl], a = acc[p], t = a ^ x;
                acc[p] = t ^ carry;
                carry = (a & x) | (carry & t);
            }
        }
#pragma unroll
        for (int p = 0; p < P; p++) red[0][p][wl] = acc[p];
    }
    __syncthreads();
    if (!live_w) return;
#pragma unroll
    for (int p = 0; p < P; p++) acc[p] = red[0][p][wl];
    const uint32_t thr = min_kmers[q];
    uint64_t gt = 0, eq = ~0ull;
    if (P < 32 && (thr >> (P & 31)) != 0) eq = 0;
#pragma unroll
    for (int p = P - 1; p >= 0; p--) {
        if ((thr >> p) & 1u) eq &= acc[p];
        else { gt |= eq & acc[p]; eq &= ~acc[p]; }
    }
    const uint64_t ge = (gt | eq) & valid_mask(w, n_cols);
    if (g == 0) hit_bitmap[(uint64_t)q * bm_stride + w] = ge;
    if (sparse && ge == 0) return;
    // group g expands byte g of the word: columns 8g .. 8g + 7 (column 8b + jj sits at bit 8b + 7 - jj)
    CountT *o = out + (uint64_t)q * out_stride + (uint64_t)w * 64 + 8 * g;
    CountT c[8];
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        const uint32_t bit = 8 * g + 7 - jj;
        uint32_t x = 0;
#pragma unroll
        for (int p = 0; p < P; p++) x |= (uint32_t)((acc[p] >> bit) & 1ull) << p;
        c[jj] = (CountT)x;
    }
    if (sizeof(CountT) == 2) {
        uint4 pk;
        pk.x = (uint32_t)c[0] | ((uint32_t)c[1] << 16); pk.y = (uint32_t)c[2] | ((uint32_t)c[3] << 16);
        pk.z = (uint32_t)c[4] | ((uint32_t)c[5] << 16); pk.w = (uint32_t)c[6] | ((uint32_t)c[7] << 16);
        *reinterpret_cast<uint4 *>(o) = pk;
    } else {
        uint4 lo{(uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2], (uint32_t)c[3]};
        uint4 hi{(uint32_t)c[4], (uint32_t)c[5], (uint32_t)c[6], (uint32_t)c[7]};
        *reinterpret_cast<uint4 *>(o) = lo;
        *reinterpret_cast<uint4 *>(o + 4) = hi;
    }
}

// ------------------------------------------------------------------------------ K4: threshold + compaction
// Result buffers are laid out [shard][seq][stride] (n_shards = 1 for a single GPU; > 1 for buffers gathered from column
// shards); hits come out as (colour, count) in (seq, shard, column) order.  Bit vectors -- every production path -- take
// two launches (k_hits_totals, k_hits_write); dense counter buffers take three passes: (a) hits per 2048-column chunk, (b) exclusive scan
// over chunks, (c) ordered write.  Colours ascend within a sequence
// (exact_filter's np.where order, graph/bigsi.py:193-204; inexact_filter's dict order before its stable sort, :215-229).
constexpr uint32_t kAllShards = 0xFFFFFFFFu;
constexpr uint32_t kChunkCols = 2048;   // counting: kBlock threads x 8 columns per chunk; exact: kBlock words (16384 columns)

__device__ __forceinline__ uint64_t chunk_index(uint32_t q, uint32_t shard, uint32_t chunk, uint32_t n_shards, uint32_t chunks)
{
    return ((uint64_t)q * n_shards + shard) * chunks + chunk;
}

// K4 for bit-vector inputs (the AND bitmap of an exact search, the hit mask of a thresholded one) in TWO launches and no
// waiting between workgroups.  Workgroup g owns the `ipb` consecutive items [g * ipb, +ipb) of the (seq, shard, chunk) order;
// the host picks ipb so that the grid is at most kHitsMaxGroups workgroups.
//   k_hits_totals  the group's hit total -> totals[g];
//   k_hits_write   every thread loads a share of the totals of the PRECEDING groups (plain loads: the launch boundary has made
//                  them visible), the workgroup sums them -- the exclusive prefix in one round trip -- and writes its items'
//                  (colour, count) in order (the words come back out of L2).
// Round 2 fused the two into one launch whose workgroups published their totals and spun on those of their predecessors:
// 7 us instead of 14 at BASELINE configs[1], but correct only while every workgroup of the grid was resident -- an assumption
// about the dispatcher and about whatever else runs on the device (other streams, other processes) that nothing enforces.
// The launch boundary costs ~5 us per batch (0.5 % of a 1 ms step at configs[3] / [4]; batches of reads do not come here, they
// write their hits from k_reads_fused) and removes the only unbounded wait the library had.
constexpr uint32_t kHitsMaxGroups = 1024;

// item ci of the (seq, shard, chunk) order; the host keeps the number of items below 2^31, so the divisions are 32-bit ones
// (a 64-bit division is a ~100-instruction routine on this hardware: three of them per item were most of K4's time on a
// latency-bound call -- 8.3 us for the seven items of one query on 100 k samples)
struct HitItem {
    uint32_t chunk, shard, q;
};
__device__ __forceinline__ HitItem hit_item(uint64_t ci, uint32_t n_shards, uint32_t chunks)
{
    const uint32_t c32 = (uint32_t)ci, sq = c32 / chunks;
    HitItem it;
    it.chunk = c32 - sq * chunks;
    it.shard = n_shards == 1 ? 0u : sq % n_shards;
    it.q = n_shards == 1 ? sq : sq / n_shards;
    return it;
}

__device__ __forceinline__ uint64_t hits_word(const uint64_t *__restrict__ bitmaps, uint64_t stride_words, uint32_t wv, uint32_t n_seqs,
                                              const HitItem &it, uint32_t *w_out)
{
    const uint32_t w = it.chunk * kBlock + threadIdx.x;   // one 64-column word per thread
    *w_out = w;
    return w < wv ? bitmaps[((uint64_t)it.shard * n_seqs + it.q) * stride_words + w] : 0ull;
}

__global__ __launch_bounds__(kBlock) void k_hits_totals(
    const uint64_t *__restrict__ bitmaps, uint64_t stride_words, uint32_t wv, uint32_t n_seqs, uint32_t n_shards, uint32_t chunks,
    uint32_t ipb, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t lds[16];
    const uint64_t grp = blockIdx.x, n_items = (uint64_t)n_seqs * n_shards * chunks;
    const uint64_t i0 = grp * ipb, i1 = i0 + ipb < n_items ? i0 + ipb : n_items;
    uint32_t mine = 0, w_unused;
    for (uint64_t ci = i0; ci < i1; ci++) mine += (uint32_t)__popcll(hits_word(bitmaps, stride_words, wv, n_seqs, hit_item(ci, n_shards, chunks), &w_unused));
    uint32_t gtot;
    block_exclusive_scan(mine, &gtot, lds);
    if (threadIdx.x == 0) totals[grp] = gtot;
}

__global__ __launch_bounds__(kBlock) void k_hits_write(
    const uint64_t *__restrict__ bitmaps, uint64_t stride_words, uint32_t wv, uint32_t n_seqs, uint32_t n_shards, uint32_t chunks,
    uint64_t shard_cols, const uint32_t *__restrict__ num_unique, uint32_t ipb, const uint32_t *__restrict__ totals,
    uint64_t *__restrict__ hit_off, uint32_t *__restrict__ hit_col, uint32_t *__restrict__ hit_cnt, uint64_t capacity,
    const void *__restrict__ counters, uint32_t counter_bytes, uint64_t counter_stride, uint32_t own_shard,
    uint64_t *exp_out /* non-null (single-workgroup launches of a one-call search only): this launch also writes the caller's block in
                         pinned memory (k_export_results' layout) and raises the flag the host spins on -- no export kernel, one launch
                         boundary less on a call that is a chain of them */,
    uint32_t exp_spec, const uint32_t *__restrict__ exp_uniq, volatile uint64_t *exp_flag, uint64_t exp_serial)
{
    __shared__ uint32_t lds[16];
    __shared__ uint64_t lds64[kBlock / 64];
    if (gridDim.x == 1 && n_shards == 1 && chunks <= 16) {
        // a latency-bound call (one or two gene-length queries, at most 16 items, ONE workgroup): every thread takes a run of
        // CONSECUTIVE words of a query -- all its loads in flight together, one scan per query instead of one per 256 words
        // (7.5 -> ~4 us for one query on 100 k samples)
        uint64_t base = 0;
        BIGSI_PHASE_AT(1000, 0);
        // (the block's header numbers: fetched now, beside the words, not in a round trip of their own behind the scans.  What remains of
        // this route's tail is the system-scope release before the flag: 1.5 us by the stamps -- the posted writes' acknowledgement)
        uint32_t hdr = 0;
        const bool hdr_early = 3u * n_seqs <= kBlock;                                 // (a handful of queries: always)
        if (exp_out && hdr_early && threadIdx.x < 3u * n_seqs) hdr = exp_uniq[threadIdx.x];
        for (uint32_t q = 0; q < n_seqs; q++) {
            const uint32_t w_first = threadIdx.x * chunks;
            uint64_t bits[16];
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t w = w_first + j;
                bits[j] = ((uint32_t)j < chunks && w < wv) ? bitmaps[(uint64_t)q * stride_words + w] : 0ull;
                cnt += (uint32_t)__popcll(bits[j]);
            }
            uint32_t tot;
            BIGSI_PHASE_AT(1000, 1);
            const uint32_t pre = block_exclusive_scan(cnt, &tot, lds);
            BIGSI_PHASE_AT(1000, 2);
            if (threadIdx.x == 0) {
                hit_off[q] = base;
                if (exp_out) exp_out[q] = base;
            }
            uint64_t o = base + pre;
            base += tot;
            if (cnt == 0 || o + cnt > capacity) continue;
            const uint32_t uq = num_unique[q];
            uint32_t *ocol = exp_out ? reinterpret_cast<uint32_t *>(exp_out + n_seqs + 2u) + ((3u * n_seqs + 1u) & ~1u) : nullptr;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                uint64_t mcol = by_column(bits[j]);
                const uint64_t col0 = (uint64_t)(w_first + j) * 64, cnt0 = (uint64_t)q * counter_stride + col0;
                while (mcol) {
                    const uint32_t c = (uint32_t)__builtin_ctzll(mcol);
                    mcol &= mcol - 1;
                    const uint32_t found = !counters ? uq
                                           : counter_bytes == 2 ? (uint32_t) reinterpret_cast<const uint16_t *>(counters)[cnt0 + c]
                                                                : reinterpret_cast<const uint32_t *>(counters)[cnt0 + c];
                    hit_col[o] = (uint32_t)(col0 + c);
                    hit_cnt[o] = found;
                    if (ocol && o < exp_spec) { ocol[o] = (uint32_t)(col0 + c); ocol[exp_spec + o] = found; }
                    o++;
                }
            }
        }
        BIGSI_PHASE_AT(1000, 3);
        if (threadIdx.x == 0) hit_off[n_seqs] = base;
        if (exp_out) {
            if (threadIdx.x == 0) { exp_out[n_seqs] = base; exp_out[n_seqs + 1] = 0; }
            uint32_t *o32 = reinterpret_cast<uint32_t *>(exp_out + n_seqs + 2u);
            if (!hdr_early)
                for (uint32_t i = threadIdx.x; i < 3u * n_seqs; i += kBlock) o32[i] = exp_uniq[i];
            else if (threadIdx.x < 3u * n_seqs) o32[threadIdx.x] = hdr;
            BIGSI_PHASE_AT(1000, 4);
            __threadfence_system();
            BIGSI_PHASE_AT(1000, 5);
            __syncthreads();
            if (threadIdx.x == 0) *exp_flag = exp_serial;
            BIGSI_PHASE_AT(1000, 6);
        }
        return;
    }
    const uint64_t grp = blockIdx.x, n_items = (uint64_t)n_seqs * n_shards * chunks;
    const uint64_t i0 = grp * ipb, i1 = i0 + ipb < n_items ? i0 + ipb : n_items;
    // totals of all preceding groups
    uint64_t part = 0;
    for (uint64_t j = threadIdx.x; j < grp; j += kBlock) part += totals[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if ((threadIdx.x & 63u) == 0) lds64[threadIdx.x >> 6] = part;
    __syncthreads();
    uint64_t base = 0;
#pragma unroll
    for (int i = 0; i < kBlock / 64; i++) base += lds64[i];
    // ordered write (the last group ends up with the grand total in `base`: a grid of ONE group needs no k_hits_totals at all)
    for (uint64_t ci = i0; ci < i1; ci++) {
        uint32_t w;
        const HitItem it = hit_item(ci, n_shards, chunks);
        const uint64_t bits = hits_word(bitmaps, stride_words, wv, n_seqs, it, &w);
        const uint32_t cnt = (uint32_t)__popcll(bits);
        uint32_t tot;
        const uint32_t pre = block_exclusive_scan(cnt, &tot, lds);
        const uint32_t chunk = it.chunk, shard = it.shard, q = it.q;
        if (threadIdx.x == 0 && chunk == 0 && shard == 0) hit_off[q] = base;
        uint64_t o = base + pre;
        base += tot;
        if (cnt == 0 || o + cnt > capacity) continue;     // overflow: the host sees total > capacity, grows the lists, runs this again
        const uint32_t uq = num_unique[q];
        const uint64_t cbase = (uint64_t)shard * shard_cols + (uint64_t)w * 64;
        const bool owned = own_shard == kAllShards || own_shard == shard;
        const uint64_t cnt0 = (own_shard == kAllShards ? (uint64_t)shard * n_seqs + q : (uint64_t)q) * counter_stride + (uint64_t)w * 64;
        for (uint32_t c = 0; c < 64; c++)
            if ((bits >> bit_of_col(c)) & 1ull) {
                hit_col[o] = (uint32_t)(cbase + c);
                hit_cnt[o] = !counters ? uq
                             : !owned ? 0u
                             : counter_bytes == 2 ? (uint32_t) reinterpret_cast<const uint16_t *>(counters)[cnt0 + c]
                                                  : reinterpret_cast<const uint32_t *>(counters)[cnt0 + c];
                o++;
            }
    }
    if (threadIdx.x == 0 && i1 == n_items) hit_off[n_seqs] = base;
}

// ------------------------------------------------------------------------------ reads: K1 + K2 + K4 in ONE launch
// A batch of reads against a narrow index (BASELINE configs[1]: 1000 x 61-mers, 10 000 samples) is three short kernels
// and three launch boundaries: 7 + 24 + 7 us of kernels in a 43 us step.  When every query has < 64 k-mer positions and a row
// fits one workgroup's lanes (<= 512 words), ONE workgroup per query does the whole path: two wavefronts k-merise / dedupe /
// hash (and leave the same arrays in global memory as the K1 kernels, for lookup / presence / fetch_rows), the row ids go to
// the other wavefronts through LDS, all of them stream and AND (or count) the rows, and the workgroup writes its query's hits
// to entries of the hit buffers it allocates with one atomic add (round 4: no order between queries, no waiting; see the
// kernel's last section).  Same results, one launch.
constexpr uint32_t kReadsMaxSeqs = 1u << 20;  // queries per launch of k_reads_fused at most
template <int H, bool EXACT, int VEC = kVec /* 64-column words per lane: 2 (16-byte loads) or 1 (8-byte loads; rows of <= kBlock words) */,
          int UNR_EXACT = 16 /* row loads a lane keeps in flight on the exact route */,
          bool SPLIT = false /* exact route, rows of <= 256 words, a handful of queries (a latency-bound call): the two halves of the
                                workgroup stream half of the query's rows each and meet in LDS -- half the chain of dependent round trips */>
__global__ __launch_bounds__(kBlock) void k_reads_fused(
    const uint64_t *__restrict__ index, uint64_t stride_words, uint32_t wv, uint64_t n_cols, uint64_t m, double threshold,
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ pos_off, uint32_t n_seqs,
    uint32_t *__restrict__ first_pos, uint32_t *__restrict__ pos_unique, uint32_t *__restrict__ rep_out, uint64_t *__restrict__ rows,
    uint32_t *__restrict__ num_kmers, uint32_t *__restrict__ num_unique, uint32_t *__restrict__ min_kmers,
    uint64_t *__restrict__ out_bits, uint64_t out_stride_words,
    uint64_t *__restrict__ q_start, uint32_t *__restrict__ q_cnt /* per query: where its hits lie in hit_col / hit_cnt, how many */,
    unsigned long long *__restrict__ alloc /* two words: hits allocated so far by this launch [slot] / by the launch before [slot ^ 1] */,
    uint32_t slot, uint32_t *__restrict__ hit_col, uint32_t *__restrict__ hit_cnt,
    uint64_t capacity, uint32_t fp_mask /* ~0; 1 = BIGSI_RUN_WEAK_FINGERPRINT */,
    uint64_t *__restrict__ pos_off_out /* as k_kmerize_lds: inputs come straight from pinned host memory */,
    uint32_t one_len /* as k_kmerize_lds */,
    uint64_t *exp_out /* non-null: a one-call search of ONE read -- the workgroup also writes the caller's block in pinned memory
                         (k_export_reads' layout) and raises the flag the host spins on: no export kernel, no launch boundary */,
    uint32_t exp_spec, volatile uint64_t *exp_flag, uint64_t exp_serial,
    const SeqArg<kSeqArgRead> sarg /* seqs == nullptr (one_len > 0): the one read of the call, passed in the kernel arguments */)
{
    constexpr int KF = 31, P = 6;
    if (pos_off_out && threadIdx.x == 0) {
        if (one_len) {
            pos_off_out[0] = 0;
            pos_off_out[1] = one_len >= (uint32_t)KF ? one_len - KF + 1 : 0u;
        } else {
            pos_off_out[blockIdx.x] = pos_off[blockIdx.x];
            if (blockIdx.x + 1 == n_seqs) pos_off_out[n_seqs] = pos_off[n_seqs];
        }
    }
    __shared__ uint64_t s_rows[64 * H], s_hrow[64 * H];   // row ids: of the unique k-mers / of every position, per seed
    __shared__ uint32_t s_seq[24], s_cmp[25];             // the query's bytes (63 positions + 30 = 93 at most); their complements
    __shared__ uint8_t s_first[64];                       // position of the j-th unique k-mer
    __shared__ uint32_t s_u, s_min;
    __shared__ uint32_t lds[16];
    __shared__ uint64_t lds64[kBlock / 64];
    const uint32_t q = blockIdx.x;
    BIGSI_PHASE(0);
    // ---- K1 on two wavefronts, on packed words.  The query's bytes (<= 93) and their complements are staged in LDS once.
    // Wavefront A finds each k-mer's first occurrence (k_kmerize_wave's scheme with a scalar fast path); wavefront B builds
    // the canonical form of EVERY position (forward words against byte-reversed complement words, compared as big-endian
    // numbers = lexicographically, utils/fncts.py:51-54) and hashes it for all H seeds, sharing the seed-independent part of
    // MurmurHash3.  The rows of the first occurrences are compacted afterwards.  A and B rotate with the query number so that
    // the workgroups sharing a CU load different SIMDs.  (One wavefront working byte by byte was 6.3 us of a 32 us kernel,
    // with the HBM idle: about 4 x 1150 VALU instructions per CU, most of them on one SIMD.)
    {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const uint32_t wave_a = (q >> 8) & 3u, wave_b = (wave_a + 2u) & 3u;
        const char *s = one_len ? seqs : seqs + seq_off[q];
        const uint32_t len = one_len ? one_len : (uint32_t)(seq_off[q + 1] - seq_off[q]);
        const uint32_t n = len >= KF ? len - KF + 1 : 0u;      // < 64 by the launch condition
        const uint64_t P0 = one_len ? 0ull : pos_off[q];
        if (threadIdx.x < 100) {                           // s_cmp carries 4 pad bytes in front (the word at byte p - 1 is read)
            const uint32_t t = threadIdx.x;
            uint8_t c = 0;
            if (t < len) c = seqs ? (uint8_t)s[t] : (uint8_t)(sarg.w[t >> 2] >> ((t & 3u) * 8u));
            if (t < 96) {
                reinterpret_cast<uint8_t *>(s_seq)[t] = c;
                reinterpret_cast<uint8_t *>(s_cmp)[4 + t] = complement(c);
            } else {
                reinterpret_cast<uint8_t *>(s_cmp)[t - 96] = 0;
            }
        }
        __syncthreads();
        BIGSI_PHASE(4);
        const bool live = lane < n;
        uint32_t wf[8];                                    // the k-mer at position `lane` (kmer31_words)
        if (wave == wave_a || wave == wave_b) kmer31_words(s_seq, lane, wf);
        if (wave == wave_a) {
            const uint32_t fp = live ? kmer31_fingerprint(wf) & fp_mask : 0u;
            BIGSI_PHASE(5);
            // rep = the first position holding this lane's k-mer.  Branch-free pass: the lowest lane with the same fingerprint
            // (independent v_readlane / compare / select triples, highest lane first so that the lowest match is kept; a serial
            // loop with a scalar early-out ran at 70 ns per position, all dependency stalls), then one word-by-word check
            // against that lane.  A fingerprint collision between different k-mers (2^-32 per pair) sends the wavefront
            // through the exact pairwise loop instead.
            uint32_t cand = lane;
            for (int jb = 48; jb >= 0; jb -= 16) {
                if ((uint32_t)jb >= n) continue;
#pragma unroll
                for (int t = 15; t >= 0; t--) {
                    const uint32_t fj = (uint32_t)__builtin_amdgcn_readlane((int)fp, jb + t);
                    cand = fp == fj ? (uint32_t)(jb + t) : cand;
                }
            }
            cand = min(cand, lane);                        // (lanes at or beyond n are not live and never representatives)
            const bool dup = live && cand < lane;
            bool same = true;
#pragma unroll
            for (int i = 0; i < 8; i++) same = same && (uint32_t)__shfl((int)wf[i], (int)cand, 64) == wf[i];
            uint32_t rep = dup ? cand : lane;
            if (__ballot(dup && !same) != 0ull) {          // wave-uniform, practically never
                rep = lane;
                for (uint32_t j = 0; j + 1 < n; j++) {
                    bool eq = live && lane > j && rep == lane;
#pragma unroll
                    for (int i = 0; i < 8; i++) eq = eq && wf[i] == (uint32_t)__builtin_amdgcn_readlane((int)wf[i], (int)j);
                    if (eq) rep = j;
                }
            }
            BIGSI_PHASE(6);
            const bool first = live && rep == lane;
            const unsigned long long mask = __ballot(first);
            const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            const uint32_t u = (uint32_t)__popcll(mask);
            if (live) {
                rep_out[P0 + lane] = rep;
                pos_unique[P0 + lane] = (uint32_t)__popcll(mask & (rep ? (~0ull >> (64 - rep)) : 0ull));
            }
            if (first) {
                const uint32_t j = (uint32_t)__popcll(mask & below);
                first_pos[P0 + j] = lane;
                s_first[j] = (uint8_t)lane;
            }
            if (lane == 0) {
                const double mk = ceil((double)u * threshold);
                const uint32_t mn = mk > 0.0 ? (uint32_t)mk : 0u;
                num_kmers[q] = n;
                num_unique[q] = u;
                min_kmers[q] = mn;
                s_u = u;
                s_min = mn;
            }
            BIGSI_PHASE(7);
        }
        if (wave == wave_b) {
            uint32_t k1[8];
            kmer31_canonical_premix(wf, s_cmp, lane, k1);
#pragma unroll
            for (int sd = 0; sd < H; sd++)
                if (live) s_hrow[lane * H + sd] = row_of_hash(murmur3_31_finish(k1, (uint32_t)sd), m);
        }
        __syncthreads();
        if (threadIdx.x < s_u * H) {                      // rows of the unique k-mers, first-occurrence order (u * H <= 252)
            const uint32_t j = threadIdx.x / H, t = threadIdx.x - j * H;
            const uint64_t r = s_hrow[(uint32_t)s_first[j] * H + t];
            s_rows[threadIdx.x] = r;
            rows[P0 * H + threadIdx.x] = r;
        }
    }
    __syncthreads();
    // ---- K2: lane = two words of the row.  (Splitting a query's rows over groups of lanes -- 3 x 79 lanes for the 157-word
    // rows of configs[1] -- so that three times the bytes are in flight measured +-0: with ~1000 workgroups resident the phase
    // already moves 5.7 TB/s of 1.25 KB rows, and the HBM is the limit, not the round trips.)
    const uint32_t u = s_u;
    BIGSI_PHASE(1);
    const uint32_t w0 = (SPLIT ? (threadIdx.x & (kBlock / 2 - 1)) : threadIdx.x) * VEC;
    const bool live = (!SPLIT || threadIdx.x < kBlock / 2) && w0 < wv;      // the lanes that own the query's words from here on
    uint64_t hitw[VEC];
    uint64_t pl[VEC][P];
#pragma unroll
    for (int v = 0; v < VEC; v++) hitw[v] = 0ull;
    if (EXACT) {
        constexpr int UNR = UNR_EXACT;
        const uint32_t R = u * H;
        RowWords<VEC> acc = RowWords<VEC>::fill(~0ull);
        if (SPLIT) {
            __shared__ RowWords<VEC> s_half[kBlock / 2];
            const uint32_t upper = threadIdx.x / (kBlock / 2), Rh = (R + 1) / 2, ra = upper ? Rh : 0u, rb = upper ? R : Rh;
            if (w0 < wv) {
                for (uint32_t r = ra; r < rb; r += UNR) {
                    RowWords<VEC> v[UNR];
#pragma unroll
                    for (int j = 0; j < UNR; j++) v[j] = r + j < rb ? RowWords<VEC>::load(index, s_rows[r + j], stride_words, w0) : RowWords<VEC>::fill(~0ull);
#pragma unroll
                    for (int j = 0; j < UNR; j++) acc &= v[j];
                }
            }
            if (upper) s_half[threadIdx.x - kBlock / 2] = acc;
            __syncthreads();
            if (live) acc &= s_half[threadIdx.x];
        }
        if (live) {
            for (uint32_t r = 0; !SPLIT && r < R; r += UNR) {
                RowWords<VEC> v[UNR];
#pragma unroll
                for (int j = 0; j < UNR; j++) v[j] = r + j < R ? RowWords<VEC>::load(index, s_rows[r + j], stride_words, w0) : RowWords<VEC>::fill(~0ull);
#pragma unroll
                for (int j = 0; j < UNR; j++) acc &= v[j];
            }
            if (R == 0) acc = RowWords<VEC>::fill(0ull);
#pragma unroll
            for (int v = 0; v < VEC; v++) hitw[v] = acc.word(v) & valid_mask((uint64_t)w0 + v, n_cols);
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; v++)
#pragma unroll
            for (int p = 0; p < P; p++) pl[v][p] = 0;
        // SPLIT: the upper half of the workgroup counts the k-mers [uh, u), the lower half [0, uh); the partial counts meet in LDS
        const uint32_t upper_c = SPLIT ? threadIdx.x / (kBlock / 2) : 0u, uh = SPLIT ? (u + 1) / 2 : u;
        const uint32_t ja = upper_c ? uh : 0u, jb = SPLIT ? (upper_c ? u : uh) : u;
        if (SPLIT ? w0 < wv : live) {
            constexpr int KM = H <= 2 ? 8 : H == 3 ? 6 : 4;
            for (uint32_t j = ja; j < jb; j += KM) {
                RowWords<VEC> v[KM * H];
#pragma unroll
                for (int t = 0; t < KM * H; t++)
                    v[t] = j + t / H < jb ? RowWords<VEC>::load(index, s_rows[(j + t / H) * H + t % H], stride_words, w0) : RowWords<VEC>::fill(0ull);
#pragma unroll
                for (int g = 0; g < KM; g++) {
                    RowWords<VEC> a = v[g * H];
#pragma unroll
                    for (int t = 1; t < H; t++) a &= v[g * H + t];
#pragma unroll
                    for (int e = 0; e < VEC; e++) {
                        uint64_t c = a.word(e);
#pragma unroll
                        for (int p = 0; p < P; p++) {
                            const uint64_t t0 = pl[e][p] & c;
                            pl[e][p] ^= c;
                            c = t0;
                        }
                    }
                }
            }
        }
        if (SPLIT) {
            __shared__ uint64_t s_pl[P][VEC][kBlock / 2];
            if (upper_c) {
#pragma unroll
                for (int p = 0; p < P; p++)
#pragma unroll
                    for (int e = 0; e < VEC; e++) s_pl[p][e][threadIdx.x - kBlock / 2] = pl[e][p];
            }
            __syncthreads();
            if (live) {          // bit-sliced ripple-carry add of the two partial counts (their sum is at most u < 2^P)
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    uint64_t carry = 0;
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const uint64_t a = pl[e][p], b2 = s_pl[p][e][threadIdx.x];
                        pl[e][p] = a ^ b2 ^ carry;
                        carry = (a & b2) | (carry & (a ^ b2));
                    }
                }
            }
        }
        if (live) {
            const uint32_t thr = s_min;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                uint64_t gt = 0, eq = ~0ull;
                if ((thr >> P) != 0) eq = 0;
#pragma unroll
                for (int p = P - 1; p >= 0; p--) {
                    if ((thr >> p) & 1u) eq &= pl[v][p];
                    else { gt |= eq & pl[v][p]; eq &= ~pl[v][p]; }
                }
                hitw[v] = (gt | eq) & valid_mask((uint64_t)w0 + v, n_cols);
            }
        }
    }
    BIGSI_PHASE(2);
    if (live) {
        uint64_t *o = out_bits + (uint64_t)q * out_stride_words + w0;
#pragma unroll
        for (int v = 0; v < VEC; v++)
            if (w0 + v < out_stride_words) o[v] = hitw[v];
    }
    // ---- K4: this query's hits, at a place of their own.  The workgroup takes `tot` consecutive entries of the batch's hit buffers
    // with ONE atomic add and leaves (start, count) for its query; the lists of different queries lie in whatever order the
    // workgroups got there, and whoever reads them -- the export kernel of the one-call / streaming searches, the host for
    // fetch_hits -- puts them in query order from the counts.  Nobody waits for anybody: rounds 2 and 3 wrote the lists in query
    // order through a publish-and-sum scan whose workgroups spun on the totals of their predecessors (bounded by a timeout and
    // a host-side repeat, but leaning on dispatch order for progress, and 3 us of every 29 us launch).
    uint32_t mine = 0;
#pragma unroll
    for (int v = 0; v < VEC; v++) mine += (uint32_t)__popcll(hitw[v]);
    uint32_t tot;
    const uint32_t pre = block_exclusive_scan(mine, &tot, lds);
    if (threadIdx.x == 0) {
        const uint64_t start = tot ? (uint64_t)atomicAdd(alloc + slot, (unsigned long long)tot) : 0ull;
        lds64[0] = start;
        q_start[q] = start;
        q_cnt[q] = tot;
        if (q == 0) alloc[slot ^ 1u] = 0;          // the next launch of this batch (launches of one batch never overlap) starts from zero
    }
    __syncthreads();
    BIGSI_PHASE(3);
    const uint64_t start = lds64[0];
    if (mine != 0 && start + tot <= capacity) {      // (else the host sees total > capacity, grows the lists and launches again)
        uint64_t o = start + pre;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            uint64_t mcol = by_column(hitw[v]);
            while (mcol) {
                const uint32_t c = (uint32_t)__builtin_ctzll(mcol);
                mcol &= mcol - 1;
                const uint32_t colour = (uint32_t)(((uint64_t)w0 + v) * 64 + c);
                uint32_t x = u;
                if (!EXACT) {
                    const uint32_t bp = bit_of_col(c);
                    x = 0;
#pragma unroll
                    for (int p = 0; p < P; p++) x |= (uint32_t)((pl[v][p] >> bp) & 1ull) << p;
                }
                hit_col[o] = colour;
                hit_cnt[o] = x;
                if (exp_out && o < exp_spec) {             // (one query: its list starts at 0)
                    uint32_t *ocol = reinterpret_cast<uint32_t *>(exp_out + 3) + 4;      // offsets (2) | 0 | counts (3, padded to 4) | colours | counts
                    ocol[o] = colour;
                    ocol[exp_spec + o] = x;
                }
                o++;
            }
        }
    }
    if (!exp_out) return;
    if (threadIdx.x == 0) {
        exp_out[0] = 0;
        exp_out[1] = tot;
        exp_out[2] = 0;
        uint32_t *o32 = reinterpret_cast<uint32_t *>(exp_out + 3);
        const uint32_t len = one_len ? one_len : (uint32_t)(seq_off[1] - seq_off[0]);
        o32[0] = len >= (uint32_t)KF ? len - KF + 1 : 0u;
        o32[1] = s_u;
        o32[2] = s_min;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *exp_flag = exp_serial;
}

template <typename CountT, bool WRITE>
__global__ __launch_bounds__(kBlock) void k_hits_count(
    const CountT *__restrict__ counts, uint64_t stride, uint32_t /*wv: unused, keeps both K4 signatures alike*/,
    uint32_t n_seqs, uint32_t n_shards, uint32_t chunks,
    uint64_t shard_cols, const uint32_t *__restrict__ min_kmers,
    uint32_t *__restrict__ chunk_hits, const uint64_t *__restrict__ chunk_off,
    uint32_t *__restrict__ hit_col, uint32_t *__restrict__ hit_cnt, uint64_t capacity, uint32_t *__restrict__ overflow)
{
    __shared__ uint32_t lds[16];
    const uint32_t chunk = blockIdx.x % chunks, sq = blockIdx.x / chunks;
    const uint32_t shard = sq % n_shards, q = sq / n_shards;
    const uint64_t c0 = (uint64_t)chunk * kChunkCols + threadIdx.x * 8u;
    const uint32_t thr = min_kmers[q];
    uint32_t c[8];
    uint32_t mine = 0;
    const CountT *src = counts + ((uint64_t)shard * n_seqs + q) * stride + c0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c[j] = (c0 + j < shard_cols) ? (uint32_t)src[j] : 0u;
        if (c0 + j < shard_cols && c[j] >= thr) mine++;
    }
    uint32_t tot;
    const uint32_t pre = block_exclusive_scan(mine, &tot, lds);
    const uint64_t ci = chunk_index(q, shard, chunk, n_shards, chunks);
    if (!WRITE) {
        if (threadIdx.x == 0) chunk_hits[ci] = tot;
        return;
    }
    if (mine == 0) return;
    uint64_t o = chunk_off[ci] + pre;
    if (o + mine > capacity) { *overflow = 1; return; }
    const uint64_t cbase = (uint64_t)shard * shard_cols + c0;
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (c0 + j < shard_cols && c[j] >= thr) { hit_col[o] = (uint32_t)(cbase + j); hit_cnt[o] = c[j]; o++; }
}

// exclusive scan of chunk_hits (n entries) by one workgroup, kScanItems consecutive entries per thread per round;
// also hit_off[q] for every sequence and the total.
constexpr int kScanItems = 16;
__global__ __launch_bounds__(kBlock) void k_scan_chunks(
    const uint32_t *__restrict__ chunk_hits, uint64_t n, uint32_t per_seq, uint32_t n_seqs,
    uint64_t *__restrict__ chunk_off, uint64_t *__restrict__ hit_off)
{
    __shared__ uint32_t lds[16];
    uint64_t carry = 0;
    for (uint64_t base = 0; base < n; base += (uint64_t)kBlock * kScanItems) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * kScanItems;
        uint32_t v[kScanItems], sum = 0;
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            v[j] = i0 + j < n ? chunk_hits[i0 + j] : 0u;
            sum += v[j];
        }
        uint32_t tot;
        uint64_t run = carry + block_exclusive_scan(sum, &tot, lds);
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            const uint64_t i = i0 + j;
            if (i < n) {
                chunk_off[i] = run;
                if (i % per_seq == 0) hit_off[i / per_seq] = run;
            }
            run += v[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) hit_off[n_seqs] = carry;
}

// ------------------------------------------------------------------------------ lookup (API parity)
// KmerSignatureIndex.lookup for one sequence (graph/index.py:42-49): out[j][w] = AND of the h rows of unique k-mer j.
__global__ __launch_bounds__(kBlock) void k_lookup(
    const uint64_t *__restrict__ index, uint64_t stride_words, uint32_t wv, const uint64_t *__restrict__ qrows,
    uint32_t h, uint32_t u, uint64_t *__restrict__ out)
{
    const uint32_t wblocks = (wv + kBlock - 1) / kBlock;
    const uint32_t j = blockIdx.x / wblocks;
    const uint32_t w = (blockIdx.x % wblocks) * kBlock + threadIdx.x;
    if (j >= u || w >= wv) return;
    uint64_t a = ~0ull;
    for (uint32_t s = 0; s < h; s++) a &= index[qrows[(uint64_t)j * h + s] * stride_words + w];
    out[(uint64_t)j * wv + w] = a;
}

// ------------------------------------------------------------------------------ K5: presence strings
// graph/bigsi.py:232-237: for hit colour c and every k-mer position i (duplicates included), '1' iff every one of the
// k-mer's h rows has bit c set.  out[hit][i] ASCII.
__global__ __launch_bounds__(kBlock) void k_presence(
    const uint64_t *__restrict__ index, uint64_t stride_words, const uint64_t *__restrict__ qrows,
    const uint32_t *__restrict__ pos_unique, uint32_t h, uint32_t n, const uint32_t *__restrict__ colours, uint32_t n_colours,
    uint8_t *__restrict__ out)
{
    const uint32_t pblocks = (n + kBlock - 1) / kBlock;
    const uint32_t hit = blockIdx.x / pblocks;
    const uint32_t i = (blockIdx.x % pblocks) * kBlock + threadIdx.x;
    if (i >= n || hit >= n_colours) return;
    const uint32_t c = colours[hit];
    const uint64_t w = c >> 6, j = pos_unique[i];
    uint64_t a = ~0ull;
    for (uint32_t s = 0; s < h; s++) a &= index[qrows[j * h + s] * stride_words + w];
    out[(uint64_t)hit * n + i] = ((a >> bit_of_col(c & 63u)) & 1ull) ? '1' : '0';
}

// ------------------------------------------------------------------------------ K5 at scale: all hits of a batch in one pass
// BIGSI.score (graph/bigsi.py:232-237) needs, for every hit (sequence q, colour c), the n-character string whose i-th
// character says whether the k-mer at position i is present in sample c.  k_presence above spends h dependent 8-byte loads
// per (hit, position); with thousands of hits per query (threshold 0.4 on a 500k-sample index) that is the wrong unit of
// work.  Here the unit is (unique k-mer, 128-column PAIR OF WORDS that contains hits): the h rows are AND-ed once per pair
// with 16-byte loads, coalesced over the consecutive hit pairs of a query, and ALL hits of the pair take their bit from
// that one AND.  A thread keeps the AND-ed words of 16 unique k-mers in registers and emits, per hit, the 16 presence bits
// as one uint16 (bits[hit][k-mer / 16]); k_presence_expand then writes the ASCII strings over the n positions
// (duplicates included) from those bits.  Bytes: u x h x 16 per hit pair -- the K2 stream restricted to the hit words.
// presence bits of hit number `rank` (rank among the call's hits, each sequence's hits in colour order), unique k-mers
// 16 * chunk .. +15: tiles of 32 ranks, [tile][chunk][rank % 32].  k_presence_bits then stores runs of consecutive ranks
// next to each other (hit-major rows cost it a third of its time in isolated 2-byte stores: 373 vs 251 us without stores,
// 319 us with this layout), and k_presence_expand reads the chunks of 4 consecutive ranks as one 8-byte word.
__device__ __forceinline__ uint64_t presence_bits_at(uint64_t rank, uint32_t chunk, uint32_t n_chunks)
{
    return ((rank >> 5) * n_chunks + chunk) * 32u + (rank & 31u);
}

struct PresencePair {
    uint32_t wpair;          // word pair: columns [128 * wpair, +128)
    uint32_t base;           // rank, among the query's hits sorted by colour, of the pair's first hit (global index into perm)
    uint64_t mask_lo, mask_hi;   // hit bits of the two words (row bit order)
    uint32_t q;              // the query the pair belongs to
    uint32_t reserved;
};
// one wavefront of k_presence_bits: `count` (<= 64) consecutive pairs of ONE query, from pairs[first] on
struct PresenceWave {
    uint32_t first, count;
};


// The grid is flat over wavefronts of pairs (x; a wavefront takes up to 64 pairs of one query: PresenceWave) and 16-k-mer chunks (y): a thresholded search of a few hundred queries of which a
// dozen have hits -- BASELINE configs[4] as benchmarked -- used to launch (pairs of the fullest query / 256) x chunks x queries
// workgroups, nearly all of them empty, and waited for slots beside the next batch's row-AND kernel (166 us against 26 us alone).
template <int H, int WAVES = 2>      // WAVES = 2: the compiler keeps all 16 x h loads of a thread in flight (~200 VGPRs), measured faster than 4
__global__ __launch_bounds__(kBlock, WAVES) void k_presence_bits(
    const uint64_t *__restrict__ index, uint64_t stride_words, const uint64_t *__restrict__ rows, const uint64_t *__restrict__ pos_off,
    const uint32_t *__restrict__ num_unique, uint32_t h_rt, uint32_t n_waves, const PresenceWave *__restrict__ waves,
    const PresencePair *__restrict__ pairs, uint16_t *__restrict__ bits /* presence_bits_at(rank, chunk) */, uint32_t bits_stride)
{
    const uint32_t h = H > 0 ? (uint32_t)H : h_rt;
    const uint32_t jc = blockIdx.y;
    // a wavefront's pairs all belong to one query (the host cuts the pair list that way: round 3 padded every query's pairs to 64
    // entries instead -- 2 KB of upload per query with hits, nearly all of it padding for reads), so the query -- and with it the
    // k-mer count and every row id of the chunk -- is the same for all 64 lanes: scalar loads (per-lane row ids cost this kernel
    // 70 % at 261 k hits)
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));
    if (w >= n_waves) return;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)waves[w].first);
    const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)waves[w].count);
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)pairs[first].q);
    const uint32_t u = num_unique[q];
    const uint32_t j0 = jc * 16u;
    if (j0 >= u) return;
    if ((threadIdx.x & 63u) >= count) return;
    const PresencePair pr = pairs[first + (threadIdx.x & 63u)];
    const uint64_t *qrows = rows + pos_off[q] * h;
    const uint32_t woff = pr.wpair * 2u;
    const u64x2 zero = {0ull, 0ull};
    u64x2 a[16];
    if (H > 0) {
        // four k-mers = 4 x h independent 16-byte loads in flight, then their ANDs
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 4) {
            u64x2 tmp[4 * (H > 0 ? H : 1)];
            constexpr int HH = H > 0 ? H : 1;
#pragma unroll
            for (int x = 0; x < 4 * H; x++) {
                const uint32_t j = j0 + t0 + x / HH;
                tmp[x] = j < u ? load_row_seg(index, qrows[(uint64_t)j * H + x % HH], stride_words, woff) : zero;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                u64x2 v = tmp[t * H];
#pragma unroll
                for (int sidx = 1; sidx < H; sidx++) v &= tmp[t * H + sidx];
                a[t0 + t] = v;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const uint32_t j = j0 + t;
            u64x2 v = zero;
            if (j < u) {
                v = load_row_seg(index, qrows[(uint64_t)j * h], stride_words, woff);
                for (uint32_t sidx = 1; sidx < h; sidx++) v &= load_row_seg(index, qrows[(uint64_t)j * h + sidx], stride_words, woff);
            }
            a[t] = v;
        }
    }
    uint32_t rank = pr.base;
    // (layout: presence_bits_at)
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint64_t m = by_column(half ? pr.mask_hi : pr.mask_lo);      // bit c set <=> column 64 * word + c is a hit
        while (m) {
            const uint32_t c = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uint32_t bp = bit_of_col(c);
            uint32_t out = 0;
#pragma unroll
            for (int t = 0; t < 16; t++) out |= (uint32_t)(((half ? a[t].y : a[t].x) >> bp) & 1ull) << t;
            bits[presence_bits_at(rank, jc, bits_stride)] = (uint16_t)out;
            rank++;
        }
    }
}

// The same bits for queries with FEW pairs (a thresholded search whose queries have a hit or two each -- BASELINE configs[4]: 259 hits in
// 256 queries): there a wavefront of the kernel above has one live lane, and 259 hits cost 15 800 wavefronts of 48 loads each.
// Here lane = unique k-mer: a wavefront takes 64 consecutive k-mers of one query (their row ids: coalesced loads), and for every
// pair of the query ANDs the h rows' 16-byte word pair of its k-mer; a hit's 64 presence bits are one ballot, stored as four
// 16-bit chunks.  Same layout out (presence_bits_at), h loads per (k-mer, pair) as before, a sixteenth of the instructions.
template <int H>
__global__ __launch_bounds__(kBlock) void k_presence_bits_sparse(
    const uint64_t *__restrict__ index, uint64_t stride_words, const uint64_t *__restrict__ rows, const uint64_t *__restrict__ pos_off,
    const uint32_t *__restrict__ num_unique, uint32_t h_rt, const PresenceWave *__restrict__ queries /* one per query: all its pairs */,
    const PresencePair *__restrict__ pairs, uint16_t *__restrict__ bits, uint32_t bits_stride)
{
    constexpr int HH = H > 0 ? H : 1;
    const uint32_t h = H > 0 ? (uint32_t)H : h_rt;
    const uint32_t jb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.y * (kBlock / 64) + (threadIdx.x >> 6))), lane = threadIdx.x & 63u;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)queries[blockIdx.x].first);
    const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)queries[blockIdx.x].count);
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)pairs[first].q);
    const uint32_t u = num_unique[q];
    if (jb * 64u >= u) return;
    const uint32_t j = jb * 64u + lane;
    const bool live = j < u;
    const uint64_t *qrows = rows + pos_off[q] * h;
    uint64_t r[HH];
    if (H > 0) {
#pragma unroll
        for (int sidx = 0; sidx < HH; sidx++) r[sidx] = live ? qrows[(uint64_t)j * H + sidx] : 0ull;
    }
    const u64x2 zero = {0ull, 0ull};
    for (uint32_t p = 0; p < count; p++) {
        const uint32_t wpair = (uint32_t)__builtin_amdgcn_readfirstlane((int)pairs[first + p].wpair);
        uint32_t rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)pairs[first + p].base);
        const uint64_t mask_lo = pairs[first + p].mask_lo, mask_hi = pairs[first + p].mask_hi;
        const uint32_t woff = wpair * 2u;
        u64x2 v = zero;
        if (live) {
            if (H > 0) {
                v = load_row_seg(index, r[0], stride_words, woff);
#pragma unroll
                for (int sidx = 1; sidx < HH; sidx++) v &= load_row_seg(index, r[sidx], stride_words, woff);
            } else {
                v = load_row_seg(index, qrows[(uint64_t)j * h], stride_words, woff);
                for (uint32_t sidx = 1; sidx < h; sidx++) v &= load_row_seg(index, qrows[(uint64_t)j * h + sidx], stride_words, woff);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint64_t m = by_column(half ? mask_hi : mask_lo);
            const uint64_t word = half ? v.y : v.x;
            while (m) {
                const uint32_t c = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint64_t ball = __ballot(((word >> bit_of_col(c)) & 1ull) != 0);      // bit L: k-mer 64 * jb + L is present in this hit's sample
                const uint32_t chunk = 4u * jb + lane;
                if (lane < 4u && chunk * 16u < u) bits[presence_bits_at(rank, chunk, bits_stride)] = (uint16_t)(ball >> (16u * lane));
                rank++;
            }
        }
    }
}

// strings: 16 characters per thread, one 16-byte store (every string starts at a multiple of 16 bytes);
// character i of hit t = '0' + bit (unique k-mer of position i) of the hit's presence bits.  Thread -> (hit, 16-character
// piece), `pieces` (a power of two) pieces per hit, flattened over the grid.  A thread needs the position -> unique k-mer map
// of its 16 consecutive positions; read directly that is a 64-byte stride between lanes (one cache line per lane and load:
// measured 0.5 TB/s).  Instead the G = min(pieces, 64) lanes that share a hit fetch its G * 16 entries with 16 coalesced
// loads and pass them through LDS (pitch 20 dwords per lane: conflict-free 16-byte reads).
// Most 16-position pieces of a query hold 16 DISTINCT k-mers that are also new to the query, i.e. consecutive unique indices
// j0 .. j0+15 (always, unless the piece touches a repeat): the piece's characters are then 16 consecutive bits of the hit's
// presence bits.  k_presence_pieces marks those pieces once per call (bit 31 + j0) and lists the others; k_presence_expand
// turns the marked pieces of every hit into characters (two shuffled 16-bit chunks, four 24-bit multiplies: 4 bits -> 4
// bytes), k_presence_expand_listed walks the position -> unique k-mer map for the listed ones.  (The map for every piece ran
// at 1.1 TB/s of string bytes, VALU-bound; keeping it as a fallback inside the fast kernel held that kernel at 4 waves per
// SIMD, latency-bound at 2.6 TB/s.)
__global__ __launch_bounds__(kBlock) void k_presence_pieces(
    const uint32_t *__restrict__ pos_unique, const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ num_kmers, uint32_t *__restrict__ desc,
    uint32_t *__restrict__ listed_count, uint2 *__restrict__ listed)
{
    const uint32_t q = blockIdx.x, n = num_kmers[q];
    const uint64_t P0 = pos_off[q];
    const uint32_t *pu = pos_unique + P0;
    uint32_t *d = desc + (P0 >> 4) + q;                    // ceil(n / 16) entries per sequence, disjoint by construction
    for (uint32_t pc = threadIdx.x; pc * 16u < n; pc += kBlock) {
        const uint32_t i0 = pc * 16u, cnt = min(16u, n - i0), j0 = pu[i0];
        bool run = true;
        for (uint32_t t = 1; t < cnt; t++) run = run && pu[i0 + t] == j0 + t;
        d[pc] = (j0 & 0x7fffffffu) | (run ? 0x80000000u : 0u);
        if (!run) listed[atomicAdd(listed_count, 1u)] = uint2{q, pc};
    }
}

// the listed pieces (they touch a repeated k-mer): one workgroup per piece at a time, its threads over the hits of the
// piece's sequence; character t of the piece = bit pos_unique[i0 + t] of the hit's presence bits
__global__ __launch_bounds__(kBlock) void k_presence_expand_listed(
    const uint16_t *__restrict__ bits, uint32_t bits_stride, const uint32_t *__restrict__ listed_count, const uint2 *__restrict__ listed,
    const uint64_t *__restrict__ seq_hit_off /* per sequence: its first hit (the hits of a call are grouped by sequence) */,
    const uint64_t *__restrict__ pos_off, const uint32_t *__restrict__ num_kmers, const uint32_t *__restrict__ pos_unique,
    const uint64_t *__restrict__ str_off, uint8_t *__restrict__ out)
{
    const uint32_t count = *listed_count;
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
        const uint2 ent = listed[e];
        const uint32_t q = ent.x, i0 = ent.y * 16u, n = num_kmers[q], cnt = min(16u, n - i0);
        const uint32_t *pu = pos_unique + pos_off[q] + i0;
        uint32_t j[16];
#pragma unroll
        for (int t = 0; t < 16; t++) j[t] = (uint32_t)t < cnt ? pu[t] : 0u;
        for (uint64_t hit = seq_hit_off[q] + threadIdx.x; hit < seq_hit_off[q + 1]; hit += kBlock) {       // hit = rank
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint32_t ch = (uint32_t)t < cnt ? '0' + (((uint32_t)bits[presence_bits_at(hit, j[t] >> 4, bits_stride)] >> (j[t] & 15u)) & 1u) : 0u;
                w[t >> 2] |= ch << (8 * (t & 3));
            }
            *reinterpret_cast<uint4 *>(out + str_off[hit] + i0) = uint4{w[0], w[1], w[2], w[3]};
        }
    }
}

// The marked pieces.  A thread writes piece `piece` of kPresenceHits consecutive hits per round, two levels of loads each: the
// hit's record (scalar loads when a whole wavefront shares the hit, ONE_HIT), then the piece's mark and -- independent of it --
// chunk `piece` of the hit's presence bits, one coalesced 2-byte load per lane; the chunk a piece really starts in lies a few
// lanes to the left (unique index <= position) and comes over by a lane shuffle.
constexpr int kPresenceHits = 4, kPresenceRounds = 4;
template <bool ONE_HIT>
__global__ __launch_bounds__(kBlock) void k_presence_expand(
    const uint16_t *__restrict__ bits, uint32_t bits_stride, uint64_t n_hits, uint32_t pieces, const uint32_t *__restrict__ hit_n,
    const uint64_t *__restrict__ hit_pos0 /* per hit: where its sequence's position -> unique map starts */, const uint64_t *__restrict__ str_off,
    uint8_t *__restrict__ out, const uint32_t *__restrict__ hit_seq /* per hit: its sequence */, const uint32_t *__restrict__ desc /* k_presence_pieces */)
    // "hit" = rank throughout: the per-hit arrays arrive in rank order, str_off[rank] is where that hit's string goes
{
    const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t shift = 31u - (uint32_t)__builtin_clz(pieces);
    uint32_t group = (uint32_t)(idx >> shift);             // < 2^31 by the launch condition
    if (ONE_HIT) group = (uint32_t)__builtin_amdgcn_readfirstlane((int)group);      // pieces >= 64: the wavefront's lanes share it
    const uint32_t piece = (uint32_t)(idx & (pieces - 1u));
    const uint32_t lane = threadIdx.x & 63u, G = pieces < 64u ? pieces : 64u, sub = lane & (G - 1u);
#pragma unroll 1
    for (uint32_t round = 0; round < (uint32_t)kPresenceRounds; round++) {
        const uint64_t hit0 = ((uint64_t)group * kPresenceRounds + round) * kPresenceHits;
        if (hit0 >= n_hits) break;
        uint32_t n[kPresenceHits], d[kPresenceHits], cw[kPresenceHits];
        uint64_t so[kPresenceHits];
#pragma unroll
        for (int u = 0; u < kPresenceHits; u++) {
            const uint64_t hit = hit0 + u;
            const bool has = hit < n_hits;
            n[u] = has ? hit_n[hit] : 0u;
            so[u] = has ? str_off[hit] : 0ull;
            d[u] = has ? (uint32_t)(hit_pos0[hit] >> 4) + hit_seq[hit] + piece : 0u;       // where the piece's mark is
        }
        {   // chunk `piece` of the four ranks: four neighbouring 16-bit words of one tile (hit0 is a multiple of 4)
            const uint64_t four = *reinterpret_cast<const uint64_t *>(bits + presence_bits_at(hit0, min(piece, bits_stride - 1u), bits_stride));
#pragma unroll
            for (int u = 0; u < kPresenceHits; u++) {
                cw[u] = (uint32_t)(four >> (16 * u)) & 0xffffu;
                d[u] = piece * 16u < n[u] ? desc[d[u]] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < kPresenceHits; u++) {
            // a run of consecutive unique k-mers j0 .. : 16 consecutive presence bits, starting in chunk c <= piece
            const uint32_t j0 = d[u] & 0x7fffffffu, c = j0 >> 4, r = j0 & 15u, delta = piece - c;
            uint32_t lo = (uint32_t)__shfl((int)cw[u], (int)(lane - delta), 64);       // (every lane takes part in the shuffles)
            uint32_t hi = (uint32_t)__shfl((int)cw[u], (int)(lane - delta + 1u), 64);
            if (!(d[u] >> 31)) continue;                                             // no piece here, or a listed one
            if (delta > sub) lo = bits[presence_bits_at(hit0 + u, c, bits_stride)];         // the chunk belongs to a lane of another wavefront
            if (r && (delta > sub + 1u || (delta == 0u && sub + 1u >= G))) hi = bits[presence_bits_at(hit0 + u, min(c + 1u, bits_stride - 1u), bits_stride)];
            const uint32_t x16 = (lo | (hi << 16)) >> r, cnt = min(16u, n[u] - piece * 16u);
            uint32_t o[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                // 4 bits -> 4 bytes: bit i of x lands on bits i, i+7, i+14, i+21 of the product, of which 0, 8, 16, 24 are kept
                const uint32_t x = (x16 >> (4 * k4)) & 15u;
                const uint32_t ch = (__umul24(x, 0x204081u) & 0x01010101u) + 0x30303030u;
                const int valid = (int)cnt - 4 * k4;
                o[k4] = valid >= 4 ? ch : valid <= 0 ? 0u : ch & ((1u << (8 * valid)) - 1u);
            }
            *reinterpret_cast<uint4 *>(out + so[u] + piece * 16u) = uint4{o[0], o[1], o[2], o[3]};
        }
    }
}

// ------------------------------------------------------------------------------ K6: BIGSI.score on the device
// graph/bigsi.py:232-239 + scoring/score.py:7-107.  The reference turns every hit's column of the n x N matrix into a
// '0'/'1' string and scores it in Python.  Here a hit never becomes a string on the device: k_presence_score gathers the
// hit's presence bits into position order -- ONE BIT per k-mer position, in the reference's own bit order
// (bitarray.tobytes() of the presence string), 8x fewer bytes over PCIe than the ASCII strings of k_presence_expand -- and
// runs remove_short_ones / tabulate_score / calculate_score on those bits (bigsi_score.hpp: integer run
// scans + IEEE doubles with Python's round()).  One thread per hit throughout: a hit is ~n / 8 bytes and a few dozen runs,
// and the presence-bit tiles ([tile of 32 ranks][chunk][rank % 32], presence_bits_at) make the threads of a wavefront read
// consecutive 16-bit words.
//
// position-ordered presence bits of every hit AND its score record: thread = rank; piece = 16 positions; marked pieces
// (k_presence_pieces) are 16 consecutive presence bits of the hit, listed ones walk the position -> unique k-mer map (uniform
// over a sequence's hits).  The thread then scores the words it has just written (its own stores: visible to it).
constexpr int kScoreLdsWords = 16;          // presence words of a hit kept in LDS for the scoring passes (16 x 64 = 1024 positions)
__global__ __launch_bounds__(kBlock) void k_presence_score(
    const uint16_t *__restrict__ bits, uint32_t bits_stride, uint64_t n_hits, const uint32_t *__restrict__ hit_n,
    const uint64_t *__restrict__ hit_pos0, const uint32_t *__restrict__ hit_seq, const uint32_t *__restrict__ desc,
    const uint32_t *__restrict__ pos_unique, const uint64_t *__restrict__ out_off /* per rank: byte offset, multiple of 8 */,
    uint8_t *out, const uint32_t *__restrict__ found, const uint32_t *__restrict__ unique, const uint32_t *__restrict__ dest,
    bigsi_score::HitScore *__restrict__ scores)
{
    __shared__ uint64_t lw[kScoreLdsWords][kBlock];       // [word][thread]: conflict-free
    const uint64_t rank = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (rank >= n_hits) return;
    const uint32_t n = hit_n[rank], pieces = (n + 15u) >> 4;
    const uint64_t pos0 = hit_pos0[rank];
    const uint32_t *d = desc + (pos0 >> 4) + hit_seq[rank];
    const uint32_t *pu = pos_unique + pos0;
    const uint16_t *mine = bits + presence_bits_at(rank, 0, bits_stride);      // chunk c of this rank: mine[32 * c]
    uint64_t *dst = reinterpret_cast<uint64_t *>(out + out_off[rank]);
    // four pieces = one 64-position word per round: the four marks, then the (up to) eight 16-bit chunks they point at, as
    // independent loads -- two dependent levels per word instead of two per piece (a thread per hit is a chain of latencies)
    for (uint32_t p0 = 0; p0 < pieces; p0 += 4u) {
        uint32_t mark[4], lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; i++) mark[i] = p0 + i < pieces ? d[p0 + i] : 0x80000000u;      // (beyond the end: "marked", masked to nothing below)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t j0 = mark[i] & 0x7fffffffu, c = j0 >> 4, r = j0 & 15u;
            const bool fast = (mark[i] >> 31) && p0 + i < pieces;
            lo[i] = fast ? (uint32_t)mine[32u * c] : 0u;
            hi[i] = fast && r ? (uint32_t)mine[32u * min(c + 1u, bits_stride - 1u)] : 0u;
        }
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t piece = p0 + i;
            if (piece >= pieces) break;
            const uint32_t cnt = min(16u, n - piece * 16u);
            uint32_t x16;
            if (mark[i] >> 31) {
                x16 = (lo[i] | (hi[i] << 16)) >> (mark[i] & 15u);
            } else {                                    // a piece that touches a repeated k-mer: walk the position -> unique map
                x16 = 0;
                for (uint32_t t = 0; t < cnt; t++) {
                    const uint32_t j = pu[piece * 16u + t];
                    x16 |= (((uint32_t)mine[32u * (j >> 4)] >> (j & 15u)) & 1u) << t;
                }
            }
            x16 &= (1u << cnt) - 1u;
            acc |= (uint64_t)x16 << (16 * i);
        }
        const uint64_t packed = by_column(acc);         // position p -> byte p / 8, mask 0x80 >> (p % 8)
        dst[p0 >> 2] = packed;
        if ((p0 >> 2) < (uint32_t)kScoreLdsWords) lw[p0 >> 2][threadIdx.x] = acc;
    }
    bigsi_score::HitScore rec;
    const uint64_t *w = dst;
    const uint32_t tid = threadIdx.x;
    bigsi_score::score_hit([w, tid](uint32_t k) { return k < (uint32_t)kScoreLdsWords ? lw[k][tid] : by_column(w[k]); }, n, found[rank], unique[rank], &rec);
    scores[dest[rank]] = rec;
}

// scores of every hit from its packed presence bits (row / bitarray order, 8-byte aligned, ceil(n / 64) words each):
// thread = hit; the record of hit t goes to out[dest ? dest[t] : t]
__global__ __launch_bounds__(kBlock) void k_score_packed(
    const uint8_t *__restrict__ packed, const uint64_t *__restrict__ off, uint64_t n_hits, const uint32_t *__restrict__ hit_n,
    const uint32_t *__restrict__ found, const uint32_t *__restrict__ unique, const uint32_t *__restrict__ dest,
    bigsi_score::HitScore *__restrict__ out)
{
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_hits) return;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(packed + off[t]);
    bigsi_score::HitScore rec;
    bigsi_score::score_hit([w](uint32_t k) { return by_column(w[k]); }, hit_n[t], found ? found[t] : 0u, unique ? unique[t] : 0u, &rec);
    out[dest ? dest[t] : t] = rec;
}

// ------------------------------------------------------------------------------ results of a run -> pinned host memory
// What BIGSI.search needs back from a run -- per query: k-mers, unique k-mers, min_kmers, hit offsets; the first `spec` hits --
// written by ONE small kernel straight into a block of pinned host memory the batch owns:
//   [hit_off: n_seqs + 2 uint64 | num_kmers, num_unique, min_kmers: 3 n_seqs uint32 | pad to 8 | colours: spec uint32 | counts: spec uint32]
// instead of three or four device-to-host copies, each with its own ~10 us of latency and a synchronisation: a call then waits
// ONCE, for this kernel's event.  Word n_seqs + 1 of hit_off is the mark of a one-launch read run that gave up (0 otherwise).
// The LAST workgroup to finish then raises a flag word in (coherent) pinned memory, which is what the host spins on: no event
// record, no hipEventSynchronize -- scripts/probe/latency_probe.hip prices the difference.
__global__ __launch_bounds__(kBlock) void k_export_results(
    const uint64_t *__restrict__ hit_off, uint32_t n_seqs, uint32_t with_mark, const uint32_t *__restrict__ uniq,
    const uint32_t *__restrict__ col, const uint32_t *__restrict__ cnt, uint32_t spec, uint64_t *out,
    uint32_t *__restrict__ done_count /* device word, zero between launches */, volatile uint64_t *flag, uint64_t serial)
{
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x, nt = gridDim.x * kBlock;
    const uint64_t total = hit_off[n_seqs];
    const uint32_t m = (uint32_t)(total < spec ? total : spec);
    for (uint32_t i = tid; i < n_seqs + 2u; i += nt) out[i] = i <= n_seqs ? hit_off[i] : (with_mark ? hit_off[i] : 0ull);
    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + n_seqs + 2u);
    for (uint32_t i = tid; i < 3u * n_seqs; i += nt) o32[i] = uniq[i];
    uint32_t *ocol = o32 + ((3u * n_seqs + 1u) & ~1u), *ocnt = ocol + spec;
    for (uint32_t i = tid; i < m; i += nt) { ocol[i] = col[i]; ocnt[i] = cnt[i]; }
    if (!flag) return;
    __threadfence_system();                     // this thread's stores to host memory are out ...
    __syncthreads();                            // ... and so are the workgroup's
    if (threadIdx.x == 0) {
        if (gridDim.x == 1 || atomicAdd(done_count, 1u) == gridDim.x - 1u) {
            if (gridDim.x > 1) *done_count = 0;
            __threadfence_system();
            *flag = serial;
        }
    }
}

// The same for a one-launch read run, whose hit lists lie in allocation order (k_reads_fused): workgroup g owns a contiguous range
// of queries, sums the hit counts of the queries before its range (plain loads: the launch boundary has published them), scans
// its own, writes the offsets and copies every query's hits to their place in query order.  No workgroup waits for another.
__global__ __launch_bounds__(kBlock) void k_export_reads(
    const uint64_t *__restrict__ q_start, const uint32_t *__restrict__ q_cnt, uint32_t n_seqs, const uint32_t *__restrict__ uniq,
    const uint32_t *__restrict__ col, const uint32_t *__restrict__ cnt, uint64_t capacity /* entries col / cnt hold */, uint32_t spec,
    uint64_t *out, uint32_t *__restrict__ done_count, volatile uint64_t *flag, uint64_t serial)
{
    __shared__ uint32_t lds[16];
    __shared__ uint64_t lds64[kBlock / 64];
    const uint32_t per_g = (n_seqs + gridDim.x - 1) / gridDim.x, q0 = blockIdx.x * per_g < n_seqs ? blockIdx.x * per_g : n_seqs;
    const uint32_t q1 = q0 + per_g < n_seqs ? q0 + per_g : n_seqs;
    uint64_t part = 0;
    for (uint32_t j = threadIdx.x; j < q0; j += kBlock) part += q_cnt[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if ((threadIdx.x & 63u) == 0) lds64[threadIdx.x >> 6] = part;
    __syncthreads();
    uint64_t base = 0;
#pragma unroll
    for (int i = 0; i < kBlock / 64; i++) base += lds64[i];
    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + n_seqs + 2u);
    uint32_t *ocol = o32 + ((3u * n_seqs + 1u) & ~1u), *ocnt = ocol + spec;
    for (uint32_t qb = q0; qb < q1; qb += kBlock) {
        const uint32_t q = qb + threadIdx.x;
        const uint32_t c = q < q1 ? q_cnt[q] : 0u;
        uint32_t tot;
        const uint32_t pre = block_exclusive_scan(c, &tot, lds);
        const uint64_t off = base + pre;
        base += tot;
        if (q < q1) {
            out[q] = off;
            // (a list the read kernel could not place -- the buffers were full: its start lies beyond them, nothing was written;
            // the total then exceeds the capacity and the host takes the route that grows the buffers)
            const uint64_t src = c ? q_start[q] : 0ull;
            if (src + c <= capacity)
                for (uint32_t j = 0; j < c && off + j < spec; j++) { ocol[off + j] = col[src + j]; ocnt[off + j] = cnt[src + j]; }
        }
    }
    if ((q1 == n_seqs && q0 < n_seqs) || (n_seqs == 0 && blockIdx.x == 0)) {
        if (threadIdx.x == 0) { out[n_seqs] = base; out[n_seqs + 1] = 0; }
    }
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x, nt = gridDim.x * kBlock;
    for (uint32_t i = tid; i < 3u * n_seqs; i += nt) o32[i] = uniq[i];
    if (!flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.x == 1 || atomicAdd(done_count, 1u) == gridDim.x - 1u) {
            if (gridDim.x > 1) *done_count = 0;
            __threadfence_system();
            *flag = serial;
        }
    }
}

// A one-call search's tables + sequences from the pinned staging to their device arrays: wide coalesced reads over the host link
// (a few hundred requests for 85 KB) in place of the thousands of 64-byte ones k_reads_fused's 1000 workgroups would issue each
// for its own offsets and sequence (measured: 1000 reads of 61 bp in one call 54.8 -> 51.9 us), and of a hipMemcpyAsync whose
// DMA start-up is ~13 us.
__global__ __launch_bounds__(kBlock) void k_stage_in(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t bytes)
{
    const uint64_t n16 = bytes / 16, tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x, nt = (uint64_t)gridDim.x * kBlock;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (uint64_t i = tid; i < n16; i += nt) d4[i] = s4[i];
    for (uint64_t i = n16 * 16 + tid; i < bytes; i += nt) dst[i] = src[i];
}

// ------------------------------------------------------------------------------ bulk ingest helpers
// n CONSECUTIVE rows [row0, row0 + n) of rb bytes each, packed, to / from the matrix (file <-> HBM: bigsi_hip_load_rows_file /
// save_rows_file when the file's row length is not the device pitch): one workgroup per row, 16-byte lanes where alignment allows.
__global__ __launch_bounds__(kBlock) void k_scatter_run(uint8_t *__restrict__ index, uint64_t stride_bytes, uint64_t row0, const uint8_t *__restrict__ packed, uint64_t rb)
{
    uint8_t *dst = index + (row0 + blockIdx.x) * stride_bytes;
    const uint8_t *src = packed + (uint64_t)blockIdx.x * rb;
    for (uint64_t b = threadIdx.x; b < stride_bytes; b += kBlock) dst[b] = b < rb ? src[b] : (uint8_t)0;
}

__global__ __launch_bounds__(kBlock) void k_gather_run(const uint8_t *__restrict__ index, uint64_t stride_bytes, uint64_t row0, uint8_t *__restrict__ packed, uint64_t rb)
{
    const uint8_t *src = index + (row0 + blockIdx.x) * stride_bytes;
    uint8_t *dst = packed + (uint64_t)blockIdx.x * rb;
    for (uint64_t b = threadIdx.x; b < rb; b += kBlock) dst[b] = b < stride_bytes ? src[b] : (uint8_t)0;
}

// ------------------------------------------------------------------------------ storage contract helpers
// scatter n packed rows (rb bytes each) to rows row_ids[i]; bytes [rb, stride) of the row are zeroed.
__global__ __launch_bounds__(kBlock) void k_scatter_rows(
    uint8_t *__restrict__ index, uint64_t stride_bytes, uint64_t m, const uint64_t *__restrict__ row_ids,
    const uint8_t *__restrict__ packed, uint64_t rb)
{
    const uint64_t i = blockIdx.x;   // one workgroup per row
    const uint64_t row = row_ids[i];
    if (row >= m) return;
    uint8_t *dst = index + row * stride_bytes;
    for (uint64_t b = threadIdx.x; b < stride_bytes; b += kBlock) dst[b] = b < rb ? packed[i * rb + b] : (uint8_t)0;
}

__global__ __launch_bounds__(kBlock) void k_gather_rows(
    const uint8_t *__restrict__ index, uint64_t stride_bytes, uint64_t m, const uint64_t *__restrict__ row_ids,
    uint8_t *__restrict__ packed, uint64_t rb)
{
    const uint64_t i = blockIdx.x;   // one workgroup per row
    const uint64_t row = row_ids[i];
    if (row >= m) return;
    const uint8_t *src = index + row * stride_bytes;
    for (uint64_t b = threadIdx.x; b < rb; b += kBlock) packed[i * rb + b] = b < stride_bytes ? src[b] : (uint8_t)0;
}

// re-stride rows when the column capacity grows (new stride > old stride); runs back to front is not needed because
// the destination is a fresh allocation.
__global__ __launch_bounds__(kBlock) void k_restride(
    const uint64_t *__restrict__ src, uint64_t src_stride, uint64_t *__restrict__ dst, uint64_t dst_stride, uint64_t m)
{
    const uint64_t total = m * dst_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = i / dst_stride, w = i - r * dst_stride;
        dst[i] = w < src_stride ? src[r * src_stride + w] : 0ull;
    }
}

// BitMatrix.insert_column / get_column (bigsi/matrix/bitmatrix.py:50-75): bloom bit r <-> bit `col` of row r.
__global__ __launch_bounds__(kBlock) void k_insert_column(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t col, const uint8_t *__restrict__ bloom)
{
    const uint64_t w = col >> 6, bit = bit_of_col((uint32_t)(col & 63u));
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < m; r += (uint64_t)gridDim.x * kBlock) {
        const uint64_t on = (bloom[r >> 3] >> (7 - (r & 7))) & 1u;
        uint64_t *p = index + r * stride_words + w;
        *p = (*p & ~(1ull << bit)) | (on << bit);
    }
}

__global__ __launch_bounds__(kBlock) void k_get_column(
    const uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t col, uint8_t *__restrict__ bloom)
{
    // one thread per output byte = 8 rows
    const uint64_t w = col >> 6, bit = bit_of_col((uint32_t)(col & 63u));
    const uint64_t nb = (m + 7) / 8;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * kBlock) {
        uint32_t v = 0;
        for (uint32_t j = 0; j < 8; j++) {
            const uint64_t r = b * 8 + j;
            if (r < m && ((index[r * stride_words + w] >> bit) & 1ull)) v |= 0x80u >> j;
        }
        bloom[b] = (uint8_t)v;
    }
}

// transpose (bigsi/matrix/transpose.py:33-43) on the device: n Bloom filters (bloom c at blooms + c*bloom_stride, m bits,
// row byte format) become columns [col0, col0+n) of the matrix.  One thread per (row, 64-column word); the 8 threads of
// 8 consecutive rows read the same Bloom byte (one L1 line per wave), the word is read-modified-written once.
__global__ __launch_bounds__(kBlock) void k_insert_columns(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t col0, uint64_t ncols,
    const uint8_t *__restrict__ blooms, uint64_t bloom_stride)
{
    const uint64_t w0 = col0 >> 6, nwords = ((col0 + ncols - 1) >> 6) - w0 + 1;
    const uint64_t total = m * nwords;
    for (uint64_t item = (uint64_t)blockIdx.x * kBlock + threadIdx.x; item < total; item += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = item % m, w = w0 + item / m;
        const uint64_t clo = col0 > w * 64 ? col0 : w * 64;
        const uint64_t chi = col0 + ncols < w * 64 + 64 ? col0 + ncols : w * 64 + 64;
        uint64_t mask = 0, val = 0;
        for (uint64_t c = clo; c < chi; c++) {
            const uint64_t bit = (blooms[(c - col0) * bloom_stride + (r >> 3)] >> (7 - (r & 7))) & 1u;
            const uint32_t b = bit_of_col((uint32_t)(c & 63u));
            mask |= 1ull << b;
            val |= bit << b;
        }
        uint64_t *p = index + r * stride_words + w;
        *p = (*p & ~mask) | val;
    }
}

constexpr int kTransposeTile = 512, kTransposeSuper = 32;      // rows of a tile pass; supertile edge, in tiles

#ifdef BIGSI_HIP_TUNING
// ROUNDS 2-6's KERNEL, kept in tuning builds only (BIGSI_HIP_TR_REGS=0; scripts/ab_transpose_regs.sh, scripts/probe/transpose_ab.hip)
// as the other side of the A/B that k_transpose_regs below is measured against.  The product library does not hold it.
constexpr int kTransposePitch = 72;
// The transpose as a bandwidth kernel (full 64-column words; k_insert_columns above keeps the ragged edges).
// A tile is 512 rows x 512 columns: 64 bytes of each of 512 filters in, 64 bytes of each of 512 rows out, as whole runs
// (16 bytes per lane), so every sector that crosses the memory interface is used in full and nothing is read-modified-written;
// a workgroup moves 2 x 2 tiles, which makes the runs 128 bytes on both sides (RT, CT below).  In between, the tile is 64 blocks of 64 x 64 bits; a wavefront
// transposes a block in registers -- lane l holds the 64 row bits of one column, six butterfly steps (exchange with lane
// l ^ j, j = 32 .. 1) leave lane i holding the 64 column bits of row i -- reading its operands from and writing its results
// to LDS (pitch 72 bytes: conflict-free 8-byte accesses for 32 lanes at a stride of one LDS row).
// Bit order: both the filters and the rows keep the reference's byte format (bit 7 - (i & 7) of byte i >> 3), so in a
// little-endian uint64 element i sits at bit_of_col(i); by_column() turns that into plain order for the butterfly, and lane
// l is given column bit_of_col(l) of the word, which puts every result bit where the row format wants it.

// 64 x 64 bit transpose across the lanes of a wavefront, on 32-bit halves (64-bit shifts run at a quarter of the rate):
// after it, bit k of lane i = bit i of (the original value of) lane k.  Step j (32, 16, .. 1) swaps, between lanes l and
// l ^ j, the off-diagonal j x j blocks.  Each lane ROTATES what its partner needs into place before the exchange (towards
// the high bits if the lane has bit j set, towards the low bits otherwise: one v_alignbit with a per-lane amount) and
// merges what it receives under a per-lane mask (one v_bfi).
// The exchange itself stays in the vector ALUs: lane ^ 1 and ^ 2 are quad permutes, ^ 4 and ^ 8 two row mirrors each (DPP
// modifiers of a v_mov), ^ 16 and ^ 32 gfx950's v_permlane16_swap / v_permlane32_swap (each checked lane by lane on the
// hardware) instead of 11 ds_bpermute through the CU's one LDS crossbar.  Neither this nor a variant with 8 x 8 bit blocks in
// registers (2x fewer VALU operations) moved the kernel: with the butterflies skipped altogether (BIGSI_HIP_TR_SKIP=1 in a
// tuning build) it runs at the same rate -- the bound is the access pattern: 128-byte runs at large strides on both sides
// (RT = CT = 2 below) move 3.5-4.1 TB/s, 64-byte runs 3.1, against the 6.3 TB/s the HBM gives a copy.
__device__ __forceinline__ uint32_t rotl32v(uint32_t x, uint32_t r) { return __builtin_amdgcn_alignbit(x, x, (32u - r) & 31u); }

template <int J> __device__ __forceinline__ uint32_t lane_xor(uint32_t x, uint32_t lane)
{
    constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kQuadReverse = 0x1B, kRowMirror = 0x140, kRowHalfMirror = 0x141;
    if (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, kQuadXor1, 0xF, 0xF, true);
    if (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, kQuadXor2, 0xF, 0xF, true);
    if (J == 4)      // i -> 7 - i within 8 lanes is i ^ 7; reversing each quad is ^ 3
        return (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)x, kRowHalfMirror, 0xF, 0xF, true), kQuadReverse, 0xF, 0xF, true);
    if (J == 8)      // i ^ 15, then i ^ 7
        return (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)x, kRowMirror, 0xF, 0xF, true), kRowHalfMirror, 0xF, 0xF, true);
    if (J == 16) {   // odd rows (16 lanes) of the first operand <-> even rows of the second
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16u) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane & 32u) ? r[0] : r[1];
}

__device__ __forceinline__ uint64_t transpose64_lanes(uint64_t a64, uint32_t lane)
{
    uint32_t lo = (uint32_t)a64, hi = (uint32_t)(a64 >> 32);
    {   // j = 32: whole halves change lanes
        const bool s = (lane & 32u) != 0;
        const uint32_t recv = lane_xor<32>(s ? lo : hi, lane);
        lo = s ? recv : lo;
        hi = s ? hi : recv;
    }
#define BIGSI_TR_STEP(J, M)                                                                                  \
    {                                                                                                        \
        const bool s = (lane & J) != 0;                                                                      \
        const uint32_t rot = s ? (uint32_t)J : 32u - (uint32_t)J;   /* s: partner wants my bits J higher; else J lower */ \
        const uint32_t keep = s ? ~(uint32_t)M : (uint32_t)M;       /* the bits of my own value that stay */  \
        const uint32_t rl = lane_xor<J>(rotl32v(lo, rot), lane);                                             \
        const uint32_t rh = lane_xor<J>(rotl32v(hi, rot), lane);                                             \
        lo = (lo & keep) | (rl & ~keep);                                                                     \
        hi = (hi & keep) | (rh & ~keep);                                                                     \
    }
    BIGSI_TR_STEP(16, 0x0000FFFFu)
    BIGSI_TR_STEP(8, 0x00FF00FFu)
    BIGSI_TR_STEP(4, 0x0F0F0F0Fu)
    BIGSI_TR_STEP(2, 0x33333333u)
    BIGSI_TR_STEP(1, 0x55555555u)
#undef BIGSI_TR_STEP
    return ((uint64_t)hi << 32) | lo;
}

// two independent blocks at once, step by step: the same operations as two calls of transpose64_lanes, written so that the
// compiler sees the two chains side by side (k_transpose_tiles, phase 2)
__device__ __forceinline__ void transpose64_lanes_x2(uint64_t &a64, uint64_t &b64, uint32_t lane)
{
    uint32_t lo = (uint32_t)a64, hi = (uint32_t)(a64 >> 32), lo2 = (uint32_t)b64, hi2 = (uint32_t)(b64 >> 32);
    {
        const bool s = (lane & 32u) != 0;
        const uint32_t recv = lane_xor<32>(s ? lo : hi, lane), recv2 = lane_xor<32>(s ? lo2 : hi2, lane);
        lo = s ? recv : lo;
        hi = s ? hi : recv;
        lo2 = s ? recv2 : lo2;
        hi2 = s ? hi2 : recv2;
    }
#define BIGSI_TR_STEP2(J, M)                                                                                 \
    {                                                                                                        \
        const bool s = (lane & J) != 0;                                                                      \
        const uint32_t rot = s ? (uint32_t)J : 32u - (uint32_t)J;                                            \
        const uint32_t keep = s ? ~(uint32_t)M : (uint32_t)M;                                                \
        const uint32_t rl = lane_xor<J>(rotl32v(lo, rot), lane), rl2 = lane_xor<J>(rotl32v(lo2, rot), lane); \
        const uint32_t rh = lane_xor<J>(rotl32v(hi, rot), lane), rh2 = lane_xor<J>(rotl32v(hi2, rot), lane); \
        lo = (lo & keep) | (rl & ~keep);                                                                     \
        hi = (hi & keep) | (rh & ~keep);                                                                     \
        lo2 = (lo2 & keep) | (rl2 & ~keep);                                                                  \
        hi2 = (hi2 & keep) | (rh2 & ~keep);                                                                  \
    }
    BIGSI_TR_STEP2(16, 0x0000FFFFu)
    BIGSI_TR_STEP2(8, 0x00FF00FFu)
    BIGSI_TR_STEP2(4, 0x0F0F0F0Fu)
    BIGSI_TR_STEP2(2, 0x33333333u)
    BIGSI_TR_STEP2(1, 0x55555555u)
#undef BIGSI_TR_STEP2
    a64 = ((uint64_t)hi << 32) | lo;
    b64 = ((uint64_t)hi2 << 32) | lo2;
}

__device__ uint32_t g_tr_skip = 0;      // experiment (BIGSI_HIP_TR_SKIP=1): move the tiles without transposing them
// RT = 1: one 512-row tile per workgroup (64-byte filter runs); 2: two stacked tiles, their 128-byte filter runs loaded in one go.
// CT = 1: 512 columns per workgroup (64-byte row runs); 2: two tiles side by side, each handled by its own 256 threads in its
// own LDS buffer, their rows stored together as 128-byte runs (half as many DRAM row activations on the write side).
template <int RT, int CT>
__global__ __launch_bounds__(kBlock * CT) void k_transpose_tiles(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t w_first /* first column word written; even */,
    uint64_t n_words /* whole 64-column words to write */, const uint8_t *__restrict__ blooms /* filter of column 64 * w_first */,
    uint64_t /* n_filters: k_transpose_regs' argument; here every word has all its filters */,
    uint64_t bstride /* bytes between filters; multiple of 16 */, uint64_t nb /* valid bytes of a filter: ceil(m / 8) */,
    uint32_t rg, uint32_t cg /* tiles per XCD group along rows / columns: powers of two, rg * cg <= 128, cg <= sup_w */,
    uint32_t sup_w /* supertile width in tiles: a power of two <= 32 */)
{
    // ONE buffer per 512 x 512 tile: line L holds, before the transpose, the 64 row-bytes of column L and, after it, the 64
    // column-bytes of row L.  Block (cw, rc) of 64 x 64 bits sits at lines [64 cw, +64), bytes [8 rc, +8) and its transpose
    // belongs at lines [64 rc, +64), bytes [8 cw, +8) -- the place of block (rc, cw) -- so blocks are transposed in mirrored
    // pairs, each written where the other was read (37 KB of LDS per tile instead of 74).
    __shared__ __attribute__((aligned(16))) uint8_t tiles[CT][kTransposeTile * kTransposePitch];
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
    const uint64_t tile_r = (sup / sup_c) * sup_h + (g / gpr) * rg + t % rg;
    const uint64_t tile_c = (sup % sup_c) * sup_w + (g % gpr) * cg + t / rg;
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
        uint64_t *d = reinterpret_cast<uint64_t *>(tile + col * kTransposePitch + (part & 3u) * 16);
        d[0] = ld[it].x;
        d[1] = ld[it].y;
    }
    __syncthreads();
    // phase 2: the 64 blocks of the 8 x 8 grid, two at a time per wavefront -- ALWAYS two, in one basic block, so that the two
    // butterfly chains (each ~45 dependent vector operations, DPP / permlane exchanges among them) interleave: the two workgroups
    // of a CU are rarely in this phase together, and with two wavefronts per SIMD a single chain leaves the ALUs waiting for
    // their own results (round 6: +3 ... +5 % over one pair {(cw, rc), (rc, cw)} per trip with the second block behind a branch).
    // Trips 0-6 of a wavefront take the off-diagonal pairs {(x, y), (y, x)}, x < y, number `wave + 4 trip` of the 28 (each block
    // written where its mirror was read: 37 KB of LDS per tile instead of 74); trip 7 takes the diagonal blocks 2 wave and
    // 2 wave + 1, each transposed in place.
    if (!(g_tr_skip & 1u))
#pragma unroll 1
    for (uint32_t trip = 0; trip < 8; trip++) {
        uint32_t x, y, xb, yb;                              // block a = (cw = x, rc = y), block b = (cw = xb, rc = yb)
        if (trip < 7) {
            const uint32_t oi = wave + (kBlock / 64) * trip;          // oi -> (x, y) with x < y: row y of the strict lower triangle starts at y (y - 1) / 2
            y = 1;
            while (y * (y + 1) / 2 <= oi) y++;
            x = oi - y * (y - 1) / 2;
            xb = y, yb = x;
        } else {
            x = y = 2 * wave;
            xb = yb = 2 * wave + 1;
        }
        uint8_t *pa = tile + (x * 64) * kTransposePitch + y * 8;        // lines of column word x, bytes of row chunk y
        uint8_t *pb = tile + (xb * 64) * kTransposePitch + yb * 8;
        uint64_t va = *reinterpret_cast<const uint64_t *>(pa + mycol * kTransposePitch);
        uint64_t vb = *reinterpret_cast<const uint64_t *>(pb + mycol * kTransposePitch);
        va = by_column(va);
        vb = by_column(vb);
        transpose64_lanes_x2(va, vb, lane);
        // every lane of the wavefront has read both blocks before any lane overwrites them (the butterflies in between are
        // wavefront-wide exchanges), and no other wavefront touches these two blocks.  The transpose of block (x, y) belongs at
        // (y, x): for an off-diagonal pair that is where b was read (and vice versa), for a diagonal block its own place
        *reinterpret_cast<uint64_t *>(tile + (y * 64 + lane) * kTransposePitch + x * 8) = va;
        *reinterpret_cast<uint64_t *>(tile + (yb * 64 + lane) * kTransposePitch + xb * 8) = vb;
    }
    __syncthreads();
    // phase 3: tiles[..][row][0 .. 64) -> the rows' words [w_first + w0, +words_here): 4 * CT lanes per row, 16 bytes each
#pragma unroll
    for (int it = 0; it < kTransposeTile * 4 / kBlock; it++) {
        const uint32_t item = it * (kBlock * CT) + threadIdx.x, row = item / (4 * CT), part = item % (4 * CT);
        const uint64_t r = r0 + row;
        if (r >= m || part * 2 >= words_here) continue;
        const uint64_t *sp = reinterpret_cast<const uint64_t *>(tiles[part >> 2] + row * kTransposePitch + (part & 3u) * 16);
        uint64_t *dst = index + r * stride_words + w_first + w0 + part * 2;
        if (part * 2 + 1 < words_here) __builtin_nontemporal_store(u64x2{sp[0], sp[1]}, reinterpret_cast<u64x2 *>(dst));
        else dst[0] = sp[0];
    }
    }
}

// (Round 6 also built the form that takes a workgroup's two column tiles ONE AFTER THE OTHER through one LDS buffer -- 256 threads,
// 37 KB, four workgroups per CU, the second tile's loads in flight under the first tile's butterflies, the first tile's rows waiting
// in registers so that both 64-byte halves of a row's run leave in consecutive store instructions: bit-equal, 3.4-3.7 TB/s against
// 4.1-4.8 for k_transpose_tiles<1,2>, with non-temporal or plain stores alike (profiles/r06_transpose_seq_ab.txt): the halves do not
// merge on their way out, and a 64-byte write run is what the bare mover prices at 3.7.  Removed.)
#endif      // BIGSI_HIP_TUNING

// ------------------------------------------------------------------------------ the transpose without lane butterflies (round 6)
// k_transpose_tiles spends ~90 vector instructions per 64 x 64 bit block on six exchange steps between lanes, and passes every
// tile through the LDS twice (in, block by block in place, out); with the filters at an aligned pitch the memory side of it moves
// 5.0-5.2 TB/s and the kernel 4.1-4.4 (profiles/r06_transpose_ab_aligned.txt: "no phase 2").  k_transpose_regs does the BIT
// level of the transpose between the REGISTERS of a thread and the BYTE level on the way out of the LDS:
//   1. a thread loads 16 bytes (128 rows) of EIGHT neighbouring filters -- columns 8 g .. 8 g + 7, the columns of ONE byte of
//      the rows -- and transposes the 8 x 8 bit blocks between its 8 registers, all 16 byte lanes at once: three steps of
//      shift + v_bfi on register pairs (192 vector instructions per 128 bytes, an eighth of the butterflies).  Register t then
//      holds, for each of the thread's 16 row-bytes e, the byte (columns 8 g .. 8 g + 7) of row 8 e + t;
//   2. it stores them as 8-byte words -- 8 bytes = 8 rows that are 8 apart, one column byte -- at LDS word (t, e / 8, p; g);
//   3. after ONE barrier, ds_read_b64_tr_b8 (gfx950's transposing LDS read: inside a group of 16 lanes, lane (a, b) receives byte
//      b of the words addressed by lanes 2 j + a, j = 0 .. 7; scripts/probe/tr_probe.hip prints the map read off the hardware)
//      hands every lane 8 CONSECUTIVE column bytes of one row: two reads = the 16 bytes of its store.  A wave instruction
//      stores 8 rows x 128 bytes.
// LDS: 64 x (128 + 4) words of 8 bytes = 66 KB per workgroup of 512 threads (512 rows x 1024 columns), two per CU; the +4
// words make the stores of phase 2 conflict-free (16 lanes = 4 column bytes x 4 pieces), the flip of word bit 3 by bit 5 the reads
// of phase 3 (32 lanes = 4 pieces x 8 column bytes, 16 words apart).
// RT = 2: 1024 rows per workgroup -- the thread loads both 64-byte halves of its filters' 128-byte lines at once and the two
// 512-row halves go through the LDS one after the other.
// 8 x 8 bit blocks between 8 registers, bytes independent: x[i] bit (7 - t) of byte q  ->  x[t] bit (7 - i) of byte q
__device__ __forceinline__ void transpose8_regs(uint32_t (&x)[8])
{
#define BIGSI_TR8_STEP(S, M)                                                                  \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                           \
        if (u & S) continue;                                                                  \
        const uint32_t xu = x[u], xv = x[u + S];                                              \
        x[u + S] = (xv & (uint32_t)M) | ((xu << S) & ~(uint32_t)M);                           \
        x[u] = (xu & ~(uint32_t)M) | ((xv >> S) & (uint32_t)M);                               \
    }
    BIGSI_TR8_STEP(4, 0x0F0F0F0Fu)
    BIGSI_TR8_STEP(2, 0x33333333u)
    BIGSI_TR8_STEP(1, 0x55555555u)
#undef BIGSI_TR8_STEP
}

typedef int tr_v2i __attribute__((ext_vector_type(2)));

template <int RT, int CW = 1 /* tile width in units of 1024 columns: 2 = 256-byte row runs, 1024 threads, 133 KB of LDS (A/B) */>
__global__ __launch_bounds__(kBlock * 2 * CW) void k_transpose_regs(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint64_t w_first /* first column word written; even */,
    uint64_t n_words /* 64-column words to write (all of each: columns at or beyond n_filters are written as zeros) */,
    const uint8_t *__restrict__ blooms /* filter of column 64 * w_first */, uint64_t n_filters /* filters there are, from that one on */,
    uint64_t bstride /* bytes between filters; multiple of 16 */, uint64_t nb /* valid bytes of a filter: ceil(m / 8) */,
    uint32_t rg, uint32_t cg /* tiles per XCD group along rows / columns: powers of two, rg * cg <= 128, cg <= sup_w */,
    uint32_t sup_w /* supertile width in tiles: a power of two <= 32 */)
{
    constexpr uint32_t kPitch = 128u * CW + 4u;           // 8-byte words per (t, eh, p) line of the image: the tile's column bytes + 4
    __shared__ __attribute__((aligned(16))) uint64_t image[64 * kPitch];
    constexpr uint32_t kWordsPerBlock = 16 * CW;        // 1024 CW columns
    auto word_at = [](uint32_t line, uint32_t cb) { return line * kPitch + (cb ^ ((cb >> 2) & 8u)); };
    // workgroup -> tile: as k_transpose_tiles (supertiles of 1024 tiles, groups of rg x cg neighbours on one XCD)
    const uint64_t tiles_c = (n_words + kWordsPerBlock - 1) / kWordsPerBlock, sup_c = (tiles_c + sup_w - 1) / sup_w;
    const uint32_t sup_h = (uint32_t)(kTransposeSuper * kTransposeSuper) / sup_w;
    const uint64_t sup = blockIdx.x / (kTransposeSuper * kTransposeSuper);
    const uint32_t within = blockIdx.x % (kTransposeSuper * kTransposeSuper);
    const uint32_t gsz = rg * cg, xcd = within & 7u, sl = within >> 3, tg = sl % gsz, grp = (sl / gsz) * 8u + xcd;
    const uint32_t gpr = sup_w / cg;
    const uint64_t tile_r = (sup / sup_c) * sup_h + (grp / gpr) * rg + tg % rg;
    const uint64_t tile_c = (sup % sup_c) * sup_w + (grp % gpr) * cg + tg / rg;
    if (tile_r * kTransposeTile * RT >= m || tile_c >= tiles_c) return;
    const uint64_t byte0 = tile_r * (kTransposeTile / 8) * RT;
    const uint64_t w0 = tile_c * kWordsPerBlock;
    const uint32_t words_here = (uint32_t)(n_words - w0 < kWordsPerBlock ? n_words - w0 : kWordsPerBlock);
    // phase 1: thread (g, p): 16 bytes at byte0 + 64 half + 16 p of the filters of columns 8 g .. 8 g + 7
    const uint32_t g = threadIdx.x >> 2, p = threadIdx.x & 3u;
    u64x2 ld[RT][8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
#pragma unroll
        for (int half = 0; half < RT; half++) {
            const uint64_t off = byte0 + 64 * half + 16 * p;
            const bool ok = g < words_here * 8u && w0 * 64 + g * 8u + it < n_filters && off + 16 <= bstride && off < nb;
            const u64x2 *src = reinterpret_cast<const u64x2 *>(blooms + (w0 * 64 + (ok ? g * 8u + it : 0u)) * bstride + (ok ? off : 0));
            // PLAIN loads: the two 64-byte halves of a filter's line are asked for by consecutive instructions, and only a line the
            // vector L1 has allocated takes the second one as a hit -- with non-temporal loads the L2 was asked 1.19 times per line
            // (1.31 with RT = 1, where the other half belongs to another workgroup; TCC_EA0_RDREQ, scripts/pmc_requests.py), with the
            // halves eight instructions apart twice; with plain loads 1.000 (+5 ... +8 % on the kernel)
            ld[half][it] = ok ? *src : u64x2{0ull, 0ull};
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // phase 3's places: the wavefront takes the 64 rows 128 pw + 64 ehw + (0 .. 63) of the half; instruction t reads line (t, ehw, pw)
    // for the rows t + 8 b.  Lane = (gi, i): it ADDRESSES column byte 16 (2 gi + (i & 1)) + (i >> 1) (+ 8 for the second read) and
    // RECEIVES (b = i & 7, a = i >> 3) the bytes 16 (2 gi + a) + 0 .. 7 (8 .. 15) of row 8 b + t
    // (CW = 2: 16 wavefronts; a wavefront takes 4 of the 8 values of t and both 128-byte halves of its rows' 256-byte runs, one right after the other)
    const uint32_t pw = wave / (2u * CW), ehw = (wave / CW) & 1u, th = wave % CW, gi = lane >> 4, li = lane & 15u;
    const uint32_t cb_addr = 16u * (2u * gi + (li & 1u)) + (li >> 1);
    const uint32_t piece = 2u * gi + (li >> 3), brow = li & 7u;
#pragma unroll
    for (int half = 0; half < RT; half++) {
        if (half) __syncthreads();                          // phase 3 of the first half has read the image
        // phases 1b + 2: bit transpose between the registers, dword by dword, then the words of the image
#pragma unroll
        for (int d2 = 0; d2 < 2; d2++) {                    // d2 = eh: row bytes 16 p + 8 eh + (0 .. 7)
            uint32_t lo[8], hi[8];
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const uint64_t v = d2 ? ld[half][it].y : ld[half][it].x;
                lo[it] = (uint32_t)v;
                hi[it] = (uint32_t)(v >> 32);
            }
            transpose8_regs(lo);
            transpose8_regs(hi);
#pragma unroll
            for (int t = 0; t < 8; t++) image[word_at((t * 2 + d2) * 4 + p, g)] = ((uint64_t)hi[t] << 32) | lo[t];
        }
        __syncthreads();
        const uint64_t r0 = (tile_r * RT + half) * kTransposeTile + 128u * pw + 64u * ehw + 8u * brow;
        tr_v2i v0[8], v1[8];
#pragma unroll
        for (int sl = 0; sl < 8; sl++) {                     // store slot sl = (t, 128-byte half of the run)
            const uint32_t t = (8u / CW) * th + sl / CW, ch = sl % CW, line = (t * 2 + ehw) * 4 + pw;
            v0[sl] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) tr_v2i *)(image + word_at(line, 128u * ch + cb_addr)));
            v1[sl] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) tr_v2i *)(image + word_at(line, 128u * ch + cb_addr + 8u)));
        }
#pragma unroll
        for (int sl = 0; sl < 8; sl++) {
            const uint32_t t = (8u / CW) * th + sl / CW, pc = 8u * (sl % CW) + piece;
            const uint64_t r = r0 + t;
            if (r >= m || pc * 2 >= words_here) continue;
            const uint64_t a = ((uint64_t)(uint32_t)v0[sl].y << 32) | (uint32_t)v0[sl].x, b = ((uint64_t)(uint32_t)v1[sl].y << 32) | (uint32_t)v1[sl].x;
            uint64_t *dst = index + r * stride_words + w_first + w0 + pc * 2;
            // (non-temporal STORES stay: plain ones measured -2 ... -4 % on three shapes, three interleaved repetitions)
            if (pc * 2 + 1 < words_here) __builtin_nontemporal_store(u64x2{a, b}, reinterpret_cast<u64x2 *>(dst));
            else dst[0] = a;
        }
    }
}

// merge_indexes (bigsi/graph/index.py:54-60): append the n2 columns of src after the n1 columns of dst, row by row,
// device to device.  One thread per (row, destination byte); bits are MSB-first inside a byte, so a column offset that
// is not a multiple of 8 is a bit shift across source bytes.
__global__ __launch_bounds__(kBlock) void k_append_columns(
    uint64_t *__restrict__ dst, uint64_t dst_stride_words, uint64_t n1, const uint64_t *__restrict__ src, uint64_t src_stride_words,
    uint64_t n2, uint64_t m)
{
    // one thread per (row, destination word): in plain column order (by_column) the appended columns are the source row shifted
    // up by n1 % 64 bits -- two source words funnel into one destination word; the first destination word keeps its own
    // low columns.  (The first version moved one byte per thread, bit by bit.)
    const uint64_t j0 = n1 >> 6, j1 = (n1 + n2 + 63) >> 6, per_row = j1 - j0;
    const uint32_t sh = (uint32_t)(n1 & 63u);
    const uint64_t src_words = (n2 + 63) >> 6;
    const uint64_t total = m * per_row;
    for (uint64_t item = (uint64_t)blockIdx.x * kBlock + threadIdx.x; item < total; item += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = item / per_row, q = item % per_row, j = j0 + q;
        const uint64_t *sr = src + r * src_stride_words;
        const uint64_t cur = q < src_words ? by_column(sr[q]) : 0ull;
        uint64_t v;
        if (sh == 0) {
            v = cur;
        } else {
            const uint64_t prev = q > 0 ? by_column(sr[q - 1]) : 0ull;
            v = (prev >> (64u - sh)) | (cur << sh);
        }
        const uint64_t end = n1 + n2;                      // columns at or beyond it stay zero
        if (end < (j + 1) * 64) v &= (end & 63u) ? ((1ull << (end & 63u)) - 1) : ~0ull;
        uint64_t *d = dst + r * dst_stride_words + j;
        if (q == 0 && sh) v |= by_column(*d) & ((1ull << sh) - 1);
        *d = by_column(v);
    }
}

// Bloom-add k-mers to one sample column of the transposed matrix (bloom/bloomfilter.py:25-32 + graph/bigsi.py:151).
__global__ __launch_bounds__(kBlock) void k_insert_kmers(
    uint64_t *__restrict__ index, uint64_t stride_words, uint64_t m, uint32_t h, uint64_t col,
    const char *__restrict__ seqs, const uint64_t *__restrict__ seq_off, uint32_t k)
{
    const uint32_t q = blockIdx.x;
    const char *s = seqs + seq_off[q];
    const uint64_t len = seq_off[q + 1] - seq_off[q];
    const uint64_t n = len >= k ? len - k + 1 : 0;
    const uint64_t w = col >> 6, bitmask = 1ull << bit_of_col((uint32_t)(col & 63u));
    for (uint64_t i = threadIdx.x; i < n; i += kBlock) {
        KmerView v{s + i, k, use_revcomp(s + i, k)};
        for (uint32_t sd = 0; sd < h; sd++)
            atomicOr((unsigned long long *)(index + row_of_hash(murmur3_32(v, sd), m) * stride_words + w), (unsigned long long)bitmask);
    }
}

// BIGSI.bloom (graph/bigsi.py:150-155): u k-mers (k bytes each, packed) -> m-bit filter in the row byte format.
__global__ __launch_bounds__(kBlock) void k_bloom(
    uint32_t *__restrict__ bloom_words, uint64_t m, uint32_t h, const char *__restrict__ kmers, uint64_t u, uint32_t k, bool raw)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < u; i += (uint64_t)gridDim.x * kBlock) {
        const char *km = kmers + i * k;
        KmerView v{km, k, raw ? false : use_revcomp(km, k)};
        for (uint32_t sd = 0; sd < h; sd++) {
            const uint64_t r = row_of_hash(murmur3_32(v, sd), m);
            // byte r/8, mask 0x80 >> (r%8), addressed through little-endian 32-bit words
            atomicOr(&bloom_words[r >> 5], 1u << (8u * (uint32_t)((r >> 3) & 3u) + 7u - (uint32_t)(r & 7u)));
        }
    }
}

// synthetic contents: must match oracle/bigsi_oracle.c: orc_synth_word / orc_valid_mask bit for bit.
__global__ __launch_bounds__(kBlock) void k_fill_synth(
    uint64_t *__restrict__ index, uint64_t m, uint64_t stride_words, uint64_t n_cols, uint64_t seed, uint64_t shard, uint32_t draws)
{
    const uint64_t base = mix64(seed + shard * 0x632BE59BD9B4E019ull);
    const uint64_t pairs_per_row = stride_words / 2, total = m * pairs_per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = i / pairs_per_row, w = (i - r * pairs_per_row) * 2;
        const uint64_t rk = mix64(base ^ (r * 0x9E3779B97F4A7C15ull));
        u64x2 v = {~0ull, ~0ull};
        for (uint32_t d = 0; d < draws; d++) {
            v.x &= mix64(rk + (w * 8 + d) * 0xD1B54A32D192ED03ull);
            v.y &= mix64(rk + ((w + 1) * 8 + d) * 0xD1B54A32D192ED03ull);
        }
        v.x &= valid_mask(w, n_cols);
        v.y &= valid_mask(w + 1, n_cols);
        *reinterpret_cast<u64x2 *>(index + r * stride_words + w) = v;
    }
}

// ------------------------------------------------------------------------------ calibration: what this box gives bare row streams
// bigsi_hip_probe_rows (MEASUREMENT): a kernel with NO BIGSI code reads row lists of the index itself the way the row-AND
// kernels do -- one wavefront per 1 KiB column segment of every row of a "query", lane = 16 bytes, eight non-temporal loads in
// flight, AND-reduce -- so that a bench line can say which fraction of THIS box's random-row (counting kernel) or
// address-ordered (exact kernel) rate a leg reached: boxes differ by several per cent at identical clocks.
// ids: random = uniform over [0, m); sorted = one uniform draw per stratum [i*m/R, (i+1)*m/R): ascending by construction.
__global__ __launch_bounds__(kBlock) void k_probe_ids(uint64_t *__restrict__ ids, uint64_t n_queries, uint32_t rows_per_query, uint64_t m, uint32_t sorted, uint64_t seed)
{
    const uint64_t total = n_queries * rows_per_query;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = mix64(seed + i * 0x9E3779B97F4A7C15ull);
        if (!sorted) {
            ids[i] = r % m;
        } else {
            const uint64_t j = i % rows_per_query, lo = (unsigned __int128)j * m / rows_per_query, hi = (unsigned __int128)(j + 1) * m / rows_per_query;
            ids[i] = lo + (hi > lo ? r % (hi - lo) : 0);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_probe_rows(const uint64_t *__restrict__ index, uint64_t stride_words, uint32_t wv, const uint64_t *__restrict__ ids,
                                                       uint32_t rows_per_query, uint32_t segs, uint32_t q0, uint32_t q1, u64x2 *__restrict__ out)
{
    // (the wavefront's number is wave-uniform, which the compiler cannot see in threadIdx.x >> 6: without the readfirstlane the row
    // ids arrive through per-lane vector loads instead of scalar loads -- 6 % of the probe's rate; scripts/probe/row_probe.hip has
    // that handicap, which is why round 3's "bare kernel" figures sat BELOW k_and_exact)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6))), lane = threadIdx.x & 63u;
    const uint32_t q = q0 + wave / segs, seg = wave % segs;
    if (q >= q1) return;
    const uint32_t w0 = seg * 128u + lane * kVec;
    if (w0 >= wv) return;
    const uint64_t *r = ids + (uint64_t)q * rows_per_query;
    u64x2 acc = {~0ull, ~0ull};
    uint32_t i = 0;
    for (; i + 8 <= rows_per_query; i += 8) {
        u64x2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = load_row_seg<true>(index, r[i + j], stride_words, w0);
#pragma unroll
        for (int j = 0; j < 8; j++) acc &= v[j];
    }
    for (; i < rows_per_query; i++) acc &= load_row_seg<true>(index, r[i], stride_words, w0);
    out[((uint64_t)(q - q0) * segs + seg) * 64 + lane] = acc;
}

}   // namespace bigsi
