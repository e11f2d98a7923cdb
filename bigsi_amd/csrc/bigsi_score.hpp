// bigsi_score.hpp -- BIGSI.score's per-hit arithmetic (bigsi/scoring/score.py:7-107) as straight-line integer / IEEE-754
// double code, compiled twice: as device code inside K6 (k_score_packed, bigsi_kernels.hpp) and -- the SAME text -- as host
// C++ by the CPU test that pins it to CPython and to the reference's golden scores (tests/c_host/score_host.cpp).
//
// What runs here, per hit (reference lines in brackets):
//   presence bits of the n k-mer positions
//     -> remove_short_ones           [score.py:7-16]   3-wide erosion, two virtual 1s after the last position
//     -> tabulate_score              [score.py:19-32]  maximal runs; every run but the last is recorded one longer than it is
//     -> Scorer.calculate_score      [score.py:54-94]  three running scores, each update followed by Python's round(x, 2),
//                                                      SNP totals, math.ceil / math.floor
//     -> BigsiQueryResult's percent  [graph/bigsi.py:97-99]  round(100 * float(found) / num_kmers, 2)
// What stays on the host (numpy exp / log10 are not reproducible bit for bit on a GPU): evalue, pvalue, log_evalue,
// log_pvalue and the integer / quotient fields derived from the mismatch counts (bigsi_amd/scoring.py: score_fields).
//
// Exactness.  Every operation is an IEEE double +, -, *, /, rint or fma on values far below 2^53; the only subtle one is
// Python's round(x, 2), which rounds the EXACT binary value of x to two decimals (half to even) and returns the double
// nearest to that decimal (floatobject.c: double_round through dtoa / strtod).  py_round2 below does exactly that.
// Contraction into fused multiply-adds would change results: both builds compile this header with contraction off
// (#pragma clang fp contract(off) / -ffp-contract=off) and the one fma() that is meant is written out.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BIGSI_HD __host__ __device__ __forceinline__
#else
#define BIGSI_HD inline
#endif

namespace bigsi_score {

// Python's round(x, 2) for |x| < 2^44 (scores are bounded by the query length).
BIGSI_HD double py_round2(double x)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double p = x * 100.0;
    const double e = fma(x, 100.0, -p);       // x * 100 == p + e exactly
    double c = rint(p);                       // nearest integer, ties to even
    const double d = p - c;                   // exact
    // p is the double nearest to the exact product, so the two lie on the same side of every half-integer unless p IS one:
    // only then can rint(p) differ from the rounding of the exact value, and e says which side the exact value is on
    if (d == 0.5) { if (e > 0.0) c += 1.0; }
    else if (d == -0.5) { if (e < 0.0) c -= 1.0; }
    return c / 100.0;                         // correctly rounded quotient of two exact integers = strtod("c/100")
}

struct HitScore {                  // what K6 returns per hit (64 bytes); mirrored by bigsi_hip_hit_score in include/bigsi_hip.h
    double score, min_score, max_score;      // calculate_score's three rounded scores (score.py:86-88)
    double percent_kmers_found;              // graph/bigsi.py:97-99
    int64_t max_mismatches, min_mismatches, mismatches;      // score.py:89-93
    uint32_t num_kmers;                      // n = length of the presence string
    uint32_t reserved;
};

// Streaming state of calculate_score's loop over score_counter["0"] (score.py:62-84) with MATCH = 1, MISMATCH = 2,
// kmer_adjust = 3 (the Scorer every BIGSI object builds, graph/bigsi.py:140)
struct GapChain {
    double best, worst, mid;       // max_score, min_score, mean_score
    double most, least;            // max_total_N_snps, min_total_N_snps
    BIGSI_HD void start(uint64_t ones_total)
    {
        best = worst = mid = (double)ones_total;      // MATCH * sum(score_counter["1"])
        most = least = 0.0;
    }
    BIGSI_HD void gap(uint64_t i)
    {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
        const double fi = (double)i;
        const double snp_t = 34.0;                     // 31 + kmer_adjust
        const double lo = fi / snp_t;                  // min_N_snps
        double hi = (double)((int64_t)i - 34 + 1);     // max_N_snps
        if (hi < lo) hi = lo;
        most += hi;
        least += lo;
        const double typical = lo + 0.05 * hi;         // mean_N_snps
        const double pen_hi = 2.0 * hi, pen_lo = 2.0 * lo, pen_ty = 2.0 * typical;
        const double pts_hi = fi - pen_hi, pts_lo = fi - pen_lo, pts_ty = fi - pen_ty;
        best = py_round2((best - pen_lo) + pts_lo);
        worst = py_round2((worst - pen_hi) + pts_hi);
        mid = py_round2((mid - pen_ty) + pts_ty);
    }
    BIGSI_HD void finish(uint32_t n, HitScore *out) const
    {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
        const double seq_len = (double)n + 30.0;       // max_possible_score + 31 - 1
        const double convert = seq_len / (double)n;
        out->score = py_round2(mid * convert);
        out->min_score = py_round2(worst * convert);
        out->max_score = py_round2(best * convert);
        out->max_mismatches = (int64_t)ceil(most);
        out->min_mismatches = (int64_t)floor(least);
        out->mismatches = (int64_t)ceil(ceil(least) + 0.05 * floor(most));
    }
};

// word k (positions 64k .. 64k+63, position p at bit p % 64) of remove_short_ones(presence) given the presence words
// cur = word k and next = word k + 1 (anything when k is the last word); n = positions.  Bits at and beyond n are zero.
BIGSI_HD uint64_t eroded_word(uint64_t cur, uint64_t next, uint32_t k, uint32_t n)
{
    const uint32_t nw = (n + 63u) >> 6, tail = n & 63u;
    const uint64_t valid = (k + 1u < nw || tail == 0u) ? ~0ull : (1ull << tail) - 1ull;
    if (n < 3u) return cur & valid;                                   // score.py:9-10
    if (k + 1u >= nw) { cur |= ~valid; next = ~0ull; }                // the two virtual 1s (score.py:12-14)
    else if (k + 2u >= nw && tail) next |= ~((1ull << tail) - 1ull);
    return cur & ((cur >> 1) | (next << 63)) & ((cur >> 2) | (next << 62)) & valid;
}

// Score one hit from its presence words (LSB-first: position p at bit p % 64 of word p / 64; `word(k)` returns word k).
// found / unique feed percent_kmers_found.  n == 0 leaves zeros (the reference divides by zero there: the host raises).
template <typename WordFn>
BIGSI_HD void score_hit(WordFn word, uint32_t n, uint32_t found, uint32_t unique, HitScore *out)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    out->num_kmers = n;
    out->reserved = 0;
    out->percent_kmers_found = unique ? py_round2(100.0 * (double)found / (double)unique) : 0.0;
    if (n == 0) {
        out->score = out->min_score = out->max_score = 0.0;
        out->max_mismatches = out->min_mismatches = out->mismatches = 0;
        return;
    }
    const uint32_t nw = (n + 63u) >> 6;
    // pass 1: sum(score_counter["1"]) = set bits + number of 1-runs, minus one if the string ends inside a 1-run
    uint64_t ones = 0, starts = 0, prev_top = 0, last = 0;
    {
        uint64_t cur = word(0);
        for (uint32_t k = 0; k < nw; k++) {
            const uint64_t next = k + 1u < nw ? word(k + 1u) : 0ull;
            const uint64_t ss = eroded_word(cur, next, k, n);
            ones += (uint64_t)__builtin_popcountll(ss);
            starts += (uint64_t)__builtin_popcountll(ss & ~((ss << 1) | prev_top));
            prev_top = ss >> 63;
            last = ss;
            cur = next;
        }
    }
    const uint64_t ends_in_one = (last >> ((n - 1u) & 63u)) & 1ull;
    GapChain chain;
    chain.start(ones + starts - ends_in_one);
    // pass 2: the 0-runs in order
    uint64_t zlen = 0;
    bool in_zero = false;
    {
        uint64_t cur = word(0);
        for (uint32_t k = 0; k < nw; k++) {
            const uint64_t next = k + 1u < nw ? word(k + 1u) : 0ull;
            const uint64_t ss = eroded_word(cur, next, k, n);
            const uint32_t vb = (k + 1u < nw || (n & 63u) == 0u) ? 64u : (n & 63u);
            if (k == 0) in_zero = !(ss & 1ull);
            uint32_t p = 0;
            while (p < vb) {
                const uint64_t rest = ss >> p;
                if (in_zero) {
                    uint32_t z = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
                    if (z > vb - p) z = vb - p;
                    zlen += z;
                    p += z;
                    if (p < vb) { chain.gap(zlen + 1); zlen = 0; in_zero = false; }      // a run that is not the last: one longer
                } else {
                    const uint64_t inv = ~rest;
                    uint32_t o = inv ? (uint32_t)__builtin_ctzll(inv) : 64u;
                    if (o > vb - p) o = vb - p;
                    p += o;
                    if (p < vb) in_zero = true;
                }
            }
            cur = next;
        }
    }
    if (in_zero && zlen) chain.gap(zlen);              // the string ends inside a 0-run: recorded as it is
    chain.finish(n, out);
}

// presence bytes in the reference's row / bitarray order (position p in byte p / 8 under mask 0x80 >> (p % 8)), read as a
// little-endian uint64 -> LSB-first word
BIGSI_HD uint64_t lsb_first(uint64_t packed)
{
    uint64_t x = packed;
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return x;
}

}  // namespace bigsi_score
