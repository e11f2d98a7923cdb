// CPU pin of bigsi_amd/csrc/bigsi_score.hpp: the header K6 (k_score_packed) compiles for the device, compiled here as host
// C++ with contraction off.  tests/test_abi_and_host.py loads it and compares
//   py_round2      with CPython's round(x, 2) on adversarial doubles (half-way cases, products that round onto .5),
//   score_packed   with the reference's golden scores (tests/golden/g5_scoring.json) and with bigsi_amd.scoring.Scorer.
// Test infrastructure only: nothing in the product loads this file.
#include "../../bigsi_amd/csrc/bigsi_score.hpp"

extern "C" {

void score_host_round2(const double *x, uint64_t n, double *out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = bigsi_score::py_round2(x[i]);
}

// same layout as bigsi_hip_score_presence: string t = num_kmers[t] positions at bits + bit_offsets[t] (multiple of 8),
// position p in byte p / 8 under mask 0x80 >> (p % 8)
void score_host_packed(const uint8_t *bits, const uint64_t *bit_offsets, const uint32_t *num_kmers, const uint32_t *found,
                       const uint32_t *unique, uint64_t n, bigsi_score::HitScore *out)
{
    for (uint64_t t = 0; t < n; t++) {
        const uint64_t *w = reinterpret_cast<const uint64_t *>(bits + bit_offsets[t]);
        bigsi_score::score_hit([w](uint32_t k) { return bigsi_score::lsb_first(w[k]); }, num_kmers[t], found ? found[t] : 0u,
                               unique ? unique[t] : 0u, &out[t]);
    }
}

}
