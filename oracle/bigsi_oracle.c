/*
 * bigsi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the BIGSI query hot path (k-merise -> canonical ->
 * MurmurHash3 -> h row fetches -> AND across h -> combine across k-mers), used only
 * as the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under bigsi_amd/ may import, link or call this file.
 *
 * Parity pin: every function here is checked by tests/test_oracle_golden.py against
 * golden vectors produced by RUNNING the unmodified reference (tests/golden/make_golden.py).
 *
 * Each function cites the reference lines (relative to /root/reference) it restates.
 * MurmurHash3_x86_32 lives in the third-party `mmh3` wheel (hajimes/mmh3, pinned 2.5.1 in
 * .conda/mmh3/meta.yaml:2), not in the reference tree; it is restated from Austin Appleby's
 * public-domain algorithm description and anchored on the reference's call site
 * (bigsi/bloom/bloomfilter.py:5-6) and known answers (bigsi/tests/bloom/test_create_bloomfilter.py:6-8).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ hashing */

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* mmh3.hash(key, seed) as used at bloom/bloomfilter.py:6 (MurmurHash3_x86_32, returned signed). */
int32_t orc_mmh3_hash(const uint8_t *key, uint64_t len, uint32_t seed)
{
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
    uint64_t nblocks = len / 4;
    for (uint64_t i = 0; i < nblocks; i++) {
        uint32_t k1 = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) |
                      ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
        k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
        h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64u;
    }
    const uint8_t *tail = key + nblocks * 4;
    uint32_t k1 = 0;
    switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1: k1 ^= tail[0];
            k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint32_t)len;
    h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
    return (int32_t)h1;
}

/* _hash(element, seed, m) = mmh3.hash(element, seed) % m with Python floor-mod
 * (bloom/bloomfilter.py:5-6): result in [0, m) also for negative hashes. */
uint64_t orc_row_of(const uint8_t *canon, uint64_t k, uint32_t seed, uint64_t m)
{
    int64_t h = (int64_t)orc_mmh3_hash(canon, k, seed);
    if (h >= 0) return (uint64_t)h % m;
    uint64_t a = (uint64_t)(-h) % m;          /* |h| mod m */
    return a == 0 ? 0 : m - a;
}

/* reverse_comp (utils/fncts.py:12,38-39): reversed, A<->T C<->G, anything else unchanged. */
void orc_reverse_comp(const uint8_t *s, uint64_t k, uint8_t *out)
{
    for (uint64_t j = 0; j < k; j++) {
        uint8_t c = s[k - 1 - j];
        switch (c) {
        case 'A': c = 'T'; break;
        case 'T': c = 'A'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        default: break;
        }
        out[j] = c;
    }
}

/* canonical (utils/fncts.py:51-54): lexicographic min of the k-mer and its reverse complement. */
void orc_canonical(const uint8_t *s, uint64_t k, uint8_t *out)
{
    orc_reverse_comp(s, k, out);
    if (memcmp(s, out, k) <= 0) memcpy(out, s, k);
}

/* generate_hashes in seed order (bloom/bloomfilter.py:9-13; the caller's set() is order-free). */
void orc_kmer_rows(const uint8_t *kmer, uint64_t k, uint32_t h, uint64_t m, uint64_t *rows_out)
{
    uint8_t stackbuf[256];
    uint8_t *canon = k <= sizeof stackbuf ? stackbuf : (uint8_t *)malloc(k);
    orc_canonical(kmer, k, canon);   /* graph/index.py:64-69: hash the canonical k-mer */
    for (uint32_t s = 0; s < h; s++) rows_out[s] = orc_row_of(canon, k, s, m);
    if (canon != stackbuf) free(canon);
}

/* the same for every k-mer position of a sequence (seq_to_kmers, utils/fncts.py:63-65): rows_out[i*h + s] */
void orc_seq_rows(const uint8_t *seq, uint64_t len, uint64_t k, uint32_t h, uint64_t m, uint64_t *rows_out)
{
    if (len < k) return;
    for (uint64_t i = 0; i + k <= len; i++) orc_kmer_rows(seq + i, k, h, m, rows_out + i * h);
}

/* ----------------------------------------------------- unique query k-mers */

typedef struct { const uint8_t *seq; uint64_t k; } cmp_ctx;
static cmp_ctx g_ctx;   /* single-threaded checker: plain qsort with a static context */

static int cmp_pos(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    int c = memcmp(g_ctx.seq + x, g_ctx.seq + y, g_ctx.k);
    if (c) return c;
    return x < y ? -1 : (x > y);
}

/* set(kmers) of seq_to_kmers(seq, k) (utils/fncts.py:63-65, graph/index.py:45): unique *query strings*.
 * first_pos[j] = position of the j-th unique k-mer, in first-occurrence order;
 * pos_to_unique[i] = index into first_pos of the k-mer at position i.  Returns u. */
uint32_t orc_unique_kmers(const uint8_t *seq, uint64_t len, uint64_t k, uint32_t *first_pos, uint32_t *pos_to_unique)
{
    if (len < k) return 0;
    uint32_t n = (uint32_t)(len - k + 1);
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint32_t *rep = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    g_ctx.seq = seq; g_ctx.k = k;
    qsort(idx, n, sizeof(uint32_t), cmp_pos);
    for (uint32_t a = 0; a < n;) {
        uint32_t b = a + 1;
        while (b < n && memcmp(seq + idx[a], seq + idx[b], k) == 0) b++;
        for (uint32_t c = a; c < b; c++) rep[idx[c]] = idx[a];   /* idx[a] = smallest position of the class */
        a = b;
    }
    uint32_t u = 0;
    for (uint32_t i = 0; i < n; i++)
        if (rep[i] == i) { first_pos[u] = i; pos_to_unique[i] = u; u++; }
    for (uint32_t i = 0; i < n; i++)
        if (rep[i] != i) pos_to_unique[i] = pos_to_unique[rep[i]];
    free(idx); free(rep);
    return u;
}

/* ------------------------------------------- rows in the reference's format
 * A row is ceil(N/8) bytes, column c at byte c/8, mask 0x80 >> (c%8), zero pad bits
 * (bitarray.tobytes(), storage/base.py:85-99).  `index` is m rows of rb bytes. */

/* __bitwise_and_kmers for one k-mer (graph/index.py:75-80, utils/fncts.py:24-25): AND of its h rows. */
void orc_and_rows(const uint8_t *index, uint64_t rb, const uint64_t *rows, uint32_t h, uint8_t *out)
{
    memcpy(out, index + rows[0] * rb, rb);            /* load_bitarray copy, storage/base.py:96-99 */
    for (uint32_t s = 1; s < h; s++) {
        const uint8_t *r = index + rows[s] * rb;
        for (uint64_t b = 0; b < rb; b++) out[b] &= r[b];
    }
}

/* KmerSignatureIndex.lookup for u k-mers given as u*k ASCII (graph/index.py:42-49): out = u rows of rb bytes. */
void orc_lookup(const uint8_t *index, uint64_t m, uint64_t rb, uint32_t h,
                const uint8_t *kmers, uint64_t u, uint64_t k, uint8_t *out)
{
    uint64_t *rows = (uint64_t *)malloc(sizeof(uint64_t) * h);
    for (uint64_t j = 0; j < u; j++) {
        orc_kmer_rows(kmers + j * k, k, h, m, rows);
        orc_and_rows(index, rb, rows, h, out + j * rb);
    }
    free(rows);
}

/* unpack_and_sum (graph/bigsi.py:35-44): one byte per bit -> int32 adds.  counts has 8*rb entries. */
void orc_unpack_and_sum(const uint8_t *rows, uint64_t u, uint64_t rb, int32_t *counts)
{
    memset(counts, 0, sizeof(int32_t) * 8 * rb);
    for (uint64_t j = 0; j < u; j++) {
        const uint8_t *r = rows + j * rb;
        for (uint64_t b = 0; b < rb; b++) {
            uint8_t v = r[b];
            int32_t *c = counts + 8 * b;
            c[0] += (v >> 7) & 1; c[1] += (v >> 6) & 1; c[2] += (v >> 5) & 1; c[3] += (v >> 4) & 1;
            c[4] += (v >> 3) & 1; c[5] += (v >> 2) & 1; c[6] += (v >> 1) & 1; c[7] += v & 1;
        }
    }
}

/* exact_filter's reduce (graph/bigsi.py:192-195): AND of all u per-k-mer rows. */
void orc_and_all(const uint8_t *rows, uint64_t u, uint64_t rb, uint8_t *out)
{
    memcpy(out, rows, rb);
    for (uint64_t j = 1; j < u; j++)
        for (uint64_t b = 0; b < rb; b++) out[b] &= rows[j * rb + b];
}

/* One whole reference-shaped query against an in-RAM index: unique k-mers -> lookup -> counts.
 * Returns u.  counts (8*rb int32) and and_all (rb bytes) may be NULL.  scratch_rows: >= n*rb bytes. */
uint32_t orc_query(const uint8_t *index, uint64_t m, uint64_t rb, uint32_t h,
                   const uint8_t *seq, uint64_t len, uint64_t k,
                   uint8_t *scratch_rows, int32_t *counts, uint8_t *and_all)
{
    if (len < k) return 0;
    uint32_t n = (uint32_t)(len - k + 1);
    uint32_t *first = (uint32_t *)malloc(sizeof(uint32_t) * n), *p2u = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint32_t u = orc_unique_kmers(seq, len, k, first, p2u);
    uint64_t *rows = (uint64_t *)malloc(sizeof(uint64_t) * h);
    for (uint32_t j = 0; j < u; j++) {
        orc_kmer_rows(seq + first[j], k, h, m, rows);
        orc_and_rows(index, rb, rows, h, scratch_rows + (uint64_t)j * rb);
    }
    if (counts) orc_unpack_and_sum(scratch_rows, u, rb, counts);
    if (and_all && u) orc_and_all(scratch_rows, u, rb, and_all);
    free(rows); free(first); free(p2u);
    return u;
}

/* ------------------------------------------------------- synthetic index
 * NOT from the reference: the seeded test-input generator shared with the HIP fill kernel
 * (bigsi_amd/csrc/bigsi_hip.hip: synth_word), so that any row of a 100+ GB device index can be
 * recomputed on the host.  Word w of a row holds columns [64w, 64w+64) in the reference byte
 * order when stored little-endian. */

static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint64_t orc_synth_word(uint64_t seed, uint64_t shard, uint64_t row, uint64_t word, uint32_t and_draws)
{
    uint64_t base = mix64(seed + shard * 0x632BE59BD9B4E019ull);
    uint64_t rk = mix64(base ^ (row * 0x9E3779B97F4A7C15ull));
    uint64_t v = ~0ull;
    for (uint32_t d = 0; d < and_draws; d++)
        v &= mix64(rk + (word * 8 + d) * 0xD1B54A32D192ED03ull);
    return v;
}

/* mask of valid column bits of word `word` for an index of n_cols columns (pad bits are zero). */
uint64_t orc_valid_mask(uint64_t word, uint64_t n_cols)
{
    uint64_t mask = 0;
    for (int b = 0; b < 8; b++) {
        uint64_t c0 = word * 64 + 8 * (uint64_t)b;
        uint64_t n = c0 >= n_cols ? 0 : (n_cols - c0 >= 8 ? 8 : n_cols - c0);
        uint64_t bm = (0xFFu << (8 - n)) & 0xFFu;
        mask |= bm << (8 * b);
    }
    return mask;
}

/* row bytes (rb = ceil(n_cols/8)) of the synthetic index. */
void orc_synth_row(uint64_t seed, uint64_t shard, uint64_t row, uint64_t n_cols, uint32_t and_draws, uint8_t *out)
{
    uint64_t rb = (n_cols + 7) / 8, words = (n_cols + 63) / 64;
    for (uint64_t w = 0; w < words; w++) {
        uint64_t v = orc_synth_word(seed, shard, row, w, and_draws) & orc_valid_mask(w, n_cols);
        for (int b = 0; b < 8 && w * 8 + b < rb; b++) out[w * 8 + b] = (uint8_t)(v >> (8 * b));
    }
}

/* rows [row0, row0+n) of a column slice [0, n_cols) into a dense table (cpu_baseline index). */
void orc_synth_fill(uint64_t seed, uint64_t shard, uint64_t row0, uint64_t n, uint64_t n_cols, uint32_t and_draws, uint8_t *out)
{
    uint64_t rb = (n_cols + 7) / 8;
    for (uint64_t r = 0; r < n; r++) orc_synth_row(seed, shard, row0 + r, n_cols, and_draws, out + r * rb);
}
