// bigsi_cpu.cpp -- libbigsi_cpu.so: the CPU twin of the CORE layer of libbigsi_hip.so (include/bigsi_cpu.h).
//
// The BIGSI query path computed on the host in the reference's own shape, behind the same C boundary: a host without a GPU binds
// it explicitly, bench.py times it as the CPU baseline THROUGH the product boundary, and tests/c_host/search_host.c builds
// against either library.  The hip-hbm backend never loads this file: there is no CPU fallback.
//
// Default search path = what the reference does per query (file:line relative to the reference root):
//   seq_to_kmers                  utils/fncts.py:63-65     every k-window of the raw bytes
//   set(kmers)                    graph/index.py:45        unique query k-mers (first-occurrence order kept for presence)
//   canonical                     utils/fncts.py:38-54     reverse complement as a new string, lexicographic min
//   generate_hashes               bloom/bloomfilter.py:5-13 mmh3 (MurmurHash3_x86_32, signed) % m with Python's floor-mod, h seeds
//   batch_get + frombytes         storage/base.py:58-59,96-109  the union of rows fetched ONCE, each COPIED out of the store
//   bitwise_and                   utils/fncts.py:24-25     a fresh buffer per AND step
//   exact_filter                  graph/bigsi.py:192-205   AND of all k-mer rows, set bits ascending
//   unpack_and_sum + threshold    graph/bigsi.py:35-44,211-230,241-242  one int32 per bit, added; count >= ceil(u * threshold)
// BIGSI_CPU_WORD_PARALLEL replaces the last four by 64-bit word operations on the resident rows (same results).
// Row format = the reference's bitarray.tobytes(): column c at byte c / 8 under mask 0x80 >> (c % 8).
#include "bigsi_cpu.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <mutex>

#include "../csrc/bigsi_bdb.hpp"
#include "../csrc/bigsi_score.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_ != BIGSI_OK) \
            return rc_;      \
    } while (0)

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline uint64_t round_up(uint64_t a, uint64_t b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------------------------- hashing
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// MurmurHash3_x86_32 (Austin Appleby, public domain): what mmh3.hash(str, seed) computes over the UTF-8 bytes
uint32_t murmur3(const uint8_t *p, size_t len, uint32_t seed)
{
    uint32_t h = seed;
    const size_t nb = len / 4;
    for (size_t i = 0; i < nb; i++) {
        uint32_t k;
        memcpy(&k, p + 4 * i, 4);             // little-endian host
        k *= 0xcc9e2d51u; k = rotl(k, 15); k *= 0x1b873593u;
        h ^= k; h = rotl(h, 13); h = h * 5u + 0xe6546b64u;
    }
    uint32_t k = 0;
    const uint8_t *t = p + 4 * nb;
    const size_t rem = len & 3;
    if (rem == 3) k ^= (uint32_t)t[2] << 16;
    if (rem >= 2) k ^= (uint32_t)t[1] << 8;
    if (rem >= 1) { k ^= t[0]; k *= 0xcc9e2d51u; k = rotl(k, 15); k *= 0x1b873593u; h ^= k; }
    h ^= (uint32_t)len;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// Python: mmh3.hash(...) % m  (signed hash, result in [0, m))
uint64_t floor_mod_row(uint32_t hash, uint64_t m)
{
    const int64_t s = (int32_t)hash;
    const int64_t r = s % (int64_t)m;       // C: sign of the dividend
    return (uint64_t)(r < 0 ? r + (int64_t)m : r);
}

inline char comp(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    default: return c;              // COMPLEMENT.get(base, base): lowercase, N, anything else unchanged
    }
}

// canonical(k-mer) into `out` (a std::string reused by the caller)
void canonical(const char *s, uint32_t k, std::string &out)
{
    out.resize(k);
    for (uint32_t i = 0; i < k; i++) out[i] = comp(s[k - 1 - i]);
    if (memcmp(s, out.data(), k) <= 0) out.assign(s, k);          // sorted([kmer, rc])[0]
}

// ---------------------------------------------------------------------------------------------- synthetic contents
// the generator of bigsi_hip_fill_synthetic (csrc/bigsi_kernels.hpp: k_fill_synth), restated: word w of row r is the AND of
// `draws` values of a counter-based hash of (seed, shard, r, w), pad bits cleared
inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint64_t valid_word_mask(uint64_t w, uint64_t n_cols)
{
    const uint64_t c0 = w * 64;
    if (c0 + 64 <= n_cols) return ~0ull;
    if (c0 >= n_cols) return 0;
    uint64_t mask = 0;
    for (int b = 0; b < 8; b++) {
        const uint64_t cb = c0 + 8ull * b, left = cb >= n_cols ? 0 : std::min<uint64_t>(n_cols - cb, 8);
        mask |= (uint64_t)((0xFFu << (8 - left)) & 0xFFu) << (8 * b);
    }
    return mask;
}

}  // namespace

// rows served from a BerkeleyDB hash FILE (bigsi_cpu_open_bdb) instead of a table in RAM: where every "<row>:bitarray" record lies
struct BdbRows {
    int fd = -1;
    BigsiBdb db;
    std::vector<BigsiBdb::Loc> loc;
    // errno of the first record read that failed since the last check (an I/O error, a corrupt overflow chain): the entry point that
    // was reading fails with it instead of answering from a zero / partial row (bdb_read_failed below; fork-pool workers have their own copy)
    mutable std::atomic<int> io_error{0};
    void note(int e) const { int none = 0; if (e) io_error.compare_exchange_strong(none, e); }
};

struct bigsi_cpu_index {
    uint64_t m = 0, n_cols = 0, cap_cols = 0, stride = 0;      // stride: bytes per row, a multiple of 128 like the device's pitch
    uint32_t h = 0;
    uint8_t *rows = nullptr;
    BdbRows *bdb = nullptr;      // non-null: no table in RAM, every row fetch reads the store's file (read-only index)
    // row r's first rb bytes: the table's own memory, or -- BerkeleyDB store -- read into `tmp` (zero-extended), as a KV store's get does
    const uint8_t *fetch(uint64_t r, uint64_t rb_, std::vector<uint8_t> &tmp, std::vector<uint8_t> &page) const
    {
        if (!bdb) return rows + r * stride;
        tmp.assign(rb_, 0);
        const BigsiBdb::Loc &l = bdb->loc[r];
        if (l.kind) bdb->note(bdb->db.read_value(l, tmp.data(), (uint32_t)std::min<uint64_t>(l.len, rb_), page));
        return tmp.data();
    }
    uint64_t rb() const { return ceil_div(n_cols, 8); }
    uint8_t *row(uint64_t r) { return rows + r * stride; }
    const uint8_t *row(uint64_t r) const { return rows + r * stride; }
};

namespace {

uint64_t stride_for(uint64_t cols) { return std::max<uint64_t>(128, round_up(ceil_div(cols, 64) * 8, 128)); }

// the result of an entry point that read rows: BIGSI_OK, or -- a BerkeleyDB-backed index whose file could not be read -- the failure
int bdb_read_failed(const bigsi_cpu_index *ix)
{
    if (!ix->bdb) return BIGSI_OK;
    const int e = ix->bdb->io_error.exchange(0);
    return e ? fail(BIGSI_ERR_INVALID, "reading a record of the BerkeleyDB store failed: %s%s%s", strerror(e), ix->bdb->db.error.empty() ? "" : ": ",
                    ix->bdb->db.error.c_str())
             : BIGSI_OK;
}

int check_seqs(const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k)
{
    if (!offsets) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (n_seqs == 0) return fail(BIGSI_ERR_INVALID, "a batch needs at least one sequence");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    for (uint64_t i = 0; i < n_seqs; i++)
        if (offsets[i + 1] < offsets[i]) return fail(BIGSI_ERR_INVALID, "offsets must be non-decreasing");
    if (offsets[n_seqs] - offsets[0] && !seqs) return fail(BIGSI_ERR_INVALID, "seqs is NULL");
    return BIGSI_OK;
}

// the h rows of one (canonical) element, seeds 0 .. h-1
inline void rows_of(const std::string &canon, uint32_t h, uint64_t m, uint64_t *out)
{
    for (uint32_t s = 0; s < h; s++) out[s] = floor_mod_row(murmur3(reinterpret_cast<const uint8_t *>(canon.data()), canon.size(), s), m);
}

struct QueryKmers {
    std::vector<uint32_t> first_pos;       // unique k-mers in first-occurrence order: position of the first occurrence
    std::vector<uint32_t> pos_unique;      // every position -> index of its unique k-mer
};

void unique_kmers(const char *s, uint64_t len, uint32_t k, QueryKmers &q)
{
    q.first_pos.clear();
    q.pos_unique.clear();
    if (len < k) return;
    const uint64_t n = len - k + 1;
    std::unordered_map<std::string_view, uint32_t> seen;
    seen.reserve(n * 2);
    q.pos_unique.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        auto it = seen.emplace(std::string_view(s + i, k), (uint32_t)q.first_pos.size());
        if (it.second) q.first_pos.push_back((uint32_t)i);
        q.pos_unique[i] = it.first->second;
    }
}

struct Hits {
    uint64_t *off;
    uint32_t *col, *cnt;
    uint64_t cap, total = 0;
    void push(uint32_t c, uint32_t n)
    {
        if (total < cap) {
            if (col) col[total] = c;
            if (cnt) cnt[total] = n;
        }
        total++;
    }
};

// one sequence, the reference's shape
void search_reference_shaped(const bigsi_cpu_index *ix, const char *s, uint64_t len, uint32_t k, double threshold, QueryKmers &q,
                             uint32_t *nk, uint32_t *nu, uint32_t *mk, Hits &hits)
{
    unique_kmers(s, len, k, q);
    const uint32_t u = (uint32_t)q.first_pos.size(), h = ix->h;
    const uint64_t rb = ix->rb(), n_cols = ix->n_cols;
    *nk = (uint32_t)q.pos_unique.size();
    *nu = u;
    const uint32_t min_kmers = (uint32_t)ceil((double)u * threshold);        // math.ceil(len(set(kmers)) * threshold), graph/bigsi.py:179
    *mk = min_kmers;
    const bool exact = threshold == 1.0;
    if (u == 0) {
        if (!exact)                                 // counts are all zero and min_kmers is 0: every sample passes `count >= 0`
            for (uint64_t c = 0; c < n_cols; c++) hits.push((uint32_t)c, 0);
        return;
    }
    // {k-mer -> set(rows)}, the union of rows fetched once and copied out of the store (batch_get + frombytes)
    std::vector<uint64_t> kmer_rows((size_t)u * h);
    std::unordered_map<uint64_t, uint32_t> fetched;
    std::vector<std::vector<uint8_t>> row_copy;
    std::vector<uint8_t> bdb_page;
    std::string canon;
    for (uint32_t j = 0; j < u; j++) {
        canonical(s + q.first_pos[j], k, canon);
        rows_of(canon, h, ix->m, &kmer_rows[(size_t)j * h]);
        for (uint32_t t = 0; t < h; t++) {
            const uint64_t r = kmer_rows[(size_t)j * h + t];
            if (fetched.emplace(r, (uint32_t)row_copy.size()).second) {
                if (ix->bdb) {          // the store's get(): the record's bytes read from the file (an overflow chain of pages for wide rows)
                    row_copy.emplace_back(rb, 0);
                    const BigsiBdb::Loc &l = ix->bdb->loc[r];
                    if (l.kind) ix->bdb->note(ix->bdb->db.read_value(l, row_copy.back().data(), (uint32_t)std::min<uint64_t>(l.len, rb), bdb_page));
                } else row_copy.emplace_back(ix->row(r), ix->row(r) + rb);
            }
        }
    }
    std::vector<uint8_t> all;                        // exact: AND of every k-mer's row
    std::vector<int32_t> sums;                       // inexact: one int32 per bit
    if (!exact) sums.assign(rb * 8, 0);
    for (uint32_t j = 0; j < u; j++) {
        std::vector<uint8_t> acc(row_copy[fetched[kmer_rows[(size_t)j * h]]]);               // reduce(x & y): a new bitarray per step
        for (uint32_t t = 1; t < h; t++) {
            const std::vector<uint8_t> &other = row_copy[fetched[kmer_rows[(size_t)j * h + t]]];
            std::vector<uint8_t> next(rb);
            for (uint64_t b = 0; b < rb; b++) next[b] = acc[b] & other[b];
            acc.swap(next);
        }
        if (exact) {
            if (j == 0) all = acc;
            else {
                std::vector<uint8_t> next(rb);
                for (uint64_t b = 0; b < rb; b++) next[b] = all[b] & acc[b];
                all.swap(next);
            }
        } else {
            // unpack(one byte per bit) -> int32 -> add (graph/bigsi.py:35-44)
            std::vector<int32_t> unpacked(rb * 8);
            for (uint64_t b = 0; b < rb; b++)
                for (int bit = 0; bit < 8; bit++) unpacked[b * 8 + bit] = (acc[b] >> (7 - bit)) & 1;
            for (uint64_t c = 0; c < rb * 8; c++) sums[c] += unpacked[c];
        }
    }
    if (exact) {
        for (uint64_t c = 0; c < n_cols; c++)
            if (all[c >> 3] & (0x80u >> (c & 7))) hits.push((uint32_t)c, u);
    } else {
        for (uint64_t c = 0; c < n_cols; c++)
            if ((uint32_t)sums[c] >= min_kmers) hits.push((uint32_t)c, (uint32_t)sums[c]);
    }
}

// the same results on 64-bit words of the resident rows: no copies, counters touched only where bits are set
void search_word_parallel(const bigsi_cpu_index *ix, const char *s, uint64_t len, uint32_t k, double threshold, QueryKmers &q,
                          uint32_t *nk, uint32_t *nu, uint32_t *mk, Hits &hits)
{
    unique_kmers(s, len, k, q);
    const uint32_t u = (uint32_t)q.first_pos.size(), h = ix->h;
    const uint64_t n_cols = ix->n_cols, words = ceil_div(n_cols, 64);
    *nk = (uint32_t)q.pos_unique.size();
    *nu = u;
    const uint32_t min_kmers = (uint32_t)ceil((double)u * threshold);
    *mk = min_kmers;
    const bool exact = threshold == 1.0;
    if (u == 0) {
        if (!exact)
            for (uint64_t c = 0; c < n_cols; c++) hits.push((uint32_t)c, 0);
        return;
    }
    std::vector<uint64_t> kr((size_t)u * h);
    std::string canon;
    for (uint32_t j = 0; j < u; j++) {
        canonical(s + q.first_pos[j], k, canon);
        rows_of(canon, h, ix->m, &kr[(size_t)j * h]);
    }
    auto word = [&](uint64_t r, uint64_t w) { uint64_t v; memcpy(&v, ix->row(r) + 8 * w, 8); return v; };
    auto col_of_bit = [](uint64_t w, int bit) { return w * 64 + (uint64_t)(bit >> 3) * 8 + 7 - (bit & 7); };      // little-endian load of row bytes
    std::vector<uint32_t> sums;
    if (!exact) sums.assign(words * 64, 0);
    std::vector<uint64_t> all(exact ? words : 0, ~0ull);
    for (uint32_t j = 0; j < u; j++)
        for (uint64_t w = 0; w < words; w++) {
            uint64_t a = word(kr[(size_t)j * h], w);
            for (uint32_t t = 1; t < h; t++) a &= word(kr[(size_t)j * h + t], w);
            if (exact) all[w] &= a;
            else
                while (a) {
                    const int bit = __builtin_ctzll(a);
                    a &= a - 1;
                    sums[col_of_bit(w, bit)]++;
                }
        }
    if (exact) {
        for (uint64_t c = 0; c < n_cols; c++) {
            const uint64_t w = c >> 6;
            const int bit = (int)(((c >> 3) & 7) * 8 + 7 - (c & 7));
            if ((all[w] >> bit) & 1) hits.push((uint32_t)c, u);
        }
    } else {
        for (uint64_t c = 0; c < n_cols; c++)
            if (sums[c] >= min_kmers) hits.push((uint32_t)c, sums[c]);
    }
}

}  // namespace

extern "C" {

const char *bigsi_cpu_last_error(void) { return g_err; }

int bigsi_cpu_device_count(int *out)
{
    if (!out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    *out = 1;
    return BIGSI_OK;
}

int bigsi_cpu_open(uint64_t num_rows, uint64_t num_cols, uint64_t col_capacity, uint32_t num_hashes, int, bigsi_cpu_index **out)
{
    if (!out) return fail(BIGSI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_rows == 0) return fail(BIGSI_ERR_INVALID, "num_rows must be > 0");
    if (num_hashes == 0) return fail(BIGSI_ERR_INVALID, "num_hashes must be > 0");
    col_capacity = std::max<uint64_t>(std::max(col_capacity, num_cols), 1);
    bigsi_cpu_index *ix = new (std::nothrow) bigsi_cpu_index();
    if (!ix) return fail(BIGSI_ERR_NOMEM, "host allocation failed");
    ix->m = num_rows;
    ix->n_cols = num_cols;
    ix->h = num_hashes;
    ix->stride = stride_for(col_capacity);
    ix->cap_cols = ix->stride * 8;
    ix->rows = static_cast<uint8_t *>(calloc(ix->m, ix->stride));
    if (!ix->rows) { delete ix; return fail(BIGSI_ERR_NOMEM, "cannot allocate %llu x %llu bytes", (unsigned long long)num_rows, (unsigned long long)ix->stride); }
    *out = ix;
    return BIGSI_OK;
}

int bigsi_cpu_close(bigsi_cpu_index *ix)
{
    if (!ix) return BIGSI_OK;
    free(ix->rows);
    if (ix->bdb) {
        if (ix->bdb->fd >= 0) close(ix->bdb->fd);
        delete ix->bdb;
    }
    delete ix;
    return BIGSI_OK;
}

// An index whose rows stay in the reference's own store: a BerkeleyDB HASH file (bigsi/storage/berkeleydb.py:6-19) holding the
// "<row>:bitarray" records and the index integers of a v0.3 BIGSI index (bigsi/storage/base.py:29-36).  Nothing is loaded: every
// row a search needs is read from the file when it is needed -- located by a table built with one scan of the hash pages (libdb
// finds a record by hashing its key to a bucket page; the table stands in for that), then its bytes gathered from the page, or the
// overflow chain of pages, they lie in.  The closest this package comes to the reference's "berkeleydb / CPU path" without libdb
// (neither bsddb3 nor the db.h headers exist on these hosts): reference-shaped search, lookup, get_rows and presence only.
int bigsi_cpu_open_bdb(const char *path, uint32_t threads, bigsi_cpu_index **out)
{
    if (!path || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(BIGSI_ERR_INVALID, "%s: %s", path, strerror(errno));
    BdbRows *b = new (std::nothrow) BdbRows();
    bigsi_cpu_index *ix = new (std::nothrow) bigsi_cpu_index();
    if (!b || !ix) { close(fd); delete b; delete ix; return fail(BIGSI_ERR_NOMEM, "host allocation failed"); }
    b->fd = fd;
    ix->bdb = b;
    auto bail = [&](int code, const std::string &msg) { bigsi_cpu_close(ix); return fail(code, "%s: %s", path, msg.c_str()); };
    if (b->db.open_fd(fd)) return bail(BIGSI_ERR_INVALID, b->db.error);
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 4));
    // the small records first (number_of_rows, number_of_cols, ksi:num_hashes), the row locations in the same pass
    std::mutex mu;
    std::vector<std::pair<uint64_t, BigsiBdb::Loc>> rows;
    std::vector<std::pair<std::string, BigsiBdb::Loc>> small;
    if (b->db.scan(threads, [&](unsigned, const uint8_t *key, uint32_t klen, const BigsiBdb::Loc &l) {
            uint64_t r;
            std::lock_guard<std::mutex> g(mu);
            if (bigsi_bdb_row_key(key, klen, &r)) rows.emplace_back(r, l);
            else if (klen < 64) small.emplace_back(std::string(reinterpret_cast<const char *>(key), klen), l);
        }))
        return bail(BIGSI_ERR_INVALID, b->db.error);
    auto integer = [&](const char *key, uint64_t *v) -> bool {
        std::vector<uint8_t> page;
        for (auto &s_ : small)
            if (s_.first == key && s_.second.len < 32) {
                char buf[32] = {0};
                if (b->db.read_value(s_.second, reinterpret_cast<uint8_t *>(buf), s_.second.len, page)) return false;
                char *end = nullptr;
                *v = strtoull(buf, &end, 10);
                return end != buf;
            }
        return false;
    };
    uint64_t m = 0, n = 0, h = 0;
    if (!integer("number_of_rows:int", &m) || !integer("number_of_cols:int", &n) || !integer("ksi:num_hashes:int", &h) || m == 0 || h == 0)
        return bail(BIGSI_ERR_INVALID, "no number_of_rows / number_of_cols / ksi:num_hashes records: not a BIGSI v0.3 index");
    ix->m = m;
    ix->n_cols = n;
    ix->h = (uint32_t)h;
    ix->stride = stride_for(std::max<uint64_t>(n, 1));
    ix->cap_cols = ix->stride * 8;
    b->loc.assign(m, BigsiBdb::Loc{});
    for (auto &e : rows)
        if (e.first < m) b->loc[e.first] = e.second;
    *out = ix;
    return BIGSI_OK;
}

int bigsi_cpu_get_info(const bigsi_cpu_index *ix, bigsi_hip_info *out)
{
    if (!ix || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    out->num_rows = ix->m;
    out->num_cols = ix->n_cols;
    out->col_capacity = ix->cap_cols;
    out->row_bytes = ix->rb();
    out->row_stride_bytes = ix->stride;
    out->index_bytes = ix->m * ix->stride;
    out->num_hashes = ix->h;
    out->device = -1;
    return BIGSI_OK;
}

int bigsi_cpu_set_num_cols(bigsi_cpu_index *ix, uint64_t num_cols)
{
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (num_cols > ix->cap_cols) return fail(BIGSI_ERR_CAPACITY, "num_cols %llu exceeds col_capacity %llu", (unsigned long long)num_cols, (unsigned long long)ix->cap_cols);
    ix->n_cols = num_cols;
    return BIGSI_OK;
}

int bigsi_cpu_set_num_hashes(bigsi_cpu_index *ix, uint32_t num_hashes)
{
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (num_hashes == 0) return fail(BIGSI_ERR_INVALID, "num_hashes must be > 0");
    ix->h = num_hashes;
    return BIGSI_OK;
}

int bigsi_cpu_reserve_cols(bigsi_cpu_index *ix, uint64_t col_capacity)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (col_capacity <= ix->cap_cols) return BIGSI_OK;
    const uint64_t stride = stride_for(col_capacity);
    uint8_t *grown = static_cast<uint8_t *>(calloc(ix->m, stride));
    if (!grown) return fail(BIGSI_ERR_NOMEM, "cannot allocate %llu x %llu bytes", (unsigned long long)ix->m, (unsigned long long)stride);
    for (uint64_t r = 0; r < ix->m; r++) memcpy(grown + r * stride, ix->row(r), ix->stride);
    free(ix->rows);
    ix->rows = grown;
    ix->stride = stride;
    ix->cap_cols = stride * 8;
    return BIGSI_OK;
}

int bigsi_cpu_synchronize(bigsi_cpu_index *ix) { return ix ? BIGSI_OK : fail(BIGSI_ERR_INVALID, "NULL index"); }

int bigsi_cpu_set_rows(bigsi_cpu_index *ix, const uint64_t *row_ids, uint64_t n, const uint8_t *bytes, uint64_t row_bytes)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix || (n && (!row_ids || !bytes))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (row_bytes > ix->stride) return fail(BIGSI_ERR_CAPACITY, "row_bytes %llu exceeds the row stride %llu", (unsigned long long)row_bytes, (unsigned long long)ix->stride);
    for (uint64_t i = 0; i < n; i++)
        if (row_ids[i] >= ix->m) return fail(BIGSI_ERR_RANGE, "row %llu out of range", (unsigned long long)row_ids[i]);
    for (uint64_t i = 0; i < n; i++) {
        uint8_t *dst = ix->row(row_ids[i]);
        memcpy(dst, bytes + i * row_bytes, row_bytes);
        memset(dst + row_bytes, 0, ix->stride - row_bytes);
    }
    return BIGSI_OK;
}

int bigsi_cpu_get_rows(bigsi_cpu_index *ix, const uint64_t *row_ids, uint64_t n, uint8_t *out, uint64_t row_bytes)
{
    if (!ix || (n && (!row_ids || !out))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (row_bytes > ix->stride) return fail(BIGSI_ERR_CAPACITY, "row_bytes %llu exceeds the row stride %llu", (unsigned long long)row_bytes, (unsigned long long)ix->stride);
    for (uint64_t i = 0; i < n; i++)
        if (row_ids[i] >= ix->m) return fail(BIGSI_ERR_RANGE, "row %llu out of range", (unsigned long long)row_ids[i]);
    std::vector<uint8_t> tmp, page;
    for (uint64_t i = 0; i < n; i++) memcpy(out + i * row_bytes, ix->fetch(row_ids[i], row_bytes, tmp, page), row_bytes);
    return bdb_read_failed(ix);
}

int bigsi_cpu_clear(bigsi_cpu_index *ix)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    memset(ix->rows, 0, ix->m * ix->stride);
    return BIGSI_OK;
}

int bigsi_cpu_insert_column(bigsi_cpu_index *ix, uint64_t col, const uint8_t *bloom)
{
    return bigsi_cpu_insert_columns(ix, col, 1, bloom, ix ? ceil_div(ix->m, 8) : 0);
}

int bigsi_cpu_insert_columns(bigsi_cpu_index *ix, uint64_t col0, uint64_t n, const uint8_t *blooms, uint64_t bloom_stride_bytes)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix || (n && !blooms)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (col0 > ix->n_cols) return fail(BIGSI_ERR_RANGE, "col0 %llu beyond num_cols %llu", (unsigned long long)col0, (unsigned long long)ix->n_cols);
    if (col0 + n > ix->cap_cols) return fail(BIGSI_ERR_CAPACITY, "columns [%llu, %llu) exceed col_capacity %llu", (unsigned long long)col0, (unsigned long long)(col0 + n), (unsigned long long)ix->cap_cols);
    if (n && bloom_stride_bytes < ceil_div(ix->m, 8)) return fail(BIGSI_ERR_INVALID, "bloom_stride_bytes is smaller than a filter");
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *f = blooms + i * bloom_stride_bytes;
        const uint64_t c = col0 + i;
        const uint8_t mask = (uint8_t)(0x80u >> (c & 7));
        for (uint64_t r = 0; r < ix->m; r++) {
            uint8_t *b = ix->row(r) + (c >> 3);
            if (f[r >> 3] & (0x80u >> (r & 7))) *b |= mask;
            else *b &= (uint8_t)~mask;
        }
    }
    ix->n_cols = std::max(ix->n_cols, col0 + n);
    return BIGSI_OK;
}

int bigsi_cpu_get_column(bigsi_cpu_index *ix, uint64_t col, uint8_t *out)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (col >= ix->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu out of range", (unsigned long long)col);
    memset(out, 0, ceil_div(ix->m, 8));
    for (uint64_t r = 0; r < ix->m; r++)
        if (ix->row(r)[col >> 3] & (0x80u >> (col & 7))) out[r >> 3] |= (uint8_t)(0x80u >> (r & 7));
    return BIGSI_OK;
}

int bigsi_cpu_insert_kmers(bigsi_cpu_index *ix, uint64_t col, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    if (n_seqs == 0) return BIGSI_OK;
    TRY(check_seqs(seqs, offsets, n_seqs, k));
    if (col >= ix->n_cols) return fail(BIGSI_ERR_RANGE, "column %llu out of range (num_cols %llu)", (unsigned long long)col, (unsigned long long)ix->n_cols);
    std::string canon;
    std::vector<uint64_t> r(ix->h);
    for (uint32_t i = 0; i < n_seqs; i++) {
        const char *s = seqs + offsets[i];
        const uint64_t len = offsets[i + 1] - offsets[i];
        for (uint64_t p = 0; p + k <= len; p++) {
            canonical(s + p, k, canon);
            rows_of(canon, ix->h, ix->m, r.data());
            for (uint32_t t = 0; t < ix->h; t++) ix->row(r[t])[col >> 3] |= (uint8_t)(0x80u >> (col & 7));
        }
    }
    return BIGSI_OK;
}

int bigsi_cpu_fill_synthetic(bigsi_cpu_index *ix, uint64_t seed, uint64_t shard, uint32_t and_draws)
{
    if (ix && ix->bdb) return fail(BIGSI_ERR_STATE, "this index serves its rows from a BerkeleyDB file (bigsi_cpu_open_bdb): read-only, search / lookup / get_rows / presence only");
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    const uint64_t base = mix64(seed + shard * 0x632BE59BD9B4E019ull), words = ix->stride / 8;
    for (uint64_t r = 0; r < ix->m; r++) {
        const uint64_t rk = mix64(base ^ (r * 0x9E3779B97F4A7C15ull));
        for (uint64_t w = 0; w < words; w++) {
            uint64_t v = ~0ull;
            for (uint32_t d = 0; d < and_draws; d++) v &= mix64(rk + (w * 8 + d) * 0xD1B54A32D192ED03ull);
            v &= valid_word_mask(w, ix->n_cols);
            memcpy(ix->row(r) + 8 * w, &v, 8);
        }
    }
    return BIGSI_OK;
}

int bigsi_cpu_bloom(int, const char *kmers, uint64_t u, uint32_t k, uint64_t m, uint32_t h, uint32_t flags, uint8_t *out)
{
    if (!out || (u && !kmers)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0 || m == 0 || h == 0) return fail(BIGSI_ERR_INVALID, "k, m and h must be > 0");
    memset(out, 0, ceil_div(m, 8));
    std::string canon;
    std::vector<uint64_t> r(h);
    for (uint64_t i = 0; i < u; i++) {
        if (flags & BIGSI_BLOOM_RAW) canon.assign(kmers + i * k, k);
        else canonical(kmers + i * k, k, canon);
        rows_of(canon, h, m, r.data());
        for (uint32_t t = 0; t < h; t++) out[r[t] >> 3] |= (uint8_t)(0x80u >> (r[t] & 7));
    }
    return BIGSI_OK;
}

int bigsi_cpu_lookup(bigsi_cpu_index *ix, const char *kmers, uint32_t k, uint64_t u, uint8_t *out_rows)
{
    if (!ix || (u && (!kmers || !out_rows))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    if (u == 0) return BIGSI_OK;
    if (ix->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    const uint64_t rb = ix->rb();
    std::string canon;
    std::vector<uint64_t> r(ix->h);
    std::vector<uint8_t> tmp, page;
    for (uint64_t i = 0; i < u; i++) {
        canonical(kmers + i * k, k, canon);
        rows_of(canon, ix->h, ix->m, r.data());
        uint8_t *o = out_rows + i * rb;
        memcpy(o, ix->fetch(r[0], rb, tmp, page), rb);
        for (uint32_t t = 1; t < ix->h; t++) {
            const uint8_t *x = ix->fetch(r[t], rb, tmp, page);
            for (uint64_t b = 0; b < rb; b++) o[b] &= x[b];
        }
    }
    return bdb_read_failed(ix);
}

int bigsi_cpu_search_stream(bigsi_cpu_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                            double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                            uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity)
{
    if (!hit_offsets) return fail(BIGSI_ERR_INVALID, "hit_offsets is NULL");
    if (!ix) return fail(BIGSI_ERR_INVALID, "NULL index");
    TRY(check_seqs(seqs, offsets, n_seqs, k));
    if (!(threshold <= 1.0)) return fail(BIGSI_ERR_INVALID, "threshold must be <= 1 (bigsi/graph/bigsi.py:176), got %g", threshold);
    if (ix->n_cols == 0) return fail(BIGSI_ERR_STATE, "index has no columns");
    if (ix->bdb && (flags & BIGSI_CPU_WORD_PARALLEL)) return fail(BIGSI_ERR_STATE, "BIGSI_CPU_WORD_PARALLEL works on rows resident in RAM, not on a BerkeleyDB-backed index");
    Hits hits{hit_offsets, colours, counts, hit_capacity};
    QueryKmers q;
    for (uint64_t i = 0; i < n_seqs; i++) {
        uint32_t nk, nu, mk;
        hit_offsets[i] = hits.total;
        if (flags & BIGSI_CPU_WORD_PARALLEL) search_word_parallel(ix, seqs + offsets[i], offsets[i + 1] - offsets[i], k, threshold, q, &nk, &nu, &mk, hits);
        else search_reference_shaped(ix, seqs + offsets[i], offsets[i + 1] - offsets[i], k, threshold, q, &nk, &nu, &mk, hits);
        if (num_kmers) num_kmers[i] = nk;
        if (num_unique) num_unique[i] = nu;
        if (min_kmers) min_kmers[i] = mk;
    }
    hit_offsets[n_seqs] = hits.total;
    TRY(bdb_read_failed(ix));
    if (hits.total > hit_capacity)
        return fail(BIGSI_ERR_CAPACITY, "hit buffers hold %llu entries, %llu needed", (unsigned long long)hit_capacity, (unsigned long long)hits.total);
    return BIGSI_OK;
}

int bigsi_cpu_search_batch(bigsi_cpu_index *ix, const char *seqs, const uint64_t *offsets, uint32_t n_seqs, uint32_t k,
                           double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                           uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity)
{
    return bigsi_cpu_search_stream(ix, seqs, offsets, n_seqs, k, threshold, flags, num_kmers, num_unique, min_kmers, hit_offsets, colours, counts, hit_capacity);
}

int bigsi_cpu_presence(bigsi_cpu_index *ix, const char *seq, uint64_t len, uint32_t k, const uint32_t *colours, uint32_t n_colours, uint8_t *out)
{
    if (!ix || (len && !seq)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (k == 0) return fail(BIGSI_ERR_INVALID, "k must be > 0");
    if (n_colours == 0 || len < k) return BIGSI_OK;
    if (!colours || !out) return fail(BIGSI_ERR_INVALID, "NULL argument");
    for (uint32_t j = 0; j < n_colours; j++)
        if (colours[j] >= ix->n_cols) return fail(BIGSI_ERR_RANGE, "colour %u >= num_cols", colours[j]);
    const uint64_t n = len - k + 1;
    std::string canon;
    std::vector<uint64_t> r(ix->h);
    std::vector<uint8_t> tmp, page;
    for (uint64_t i = 0; i < n; i++) {
        canonical(seq + i, k, canon);
        rows_of(canon, ix->h, ix->m, r.data());
        for (uint32_t j = 0; j < n_colours; j++) {
            const uint32_t c = colours[j];
            bool present = true;
            for (uint32_t t = 0; t < ix->h && present; t++) present = (ix->fetch(r[t], ix->rb(), tmp, page)[c >> 3] & (0x80u >> (c & 7))) != 0;
            out[(uint64_t)j * n + i] = present ? '1' : '0';
        }
    }
    return bdb_read_failed(ix);
}

int bigsi_cpu_score_presence(int, const uint8_t *bits, const uint64_t *bit_offsets, const uint32_t *num_kmers, const uint32_t *found,
                             const uint32_t *unique, uint64_t n, bigsi_hip_hit_score *scores)
{
    if (n == 0) return BIGSI_OK;
    if (!bits || !bit_offsets || !num_kmers || !scores) return fail(BIGSI_ERR_INVALID, "NULL argument");
    static_assert(sizeof(bigsi_hip_hit_score) == sizeof(bigsi_score::HitScore), "score record layout");
    for (uint64_t t = 0; t < n; t++) {
        if (bit_offsets[t] & 7u) return fail(BIGSI_ERR_INVALID, "bit_offsets[%llu] is not a multiple of 8", (unsigned long long)t);
        const uint8_t *p = bits + bit_offsets[t];
        bigsi_score::score_hit([p](uint32_t kk) { uint64_t w; memcpy(&w, p + 8ull * kk, 8); return bigsi_score::lsb_first(w); }, num_kmers[t],
                               found ? found[t] : 0u, unique ? unique[t] : 0u, reinterpret_cast<bigsi_score::HitScore *>(&scores[t]));
    }
    return BIGSI_OK;
}

// BIGSI.search(..., score=True) for any number of sequences (the twin of bigsi_hip_search_stream_scored): the hit lists of
// bigsi_cpu_search_stream, then per sequence with hits its presence bits (graph/bigsi.py:232-237, one test per k-mer position,
// colour and hash) and Scorer.score's record (bigsi_score.hpp, the code the device compiles).
int bigsi_cpu_search_stream_scored(bigsi_cpu_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, uint32_t k,
                                   double threshold, uint32_t flags, uint32_t *num_kmers, uint32_t *num_unique, uint32_t *min_kmers,
                                   uint64_t *hit_offsets, uint32_t *colours, uint32_t *counts, uint64_t hit_capacity, uint8_t *bits,
                                   uint64_t bits_capacity, uint64_t *bit_offsets, bigsi_hip_hit_score *scores, uint64_t *bits_needed)
{
    if (!bit_offsets) return fail(BIGSI_ERR_INVALID, "bit_offsets is NULL");
    if (hit_capacity && (!colours || !scores)) return fail(BIGSI_ERR_INVALID, "colours / scores is NULL");
    bit_offsets[0] = 0;
    if (bits_needed) *bits_needed = 0;
    std::vector<uint32_t> nk_own, nu_own;
    if (!num_kmers) { nk_own.resize(n_seqs); num_kmers = nk_own.data(); }
    if (!num_unique) { nu_own.resize(n_seqs); num_unique = nu_own.data(); }
    const int rc = bigsi_cpu_search_stream(ix, seqs, offsets, n_seqs, k, threshold, flags, num_kmers, num_unique, min_kmers, hit_offsets, colours,
                                           counts, hit_capacity);
    if (rc != BIGSI_OK && rc != BIGSI_ERR_CAPACITY) return rc;
    uint64_t need = 0;
    for (uint64_t i = 0; i < n_seqs; i++) need += (hit_offsets[i + 1] - hit_offsets[i]) * (((uint64_t)num_kmers[i] + 63) / 64 * 8);
    if (bits_needed) *bits_needed = need;
    if (rc == BIGSI_ERR_CAPACITY) return rc;
    const bool fits = need <= bits_capacity && (bits || need == 0);
    std::vector<uint8_t> text;
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_seqs; i++) {
        const uint64_t lo = hit_offsets[i], hi = hit_offsets[i + 1], n = num_kmers[i], bytes = (n + 63) / 64 * 8;
        if (hi == lo) continue;
        if (fits && n) {
            text.resize((hi - lo) * n);
            TRY(bigsi_cpu_presence(ix, seqs + offsets[i], offsets[i + 1] - offsets[i], k, colours + lo, (uint32_t)(hi - lo), text.data()));
        }
        for (uint64_t t = lo; t < hi; t++) {
            bit_offsets[t] = at;
            if (fits) {
                memset(&scores[t], 0, sizeof scores[t]);
                if (n) {
                    uint8_t *p = bits + at;
                    memset(p, 0, bytes);
                    const uint8_t *s = text.data() + (t - lo) * n;
                    for (uint64_t j = 0; j < n; j++)
                        if (s[j] == '1') p[j >> 3] |= (uint8_t)(0x80u >> (j & 7));
                    bigsi_score::score_hit([p](uint32_t kk) { uint64_t w; memcpy(&w, p + 8ull * kk, 8); return bigsi_score::lsb_first(w); }, (uint32_t)n,
                                           counts ? counts[t] : num_unique[i], num_unique[i], reinterpret_cast<bigsi_score::HitScore *>(&scores[t]));
                }
            }
            at += bytes;
        }
    }
    bit_offsets[hit_offsets[n_seqs]] = at;
    if (!fits) return fail(BIGSI_ERR_CAPACITY, "bit buffer holds %llu bytes, %llu needed", (unsigned long long)bits_capacity, (unsigned long long)need);
    return BIGSI_OK;
}

}  // extern "C"
