// Host-side text of the batch front-end: the FASTA records a bulk search reads and the JSON / CSV text it
// returns, for callers that want the reference's text and not Python objects.  No device code: what the reference does per record
// in Python -- pyfasta's records, json.dumps(list of records, indent=4), csv.writer rows (bigsi/__main__.py:41-72, 261-314) -- done
// over the arrays bigsi_hip_search_stream leaves, by a few host threads.  bigsi_amd/frontend.py calls these for unscored bulk
// searches of ASCII files and keeps its Python route for everything else (scores, non-ASCII text, queries on which the reference
// raises); tests/test_frontend_text.py pins both routes to the same text and to golden G9.
#include <emmintrin.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bigsi_internal.hpp"
#include "bigsi_score.hpp"

namespace {

// str.strip()'s ASCII whitespace
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f); }

// ------------------------------------------------------------------------------ sinks: one formatter, run twice
// (first to size every block of records, then to write it at its place: the blocks are independent, so both passes are threaded)
struct CountSink {
    uint64_t n = 0;
    inline void put(const char *, size_t len) { n += len; }
    inline void put_c(char) { n += 1; }
    inline void put_u32(uint32_t v) { n += v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : v < 1000000 ? 6 : v < 10000000 ? 7 : v < 100000000 ? 8 : v < 1000000000 ? 9 : 10; }
};
struct WriteSink {
    char *p;
    inline void put(const char *s, size_t len) { memcpy(p, s, len); p += len; }
    inline void put_c(char c) { *p++ = c; }
    inline void put_u32(uint32_t v)
    {
        char tmp[10];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) *p++ = tmp[--n];
    }
};
struct StringSink {          // (scored text: formatting a float costs more than copying it, so a block is formatted once, into a string)
    std::string t;
    inline void put(const char *s, size_t len) { t.append(s, len); }
    inline void put_c(char c) { t.push_back(c); }
    inline void put_u32(uint32_t v) { char tmp[12]; const int n = snprintf(tmp, sizeof tmp, "%u", v); t.append(tmp, (size_t)n); }
};
#define PUT_LIT(sink, lit) (sink).put(lit, sizeof(lit) - 1)

// repr(round(100 * float(found) / num_kmers, 2)) (graph/bigsi.py:97-99): the rounded value is the double nearest to a decimal of
// at most two places, whose shortest round-trip text is that decimal without trailing zeros (one place at least)
template <typename S>
inline void put_percent(S &s, uint32_t found, uint32_t num)
{
    const double y = bigsi_score::py_round2(100.0 * (double)found / (double)num);
    const uint32_t c = (uint32_t)llrint(y * 100.0), whole = c / 100, frac = c % 100;
    s.put_u32(whole);
    s.put_c('.');
    if (frac % 10 == 0) s.put_c((char)('0' + frac / 10));
    else { s.put_c((char)('0' + frac / 10)); s.put_c((char)('0' + frac % 10)); }
}

// repr(x) of a Python float (what json.dumps and csv.writer print): the shortest decimal string that reads back as x -- found as the
// first of the correctly rounded 15-, 16- and 17-digit decimals that does (the closest n-digit decimal round-trips whenever an
// n-digit one does) -- laid out as float_repr_style 'short' does: exponent form when the decimal point would sit more than 16
// digits right of the first digit or 4 or more zeros left of it, else plain digits with at least one decimal place
// (Python/pystrtod.c: format_float_short, 'r').  `json`: Infinity / NaN as json.dumps writes them (csv: inf / nan).
template <typename S>
inline void put_repr(S &s, double x, bool json)
{
    if (x != x) { if (json) PUT_LIT(s, "NaN"); else PUT_LIT(s, "nan"); return; }
    if (x == HUGE_VAL || x == -HUGE_VAL) {
        if (x < 0) s.put_c('-');
        if (json) PUT_LIT(s, "Infinity"); else PUT_LIT(s, "inf");
        return;
    }
    if (x == 0.0) { if (signbit(x)) PUT_LIT(s, "-0.0"); else PUT_LIT(s, "0.0"); return; }
    {
        // most fields of a score record are decimals of at most two places (rounded scores, percentages, 0.0 / 1.0 / 100.0): if
        // c / 100 is x for the integer c nearest to 100 x, "c / 100" written out is the shortest decimal that reads back as x (two
        // decimals that both do are closer together than 0.01 unless they are the same number)
        const double c = rint(x * 100.0);
        if (fabs(c) < 1e13 && c / 100.0 == x) {
            uint64_t m = (uint64_t)fabs(c);
            if (x < 0) s.put_c('-');
            char tmp[24];
            int n = 0;
            uint64_t whole = m / 100;
            do { tmp[n++] = (char)('0' + whole % 10); whole /= 10; } while (whole);
            while (n) s.put_c(tmp[--n]);
            s.put_c('.');
            const uint32_t frac = (uint32_t)(m % 100);
            s.put_c((char)('0' + frac / 10));
            if (frac % 10) s.put_c((char)('0' + frac % 10));
            return;
        }
    }
    char buf[40];
    // (a normal double is within 1.2e-16 of any decimal that reads back as it, so a shorter one shows as trailing zeros of the 15-digit
    // rounding; subnormals are spaced wider than that and are tried from one digit up)
    for (int digits = fabs(x) >= 2.2250738585072014e-308 ? 15 : 1;; digits++) {
        snprintf(buf, sizeof buf, "%.*e", digits - 1, x);
        if (digits == 17 || strtod(buf, nullptr) == x) break;
    }
    // buf = [-]d.ddddde[+-]XX
    const char *p = buf;
    if (*p == '-') { s.put_c('-'); p++; }
    char dg[20];
    int nd = 0;
    dg[nd++] = *p++;
    if (*p == '.') { p++; while (*p != 'e') dg[nd++] = *p++; }
    while (nd > 1 && dg[nd - 1] == '0') nd--;
    const int decpt = atoi(p + 1) + 1;          // the value is 0.d1d2... x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        s.put_c(dg[0]);
        if (nd > 1) { s.put_c('.'); s.put(dg + 1, (size_t)nd - 1); }
        s.put_c('e');
        int e = decpt - 1;
        if (e < 0) { s.put_c('-'); e = -e; } else s.put_c('+');
        if (e < 10) s.put_c('0');
        s.put_u32((uint32_t)e);
    } else if (decpt <= 0) {
        PUT_LIT(s, "0.");
        for (int i = 0; i < -decpt; i++) s.put_c('0');
        s.put(dg, (size_t)nd);
    } else if (decpt >= nd) {
        s.put(dg, (size_t)nd);
        for (int i = nd; i < decpt; i++) s.put_c('0');
        PUT_LIT(s, ".0");
    } else {
        s.put(dg, (size_t)decpt);
        s.put_c('.');
        s.put(dg + decpt, (size_t)(nd - decpt));
    }
}

template <typename S>
inline void put_i64(S &s, int64_t v)
{
    char tmp[24];
    const int n = snprintf(tmp, sizeof tmp, "%lld", (long long)v);
    s.put(tmp, (size_t)n);
}

// the presence string of a hit ('0' / '1' per k-mer position) from its bits (bitarray(s).tobytes(): position p under 0x80 >> (p % 8))
template <typename S>
inline void put_presence(S &s, const uint8_t *bits, uint32_t n)
{
    char tmp[64];
    for (uint32_t p0 = 0; p0 < n; p0 += 64) {
        const uint32_t m = n - p0 < 64 ? n - p0 : 64;
        for (uint32_t i = 0; i < m; i++) tmp[i] = (char)('0' + ((bits[(p0 + i) >> 3] >> (7 - ((p0 + i) & 7))) & 1));
        s.put(tmp, m);
    }
}

// json.dumps of an ASCII string: quotes, backslashes, control characters and DEL escaped
template <typename S>
inline void put_json_string(S &s, const char *t, size_t len)
{
    static const char hex[] = "0123456789abcdef";
    s.put_c('"');
    size_t run = 0;
    for (size_t i = 0; i < len; i++) {
        const unsigned char c = (unsigned char)t[i];
        if (c >= 0x20 && c < 0x7f && c != '"' && c != '\\') continue;      // (json's ESCAPE_ASCII: everything outside ' '..'~')
        s.put(t + run, i - run);
        run = i + 1;
        switch (c) {
        case '"': PUT_LIT(s, "\\\""); break;
        case '\\': PUT_LIT(s, "\\\\"); break;
        case '\n': PUT_LIT(s, "\\n"); break;
        case '\r': PUT_LIT(s, "\\r"); break;
        case '\t': PUT_LIT(s, "\\t"); break;
        case '\b': PUT_LIT(s, "\\b"); break;
        case '\f': PUT_LIT(s, "\\f"); break;
        default: { const char u[6] = {'\\', 'u', '0', '0', hex[c >> 4], hex[c & 15]}; s.put(u, 6); }
        }
    }
    s.put(t + run, len - run);
    s.put_c('"');
}

// csv.writer, QUOTE_NONNUMERIC: a string in quotes, its quotes doubled
template <typename S>
inline void put_csv_string(S &s, const char *t, size_t len)
{
    s.put_c('"');
    size_t run = 0;
    for (size_t i = 0; i < len; i++)
        if (t[i] == '"') { s.put(t + run, i + 1 - run); s.put_c('"'); run = i + 1; }
    s.put(t + run, len - run);
    s.put_c('"');
}

struct Job {
    int format;
    const char *seqs;
    const uint64_t *offsets;
    const char *thr;
    size_t thr_len;
    const char *cit;
    size_t cit_len;
    bool exact;
    const uint32_t *num_unique;
    const uint64_t *hit_offsets;
    const uint32_t *colours, *counts;
    const char *names;
    const uint64_t *name_offsets;
    const uint8_t *name_deleted;
    uint64_t n_names;
    const bigsi_hip_scored_text *sc;      // score=True: per-hit records, presence bits and the caller's closed-form columns (else NULL)
};

// the 17 fields of Scorer.score (scoring/score.py:96-121) + "kmer-presence" of hit t, as `field(key index, writer)` calls in the
// reference's key order: score, min_score, max_score, max_mismatches, min_mismatches, mismatches, max_nident, nident, min_nident,
// pident, max_pident, min_pident, length, evalue, pvalue, log_evalue, log_pvalue
struct ScoredHit {
    double f[10];        // score, min_score, max_score, pident, max_pident, min_pident, evalue, pvalue, log_evalue, log_pvalue
    int64_t i[7];        // max_mismatches, min_mismatches, mismatches, max_nident, nident, min_nident, length
    double percent;
    const uint8_t *bits;
    uint32_t n;
    ScoredHit(const bigsi_hip_scored_text &sc, uint64_t t)
    {
        const bigsi_hip_hit_score &r = sc.scores[t];
        const int64_t len = (int64_t)r.num_kmers + (int64_t)sc.k - 1;
        const double fl = (double)len;
        i[0] = r.max_mismatches; i[1] = r.min_mismatches; i[2] = r.mismatches;
        i[3] = len - r.min_mismatches; i[4] = len - r.mismatches; i[5] = len - r.max_mismatches; i[6] = len;
        f[0] = r.score; f[1] = r.min_score; f[2] = r.max_score;
        f[3] = 100.0 * (double)i[4] / fl; f[4] = 100.0 * (double)i[3] / fl; f[5] = 100.0 * (double)i[5] / fl;      // score.py:110-112
        f[6] = sc.evalue[t]; f[7] = sc.pvalue[t]; f[8] = sc.log_evalue[t]; f[9] = sc.log_pvalue[t];
        percent = r.percent_kmers_found;
        bits = sc.bits + sc.bit_offsets[t];
        n = r.num_kmers;
    }
};

// the records [r0, r1) of a search, each after its separator (every record but the very first has one)
template <typename S>
void format_records(const Job &j, uint64_t r0, uint64_t r1, S &s, std::vector<uint64_t> &order)
{
    for (uint64_t r = r0; r < r1; r++) {
        const char *q = j.seqs + j.offsets[r];
        const size_t qlen = j.offsets[r + 1] - j.offsets[r];
        const uint32_t u = j.num_unique[r];
        // the hits that make it into the record: exact_filter takes every set bit in ascending order (graph/bigsi.py:192-205),
        // inexact_filter the colours below num_samples in a STABLE sort by count, descending (:211-230); deleted samples dropped
        // from either (:186-190)
        order.clear();
        for (uint64_t t = j.hit_offsets[r]; t < j.hit_offsets[r + 1]; t++) {
            const uint32_t c = j.colours[t];
            if (c >= j.n_names || j.name_deleted[c]) continue;
            order.push_back(t);
        }
        if (!j.exact && order.size() > 1)
            std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return j.counts[a] > j.counts[b]; });
        if (j.format == 1) {
            if (r) s.put_c('\n');
            for (size_t i = 0; i < order.size(); i++) {
                const uint64_t t = order[i];
                const uint32_t c = j.colours[t], f = j.exact ? u : j.counts[t];
                put_csv_string(s, q, qlen);
                s.put_c(',');
                if (j.sc) {
                    // d_to_csv: the result's values in sorted-key order (__main__.py:41-63): evalue, kmer-presence, length, log_evalue,
                    // log_pvalue, max_mismatches, max_nident, max_pident, max_score, min_mismatches, min_nident, min_pident, min_score,
                    // mismatches, nident, num_kmers, num_kmers_found, percent_kmers_found, pident, pvalue, sample_name, score
                    const ScoredHit h(*j.sc, t);
                    put_repr(s, h.f[6], false); s.put_c(',');
                    s.put_c('"'); put_presence(s, h.bits, h.n); s.put_c('"'); s.put_c(',');
                    put_i64(s, h.i[6]); s.put_c(',');
                    put_repr(s, h.f[8], false); s.put_c(',');
                    put_repr(s, h.f[9], false); s.put_c(',');
                    put_i64(s, h.i[0]); s.put_c(',');
                    put_i64(s, h.i[3]); s.put_c(',');
                    put_repr(s, h.f[4], false); s.put_c(',');
                    put_repr(s, h.f[2], false); s.put_c(',');
                    put_i64(s, h.i[1]); s.put_c(',');
                    put_i64(s, h.i[5]); s.put_c(',');
                    put_repr(s, h.f[5], false); s.put_c(',');
                    put_repr(s, h.f[1], false); s.put_c(',');
                    put_i64(s, h.i[2]); s.put_c(',');
                    put_i64(s, h.i[4]); s.put_c(',');
                    s.put_u32(u); s.put_c(',');
                    s.put_u32(f); s.put_c(',');
                    put_repr(s, h.percent, false); s.put_c(',');
                    put_repr(s, h.f[3], false); s.put_c(',');
                    put_repr(s, h.f[7], false); s.put_c(',');
                    put_csv_string(s, j.names + j.name_offsets[c], j.name_offsets[c + 1] - j.name_offsets[c]); s.put_c(',');
                    put_repr(s, h.f[0], false);
                    s.put_c('\r');
                    if (i + 1 < order.size()) s.put_c('\n');
                    continue;
                }
                s.put_u32(u);
                s.put_c(',');
                s.put_u32(f);
                s.put_c(',');
                put_percent(s, f, u);
                s.put_c(',');
                put_csv_string(s, j.names + j.name_offsets[c], j.name_offsets[c + 1] - j.name_offsets[c]);
                s.put_c('\r');
                if (i + 1 < order.size()) s.put_c('\n');      // (the reference drops the record's last character)
            }
            continue;
        }
        if (r) PUT_LIT(s, ",\n");
        PUT_LIT(s, "    {\n        \"query\": ");
        put_json_string(s, q, qlen);
        PUT_LIT(s, ",\n        \"threshold\": ");
        s.put(j.thr, j.thr_len);
        if (order.empty()) {
            PUT_LIT(s, ",\n        \"results\": [],\n        \"citation\": ");
        } else {
            PUT_LIT(s, ",\n        \"results\": [\n");
            for (size_t i = 0; i < order.size(); i++) {
                const uint64_t t = order[i];
                const uint32_t c = j.colours[t], f = j.exact ? u : j.counts[t];
                PUT_LIT(s, "            {\n                \"percent_kmers_found\": ");
                if (j.sc) put_repr(s, j.sc->scores[t].percent_kmers_found, true);
                else put_percent(s, f, u);
                PUT_LIT(s, ",\n                \"num_kmers\": ");
                s.put_u32(u);
                PUT_LIT(s, ",\n                \"num_kmers_found\": ");
                s.put_u32(f);
                PUT_LIT(s, ",\n                \"sample_name\": ");
                put_json_string(s, j.names + j.name_offsets[c], j.name_offsets[c + 1] - j.name_offsets[c]);
                if (j.sc) {
                    static const char *const fkeys[10] = {"score", "min_score", "max_score", "pident", "max_pident", "min_pident", "evalue", "pvalue", "log_evalue", "log_pvalue"};
                    static const char *const ikeys[7] = {"max_mismatches", "min_mismatches", "mismatches", "max_nident", "nident", "min_nident", "length"};
                    const ScoredHit h(*j.sc, t);
                    auto key = [&](const char *k) { PUT_LIT(s, ",\n                \""); s.put(k, strlen(k)); PUT_LIT(s, "\": "); };
                    for (int x = 0; x < 3; x++) { key(fkeys[x]); put_repr(s, h.f[x], true); }
                    for (int x = 0; x < 6; x++) { key(ikeys[x]); put_i64(s, h.i[x]); }
                    for (int x = 3; x < 6; x++) { key(fkeys[x]); put_repr(s, h.f[x], true); }
                    key(ikeys[6]); put_i64(s, h.i[6]);
                    for (int x = 6; x < 10; x++) { key(fkeys[x]); put_repr(s, h.f[x], true); }
                    key("kmer-presence"); s.put_c('"'); put_presence(s, h.bits, h.n); s.put_c('"');
                }
                PUT_LIT(s, "\n            }");
                if (i + 1 < order.size()) PUT_LIT(s, ",\n");
            }
            PUT_LIT(s, "\n        ],\n        \"citation\": ");
        }
        s.put(j.cit, j.cit_len);
        PUT_LIT(s, "\n    }");
    }
}

template <typename F>
void run_blocks(uint64_t n_blocks, uint32_t threads, F f)
{
    if (threads <= 1 || n_blocks <= 1) {
        for (uint64_t b = 0; b < n_blocks; b++) f(b);
        return;
    }
    std::vector<std::thread> pool;
    const uint32_t T = (uint32_t)std::min<uint64_t>(threads, n_blocks);
    for (uint32_t t = 0; t < T; t++)
        pool.emplace_back([=] { for (uint64_t b = t; b < n_blocks; b += T) f(b); });
    for (auto &th : pool) th.join();
}


// One chunk of a FASTA text (it starts right after a line end, or at 0): its lines, stripped as str.strip() would, handed to
// `header()` (a line that starts with '>') or `bases(a, b)` (any other non-empty line).  Line ends ('\n', '\r') are found 16 bytes at
// a time (SSE2, the x86-64 baseline: a million 61-bp reads are a 64 MB file, and a byte loop over it took longer than the search).
// Returns non-zero if a byte >= 0x80 was seen.
template <typename H, typename B>
int scan_fasta(const char *text, uint64_t lo, uint64_t hi, H header, B bases)
{
    uint64_t start = lo, pos = lo;
    int high = 0;
    auto line = [&](uint64_t a, uint64_t b) {
        while (a < b && is_space((unsigned char)text[a])) a++;
        while (b > a && is_space((unsigned char)text[b - 1])) b--;
        if (a == b) return;
        if (text[a] == '>') header();
        else bases(a, b);
    };
    const __m128i lf = _mm_set1_epi8('\n'), cr = _mm_set1_epi8('\r');
    for (; pos + 16 <= hi; pos += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(text + pos));
        high |= _mm_movemask_epi8(v);
        unsigned m = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, lf), _mm_cmpeq_epi8(v, cr)));
        while (m) {
            const uint64_t e = pos + (unsigned)__builtin_ctz(m);
            m &= m - 1;
            line(start, e);
            start = e + 1;
        }
    }
    for (; pos < hi; pos++) {
        const unsigned char c = (unsigned char)text[pos];
        high |= c & 0x80;
        if (c == '\n' || c == '\r') { line(start, pos); start = pos + 1; }
    }
    if (start < hi) line(start, hi);
    return high;
}

}   // namespace

// bigsi_hip.h: the sequences of a FASTA text, packed for the search entry points.  The text is cut into chunks at line ends and scanned
// twice by a few host threads: once to count every chunk's records and sequence bytes (the bytes before a chunk's first header belong
// to the last record of the chunks before it), once to write them at their places.
extern "C" int bigsi_hip_fasta_pack(const char *text, uint64_t n_bytes, char *out_seqs, uint64_t *out_offsets, uint64_t max_records, uint64_t *n_records)
{
    if ((n_bytes && !text) || !n_records) return fail(BIGSI_ERR_INVALID, "NULL argument");
    const uint32_t threads = (uint32_t)std::min<uint64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n_bytes / (4u << 20) + 1);
    const uint64_t n_chunks = threads;
    std::vector<uint64_t> cut(n_chunks + 1, n_bytes);
    cut[0] = 0;
    for (uint64_t c = 1; c < n_chunks; c++) {          // a chunk starts right after a line end
        uint64_t p = std::max(cut[c - 1], n_bytes / n_chunks * c);
        while (p < n_bytes && text[p] != '\n' && text[p] != '\r') p++;
        cut[c] = p < n_bytes ? p + 1 : n_bytes;
    }
    struct Tally { uint64_t records = 0, lead = 0, body = 0; int high = 0; };
    std::vector<Tally> tally(n_chunks);
    run_blocks(n_chunks, threads, [&](uint64_t c) {
        Tally t;
        t.high = scan_fasta(text, cut[c], cut[c + 1], [&] { t.records++; }, [&](uint64_t a, uint64_t b) { (t.records ? t.body : t.lead) += b - a; });
        tally[c] = t;
    });
    uint64_t n = 0, w = 0;
    std::vector<uint64_t> rec0(n_chunks), w0(n_chunks);
    for (uint64_t c = 0; c < n_chunks; c++) {
        if (tally[c].high) return fail(BIGSI_ERR_INVALID, "not plain ASCII");
        rec0[c] = n;
        w0[c] = w;
        w += (n ? tally[c].lead : 0) + tally[c].body;      // (bases before the text's first header belong to no record)
        n += tally[c].records;
    }
    *n_records = n;
    if (!out_offsets && !out_seqs) return BIGSI_OK;      // (the sizing call: n + 1 offsets, at most n_bytes sequence bytes)
    if (out_offsets && n > max_records) return fail(BIGSI_ERR_CAPACITY, "%llu records, room for %llu", (unsigned long long)n, (unsigned long long)max_records);
    run_blocks(n_chunks, threads, [&](uint64_t c) {
        uint64_t r = rec0[c], at = w0[c];
        bool in_record = r > 0;
        scan_fasta(text, cut[c], cut[c + 1],
                   [&] { if (out_offsets) out_offsets[r] = at; r++; in_record = true; },
                   [&](uint64_t a, uint64_t b) {
                       if (!in_record) return;
                       if (out_seqs) memcpy(out_seqs + at, text + a, b - a);
                       at += b - a;
                   });
    });
    if (out_offsets) out_offsets[n] = w;
    return BIGSI_OK;
}

extern "C" int bigsi_hip_format_results(int format, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, const char *threshold_text,
                                        const char *citation_text, int exact, const uint32_t *num_unique, const uint64_t *hit_offsets,
                                        const uint32_t *colours, const uint32_t *counts, const char *names, const uint64_t *name_offsets,
                                        const uint8_t *name_deleted, uint64_t n_names, uint32_t threads, char **out_text, uint64_t *out_bytes)
{
    return bigsi_hip_format_results_scored(format, seqs, offsets, n_seqs, threshold_text, citation_text, exact, num_unique, hit_offsets, colours, counts,
                                           names, name_offsets, name_deleted, n_names, nullptr, threads, out_text, out_bytes);
}

extern "C" int bigsi_hip_format_results_scored(int format, const char *seqs, const uint64_t *offsets, uint64_t n_seqs, const char *threshold_text,
                                               const char *citation_text, int exact, const uint32_t *num_unique, const uint64_t *hit_offsets,
                                               const uint32_t *colours, const uint32_t *counts, const char *names, const uint64_t *name_offsets,
                                               const uint8_t *name_deleted, uint64_t n_names, const bigsi_hip_scored_text *scored, uint32_t threads,
                                               char **out_text, uint64_t *out_bytes)
{
    if (!out_text || !out_bytes || (format != 0 && format != 1)) return fail(BIGSI_ERR_INVALID, "bad argument");
    if (n_seqs && (!seqs || !offsets || !num_unique || !hit_offsets)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (format == 0 && (!threshold_text || !citation_text)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    const uint64_t n_hits = n_seqs ? hit_offsets[n_seqs] : 0;
    if (n_hits && (!colours || (!exact && !counts) || (n_names && (!names || !name_offsets || !name_deleted)))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (scored && n_hits && (!scored->scores || !scored->bits || !scored->bit_offsets || !scored->evalue || !scored->pvalue || !scored->log_evalue || !scored->log_pvalue))
        return fail(BIGSI_ERR_INVALID, "NULL score column");
    char *const caller_buf = *out_text;             // non-NULL: the caller's own buffer of *out_bytes bytes (e.g. the body of a string object)
    const uint64_t caller_cap = *out_bytes;
    if (!caller_buf) *out_bytes = 0;
    // what the reference does not answer with text is the caller's to raise (in the order of the records)
    for (uint64_t r = 0; r < n_seqs; r++)
        if (num_unique[r] == 0) return fail(BIGSI_ERR_STATE, "record %llu has no k-mers: the reference raises (graph/bigsi.py:35-44, utils/fncts.py:24-25)", (unsigned long long)r);
    if (exact)
        for (uint64_t t = 0; t < n_hits; t++)
            if (colours[t] >= n_names) return fail(BIGSI_ERR_STATE, "colour %u has no sample name: the reference raises KeyError", colours[t]);
    if (scored)          // score=True on a one-k-mer query that has hits: IndexError in the reference (graph/bigsi.py:47-56, 236)
        for (uint64_t t = 0; t < n_hits; t++)
            if (scored->scores[t].num_kmers < 2) return fail(BIGSI_ERR_STATE, "a scored hit of a query with %u k-mer(s): the reference raises", scored->scores[t].num_kmers);
    Job j{format, seqs, offsets, threshold_text, threshold_text ? strlen(threshold_text) : 0, citation_text, citation_text ? strlen(citation_text) : 0,
          exact != 0, num_unique, hit_offsets, colours, counts, names, name_offsets, name_deleted, n_names, n_hits ? scored : nullptr};
    const uint64_t per = 4096, n_blocks = (n_seqs + per - 1) / per;
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<uint64_t> at(n_blocks + 1, 0);
    std::vector<std::string> parts(j.sc ? n_blocks : 0);
    run_blocks(n_blocks, threads, [&](uint64_t b) {
        std::vector<uint64_t> order;
        if (j.sc) {
            StringSink s;
            format_records(j, b * per, std::min(n_seqs, (b + 1) * per), s, order);
            parts[b].swap(s.t);
            at[b + 1] = parts[b].size();
            return;
        }
        CountSink s;
        format_records(j, b * per, std::min(n_seqs, (b + 1) * per), s, order);
        at[b + 1] = s.n;
    });
    const uint64_t head = format == 0 ? (n_seqs ? 2 : 0) : 0, tail = format == 0 ? 2 : 0;      // "[\n" ... "\n]", or "[]"
    at[0] = head;
    for (uint64_t b = 0; b < n_blocks; b++) at[b + 1] += at[b];
    const uint64_t total = at[n_blocks] + tail;
    if (caller_buf && caller_cap < total) {
        *out_bytes = total;
        return fail(BIGSI_ERR_CAPACITY, "the text is %llu bytes, the buffer holds %llu", (unsigned long long)total, (unsigned long long)caller_cap);
    }
    char *text = caller_buf ? caller_buf : static_cast<char *>(malloc(total + 1));
    if (!text) return fail(BIGSI_ERR_NOMEM, "no memory for %llu bytes of text", (unsigned long long)total);
    if (format == 0) {
        if (n_seqs) { memcpy(text, "[\n", 2); memcpy(text + total - 2, "\n]", 2); }
        else memcpy(text, "[]", 2);
    }
    run_blocks(n_blocks, threads, [&](uint64_t b) {
        if (j.sc) { memcpy(text + at[b], parts[b].data(), parts[b].size()); return; }
        WriteSink s{text + at[b]};
        std::vector<uint64_t> order;
        format_records(j, b * per, std::min(n_seqs, (b + 1) * per), s, order);
    });
    if (!caller_buf || caller_cap > total) text[total] = 0;
    *out_text = text;
    *out_bytes = total;
    return BIGSI_OK;
}

extern "C" void bigsi_hip_free_text(char *text) { free(text); }
