// Host-side text of the batch front-end: the FASTA records a bulk search reads and the JSON / CSV text it
// returns, for callers that want the reference's text and not Python objects.  No device code: what the reference does per record
// in Python -- pyfasta's records, json.dumps(list of records, indent=4), csv.writer rows (bigsi/__main__.py:41-72, 261-314) -- done
// over the arrays bigsi_hip_search_stream leaves, by a few host threads.  bigsi_amd/frontend.py calls these for unscored bulk
// searches of ASCII files and keeps its Python route for everything else (scores, non-ASCII text, queries on which the reference
// raises); tests/test_frontend_text.py pins both routes to the same text and to golden G9.
#include <emmintrin.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
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
                put_percent(s, f, u);
                PUT_LIT(s, ",\n                \"num_kmers\": ");
                s.put_u32(u);
                PUT_LIT(s, ",\n                \"num_kmers_found\": ");
                s.put_u32(f);
                PUT_LIT(s, ",\n                \"sample_name\": ");
                put_json_string(s, j.names + j.name_offsets[c], j.name_offsets[c + 1] - j.name_offsets[c]);
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
    if (!out_text || !out_bytes || (format != 0 && format != 1)) return fail(BIGSI_ERR_INVALID, "bad argument");
    if (n_seqs && (!seqs || !offsets || !num_unique || !hit_offsets)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    if (format == 0 && (!threshold_text || !citation_text)) return fail(BIGSI_ERR_INVALID, "NULL argument");
    const uint64_t n_hits = n_seqs ? hit_offsets[n_seqs] : 0;
    if (n_hits && (!colours || (!exact && !counts) || (n_names && (!names || !name_offsets || !name_deleted)))) return fail(BIGSI_ERR_INVALID, "NULL argument");
    char *const caller_buf = *out_text;             // non-NULL: the caller's own buffer of *out_bytes bytes (e.g. the body of a string object)
    const uint64_t caller_cap = *out_bytes;
    if (!caller_buf) *out_bytes = 0;
    // what the reference does not answer with text is the caller's to raise (in the order of the records)
    for (uint64_t r = 0; r < n_seqs; r++)
        if (num_unique[r] == 0) return fail(BIGSI_ERR_STATE, "record %llu has no k-mers: the reference raises (graph/bigsi.py:35-44, utils/fncts.py:24-25)", (unsigned long long)r);
    if (exact)
        for (uint64_t t = 0; t < n_hits; t++)
            if (colours[t] >= n_names) return fail(BIGSI_ERR_STATE, "colour %u has no sample name: the reference raises KeyError", colours[t]);
    Job j{format, seqs, offsets, threshold_text, threshold_text ? strlen(threshold_text) : 0, citation_text, citation_text ? strlen(citation_text) : 0,
          exact != 0, num_unique, hit_offsets, colours, counts, names, name_offsets, name_deleted, n_names};
    const uint64_t per = 4096, n_blocks = (n_seqs + per - 1) / per;
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<uint64_t> at(n_blocks + 1, 0);
    run_blocks(n_blocks, threads, [&](uint64_t b) {
        CountSink s;
        std::vector<uint64_t> order;
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
