// CPython extension: the reference's result dicts (BigsiQueryResult.todict, graph/bigsi.py:91-126, with the score fields of
// scoring/score.py:96-121 when score=True) built straight from the arrays a streaming search returns -- what BIGSI.search_stream
// otherwise does in a Python loop per hit (2-4 us per dict; here ~0.3-0.8).  Host code only; the values are the ones the Python
// route computes (tests/test_results_ext.py holds the two routes equal), the percent through the same py_round2 as the device's
// scorer.  Built by bigsi_amd/pyext_build.sh (g++, -ffp-contract=off) next to the package; graph/bigsi.py uses it when importable.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "csrc/bigsi_score.hpp"

namespace {

struct Buf {
    Py_buffer b{};
    bool held = false;
    ~Buf() { if (held) PyBuffer_Release(&b); }
    bool get(PyObject *o, const char *what, Py_ssize_t itemsize)
    {
        if (PyObject_GetBuffer(o, &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return false;
        held = true;
        if (b.itemsize != itemsize) {
            PyErr_Format(PyExc_TypeError, "%s: items of %zd bytes expected, got %zd", what, itemsize, b.itemsize);
            return false;
        }
        return true;
    }
    Py_ssize_t n() const { return b.len / b.itemsize; }
};

struct Column {
    Buf buf;
    bool is_float = false;
};

// ---- values written straight into a freshly copied dict's entry table (CPython 3.10 only, verified before every use)
// PyDict_Copy of the 22-key template clones its combined key table: entry i holds key i, in insertion order.  Replacing the 22 values
// through PyDict_SetItem costs 22 hash-table lookups per hit (0.26 of the 0.70 us a scored dict takes); the values can instead be
// stored into the entries directly -- what insertdict does once it has found the slot -- IF the object layout is the one this code
// was written against.  That layout (Objects/dict-common.h of CPython 3.10: PyDictKeysObject {dk_refcnt, dk_size, dk_lookup,
// dk_usable, dk_nentries, dk_indices[]}, entries {me_hash, me_key, me_value} behind dk_size one-byte indices) is not public, so:
// compiled only for 3.10, enabled only when the running interpreter says 3.10, and before each dict is touched its table is CHECKED
// read-only (combined table, 64 slots, 22 entries, entry i's key IS key i); anything unexpected -> the PyDict_SetItem route.
#if PY_VERSION_HEX >= 0x030A0000 && PY_VERSION_HEX < 0x030B0000
#define BIGSI_FAST_DICT 1
struct KeyEntry310 { Py_hash_t me_hash; PyObject *me_key; PyObject *me_value; };
struct Keys310 { Py_ssize_t dk_refcnt, dk_size; void *dk_lookup; Py_ssize_t dk_usable, dk_nentries; char dk_indices[1]; };
bool g_fast_dict = false;          // OFF until _results.fast_dict(True): bigsi_amd/graph/bigsi.py switches it on only after its import-time self-test
                                   // (one batch of dicts built both ways: equal, then mutated, copied, serialised: still equal) has passed
bool g_split_dict = true;          // _results.fast_dict(True, False): direct stores into copies of a combined template (A/B, tests)

// the 22 value slots of `d` (a fresh PyDict_Copy of the template), or nullptr if the table is not what is expected
inline KeyEntry310 *entries_of(PyObject *d, PyObject *const *k)
{
    PyDictObject *mp = reinterpret_cast<PyDictObject *>(d);
    if (mp->ma_values != nullptr || mp->ma_used != 22) return nullptr;                    // a split table, or not our template
    Keys310 *keys = reinterpret_cast<Keys310 *>(mp->ma_keys);
    if (keys->dk_size != 64 || keys->dk_nentries != 22 || keys->dk_refcnt != 1) return nullptr;      // (22 entries need 64 slots: one-byte indices)
    KeyEntry310 *e = reinterpret_cast<KeyEntry310 *>(keys->dk_indices + keys->dk_size);
    for (int i = 0; i < 22; i++)
        if (e[i].me_key != k[i] || e[i].me_value != Py_None) return nullptr;
    return e;
}

// ---- the template as a SPLIT-table dict (CPython 3.10 only, verified like the above).  Copying a combined 22-key dict clones a
// 1.1 KB key table (malloc + memcpy, and a free when the dict dies): 1.2 us + 0.6 us per dict in a plain Python loop.  The dicts of a
// class's instances share ONE key table and own only their values array (42 pointers, from the small-object allocator): 0.32 + 0.08 us.
// PyDict_Copy keeps that sharing, so the template is the __dict__ of an instance of a private class whose 22 attributes are the result
// keys, set in order (22 for scored hits, 4 for plain ones: two classes); every result dict is then a copy of it -- a real dict in every respect (a consumer that adds a key extends the
// shared table or converts that one dict, as for any instance dict) -- and the values go straight into its values array.  The
// template is untracked by the collector (it holds None and strs only, and so do its copies until a consumer stores something else):
// what CPython itself does to such dicts at its next collection.
struct SplitTemplate {
    int n = 0;                                   // keys
    Py_ssize_t dk_size = 0;                      // slots of the shared table a class of n attributes ends up with (one-byte indices)
    PyObject *obj = nullptr;                     // the instance; its __dict__ is the template
    PyObject *keys[22] = {};                     // the keys it was made for (strong references)
    PyObject *interned[22] = {};                 // the key objects the shared table actually holds (borrowed from it)
};
SplitTemplate g_split22{22, 64}, g_split4{4, 8};

// new reference to the split template for these keys, or nullptr (no error set): the caller then takes the combined route
PyObject *split_template(SplitTemplate &T, PyObject *const *k)
{
    const int n = T.n;
    bool same = T.obj != nullptr;
    for (int i = 0; i < n && same; i++)
        if (T.keys[i] != k[i]) {
            const int eq = PyObject_RichCompareBool(T.keys[i], k[i], Py_EQ);
            if (eq != 1) { same = false; PyErr_Clear(); }
        }
    if (!same) {
        Py_CLEAR(T.obj);
        for (int i = 0; i < n; i++) {
            if (!PyUnicode_Check(k[i])) return nullptr;
            Py_CLEAR(T.keys[i]);
        }
        PyObject *type = PyObject_CallFunction(reinterpret_cast<PyObject *>(&PyType_Type), "s(){}", n == 22 ? "_ScoredHit" : "_Hit");
        PyObject *obj = type ? PyObject_CallObject(type, nullptr) : nullptr;
        Py_XDECREF(type);
        if (!obj) { PyErr_Clear(); return nullptr; }
        for (int i = 0; i < n; i++)
            if (PyObject_SetAttr(obj, k[i], Py_None) != 0) { PyErr_Clear(); Py_DECREF(obj); return nullptr; }
        T.obj = obj;
        for (int i = 0; i < n; i++) { Py_INCREF(k[i]); T.keys[i] = k[i]; T.interned[i] = nullptr; }
    }
    PyObject *d = PyObject_GenericGetDict(T.obj, nullptr);
    if (!d) { PyErr_Clear(); return nullptr; }
    PyDictObject *mp = reinterpret_cast<PyDictObject *>(d);
    bool ok = PyDict_CheckExact(d) && mp->ma_values != nullptr && mp->ma_used == n;
    if (ok) {
        Keys310 *keys = reinterpret_cast<Keys310 *>(mp->ma_keys);
        ok = keys->dk_size == T.dk_size && keys->dk_nentries >= n;
        KeyEntry310 *e = reinterpret_cast<KeyEntry310 *>(keys->dk_indices + keys->dk_size);
        for (int i = 0; i < n && ok; i++) {
            if (!T.interned[i]) {                  // first use: entry i must hold key i (an interned equal of it)
                ok = e[i].me_key == k[i] || PyObject_RichCompareBool(e[i].me_key, k[i], Py_EQ) == 1;
                if (ok) T.interned[i] = e[i].me_key;
            } else {
                ok = e[i].me_key == T.interned[i];
            }
            ok = ok && mp->ma_values[i] == Py_None;
        }
        PyErr_Clear();
    }
    if (!ok) { Py_DECREF(d); Py_CLEAR(T.obj); return nullptr; }
    if (PyObject_GC_IsTracked(d)) PyObject_GC_UnTrack(d);
    return d;
}

// the values array of `d` (a fresh PyDict_Copy of the split template `tmpl` of n keys), or nullptr if it is not what is expected
inline PyObject **values_of(PyObject *d, PyObject *tmpl, int n)
{
    PyDictObject *mp = reinterpret_cast<PyDictObject *>(d);
    if (mp->ma_values == nullptr || mp->ma_keys != reinterpret_cast<PyDictObject *>(tmpl)->ma_keys || mp->ma_used != n) return nullptr;
    for (int i = 0; i < n; i++)
        if (mp->ma_values[i] != Py_None) return nullptr;
    return mp->ma_values;
}
#else
#define BIGSI_FAST_DICT 0
#endif


// build(nu, off, cols, cnts, exact, names, keys, columns, text, text_start, text_len, lo, hi) -> [results of sequence lo, ..., hi - 1]
//   nu uint32[n], off int64[n + 1], cols / cnts uint32[hits]: what bigsi_hip_search_stream returns (hits of a sequence ascending by colour)
//   names: list, names[c] = sample name of colour c, or None for a deleted sample (dropped); colours beyond the list are dropped on
//          the thresholded route (inexact_filter zips with range(num_samples)) -- the caller checks the exact route's KeyError itself
//   keys: the dict keys in order: 4 (percent_kmers_found, num_kmers, num_kmers_found, sample_name), or those + 17 score keys + "kmer-presence"
//   columns: None, or 18 arrays over the hits: percent_kmers_found, then the 17 score fields (float64 or int64 each)
//   text / text_start / text_len: the presence characters of all hits as one str, hit t = text[text_start[t] : text_start[t] + text_len[t]]
PyObject *build(PyObject *, PyObject *args)
{
    PyObject *o_nu, *o_off, *o_cols, *o_cnts, *names, *keys, *columns, *text, *o_tstart, *o_tlen;
    int exact;
    Py_ssize_t lo, hi;
    if (!PyArg_ParseTuple(args, "OOOOpOOOOOOnn", &o_nu, &o_off, &o_cols, &o_cnts, &exact, &names, &keys, &columns, &text, &o_tstart, &o_tlen, &lo, &hi))
        return nullptr;
    Buf nu, off, cols, cnts, tstart, tlen;
    if (!nu.get(o_nu, "nu", 4) || !off.get(o_off, "off", 8) || !cols.get(o_cols, "cols", 4) || !cnts.get(o_cnts, "cnts", 4)) return nullptr;
    if (!PyList_Check(names) || !PyTuple_Check(keys)) { PyErr_SetString(PyExc_TypeError, "names must be a list, keys a tuple"); return nullptr; }
    const bool scored = columns != Py_None;
    const Py_ssize_t n_keys = PyTuple_GET_SIZE(keys), n_seqs = nu.n();
    if (n_keys != (scored ? 22 : 4)) { PyErr_SetString(PyExc_ValueError, "4 keys, or 22 with score columns"); return nullptr; }
    if (lo < 0 || hi < lo || hi > n_seqs || off.n() < n_seqs + 1) { PyErr_SetString(PyExc_ValueError, "bad sequence range"); return nullptr; }
    const uint32_t *p_nu = static_cast<const uint32_t *>(nu.b.buf), *p_col = static_cast<const uint32_t *>(cols.b.buf), *p_cnt = static_cast<const uint32_t *>(cnts.b.buf);
    const int64_t *p_off = static_cast<const int64_t *>(off.b.buf);
    const Py_ssize_t n_hits = cols.n(), n_names = PyList_GET_SIZE(names);
    if (p_off[hi] > n_hits || cnts.n() < n_hits) { PyErr_SetString(PyExc_ValueError, "hit offsets beyond the hit arrays"); return nullptr; }
    std::vector<Column> col(scored ? 18 : 0);
    const int64_t *p_tstart = nullptr, *p_tlen = nullptr;
    if (scored) {
        if (!PyTuple_Check(columns) || PyTuple_GET_SIZE(columns) != 18 || !PyUnicode_Check(text)) { PyErr_SetString(PyExc_TypeError, "columns: a tuple of 18 arrays; text: a str"); return nullptr; }
        for (int i = 0; i < 18; i++) {
            if (!col[i].buf.get(PyTuple_GET_ITEM(columns, i), "score column", 8)) return nullptr;
            const char *f = col[i].buf.b.format ? col[i].buf.b.format : "";
            while (*f == '<' || *f == '=' || *f == '@') f++;
            if (*f == 'd') col[i].is_float = true;
            else if (*f != 'l' && *f != 'q') { PyErr_Format(PyExc_TypeError, "score column %d: float64 or int64 expected, got '%s'", i, col[i].buf.b.format); return nullptr; }
            if (col[i].buf.n() < n_hits) { PyErr_SetString(PyExc_ValueError, "score column shorter than the hit arrays"); return nullptr; }
        }
        if (!tstart.get(o_tstart, "text_start", 8) || !tlen.get(o_tlen, "text_len", 8)) return nullptr;
        if (tstart.n() < n_hits || tlen.n() < n_hits) { PyErr_SetString(PyExc_ValueError, "text offsets shorter than the hit arrays"); return nullptr; }
        p_tstart = static_cast<const int64_t *>(tstart.b.buf);
        p_tlen = static_cast<const int64_t *>(tlen.b.buf);
    }
    // (plain hits, CPython 3.10: copies of a 4-key split-table template with the values stored straight into them, as build_scored)
    PyObject *tmpl4 = nullptr;
#if BIGSI_FAST_DICT
    if (!scored && g_fast_dict && g_split_dict && strncmp(Py_GetVersion(), "3.10.", 5) == 0) {
        PyObject *k4[4];
        for (int i = 0; i < 4; i++) k4[i] = PyTuple_GET_ITEM(keys, i);
        tmpl4 = split_template(g_split4, k4);
    }
#endif
    struct Drop { PyObject *&o; ~Drop() { Py_XDECREF(o); } } drop_tmpl4{tmpl4};
    PyObject *out = PyList_New(hi - lo);
    if (!out) return nullptr;
    std::vector<int64_t> order;
    for (Py_ssize_t i = lo; i < hi; i++) {
        const uint32_t u = p_nu[i];
        order.clear();
        for (int64_t t = p_off[i]; t < p_off[i + 1]; t++) {
            const uint32_t c = p_col[t];
            if ((Py_ssize_t)c >= n_names || PyList_GET_ITEM(names, c) == Py_None) continue;
            order.push_back(t);
        }
        if (!exact && order.size() > 1)          // inexact_filter: stable sort by count, descending (graph/bigsi.py:215-229)
            std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return p_cnt[a] > p_cnt[b]; });
        PyObject *res = PyList_New((Py_ssize_t)order.size());
        if (!res) { Py_DECREF(out); return nullptr; }
        PyList_SET_ITEM(out, i - lo, res);
        if (order.empty()) continue;
        PyObject *py_u = PyLong_FromUnsignedLong(u);
        if (!py_u) { Py_DECREF(out); return nullptr; }
        for (size_t r = 0; r < order.size(); r++) {
            const int64_t t = order[r];
            const uint32_t f = exact ? u : p_cnt[t];
            PyObject *d = tmpl4 ? PyDict_Copy(tmpl4) : _PyDict_NewPresized(n_keys);      // (no growth steps on the way to 22 keys)
            if (!d) { Py_DECREF(py_u); Py_DECREF(out); return nullptr; }
            PyList_SET_ITEM(res, (Py_ssize_t)r, d);
            bool ok = true;
#if BIGSI_FAST_DICT
            PyObject **vals = tmpl4 ? values_of(d, tmpl4, 4) : nullptr;
#endif
            auto put = [&](Py_ssize_t k, PyObject *v) {      // steals v
                if (!v) { ok = false; return; }
#if BIGSI_FAST_DICT
                if (vals) {                                  // values[k] belongs to key k and holds None: the value takes its place
                    vals[k] = v;
                    Py_DECREF(Py_None);
                    return;
                }
#endif
                if (ok && PyDict_SetItem(d, PyTuple_GET_ITEM(keys, k), v) != 0) ok = false;
                Py_DECREF(v);
            };
            // percent_kmers_found = round(100 * float(found) / num_kmers, 2) (graph/bigsi.py:97-99); a scored search carries K6's value
            put(0, PyFloat_FromDouble(scored ? static_cast<const double *>(col[0].buf.b.buf)[t] : bigsi_score::py_round2(100.0 * (double)f / (double)u)));
            Py_INCREF(py_u);
            put(1, py_u);
            put(2, PyLong_FromUnsignedLong(f));
            PyObject *nm = PyList_GET_ITEM(names, p_col[t]);
            Py_INCREF(nm);
            put(3, nm);
            if (!PyUnicode_CheckExact(nm) && !PyObject_GC_IsTracked(d)) PyObject_GC_Track(d);      // (copies of the untracked template: only atomic values may go unseen by the collector)
            if (scored) {
                for (int j = 1; j < 18 && ok; j++)
                    put(3 + j, col[j].is_float ? PyFloat_FromDouble(static_cast<const double *>(col[j].buf.b.buf)[t])
                                               : PyLong_FromLongLong(static_cast<const int64_t *>(col[j].buf.b.buf)[t]));
                put(21, PyUnicode_Substring(text, (Py_ssize_t)p_tstart[t], (Py_ssize_t)(p_tstart[t] + p_tlen[t])));
            }
            if (!ok) { Py_DECREF(py_u); Py_DECREF(out); return nullptr; }
        }
        Py_DECREF(py_u);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------------
// build_scored: the same lists for score=True, straight from what K6 returns -- 64-byte score records + packed presence bits --
// without the intermediate numpy columns, the one big text str and a substring per hit of build() above (1.7 us per hit there,
// most of it outside the dict: 18 column arrays made, 18 x indexed, a 1 KB substring copied out of a 10 MB str).  Per hit:
//   * the integer / quotient fields of Scorer.score (score.py:99-111) from the record: plain IEEE arithmetic, the same
//     expressions as scoring.score_columns (this file is compiled with -ffp-contract=off);
//   * evalue, pvalue, log_evalue, log_pvalue arrive as four float64 arrays: numpy's exp / log10 (what the reference calls) are
//     not reproducible by libm bit for bit, so they stay numpy's (scoring.score_transcendentals);
//   * "kmer-presence": the hit's n bits (bitarray order: position p = byte p / 8, mask 0x80 >> p % 8) -> n characters, 8 at a
//     time through a 256-entry table, written into the new str's own body;
//   * the dict is a COPY of a 22-key template whose values are then replaced in key order -- no probing for free slots, no growth.
//     On CPython 3.10 the template is a split-table dict (the copies share its key table and own a values array each: see
//     split_template below); elsewhere a combined one (one allocation + memcpy of the key table per copy);
//   * small non-negative integers (lengths, identities, mismatch counts: < 2^14 for anything up to 16 kbp) come from a table of
//     ready-made int objects.
// build_scored(nu, off, cols, cnts, exact, names, keys, rec, bits, boff, trans, db_size_unused, lo, hi)
//   rec: buffer of 64-byte records (HIT_SCORE_DTYPE), bits uint8[], boff uint64[hits + 1] (byte offsets, multiples of 8),
//   trans: tuple of 4 float64 arrays over the hits
struct IntCache {
    std::vector<PyObject *> v;
    ~IntCache() { /* objects are immortal for the life of the module */ }
    PyObject *get(int64_t x)      // new reference
    {
        if (x >= 0 && x < (int64_t)(1 << 14)) {
            if (v.empty()) v.assign(1 << 14, nullptr);
            PyObject *&o = v[(size_t)x];
            if (!o) { o = PyLong_FromLongLong(x); if (!o) return nullptr; }
            Py_INCREF(o);
            return o;
        }
        return PyLong_FromLongLong(x);
    }
};
IntCache g_ints;

struct Rec {                 // bigsi_score::HitScore / HIT_SCORE_DTYPE
    double score, min_score, max_score, percent;
    int64_t max_mm, min_mm, mm;
    uint32_t n, reserved;
};
static_assert(sizeof(Rec) == 64, "score record layout");

uint64_t g_lut[256];
bool g_lut_ready = false;
void lut_init()
{
    for (int b = 0; b < 256; b++) {
        uint64_t w = 0;
        for (int i = 0; i < 8; i++) w |= (uint64_t)('0' + ((b >> (7 - i)) & 1)) << (8 * i);      // character i = bit 0x80 >> i (little-endian store)
        g_lut[b] = w;
    }
    g_lut_ready = true;
}

PyObject *presence_str(const uint8_t *bits, uint32_t n)
{
    PyObject *s = PyUnicode_New((Py_ssize_t)n, 127);
    if (!s) return nullptr;
    uint8_t *dst = static_cast<uint8_t *>(PyUnicode_DATA(s));
    const uint32_t full = n >> 3;
    for (uint32_t i = 0; i < full; i++) memcpy(dst + 8 * (size_t)i, &g_lut[bits[i]], 8);
    if (n & 7u) memcpy(dst + 8 * (size_t)full, &g_lut[bits[full]], n & 7u);
    return s;
}

PyObject *build_scored(PyObject *, PyObject *args)
{
    PyObject *o_nu, *o_off, *o_cols, *o_cnts, *names, *keys, *o_rec, *o_bits, *o_boff, *trans;
    int exact;
    Py_ssize_t lo, hi;
    if (!PyArg_ParseTuple(args, "OOOOpOOOOOOnn", &o_nu, &o_off, &o_cols, &o_cnts, &exact, &names, &keys, &o_rec, &o_bits, &o_boff, &trans, &lo, &hi))
        return nullptr;
    Buf nu, off, cols, cnts, rec, bits, boff, tr[4];
    if (!nu.get(o_nu, "nu", 4) || !off.get(o_off, "off", 8) || !cols.get(o_cols, "cols", 4) || !cnts.get(o_cnts, "cnts", 4)) return nullptr;
    if (!rec.get(o_rec, "rec", 64) || !bits.get(o_bits, "bits", 1) || !boff.get(o_boff, "boff", 8)) return nullptr;
    if (!PyList_Check(names) || !PyTuple_Check(keys) || PyTuple_GET_SIZE(keys) != 22) { PyErr_SetString(PyExc_TypeError, "names must be a list, keys a tuple of 22"); return nullptr; }
    if (!PyTuple_Check(trans) || PyTuple_GET_SIZE(trans) != 4) { PyErr_SetString(PyExc_TypeError, "trans: a tuple of 4 float64 arrays"); return nullptr; }
    const Py_ssize_t n_seqs = nu.n(), n_hits = cols.n(), n_names = PyList_GET_SIZE(names);
    for (int i = 0; i < 4; i++) {
        if (!tr[i].get(PyTuple_GET_ITEM(trans, i), "trans", 8)) return nullptr;
        const char *f = tr[i].b.format ? tr[i].b.format : "";
        while (*f == '<' || *f == '=' || *f == '@') f++;
        if (*f != 'd' || tr[i].n() < n_hits) { PyErr_SetString(PyExc_TypeError, "trans: float64 arrays over the hits expected"); return nullptr; }
    }
    if (lo < 0 || hi < lo || hi > n_seqs || off.n() < n_seqs + 1) { PyErr_SetString(PyExc_ValueError, "bad sequence range"); return nullptr; }
    const uint32_t *p_nu = static_cast<const uint32_t *>(nu.b.buf), *p_col = static_cast<const uint32_t *>(cols.b.buf), *p_cnt = static_cast<const uint32_t *>(cnts.b.buf);
    const int64_t *p_off = static_cast<const int64_t *>(off.b.buf);
    const Rec *p_rec = static_cast<const Rec *>(rec.b.buf);
    const uint8_t *p_bits = static_cast<const uint8_t *>(bits.b.buf);
    const uint64_t *p_boff = static_cast<const uint64_t *>(boff.b.buf);
    const double *p_tr[4];
    for (int i = 0; i < 4; i++) p_tr[i] = static_cast<const double *>(tr[i].b.buf);
    if (p_off[hi] > n_hits || cnts.n() < n_hits || rec.n() < n_hits || boff.n() < n_hits + 1) { PyErr_SetString(PyExc_ValueError, "hit offsets beyond the hit arrays"); return nullptr; }
    if (n_hits && p_boff[n_hits] > (uint64_t)bits.n()) { PyErr_SetString(PyExc_ValueError, "bit offsets beyond the bits"); return nullptr; }
    if (!g_lut_ready) lut_init();
    PyObject *k[22];
    for (int i = 0; i < 22; i++) k[i] = PyTuple_GET_ITEM(keys, i);
    PyObject *tmpl = nullptr;
#if BIGSI_FAST_DICT
    static const bool is_310 = strncmp(Py_GetVersion(), "3.10.", 5) == 0;
    const bool fast = g_fast_dict && is_310;
    const bool split = fast && g_split_dict && (tmpl = split_template(g_split22, k)) != nullptr;
#endif
    if (!tmpl) {
        tmpl = PyDict_New();
        if (!tmpl) return nullptr;
        for (int i = 0; i < 22; i++)
            if (PyDict_SetItem(tmpl, k[i], Py_None) != 0) { Py_DECREF(tmpl); return nullptr; }
    }
    PyObject *out = PyList_New(hi - lo);
    if (!out) { Py_DECREF(tmpl); return nullptr; }
    std::vector<int64_t> order;
    bool ok = true;
    for (Py_ssize_t i = lo; i < hi && ok; i++) {
        const uint32_t u = p_nu[i];
        order.clear();
        for (int64_t t = p_off[i]; t < p_off[i + 1]; t++) {
            const uint32_t c = p_col[t];
            if ((Py_ssize_t)c >= n_names || PyList_GET_ITEM(names, c) == Py_None) continue;
            order.push_back(t);
        }
        if (!exact && order.size() > 1)          // inexact_filter: stable sort by count, descending (graph/bigsi.py:215-229)
            std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return p_cnt[a] > p_cnt[b]; });
        PyObject *res = PyList_New((Py_ssize_t)order.size());
        if (!res) { ok = false; break; }
        PyList_SET_ITEM(out, i - lo, res);
        if (order.empty()) continue;
        PyObject *py_u = g_ints.get(u);
        if (!py_u) { ok = false; break; }
        for (size_t r = 0; r < order.size() && ok; r++) {
            const int64_t t = order[r];
            const Rec &rc = p_rec[t];
            const uint32_t f = exact ? u : p_cnt[t];
            PyObject *d = PyDict_Copy(tmpl);
            if (!d) { ok = false; break; }
            PyList_SET_ITEM(res, (Py_ssize_t)r, d);
#if BIGSI_FAST_DICT
            PyObject **vals = split ? values_of(d, tmpl, 22) : nullptr;
            KeyEntry310 *slots = (fast && !split) ? entries_of(d, k) : nullptr;
#endif
            auto put = [&](int kk, PyObject *v) {      // steals v
                if (!v) { ok = false; return; }
#if BIGSI_FAST_DICT
                if (vals) {                             // values[kk] belongs to key kk and holds None: the value takes its place
                    vals[kk] = v;
                    Py_DECREF(Py_None);
                    return;
                }
                if (slots) {                            // entry kk holds key kk and None: the value takes None's place
                    slots[kk].me_value = v;
                    Py_DECREF(Py_None);
                    return;
                }
#endif
                if (ok && PyDict_SetItem(d, k[kk], v) != 0) ok = false;
                Py_DECREF(v);
            };
            if (rc.n == 0) { PyErr_SetString(PyExc_ZeroDivisionError, "division by zero"); ok = false; break; }      // score.py:99-100
            const int64_t seq_len = (int64_t)rc.n + 30;                  // n + k - 1, k = 31 (score.py:61,99)
            const double fl = (double)seq_len;
            const int64_t max_nident = seq_len - rc.min_mm, nident = seq_len - rc.mm, min_nident = seq_len - rc.max_mm;
            put(0, PyFloat_FromDouble(rc.percent));
            Py_INCREF(py_u);
            put(1, py_u);
            put(2, g_ints.get(f));
            PyObject *nm = PyList_GET_ITEM(names, p_col[t]);
            Py_INCREF(nm);
            put(3, nm);
            if (!PyUnicode_CheckExact(nm) && !PyObject_GC_IsTracked(d)) PyObject_GC_Track(d);      // (as in build: a name that is not a plain str may hold references)
            put(4, PyFloat_FromDouble(rc.score));
            put(5, PyFloat_FromDouble(rc.min_score));
            put(6, PyFloat_FromDouble(rc.max_score));
            put(7, g_ints.get(rc.max_mm));
            put(8, g_ints.get(rc.min_mm));
            put(9, g_ints.get(rc.mm));
            put(10, g_ints.get(max_nident));
            put(11, g_ints.get(nident));
            put(12, g_ints.get(min_nident));
            put(13, PyFloat_FromDouble(100.0 * (double)nident / fl));           // pident = 100 * float(nident) / seq_len (score.py:106-111)
            put(14, PyFloat_FromDouble(100.0 * (double)max_nident / fl));
            put(15, PyFloat_FromDouble(100.0 * (double)min_nident / fl));
            put(16, g_ints.get(seq_len));
            for (int j = 0; j < 4; j++) put(17 + j, PyFloat_FromDouble(p_tr[j][t]));
            put(21, presence_str(p_bits + p_boff[t], rc.n));
        }
        Py_DECREF(py_u);
    }
    Py_DECREF(tmpl);
    if (!ok) { Py_DECREF(out); return nullptr; }
    return out;
}

// pack_rows(raws, block, last_byte_mask): rows as a KV store hands them over (a list of bytes objects: bigsi/storage/base.py:58-59,
// 96-99) -> the rows of `block` (a writable C-contiguous uint8[n, rb] buffer), each cut or zero-extended to rb bytes, the last byte
// ANDed with last_byte_mask (columns beyond number_of_cols are not part of the index).  The copies run on a few threads without
// the GIL: what migrate_index did in a Python loop per row (the importer's own 7-8 GB/s of round 4).
PyObject *pack_rows(PyObject *, PyObject *args)
{
    PyObject *raws, *o_block;
    int mask;
    if (!PyArg_ParseTuple(args, "OOi", &raws, &o_block, &mask)) return nullptr;
    if (!PyList_Check(raws)) { PyErr_SetString(PyExc_TypeError, "raws must be a list"); return nullptr; }
    Py_buffer blk;
    if (PyObject_GetBuffer(o_block, &blk, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_ND) != 0) return nullptr;
    const Py_ssize_t n = PyList_GET_SIZE(raws);
    if (blk.ndim != 2 || blk.itemsize != 1 || blk.shape[0] != n) {
        PyBuffer_Release(&blk);
        PyErr_SetString(PyExc_ValueError, "block must be a uint8[len(raws), rb] array");
        return nullptr;
    }
    const size_t rb = (size_t)blk.shape[1];
    std::vector<const char *> ptr((size_t)n);
    std::vector<size_t> len((size_t)n);
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *o = PyList_GET_ITEM(raws, i);
        char *p;
        Py_ssize_t l;
        if (PyBytes_Check(o)) { p = PyBytes_AS_STRING(o); l = PyBytes_GET_SIZE(o); }
        else if (PyByteArray_Check(o)) { p = PyByteArray_AS_STRING(o); l = PyByteArray_GET_SIZE(o); }
        else { PyBuffer_Release(&blk); PyErr_SetString(PyExc_TypeError, "rows must be bytes or bytearray objects"); return nullptr; }
        ptr[(size_t)i] = p;
        len[(size_t)i] = (size_t)l;
    }
    uint8_t *dst = static_cast<uint8_t *>(blk.buf);
    Py_BEGIN_ALLOW_THREADS
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(8, ((size_t)n * rb) >> 22));
    auto work = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            const size_t take = std::min(len[i], rb);
            memcpy(dst + i * rb, ptr[i], take);
            if (take < rb) memset(dst + i * rb + take, 0, rb - take);
            if (rb) dst[i * rb + rb - 1] &= (uint8_t)mask;
        }
    };
    if (T == 1) work(0, (size_t)n);
    else {
        std::vector<std::thread> pool;
        const size_t per = ((size_t)n + T - 1) / T;
        for (unsigned t = 0; t < T; t++) pool.emplace_back(work, std::min((size_t)n, t * per), std::min((size_t)n, (t + 1) * per));
        for (auto &th : pool) th.join();
    }
    Py_END_ALLOW_THREADS
    PyBuffer_Release(&blk);
    Py_RETURN_NONE;
}

// fast_dict(on) -> bool: switch the direct-store route of build_scored on / off (tests, A/B); returns whether it is compiled in and active
PyObject *fast_dict(PyObject *, PyObject *args)
{
    int on = -1, split = -1;
    if (!PyArg_ParseTuple(args, "|pp", &on, &split)) return nullptr;
#if BIGSI_FAST_DICT
    if (on >= 0) g_fast_dict = on != 0;
    if (split >= 0) g_split_dict = split != 0;
    return PyBool_FromLong(g_fast_dict && strncmp(Py_GetVersion(), "3.10.", 5) == 0);
#else
    Py_RETURN_FALSE;
#endif
}

// ascii_str(n) -> (s, address): a new str of n ASCII characters whose body (n bytes at `address`) the caller fills before anything
// reads s -- bigsi_hip_format_results writes the text of a bulk search straight into it (no bytes -> str copy of a few hundred MB)
PyObject *ascii_str(PyObject *, PyObject *args)
{
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "n", &n)) return nullptr;
    if (n < 0) { PyErr_SetString(PyExc_ValueError, "negative length"); return nullptr; }
    PyObject *s = PyUnicode_New(n, 127);
    if (!s) return nullptr;
    return Py_BuildValue("(Nn)", s, (Py_ssize_t)(uintptr_t)PyUnicode_DATA(s));
}

PyMethodDef methods[] = {{"build", build, METH_VARARGS, "result dicts of the sequences [lo, hi) of a streaming search (see _results.cpp)"},
                         {"build_scored", build_scored, METH_VARARGS, "the same for score=True, from K6's records and presence bits"},
                         {"fast_dict", fast_dict, METH_VARARGS, "fast_dict([on[, split]]) -> whether build_scored stores values straight into the copied dict (CPython 3.10); split: copies of a split-table template (default) or of a combined one"},
                         {"pack_rows", pack_rows, METH_VARARGS, "rows (list of bytes) -> uint8[n, rb] block, cut / zero-extended, threaded, without the GIL"},
                         {"ascii_str", ascii_str, METH_VARARGS, "(str of n ASCII characters to be filled, address of its body)"},
                         {nullptr, nullptr, 0, nullptr}};
PyModuleDef module = {PyModuleDef_HEAD_INIT, "_results", "result dicts of BIGSI.search_stream, built in C++", -1, methods, nullptr, nullptr, nullptr, nullptr};

}   // namespace

PyMODINIT_FUNC PyInit__results(void) { return PyModule_Create(&module); }
