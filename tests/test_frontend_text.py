"""The host-side text helpers of the batch front-end (include/bigsi_hip.h "FRONT-END TEXT": bigsi_hip_fasta_pack,
bigsi_hip_format_results) against the Python route they stand in for -- frontend.read_fasta, json.dumps(records, indent=4),
frontend.d_to_csv -- which golden G9 pins to the reference's own output.  No device needed: the functions are host code."""
import json
import os
import tempfile

import numpy as np
import pytest

from bigsi_amd import _lib, frontend

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libbigsi_hip.so not built")

FASTAS = [
    ">r1\nACGT\n>r2\nGGCC\n",
    ">r1 desc\r\nACGT\r\nTTAA\r\n\r\n>r2\r\n  GG CC \t\r\n",
    "junk before the first header\nACGT\n>a\nAC\n\n\nGT\n>empty\n>b\nTT",
    "",
    "\n\n",
    ">only\n",
    ">x\rAC\rGT\r>y\rTT\r",
    "  >indented header\n\x0bACGT\x1c\n>z\n>\nAA\n",
    ">a\n >b\n",                      # a header in a sequence position: two empty records on every route (path, file object, C packer)
    ">a\nACGT\n\t>b\nGG\n",
]


@pytest.mark.parametrize("text", FASTAS)
def test_fasta_pack_gives_read_fastas_sequences(text):
    fn = tempfile.mktemp(suffix=".fa")
    with open(fn, "w", newline="") as f:
        f.write(text)
    try:
        want = [s for _, s in frontend.read_fasta(fn)]
        with open(fn, "rb") as f:
            blob, off = _lib.fasta_pack(f.read())
    finally:
        os.remove(fn)
    raw = blob.tobytes().decode()
    got = [raw[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    assert got == want
    import io
    assert [s for _, s in frontend.read_fasta(io.StringIO(text, newline=""))] == want      # a file object takes the per-line loop: same records


def test_fasta_pack_of_a_text_cut_into_chunks():
    """A text large enough for several chunks / host threads (one per 4 MB): records of one to five lines of random lengths, some
    empty, CRLF and bare CR line ends here and there, bases before the first header, so that chunk boundaries fall inside records,
    between them and next to blank lines -- against read_fasta."""
    rng = np.random.default_rng(9)
    parts = ["ACGTACGT\nTTTT\n"]                       # bases that belong to no record
    size = len(parts[0])
    i = 0
    while size < 13 << 20:
        lines = [">rec%d %s" % (i, "x" * int(rng.integers(0, 30)))]
        for _ in range(int(rng.integers(0, 6))):
            lines.append("".join(rng.choice(list("ACGTN"), size=int(rng.integers(1, 2000)))))
            if rng.random() < 0.1:
                lines.append("" if rng.random() < 0.5 else "   ")
        end = "\r\n" if i % 7 == 0 else "\r" if i % 11 == 0 else "\n"
        text = end.join(lines) + end
        parts.append(text)
        size += len(text)
        i += 1
    data = "".join(parts)
    fn = tempfile.mktemp(suffix=".fa")
    with open(fn, "w", newline="") as f:
        f.write(data)
    try:
        want = [s for _, s in frontend.read_fasta(fn)]
        blob, off = _lib.fasta_pack(data.encode())
    finally:
        os.remove(fn)
    assert len(off) - 1 == len(want) == i
    raw = blob.tobytes().decode()
    assert int(off[-1]) == len(raw) == sum(len(s) for s in want)
    assert all(raw[int(off[j]):int(off[j + 1])] == want[j] for j in range(len(want)))
    # a non-ASCII byte in a late chunk is still seen
    assert _lib.fasta_pack(data.encode() + ">z\nAC\xc3\xa9GT\n".encode("latin-1")) is None


def test_fasta_pack_leaves_non_ascii_text_to_the_caller():
    assert _lib.fasta_pack(">r\nAC\xc3\xa9GT\n".encode("latin-1")) is None


def _python_text(fmt, seqs, threshold, nu, off, col, cnt, names, deleted):
    """the per-record route of frontend.bulk_search over the same arrays"""
    exact = threshold == 1.0
    recs = []
    for i, s in enumerate(seqs):
        ts = [t for t in range(int(off[i]), int(off[i + 1])) if col[t] < len(names)]
        if not exact:
            ts = sorted(ts, key=lambda t: -int(cnt[t]))
        r = []
        for t in ts:
            if deleted[col[t]]:
                continue
            f = int(nu[i]) if exact else int(cnt[t])
            r.append({"percent_kmers_found": round(100 * float(f) / int(nu[i]), 2), "num_kmers": int(nu[i]), "num_kmers_found": f, "sample_name": names[col[t]]})
        recs.append(frontend.search_record(s, threshold, r))
    if fmt == "json":
        return json.dumps(recs, indent=4)
    return "\n".join(frontend.d_to_csv(d, False, False) if d["results"] else "" for d in recs)


@pytest.mark.parametrize("fmt", ["json", "csv"])
@pytest.mark.parametrize("threshold", [1.0, 0.4, 0.0])
@pytest.mark.parametrize("n", [0, 1, 7, 9000])
def test_format_results_is_the_python_routes_text(fmt, threshold, n):
    rng = np.random.default_rng(n + int(threshold * 10))
    alphabet = list("ACGTN") + ['"', "\\", "\t", "\n", "\r", "\x01", "\x1f", "/", " ", "\x7f", "a"]
    seqs = ["".join(rng.choice(alphabet, size=int(rng.integers(0, 40)), p=[0.18] * 5 + [0.1 / 11] * 11)) for _ in range(n)]
    n_names = 37
    names = ["s%d" % c if c % 5 else 'na"me\\%d,\t' % c for c in range(n_names)]
    deleted = np.zeros(n_names, np.uint8)
    deleted[[3, 20]] = 1
    nu = rng.integers(1, 3000, size=n).astype(np.uint32)
    n_hits = np.where(rng.random(n) < 0.7, 0, rng.integers(1, 12, size=n)) if n else np.zeros(0, np.int64)
    off = np.zeros(n + 1, np.uint64)
    np.cumsum(n_hits, out=off[1:])
    total = int(off[-1])
    col = np.zeros(total, np.uint32)
    cnt = np.zeros(total, np.uint32)
    for i in range(n):
        lo, hi = int(off[i]), int(off[i + 1])
        # ascending colours as the device leaves them; some beyond the named samples (pad columns of a threshold-0 search)
        col[lo:hi] = np.sort(rng.choice(n_names + (0 if threshold == 1.0 else 6), size=hi - lo, replace=False))
        cnt[lo:hi] = rng.integers(0, int(nu[i]) + 1, size=hi - lo) // (1 if rng.random() < 0.5 else 7) * (1 if rng.random() < 0.5 else 7) % (int(nu[i]) + 1)
    blob, soff = _lib.pack_seqs(seqs) if n else (b"", np.zeros(1, np.uint64))
    enc = [nm.encode() for nm in names]
    name_off = np.zeros(n_names + 1, np.uint64)
    np.cumsum([len(e) for e in enc], out=name_off[1:])
    for threads in (1, 0):
        got = _lib.format_results(1 if fmt == "csv" else 0, blob, soff, threshold, json.dumps(frontend.CITATION), nu, off, col, cnt, b"".join(enc) + b"\0",
                                  name_off, deleted, threads=threads)
        assert got == _python_text(fmt, seqs, threshold, nu, off, col, cnt, names, deleted)


def test_format_results_percentages_are_pythons_round_and_repr():
    """every (found, num_kmers) pair up to 400 k-mers plus the half-way cases of larger queries, one hit per record"""
    pairs = [(f, u) for u in range(1, 401) for f in range(0, u + 1)]
    pairs += [(f, u) for u in (800, 970, 1600, 4000, 8000, 65535) for f in range(0, u + 1, max(1, u // 997))]
    n = len(pairs)
    nu = np.array([u for _, u in pairs], np.uint32)
    cnt = np.array([f for f, _ in pairs], np.uint32)
    col = np.zeros(n, np.uint32)
    off = np.arange(n + 1, dtype=np.uint64)
    seqs = ["A"] * n
    blob, soff = _lib.pack_seqs(seqs)
    text = _lib.format_results(1, blob, soff, 0.5, json.dumps(frontend.CITATION), nu, off, col, cnt, b"s\0", np.array([0, 1], np.uint64), np.zeros(1, np.uint8))
    rows = text.split("\n")
    assert len(rows) == n
    for (f, u), row in zip(pairs, rows):
        assert row == '"A",%d,%d,%r,"s"\r' % (u, f, round(100 * float(f) / u, 2)), (f, u)


def test_format_results_leaves_the_references_errors_to_the_caller():
    blob, soff = _lib.pack_seqs(["ACGT", "AC"])
    nu, off = np.array([2, 0], np.uint32), np.zeros(3, np.uint64)
    empty = np.zeros(0, np.uint32)
    with pytest.raises(_lib.BigsiHipError) as e:
        _lib.format_results(0, blob, soff, 1.0, "\"c\"", nu, off, empty, empty, b"\0", np.zeros(1, np.uint64), np.zeros(1, np.uint8))
    assert e.value.code == _lib.ERR_STATE
    # an exact hit on a colour without a name: KeyError in the reference
    nu, off = np.array([2, 2], np.uint32), np.array([0, 1, 1], np.uint64)
    with pytest.raises(_lib.BigsiHipError) as e:
        _lib.format_results(0, blob, soff, 1.0, "\"c\"", nu, off, np.array([5], np.uint32), np.array([2], np.uint32), b"ab\0", np.array([0, 1, 2], np.uint64),
                            np.zeros(2, np.uint8))
    assert e.value.code == _lib.ERR_STATE


def test_format_results_into_the_callers_buffer():
    """*out_text != NULL on entry: the text goes into the caller's buffer.  A zero-byte buffer is the sizing call (BIGSI_ERR_CAPACITY
    and the size), a buffer one byte short is refused the same way, the exact size and a larger one hold the same characters as the
    malloc'ed route; _lib.format_results (which fills the body of the str it returns when the extension is built) gives them too."""
    import ctypes as C
    seqs = ["ACGT" * 9, 'A"C\\G', "TTTTT"]
    blob, soff = _lib.pack_seqs(seqs)
    nu, off = np.array([6, 2, 1], np.uint32), np.array([0, 2, 2, 3], np.uint64)
    col, cnt = np.array([0, 1, 1], np.uint32), np.array([6, 3, 1], np.uint32)
    names, name_off, deleted = b"s0s1\0", np.array([0, 2, 4], np.uint64), np.zeros(2, np.uint8)
    L = _lib.lib()
    for fmt in (0, 1):
        args = (fmt, blob, _lib.ptr(soff), 3, b"0.5", b'"c"', 0, _lib.ptr(nu), _lib.ptr(off), _lib.ptr(col), _lib.ptr(cnt), names, _lib.ptr(name_off),
                _lib.ptr(deleted), 2, 1)
        text, size = C.c_void_p(), C.c_uint64(0)
        _lib.check(L.bigsi_hip_format_results(*args, C.byref(text), C.byref(size)))
        want = C.string_at(text.value, size.value)
        L.bigsi_hip_free_text(text)
        assert len(want) > 20
        for cap in (0, len(want) - 1):
            buf = C.create_string_buffer(max(cap, 1))
            text, size = C.c_void_p(C.addressof(buf)), C.c_uint64(cap)
            assert L.bigsi_hip_format_results(*args, C.byref(text), C.byref(size)) == _lib.ERR_CAPACITY and size.value == len(want)
        for cap in (len(want), len(want) + 5):
            buf = C.create_string_buffer(b"\xff" * (cap + 1), cap + 1)
            text, size = C.c_void_p(C.addressof(buf)), C.c_uint64(cap)
            _lib.check(L.bigsi_hip_format_results(*args, C.byref(text), C.byref(size)))
            assert size.value == len(want) and buf.raw[:len(want)] == want and text.value == C.addressof(buf)
            assert buf.raw[cap:cap + 1] == b"\xff"                       # nothing written beyond the buffer
        got = _lib.format_results(fmt, blob, soff, 0.5, '"c"', nu, off, col, cnt, names, name_off, deleted, threads=1)
        assert got == want.decode()


def _scored_inputs(rng, n, threshold):
    """synthetic arrays of a score=True search: hit lists, K6 records, presence bits, the closed-form columns"""
    from bigsi_amd.scoring import HIT_SCORE_DTYPE, score_columns
    nk = rng.integers(2, 150, size=n).astype(np.uint32)
    nu = np.minimum(nk, rng.integers(1, 150, size=n)).astype(np.uint32)
    n_hits = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 6, size=n)) if n else np.zeros(0, np.int64)
    off = np.zeros(n + 1, np.uint64)
    np.cumsum(n_hits, out=off[1:])
    total = int(off[-1])
    col, cnt = np.zeros(total, np.uint32), np.zeros(total, np.uint32)
    for i in range(n):
        lo, hi = int(off[i]), int(off[i + 1])
        col[lo:hi] = np.sort(rng.choice(20, size=hi - lo, replace=False))
        cnt[lo:hi] = rng.integers(0, int(nu[i]) + 1, size=hi - lo) if threshold < 1 else nu[i]
    per_hit = np.repeat(nk.astype(np.int64), n_hits)
    boff = np.zeros(total + 1, np.uint64)
    np.cumsum((per_hit + 63) // 64 * 8, out=boff[1:])
    bits = rng.integers(0, 256, size=int(boff[-1]), dtype=np.uint8)
    rec = np.zeros(total, HIT_SCORE_DTYPE)
    rec["num_kmers"] = per_hit
    for f in ("score", "min_score", "max_score"):
        rec[f] = np.round(rng.random(total) * 300 - 40, 2)
    rec["percent_kmers_found"] = np.round(rng.random(total) * 100, 2)
    for f in ("max_mismatches", "min_mismatches", "mismatches"):
        rec[f] = rng.integers(0, 25, size=total)
    cols = score_columns(rec, 20, as_arrays=True)
    return nk, nu, off, col, cnt, bits, boff, rec, cols


@pytest.mark.parametrize("fmt", ["json", "csv"])
@pytest.mark.parametrize("threshold", [1.0, 0.4])
@pytest.mark.parametrize("n", [0, 1, 6, 5000])
def test_scored_text_is_the_python_routes_text(fmt, threshold, n):
    """bigsi_hip_format_results_scored against json.dumps / d_to_csv of the dicts the Python route builds from the same arrays:
    the 22 keys in the reference's order (CSV: sorted), floats as repr() writes them, presence strings from the bits."""
    from bigsi_amd.scoring import SCORE_KEYS, unpack_presence
    rng = np.random.default_rng(3 * n + int(10 * threshold))
    alphabet = list("ACGT")
    seqs = ["".join(rng.choice(alphabet, size=int(rng.integers(1, 30)))) for _ in range(n)]
    nk, nu, off, col, cnt, bits, boff, rec, cols = _scored_inputs(rng, n, threshold)
    n_names = 17                                        # colours 17..19 have no name (dropped on the thresholded route)
    if threshold == 1.0:
        col[:] = col % n_names
        for i in range(n):                              # (keep each sequence's colours distinct and ascending)
            lo, hi = int(off[i]), int(off[i + 1])
            col[lo:hi] = np.sort(rng.choice(n_names, size=hi - lo, replace=False))
    names = ["s%d" % c if c % 4 else 'n"%d,' % c for c in range(n_names)]
    deleted = np.zeros(n_names, np.uint8)
    deleted[2] = 1
    text_all = unpack_presence(bits, boff) if len(boff) > 1 else ""
    lists = [c.tolist() for c in cols]
    recs = []
    for i, s in enumerate(seqs):
        ts = [t for t in range(int(off[i]), int(off[i + 1])) if col[t] < n_names]
        if threshold < 1:
            ts = sorted(ts, key=lambda t: -int(cnt[t]))
        r = []
        for t in ts:
            if deleted[col[t]]:
                continue
            f = int(nu[i]) if threshold == 1.0 else int(cnt[t])
            d = {"percent_kmers_found": float(rec["percent_kmers_found"][t]), "num_kmers": int(nu[i]), "num_kmers_found": f, "sample_name": names[col[t]]}
            d.update({k_: lists[j][t] for j, k_ in enumerate(SCORE_KEYS)})
            d["kmer-presence"] = text_all[8 * int(boff[t]): 8 * int(boff[t]) + int(rec["num_kmers"][t])]
            r.append(d)
        recs.append(frontend.search_record(s, threshold, r))
    want = json.dumps(recs, indent=4) if fmt == "json" else "\n".join(frontend.d_to_csv(d, False, False) if d["results"] else "" for d in recs)
    blob, soff = _lib.pack_seqs(seqs) if n else (b"", np.zeros(1, np.uint64))
    enc = [nm.encode() for nm in names]
    name_off = np.zeros(n_names + 1, np.uint64)
    np.cumsum([len(e) for e in enc], out=name_off[1:])
    scored = (rec, bits if len(bits) else np.zeros(8, np.uint8), boff, cols[13], cols[14], cols[15], cols[16], 31)
    for threads in (1, 0):
        got = _lib.format_results(1 if fmt == "csv" else 0, blob, soff, threshold, json.dumps(frontend.CITATION), nu, off, col, cnt, b"".join(enc) + b"\0",
                                  name_off, deleted, threads=threads, scored=scored)
        assert got == want


def test_scored_text_writes_floats_as_pythons_repr():
    """repr() of 60 000 doubles -- random bit patterns, every power of ten from 1e-30 to 1e30 and its neighbours, the 1e16 / 1e-4
    switches to exponent form, integers, two-decimal values, subnormals, zeros, infinities, NaN -- through the evalue column of one
    scored hit per record (CSV: inf / nan; JSON: Infinity / NaN as json.dumps writes them)."""
    from bigsi_amd.scoring import HIT_SCORE_DTYPE
    rng = np.random.default_rng(8)
    xs = rng.integers(0, 2 ** 63, size=40000, dtype=np.uint64).view(np.float64).tolist()
    xs += (rng.random(6000) * 10.0 ** rng.integers(-20, 20, size=6000)).tolist()
    xs += np.round(rng.random(4000) * 1000 - 500, 2).tolist()
    for e in range(-30, 31):
        p = float("1e%d" % e)
        xs += [p, np.nextafter(p, 0), np.nextafter(p, np.inf), -p, 3 * p, p / 3]
    xs += [0.0001, 0.00001, 0.00012345, 9999999999999998.0, 1e16, 1.2345e16, 123456789012345678.0, 1.0, -1.0, 100.0, 0.1, 0.2, 0.3, 1 / 3, 2 / 3, 5e-324,
           2.2250738585072014e-308, 1.7976931348623157e308, 0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e22, 1e23, 4.35, 0.285, 2.675]
    xs = [float(x) for x in xs if x == x and abs(x) != float("inf")] + [float("inf"), float("-inf"), float("nan")]
    n = len(xs)
    rec = np.zeros(n, HIT_SCORE_DTYPE)
    rec["num_kmers"] = 8
    rec["score"] = 1.0
    off = np.arange(n + 1, dtype=np.uint64)
    col, cnt, nu = np.zeros(n, np.uint32), np.full(n, 8, np.uint32), np.full(n, 8, np.uint32)
    bits, boff = np.zeros(8 * n, np.uint8), np.arange(n + 1, dtype=np.uint64) * 8
    blob, soff = _lib.pack_seqs(["A"] * n)
    ev = np.array(xs, np.float64)
    z = np.zeros(n, np.float64)
    scored = (rec, bits, boff, ev, z, z, z, 31)
    text = _lib.format_results(1, blob, soff, 1.0, '"c"', nu, off, col, cnt, b"s\0", np.array([0, 1], np.uint64), np.zeros(1, np.uint8), scored=scored)
    rows = text.split("\n")
    assert len(rows) == n
    for x, row in zip(xs, rows):
        assert row.split(",")[1] == repr(x), (x.hex() if x == x else x, row.split(",")[1])
    text = _lib.format_results(0, blob[-3:], soff[:4], 1.0, '"c"', nu[-3:], off[:4], col[-3:], cnt[-3:], b"s\0", np.array([0, 1], np.uint64), np.zeros(1, np.uint8),
                               scored=(rec[-3:], bits[-24:], boff[:4], ev[-3:], z[-3:], z[-3:], z[-3:], 31))
    assert [r_["results"][0]["evalue"] for r_ in json.loads(text)][:2] == [float("inf"), float("-inf")] and '"evalue": NaN' in text
