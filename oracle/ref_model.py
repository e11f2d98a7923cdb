"""Python-level model of the reference's BIGSI object over a numpy row table -- TEST INFRASTRUCTURE ONLY.

Heavy lifting (hashing, canonical k-mers, row AND, unpack-and-sum) is done by the C oracle
(oracle/bigsi_oracle.c); this file restates the orchestration and result assembly around it so the
HIP path's `search()` dicts can be checked end to end.  Citations are into /root/reference.
Pinned against golden vectors of the real reference by tests/test_oracle_golden.py.
"""
import math
import re

import numpy as np

from . import coracle

DELETED = "D3L3T3D"          # graph/metadata.py:1


def seq_to_kmers(seq, k):    # utils/fncts.py:63-65
    return [seq[i:i + k] for i in range(len(seq) - k + 1)]


def bytes_to_01(row_bytes, nbits=None):
    bits = np.unpackbits(np.frombuffer(bytes(row_bytes), dtype=np.uint8))   # MSB first == bitarray order
    if nbits is not None:
        bits = bits[:nbits]
    return "".join("1" if b else "0" for b in bits)


# ------------------------------------------------------------------ scoring
def remove_short_ones(s):    # scoring/score.py:7-16 -- 3-wide erosion, virtual 1-padding on the right
    n = len(s)
    if n < 3:
        return s
    pad = s + "11"
    return "".join("1" if pad[i] == "1" and pad[i + 1] == "1" and pad[i + 2] == "1" else "0" for i in range(n))


def tabulate_score(ss):      # scoring/score.py:19-32 -- run lengths, every run but the last counted +1
    runs = [(m.group(0)[0], len(m.group(0))) for m in re.finditer(r"0+|1+", ss)]
    out = {"0": [], "1": []}
    for i, (sym, ln) in enumerate(runs):
        # the counter restarts at 1 after each change and is bumped before every comparison, so an
        # interior run of length L is recorded as L+1 and the final run as L
        out[sym].append(ln if i == len(runs) - 1 else ln + 1)
    return out


class Scorer:                # scoring/score.py:35-151
    LAMBDA, K = 1.330, 0.621

    def __init__(self, db_size):
        self.db_size = db_size

    def _calc(self, sc, convert):        # score.py:55-94
        hi = lo = mean = 1 * sum(sc["1"])
        snp_t = 31 + 3
        max_tot = min_tot = 0
        for i in sc["0"]:
            mn = float(i) / snp_t
            mx = (i - snp_t) + 1
            if mx < mn:
                mx = mn
            max_tot += mx
            min_tot += mn
            mid = mn + 0.05 * mx
            hi = round(hi - 2 * mn + 1 * (i - 2 * mn), 2)
            lo = round(lo - 2 * mx + 1 * (i - 2 * mx), 2)
            mean = round(mean - 2 * mid + 1 * (i - 2 * mid), 2)
        return {
            "score": round(mean * convert, 2),
            "min_score": round(lo * convert, 2),
            "max_score": round(hi * convert, 2),
            "max_mismatches": math.ceil(max_tot),
            "min_mismatches": math.floor(min_tot),
            "mismatches": math.ceil(math.ceil(min_tot) + (0.05 * math.floor(max_tot))),
        }

    def score(self, s):                  # score.py:96-116
        ss = remove_short_ones(s)
        n = len(ss)
        seq_len = n + 31 - 1
        d = self._calc(tabulate_score(ss), seq_len / n)
        d["max_nident"] = seq_len - d["min_mismatches"]
        d["nident"] = seq_len - d["mismatches"]
        d["min_nident"] = seq_len - d["max_mismatches"]
        d["pident"] = 100 * float(d["nident"]) / seq_len
        d["max_pident"] = 100 * float(d["max_nident"]) / seq_len
        d["min_pident"] = 100 * float(d["min_nident"]) / seq_len
        d["length"] = seq_len
        d["evalue"] = self.K * self.db_size * seq_len * np.exp(-self.LAMBDA * d["score"])   # score.py:125-129
        d["pvalue"] = 1 - np.exp(-d["evalue"])                                              # score.py:131-132
        m = self.db_size if self.db_size != 0 else 1
        le = round(np.log10(self.K * m * seq_len) - self.LAMBDA * d["score"], 2)            # score.py:134-140 (numpy round)
        d["log_evalue"] = round(le, 2)
        ev = 10 ** d["log_evalue"]                                                          # score.py:142-151
        with np.errstate(divide="ignore"):
            logp = np.log10(1 - np.exp(-ev)) if 1 - np.exp(-ev) > 0 else -np.inf
        lp = round(d["log_evalue"], 2) if logp == -np.inf else round(logp, 2)
        d["log_pvalue"] = round(lp, 2)
        return d


# ---------------------------------------------------------------- non-ASCII text
# The reference works on Python str: k CHARACTERS per k-mer, reverse_comp / canonical character by character
# (utils/fncts.py:38-54), mmh3.hash of the str = of its UTF-8 bytes (bloom/bloomfilter.py:5-6).  bigsi_oracle.c works on bytes,
# which is the same thing for ASCII; these restate the character-wise forms for everything else (pinned by golden G13).
_COMPLEMENT = str.maketrans("ACGT", "TGCA")


def canonical_chars(km):             # fncts.py:38-39, 51-54
    rc = km[::-1].translate(_COMPLEMENT)
    return km if km <= rc else rc


def kmer_rows_chars(km, h, m):       # bloomfilter.py:5-6 on the canonical k-mer
    c = canonical_chars(km)
    return [coracle.row_of(c, seed, m) for seed in range(h)]


def _rows_of(km, h, m):
    return coracle.kmer_rows(km, h, m) if km.isascii() else kmer_rows_chars(km, h, m)


# ---------------------------------------------------------------- the index
class OracleBIGSI:
    """rows: uint8[m, rb] in the reference's storage format; names[c] = sample name of colour c."""

    def __init__(self, rows, names, k, h):
        self.rows = np.ascontiguousarray(rows, dtype=np.uint8)
        self.m, self.rb = self.rows.shape
        self.names = list(names)
        self.k, self.h = k, h
        self.scorer = Scorer(len(self.names))     # graph/bigsi.py:140

    # BIGSI.bloom (graph/bigsi.py:150-155) with the filter zero-initialised (harness patch H1)
    @staticmethod
    def bloom(kmers, m, h):
        bits = np.zeros(m, dtype=np.uint8)
        for km in kmers:
            for r in _rows_of(km, h, m):
                bits[r] = 1
        return np.packbits(bits)       # == bitarray.tobytes(): MSB first, zero padded

    # BIGSI.build (graph/bigsi.py:157-172) + transpose (matrix/transpose.py:33-43)
    @classmethod
    def build(cls, blooms, names, k, m, h):
        cols = np.stack([np.unpackbits(np.asarray(b, dtype=np.uint8))[:m] for b in blooms], axis=1)   # m x N
        return cls(np.packbits(cols, axis=1), names, k, h)

    @property
    def num_samples(self):
        return len(self.names)

    def lookup(self, kmers, remove_trailing_zeros=True):     # graph/index.py:42-49
        if isinstance(kmers, str):
            kmers = [kmers]
        uniq = list(dict.fromkeys(kmers))
        if not uniq:
            return {}
        out = self._per_kmer(uniq, len(uniq[0]))
        nb = self.num_samples if remove_trailing_zeros else None
        return {km: bytes_to_01(out[i].tobytes(), nb) for i, km in enumerate(uniq)}

    def _per_kmer(self, uniq, k):
        """uint8[u, rb]: the AND of each k-mer's h rows (graph/index.py:42-49 + bitvector_index.py:36-41)."""
        if all(km.isascii() for km in uniq):
            return coracle.lookup(self.rows, self.h, uniq, k)
        return np.stack([np.bitwise_and.reduce(self.rows[_rows_of(km, self.h, self.m)], axis=0) for km in uniq])

    def counts(self, seq):
        """(u, int32[num_samples]) of graph/bigsi.py:212-215 for one query."""
        u, cnt, _ = coracle.query(self.rows, self.h, seq, self.k, want_counts=True, want_and=False)
        return u, cnt[: self.num_samples]

    def search(self, seq, threshold=1.0, score=False):       # graph/bigsi.py:174-190
        assert threshold <= 1
        kmers = seq_to_kmers(seq, self.k)
        uniq = list(dict.fromkeys(kmers))
        u = len(uniq)
        min_kmers = math.ceil(u * threshold)
        per_kmer = self._per_kmer(uniq, self.k) if u else np.zeros((0, self.rb), np.uint8)
        if threshold == 1.0:                                  # exact_filter, graph/bigsi.py:192-205
            if u == 0:
                raise TypeError("reduce() of empty sequence with no initial value")
            bits = np.unpackbits(coracle.and_all(per_kmer))
            hits = [(int(c), u) for c in np.nonzero(bits)[0]]
        else:                                                 # inexact_filter, graph/bigsi.py:211-230
            if u == 0:
                raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
            cnt = coracle.unpack_and_sum(per_kmer)[: self.num_samples]
            hits = [(c, int(v)) for c, v in enumerate(cnt) if v >= min_kmers]
            hits.sort(key=lambda x: -x[1])                    # stable: count desc, colour asc
        results = []
        for c, found in hits:
            d = {"percent_kmers_found": round(100 * float(found) / u, 2), "num_kmers": u,
                 "num_kmers_found": found, "sample_name": self.names[c]}
            results.append((c, d))
        if score and results:                                 # graph/bigsi.py:232-239
            if len(kmers) == 1:
                raise IndexError("too many indices for array")
            idx = {km: i for i, km in enumerate(uniq)}
            bits = np.unpackbits(per_kmer, axis=1)            # u x 8rb
            for c, d in results:
                col = "".join("1" if bits[idx[km], c] else "0" for km in kmers)
                sd = self.scorer.score(col)
                sd["kmer-presence"] = col
                d.update(sd)
        return [d for _, d in results if d["sample_name"] != DELETED]


# ---------------------------------------------------- synthetic-index oracle
class SynthOracle:
    """Recomputes rows of the seeded synthetic index (oracle/bigsi_oracle.c: orc_synth_row) plus the
    bits planted by insert_kmers, for parity checks at sizes no host table could hold."""

    def __init__(self, seed, shard, m, n_cols, h, k, and_draws):
        self.seed, self.shard, self.m, self.n_cols, self.h, self.k, self.and_draws = seed, shard, m, n_cols, h, k, and_draws
        self.rb = (n_cols + 7) // 8
        self.planted = {}           # row -> set(cols)

    def insert_kmers(self, colour, seq):
        """Bloom-add every k-mer of seq to sample `colour` (bloom/bloomfilter.py:25-32 on the transposed matrix)."""
        for km in seq_to_kmers(seq, self.k):
            for r in coracle.kmer_rows(km, self.h, self.m):
                self.planted.setdefault(r, set()).add(colour)

    def insert_kmer_masks(self, rows_of_kmers, cols, masks, only_rows=None):
        """The same for many samples at once: k-mer i of a sequence (rows_of_kmers[i] = its h rows, from kmer_rows) is added to
        sample cols[j] where masks[j][i]; only_rows (a uint64 array) keeps the bookkeeping to the rows a later check will read."""
        masks = np.asarray(masks, dtype=bool)
        cols = np.asarray(cols)
        rows_of_kmers = np.asarray(rows_of_kmers, dtype=np.uint64)
        if only_rows is not None:
            keep = np.isin(rows_of_kmers, only_rows)
            todo = np.flatnonzero(keep.any(axis=1))
        else:
            keep, todo = None, range(len(rows_of_kmers))
        for i in todo:
            sel = cols[masks[:, i]].tolist()
            if not sel:
                continue
            for s_, r in enumerate(rows_of_kmers[i].tolist()):
                if keep is None or keep[i, s_]:
                    self.planted.setdefault(r, set()).update(sel)

    def rows_of(self, seq):
        """uint64[n, h]: the h rows of every k-mer position of seq (duplicates included)."""
        return coracle.seq_rows(seq, self.k, self.h, self.m)

    def row(self, r):
        out = coracle.synth_row(self.seed, self.shard, r, self.n_cols, self.and_draws)
        cols = self.planted.get(r)
        if cols:
            c = np.fromiter(cols, dtype=np.int64, count=len(cols))
            np.bitwise_or.at(out, c >> 3, (0x80 >> (c & 7)).astype(np.uint8))
        return out

    def per_kmer_rows(self, seq):
        kmers = seq_to_kmers(seq, self.k)
        uniq = list(dict.fromkeys(kmers))
        out = np.empty((len(uniq), self.rb), dtype=np.uint8)
        for j, km in enumerate(uniq):
            rows = coracle.kmer_rows(km, self.h, self.m)
            acc = self.row(rows[0])
            for r in rows[1:]:
                acc &= self.row(r)
            out[j] = acc
        return kmers, uniq, out

    def counts(self, seq):
        _, uniq, rows = self.per_kmer_rows(seq)
        if not len(uniq):
            return 0, np.zeros(self.n_cols, np.int32)
        return len(uniq), coracle.unpack_and_sum(rows)[: self.n_cols]

    def exact_bitmap(self, seq):
        _, uniq, rows = self.per_kmer_rows(seq)
        return len(uniq), coracle.and_all(rows)
