"""`BIGSI`: the reference's index object (bigsi/graph/bigsi.py:129-275) over the hip-hbm backend.

`search()` / `lookup()` keep the reference's signatures, return shapes, ordering, rounding and error behaviour, but
the work between the query string and the hit list -- k-merise, dedupe, canonicalise, MurmurHash3, row fetch, AND,
per-sample counts, threshold, compaction -- is ONE device batch (include/bigsi_hip.h: bigsi_hip_batch_run).  The host
only assembles result dicts (names, percentages, optional score)."""
import json
import logging

import numpy as np

from ..bitrow import BitRow, row_bytes_of
from ..bloom import _device_bloom
from ..scoring import Scorer
from ..storage import get_storage
from ..utils import seq_to_kmers
from .index import KmerSignatureIndex
from .metadata import DELETION_SPECIAL_SAMPLE_NAME, SampleMetadata

try:          # the C++ assembly of result dicts (bigsi_amd/_results.cpp, built by bigsi_amd/pyext_build.sh); the loop in _emit is its definition
    from .. import _results
except ImportError:
    _results = None

logger = logging.getLogger(__name__)

DEFAULT_NPROC = 4
MIN_UNIQUE_KMERS_IN_QUERY = 0
DEFAULT_CONFIG = {"storage-engine": "hip-hbm", "storage-config": {"name": "default"}, "k": 31, "m": 25 * 10 ** 6, "h": 3}


def validate_build_params(bloomfilters, samples):
    if len(bloomfilters) != len(samples):
        raise ValueError("There must be the same number of bloomfilters and sample names")


# sequences of a search_batch call up to this many bytes in all go through the C ABI's one-call entry point
ONE_CALL_BYTES = 192 << 10

# host memory one slice of scored hits may take as characters (the presence strings are the bulk of a scored result)
SCORE_SLICE_CHARS = 256 << 20


def scored_rows(rec, bits, boff, lengths, db_size):
    """Host half of a scored search: K6's records + presence bits (QueryBatch.score_hits / score_hits_end) -> one tuple per hit,
    (percent_kmers_found, the 17 fields of Scorer.score in the reference's key order, presence string); `lengths` = k-mer
    positions of every hit's sequence."""
    from ..scoring import score_columns, unpack_presence
    text = unpack_presence(bits, boff)
    starts = (boff[:-1].astype(np.int64) * 8).tolist()
    strings = [text[a:a + n] for a, n in zip(starts, np.asarray(lengths).tolist())]
    return list(zip(rec["percent_kmers_found"].tolist(), zip(*score_columns(rec, db_size)), strings))


def score_hit_rows(batch, off, colours, counts, num_kmers, n_seqs, db_size, slice_chars=None):
    """graph/bigsi.py:232-239 for every hit of a batch: the device extracts each hit's presence bits and runs
    remove_short_ones / tabulate_score / calculate_score on them (K6, QueryBatch.score_hits); the host derives the closed-form
    fields for all hits at once (scoring.score_columns) and expands the bits into the "kmer-presence" strings.  Returns one
    tuple per hit, in hit-list order: (percent_kmers_found, the 17 score fields in the reference's key order, presence string).
    Hits are processed in slices of at most `slice_chars` characters, so a low threshold on a wide index cannot ask for tens of
    GB of host memory at once."""
    slice_chars = slice_chars or SCORE_SLICE_CHARS
    off64 = off.astype(np.int64)
    per_hit = np.repeat(np.asarray(num_kmers[:n_seqs], dtype=np.int64), np.diff(off64[:n_seqs + 1]))
    first, total = int(off64[0]), int(off64[n_seqs])
    ends = np.cumsum((per_hit + 63) // 64 * 64)
    rows, lo = [], 0
    while first + lo < total:
        base = int(ends[lo - 1]) if lo else 0
        hi = max(int(np.searchsorted(ends, base + slice_chars, side="right")), lo + 1)
        part = np.clip(off, first + lo, first + hi).astype(np.uint64)          # the same hit lists, restricted to hits [lo, hi)
        rec, bits, boff = batch.score_hits(part, colours, counts, num_kmers)
        rows.extend(scored_rows(rec, bits, boff, per_hit[lo:hi], db_size))
        lo = hi
    return rows


def native_result_lists(nk, nu, off64, cols, counts, exact, names, scored, db_size, block=4096):
    """Generator over the result lists of the sequences 0 .. len(nu) - 1 of a search, built by the C++ extension (bigsi_amd/_results.cpp)
    from the arrays the C ABI returns: nk / nu uint32 per sequence, off64 int64 hit offsets, cols / counts uint32 per hit (ascending
    colours per sequence), names[c] = sample name or None (deleted: dropped), scored = None or (records, bits, bit_offsets) of K6
    (QueryBatch.score_hits_end / search_many_scored).  Same dicts as BIGSI._emit's Python loop and as search(); the caller deals with
    the queries the reference raises on.  `block` sequences are assembled per call of the extension."""
    from ..scoring import SCORE_KEYS, score_transcendentals
    n, total = len(nu), int(off64[len(nu)])
    keys = ("percent_kmers_found", "num_kmers", "num_kmers_found", "sample_name")
    nu = np.ascontiguousarray(nu, dtype=np.uint32)
    off64 = np.ascontiguousarray(off64, dtype=np.int64)
    cols = np.ascontiguousarray(cols[:total], dtype=np.uint32)
    cnts = np.ascontiguousarray(counts[:total], dtype=np.uint32) if counts is not None and len(counts) >= total else np.zeros(total, np.uint32)
    if scored is not None:
        # K6's records and presence bits go to the extension as they are; only the four fields that need numpy's exp / log10 are
        # computed here, for all hits at once (scoring.score_transcendentals)
        keys = keys + SCORE_KEYS + ("kmer-presence",)
        rec, bits, boff = scored
        rec = np.ascontiguousarray(rec[:total])
        trans = tuple(np.ascontiguousarray(c, dtype=np.float64) for c in score_transcendentals(rec, db_size))
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        boff = np.ascontiguousarray(boff[:total + 1], dtype=np.uint64)
        # (blocks of ~8 k hits: a block's dicts -- ~2 KB each with the presence string -- are then made in memory the consumer has just
        # released; 240 k hits built in one go measured 1.5x slower per dict, all of it page faults)
        lo = 0
        while lo < n:
            hi = min(n, lo + block, max(lo + 1, int(np.searchsorted(off64, off64[lo] + 8192, side="right")) - 1))
            yield from _results.build_scored(nu, off64, cols, cnts, bool(exact), names, keys, rec, bits, boff, trans, lo, hi)
            lo = hi
        return
    for lo in range(0, n, block):
        yield from _results.build(nu, off64, cols, cnts, bool(exact), names, keys, None, None, None, None, lo, min(n, lo + block))


def _fast_dict_selftest():
    """Whether the extension's direct-store route for result dicts may be used in THIS interpreter.  That route (bigsi_amd/_results.cpp:
    copies of a split-table template, values written into the copy's values array) leans on CPython 3.10's private dict layout; it is
    compiled for 3.10 only, checks every dict's table before it touches it -- and is OFF unless this test, run once at import, passes:
    one small batch of scored and plain result lists is built both ways and must be equal in every respect a consumer can see (values,
    key order, types, json text), stay equal under the same mutations (add / delete / pop / update a key, copy, pickle round trip) and
    survive a collection.  Anything unexpected -- an exception included -- leaves the portable PyDict_SetItem route in place."""
    if _results is None or not hasattr(_results, "fast_dict"):
        return False
    import copy
    import gc
    import json
    import pickle
    from ..scoring import HIT_SCORE_DTYPE

    class Name(str):          # (a sample name that is not a plain str: such dicts must stay visible to the collector)
        pass
    try:
        nk = np.array([40, 70, 33], np.uint32)
        off = np.array([0, 3, 3, 8], np.int64)
        cols = np.array([0, 2, 5, 1, 2, 3, 4, 6], np.uint32)
        cnts = np.array([40, 31, 40, 33, 20, 33, 9, 14], np.uint32)
        names = ["s0", "s1", Name("s2"), "s3", None, "s5", "s6"]
        rec = np.zeros(8, HIT_SCORE_DTYPE)
        rec["num_kmers"] = np.repeat(nk, np.diff(off))
        rec["score"], rec["min_score"], rec["max_score"] = np.arange(8) * 1.25 + 3, np.arange(8) * 0.5, np.arange(8) * 2.0 + 7
        rec["percent_kmers_found"] = np.round(100.0 * cnts / rec["num_kmers"], 2)
        rec["max_mismatches"], rec["min_mismatches"], rec["mismatches"] = 5, 1, 3
        boff = np.zeros(9, np.uint64)
        boff[1:] = np.cumsum((rec["num_kmers"].astype(np.int64) + 63) // 64 * 8)
        bits = (np.arange(int(boff[-1])) * 37 % 251).astype(np.uint8)

        def build():
            return (list(native_result_lists(nk, nk, off, cols, cnts, False, names, (rec, bits, boff), 1000)),
                    list(native_result_lists(nk, nk, off, cols, cnts, True, names, None, 1000)))

        def same(a, b):
            if len(a) != len(b) or [len(x) for x in a] != [len(x) for x in b]:
                return False
            for x, y in zip(a, b):
                for d, e in zip(x, y):
                    if type(d) is not dict or type(e) is not dict or d != e or list(d) != list(e) or [type(v) for v in d.values()] != [type(v) for v in e.values()]:
                        return False
                    if json.dumps(d) != json.dumps(e) or dict(d) != e or len(d) != len(e):
                        return False
            return True
        _results.fast_dict(False)
        slow = build()
        if not _results.fast_dict(True):
            return False
        fast = build()
        if not all(same(a, b) for a, b in zip(slow, fast)):
            raise ValueError("the two routes differ")
        for lists in (slow, fast):              # the same mutations on both
            for part in lists:
                for res in part:
                    for i, d in enumerate(res):
                        d["extra"] = [i]
                        d["num_kmers"] += 1
                        del d["sample_name"]
                        d.pop("percent_kmers_found")
                        d.update(zzz=i)
                        d.setdefault("sample_name", "again")
        gc.collect()
        if not all(same(a, b) for a, b in zip(slow, fast)):
            raise ValueError("the two routes differ after mutation")
        if pickle.loads(pickle.dumps(fast)) != slow or copy.deepcopy(fast) != slow:
            raise ValueError("the two routes differ after a round trip")
        del slow, fast
        gc.collect()
        return True
    except Exception as e:  # noqa: BLE001 -- whatever went wrong, the portable route is the answer
        logger.warning("bigsi_amd: the direct-store route for result dicts stays off (%s: %s)", type(e).__name__, e)
        _results.fast_dict(False)
        return False


FAST_DICT_ACTIVE = _fast_dict_selftest()


class _StreamTuning(object):
    """The two interpreter-wide settings a running search_stream changes -- a short thread switch interval, the cyclic collector's
    automatic passes paused -- owned by ALL live streams together: the first stream in saves and sets, the last one out restores,
    under a lock, so that streams in several threads (or nested ones) cannot restore each other's values out of order."""
    lock = __import__("threading").Lock()
    users = 0
    gc_users = 0
    interval = None
    gc_was_on = False

    @classmethod
    def enter(cls, pause_gc):
        import gc
        import sys
        with cls.lock:
            if cls.users == 0:
                cls.interval = sys.getswitchinterval()
                sys.setswitchinterval(min(cls.interval, 2e-4))
            cls.users += 1
            if pause_gc:
                if cls.gc_users == 0:
                    cls.gc_was_on = gc.isenabled()
                    if cls.gc_was_on:
                        gc.disable()
                cls.gc_users += 1
            return bool(pause_gc) and cls.gc_was_on

    @classmethod
    def leave(cls, pause_gc):
        import gc
        import sys
        with cls.lock:
            cls.users -= 1
            if cls.users == 0 and cls.interval is not None:
                sys.setswitchinterval(cls.interval)
            if pause_gc:
                cls.gc_users -= 1
                if cls.gc_users == 0 and cls.gc_was_on:
                    gc.enable()


class BigsiQueryResult(object):
    """One hit; `todict()` key order is part of the contract (tests/graph/test_end_to_end.py:114-124)."""

    def __init__(self, colour, sample_name, num_kmers_found, num_kmers):
        self.colour = colour
        self.sample_name = sample_name
        self.num_kmers_found = num_kmers_found
        self.num_kmers = num_kmers
        self.percent_kmers_found = round(100 * float(num_kmers_found) / num_kmers, 2)
        self.score = None

    def todict(self):
        out = {"percent_kmers_found": self.percent_kmers_found, "num_kmers": self.num_kmers,
               "num_kmers_found": self.num_kmers_found, "sample_name": self.sample_name}
        if self.score:
            out.update(self.score)
        return out

    def tojson(self):
        return json.dumps(self.todict())

    __repr__ = tojson

    def __eq__(self, other):
        return self.todict() == other.todict()

    def add_score(self, score):
        self.score = score


class _Done(object):
    """search_stream: a chunk whose results are already there (it held non-ASCII sequences)."""

    def __init__(self, results):
        self.results = results


class BIGSI(SampleMetadata, KmerSignatureIndex):
    def __init__(self, config=None):
        self.config = DEFAULT_CONFIG if config is None else config
        self.storage = get_storage(self.config)
        SampleMetadata.__init__(self, self.storage)
        KmerSignatureIndex.__init__(self, self.storage)
        self.min_unique_kmers_in_query = MIN_UNIQUE_KMERS_IN_QUERY
        self.scorer = Scorer(self.num_samples)

    @property
    def kmer_size(self):
        return self.config["k"]

    @property
    def nproc(self):
        return self.config.get("nproc", DEFAULT_NPROC)

    # ------------------------------------------------------------------ construction
    @classmethod
    def bloom(cls, config, kmers):
        """Bloom filter (m bits) of the canonical forms of `kmers`, built on the device."""
        device = (config.get("storage-config") or {}).get("device", 0)
        raw = _device_bloom(list(kmers), config["m"], config["h"], raw=False, device=device)
        return BitRow.frombytes(raw.tobytes(), int(config["m"]))

    @classmethod
    def build(cls, config, bloomfilters, samples):
        storage = get_storage(config)
        validate_build_params(bloomfilters, samples)
        SampleMetadata(storage).add_samples(samples)
        KmerSignatureIndex.create(storage, bloomfilters, config["m"], config["h"], config.get("low_mem_build", False))
        storage.close()
        return cls(config)

    @classmethod
    def build_from_sequences(cls, config, samples):
        """Build an index straight from sequences, no Bloom files and no transpose: `samples` maps sample name ->
        list of sequences; every k-mer of every sequence is Bloom-added to that sample's column on the device
        (k_insert_kmers = BIGSI.bloom + build of the reference, graph/bigsi.py:150-172, fused).  Same rows as
        build(config, [bloom(config, kmers of s) for s in samples], names)."""
        storage = get_storage(config)
        names = list(samples.keys())
        SampleMetadata(storage).add_samples(names)
        storage.set_integer("ksi:bloomfilter_size", config["m"])
        storage.set_integer("ksi:num_hashes", config["h"])
        storage.set_integer("number_of_rows", config["m"])
        storage.set_integer("number_of_cols", len(names))
        storage.res.ensure_open()
        storage.res.written[:] = True
        for colour, name in enumerate(names):
            seqs = [samples[name]] if isinstance(samples[name], str) else list(samples[name])
            if seqs:
                storage.insert_kmers(colour, seqs, config["k"])
        storage.sync()
        storage.close()
        return cls(config)

    def insert(self, bloomfilter, sample):
        logger.warning("Build and merge is preferable to insert in most cases")
        colour = self.add_sample(sample)
        self.insert_bloom(bloomfilter, colour - 1)

    def insert_many(self, bloomfilters, samples):
        """Append several samples at once: their filters become the next columns through ONE device transpose
        (bigsi_hip_insert_columns) instead of a column-at-a-time loop.  Same result as insert() called for each in turn
        (graph/bigsi.py:244-247)."""
        validate_build_params(bloomfilters, samples)
        col0 = self.num_samples
        self.add_samples(samples)
        nb = (int(self.bloomfilter_size) + 7) // 8
        arr = np.zeros((len(bloomfilters), nb), dtype=np.uint8)
        for i, bf in enumerate(bloomfilters):
            data = np.frombuffer(row_bytes_of(getattr(bf, "bitarray", bf))[0], dtype=np.uint8)[:nb]
            arr[i, : data.size] = data
        self.storage.insert_columns(col0, arr)
        self.bitmatrix.set_num_cols(max(self.bitmatrix.num_cols, col0 + len(bloomfilters)))

    def delete(self):
        for batch in self.__dict__.pop("_workspaces", {}).values():
            batch.close()
        self.storage.delete_all()

    def merge(self, bigsi):
        assert self.bloomfilter_size == bigsi.bloomfilter_size
        assert self.num_hashes == bigsi.num_hashes
        assert self.kmer_size == bigsi.kmer_size
        self.merge_indexes(bigsi)
        self.merge_metadata(bigsi)

    # ------------------------------------------------------------------ queries
    def seq_to_kmers(self, seq):
        return seq_to_kmers(seq, self.kmer_size)

    def search(self, seq, threshold=1.0, score=False):
        if len(seq) - self.kmer_size + 1 <= self.min_unique_kmers_in_query:
            logger.warning("Query string should contain at least %i unique kmers. Your query contained %i unique kmers, "
                           "and as a result the false discovery rate may be high. In future this will become an error."
                           % (self.min_unique_kmers_in_query, max(len(seq) - self.kmer_size + 1, 0)))
        assert threshold <= 1
        return self.search_batch([seq], threshold, score)[0]

    def _workspace(self, slot, seqs, ws=None):
        """Batch workspace `slot` (of `ws`, else of this index object), staged with `seqs` (created once, then only reloaded).
        search() / search_batch() use the object's slot 0; every stream generator brings its own dict, so that a search()
        made while a stream is being consumed cannot reload a batch the stream still has in flight."""
        if ws is None:
            ws = self.__dict__.setdefault("_workspaces", {})
        batch = ws.get(slot)
        if batch is None or batch.b is None or batch.storage.res is not self.storage.res:
            batch = ws[slot] = self.storage.new_batch(seqs, self.kmer_size)
        else:
            batch.reload(seqs, self.kmer_size)
        return batch

    def _launch(self, batch, threshold):
        # hit lists only: counters of non-hits are never stored; config key `early_exit: true` additionally lets an exact
        # search stop reading a query's rows once no sample can match any more (identical results)
        batch.run(threshold, sparse_counts=True, early_exit=bool(self.config.get("early_exit", False)))

    def _collect(self, batch, n_seqs, threshold, score):
        return self._collect_end(self._collect_begin(batch, n_seqs, threshold, score))

    def _collect_begin(self, batch, n_seqs, threshold, score, deferred=False):
        """First half of a batch's collection: per-query counts and hit lists to the host, the reference's errors for degenerate
        queries, and -- score=True -- the scored hits: at once (K6 through the synchronous call), or `deferred`: only queued
        (bigsi_hip_batch_score_hits_begin) for _collect_end to pick up, so that a stream can launch its next batch in between."""
        num_kmers, num_unique, _ = batch.unique()
        off, colours, counts = batch.hits()
        exact = threshold == 1.0
        if num_unique[:n_seqs].all() and int(off[n_seqs]) == 0:
            return None, n_seqs                           # the common bulk case: nothing found anywhere in the batch
        nu = num_unique[:n_seqs]
        if not nu.all():
            # the reference fails on a query without k-mers: reduce() over nothing on the exact branch
            # (utils/fncts.py:24-25), an unbound accumulator on the other (graph/bigsi.py:35-44)
            if exact:
                raise TypeError("reduce() of empty sequence with no initial value")
            raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
        off64 = off.astype(np.int64)
        scored = None
        if score:
            if ((np.diff(off64[:n_seqs + 1]) > 0) & (num_kmers[:n_seqs] == 1)).any():
                # the reference builds a 1-D matrix from a single row and then indexes it with two subscripts
                raise IndexError("too many indices for array: array is 1-dimensional, but 2 were indexed")
            per_hit = np.repeat(num_kmers[:n_seqs].astype(np.int64), np.diff(off64[:n_seqs + 1]))
            fits = int(((per_hit + 63) // 64 * 64).sum()) <= SCORE_SLICE_CHARS
            if deferred and not batch.group and fits:
                batch.score_hits_begin(off, colours, None if exact else counts, num_kmers)
                scored = ("pending", per_hit)
            elif _results is not None and fits and int(off[0]) == 0:
                # K6's records and presence bits as arrays: _collect_end hands them to the C++ builder (bigsi_amd/_results.cpp:
                # build_scored) instead of a tuple and a 22-key dict per hit made in Python
                scored = ("arrays", batch.score_hits(off, colours, None if exact else counts, num_kmers), num_kmers)
            else:
                scored = self._score_hits(batch, off, colours, None if exact else counts, num_kmers, n_seqs)
        return (batch, off64, colours, counts, nu, exact, scored), n_seqs

    def _collect_end(self, begun):
        state, n_seqs = begun
        if state is None:
            return [[] for _ in range(n_seqs)]
        batch, off64, colours, counts, nu, exact, scored = state
        if isinstance(scored, tuple) and scored[0] == "pending":
            rec, bits, boff = batch.score_hits_end()
            scored = scored_rows(rec, bits, boff, scored[1], self.scorer.DB_SIZE)
        elif isinstance(scored, tuple) and scored[0] == "arrays":
            (rec, bits, boff), num_kmers = scored[1], scored[2]
            total = int(off64[n_seqs])
            names = self._names_of(colours[:total], exact)
            if names is not None:
                return list(native_result_lists(num_kmers[:n_seqs], nu[:n_seqs], off64[:n_seqs + 1], colours, counts, exact, names, (rec, bits, boff), self.scorer.DB_SIZE))
            per_hit = np.repeat(num_kmers[:n_seqs].astype(np.int64), np.diff(off64[:n_seqs + 1]))
            scored = scored_rows(rec, bits, boff, per_hit, self.scorer.DB_SIZE)          # (a colour without a name: the per-record loop raises the reference's KeyError in order)
        if _results is not None and scored is None:
            # unscored: the dicts straight from the arrays (bigsi_amd/_results.cpp), as search_stream builds them
            total = int(off64[n_seqs])
            names = self._names_of(colours[:total], exact)
            if names is not None:
                return list(native_result_lists(None, nu[:n_seqs], off64[:n_seqs + 1], colours, counts, exact, names, None, self.scorer.DB_SIZE))
        out = [[] for _ in range(n_seqs)]
        for i in np.flatnonzero(np.diff(off64[:n_seqs + 1])).tolist():      # only the sequences that have hits
            lo, hi = int(off64[i]), int(off64[i + 1])
            out[i] = self._assemble(lo, colours[lo:hi], counts[lo:hi], int(nu[i]), exact, scored)
        return out

    def _score_hits(self, batch, off, colours, counts, num_kmers, n_seqs):
        return score_hit_rows(batch, off, colours, counts, num_kmers, n_seqs, self.scorer.DB_SIZE)

    def _elements_of(self, seq):
        """A non-ASCII query as the device takes it: its unique k-mers (k CHARACTERS each, utils/fncts.py:63-65) in
        first-occurrence order, canonical (character-wise, fncts.py:38-54) and UTF-8 encoded -- the bytes the reference hashes
        (bloom/bloomfilter.py:5-6) -- plus, for every position, the index of its unique k-mer."""
        from ..utils import canonical
        k, index, uniq, pos = self.kmer_size, {}, [], []
        for i in range(len(seq) - k + 1):
            km = seq[i:i + k]
            j = index.get(km)
            if j is None:
                j = index[km] = len(uniq)
                uniq.append(canonical(km).encode("utf-8"))
            pos.append(j)
        return uniq, pos

    def _search_wide(self, seqs, threshold, score):
        """search_batch for sequences with non-ASCII characters: k-merised here, hashed / fetched / combined on the device."""
        batch = self.storage.new_element_batch([self._elements_of(s) for s in seqs])
        try:
            self._launch(batch, threshold)
            return self._collect(batch, len(seqs), threshold, score)
        finally:
            batch.close()

    def search_batch(self, seqs, threshold=1.0, score=False):
        """search() for many sequences in one device batch; a list of result lists in input order."""
        assert threshold <= 1
        with self._device_lock():          # (a search_stream being consumed has a worker thread on this index between yields)
            return self._search_batch_locked(seqs, threshold, score)

    def _search_batch_locked(self, seqs, threshold, score):
        seqs = list(seqs)
        if not seqs:
            return []
        wide = [i for i, s in enumerate(seqs) if not s.isascii()]
        if wide:
            out = [None] * len(seqs)
            rest = [i for i in range(len(seqs)) if seqs[i].isascii()]
            for i, r in zip(wide, self._search_wide([seqs[i] for i in wide], threshold, score)):
                out[i] = r
            if rest:
                for i, r in zip(rest, self._search_batch_locked([seqs[i] for i in rest], threshold, score)):
                    out[i] = r
            return out
        if not score and sum(map(len, seqs)) <= ONE_CALL_BYTES:
            # unscored (the reference's `search` endpoint, bigsi/__main__.py:195-209): ONE call of the C ABI -- bigsi_hip_search_batch:
            # zero-copy input, results written into pinned memory by the last kernel -- instead of reload + run + two fetches
            # (a 1 kbp query against a 125 GB index: 35 us in the C call against ~90 through the batch object)
            from .. import _lib
            flags = _lib.RUN_EARLY_EXIT if self.config.get("early_exit", False) else 0
            nk, nu, off, colours, counts = self.storage.search_batch_arrays(seqs, self.kmer_size, threshold, flags, borrow=True)
            return self._collect_end(self._check_degenerate(None, len(seqs), threshold, nk, nu, off, colours, counts))
        batch = self._workspace(0, seqs)
        self._launch(batch, threshold)
        return self._collect(batch, len(seqs), threshold, score)

    def _check_degenerate(self, batch, n_seqs, threshold, num_kmers, num_unique, off, colours, counts):
        """The unscored half of _collect_begin on arrays: the reference's errors for queries without k-mers, the no-hit shortcut."""
        exact = threshold == 1.0
        if num_unique[:n_seqs].all() and int(off[n_seqs]) == 0:
            return None, n_seqs
        nu = num_unique[:n_seqs]
        if not nu.all():
            if exact:
                raise TypeError("reduce() of empty sequence with no initial value")
            raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
        return (batch, off.astype(np.int64), colours, counts, nu, exact, None), n_seqs

    def search_stream(self, seqs, threshold=1.0, score=False, batch_size=None, batch_kmers=1 << 19, pause_gc=True):
        """Generator over (sequence, results) for an arbitrarily long iterable of sequences (bulk_search, bigsi/__main__.py:261-314,
        runs one BIGSI.search per query in a fork pool).  The sequences go through the C ABI's streaming entry points --
        bigsi_hip_search_stream, or bigsi_hip_search_stream_scored with score=True: one call per slice of about 8 x `batch_kmers`
        k-mers (or 16 x `batch_size` sequences), inside which the library keeps four device batches in flight and, scored, runs
        K5 + K6 of one beside the row-AND of the next -- on a worker thread (the call holds no GIL), while this thread turns the
        arrays of the slice before into the reference's result dicts.  Results, order and the reference's errors for degenerate
        queries (raised when the offending sequence's turn comes, after everything before it was yielded) are those of search()
        per sequence.  While a stream is being consumed the index handle is in use by the worker between yields: search() /
        search_batch() / lookup() take the same lock and simply wait; do not modify the index.  `pause_gc` (default on): full
        passes of Python's cyclic garbage collector are deferred until the stream ends (see below); pass False to leave it alone.
        Exhaust the generator or .close() it: while it is suspended the process runs with the short switch interval and (pause_gc)
        without automatic collections; several streams at once share the two settings and the last one to end restores them.
        A multi-GPU (devices=[...]) index streams through its batch objects instead (_search_stream_batches)."""
        assert threshold <= 1
        if self.storage.res.is_group:
            yield from self._search_stream_batches(seqs, threshold, score, batch_size, batch_kmers)
            return
        from concurrent.futures import ThreadPoolExecutor
        from ..storage.hip_hbm import TooManyHits
        k, st, lock = self.kmer_size, self.storage, self._device_lock()

        from .. import _lib

        def pack(chunk):
            """(blob, offsets) of an all-ASCII slice -- packed on the CALLER's thread, so that the worker goes from one C call straight
            into the next -- or None: the slice then takes search_batch's route for non-ASCII text."""
            try:
                blob, soff = _lib.pack_seqs(chunk)
            except ValueError:
                return None
            return (blob, soff) if blob.isascii() else None          # (bytes among the sequences may hold anything)

        def work(chunk, packed):
            with lock:
                try:
                    if packed is None:                           # rare: answered through search_batch's non-ASCII route
                        return "done", chunk, self._search_batch_locked(chunk, threshold, score)
                    if not score:
                        return "arrays", chunk, st.search_many_packed(packed[0], packed[1], k, threshold)
                    try:
                        return "arrays", chunk, st.search_many_scored(None, k, threshold, max_bits=SCORE_SLICE_CHARS // 8, packed=packed)
                    except TooManyHits:                          # (a low threshold on a wide index: the batch route scores in slices)
                        return "done", chunk, self._search_batch_locked(chunk, threshold, score)
                except Exception as e:  # noqa: BLE001 -- surfaces in stream order, from emit()
                    return "error", chunk, e

        take = batch_size * 16 if batch_size else 64
        slices = self._slices(iter(seqs), take, take if batch_size else None, batch_kmers * 8, k)
        # the worker needs the GIL a few times per slice (arguments in, arrays out) while this thread assembles dicts in pure Python
        # and never lets go of it voluntarily: at the default 5 ms switch interval those handoffs cost a scored stream a third of its
        # rate (128 -> 165+ M lookups/s on BASELINE configs[4]'s shard)
        import gc
        # The stream makes two containers per sequence (the pair, the result list); every ~35 000 of them the cyclic collector
        # starts a FULL collection, which walks every object of the process -- with torch and numpy imported ~40 ms, GIL held, the
        # worker stuck at the end of its C call: a quarter of a scored stream's time (163 -> 120 ms per 24 576 queries of 1 kbp).
        # While the stream runs, full collections wait: the collector is off, and the two young generations are collected at every
        # slice boundary (cheap: only what was made since), so that cyclic garbage of the consumer does not pile up unseen.
        # Both settings are process-wide: _StreamTuning shares them between all running streams (first in sets, last out restores).
        # They are restored when the generator ends, is closed or is finalised -- a stream that is abandoned half-consumed should be
        # .close()d (a suspended generator kept alive keeps them in force).
        gc_was_on = _StreamTuning.enter(pause_gc)
        with ThreadPoolExecutor(1) as pool:
            pending = None
            try:
                for chunk in slices:
                    nxt = pool.submit(work, chunk, pack(chunk))
                    if pending is not None:
                        yield from self._emit(pending.result(), threshold, score)
                        if gc_was_on:
                            gc.collect(1)
                    pending = nxt
                if pending is not None:
                    last, pending = pending, None
                    yield from self._emit(last.result(), threshold, score)
            finally:
                _StreamTuning.leave(pause_gc)
                if pending is not None:
                    pending.result()                             # (the consumer stopped early: let the worker leave the index alone)

    def _device_lock(self):
        import threading
        return self.storage.res.__dict__.setdefault("_lock", threading.RLock())

    def _names_of(self, cols, exact):
        """names[c] for the C++ assembly of result dicts: the sample name of every colour that occurs in `cols` (looked up once),
        None for a deleted sample (its hits are dropped).  Returns None instead when a colour has no name (beyond num_samples on the
        exact route, or below it with no metadata record): a KeyError in the reference, which the Python loop raises at the right place."""
        ns = self.num_samples
        names = [None] * ns
        for c in np.unique(cols).tolist():
            if c < ns:
                try:
                    name = self.colour_to_sample(c)
                except KeyError:          # a colour below num_samples without a name: the per-record Python loop raises it when its turn comes
                    return None
                names[c] = None if name == DELETION_SPECIAL_SAMPLE_NAME else name
            elif exact:
                return None
        return names

    def _emit_native(self, res, nk, nu, off64, n_hits, colours, counts, threshold, score):
        """_emit's loop in the C++ extension (bigsi_amd/_results.cpp): the same dicts, built from the arrays without a Python
        statement per hit.  What stays here: the reference's errors (raised when the offending sequence's turn comes), the sample
        names of the colours that occur (looked up once per slice), the closed-form score columns (numpy, all hits at once)."""
        _, chunk, payload = res
        exact = threshold == 1.0
        ns = self.num_samples
        # where the reference raises: a query without k-mers (either branch); score=True on a one-k-mer query that has hits
        bad = nu == 0
        if score:
            bad = bad | ((nk == 1) & (n_hits > 0))
        stop = int(np.argmax(bad)) if bad.any() else len(chunk)
        total = int(off64[stop])
        cols = np.ascontiguousarray(colours[:total])
        names = self._names_of(cols, exact)
        if names is None:
            yield from self._emit(res, threshold, score, native=False)
            return
        scored = (payload[7][:total], payload[5], payload[6]) if score else None
        # (blocks of sequences: the consumer gets its first results before the whole slice is assembled)
        yield from zip(chunk[:stop], native_result_lists(nk[:stop], nu[:stop], off64[:stop + 1], cols, counts, exact, names, scored, self.scorer.DB_SIZE))
        if stop < len(chunk):
            if nu[stop] == 0:
                if exact:
                    raise TypeError("reduce() of empty sequence with no initial value")
                raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
            raise IndexError("too many indices for array: array is 1-dimensional, but 2 were indexed")

    def _emit(self, res, threshold, score, native=True):
        """(sequence, results) pairs of one slice from what the worker left: the reference's errors in stream order.  Plain Python
        over lists made once per slice (a numpy call per sequence costs more than the few hits a read has)."""
        from ..scoring import SCORE_KEYS
        kind, chunk, payload = res
        if kind == "error":
            raise payload
        if kind == "done":
            yield from zip(chunk, payload)
            return
        nk, nu, off, colours, counts = payload[:5]
        exact = threshold == 1.0
        off64 = off.astype(np.int64)
        n_hits = np.diff(off64)
        special = np.flatnonzero((n_hits > 0) | (nu == 0)).tolist()
        if not special:
            for s in chunk:
                yield s, []
            return
        if _results is not None and native:
            yield from self._emit_native(res, nk, nu, off64, n_hits, colours, counts, threshold, score)
            return
        scored = None
        if score and int(off64[-1]):
            # the scored rows (closed-form fields, presence strings) of a slice are made 256 hits at a time, as they are needed: 4096
            # hits of 970 positions at once are 4 MB of fresh strings per slice -- page faults and cache misses made that 7 us per hit
            # where blocks that stay in the cache take 4
            bits, boff, rec = payload[5:8]
            lengths, db, blk = np.repeat(nk.astype(np.int64), n_hits), self.scorer.DB_SIZE, {"lo": 0, "rows": []}

            class _Blocks(object):
                def __getitem__(self_, t):
                    lo = blk["lo"]
                    if not lo <= t < lo + len(blk["rows"]):
                        lo = t - t % 256
                        hi = min(lo + 256, len(lengths))
                        b0 = int(boff[lo])
                        blk["rows"] = scored_rows(rec[lo:hi], bits[b0:int(boff[hi])], boff[lo:hi + 1] - boff[lo], lengths[lo:hi], db)
                        blk["lo"] = lo
                    return blk["rows"][t - lo]
            scored = _Blocks()
        offs, nus, nks, cols, cnts = off64.tolist(), nu.tolist(), nk.tolist(), colours.tolist(), counts.tolist()
        ns, name_of, deleted = self.num_samples, self.colour_to_sample, DELETION_SPECIAL_SAMPLE_NAME
        names = {}
        keys = ("percent_kmers_found", "num_kmers", "num_kmers_found", "sample_name") + SCORE_KEYS + ("kmer-presence",)
        prev = 0
        for i in special:
            for s in chunk[prev:i]:
                yield s, []
            prev = i + 1
            u = nus[i]
            if u == 0:
                # the reference fails on a query without k-mers: reduce() over nothing on the exact branch
                # (utils/fncts.py:24-25), an unbound accumulator on the other (graph/bigsi.py:35-44)
                if exact:
                    raise TypeError("reduce() of empty sequence with no initial value")
                raise UnboundLocalError("local variable 'cumsum' referenced before assignment")
            if score and nks[i] == 1:
                # the reference builds a 1-D matrix from a single row and then indexes it with two subscripts
                raise IndexError("too many indices for array: array is 1-dimensional, but 2 were indexed")
            lo, hi = offs[i], offs[i + 1]
            if exact:
                # exact_filter (graph/bigsi.py:192-205): every set bit, ascending; a colour without a name is a KeyError
                order = range(lo, hi)
            else:
                # inexact_filter (:211-230): only colours < num_samples are zipped in; stable sort by count, descending
                order = sorted((t for t in range(lo, hi) if cols[t] < ns), key=lambda t: -cnts[t])
            out = []
            for t in order:
                c = cols[t]
                name = names.get(c)
                if name is None:
                    name = names[c] = name_of(c)
                if name == deleted:
                    continue
                f = u if exact else cnts[t]
                if scored is None:
                    out.append({"percent_kmers_found": round(100 * float(f) / u, 2), "num_kmers": u, "num_kmers_found": f, "sample_name": name})
                else:
                    pct, fields, text = scored[t]
                    out.append(dict(zip(keys, (pct, u, f, name) + fields + (text,))))
            yield chunk[i], out
        for s in chunk[prev:]:
            yield s, []

    def _search_stream_batches(self, seqs, threshold=1.0, score=False, batch_size=None, batch_kmers=1 << 19):
        """search_stream over this object's own batch workspaces (what a multi-GPU index uses), two workspaces deep: while
        the GPU runs batch i+1 the host fetches and assembles batch i (fetches wait on the batch's own completion event, not
        on the stream).  A device batch closes after `batch_size` sequences if given, else once it holds about
        `batch_kmers` k-mers (about 540 x 1 kbp, or ~17000 reads of 61 bp: 4 ms of device work on a 125 GB index, enough to hide the
        per-batch host work; measured 117 M lookups/s with 2^18 and 124 M with 2^19 k-mers per batch at C3)."""
        assert threshold <= 1
        k, ws = self.kmer_size, {}

        def submit(chunk, slot):
            try:
                plain = "".join(chunk).isascii()           # one pass at C speed
            except TypeError:                             # (bytes among the sequences)
                plain = all(s.isascii() for s in chunk)
            if not plain:                                 # rare: answered at once through search_batch's non-ASCII route
                return [_Done(self.search_batch(chunk, threshold, score)), chunk]
            batch = self._workspace(slot, chunk, ws)
            self._launch(batch, threshold)
            return [batch, chunk]

        def begin(p):
            # score=True, three batches deep: the scored hits of the batch launched one step ago are only QUEUED here (K5 + K6 beside
            # the batch just launched); done() picks them up a step later.  The reference's errors for degenerate queries are
            # raised from here: kept for done(), so that they surface in stream order (after the results of the batch before)
            if not isinstance(p[0], _Done):
                try:
                    p.append(self._collect_begin(p[0], len(p[1]), threshold, score, deferred=True))
                except Exception as e:  # noqa: BLE001
                    p.append(e)

        def done(p):
            if isinstance(p[0], _Done):
                return p[0].results
            if len(p) > 2 and isinstance(p[2], Exception):
                raise p[2]
            return self._collect_end(p[2]) if len(p) > 2 else self._collect(p[0], len(p[1]), threshold, score)

        # sequences are taken from the iterable in slices (C speed: millions of reads go through here); without a
        # batch_size the slice length follows the k-mers per sequence seen so far, so that a batch holds ~batch_kmers of them
        it = iter(seqs)
        take = batch_size if batch_size else 64
        try:
            yield from self._stream_loop(it, take, batch_size, batch_kmers, k, submit, done, begin if score else None, 3 if score else 2)
        finally:
            for b_ in ws.values():
                b_.close()

    @staticmethod
    def _slices(it, take, batch_size, batch_kmers, k):
        """Cut an iterable of sequences into lists: `batch_size` sequences each if given, else about `batch_kmers` k-mers -- the slice
        length follows the k-mers per sequence seen so far, and a slice that overshoots (reads first, genomes later) is cut at the
        sequence that reaches the target, the rest held back for the next slice (a plain list: nothing is wrapped around the iterator)."""
        from itertools import islice
        held_back = []

        def pull(n):
            out = held_back[:n]
            del held_back[:n]
            if len(out) < n:
                out += list(islice(it, n - len(out)))
            return out

        while True:
            chunk = pull(take)
            if not chunk:
                return
            if not batch_size:
                held = sum(map(len, chunk)) - (k - 1) * len(chunk)
                while held < batch_kmers:                # top the slice up to a full batch
                    per = max(held // len(chunk), 1)
                    more = pull(max((batch_kmers - held + per - 1) // per, 1))
                    if not more:
                        break
                    chunk += more
                    held = sum(map(len, chunk)) - (k - 1) * len(chunk)
                    held = max(held, len(chunk))
                if held > batch_kmers and len(chunk) > 1:
                    csum = np.cumsum(np.maximum(np.fromiter(map(len, chunk), dtype=np.int64, count=len(chunk)) - (k - 1), 1))
                    cut = int(np.searchsorted(csum, batch_kmers, side="left")) + 1
                    if cut < len(chunk):
                        held_back[:0] = chunk[cut:]
                        chunk = chunk[:cut]
                take = len(chunk)
            yield chunk

    @staticmethod
    def _stream_loop(it, take, batch_size, batch_kmers, k, submit, done, begin=None, depth=2):
        """submit(chunk, slot) launches a batch on workspace `slot` (of `depth`) and returns its handle; `depth` - 1 batches later
        done(handle) yields its results.  With `begin`, begin(handle) is called once the NEXT batch has been launched (the middle
        stage of a pipeline three deep: work queued beside that batch, collected by done() a step later)."""
        flight, slot = [], 0
        for chunk in BIGSI._slices(it, take, batch_size, batch_kmers, k):
            if len(flight) == depth:                     # the workspace about to be reused: its results first
                p = flight.pop(0)
                yield from zip(p[1], done(p))
            nxt = submit(chunk, slot)
            slot = (slot + 1) % depth
            if begin is not None and flight:
                begin(flight[-1])
            flight.append(nxt)
            if len(flight) == depth:
                p = flight.pop(0)
                yield from zip(p[1], done(p))
        if begin is not None and flight:
            begin(flight[-1])
        for p in flight:
            yield from zip(p[1], done(p))

    def lookup(self, kmers, remove_trailing_zeros=True):
        with self._device_lock():
            return KmerSignatureIndex.lookup(self, kmers, remove_trailing_zeros)

    def search_stream_arrays(self, seqs, threshold=1.0, batch_size=1 << 15):
        """The device's own answer, batch by batch, for callers to whom a Python dict per hit is too slow (millions of reads):
        yields (chunk, num_unique, hit_offsets, colours, counts) -- the sequences of the batch, their unique k-mer counts
        (uint32[n]), and its hit lists: sequence i matched colours[hit_offsets[i]:hit_offsets[i+1]] with that many of its
        k-mers (ascending colour; every colour < num_samples, deleted samples included: map names with colour_to_sample).
        Same two-deep pipeline as search_stream; ASCII sequences only (ValueError otherwise).  Not part of the reference's API."""
        assert threshold <= 1
        from itertools import islice
        it, pending, slot, ws = iter(seqs), None, 0, {}

        def finish(p):
            batch, chunk = p
            _, nu, _ = batch.unique()
            off, colours, counts = batch.hits()
            keep = colours < self.num_samples            # (columns beyond the last sample exist only as padding)
            if not keep.all():
                csum = np.concatenate([[0], np.cumsum(keep)])
                off, colours, counts = csum[off.astype(np.int64)].astype(np.uint64), colours[keep], counts[keep]
            return chunk, nu[:len(chunk)].copy(), off, colours, counts

        try:
            while True:
                chunk = list(islice(it, batch_size))
                if not chunk:
                    break
                batch = self._workspace(slot, chunk, ws)
                self._launch(batch, threshold)
                if pending is not None:
                    yield finish(pending)
                pending, slot = (batch, chunk), slot ^ 1
            if pending is not None:
                yield finish(pending)
        finally:
            for b_ in ws.values():
                b_.close()

    def _assemble(self, first_hit, colours, counts, u, exact, scored):
        """Result dicts of one sequence from its slice of the batch's hit lists; `scored` (score=True) = _score_hits' per-hit
        tuples, indexed by position in the hit lists, `first_hit` = position of this slice's first hit."""
        from ..scoring import SCORE_KEYS
        idx = np.arange(len(colours))
        if exact:
            # exact_filter (graph/bigsi.py:192-205): every set bit, ascending; a colour without a name is a KeyError
            order = idx
            found = [u] * len(colours)
        else:
            # inexact_filter (:211-230): only colours < num_samples are zipped in; stable sort by count, descending
            keep = colours < self.num_samples
            colours, counts, idx = colours[keep], counts[keep], idx[keep]
            order = np.argsort(-counts.astype(np.int64), kind="stable")
            colours, idx = colours[order], idx[order]
            found = counts[order].tolist()
        names = [self.colour_to_sample(c) for c in colours.tolist()]
        if scored is None:
            results = [BigsiQueryResult(0, name, f, u) for name, f in zip(names, found)]
            return [r.todict() for r in results if r.sample_name != DELETION_SPECIAL_SAMPLE_NAME]
        # BigsiQueryResult.todict (graph/bigsi.py:105-114) + add_score, written out: percent and scores come from the device
        keys = ("percent_kmers_found", "num_kmers", "num_kmers_found", "sample_name") + SCORE_KEYS + ("kmer-presence",)
        out = []
        for name, f, t in zip(names, found, idx.tolist()):
            if name == DELETION_SPECIAL_SAMPLE_NAME:
                continue
            pct, fields, col = scored[first_hit + t]
            out.append(dict(zip(keys, (pct, u, f, name) + fields + (col,))))
        return out
