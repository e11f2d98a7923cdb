"""Batch front-end: the text the reference's `search` / `bulk_search` commands produce (bigsi/__main__.py:41-72, 195-209,
261-314), over ONE device batch per call instead of a fork pool of per-sequence searches.

Pinned byte for byte (JSON indent, key order, csv quoting, the '\\r' the reference leaves when it strips a trailing
newline, the header/terminator state machine of the streaming branch) by tests/golden/g9_frontend.json, which was
produced by running the reference's own `bigsi.__main__`."""
import csv
import io
import json
import sys

CITATION = "http://dx.doi.org/10.1038/s41587-018-0010-1"      # __main__.py:71


def read_fasta(path_or_file):
    """[(name, sequence)] in file order; sequence lines concatenated, no case folding (pyfasta's str(record))."""
    opened = f = open(path_or_file) if isinstance(path_or_file, str) else path_or_file
    try:
        if isinstance(path_or_file, str):
            # the common layout of read sets -- every record a header line and ONE sequence line -- without a Python loop per line
            lines = f.read().split("\n")
            while lines and not lines[-1].strip():
                lines.pop()
            heads, seqs = lines[0::2], lines[1::2]
            # (tested on the stripped lines, as the loop below and bigsi_hip_fasta_pack do: "  >b" is a header there)
            if len(lines) % 2 == 0 and all(h.lstrip().startswith(">") for h in heads) and not any(s_.lstrip().startswith(">") or not s_.strip() for s_ in seqs):
                return [(h.strip()[1:], s_.strip()) for h, s_ in zip(heads, seqs)]
            f = iter(lines)
        recs, name, buf = [], None, []
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(buf)))
                name, buf = line[1:], []
            elif line and name is not None:
                buf.append(line)
        if name is not None:
            recs.append((name, "".join(buf)))
        return recs
    finally:
        if isinstance(path_or_file, str):
            opened.close()


def search_record(seq, threshold, results):
    return {"query": seq, "threshold": threshold, "results": results, "citation": CITATION}      # __main__.py:66-72


def d_to_csv(d, with_header=True, carriage_return=True):
    """One csv row per result: query, then the result's values in sorted-key order; QUOTE_NONNUMERIC, '\\r\\n' line ends;
    carriage_return=False drops the final character only (the '\\n'), as the reference does (__main__.py:41-63)."""
    results = d["results"]
    out = io.StringIO()
    w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
    if results:
        header = sorted(results[0].keys())
        if with_header:
            w.writerow(["query"] + header)
        for res in results:
            w.writerow([d["query"]] + [res[k] for k in header])
    text = out.getvalue()
    return text if carriage_return else text[:-1]


def search(bigsi, seq, threshold=1.0, score=False, format="json"):
    d = search_record(seq, threshold, bigsi.search(seq, threshold, score))
    return d_to_csv(d) if format == "csv" else json.dumps(d, indent=4)


def _bulk_text_native(bigsi, fasta, threshold, format, score=False):
    """The text of a bulk search without a Python object per record: bigsi_hip_fasta_pack -> bigsi_hip_search_stream(_scored) ->
    bigsi_hip_format_results(_scored) (include/bigsi_hip.h, "FRONT-END TEXT").  None when this route does not apply -- text that is
    not plain ASCII (file or sample names), a multi-GPU index, a record on which the reference raises, more scored hits than one
    call should hold: the per-record route below then does what it always did."""
    import numpy as np
    from . import _lib
    from .graph.metadata import DELETION_SPECIAL_SAMPLE_NAME
    st = getattr(bigsi, "storage", None)
    if st is None or not hasattr(st, "search_many_packed") or st.res.is_group or not isinstance(fasta, str):
        return None
    with open(fasta, "rb") as f:
        packed = _lib.fasta_pack(f.read())
    if packed is None:
        return None
    blob, soff = packed
    scored = None
    if len(soff) > 1 and int(np.diff(soff.astype(np.int64)).min()) < bigsi.kmer_size:
        return None                           # a record without k-mers: the reference raises in record order (no device pass wasted on it)
    with bigsi._device_lock():
        if score:
            from .graph.bigsi import SCORE_SLICE_CHARS
            from .scoring import score_columns
            from .storage.hip_hbm import TooManyHits
            try:
                nk, nu, off, col, cnt, bits, boff, rec = st.search_many_scored(None, bigsi.kmer_size, threshold, max_bits=SCORE_SLICE_CHARS // 8, packed=(blob, soff))
            except TooManyHits:
                return None
        else:
            nk, nu, off, col, cnt = st.search_many_packed(blob, soff, bigsi.kmer_size, threshold)
    if len(nu) and int(nu.min()) == 0:
        return None
    if score:
        if len(rec) and int(rec["num_kmers"].min()) < 2:
            return None                       # (a scored hit of a one-k-mer query: IndexError in the reference, in record order)
        c_ = score_columns(rec, bigsi.scorer.DB_SIZE, as_arrays=True)          # evalue, pvalue, log_evalue, log_pvalue: numpy's exp / log10
        scored = (rec, bits, boff, c_[13], c_[14], c_[15], c_[16], 31)
    ns = bigsi.num_samples
    used = np.unique(col)
    used = used[used < ns].tolist()
    try:
        names = [bigsi.colour_to_sample(c) for c in used]
        enc = [nm.encode("ascii") for nm in names]
    except (KeyError, UnicodeEncodeError):
        return None
    name_off, deleted = np.zeros(ns + 1, np.uint64), np.zeros(max(ns, 1), np.uint8)
    if used:
        lens = np.zeros(ns, np.uint64)
        lens[used] = [len(e) for e in enc]
        np.cumsum(lens, out=name_off[1:])
        deleted[[c for c, nm in zip(used, names) if nm == DELETION_SPECIAL_SAMPLE_NAME]] = 1
    try:
        return _lib.format_results(1 if format == "csv" else 0, blob, soff, threshold, json.dumps(CITATION), nu, off, col, cnt, b"".join(enc) + b"\0",
                                   name_off, deleted, scored=scored)
    except _lib.BigsiHipError as e:
        if e.code == _lib.ERR_STATE:
            return None
        raise


def bulk_search(bigsi, fasta, threshold=1.0, score=False, format="json", stream=False, out=None):
    """All records of a FASTA file in one device batch.  Returns the combined text (stream=False) or prints one record
    per line as the reference's streaming branch does and returns None."""
    if not stream and hasattr(bigsi, "_device_lock"):
        text = _bulk_text_native(bigsi, fasta, threshold, format, score)
        if text is not None:
            return text
    seqs = [s for _, s in read_fasta(fasta)]
    if hasattr(bigsi, "search_stream"):       # the C ABI's streaming searches; the text of one slice is assembled while the next one runs
        size = getattr(bigsi, "config", {}).get("batch_size")          # default: slices of ~4M k-mers
        pairs = bigsi.search_stream(seqs, threshold, score, batch_size=size)
    else:
        pairs = zip(seqs, bigsi.search_batch(seqs, threshold, score) if seqs else [])
    if not stream:
        # the reference's text -- json.dumps(list of records, indent=4) / the csv rows of every record -- written record by record: most
        # records of a bulk search have no results, and their text is a constant around the query (json.dumps with indent runs the
        # pure-Python encoder: ~20 us per record; this: ~0.3 us)
        plain = ("percent_kmers_found", "num_kmers", "num_kmers_found", "sample_name")      # an unscored result's keys, in the reference's order
        if format == "csv":
            def rows(s, r):
                if tuple(r[0]) != plain:
                    return d_to_csv(search_record(s, threshold, r), False, False)
                q = s.replace('"', '""')          # csv.writer, QUOTE_NONNUMERIC: strings quoted, quotes doubled; numbers as repr(); sorted keys
                text = "".join('"%s",%d,%d,%r,"%s"\r\n' % (q, x["num_kmers"], x["num_kmers_found"], x["percent_kmers_found"], x["sample_name"].replace('"', '""')) for x in r)
                return text[:-1]
            return "\n".join(rows(s, r) if r else "" for s, r in pairs)
        th, cit = json.dumps(threshold), json.dumps(CITATION)
        head = "    {\n        \"query\": "
        tail = ",\n        \"threshold\": %s,\n        \"results\": [],\n        \"citation\": %s\n    }" % (th, cit)
        mid = ",\n        \"threshold\": %s,\n        \"results\": [\n" % th
        end = "\n        ],\n        \"citation\": %s\n    }" % cit
        one = ('            {\n                "percent_kmers_found": %r,\n                "num_kmers": %d,\n                "num_kmers_found": %d,\n'
               '                "sample_name": %s\n            }')

        def record(s, r):
            if not r:
                return head + json.dumps(s) + tail
            if tuple(r[0]) != plain:                     # scored results: the general encoder, indented one level
                return "\n".join("    " + line for line in json.dumps(search_record(s, threshold, r), indent=4).split("\n"))
            return head + json.dumps(s) + mid + ",\n".join(one % (x["percent_kmers_found"], x["num_kmers"], x["num_kmers_found"], json.dumps(x["sample_name"])) for x in r) + end
        parts = [record(s, r) for s, r in pairs]
        return "[\n" + ",\n".join(parts) + "\n]" if parts else "[]"
    dd = [search_record(s, threshold, r) for s, r in pairs]
    out = out or sys.stdout
    with_header, carriage_return = True, True
    for i, d in enumerate(dd):
        if format == "csv":
            # the reference's flags persist across iterations (__main__.py:298-305): header on the first record, and on
            # the last record only if the record before it was the first
            if i == 0:
                with_header, carriage_return = True, False
            elif i == len(dd) - 1:
                carriage_return = True
            else:
                with_header, carriage_return = False, False
            print(d_to_csv(d, with_header, carriage_return), file=out)
        else:
            print(json.dumps(d), file=out)
    return None


def variant_search(bigsi, reference, ref, pos, alt, gene=None, genbank=None, format="json", probes=None):
    """Text of the reference's `variant_search` command (bigsi/__main__.py:222-246)."""
    from .variant_search import BIGSIAminoAcidMutationSearch, BIGSIVariantSearch
    if genbank and gene:
        d = BIGSIAminoAcidMutationSearch(bigsi, reference, genbank).search(gene, ref, pos, alt, probes=probes)
    elif genbank or gene:
        raise ValueError("genbank and gene must be supplied together")
    else:
        d = BIGSIVariantSearch(bigsi, reference).search(ref, pos, alt, probes=probes)
    d["citation"] = CITATION
    return d_to_csv(d) if format == "csv" else json.dumps(d, indent=4)
