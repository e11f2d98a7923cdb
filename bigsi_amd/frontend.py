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
    f = open(path_or_file) if isinstance(path_or_file, str) else path_or_file
    try:
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
            f.close()


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


def bulk_search(bigsi, fasta, threshold=1.0, score=False, format="json", stream=False, out=None):
    """All records of a FASTA file in one device batch.  Returns the combined text (stream=False) or prints one record
    per line as the reference's streaming branch does and returns None."""
    seqs = [s for _, s in read_fasta(fasta)]
    if hasattr(bigsi, "search_stream"):       # device batches of `batch_size`, host assembly overlapped with the next batch
        size = getattr(bigsi, "config", {}).get("batch_size")          # default: batches of ~524k k-mers
        dd = [search_record(s, threshold, r) for s, r in bigsi.search_stream(seqs, threshold, score, batch_size=size)]
    else:
        results = bigsi.search_batch(seqs, threshold, score) if seqs else []
        dd = [search_record(s, threshold, r) for s, r in zip(seqs, results)]
    if not stream:
        if format == "csv":
            return "\n".join(d_to_csv(d, False, False) for d in dd)
        return json.dumps(dd, indent=4)
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
