"""Front-end text output (search / bulk_search, JSON + CSV + streaming) against goldens captured from the reference's
own bigsi.__main__ (tests/golden/g9_frontend.json).  CPU part: formatting with a stub index that replays the golden
results; GPU part: the whole thing end to end on the device."""
import io
import json

import pytest

from conftest import load_golden


def _golden_results(g):
    """(seq, threshold, score) -> result list, recovered from the golden JSON outputs."""
    table = {}
    for c in g["cases"]:
        if c["format"] != "json":
            continue
        if c["cmd"] == "search":
            d = json.loads(c["out"])
            table[(d["query"], c["threshold"], c["score"])] = d["results"]
        elif not c["stream"]:
            for d in json.loads(c["out"]):
                table[(d["query"], c["threshold"], c["score"])] = d["results"]
    return table


class ReplayIndex(object):
    def __init__(self, table):
        self.table = table

    def search(self, seq, threshold=1.0, score=False):
        return self.table[(seq, threshold, score)]

    def search_batch(self, seqs, threshold=1.0, score=False):
        return [self.table[(s, threshold, score)] for s in seqs]


def _run_cases(index, g, tmp_path, compare):
    from bigsi_amd import frontend
    paths = {}
    for name, text in g["fasta_text"].items():
        p = tmp_path / (name + ".fasta")
        p.write_text(text)
        paths[name] = str(p)
    n = 0
    for c in g["cases"]:
        if c["cmd"] == "search":
            got = frontend.search(index, c["seq"], c["threshold"], c["score"], c["format"])
            compare(got, c["out"], c)
        else:
            buf = io.StringIO()
            got = frontend.bulk_search(index, paths[c["fasta"]], c["threshold"], c["score"], c["format"], c["stream"], out=buf)
            compare(got, c["out"], c)
            if c["stream"]:
                compare(buf.getvalue(), c["stdout"], c)
        n += 1
    assert n == len(g["cases"]) and n > 40


def test_frontend_formatting_replay(tmp_path):
    g = load_golden("g9_frontend.json")

    def same(got, want, c):
        assert got == want, (c["cmd"], c.get("fasta"), c["threshold"], c["score"], c["format"], c.get("stream"))

    _run_cases(ReplayIndex(_golden_results(g)), g, tmp_path, same)


def test_read_fasta(tmp_path):
    from bigsi_amd.frontend import read_fasta
    p = tmp_path / "a.fa"
    p.write_text(">r1 desc\nACGT\nAC\n\n>r2\nGG\n>empty\n")
    assert read_fasta(str(p)) == [("r1 desc", "ACGTAC"), ("r2", "GG"), ("empty", "")]


@pytest.mark.gpu
def test_frontend_end_to_end_on_device(tmp_path):
    import bigsi_amd
    from conftest import FLOAT_TOL_KEYS
    g = load_golden("g9_frontend.json")
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "g9"}, "k": g["k"], "m": g["m"], "h": g["h"]}
    blooms = [bigsi_amd.BIGSI.bloom(cfg, ks) for ks in g["samples"].values()]
    b = bigsi_amd.BIGSI.build(cfg, blooms, list(g["samples"].keys()))

    def same(got, want, c):
        if got == want:
            return
        # the only permitted difference: last-ulp evalue/pvalue digits inside score dicts (see conftest.FLOAT_TOL_KEYS)
        assert c["score"] and c["format"] == "json", (c["cmd"], c["threshold"], c["format"])
        def strip(x):
            if isinstance(x, dict):
                return {k: strip(v) for k, v in x.items() if k not in FLOAT_TOL_KEYS}
            if isinstance(x, list):
                return [strip(v) for v in x]
            return x
        parse = (lambda t: [json.loads(l) for l in t.splitlines()]) if c.get("stream") else json.loads
        assert strip(parse(got)) == strip(parse(want))

    cases = dict(g)
    # csv with score embeds evalue/pvalue digits too: compare those through the json cases only
    cases["cases"] = [c for c in g["cases"] if not (c["score"] and c["format"] == "csv")]
    n_all = len(g["cases"])
    try:
        from bigsi_amd import frontend
        paths = {}
        for name, text in g["fasta_text"].items():
            p = tmp_path / (name + ".fasta")
            p.write_text(text)
            paths[name] = str(p)
        for c in cases["cases"]:
            if c["cmd"] == "search":
                same(frontend.search(b, c["seq"], c["threshold"], c["score"], c["format"]), c["out"], c)
            else:
                buf = io.StringIO()
                got = frontend.bulk_search(b, paths[c["fasta"]], c["threshold"], c["score"], c["format"], c["stream"], out=buf)
                if c["stream"]:
                    assert got is None
                    same(buf.getvalue(), c["stdout"], c)
                else:
                    same(got, c["out"], c)
        assert n_all > 40
    finally:
        b.delete()


@pytest.mark.gpu
def test_bulk_search_native_text_route_is_the_per_record_routes_text(tmp_path, monkeypatch):
    """bulk_search's text (score=True too: bigsi_hip_search_stream_scored -> bigsi_hip_format_results_scored) comes from
    bigsi_hip_fasta_pack -> bigsi_hip_search_stream -> bigsi_hip_format_results; forcing
    the per-record Python route on the same file must give the same characters: 3000 reads of which every tenth was added to a few
    samples (one of them deleted afterwards, one named with characters JSON and CSV escape), exact and thresholded, both formats.
    A file with a record the reference raises on, and a non-ASCII file, fall through to the per-record route."""
    import numpy as np
    import bigsi_amd
    from bigsi_amd import frontend
    rng = np.random.default_rng(11)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(3000, 70), dtype=np.uint8)]
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "native_text"}, "k": 31, "m": 200003, "h": 3}
    names = ["s%02d" % i for i in range(12)]
    names[4] = 'odd "name", \\ with\ttab'
    groups = [[] for _ in names]
    for i in range(0, len(reads), 10):
        r = reads[i]
        kms = [r[j:j + 31] for j in range(len(r) - 30)]
        for t in range(1 + i % 4):
            c = (i // 10 + 5 * t) % len(names)
            groups[c].extend(kms if t % 2 == 0 else kms[:25])      # whole reads in some samples, part of them in others
    b = bigsi_amd.BIGSI.build(cfg, [bigsi_amd.BIGSI.bloom(cfg, ks or ["A" * 31]) for ks in groups], names)
    try:
        b.delete_sample(names[7])
        fa = tmp_path / "reads.fa"
        fa.write_text("".join(">r%d some text\n%s\n%s\n" % (i, r[:40], r[40:]) for i, r in enumerate(reads)))
        taken = []
        real = frontend._bulk_text_native
        monkeypatch.setattr(frontend, "_bulk_text_native", lambda *a: taken.append(real(*a)) or taken[-1])
        for fmt in ("json", "csv"):
            for thr, score in ((1.0, False), (0.6, False), (0.0, False), (1.0, True), (0.6, True), (0.0, True)):
                taken.clear()
                got = frontend.bulk_search(b, str(fa), thr, score, fmt)
                assert taken and taken[0] is not None and got is taken[0]
                monkeypatch.setattr(frontend, "_bulk_text_native", lambda *a: None)
                want = frontend.bulk_search(b, str(fa), thr, score, fmt)
                monkeypatch.setattr(frontend, "_bulk_text_native", lambda *a: taken.append(real(*a)) or taken[-1])
                assert got == want, (fmt, thr, score)
                if score and fmt == "json" and thr == 0.6:
                    assert sum(len(x) == 22 for r in json.loads(got) for x in r["results"]) >= 300
                if fmt == "json" and thr == 1.0:
                    recs = json.loads(got)
                    assert len(recs) == len(reads) and sum(len(r["results"]) for r in recs) >= 300
                    assert any(x["sample_name"] == names[4] for r in recs for x in r["results"])
                    assert not any(x["sample_name"] == names[7] for r in recs for x in r["results"])
        # a record too short to have k-mers: the reference raises, and so does bulk_search (through the per-record route)
        short = tmp_path / "short.fa"
        short.write_text(">a\n%s\n>b\nACGT\n" % reads[0])
        taken.clear()
        with pytest.raises(TypeError):
            frontend.bulk_search(b, str(short), 1.0)
        assert taken == [None]
        # non-ASCII text in the file: not this route's business
        odd = tmp_path / "odd.fa"
        odd.write_text(">a\n%s\n>b\n%sé%s\n" % (reads[0], reads[1][:35], reads[1][35:]), encoding="utf-8")
        taken.clear()
        text = frontend.bulk_search(b, str(odd), 1.0)
        assert taken == [None] and len(json.loads(text)) == 2
    finally:
        b.delete()


@pytest.mark.gpu
def test_c_host_bulk_search_prints_the_references_text(tmp_path):
    """tests/c_host/bulk_host.c -- C99, links libbigsi_hip.so only -- does `bigsi bulk_search` in three calls of the C ABI
    (bigsi_hip_fasta_pack, bigsi_hip_search_stream, bigsi_hip_format_results); what it prints for golden G9's index and FASTA files
    must be the reference's own bulk_search output, byte for byte, JSON and CSV, at every threshold G9 holds."""
    import os
    import subprocess
    from bigsi_amd import _lib, frontend
    g = load_golden("g9_frontend.json")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bulk_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "c_host", "bulk_host.c"), "-L", os.path.dirname(_lib.LIB_PATH), "-lbigsi_hip",
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    index = "%d %d %d %d\n" % (g["m"], g["h"], g["k"], len(g["samples"]))
    index += "".join("%s\n%d %s\n" % (name, len(kmers), " ".join(kmers)) for name, kmers in g["samples"].items())
    n = 0
    for c in g["cases"]:
        if c["cmd"] != "bulk_search" or c["score"] or c["stream"]:
            continue
        fa = tmp_path / (c["fasta"] + ".fasta")
        fa.write_text(g["fasta_text"][c["fasta"]])
        r = subprocess.run([exe, str(fa), repr(c["threshold"]), c["format"], json.dumps(c["threshold"]), json.dumps(frontend.CITATION)],
                           input=index.encode(), capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout.decode() == c["out"], (c["fasta"], c["threshold"], c["format"])
        n += 1
    assert n >= 4
