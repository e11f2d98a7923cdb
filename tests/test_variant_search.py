"""Variant genotyping (bigsi_amd/variant_search.py) against goldens captured from the reference's own
bigsi/cmds/variant_search.py and `variant_search` command (tests/golden/g12_variant.json).  The reference emits results
in set-iteration order, so lists are compared sorted by sample name.  CPU part: genotype logic / probe split / text with
a stub index; GPU part: the searches on the device."""
import json

import pytest

from conftest import load_golden


def norm(res):
    return sorted(res, key=lambda r: r["sample_name"])


class StubIndex(object):
    kmer_size = 21

    def __init__(self, table):
        self.table, self.batches = table, 0

    def search_batch(self, seqs, threshold=1.0, score=False):
        assert threshold == 1 and not score
        self.batches += 1
        return [[{"sample_name": n} for n in self.table[s]] for s in seqs]


def _probe_hits(case):
    """probe sequence -> sample names, reconstructed from one golden case with a single alt probe."""
    refs = [r["sample_name"] for r in case["results"] if r["genotype"] in ("0/0", "0/1")]
    alts = [r["sample_name"] for r in case["results"] if r["genotype"] in ("1/1", "0/1")]
    return {case["refs"][0]: refs, case["alts"][0]: alts}


def test_genotypes_and_probe_split():
    from bigsi_amd.variant_search import BIGSIVariantSearch, genotypes, split_probes
    assert genotypes(["a", "b", "a"], ["b", "c"]) == [{"sample_name": "a", "genotype": "0/0"}, {"sample_name": "b", "genotype": "0/1"},
                                                     {"sample_name": "c", "genotype": "1/1"}]
    assert genotypes([], []) == []
    g = load_golden("g12_variant.json")
    for case in g["cases"]:
        refs, alts = split_probes(case["probes_fasta"])
        assert (refs, alts) == (case["refs"], case["alts"])
        assert split_probes(case["probes_fasta"].encode()) == (refs, alts)
    singles = [c for c in g["cases"] if len(c["alts"]) == 1]
    assert len(singles) >= 8
    for case in singles:
        stub = StubIndex(_probe_hits(case))
        d = BIGSIVariantSearch(stub, "ref.fa").search(case["ref_base"], case["pos"], case["alt_base"], probes=case["probes_fasta"])
        assert d["query"] == case["query"] and norm(d["results"]) == case["results"] and stub.batches == 1


def test_variant_search_text_replay(tmp_path):
    from bigsi_amd import frontend
    g = load_golden("g12_variant.json")
    case = g["cases"][0]
    stub = StubIndex(_probe_hits(case))
    p = tmp_path / "probes.fa"
    p.write_text(case["probes_fasta"])
    for rec in g["cli"]:
        text = frontend.variant_search(stub, "ref.fa", case["ref_base"], case["pos"], case["alt_base"], rec["gene"], rec["genbank"],
                                       rec["format"], probes=str(p))
        _check_text(text, rec)
    with pytest.raises(ValueError):
        frontend.variant_search(stub, "ref.fa", "A", 1, "T", gene="rpoB", probes=str(p))
    assert g["gene_without_genbank"] == "ValueError"


def _check_text(text, rec):
    want = rec["out"]
    if rec["format"] == "json":
        d = json.loads(text)
        assert list(d.keys()) == want["key_order"] and text.startswith('{\n    "') == want["indent4"]
        d["results"] = norm(d["results"])
        assert d == want["parsed"]
    else:
        lines = text.split("\r\n")
        assert lines[0] == want["header"] and sorted(lines[1:-1]) == want["rows_sorted"] and lines[-1] == want["tail"]


def test_missing_probe_tool_is_an_error():
    from bigsi_amd.variant_search import BIGSIVariantSearch
    import shutil
    if shutil.which("mykrobe"):
        pytest.skip("mykrobe is installed")
    with pytest.raises(FileNotFoundError):        # same failure as the reference's subprocess call on a host without it
        BIGSIVariantSearch(StubIndex({}), "ref.fa").search("A", 1, "T")


@pytest.mark.gpu
def test_variant_search_on_device(tmp_path):
    import bigsi_amd
    from bigsi_amd import frontend
    from bigsi_amd.utils import seq_to_kmers
    from bigsi_amd.variant_search import BIGSIVariantSearch
    g = load_golden("g12_variant.json")
    cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "g12"}, "k": g["k"], "m": g["m"], "h": g["h"]}
    blooms = [bigsi_amd.BIGSI.bloom(cfg, [km for s in seqs for km in seq_to_kmers(s, g["k"])]) for seqs in g["samples"].values()]
    b = bigsi_amd.BIGSI.build(cfg, blooms, list(g["samples"].keys()))
    try:
        for name in g["deleted"]:
            b.delete_sample(name)
        vs = BIGSIVariantSearch(b, "ref.fa")
        for case in g["cases"]:
            d = vs.search(case["ref_base"], case["pos"], case["alt_base"], probes=case["probes_fasta"])
            assert d["query"] == case["query"] and norm(d["results"]) == case["results"], case["query"]
            assert norm(vs.genotype_alleles(case["refs"], case["alts"])) == case["results"]
        many = vs.genotype_many([(c["refs"], c["alts"]) for c in g["cases"]])
        assert [norm(r) for r in many] == [c["results"] for c in g["cases"]]
        case = g["cases"][0]
        for rec in g["cli"]:
            _check_text(frontend.variant_search(b, "ref.fa", case["ref_base"], case["pos"], case["alt_base"], rec["gene"], rec["genbank"],
                                                rec["format"], probes=case["probes_fasta"]), rec)
    finally:
        b.delete()
