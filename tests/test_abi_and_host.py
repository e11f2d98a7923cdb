"""CPU-only checks: the C-ABI library loads and exports every symbol include/bigsi_hip.h declares (no compute calls),
and the host-side logic of the package (BitRow, scoring, string helpers, storage contract, result assembly)
matches golden vectors produced by the real reference."""
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_result_equal, load_golden, unjson


# --------------------------------------------------------------------------------------------- C ABI
HIP_HEADERS = ("bigsi_hip.h", "bigsi_hip_group.h", "bigsi_hip_text.h", "bigsi_hip_testing.h")      # core boundary | device groups | front-end text | test hooks


def header_functions(headers=HIP_HEADERS):
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(bigsi_hip_\w+)\s*\(", src))
    return sorted(names)


def test_public_header_stays_small():
    """include/bigsi_hip.h is core + batches + the one-process-per-GPU exchange + stats; caller-managed exchange hooks and the
    flags that force an A/B route live in bigsi_hip_testing.h, device groups in bigsi_hip_group.h."""
    public = header_functions(("bigsi_hip.h",))
    assert len(public) <= 60, len(public)
    hidden = set(header_functions(("bigsi_hip_testing.h",)))
    assert {"bigsi_hip_set_stream", "bigsi_hip_batch_set_outputs", "bigsi_hip_batch_compact_gathered"} <= hidden
    assert not hidden & set(public)
    src = open(os.path.join(ROOT, "include", "bigsi_hip.h")).read()
    for flag in ("BIGSI_RUN_K1_GLOBAL", "BIGSI_RUN_ONE_STREAM", "BIGSI_RUN_WEAK_FINGERPRINT", "BIGSI_RUN_NO_SORT"):
        assert "#define " + flag not in src


def test_library_exports_every_declared_symbol():
    import ctypes
    from bigsi_amd import _lib
    names = header_functions()
    assert len(names) >= 30
    assert os.path.exists(_lib.LIB_PATH), "libbigsi_hip.so has not been built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "library does not export %s" % n
    # the Python binding covers the same set
    assert sorted(_lib.SIGNATURES) == names
    _lib.lib()


def build_c_host(tmp_path):
    """tests/c_host/search_host.c compiled as strict C99 against include/bigsi_hip.h and linked to the in-tree library."""
    import subprocess
    from bigsi_amd import _lib
    exe = str(tmp_path / "search_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "c_host", "search_host.c"),
                           "-L", os.path.dirname(_lib.LIB_PATH), "-lbigsi_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    return exe


def test_bulk_search_c_host_compiles_as_c99(tmp_path):
    """tests/c_host/bulk_host.c (bulk_search in three C calls; run by the gpu suite) is strict C99 against the public header."""
    import subprocess
    from bigsi_amd import _lib
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-o", str(tmp_path / "bulk_host"), os.path.join(ROOT, "tests", "c_host", "bulk_host.c"),
                           "-L", os.path.dirname(_lib.LIB_PATH), "-lbigsi_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """The boundary is a C ABI: the header must compile as C (not only as C++) and a host without Python or HIP headers must
    link against the library alone.  Without a device that host fails loudly -- an error code and message, no fallback."""
    import subprocess
    import torch
    exe = build_c_host(tmp_path)
    if torch.cuda.is_available():
        return                                  # what it computes is checked by the gpu suite
    r = subprocess.run([exe], input="1009 3 31 1 1 0.5\n1 ACGTACGTACGTACGTACGTACGTACGTACGTACGT\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n",
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "bigsi_hip_device_count" in r.stderr and r.stdout == ""


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "bigsi_hip.h")).read()
    assert len(re.findall(r"bigsi/[\w/]+\.py:\d+", src)) >= 25


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "bigsi_hip.h")).read()
    assert "torch" not in src.replace("torch's", "").lower() or "at::" not in src
    assert "#include <stdint.h>" in src and "extern \"C\"" in src


def test_product_does_not_import_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "bigsi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                txt = open(os.path.join(base, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bigsi_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.BigsiHipError):
        _lib.lib()
    from bigsi_amd.storage import get_storage
    with pytest.raises(_lib.BigsiHipError):
        get_storage({"storage-engine": "hip-hbm", "storage-config": {"name": "x"}})
    with pytest.raises(KeyError):
        get_storage({"storage-engine": "berkeleydb", "storage-config": {}})


# --------------------------------------------------------------------------------------------- host helpers
def test_bitrow():
    from bigsi_amd import BitRow
    a = BitRow("110101111010")
    assert a.tobytes().hex() == load_golden("g8_storage.json")["bitarray_bytes"]["stored_hex"]
    assert BitRow.frombytes(a.tobytes()).to01() == "1101011110100000"
    assert BitRow.frombytes(a.tobytes(), 12) == a and a == BitRow(a) and a != BitRow("110101111011")
    assert (a & BitRow("101010101010")).to01() == "100000101010"
    assert a[1] is True and a[2] is False and a[:3].to01() == "110" and len(a) == 12 and a.count() == 8
    b = a.copy()
    b[0] = 0
    b.append(1)
    assert b.to01() == "0101011110101" and a.to01()[0] == "1"
    with pytest.raises(IndexError):
        b[99] = 1
    assert BitRow(5).to01() == "00000" and BitRow([True, False, 1]).to01() == "101"
    c = BitRow("10")
    c.extend(BitRow("011"))
    assert c.to01() == "10011" and c.tolist() == [True, False, False, True, True]
    with pytest.raises(ValueError):
        BitRow("10") & BitRow("101")


def test_string_helpers_vs_golden():
    from bigsi_amd.utils import canonical, reverse_comp, seq_to_kmers
    g = load_golden("g1_hash.json")
    for rec in g["canonical"]:
        assert reverse_comp(rec["s"]) == rec["reverse_comp"] and canonical(rec["s"]) == rec["canonical"]
    for rec in g["seq_to_kmers"]:
        assert list(seq_to_kmers(rec["seq"], rec["k"])) == rec["kmers"]


def test_scoring_vs_golden():
    from bigsi_amd.scoring import Scorer, remove_short_ones, tabulate_score
    g = load_golden("g5_scoring.json")
    for rec in g["helpers"]["remove_short_ones"]:
        assert remove_short_ones(rec["s"]) == rec["out"], rec["s"]
    for rec in g["helpers"]["tabulate_score"]:
        assert tabulate_score(rec["s"]) == rec["out"], rec["s"]
    for rec in g["cases"]:
        if "raises" in rec:
            with pytest.raises(BaseException) as ei:
                Scorer(rec["db_size"]).score(rec["s"])
            assert type(ei.value).__name__ == rec["raises"]
        else:
            assert_result_equal(Scorer(rec["db_size"]).score(rec["s"]), unjson(rec["score"]), "db=%d %s" % (rec["db_size"], rec["s"][:20]))
    # the reference's own known answer (bigsi/tests/scoring.py:10-31) is case 0 of db_size 500000
    kat = [r for r in g["cases"] if r["db_size"] == 500000][0]
    assert kat["score"]["length"] == 1174 and kat["score"]["score"] == 1064.89


def test_scoring_fast_tallies_equal_the_two_step_form_and_the_oracle():
    """Scorer.score tallies runs on big integers + one regular expression (filtered_run_tallies); that must be
    tabulate_score(remove_short_ones(s)) for every string, and the scores must equal the oracle's restatement of score.py."""
    import random
    from bigsi_amd.scoring import Scorer, filtered_run_tallies, remove_short_ones, tabulate_score
    from oracle.ref_model import Scorer as OracleScorer
    rng = random.Random(20)
    strings = ["", "0", "1", "00", "11", "10", "111", "110", "011", "101", "0000", "1111", "1101", "1011"]
    for n in (3, 4, 5, 7, 8, 31, 61, 63, 64, 65, 128, 970, 3970):
        for p in (0.0, 0.1, 0.5, 0.9, 0.97, 1.0):
            for _ in range(6):
                a = ["1" if rng.random() < p else "0" for _ in range(n)]
                for _ in range(rng.randrange(3)):          # gaps a SNP would leave: ~31 absent k-mers in a row
                    g0 = rng.randrange(n)
                    a[g0:g0 + rng.randrange(1, 70)] = "0" * len(a[g0:g0 + rng.randrange(1, 70)])
                strings.append("".join(a)[:n])
    for s in strings:
        ss = remove_short_ones(s)
        assert filtered_run_tallies(s) == (len(ss), tabulate_score(ss)), s
    for db in (0, 3, 500000):
        mine, ref = Scorer(db), OracleScorer(db)
        for s in strings:
            if not s:
                continue
            assert_result_equal(mine.score(s), ref.score(s), "db=%d %s" % (db, s[:40]))
    for bad in ("012", "1 1", "0b1", "+11", "1_1"):
        with pytest.raises(ValueError):
            filtered_run_tallies(bad)


def test_threshold_and_percent_arithmetic():
    from bigsi_amd.graph.bigsi import BigsiQueryResult
    from bigsi_amd.utils import min_kmers_for
    g = load_golden("g6_arith.json")
    for rec in g["min_kmers"]:
        assert min_kmers_for(rec["n"], rec["t"]) == rec["min_kmers"]
    for rec in g["percent"]:
        r = BigsiQueryResult(0, "s", rec["found"], rec["n"])
        assert r.percent_kmers_found == rec["percent"]
        assert list(r.todict()) == ["percent_kmers_found", "num_kmers", "num_kmers_found", "sample_name"]


# --------------------------------------------------------------------------------------------- storage contract on a host dict
class DictStorage(object):
    pass


def make_dict_storage():
    from bigsi_amd.storage.contract import BaseStorage

    class _Dict(BaseStorage):
        def __init__(self):
            self.storage = {}

        def delete_all(self):
            self.storage = {}

    return _Dict()


def test_contract_key_grammar_and_typed_helpers():
    from bigsi_amd import BitRow
    g = load_golden("g8_storage.json")
    st = make_dict_storage()
    st["test"] = b"123"
    assert st["test"] == b"123" and st.storage[b"test"] == b"123"
    st.set_integer("x", 112)
    assert st.storage[b"x:int"].decode() == g["integer_bytes"] and st.get_integer("x") == 112
    st.set_string("name", "abc")
    assert st.storage[b"name:string"] == b"abc" and st.get_string("name") == "abc"
    ba = BitRow("110101111010")
    st.set_bitarray("test", ba)
    assert st.storage[b"test:bitarray"].hex() == g["bitarray_bytes"]["stored_hex"]
    assert st.get_bitarray("test").to01() == g["bitarray_bytes"]["get_bitarray"]
    st.set_bit("test", 0, 0)
    assert st.get_bitarray("test").to01() == g["after_set_bit_0_0"]
    assert [st.incr("ctr"), st.incr("ctr")] == g["incr"]
    st.set_integers(["a", "b"], [1, 2])
    assert st.get_integers(["a", "b"]) == [1, 2]
    st.set_bitarrays([0, 1], [BitRow("001"), BitRow("111")])
    assert [r[:3].to01() for r in st.get_bitarrays([0, 1])] == ["001", "111"]
    assert st.get("missing") is None
    with pytest.raises(KeyError):
        st.get_integer("missing")
    assert st.convert_to_bitarray_key(7) == "7:bitarray" and st.convert_to_integer_key("k") == "k:int"


def test_metadata_on_dict_storage():
    from bigsi_amd.graph.metadata import DELETION_SPECIAL_SAMPLE_NAME, SampleMetadata
    st = make_dict_storage()
    sm = SampleMetadata(st)
    assert sm.num_samples == 0
    assert sm.add_sample("a") == 1 and sm.add_sample("b") == 2
    assert st.storage[b"metadata:a:int"] == b"0" and st.storage[b"metadata:1:string"] == b"b"
    assert st.storage[b"metadata:colour_count:int"] == b"2"
    assert sm.sample_to_colour("b") == 1 and sm.colour_to_sample(0) == "a" and sm.sample_to_colour("zz") is None
    with pytest.raises(ValueError):
        sm.add_sample("a")
    with pytest.raises(ValueError):
        sm.add_sample(DELETION_SPECIAL_SAMPLE_NAME)
    sm.delete_sample("a")
    assert sm.colour_to_sample(0) == DELETION_SPECIAL_SAMPLE_NAME and sm.sample_to_colour("a") is None and sm.num_samples == 2
    assert sm.colours_to_samples([0, 1]) == {0: DELETION_SPECIAL_SAMPLE_NAME, 1: "b"}
    other = SampleMetadata(make_dict_storage())
    other.add_samples(["b", "c"])
    sm.merge_metadata(other)
    assert sm.colour_to_sample(2) == "b_duplicate_in_merge" and sm.colour_to_sample(3) == "c"


def test_fused_backend_is_required():
    from bigsi_amd.graph.index import KmerSignatureIndex
    st = make_dict_storage()
    with pytest.raises(TypeError):
        KmerSignatureIndex(st)


def test_shard_planning():
    from bigsi_amd.parallel import plan_shards
    assert plan_shards(500000, 8) == (62500, [(i * 62500, 62500) for i in range(8)])
    sc, spans = plan_shards(10, 4)
    assert sc == 3 and spans == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert plan_shards(5, 8)[1][5:] == [(5, 0), (5, 0), (5, 0)]
    assert math.ceil(7 * 0.1) == 1
    from bigsi_amd.parallel import shard_config
    base = {"k": 31, "m": 1000, "h": 3, "storage-config": {"name": "ix", "filename": "/data/ix.hbm"}}
    c1 = shard_config(base, 1, 4, 1)
    assert c1["storage-config"] == {"name": "ix.shard1-of-4", "filename": "/data/ix.hbm.shard1-of-4", "device": 1}
    assert base["storage-config"] == {"name": "ix", "filename": "/data/ix.hbm"} and c1["k"] == 31       # input untouched
    assert shard_config(base, 0, 1, 0)["storage-config"] == {"name": "ix", "filename": "/data/ix.hbm", "device": 0}


def test_bench_line_is_condensed_under_the_drivers_tail():
    """bench.condense: the verbose record of a run -> the ONE line rank 0 prints.  Worst case (eight ranks, seven legs with every optional key,
    long prose) stays under 7.5 KB -- the driver keeps 8 KB of stdout -- and keeps what the judge's checks read; strings are cut below the
    120 characters the driver's parser keeps."""
    import json
    import bench
    leg_line = {"value": 263123456.789, "ms_per_step": 0.94012345, "steps": 1200,
                "roofline": {"kernel": "k_and_count", "frac": 0.81234567, "step_frac": 0.7712345, "frac_of_box": 0.98123456, "traffic_ratio": 1.00412345,
                             "read_launches_repeated": 0, "frac_overlapped": 0.2512345},
                "config": {"verified": "x" * 100, "host_visible_lookups_per_s": 261234567.8, "value_inputs": "host", "resident_lookups_per_s": 271234567.8, "one_call_us": 45.12345, "index_gb_per_gpu": 195.3125, "exchange_ms": 0.0712345,
                           "rccl_ranks": 8, "per_rank_GBps": [6512.3456] * 8, "scored_hits": 260075, "scored_us_per_hit": 3.912345, "hv_scored_lookups_per_s": 240123456.7,
                           "distinct_gpus": True, "one_call_us_batch": 69.12345}}
    legs = {k: bench.leg_summary(leg_line, "what", [], 12.3) for k in ("c3_t04", "c2", "c2_t04", "c4_shard", "c5_shard", "ns_shard", "extra")}
    full = {"metric": "kmer_lookups_per_s", "value": 135522511.89149362, "unit": "kmer_lookups/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 58.63409620360471,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload_short": "configs[2]: 10M x 100k over 8 GPUs, h=4, 8192 x 1000 bp/step, t=1 exact" + " padding" * 10, "workload_key": "c3", "rows": 10000000,
                       "cols_per_gpu": 12500, "total_cols": 100000, "index_gb_per_gpu": 15.68, "hashes": 4, "batch": 8192, "qlen": 1000, "unique_kmers_per_batch": 7946240,
                       "hits_first_batch": 64, "value_inputs": "host", "resident_lookups_per_s": 135522511.9, "host_visible": {"stream": {"kmer_lookups_per_s": 132123456.7}, "stream_scored": {"kmer_lookups_per_s": 1.2e8},
                                                                 "one_call_us": {"single_query": 57.123, "whole_batch_of_1000": 69.2}},
                       "verified": "planted hits on 8 shard(s) + 4 queries == oracle (colours, counts) on EVERY shard; 264 scored dicts == oracle",
                       "parallelism": "column-shard x8 + ncclAllGather of 1 bit/sample (library-owned RCCL communicator)", "exchange": "rccl", "rccl_ranks": 8, "exchange_ms": 0.0712345,
                       "ranks": {"ranks": [{"pci_bus_id": "0000:%02x:00" % (5 + 16 * i)} for i in range(8)], "distinct_gpus": True, "peer_access_from_rank0": True},
                       "per_rank_GBps": [6251.123456] * 8, "presence": {"in_timed_region": {"hits_scored": 260075, "host_us_per_hit": 3.9123}},
                       "clocks": {"after_timed_region": {"sclk_mhz": 2400.0, "mclk_mhz": 2000.0, "power_w": 712.0}}, "index_fill_s": 0.0478,
                       "calibration": {"sorted_GBps": 6365.1234, "random_GBps": 5898.1234}, "also": legs},
            "roofline": {"achieved": 6827.333765473314, "peak": 8000.0, "frac": 0.8534167206841643, "traffic": 6235709691.961538,
                         "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per the guide" * 2,
                         "kernel": "k_and_exact", "kernel_ms": 0.9096327185630798, "alg_bytes_per_launch": 6210366173.625, "launches_per_step": 64.0, "launches_timed": 1280,
                         "step_frac": 0.8473385385949821, "frac_overlapped": None, "kernel_ms_overlapped": None, "concurrent_launches": 1, "read_launches_repeated": 0,
                         "frac_of_box": 1.0726, "kmerize_ms": 0.2814119979739189, "compact_ms": 0.13012100011110306},
            "cpu_baseline": {"value": 62890.83504343365, "unit": "kmer_lookups/s", "cores": 1, "kind": "port", "sample_short": "s" * 150,
                             "pool": {"value": 431234.5, "cores": 128}, "word_parallel_pool": {"value": 1551234.5, "cores": 128}, "oracle_port": {"value": 85123.4}}}
    line = bench.condense(full)
    text = json.dumps(line)
    assert len(text) <= 7500, len(text)
    assert line["metric"] == "kmer_lookups_per_s" and line["n_gpus"] == 8 and line["roofline"]["bound"] == "hbm" and line["roofline"]["unit"] == "GB/s"
    assert line["roofline"]["frac"] == pytest.approx(0.85342, abs=1e-4) and line["roofline"]["traffic_ratio"] == pytest.approx(1.0041, abs=1e-4)
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["best_cpu_cores"] == 128
    # `value` is the host-visible rate (value_inputs "host"): the resident figure stands beside it, the one-call stream figure is not repeated
    assert line["config"]["value_inputs"] == "host" and line["config"]["resident_lookups_per_s"] == pytest.approx(1.3552e8, rel=1e-3)
    assert "host_visible_lookups_per_s" not in line["config"] and line["config"]["rccl_ranks"] == 8
    assert line["rccl_ranks"] == 8 and len(line["per_rank_GBps"]) == 8 and line["per_rank_GBps"][0] == pytest.approx(6251, rel=1e-3)      # top level at N > 1
    assert all(l["in"] == "h" and l["rv"] > 0 for l in line["config"]["also"].values())
    full["config"]["value_inputs"] = "resident"          # (score=True workloads, batches of short reads: resident steps, the stream figure beside them)
    assert bench.condense(full)["config"]["host_visible_lookups_per_s"] == pytest.approx(1.3212e8, rel=1e-3)
    assert set(line["config"]["also"]) == set(legs) and all(l["ok"] == 1 and l["ranks"] == 8 for l in line["config"]["also"].values())

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(t) for t in strings(line)) < 120
    # every world size the driver uses has its legs; 8 GPUs run the configurations that need 8 GPUs as WHOLE indexes
    assert [k for k, *_ in bench.also_legs_for(8)] == ["c4", "c5", "northstar", "c3_t04"] and [k for k, *_ in bench.also_legs_for(4)][0] == "northstar"
    assert all("--shard-of" not in extra for n in (2, 4, 8) for _, _, extra, _ in bench.also_legs_for(n))
    assert [k for k, *_ in bench.also_legs_for(1)][-1] == "ingest" and len(bench.also_legs_for(1)) == 11 and {"c5_dense", "c2_dense", "c5_ee", "c3_ee"} <= {k for k, *_ in bench.also_legs_for(1)} and bench.also_legs_for(3)[0][0] == "c3_t04"


def test_cortex_reader_vs_reference(tmp_path):
    """bigsi_amd.cortex against the reference's own reader on its three .ctx files (tests/golden/g10_cortex.json)."""
    import base64
    from bigsi_amd.cortex import CortexFormatError, extract_kmers_from_ctx, read_kmers
    for case in load_golden("g10_cortex.json"):
        p = tmp_path / os.path.basename(case["file"])
        p.write_bytes(base64.b64decode(case["ctx_base64"]))
        ksz, kmers = read_kmers(str(p))
        assert ksz == case["kmer_size"] and len(kmers) == case["num_records"]
        assert list(extract_kmers_from_ctx(str(p), 31)) == case["kmers_k31"]
        assert list(extract_kmers_from_ctx(str(p), 21)) == case["kmers_k21"]
    bad = tmp_path / "bad.ctx"
    bad.write_bytes(b"CORTEX\x05\x00\x00\x00")
    with pytest.raises(CortexFormatError):
        read_kmers(str(bad))


def test_cli_build_inputs_and_sizes(tmp_path):
    """Host logic of the `build` command line (bigsi/__main__.py:139-160): --from_file XOR -b, sample names default to the
    filter paths; max_build_mem_bytes sizes as humanfriendly reads them."""
    import argparse
    from bigsi_amd.__main__ import build_inputs, parse_size
    tsv = tmp_path / "f.tsv"
    tsv.write_text("a.bloom\tsample a\nb.bloom\tb\n")
    ns = argparse.Namespace(bloomfilters=[], samples=[], from_file=str(tsv))
    assert build_inputs(ns) == (["a.bloom", "b.bloom"], ["sample a", "b"])
    ns = argparse.Namespace(bloomfilters=["x", "y"], samples=[], from_file=None)
    assert build_inputs(ns) == (["x", "y"], ["x", "y"])
    with pytest.raises(ValueError):
        build_inputs(argparse.Namespace(bloomfilters=["x"], samples=[], from_file=str(tsv)))
    with pytest.raises(AssertionError):
        build_inputs(argparse.Namespace(bloomfilters=["x", "y"], samples=["only one"], from_file=None))
    assert [parse_size(t) for t in ("4GB", "512 MiB", 1000, "100", "2kb", "10 bytes", "1.5 GiB")] == \
        [4_000_000_000, 536_870_912, 1000, 100, 2000, 10, 1_610_612_736]
    with pytest.raises(ValueError):
        parse_size("lots")


def test_plan_shards_and_bench_workloads():
    """Column-range plan of a sharded index and the bench's named workloads: every BASELINE config is there, the ones that
    do not fit one GPU say so before touching a device."""
    import subprocess
    import sys
    from bigsi_amd.parallel import plan_shards
    assert plan_shards(129, 2) == (65, [(0, 65), (65, 64)])
    assert plan_shards(1000, 3) == (334, [(0, 334), (334, 334), (668, 332)])
    assert plan_shards(5, 8)[1][5:] == [(5, 0), (5, 0), (5, 0)]
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.WORKLOADS) == {"c2", "c3", "c4", "c5", "northstar"}
    assert bench.WORKLOADS["c4"]["rows"] * bench.WORKLOADS["c4"]["cols"] == 25_000_000 * 500_000 and bench.WORKLOADS["c5"]["threshold"] == 0.4
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "northstar", "--gpus", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "more than one MI355X holds" in r.stderr


# --------------------------------------------------------------------------------------------- K6's arithmetic, pinned on the CPU
# bigsi_amd/csrc/bigsi_score.hpp is the text the device compiles for K6 (k_score_packed); here the same header is compiled as host
# C++ (tests/c_host/score_host.cpp, contraction off) and pinned to CPython's round(), to the reference's golden scores (G5) and to
# the scalar restatement of scoring/score.py.  The -m gpu suite runs the same cases through the device build.
def build_score_host(tmp_path):
    import ctypes
    import subprocess
    so = str(tmp_path / "libscore_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
                           "-o", so, os.path.join(ROOT, "tests", "c_host", "score_host.cpp")])
    return ctypes.CDLL(so)


def score_strings_on_host(lib, strings, found=None, unique=None):
    import ctypes
    from bigsi_amd.scoring import HIT_SCORE_DTYPE, pack_presence
    bits, off, lens = pack_presence(strings)
    rec = np.zeros(len(strings), HIT_SCORE_DTYPE)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    lib.score_host_packed(p(bits), p(off), p(lens), p(found), p(unique), ctypes.c_uint64(len(strings)), p(rec))
    return rec


def adversarial_round_inputs():
    rng = np.random.default_rng(0)
    return np.ascontiguousarray(np.concatenate([
        rng.uniform(-5000, 5000, 100000), rng.integers(-500000, 500000, 100000) / 1000.0, rng.integers(-500000, 500000, 100000) / 200.0,
        (rng.integers(-5000000, 5000000, 100000) + 0.5) / 100.0,        # x.xx5: the decimal half-way cases, none exactly representable
        rng.integers(-40000, 40000, 50000) / 8.0,                       # exact binary ties (x.125, x.375, ...): half to even
        np.array([2.675, 0.125, 0.375, -0.125, 1.005, 1e-9, -1e-9, 0.0, -0.0, 0.005, 0.015, 0.025, 1064.885, 96.045, 1e12 + 0.005])]),
        dtype=np.float64)


def score_strings():
    import random
    rng = random.Random(5)
    strings = ["0", "1", "00", "11", "10", "01", "111", "110", "011", "101", "000", "0000", "1111", "1101", "1011"]
    for n in (3, 4, 5, 7, 8, 31, 61, 63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193, 970, 3970):
        for p in (0.0, 0.05, 0.5, 0.9, 0.97, 1.0):
            for _ in range(4):
                a = ["1" if rng.random() < p else "0" for _ in range(n)]
                for _ in range(rng.randrange(4)):          # gaps a SNP would leave: ~31 absent k-mers in a row
                    g0, ln = rng.randrange(n), rng.randrange(1, 70)
                    a[g0:g0 + ln] = ["0"] * len(a[g0:g0 + ln])
                strings.append("".join(a)[:n])
    return strings


def test_k6_round_is_cpythons_round(tmp_path):
    import ctypes
    lib = build_score_host(tmp_path)
    xs = adversarial_round_inputs()
    out = np.zeros_like(xs)
    lib.score_host_round2(xs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(xs.size), out.ctypes.data_as(ctypes.c_void_p))
    want = np.array([round(float(x), 2) for x in xs])
    assert np.array_equal(out, want) and np.array_equal(np.signbit(out), np.signbit(want))
    assert (np.round(xs, 2) != want).any()          # (the obvious x*100 -> rint -> /100 is NOT Python's round)


def check_records_against_golden_and_scalar(score_fn):
    """score_fn(strings, found, unique) -> HIT_SCORE_DTYPE records.  All 438 cases of G5 (the reference's own Scorer.score outputs,
    incl. its known answer bigsi/tests/scoring.py:10-31) and seeded strings against the scalar restatement, every field exact."""
    from bigsi_amd.scoring import SCORE_KEYS, Scorer, score_columns, unpack_presence, pack_presence
    g = load_golden("g5_scoring.json")
    checked = 0
    for db in sorted({r["db_size"] for r in g["cases"]}):
        good = [r for r in g["cases"] if r["db_size"] == db and "raises" not in r]
        cols = score_columns(score_fn([r["s"] for r in good], None, None), db)
        for i, r in enumerate(good):
            assert_result_equal(dict(zip(SCORE_KEYS, [c[i] for c in cols])), unjson(r["score"]), "db=%d %s" % (db, r["s"][:20]))
            checked += 1
    assert checked == 438
    strings = score_strings()
    rng = np.random.default_rng(3)
    unique = rng.integers(1, 4000, len(strings)).astype(np.uint32)
    found = (rng.random(len(strings)) * (unique + 1)).astype(np.uint32).clip(0, unique)
    rec = score_fn(strings, found, unique)
    assert rec["percent_kmers_found"].tolist() == [round(100 * float(f) / u, 2) for f, u in zip(found.tolist(), unique.tolist())]
    assert rec["num_kmers"].tolist() == [len(s) for s in strings]
    for db in (0, 1, 500000):
        sc, cols = Scorer(db), score_columns(rec, db)
        for i, s in enumerate(strings):
            want, got = sc.score(s), dict(zip(SCORE_KEYS, [c[i] for c in cols]))
            assert list(got) == list(want)
            for k in want:
                assert (got[k] == want[k] or (got[k] != got[k] and want[k] != want[k])) and math.copysign(1, got[k]) == math.copysign(1, want[k]), (db, s[:30], len(s), k, got[k], want[k])
    bits, off, _ = pack_presence(strings)
    text = unpack_presence(bits, off)
    assert all(text[8 * int(off[i]):8 * int(off[i]) + len(s)] == s for i, s in enumerate(strings))
    with pytest.raises(ZeroDivisionError):
        score_columns(score_fn([""], None, None), 4)


def test_k6_scoring_header_vs_golden_scores_and_scalar_scorer(tmp_path):
    lib = build_score_host(tmp_path)
    check_records_against_golden_and_scalar(lambda strings, found, unique: score_strings_on_host(lib, strings, found, unique))


def test_search_stream_batches_stay_bounded_when_sequences_grow_along_the_stream():
    """search_stream sizes its slices by the sequences seen so far; reads followed by genome-length sequences used to give a
    batch of ~1e9 k-mers (tens of GB of row ids).  Every batch must stay within batch_kmers + one sequence, order preserved."""
    from bigsi_amd.graph.bigsi import BIGSI
    sizes = []

    def submit(chunk, slot):
        sizes.append(sum(max(len(s) - 30, 1) for s in chunk))
        return None, chunk

    seqs = ["A" * 61] * 40000 + ["C" * 100000] * 300 + ["G" * 61] * 1000 + ["T" * 5000] * 300
    out = list(BIGSI._stream_loop(iter(seqs), 64, None, 1 << 19, 31, submit, lambda p: [[] for _ in p[1]]))
    assert [s for s, _ in out] == seqs
    assert max(sizes) <= (1 << 19) + 100000 and len(sizes) < 200
    sizes.clear()
    out = list(BIGSI._stream_loop(iter(seqs[:1000]), 10, 10, 1 << 19, 31, submit, lambda p: [[] for _ in p[1]]))     # explicit batch_size: untouched
    assert len(sizes) == 100 and [s for s, _ in out] == seqs[:1000]


def test_bench_failure_leaves_a_json_line():
    """A run that cannot go ahead (here: a workload that does not fit one GPU, refused before any device is touched) exits non-zero AND
    prints the one JSON line with value null, rc 1 and the error -- the driver's record then says why."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4", "--gpus", "1"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "more than one MI355X holds" in r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["value"] is None and d["rc"] == 1 and d["n_gpus"] == 1 and "more than one MI355X holds" in d["error"] and d["where"]


def _library_kernels():
    """short names (template arguments kept, parameter lists dropped) of the device kernels libbigsi_hip.so holds: the mangled names
    sit in the embedded code object."""
    import re
    import subprocess
    from bigsi_amd import _lib
    raw = open(_lib.LIB_PATH, "rb").read()
    mangled = sorted({m.group(0).decode() for m in re.finditer(rb"_ZN5bigsi[A-Za-z0-9_]+", raw)})
    filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt):
        import shutil
        filt = shutil.which("c++filt")
    if not filt:
        pytest.skip("no demangler on this machine")
    out = subprocess.run([filt], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
    return {l.split("(")[0].replace("void ", "").strip() for l in out if "bigsi::" in l}


def test_profiles_of_this_round_describe_the_built_library():
    """Every kernel a rocprofv3 summary of THIS round names (profiles/r06_*_kernel_stats.csv) is a kernel of the library the tree
    builds -- a summary made before a kernel changed its template parameters names one that no longer exists -- and the round's stamp
    (profiles/r06_stamp.json: commit and library hash the profiles were made with) is present."""
    import csv
    import glob
    from bigsi_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libbigsi_hip.so not built")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*_kernel_stats.csv")))
    if not files:
        pytest.skip("no round-6 profiles yet")
    have = _library_kernels()
    assert len(have) > 30
    norm = lambda s_: s_.replace(" ", "")      # noqa: E731 -- (demanglers differ in spacing)
    have_n = {norm(h) for h in have}
    missing = {}
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                name = r["Name"].split("(")[0].replace("void ", "").strip()
                if name.startswith("bigsi::") and norm(name) not in have_n:
                    missing.setdefault(os.path.basename(fn), []).append(name)
    assert not missing, "profiles name kernels the built library does not hold: %r" % missing
    import json
    st = json.load(open(os.path.join(ROOT, "profiles", "r06_stamp.json")))
    assert len(st.get("git_head", "")) == 40 and len(st.get("libbigsi_hip_so_sha256", "")) == 64


def test_stream_tuning_is_shared_between_streams_and_restored_by_the_last_one():
    """search_stream's two interpreter-wide settings (short switch interval, automatic GC paused) belong to all running streams together:
    interleaved entries and exits from two streams restore the ORIGINAL values, whatever the order (a per-stream save / restore left the
    process at the short interval, or with the collector off, for good)."""
    import gc
    import sys
    from bigsi_amd.graph.bigsi import _StreamTuning as T
    i0, g0 = sys.getswitchinterval(), gc.isenabled()
    try:
        gc.enable()
        sys.setswitchinterval(0.005)
        a = T.enter(True)                        # stream A
        assert a and not gc.isenabled() and sys.getswitchinterval() <= 2.1e-4
        b = T.enter(True)                        # stream B, nested / in another thread
        assert b and not gc.isenabled()
        T.leave(True)                            # A ends first: B still runs
        assert not gc.isenabled() and sys.getswitchinterval() <= 2.1e-4
        T.leave(True)                            # the last one out restores
        assert gc.isenabled() and abs(sys.getswitchinterval() - 0.005) < 1e-9
        c = T.enter(False)                       # a stream that leaves the collector alone
        assert not c and gc.isenabled() and sys.getswitchinterval() <= 2.1e-4
        d = T.enter(True)
        T.leave(False)
        assert not gc.isenabled()
        T.leave(True)
        assert gc.isenabled() and abs(sys.getswitchinterval() - 0.005) < 1e-9 and T.users == 0 and T.gc_users == 0
    finally:
        sys.setswitchinterval(i0)
        (gc.enable if g0 else gc.disable)()


def test_pack_rows_extension_is_the_python_loop():
    """_results.pack_rows (row blocks of migrate_index packed by threads below Python) == the loop it replaces: rows cut or
    zero-extended to rb bytes, the last byte masked."""
    from bigsi_amd import migrate
    if migrate._ext is None:
        pytest.skip("bigsi_amd/_results extension not built")
    rng = np.random.default_rng(3)
    for n, rb, mask in ((1, 1, 0xFF), (7, 13, 0xF0), (300, 62500, 0x80), (2000, 1251, 0xFE)):
        raws = [rng.integers(0, 256, size=int(rng.choice([0, 1, rb - 1, rb, rb + 5, max(rb // 2, 1)])), dtype=np.uint8).tobytes() for _ in range(n)]
        raws[0] = bytearray(raws[0])
        want = np.zeros((n, rb), np.uint8)
        for j, raw in enumerate(raws):
            a = np.frombuffer(bytes(raw), np.uint8)[:rb]
            want[j, : a.size] = a
        want[:, rb - 1] &= mask
        got = np.full((n, rb), 0xAA, np.uint8)
        migrate._ext.pack_rows(raws, got, mask)
        assert np.array_equal(got, want), (n, rb)
    with pytest.raises(TypeError):
        migrate._ext.pack_rows(["not bytes"], np.zeros((1, 4), np.uint8), 0xFF)
    with pytest.raises(ValueError):
        migrate._ext.pack_rows([b"x", b"y"], np.zeros((1, 4), np.uint8), 0xFF)


def test_dense_bench_plants_and_the_oracles_bulk_bookkeeping():
    """bench.py --dense: a sample's stretches of a query <-> the k-mer positions planted (seg_masks), and the oracle's bulk
    bookkeeping of such plants (SynthOracle.insert_kmer_masks, restricted to the rows a check reads) == insert_kmers per stretch."""
    import bench
    from oracle.ref_model import SynthOracle
    k, qlen = 31, 400
    w = {"batch": 20, "qlen": qlen}
    plants = list(bench.dense_gene_plants(w, 2, 0, 5000, k))
    assert len(plants) == 2 * 16 and all(len(c) == 50 and len(np.unique(c)) == 50 for _, _, c, _ in plants)
    rng = np.random.default_rng(4)
    seq = "".join(rng.choice(list("ACGT"), size=qlen))
    _, _, cols, segs = plants[3]
    masks = bench.seg_masks(segs, qlen - k + 1, k)
    for j, sg in enumerate(segs):
        inside = np.zeros(qlen - k + 1, bool)
        for a, b in sg:
            assert b - a >= k
            inside[a:b - k + 1] = True
        assert np.array_equal(masks[j], inside)
    assert 0.3 < masks.mean() <= 1.0
    a, b = SynthOracle(1, 0, 100003, 5000, 3, k, 2), SynthOracle(1, 0, 100003, 5000, 3, k, 2)
    for c, sg in zip(cols.tolist(), segs):
        for x, y in sg:
            a.insert_kmers(c, seq[x:y])
    b.insert_kmer_masks(b.rows_of(seq), cols, masks)
    assert a.planted == b.planted
    only = np.unique(b.rows_of(seq)[::7].ravel())
    c_ = SynthOracle(1, 0, 100003, 5000, 3, k, 2)
    c_.insert_kmer_masks(c_.rows_of(seq), cols, masks, only)
    assert c_.planted == {r: v for r, v in a.planted.items() if r in set(only.tolist())}
    assert sorted(bench.dense_read_cols(1, 5, 0, 10000)) == sorted(set(bench.dense_read_cols(1, 5, 0, 10000))) and len(bench.dense_read_cols(1, 5, 0, 10000)) == 8
