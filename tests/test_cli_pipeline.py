"""The reference's command-line workflow on this package, one process per command as a user would run it:
    bloom CTX OUT  ->  build -b ... -s ...  ->  search / bulk_search
The index persists between processes through the backend's snapshot file (`storage-config: filename`).  Expected output:
goldens of the reference's own Bloom filter (G4) and front-end text (G9)."""
import base64
import json
import os
import subprocess
import sys

import pytest
import yaml

from conftest import ROOT, load_golden


def nl(text):
    """subprocess's text mode folds the csv writer's '\\r\\n' into '\\n'; the byte-exact text is pinned in test_frontend.py"""
    return text.replace("\r\n", "\n").replace("\r", "\n")


def cli(args, cwd):
    r = subprocess.run([sys.executable, "-m", "bigsi_amd"] + args, cwd=cwd, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_bloom_build_search_commands(tmp_path):
    g4, g9, g10 = load_golden("g4_config1.json"), load_golden("g9_frontend.json"), load_golden("g10_cortex.json")
    cfg = {"storage-engine": "hip-hbm", "k": 31, "m": 1000, "h": 3,
           "storage-config": {"name": "cli", "filename": str(tmp_path / "index.hbm")}}
    cf = tmp_path / "config.yaml"
    cf.write_text(yaml.safe_dump(cfg))
    ctx = tmp_path / "test_kmers.ctx"
    ctx.write_bytes(base64.b64decode(g10[0]["ctx_base64"]))
    # bloom from the Cortex graph == the reference's filter of the same 100 k-mers (G4 sample s1)
    cli(["bloom", str(ctx), str(tmp_path / "s1.bloom"), "--config", str(cf)], str(tmp_path))
    assert (tmp_path / "s1.bloom").read_bytes().hex() == g4["blooms"][0]
    # the other two G9 samples from k-mer list files
    names = list(g9["samples"].keys())
    blooms = []
    for i, nme in enumerate(names):
        kf = tmp_path / ("s%d.kmers" % i)
        kf.write_text("\n".join(g9["samples"][nme]) + "\n")
        cli(["bloom", str(kf), str(tmp_path / ("s%d.bloom" % i)), "--config", str(cf)], str(tmp_path))
        blooms.append(str(tmp_path / ("s%d.bloom" % i)))
    args = ["build", "--config", str(cf)]
    for b, nme in zip(blooms, names):
        args += ["-b", b, "-s", nme]
    assert "success" in cli(args, str(tmp_path))
    assert os.path.exists(cfg["storage-config"]["filename"])
    fasta = tmp_path / "tests.fasta"
    fasta.write_text(g9["fasta_text"]["tests"])
    n = 0
    for c in g9["cases"]:
        if c["score"]:
            continue
        if c["cmd"] == "search":
            out = cli(["search", c["seq"], "--threshold", str(c["threshold"]), "--format", c["format"], "--config", str(cf)], str(tmp_path))
            assert out == nl(c["out"] + "\n"), (c["seq"][:20], c["threshold"], c["format"])
            n += 1
        elif c["fasta"] == "tests" and not c["stream"]:
            out = cli(["bulk_search", str(fasta), "--threshold", str(c["threshold"]), "--format", c["format"], "--config", str(cf)], str(tmp_path))
            assert out == nl(c["out"] + "\n")
            n += 1
    assert n >= 12
    cli(["delete", "--config", str(cf)], str(tmp_path))
