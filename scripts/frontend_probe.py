"""Application-level throughput of the batch front-end on BASELINE configs[1]'s index (1M rows x 10k samples, h=3), reads of 61 bp:
Python strings / a FASTA file in -> the reference's result dicts / its JSON and CSV text out.  Two read sets: random reads (no
hits: the common bulk case) and reads that each match 8 samples (hits present: every record carries 8 result dicts).

    python scripts/frontend_probe.py [n_reads]         # one JSON object on stdout
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigsi_amd import BIGSI, frontend  # noqa: E402
from bigsi_amd.storage import get_storage  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m, n_cols, h, k = 1_000_000, 10_000, 3, 31
cfg = {"storage-engine": "hip-hbm", "storage-config": {"name": "frontend_probe", "max_cols": n_cols}, "k": k, "m": m, "h": h}
st = get_storage(cfg)
st.delete_all()
for key, v in (("number_of_rows", m), ("number_of_cols", n_cols), ("ksi:bloomfilter_size", m), ("ksi:num_hashes", h)):
    st.set_integer(key, v)
st.set_integer("metadata:colour_count", n_cols)
for c in range(n_cols):
    st.set_string("metadata:%d" % c, "sample%05d" % c)
st.fill_synthetic(1, 0, 2)
rng = np.random.default_rng(0)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
rand_reads = [lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(n_reads, 61), dtype=np.uint8)]
# 2000 distinct reads, each Bloom-added to 8 samples; the hit set repeats them
base = [lut[r].tobytes().decode() for r in rng.integers(0, 4, size=(2000, 61), dtype=np.uint8)]
for i, s in enumerate(base):
    for t in range(8):
        st.insert_kmers((i * 8 + t) % n_cols, [s], k)
hit_reads = [base[i % 2000] for i in range(n_reads)]
b = BIGSI(cfg)
out = {"reads": n_reads, "index": "1M x 10k, h=3", "kmers_per_read": 31}


def fasta_of(reads):
    fn = tempfile.mktemp(suffix=".fa")
    with open(fn, "w") as f:
        f.write("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    return fn


for tag, reads in (("no_hits", rand_reads), ("8_hits_per_read", hit_reads)):
    rec = {}
    for thr in (1.0, 0.4):
        list(b.search_stream(reads[:20000], thr))          # warm
        t0 = time.perf_counter()
        n_res = sum(len(r) for _, r in b.search_stream(reads, thr))
        dt = time.perf_counter() - t0
        rec["search_stream_dicts_t%g" % thr] = {"reads_per_s": n_reads / dt, "lookups_per_s": 31 * n_reads / dt, "results": n_res}
    t0 = time.perf_counter()
    nk, nu, off, col, cnt = st.search_many(reads, k, 1.0)
    rec["search_many_arrays"] = {"reads_per_s": n_reads / (time.perf_counter() - t0), "hits": int(off[-1])}
    if tag != "no_hits":
        # score=True on reads (K5 + K6 inside the stream): 200 k reads with 8 hits each, arrays out
        sub = reads[:200_000]
        st.search_many_scored(sub[:20000], k, 1.0)
        t0 = time.perf_counter()
        r_ = st.search_many_scored(sub, k, 1.0)
        dt = time.perf_counter() - t0
        rec["search_many_scored_arrays"] = {"reads_per_s": len(sub) / dt, "scored_hits_per_s": int(r_[2][-1]) / dt}
    fn = fasta_of(reads)
    try:
        for fmt in ("json", "csv"):
            t0 = time.perf_counter()
            text = frontend.bulk_search(b, fn, 1.0, False, fmt)
            dt = time.perf_counter() - t0
            rec["bulk_search_%s" % fmt] = {"reads_per_s": n_reads / dt, "text_MB": len(text) / 1e6, "text_MBps": len(text) / 1e6 / dt}
        t0 = time.perf_counter()
        n_fa = len(frontend.read_fasta(fn))
        rec["read_fasta_reads_per_s"] = n_fa / (time.perf_counter() - t0)
        # the stages of the native text route (frontend._bulk_text_native), each on its own
        from bigsi_amd import _lib
        stages = {}
        t0 = time.perf_counter()
        with open(fn, "rb") as f:
            data = f.read()
        stages["file_read_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        blob_, soff_ = _lib.fasta_pack(data)
        stages["fasta_pack_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        nk_, nu_, off_, col_, cnt_ = st.search_many_packed(blob_, soff_, k, 1.0)
        stages["search_stream_ms"] = (time.perf_counter() - t0) * 1e3
        name_off_ = np.arange(n_cols + 1, dtype=np.uint64) * 11
        names_ = b"".join(b"sample%05d" % c for c in range(n_cols)) + b"\0"
        for fmt_, key in ((0, "format_json_ms"), (1, "format_csv_ms")):
            t0 = time.perf_counter()
            text = _lib.format_results(fmt_, blob_, soff_, 1.0, json.dumps(frontend.CITATION), nu_, off_, col_, cnt_, names_, name_off_, np.zeros(n_cols, np.uint8))
            stages[key] = (time.perf_counter() - t0) * 1e3
        rec["native_route_stages"] = stages
        if tag != "no_hits":
            # score=True as text: 200 k reads with 8 scored hits each through the native route; the per-record route on 2000 for scale
            sfn = fasta_of(reads[:200_000])
            for fmt in ("json", "csv"):
                t0 = time.perf_counter()
                text = frontend.bulk_search(b, sfn, 1.0, True, fmt)
                dt = time.perf_counter() - t0
                rec["bulk_search_scored_%s" % fmt] = {"reads_per_s": 200_000 / dt, "scored_hits_per_s": 1_600_000 / dt, "text_MB": len(text) / 1e6, "text_MBps": len(text) / 1e6 / dt}
            os.remove(sfn)
            sfn = fasta_of(reads[:2000])
            native = frontend._bulk_text_native
            frontend._bulk_text_native = lambda *a: None
            t0 = time.perf_counter()
            slow = frontend.bulk_search(b, sfn, 1.0, True, "json")
            rec["bulk_search_scored_json_per_record_route"] = {"reads_per_s": 2000 / (time.perf_counter() - t0)}
            frontend._bulk_text_native = native
            assert frontend.bulk_search(b, sfn, 1.0, True, "json") == slow
            os.remove(sfn)
        # the text equals the reference's (json.dumps of the record list, indent=4) on a sample of the file
        small = fasta_of(reads[:300])
        want = json.dumps([frontend.search_record(s, 1.0, r) for s, r in zip(reads[:300], b.search_batch(reads[:300], 1.0))], indent=4)
        assert frontend.bulk_search(b, small, 1.0) == want
        os.remove(small)
    finally:
        os.remove(fn)
    out[tag] = rec
b.delete()
print(json.dumps(out))
