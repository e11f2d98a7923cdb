"""HBM traffic of EVERY kernel a bench line quotes, from the PMC counters, on the GPU box:

    python scripts/pmc_all.py [OUTDIR]        # default gpurun_out/pmc

For each workload: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc WRITE_SIZE: one counter per pass, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes; --kernel-trace only, no other trace domain) around a short run of the SAME
command the bench leg uses, reduced to per-kernel averages per dispatch.  traffic = FETCH_SIZE x 2 (gfx950 tallies 128-byte
requests at 64, per the guide; calibrated on k_fill_synth, whose WRITE_SIZE must equal the index bytes) + WRITE_SIZE, both
KiB.  Writes OUTDIR/pmc_traffic.json (keyed by bench.py's workload key -> what bench.py reports as roofline.traffic) and
OUTDIR/pmc_kernels.json (every bigsi:: kernel of every pass, for the record).  The caller copies both into profiles/."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc"))
B = ["--cpu-seconds", "0", "--also", "none", "--host-visible", "0", "--no-verify", "--alone-steps", "0"]

# (tag, command after `python`, kernel-name substrings whose traffic the line of that workload reports)
WORKLOADS = [
    # (--timed resident: every launch of the pass has ONE shape -- whole batches --, so that the average over dispatches is the traffic of that
    #  shape; a streamed thresholded batch goes out in two launches beside the whole-batch launches of the resident leg.  bench.py scales
    #  the RATIO measured here to the launches of its own run)
    ("c3_exact", ["bench.py", "--steps", "2", "--warmup", "1", "--timed", "resident"] + B, ["k_and_exact"]),
    ("c3_t04", ["bench.py", "--steps", "2", "--warmup", "1", "--threshold", "0.4", "--timed", "resident"] + B, ["k_and_count"]),
    ("c2", ["bench.py", "--workload", "c2", "--steps", "64", "--warmup", "8"] + B, ["k_reads_fused"]),
    ("c2_t04", ["bench.py", "--workload", "c2", "--steps", "64", "--warmup", "8", "--threshold", "0.4"] + B, ["k_reads_fused"]),
    ("c4_shard", ["bench.py", "--workload", "c4", "--shard-of", "8", "--steps", "8", "--warmup", "2"] + B, ["k_and_exact"]),
    ("c5_shard", ["bench.py", "--workload", "c5", "--shard-of", "8", "--steps", "8", "--warmup", "2"] + B, ["k_and_count", "k_presence_bits", "k_presence_score"]),
    ("c5_dense", ["bench.py", "--workload", "c5", "--shard-of", "8", "--dense", "1", "--steps", "8", "--warmup", "2"] + B, ["k_and_count", "k_presence_bits", "k_presence_score"]),
    ("c5_ee", ["bench.py", "--workload", "c5", "--shard-of", "8", "--dense", "1", "--early-exit", "1", "--score", "0", "--steps", "8", "--warmup", "2"] + B, ["k_and_count"]),
    ("c3_ee", ["bench.py", "--steps", "8", "--warmup", "2", "--early-exit", "1", "--timed", "resident"] + B, ["k_and_exact"]),
    ("c2_dense", ["bench.py", "--workload", "c2", "--dense", "1", "--steps", "64", "--warmup", "8"] + B, ["k_reads_fused"]),
    ("ns_shard", ["bench.py", "--workload", "northstar", "--shard-of", "8", "--steps", "8", "--warmup", "2"] + B, ["k_and_exact"]),
    ("ns_shard_t04", ["bench.py", "--workload", "northstar", "--shard-of", "8", "--steps", "8", "--warmup", "2", "--threshold", "0.4"] + B, ["k_and_count"]),
    ("transpose", ["scripts/measure.py", "transpose"], ["k_transpose_regs"]),
]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def reduce_counters(path):
    acc = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            a = acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: {c: {"dispatches": n, "avg": tot / n} for c, (n, tot) in v.items()} for k, v in acc.items()}


def one_pass(tag, counter, cmd):
    raw = os.path.join(OUT, "raw_%s_%s" % (tag, counter))
    details = os.path.join(OUT, "%s_details.json" % tag)
    full = [sys.executable] + cmd + (["--details", details] if cmd[0] == "bench.py" else [])
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(OUT, "%s_%s.stdout" % (tag, counter)), "w") as so, open(os.path.join(OUT, "%s_%s.stderr" % (tag, counter)), "w") as se:
        subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", raw, "-o", "p", "--"] + full,
                       cwd=ROOT, env=env, stdout=so, stderr=se, timeout=1200)
    files = glob.glob(os.path.join(raw, "**", "p_counter_collection.csv"), recursive=True)
    red = reduce_counters(files[0]) if files else {}
    shutil.rmtree(raw, ignore_errors=True)
    return red


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[2:])
    traffic, kernels = {}, {}
    for tag, cmd, wanted in WORKLOADS:
        if only and tag not in only:
            continue
        fe, wr = one_pass(tag, "FETCH_SIZE", cmd), one_pass(tag, "WRITE_SIZE", cmd)
        rec = {}
        for k in sorted(set(fe) | set(wr)):
            if not k.startswith("bigsi::"):
                continue
            f_, w_ = fe.get(k, {}).get("FETCH_SIZE", {}), wr.get(k, {}).get("WRITE_SIZE", {})
            rec[k] = {"FETCH_SIZE_avg_kib": f_.get("avg"), "WRITE_SIZE_avg_kib": w_.get("avg"), "dispatches": f_.get("dispatches") or w_.get("dispatches"),
                      "traffic_bytes_per_dispatch": (f_.get("avg") or 0.0) * 1024 * 2 + (w_.get("avg") or 0.0) * 1024}
        kernels[tag] = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python " + " ".join(cmd), "kernels": rec}
        details = os.path.join(OUT, "%s_details.json" % tag)
        if os.path.exists(details):
            d = json.load(open(details))
            key, alg = d["roofline"]["pmc_key"], d["roofline"]["alg_bytes_per_launch"]
            ent = {}
            for want in wanted:
                names = [k for k in rec if want in k]
                if not names:
                    continue
                k = max(names, key=lambda n_: rec[n_]["dispatches"] or 0)
                ent[want] = {"kernel": k, "traffic_bytes_per_launch": rec[k]["traffic_bytes_per_dispatch"], "dispatches": rec[k]["dispatches"]}
            if wanted[0] in ent:
                t = ent[wanted[0]]
                traffic[key] = {"kernel": t["kernel"], "traffic_bytes_per_launch": t["traffic_bytes_per_launch"], "alg_bytes_per_launch": alg,
                                "ratio": t["traffic_bytes_per_launch"] / alg, "dispatches": t["dispatches"], "workload": tag,
                                "other_kernels": {w_: v for w_, v in ent.items() if w_ != wanted[0]},
                                "presence_alg_bytes_per_call": (d["config"].get("presence") or {}).get("alg_bytes"),
                                "presence_line_floor_bytes_per_call": (d["config"].get("presence") or {}).get("line_floor_bytes"),
                                "source": "profiles/pmc_kernels.json [%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 per MI355X_MICROARCH.md" % tag}
        # calibration of the x2: k_fill_synth writes the whole index once
        fill = [k for k in rec if "k_fill_synth" in k]
        if fill and os.path.exists(details):
            idx_bytes = json.load(open(details))["config"]["index_gb_per_gpu"] * 1e9
            kernels[tag]["fill_synth_write_over_index_bytes"] = rec[fill[0]]["WRITE_SIZE_avg_kib"] * 1024 / idx_bytes if rec[fill[0]]["WRITE_SIZE_avg_kib"] else None
        print(tag, json.dumps({k: (round(v["traffic_bytes_per_dispatch"] / 1e6, 3), v["dispatches"]) for k, v in rec.items() if any(w_ in k for w_ in wanted)}), flush=True)
    with open(os.path.join(OUT, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    with open(os.path.join(OUT, "pmc_kernels.json"), "w") as f:
        json.dump(kernels, f, indent=1)
    for k, v in traffic.items():
        print("%-14s %-40s traffic %.4g B / alg %.4g B = %.4f" % (v["workload"], v["kernel"][:40], v["traffic_bytes_per_launch"], v["alg_bytes_per_launch"], v["ratio"]))


if __name__ == "__main__":
    main()
