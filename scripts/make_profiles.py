"""Condense gpurun_out/prof (scripts/profile_all.sh) into profiles/<round>_* (ROUND, default r03): per workload one JSON (the command's own line(s),
the rocprofv3 average duration of the dominant kernel from the SAME run, algorithmic bytes, fraction of the 8 TB/s HBM peak)
plus the kernel-stats CSV it came from; and profiles/pmc_traffic.json from the PMC passes.

    python scripts/make_profiles.py                      # after a profile_all.sh run came back
    python scripts/make_profiles.py --pmc-reduce IN OUT  # on the GPU box: counter_collection.csv -> per-kernel averages"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
PEAK = 8000.0
R = os.environ.get("ROUND", "r06")      # prefix of this round's files


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def pmc_reduce(src, dst):
    """rocprofv3 counter_collection.csv (one row per dispatch and counter) -> {kernel: {counter: {dispatches, avg}}}."""
    acc = {}
    with open(src) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    out = {k: {c: {"dispatches": n, "avg": tot / n} for c, (n, tot) in v.items()} for k, v in acc.items()}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)


def kernel_stats(tag):
    path = os.path.join(SRC, tag + "_kernel_stats.csv")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {short(r["Name"]): {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
                for r in csv.DictReader(f)}


def json_lines(tag):
    path = os.path.join(SRC, tag + ".stdout")
    if not os.path.exists(path):
        return []
    return [json.loads(l) for l in open(path) if l.startswith("{")]


def stamp():
    """What the files of this round describe: the commit the snapshot was sent from and the library that ran (the .so travels with the
    snapshot: the file here is the file there), plus the kernels that library holds -- tests/test_abi_and_host.py holds every kernel
    name in profiles/<round>_*_kernel_stats.csv against the library built from the tree."""
    import hashlib
    import subprocess
    so = os.path.join(ROOT, "bigsi_amd", "libbigsi_hip.so")
    out = {"round": R}
    try:
        out["git_head"] = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        out["git_dirty"] = bool(subprocess.run(["git", "status", "--porcelain", "--", "bigsi_amd/csrc", "bench.py"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
    except OSError:
        pass
    try:
        out["libbigsi_hip_so_sha256"] = hashlib.sha256(open(so, "rb").read()).hexdigest()
    except OSError:
        pass
    import glob
    for ext in sorted(glob.glob(os.path.join(ROOT, "bigsi_amd", "_results*.so"))):       # the result-dict builder (host code: the dict rates)
        out["results_ext_sha256"] = hashlib.sha256(open(ext, "rb").read()).hexdigest()
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    summary = {}
    st = stamp()
    try:          # the GPU box's own record of the library it ran (scripts/profile_all.sh)
        ran = json.load(open(os.path.join(SRC, R + "_build.json")))
        st["ran_so_sha256"] = ran.get("so_sha256")
        assert ran.get("so_sha256") in (None, st.get("libbigsi_hip_so_sha256")), "the profiles were made with another build of libbigsi_hip.so than the one in the tree"
        assert ran.get("results_ext_sha256") in (None, st.get("results_ext_sha256")), "the profiles were made with another build of bigsi_amd/_results than the one in the tree"
    except OSError:
        pass
    with open(os.path.join(DST, R + "_stamp.json"), "w") as f:
        json.dump(st, f, indent=1)
    for path in sorted(glob.glob(os.path.join(SRC, R + "_*_kernel_stats.csv"))):
        tag = os.path.basename(path)[: -len("_kernel_stats.csv")]
        ks, lines = kernel_stats(tag), json_lines(tag)
        rec = {"command": "scripts/profile_all.sh: " + tag, "git_head": st.get("git_head"), "libbigsi_hip_so_sha256": st.get("libbigsi_hip_so_sha256"),
               "kernels": {k: v for k, v in ks.items() if k.startswith("bigsi::")}}
        if lines and "roofline" in lines[-1]:              # a bench.py line
            d = lines[-1]
            r = d["roofline"]
            want = "k_reads_fused" if "k_reads_fused" in r["kernel"] else r["kernel"]
            names = [k for k in ks if want in k] or [k for k in ks if "k_reads_fused" in k]
            dom = max(names, key=lambda k: ks[k]["calls"] * ks[k]["avg_ns"]) if names else None
            rec["bench_same_run"] = d
            if dom:
                ns = ks[dom]["avg_ns"]
                rec["dominant_kernel"] = {"name": dom, "rocprof_avg_ns": ns, "calls": ks[dom]["calls"], "hip_event_avg_ns": r["kernel_ms"] * 1e6,
                                          "alg_bytes_per_launch": r["alg_bytes_per_launch"], "GBps_rocprof": r["alg_bytes_per_launch"] / ns,
                                          "frac_of_8TBps": r["alg_bytes_per_launch"] / ns / PEAK}
                summary[tag] = {"workload": d["config"]["workload"], "value_lookups_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                                "kernel": dom, "kernel_ns": ns, "GBps": r["alg_bytes_per_launch"] / ns, "frac": r["alg_bytes_per_launch"] / ns / PEAK,
                                "k1_ms": r["kmerize_ms"], "k4_ms": r["compact_ms"],
                                "step_GBps": r.get("step_GBps"), "step_frac": r.get("step_frac"), "concurrent_launches": r.get("concurrent_launches", 1)}
        else:
            rec["measure_lines"] = lines
            summary[tag] = {"lines": [{k: v for k, v in l.items() if k in ("workload", "threshold", "n_seqs", "hits", "m", "cols", "GBps", "frac",
                                                                          "k2_ms", "kernels_ms", "step_ms", "lookups_per_s")} for l in lines]}
        with open(os.path.join(DST, tag + ".json"), "w") as f:
            json.dump(rec, f, indent=1)
        shutil.copy(path, os.path.join(DST, tag + "_kernel_stats.csv"))
    for name in (R + "_bench_default", R + "_bench_t04"):
        lines = json_lines(name)
        if lines:
            with open(os.path.join(DST, name + ".json"), "w") as f:
                json.dump(lines[-1], f)
    # PMC: traffic per launch of the row-AND kernel = FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64, MI355X_MICROARCH.md) + WRITE_SIZE,
    # both reported in KiB; calibration: WRITE_SIZE of k_fill_synth must equal the index bytes
    traffic, pmc = {}, {}
    for thr, kern in (("1.0", "k_and_exact"), ("0.4", "k_and_count")):
        try:
            fe = json.load(open(os.path.join(SRC, "pmc_FETCH_SIZE_%s.json" % thr)))
            wr = json.load(open(os.path.join(SRC, "pmc_WRITE_SIZE_%s.json" % thr)))
        except OSError:
            continue
        pmc["threshold=" + thr] = {k: {"FETCH_SIZE_avg_kib": fe.get(k, {}).get("FETCH_SIZE", {}).get("avg"), "WRITE_SIZE_avg_kib": wr.get(k, {}).get("WRITE_SIZE", {}).get("avg"),
                                       "dispatches": fe.get(k, {}).get("FETCH_SIZE", {}).get("dispatches")} for k in sorted(set(fe) | set(wr)) if k.startswith("bigsi::")}
        name = [k for k in fe if kern in k]
        if not name:
            continue
        k = name[0]
        fetch = fe[k]["FETCH_SIZE"]["avg"] * 1024 * 2
        write = wr[k]["WRITE_SIZE"]["avg"] * 1024
        key = "rows=10000000 cols=100000 hashes=4 batch=8192 qlen=1000 k=31 threshold=%s draws=2" % thr
        traffic[key] = {"kernel": k, "traffic_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
                        "source": "profiles/" + R + "_c3_pmc.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; FETCH_SIZE doubled per "
                                  "MI355X_MICROARCH.md, calibrated on k_fill_synth's WRITE_SIZE = index bytes)"}
    # the one-launch read kernel on BASELINE configs[1]
    try:
        fe = json.load(open(os.path.join(SRC, "pmc_c2_FETCH_SIZE.json")))
        wr = json.load(open(os.path.join(SRC, "pmc_c2_WRITE_SIZE.json")))
        name = [k for k in fe if "k_reads_fused" in k]
        if name:
            k = name[0]
            fetch, write = fe[k]["FETCH_SIZE"]["avg"] * 1024 * 2, wr[k]["WRITE_SIZE"]["avg"] * 1024
            traffic["rows=1000000 cols=10000 hashes=3 batch=1000 qlen=61 k=31 threshold=1.0 draws=2"] = {
                "kernel": k, "traffic_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
                "source": "profiles/" + R + "_c2_pmc.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; FETCH_SIZE doubled per "
                          "MI355X_MICROARCH.md, which calibrates that factor on wide streaming reads: 1.25 KB rows fetch whole 128-byte lines)"}
            with open(os.path.join(DST, R + "_c2_pmc.json"), "w") as f:
                json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --workload c2 --steps 64 --warmup 8 "
                                      "--cpu-seconds 0 --no-verify (one counter per pass; scripts/profile_all.sh)",
                           "units": "KiB per dispatch, averaged over the dispatches of a kernel",
                           "kernels": {kk: {"FETCH_SIZE_avg_kib": fe.get(kk, {}).get("FETCH_SIZE", {}).get("avg"), "WRITE_SIZE_avg_kib": wr.get(kk, {}).get("WRITE_SIZE", {}).get("avg"),
                                            "dispatches": fe.get(kk, {}).get("FETCH_SIZE", {}).get("dispatches")} for kk in sorted(set(fe) | set(wr)) if kk.startswith("bigsi::")}}, f, indent=1)
    except OSError:
        pass
    if pmc:
        with open(os.path.join(DST, R + "_c3_pmc.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 "
                                  "--no-verify [--threshold 0.4]  (one counter per pass; scripts/profile_all.sh)",
                       "units": "KiB per dispatch, averaged over the dispatches of a kernel", "runs": pmc}, f, indent=1)
    if traffic:
        old = {}
        try:
            old = json.load(open(os.path.join(DST, "pmc_traffic.json")))
        except OSError:
            pass
        old.update(traffic)
        with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
            json.dump(old, f, indent=1)
    # round 4: scripts/pmc_all.py leaves its own reduced tables (every quoted kernel): merge / copy them
    try:
        new = json.load(open(os.path.join(SRC, "pmc", "pmc_traffic.json")))
        for v in new.values():
            v["source"] = v["source"].replace("profiles/pmc_kernels.json", "profiles/%s_pmc_kernels.json" % R)
        old = {}
        try:
            old = json.load(open(os.path.join(DST, "pmc_traffic.json")))
        except OSError:
            pass
        old.update(new)
        with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
            json.dump(old, f, indent=1)
        shutil.copy(os.path.join(SRC, "pmc", "pmc_kernels.json"), os.path.join(DST, R + "_pmc_kernels.json"))
    except OSError:
        pass
    for name in (R + "_latency_probe.txt", R + "_frontend_probe.json", R + "_bench_default_full.json", R + "_pmc_requests.json"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))
    for name in (R + "_row_probe.txt", R + "_call_breakdown.txt", R + "_call_trace.txt", R + "_k1_phases.txt", R + "_one_call_timeline.txt", R + "_python_stack_latency.txt", R + "_ingest.json", R + "_import.json", R + "_tmpfs_write_probe.txt",
                 R + "_bench_default.json", R + "_results_bench.txt", R + "_build_bench.jsonl", R + "_transpose_regs_ab.txt", R + "_tr_probe.txt"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))
    with open(os.path.join(DST, R + "_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    for k, v in summary.items():
        if "frac" in v:
            print("%-28s %8.1f M lookups/s  step %9.4f ms  %-34s %10.0f ns  %6.0f GB/s  frac %.3f" % (k, v["value_lookups_per_s"] / 1e6, v["ms_per_step"], v["kernel"][:34], v["kernel_ns"], v["GBps"], v["frac"]))
        else:
            for l in v["lines"]:
                print("%-28s %s" % (k, json.dumps(l)))


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--pmc-reduce":
        pmc_reduce(sys.argv[2], sys.argv[3])
    else:
        main()
