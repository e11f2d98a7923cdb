"""What the L2 asks the memory side for, BY REQUEST SIZE, per kernel (round 6: K5's over-fetch, the transpose's write side):

    python scripts/pmc_requests.py OUTDIR [tag ...]      # default: every workload below

FETCH_SIZE on gfx950 is an expression over the TCC's fabric-side request counters
(`rocprofv3 -L`: (TCC_BUBBLE*128 + (TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B)*64 + TCC_EA0_RDREQ_32B*32) / 1024, TCC_BUBBLE being
"128-byte read requests sent to EA" there), and the hardware guide's "double it" is calibrated for wide streaming reads only.  For a
kernel that gathers 16 bytes here and there the doubling is a guess; this script reads the size classes themselves --
TCC_EA0_RDREQ_32B / _64B / _128B (+ the total and TCC_BUBBLE), one counter per pass as the guide prescribes, --kernel-trace only --
so that bytes = 32 n32 + 64 n64 + 128 n128 needs no correction, and checks that on the row-AND kernel, whose bytes are known.
Writes OUTDIR/pmc_requests.json: per workload and kernel, the average of every counter per dispatch and the derived bytes."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_req"))
B = ["--cpu-seconds", "0", "--also", "none", "--host-visible", "0", "--no-verify", "--alone-steps", "0"]
READ = ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum", "TCC_BUBBLE_sum"]
WRITE = ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]
# address translation and read latency as the vector L1 sees them (round 6: why the counting kernel's hash-ordered rows run
# below the exact kernel's address-ordered ones): the bare row streams of scripts/probe/row_probe over a 125 GB matrix
TLB = ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum",
       "TCP_PENDING_STALL_CYCLES_sum"]
PROBE = [os.path.join(ROOT, "scripts", "probe", "row_probe"), "--gb", "125", "--row-bytes", "12500", "--rows-per-query", "3880", "--queries", "768", "--modes"]
PY = [sys.executable]
WORKLOADS = {
    "c3_exact": (PY + ["bench.py", "--steps", "2", "--warmup", "1", "--timed", "resident"] + B, READ),
    "c5_shard": (PY + ["bench.py", "--workload", "c5", "--shard-of", "8", "--steps", "8", "--warmup", "2"] + B, READ),
    "c5_dense": (PY + ["bench.py", "--workload", "c5", "--shard-of", "8", "--dense", "1", "--steps", "8", "--warmup", "2"] + B, READ),
    "transpose": (PY + ["scripts/measure.py", "transpose"], READ + WRITE + ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]),
    "probe_random": (PROBE + ["random"], TLB),
    "probe_sorted": (PROBE + ["sorted"], TLB),
    "probe_kfirst": (PROBE + ["kfirst"], TLB),
    "c3_exact_tlb": (PY + ["bench.py", "--steps", "2", "--warmup", "1", "--timed", "resident"] + B, TLB),
    "c3_t04_tlb": (PY + ["bench.py", "--steps", "2", "--warmup", "1", "--timed", "resident", "--threshold", "0.4"] + B, TLB),
}


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def one_pass(tag, counter, cmd):
    raw = os.path.join(OUT, "raw_%s_%s" % (tag, counter))
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(OUT, "%s_%s.log" % (tag, counter)), "w") as log:
        subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", raw, "-o", "p", "--"] + cmd,
                       cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, timeout=1500)
    acc = {}
    for path in glob.glob(os.path.join(raw, "**", "p_counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                a = acc.setdefault(short(r["Kernel_Name"]), [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    shutil.rmtree(raw, ignore_errors=True)
    return {k: {"dispatches": n, "avg": tot / n} for k, (n, tot) in acc.items() if k.startswith("bigsi::") or k.startswith("k_stream_rows")}


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[2:])
    result = {}
    for tag, (cmd, counters) in WORKLOADS.items():
        if only and tag not in only:
            continue
        per = {}
        for c in counters:
            for k, v in one_pass(tag, c, cmd).items():
                per.setdefault(k, {"dispatches": v["dispatches"]})[c] = v["avg"]
        for k, v in per.items():
            if all(c in v for c in READ[:3]):
                v["read_bytes"] = 32 * v[READ[0]] + 64 * v[READ[1]] + 128 * v[READ[2]]
                v["fetch_size_expr_bytes"] = v.get("TCC_BUBBLE_sum", 0) * 128 + (v.get("TCC_EA0_RDREQ_sum", 0) - v.get("TCC_BUBBLE_sum", 0) - v[READ[0]]) * 64 + v[READ[0]] * 32
            if all(c in v for c in WRITE):
                v["write_bytes"] = 32 * (v[WRITE[0]] - v[WRITE[1]]) + 64 * v[WRITE[1]]
        for k, v in per.items():
            if v.get("TCP_UTCL1_REQUEST_sum"):
                v["utcl1_miss_per_request"] = v.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / v["TCP_UTCL1_REQUEST_sum"]
            if v.get("TCP_TCC_READ_REQ_sum"):
                v["read_latency_cycles"] = v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / v["TCP_TCC_READ_REQ_sum"]
        result[tag] = {"command": "rocprofv3 --pmc <one counter> --kernel-trace -- " + " ".join(c_.replace(ROOT + "/", "").replace(sys.executable, "python") for c_ in cmd), "kernels": per}
        print(tag, json.dumps({k: {c: round(x, 4) for c, x in v.items()} for k, v in per.items() if v.get("read_bytes", 0) > 1e6 or v.get("TCP_UTCL1_REQUEST_sum", 0) > 1e5}), flush=True)
        with open(os.path.join(OUT, "pmc_requests.json"), "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
