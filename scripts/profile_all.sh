#!/bin/bash
# Everything DESIGN.md section 5 quotes, measured in one go on a GPU box: each workload under `rocprofv3 --kernel-trace --stats`
# (per-kernel durations; the command's own JSON -- HIP-event figures of the SAME run -- lands next to the CSV), the PMC passes
# for the dominant kernel (one counter per pass, as the hardware guide prescribes), and the unprofiled default bench line.
# scripts/make_profiles.py then condenses gpurun_out/prof into profiles/<round>_* (ROUND=r03 by default).
cd "${GRAFT_REPO_ROOT:-.}"
R=${ROUND:-r06}
P=gpurun_out/prof
mkdir -p $P
B="--cpu-seconds 0 --also none --host-visible 0 --alone-steps 0"      # (rocprofv3 averages then cover launches of the timed shape only)
[ -f bigsi_amd/libbigsi_hip_tuning.so ] || bash bigsi_amd/csrc/build.sh tuning > /dev/null      # (two legs below A/B through it)
run() { scripts/prof.sh "$@" > /dev/null; }
python - <<PYEOF > $P/${R}_build.json
import glob, hashlib, json
sha = lambda f: hashlib.sha256(open(f, "rb").read()).hexdigest()
print(json.dumps({"so_sha256": sha("bigsi_amd/libbigsi_hip.so"), "results_ext_sha256": next((sha(f) for f in sorted(glob.glob("bigsi_amd/_results*.so"))), None)}))
PYEOF
run ${R}_c3_exact        -- python bench.py --steps 20 --warmup 5 $B
run ${R}_c3_t04          -- python bench.py --steps 20 --warmup 5 $B --threshold 0.4
run ${R}_c3_256x1kbp     -- python bench.py --steps 200 --warmup 10 $B --batch 256
run ${R}_c3_256x1kbp_t04 -- python bench.py --steps 200 --warmup 10 $B --batch 256 --threshold 0.4
run ${R}_c2              -- python bench.py --workload c2 --steps 4000 --warmup 100 $B
run ${R}_c2_t04          -- python bench.py --workload c2 --steps 4000 --warmup 100 $B --threshold 0.4
run ${R}_c2_one_stream   -- python bench.py --workload c2 --steps 4000 --warmup 100 $B --one-stream      # the read kernel alone on the device: what roofline.frac of read workloads is priced on
run ${R}_c2_t04_one_stream -- python bench.py --workload c2 --steps 4000 --warmup 100 $B --one-stream --threshold 0.4
run ${R}_c2_unfused BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_HIP_FUSE_READS=0 -- python bench.py --workload c2 --steps 4000 --warmup 100 $B
run ${R}_c2_32k_reads    -- python bench.py --workload c2 --steps 200 --warmup 16 $B --batch 32768 --distinct-batches 8
run ${R}_c4_shard        -- python bench.py --workload c4 --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B
run ${R}_c5_shard        -- python bench.py --workload c5 --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B
run ${R}_northstar_shard -- python bench.py --workload northstar --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B
run ${R}_northstar_shard_t04 -- python bench.py --workload northstar --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B --threshold 0.4
run ${R}_northstar_shard_h4  -- python bench.py --workload northstar --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B --hashes 4
run ${R}_c3_strong8_shard    -- python bench.py --workload c3 --shard-of 8 --gpus 1 --steps 40 --warmup 5 $B
run ${R}_c3_strong8_rccl1    -- python bench.py --workload c3 --shard-of 8 --gpus 1 --steps 40 --warmup 5 $B --force-dist
# round 6: the headline as rounds 1-5 timed it (batches resident in HBM, hit lists left on the device), and configs[1] on an index far beyond the
# 256 MB Infinity Cache (8 M x 10 k = 10 GB)
run ${R}_c3_exact_resident -- python bench.py --steps 20 --warmup 5 $B --timed resident
run ${R}_c2_10gb         -- python bench.py --workload c2 --rows 8000000 --steps 4000 --warmup 100 $B
run ${R}_c2_10gb_one_stream -- python bench.py --workload c2 --rows 8000000 --steps 4000 --warmup 100 $B --one-stream
# the hit-dense regime (round 5): ~10 k scored hits per batch on the c5 shard, 8 hits per read on c2, early exit on the dense shard
run ${R}_c5_dense        -- python bench.py --workload c5 --shard-of 8 --dense 1 --steps 60 --warmup 6 $B
run ${R}_c2_dense        -- python bench.py --workload c2 --dense 1 --steps 4000 --warmup 100 $B
run ${R}_c5_ee           -- python bench.py --workload c5 --shard-of 8 --dense 1 --early-exit 1 --score 0 --steps 200 --warmup 10 $B
run ${R}_long_queries    -- python scripts/measure.py p16
run ${R}_k5              -- python scripts/measure.py k5
run ${R}_transpose       -- python scripts/measure.py transpose
run ${R}_scored_stream   -- python scripts/scored_stream_probe.py
# what the memory system gives the row-AND access pattern with no BIGSI code in the way (scripts/probe/row_probe.hip)
{ for a in "" "--vmm"; do scripts/probe/row_probe --gb 125 --row-bytes 12500 --rows-per-query 3880 --queries 768 $a; done
  scripts/probe/row_probe --gb 16 --row-bytes 12500 --rows-per-query 3880 --queries 768
  scripts/probe/row_probe --gb 125 --row-bytes 7813 --rows-per-query 2900 --queries 1024
  scripts/probe/row_probe --gb 1.25 --row-bytes 1250 --rows-per-query 93 --queries 8192 --wgs 2048
  scripts/probe/row_probe --gb 10 --row-bytes 1250 --rows-per-query 93 --queries 8192 --wgs 2048 --modes random,sorted
  # round 6: the counting kernel's constraint (a k-mer's h rows together) with the k-mers ordered by their first row
  scripts/probe/row_probe --gb 125 --row-bytes 12500 --rows-per-query 3880 --queries 768 --modes random,sorted,kfirst,kpage; } > $P/${R}_row_probe.txt 2>&1
python scripts/call_breakdown.py > $P/${R}_call_breakdown.txt 2>/dev/null
BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so python scripts/call_breakdown.py 2>/dev/null | grep "inside the call" > $P/${R}_call_trace.txt      # host clock inside the one-call entry point (tuning build)
BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so python scripts/ab_k1_phases.py > $P/${R}_k1_phases.txt 2>/dev/null
BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so python scripts/ab_k1_phases.py 0.4 >> $P/${R}_k1_phases.txt 2>/dev/null
# the device timeline of a one-call search of one 1 kbp query (kernel durations and gaps), exact and at 0.4
( export TMPDIR=/tmp; mkdir -p $P/tl; rocprofv3 --kernel-trace --output-format csv -d $P/tl -o t -- python scripts/one_call_timeline.py run > $P/tl_run.log 2>&1
  python scripts/one_call_timeline.py report $P/tl > $P/${R}_one_call_timeline.txt 2>&1 )
scripts/probe/latency_probe > $P/${R}_latency_probe.txt 2>&1
# round 5: file <-> HBM (striped snapshot, one-file save for comparison, two-shard group), importers, the tmpfs write probe, the dict builder
python scripts/ingest_bench.py --gb 32 > $P/${R}_ingest.json 2>/dev/null
python scripts/import_bench.py --gb 16 --bdb-gb 6 > $P/${R}_import.json 2>/dev/null
# PMC: HBM traffic of every quoted kernel (FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only)
export TMPDIR=/tmp
python scripts/pmc_all.py $P/pmc > $P/${R}_pmc_all.log 2>&1
# round 6: the request-size classes themselves (no x2 to argue about): calibration on the row-AND kernel, K5 on both c5 legs, the transpose
python scripts/pmc_requests.py $P/pmc_req c3_exact c5_shard c5_dense transpose > $P/${R}_pmc_requests.log 2>&1
cp $P/pmc_req/pmc_requests.json $P/${R}_pmc_requests.json
python bench.py --steps 20 --warmup 5 --details $P/${R}_bench_default_full.json > $P/${R}_bench_default.stdout 2> $P/${R}_bench_default.stderr      # the driver's command: headline + every other config as config.also legs + the CPU baseline
python bench.py --steps 20 --warmup 5 --threshold 0.4 --also none > $P/${R}_bench_t04.stdout 2> $P/${R}_bench_t04.stderr
# round 6: the Python stack's latency, the host-visible build, the transpose A/B against the previous kernel (tuning build), the transposing read's lane map
python scripts/latency_probe.py > $P/${R}_python_stack_latency.txt 2>&1
{ python scripts/build_bench.py 4000000 16384; python scripts/build_bench.py 1000000 100000; } > $P/${R}_build_bench.jsonl 2>/dev/null
CFGS="1 1 4 1,0 0 4 1,0 1 4 1" REPS=3 scripts/ab_transpose_regs.sh > $P/${R}_transpose_regs_ab.txt 2>&1
scripts/probe/tr_probe > $P/${R}_tr_probe.txt 2>&1
ls $P | wc -l
