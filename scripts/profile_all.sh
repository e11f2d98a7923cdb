#!/bin/bash
# Everything DESIGN.md section 5 quotes, measured in one go on a GPU box: each workload under `rocprofv3 --kernel-trace --stats`
# (per-kernel durations; the command's own JSON -- HIP-event figures of the SAME run -- lands next to the CSV), the PMC passes
# for the dominant kernel (one counter per pass, as the hardware guide prescribes), and the unprofiled default bench line.
# scripts/make_profiles.py then condenses gpurun_out/prof into profiles/r02_*.
cd "${GRAFT_REPO_ROOT:-.}"
P=gpurun_out/prof
mkdir -p $P
B="--cpu-seconds 0"
run() { scripts/prof.sh "$@" > /dev/null; }
run r02_c3_exact        -- python bench.py --steps 20 --warmup 5 $B
run r02_c3_t04          -- python bench.py --steps 20 --warmup 5 $B --threshold 0.4
run r02_c3_256x1kbp     -- python bench.py --steps 200 --warmup 10 $B --batch 256
run r02_c3_256x1kbp_t04 -- python bench.py --steps 200 --warmup 10 $B --batch 256 --threshold 0.4
run r02_c2              -- python bench.py --workload c2 --steps 4000 --warmup 100 $B
run r02_c2_t04          -- python bench.py --workload c2 --steps 4000 --warmup 100 $B --threshold 0.4
run r02_c2_one_stream BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_HIP_READ_STREAMS=1 -- python bench.py --workload c2 --steps 4000 --warmup 100 $B
run r02_c2_unfused BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_HIP_FUSE_READS=0 -- python bench.py --workload c2 --steps 4000 --warmup 100 $B
run r02_c2_32k_reads    -- python bench.py --workload c2 --steps 200 --warmup 16 $B --batch 32768 --distinct-batches 8
run r02_c4_shard        -- python bench.py --workload c4 --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B
run r02_c5_shard        -- python bench.py --workload c5 --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B
run r02_northstar_shard -- python bench.py --workload northstar --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B
run r02_northstar_shard_t04 -- python bench.py --workload northstar --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B --threshold 0.4
run r02_northstar_shard_h4  -- python bench.py --workload northstar --shard-of 8 --gpus 1 --steps 200 --warmup 10 $B --hashes 4
run r02_c3_strong8_shard    -- python bench.py --workload c3 --shard-of 8 --gpus 1 --steps 40 --warmup 5 $B
run r02_c3_strong8_rccl1    -- python bench.py --workload c3 --shard-of 8 --gpus 1 --steps 40 --warmup 5 $B --force-dist
run r02_long_queries    -- python scripts/measure.py p16
run r02_k5              -- python scripts/measure.py k5
run r02_transpose       -- python scripts/measure.py transpose
# PMC: HBM traffic of the row-AND kernels (FETCH_SIZE / WRITE_SIZE in separate passes)
export TMPDIR=/tmp
for thr in 1.0 0.4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/raw_pmc -o pmc_${c}_$thr -- python bench.py --steps 2 --warmup 1 $B --no-verify --threshold $thr > $P/pmc_${c}_$thr.stdout 2> $P/pmc_${c}_$thr.stderr
    f=$(find $P/raw_pmc -name "pmc_${c}_${thr}_counter_collection.csv" | head -1)
    [ -n "$f" ] && python scripts/make_profiles.py --pmc-reduce "$f" $P/pmc_${c}_$thr.json
    rm -rf $P/raw_pmc
  done
done
# the same two counters for the one-launch read kernel (BASELINE configs[1])
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/raw_pmc -o pmc_c2_${c} -- python bench.py --workload c2 --steps 64 --warmup 8 $B --no-verify > $P/pmc_c2_${c}.stdout 2> $P/pmc_c2_${c}.stderr
  f=$(find $P/raw_pmc -name "pmc_c2_${c}_counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/make_profiles.py --pmc-reduce "$f" $P/pmc_c2_${c}.json
  rm -rf $P/raw_pmc
done
python bench.py --steps 20 --warmup 5 > $P/r02_bench_default.stdout 2> $P/r02_bench_default.stderr
python bench.py --steps 20 --warmup 5 --threshold 0.4 > $P/r02_bench_t04.stdout 2> $P/r02_bench_t04.stderr
ls $P | wc -l
