#!/bin/bash
# Where do rows at random addresses lose their bandwidth?  Translation (UTCL1 / UTCL2), memory-side latency (TCC -> EA), DRAM
# credits: one rocprofv3 --pmc pass per counter group (kernel trace only, as the hardware guide prescribes) for three variants of
# the row-AND kernel on the 125 GB C3 index:
#   exact_sorted    k_and_exact, address-ordered row lists, launches of 512 workgroups (the shipped path, 0.857 of peak)
#   exact_unsorted  the same kernel with BIGSI_HIP_SORT_ROWS=0 (tuning build): rows in hash order (0.78-0.79)
#   count           k_and_count<10,4> at threshold 0.4 (0.786)
# Output: gpurun_out/pmc_attr/<variant>_<group>.csv reduced by scripts/pmc_attribution_reduce.py into profiles/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_attr
mkdir -p $O
B="--steps 2 --warmup 1 --no-verify --cpu-seconds 0 --also none"
declare -A CGRP
CGRP[utcl1]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
CGRP[utcl2]="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
CGRP[ealat]="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
CGRP[tcplat]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
CGRP[dram]="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum"
CGRP[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
CGRP[stall]="TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
for variant in exact_sorted exact_unsorted count; do
  case $variant in
    exact_sorted)   envs=(); extra="" ;;
    exact_unsorted) envs=(BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_HIP_SORT_ROWS=0); extra="" ;;
    count)          envs=(); extra="--threshold 0.4" ;;
  esac
  for g in "${!CGRP[@]}"; do
    rm -rf $O/raw
    env "${envs[@]}" rocprofv3 --pmc ${CGRP[$g]} --kernel-trace --output-format csv -d $O/raw -o p -- python bench.py $B $extra > $O/${variant}_$g.stdout 2> $O/${variant}_$g.stderr
    f=$(find $O/raw -name "p_counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python scripts/pmc_attribution_reduce.py "$f" > $O/${variant}_$g.json; else echo "no counters for $variant $g" >&2; tail -3 $O/${variant}_$g.stderr >&2; fi
    rm -rf $O/raw
  done
done
ls $O | wc -l
