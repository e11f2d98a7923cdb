#!/bin/bash
# round 6: k_transpose_regs (register bit transpose + transposing LDS read) against k_transpose_tiles, its RT variants and
# XCD groupings; tuning build, scripts/measure.py transpose (filters at a 128-byte pitch).
#   CFGS="regs double rg cg,..." REPS=n scripts/ab_transpose_regs.sh      (regs = 0: k_transpose_tiles<RT, 2> of rounds 2-6)
cd "${GRAFT_REPO_ROOT:-.}"
export BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_TR_SHAPES=${BIGSI_TR_SHAPES:-10000000x8192,1000000x100000,4000000x32768}
IFS=',' read -ra LIST <<< "${CFGS:-1 1 4 1,1 0 4 1,0 0 4 1,0 1 4 1,1 1 2 1,1 1 1 1,1 1 8 1,1 1 2 2,1 1 4 2,1 1 1 4,1 1 4 1}"
for rep in $(seq 1 ${REPS:-1}); do
for cfg in "${LIST[@]}"; do
    set -- $cfg
    echo "== rep $rep regs=$1 RT=$((1 + $2)) rg=$3 cg=$4"
    BIGSI_HIP_TR_REGS=$1 BIGSI_HIP_TR_DOUBLE=$2 BIGSI_HIP_TR_RG=$3 BIGSI_HIP_TR_CG=$4 python scripts/measure.py transpose 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %d x %d: %.0f GB/s (%.3f)' % (d['m'], d['cols'], d['GBps'], d['frac']))"
done
done
