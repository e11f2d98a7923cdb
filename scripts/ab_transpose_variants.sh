#!/bin/bash
# round 6: the four tile shapes of k_transpose_tiles (RT x CT) and their XCD groupings, with the filters at a 128-byte pitch (every earlier
# A/B of these ran at the packed, 16-byte pitch, where a 128-byte run straddles two lines); tuning build, scripts/measure.py transpose
cd "${GRAFT_REPO_ROOT:-.}"
export BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_TR_SHAPES=10000000x8192,1000000x100000 BIGSI_HIP_TR_REGS=0      # (the kernel of rounds 2-6: tuning builds only)
for cfg in "1 1 4 1" "1 1 2 2" "1 1 4 2" "1 1 1 1" "0 0 2 2" "0 0 4 4" "0 1 2 1" "0 1 4 1" "1 0 1 2" "1 0 2 2"; do
    set -- $cfg
    echo "== RT=$((1 + $1)) CT=$((1 + $2)) rg=$3 cg=$4"
    BIGSI_HIP_TR_DOUBLE=$1 BIGSI_HIP_TR_WIDE=$2 BIGSI_HIP_TR_RG=$3 BIGSI_HIP_TR_CG=$4 python scripts/measure.py transpose 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %d x %d: %.0f GB/s (%.3f)' % (d['m'], d['cols'], d['GBps'], d['frac']))"
done
