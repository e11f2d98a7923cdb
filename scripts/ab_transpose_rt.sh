cd "${GRAFT_REPO_ROOT:-.}"
export BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_TR_SHAPES=10000000x8192,1000000x100000,4000000x32768 BIGSI_HIP_TR_REGS=0      # (the kernel of rounds 2-6: tuning builds only)
for rep in 1 2 3; do for cfg in "1 1 4 1" "0 1 4 1" "0 1 8 1"; do
    set -- $cfg
    echo "== rep $rep RT=$((1 + $1)) CT=$((1 + $2)) rg=$3 cg=$4"
    BIGSI_HIP_TR_DOUBLE=$1 BIGSI_HIP_TR_WIDE=$2 BIGSI_HIP_TR_RG=$3 BIGSI_HIP_TR_CG=$4 python scripts/measure.py transpose 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %d x %d: %.0f GB/s (%.3f)' % (d['m'], d['cols'], d['GBps'], d['frac']))"
done; done
