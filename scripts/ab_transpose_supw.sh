cd "${GRAFT_REPO_ROOT:-.}"
export BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so BIGSI_TR_SHAPES=1000000x100000,2000000x65536,10000000x8192
for rep in 1 2; do for cfg in "32 4 1" "64 4 1" "128 4 1" "128 2 1" "128 8 1" "256 4 1" "16 4 1" "128 4 2"; do set -- $cfg
echo "== rep $rep supw=$1 rg=$2 cg=$3"
BIGSI_HIP_TR_SUPW=$1 BIGSI_HIP_TR_RG=$2 BIGSI_HIP_TR_CG=$3 python scripts/measure.py transpose 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %d x %d: %.0f GB/s (%.3f)' % (d['m'], d['cols'], d['GBps'], d['frac']))"
done; done
