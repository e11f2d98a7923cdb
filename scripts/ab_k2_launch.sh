cd "${GRAFT_REPO_ROOT:-.}"
export BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so
B="--cpu-seconds 0 --also none --host-visible 0 --alone-steps 0 --timed resident --steps 10 --warmup 3"
python bench.py $B > /dev/null 2>&1   # warm the box: the first run after a lease is the fast one
for rep in 1 2; do for v in "WAVES 1600" "WAVES 1280" "WAVES 2048" "WAVES 2400" "BLOCKS 512" "BLOCKS 768" "BLOCKS 640" "AND_UNROLL 4" "AND_UNROLL 16"; do set -- $v
echo -n "rep $rep K2_$1=$2: "
env BIGSI_HIP_K2_$1=$2 BIGSI_HIP_$1=$2 python bench.py $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.1f M/s  frac %.4f  kernel_ms %.4f launches/step %s' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['launches_per_step']))"
done; done
