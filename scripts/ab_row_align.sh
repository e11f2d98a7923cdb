cd "${GRAFT_REPO_ROOT:-.}"
export BIGSI_HIP_LIB=$PWD/bigsi_amd/libbigsi_hip_tuning.so
B="--cpu-seconds 0 --also none --host-visible 0 --alone-steps 0 --timed resident"
for rep in 1 2; do for a in 16 256; do
echo "== rep $rep align_words=$a"
for w in "--steps 10 --warmup 3" "--steps 10 --warmup 3 --threshold 0.4" "--workload c4 --shard-of 8 --gpus 1 --steps 100 --warmup 10" "--workload c5 --shard-of 8 --gpus 1 --steps 100 --warmup 10" "--workload northstar --shard-of 8 --gpus 1 --steps 100 --warmup 10"; do
BIGSI_HIP_ROW_ALIGN_WORDS=$a python bench.py $w $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %-60s %.1f M/s  frac %.4f  kernel_ms %.4f  gb %.1f' % (d['config']['workload'][:60], d['value']/1e6, d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['index_gb_per_gpu']))"
done; done; done
