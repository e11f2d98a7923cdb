#!/bin/bash
# interleaved A/B of environment-variable variants of the bench on one box: ab_env.sh ROUNDS "VAR=1 VAR2=3" "..." -- bench args
rounds=$1; shift
variants=()
while [ "$1" != "--" ]; do variants+=("$1"); shift; done
shift
for r in $(seq $rounds); do
  for v in "${variants[@]}"; do
    out=$(env $v python bench.py --cpu-seconds 0 "$@" 2>gpurun_out/ab_env.err | tail -1)
    [ -z "$out" ] && { echo "$v: no output"; tail -3 gpurun_out/ab_env.err; continue; }
    echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s value %.2fM  step %.4f ms  K2 %.4f ms  %.0f GB/s  step_frac %.4f  K1 %s ms  verified %s' % ('$v', d['value']/1e6, d['ms_per_step'], r['kernel_ms'], r['achieved'], r.get('step_frac') or 0, r.get('kmerize_ms'), bool(d['config'].get('verified'))))"
  done
done
