#!/bin/bash
# does the counting kernel's rate depend on how long the run is (power / thermal state)?  C3 at threshold 0.4, different step counts,
# clocks / power / temperature from rocm-smi before and after the timed region (bench.py logs them)
run() { python bench.py --threshold 0.4 --steps $1 --warmup 3 --cpu-seconds 0 --also none --host-visible 0 --no-verify 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);c=d['config']['clocks'];a=c['after_timed_region'];b=c['before_timed_region']
pick=lambda x:x
print('steps $1', round(d['value']/1e6,2), round(d['roofline']['frac'],4), 'before', pick(b), 'after', pick(a))"; }
for s in 4 8 16 32 8 32 4; do run $s; done
