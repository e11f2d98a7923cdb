#!/bin/bash
# A/B of the index allocation (tuning build): BIGSI_HIP_CONTIGUOUS=1 (hipDeviceMallocContiguous) against 0 (plain hipMalloc), interleaved on
# one box, counting kernel at C3 (threshold 0.4); prints lookups/s, the kernel's fraction of peak and whether the index really got
# contiguous memory, per run (processes differ: the rate is bimodal from process to process).
L=$PWD/bigsi_amd/libbigsi_hip_tuning.so
run() { python bench.py $2 --steps 8 --warmup 3 --cpu-seconds 0 --also none --host-visible 0 --no-verify 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', '$2', round(d['value']/1e6,2), round(d['roofline']['frac'],4), 'contiguous', d['config']['index_contiguous'])"; }
for rep in 1 2 3 4 5 6 7 8; do
  BIGSI_HIP_LIB=$L BIGSI_HIP_CONTIGUOUS=1 run contig1 "--threshold 0.4"
  BIGSI_HIP_LIB=$L BIGSI_HIP_CONTIGUOUS=0 run contig0 "--threshold 0.4"
done
