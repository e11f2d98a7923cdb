#!/bin/bash
# builds scripts/probe/row_probe, tile_probe, transpose_ab, latency_probe and tr_probe for gfx950 (measurement tools; the binaries travel with gpurun snapshots and are git-ignored)
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
for t in row_probe tile_probe transpose_ab latency_probe tr_probe; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBIGSI_HIP_TUNING -o "$HERE/$t" "$HERE/$t.hip"
    echo built "$HERE/$t"
done
