#!/bin/bash
# builds scripts/probe/row_probe for gfx950 (a measurement tool; the binary travels with gpurun snapshots and is git-ignored)
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o "$HERE/row_probe" "$HERE/row_probe.hip"
echo built "$HERE/row_probe"
