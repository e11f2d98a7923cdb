#!/bin/bash
# Build libbigsi_hip.so for gfx950 (MI355X), in-tree next to the Python package.
# hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libbigsi_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# `build.sh tuning` builds libbigsi_hip_tuning.so instead: the same code with the A/B knobs of DESIGN.md section 6 read from
# the environment (-DBIGSI_HIP_TUNING); select it with BIGSI_HIP_LIB=... .  The product library never calls getenv.
FLAGS="${BIGSI_HIP_EXTRA_FLAGS:-}"
if [ "${1:-}" = "tuning" ]; then
    OUT="$HERE/../libbigsi_hip_tuning.so"
    FLAGS="$FLAGS -DBIGSI_HIP_TUNING"
fi
# librccl is NOT linked: bigsi_shard.hip loads it with dlopen at the first multi-GPU call (see there)
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wextra -Wno-unused-parameter \
    -I"$ROOT/include" -I/opt/rocm/include $FLAGS \
    -o "$OUT" "$HERE/bigsi_hip.hip" "$HERE/bigsi_shard.hip" "$HERE/bigsi_text.cpp" -ldl -lpthread
echo "built $OUT"
