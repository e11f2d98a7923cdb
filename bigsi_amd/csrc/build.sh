#!/bin/bash
# Build libbigsi_hip.so for gfx950 (MI355X), in-tree next to the Python package.
# hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libbigsi_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wextra -Wno-unused-parameter \
    -I"$ROOT/include" ${BIGSI_HIP_EXTRA_FLAGS:-} \
    -o "$OUT" "$HERE/bigsi_hip.hip"
echo "built $OUT"
