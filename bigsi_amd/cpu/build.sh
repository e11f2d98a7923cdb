#!/bin/bash
# Build libbigsi_cpu.so (include/bigsi_cpu.h): the CPU twin of the CORE layer of the C ABI, in-tree next to libbigsi_hip.so.
# Plain g++; contraction off because csrc/bigsi_score.hpp reproduces CPython's arithmetic operation by operation.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libbigsi_cpu.so"
# libstdc++ / libgcc linked statically: the twin is what a host WITHOUT this image's toolchain binds (any interpreter, e.g. the
# conda python that runs the reference in tests/golden/run_reference_suite.py ships an older libstdc++.so.6 than g++ here links against)
"${CXX:-g++}" -O2 -std=c++17 -ffp-contract=off -fPIC -shared -static-libstdc++ -static-libgcc -Wall -Wextra -Werror -I"$ROOT/include" -o "$OUT" "$HERE/bigsi_cpu.cpp" -lpthread
echo "built $OUT"
