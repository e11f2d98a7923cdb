#!/bin/bash
# Build bigsi_amd/_results.<abi>.so: the CPython extension that assembles the result dicts of BIGSI.search_stream (host code; g++).
# -ffp-contract=off: py_round2 (csrc/bigsi_score.hpp) writes out the one fma it means.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PY="${PYTHON:-python3}"
INC="$($PY -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
SUF="$($PY -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
g++ -O2 -std=c++17 -shared -fPIC -pthread -ffp-contract=off -Wall -Wextra -I"$INC" -I"$HERE" -o "$HERE/_results$SUF" "$HERE/_results.cpp"
echo "built $HERE/_results$SUF"
