#!/bin/bash
# Micro-benchmark probes (NOT part of the product library): built into probes/libseeme_probes.so for gfx950.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/libseeme_probes.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -I"$here/../seeme_amd/csrc" -o "$out" "$here"/*.hip "$here/../seeme_amd/csrc/api.hip" "$@"
echo "built $out"
