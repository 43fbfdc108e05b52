#!/bin/bash
# Build libseeme_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${SEEME_BUILD_OUT:-$here/../libseeme_hip.so}"
srcs=("$here"/api.hip "$here"/vae_kernels.hip "$here"/den_kernels.hip "$here"/misc_kernels.hip "$here"/pointnet_bf16.hip "$here"/pointnet_v2.hip "$here"/vae_h16.hip "$here"/glue_kernels.hip "$here"/vae_train.hip)
[ -f "$here/smpl_kernels.hip" ] && srcs+=("$here/smpl_kernels.hip")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function -o "$out" "${srcs[@]}" "$@"
echo "built $out"
