#!/bin/bash
# Variant builds of pointnet_v2.hip (probes/variants/libseeme_<name>.so; the other sources compiled once), for scripts/gpu_pn2_libs*.sh.
# usage: build_pn2_libs.sh name1 "flags1" name2 "flags2" ...
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/probes/variants"; mkdir -p "$out"
src="$root/seeme_amd/csrc"
objs=()
for f in api vae_kernels den_kernels misc_kernels pointnet_bf16 vae_h16 smpl_kernels glue_kernels vae_train; do
  [ -f "$out/$f.o" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -o "$out/$f.o" "$src/$f.hip" &
  objs+=("$out/$f.o")
done
wait
while [ $# -ge 2 ]; do
  name="$1"; flags="$2"; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function $flags -o "$out/pn2_$name.o" "$src/pointnet_v2.hip"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$out/libseeme_$name.so" "$out/pn2_$name.o" "${objs[@]}"
    echo "built $name [$flags]" ) &
done
wait
