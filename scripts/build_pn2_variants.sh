#!/bin/bash
# Timing-only ablation builds of k_pn_block2 (results are WRONG by construction; they bound what each part of the kernel costs).
# Built here (hipcc cross-compiles), shipped with the snapshot, selected on the GPU box with SEEME_HIP_LIB.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/probes/variants"; mkdir -p "$out"
src="$root/seeme_amd/csrc"
objs=()
for f in api vae_kernels den_kernels misc_kernels pointnet_bf16 vae_h16 smpl_kernels glue_kernels vae_train; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -o "$out/$f.o" "$src/$f.hip" &
  objs+=("$out/$f.o")
done
wait
build() { # name flags...
  name="$1"; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function "$@" -o "$out/pn2_$name.o" "$src/pointnet_v2.hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$out/libseeme_hip_$name.so" "$out/pn2_$name.o" "${objs[@]}"
}
build 1_mfma_only   -DP2_ABL_NOXLOAD -DP2_ABL_NORING -DP2_ABL_NOLDS -DP2_ABL_NOEPI -DP2_ABL_NOSTORE &
build 2_plus_xloads                  -DP2_ABL_NORING -DP2_ABL_NOLDS -DP2_ABL_NOEPI -DP2_ABL_NOSTORE &
build 3_plus_ring                                    -DP2_ABL_NOLDS -DP2_ABL_NOEPI -DP2_ABL_NOSTORE &
build 4_plus_ldsreads                                               -DP2_ABL_NOEPI -DP2_ABL_NOSTORE &
wait
build 5_plus_epilogue                                                              -DP2_ABL_NOSTORE &
build 6_full &
build 7_full_nobarrier -DP2_ABL_NOBAR &
wait
rm -f "$out"/*.o
ls -la "$out"
