#!/bin/bash
# GPU box: time the cumulative timing-only ablations of k_pn_block2 built by scripts/build_pn2_variants.sh (B=64 x 20000 points).
mkdir -p gpurun_out/r02
out=gpurun_out/r02/pointnet_v2_anatomy.jsonl; : > $out
for so in probes/variants/libseeme_hip_*.so; do
  n=$(basename $so .so); n=${n#libseeme_hip_}
  SEEME_HIP_LIB=$PWD/$so timeout -k 10 120 python scripts/pn_bench.py 2>/dev/null | sed "s/^{/{\"variant\": \"$n\", /" >> $out || exit 1
done
cat $out
