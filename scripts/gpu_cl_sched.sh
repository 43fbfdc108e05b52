out=gpurun_out/${1:-r3r}; mkdir -p $out
for lib in seeme_amd/libseeme_hip_dbg_s*.so; do
  echo "== $lib"
  SEEME_HIP_LIB=$PWD/$lib SEEME_DEN_CLUSTER=8 SEEME_DEN_CLUSTER_PLACE=1 python scripts/cl_times.py fp16 32 2>&1 | grep -v amdgpu.ids
done > $out/times.txt
cat $out/times.txt
