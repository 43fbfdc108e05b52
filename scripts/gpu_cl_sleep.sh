# kernel time + stamps of k_den_cluster for several DCL_POLL_SLEEP builds (debug libraries built locally as libseeme_hip_dbg_s<N>.so)
out=gpurun_out/${1:-r3q}; mkdir -p $out
for lib in seeme_amd/libseeme_hip_dbg.so seeme_amd/libseeme_hip_dbg_s*.so; do
  echo "== $lib"
  for p in 0 1; do
    SEEME_HIP_LIB=$PWD/$lib SEEME_DEN_CLUSTER=8 SEEME_DEN_CLUSTER_PLACE=$p python scripts/cl_times.py fp16 32 2>&1 | grep -v amdgpu.ids
  done
done > $out/times.txt
cat $out/times.txt
