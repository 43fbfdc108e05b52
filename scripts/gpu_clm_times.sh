# cycle stamps of k_den_cluster_ms (debug library built HERE into seeme_amd/libseeme_hip_dbg.so with -DDEN_DBG_TIMES): one step, per phase
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export SEEME_HIP_LIB=$PWD/seeme_amd/libseeme_hip_dbg.so
for B in ${1:-128 256 512}; do
  SEEME_DEN_CLUSTER=auto timeout -k 10 120 python scripts/cl_times.py fp16 $B 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/clm_times.txt
