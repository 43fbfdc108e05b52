# A/B of prebuilt library variants (probes/variants/libseeme_<name>.so) on the 50-step DDIM launch at B = 32 (or the sizes in $B): scripts/clm_time_only.py, twice each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do
  timeout -k 10 100 python scripts/clm_time_only.py ${B:-32} 2>&1 | grep '^{' | tee -a gpurun_out/cl_libs.txt
  for n in "$@"; do
    SEEME_HIP_LIB=$PWD/probes/variants/libseeme_$n.so timeout -k 10 100 python scripts/clm_time_only.py ${B:-32} 2>&1 | grep '^{' | tee -a gpurun_out/cl_libs.txt
  done
done
