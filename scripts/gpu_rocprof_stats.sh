# rocprofv3 --kernel-trace --stats of the default bench command (whole process: includes model construction and the parity /
# MPJPE passes; the steady-state cut of the same trace is scripts/gpu_r02_evidence.sh's kernel_stats_sample_steady.csv).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -rf gpurun_out/rp_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_stats -- python bench.py --no-cpu-baseline > gpurun_out/rp_stats_bench.json 2> gpurun_out/rp_stats.err
f=$(ls gpurun_out/rp_stats/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/rocprof_kernel_stats_default.csv
rm -rf gpurun_out/rp_stats
head -8 gpurun_out/rocprof_kernel_stats_default.csv; grep '^{' gpurun_out/rp_stats_bench.json | cut -c1-200
