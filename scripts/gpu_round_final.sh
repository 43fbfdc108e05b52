# Round-end evidence: full GPU suite, smoke, default bench (with cpu_baseline), variants, kernel-trace stats, PMC traffic.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/tests.log 2>&1; grep -E "rel err|MPJPE|passed|failed|Error" gpurun_out/tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 &&
T0=$(date +%s); python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench wall $(( $(date +%s) - T0 )) s"; cat gpurun_out/bench_default.json &&
bash scripts/gpu_test_bench.sh 2>&1 | tail -4 &&
bash scripts/gpu_prof.sh fp16 | tail -25 &&
cp gpurun_out/prof_cur/*/*kernel_stats.csv gpurun_out/final_kernel_stats.csv &&
bash scripts/gpu_traffic.sh fp16
