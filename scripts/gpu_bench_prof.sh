cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 --weights fp32 --no-cpu-baseline > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err
python bench.py --steps 10 --warmup 2 --weights bf16 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python bench.py --steps 5 --warmup 1 --weights bf16 --no-cpu-baseline > gpurun_out/prof_r1.log 2>&1
cat gpurun_out/bench_fp32.json gpurun_out/bench_bf16.json
