# GPU box: full gpu tests, the training bench line and steady-state kernel stats of the training step (tag = $1, default r02b).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && T=${1:-r02b} && mkdir -p gpurun_out/$T && O=gpurun_out/$T
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log; fi
timeout -k 10 300 python bench.py --mode train --steps 20 > $O/bench_train_scene.json 2>$O/bench_train_scene.err; cat $O/bench_train_scene.json | cut -c1-700
timeout -k 10 300 python bench.py --mode train --train-config gimo --steps 20 > $O/bench_train_gimo.json 2>/dev/null
rm -rf gpurun_out/kt_t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_t -- python bench.py --mode train --steps 10 --warmup 3 > $O/kt_train.log 2>&1
python scripts/steady_stats.py kt_t k_adamw 3 10 $O/kernel_stats_train_steady.csv k_gg
rm -rf gpurun_out/kt_t
