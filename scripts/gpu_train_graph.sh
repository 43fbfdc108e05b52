cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 200 python scripts/train_step_bench.py --scene-precision bf16 --steps 4 2>&1 | grep step | tail -2
