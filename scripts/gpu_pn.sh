cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -s -k "pointnet or training_step" > gpurun_out/tests_pn.log 2>&1; grep -E "rel err|passed|failed|Error|error" gpurun_out/tests_pn.log | tail -6
python scripts/train_breakdown.py vae16 2>&1 | tail -1
