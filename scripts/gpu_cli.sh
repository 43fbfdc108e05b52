cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -s -k "cli or variants or stage1" > gpurun_out/tests_cli.log 2>&1; tail -25 gpurun_out/tests_cli.log
