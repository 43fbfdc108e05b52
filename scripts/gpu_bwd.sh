cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -s -k "hip_backward" > gpurun_out/tests_bwd.log 2>&1; grep -v amdgpu.ids gpurun_out/tests_bwd.log | tail -${1:-25}
