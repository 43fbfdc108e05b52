cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "pointnet or training_step or hip_backward" > gpurun_out/tests_pn.log 2>&1; tail -3 gpurun_out/tests_pn.log
for f in 0 1; do
bash seeme_amd/csrc/build.sh -DPN_DBG_TIMES -DPN_DBG_FIRST=$f > gpurun_out/build_dbg.log 2>&1 || { tail -5 gpurun_out/build_dbg.log; exit 1; }
echo "== first block: $f"
timeout -k 10 300 python scripts/pn_times.py 2>&1 | tail -11
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
