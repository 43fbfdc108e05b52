cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for k in ${@:-1 2 3 4}; do
bash seeme_amd/csrc/build.sh -DH16_DBG_TIMES -DH16_DBG_KERNEL=$k > gpurun_out/build_dbg.log 2>&1 || { tail -5 gpurun_out/build_dbg.log; exit 1; }
timeout -k 10 300 python scripts/h16_times.py $k 2>&1 | grep -v "^{" | tail -14
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
