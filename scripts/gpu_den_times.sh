cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for flag in "-DDEN_DBG_TIMES" "-DDEN_DBG_TIMES -DDEN_DBG_NOLOAD"; do
bash seeme_amd/csrc/build.sh $flag > gpurun_out/build_dbg.log 2>&1 || { tail -5 gpurun_out/build_dbg.log; exit 1; }
echo "== $flag"
SEEME_DEBUG_NOCHECK=1 python scripts/den_times.py ${1:-fp16} 2>&1 | grep -v "^{" | tail -6
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
