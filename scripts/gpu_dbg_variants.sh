# timing probes of the sampling kernel (debug builds on the GPU box; results of these builds are garbage by design)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export SEEME_DEBUG_NOCHECK=1
for flag in "" "-DDEN_DBG_NOFMA" "-DDEN_DBG_NOEPI" "-DDEN_DBG_NOLOAD" "-DDEN_DBG_NOFMA -DDEN_DBG_NOEPI"; do
  bash seeme_amd/csrc/build.sh $flag > gpurun_out/build_dbg.log 2>&1 || { tail -5 gpurun_out/build_dbg.log; exit 1; }
  for w in ${WLIST:-fp16 fp32}; do
    timeout -k 10 120 python bench.py --steps 5 --warmup 1 --weights $w --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('[$flag]', '$w', 'den ms', r['roofline']['ms_per_launch'], 'pass ms', r['ms_per_step'])"
  done
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
