# tuning probes of the sampling kernel (valid results)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for flag in ${FLAGS:-"-DDEN_SLEEP=2" "-DDEN_SLEEP=4" "-DDEN_SLEEP=8" "-DDEN_SLEEP=16"}; do
  bash seeme_amd/csrc/build.sh $flag > gpurun_out/build_dbg.log 2>&1 || { tail -5 gpurun_out/build_dbg.log; exit 1; }
  for w in ${WLIST:-fp16}; do
    timeout -k 10 120 python bench.py --steps 5 --warmup 1 --weights $w --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('[$flag]', '$w', 'den ms', r['roofline']['ms_per_launch'], 'pass ms', r['ms_per_step'])"
  done
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
