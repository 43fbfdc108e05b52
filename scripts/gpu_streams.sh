cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for st in 1 2 4 8; do
  timeout -k 10 200 python bench.py --steps 48 --warmup 8 --streams $st --graph --no-cpu-baseline 2>gpurun_out/bench_q.err | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('graph streams=$st', 'seqs/s', r['value'], 'ms/pass', r['ms_per_step'], 'den ms', r['roofline']['ms_per_launch'])" || tail -3 gpurun_out/bench_q.err
done
