cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -s -k "vae or mld or smoke" > gpurun_out/tests_vae.log 2>&1; grep -E "rel err|passed|failed|Error|error" gpurun_out/tests_vae.log | tail -8
for b in 32 256; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --batch $b --no-cpu-baseline 2>gpurun_out/bench_q.err | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('B=$b', 'seqs/s', r['value'], 'pass ms', r['ms_per_step'], 'den ms', r['roofline']['ms_per_launch'])"
done
