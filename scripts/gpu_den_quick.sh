# quick loop for the sampling kernel: denoiser/DDIM parity tests + bench lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -s -k "denoiser or ddim or ddpm or sample or multihead or extreme or weights or variants or pairs" > gpurun_out/tests_den.log 2>&1; grep -E "rel err|MPJPE|passed|failed|Error|error" gpurun_out/tests_den.log | tail -12
for w in fp16 fp32; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --weights $w --no-cpu-baseline 2>gpurun_out/bench_q.err | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$w', 'seqs/s', r['value'], 'pass ms', r['ms_per_step'], 'den ms', r['roofline']['ms_per_launch'])"
done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --batch 512 --scheduler ddpm --no-cpu-baseline 2>gpurun_out/bench_q.err | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('ddpm1000 B=512', 'seqs/s', r['value'], 'pass ms', r['ms_per_step'], 'den ms', r['roofline']['ms_per_launch'])"
