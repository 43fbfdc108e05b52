cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
true
python bench.py --steps 10 --warmup 2 --weights fp32 --vae fp32 --no-cpu-baseline > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err
python bench.py --steps 10 --warmup 2 --weights bf16 --vae fp32 --no-cpu-baseline > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
python bench.py --steps 10 --warmup 2 --weights fp16 --vae fp16 --no-cpu-baseline > gpurun_out/bench_fp16.json 2> gpurun_out/bench_fp16.err
python bench.py --steps 10 --warmup 2 --weights fp16 --vae fp16 --batch 256 --no-cpu-baseline > gpurun_out/bench_fp16_b256.json 2> gpurun_out/bench_fp16_b256.err
cat gpurun_out/bench_fp32.json gpurun_out/bench_bf16.json gpurun_out/bench_fp16.json gpurun_out/bench_fp16_b256.json | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['dtype'], r['config']['batch_per_gpu'], 'seqs/s', r['value'], 'ms/pass', r['ms_per_step'], 'den ms', r['roofline']['ms_per_launch'], 'GB/s', r['roofline']['achieved'])
"
