# Sampling kernel with its workgroups packed onto 1 / 2 / 4 / 8 XCDs (SEEME_DEN_XCDS): pass time at B=32 and B=64.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do for k in 8 4 2 1; do
for b in 32 64; do
SEEME_DEN_XCDS=$k python bench.py --no-cpu-baseline --no-parity-check --batch $b > gpurun_out/xcds.json 2>/dev/null
python -c "
import json; r=json.load(open('gpurun_out/xcds.json')); print('xcds $k B $b', r['value'], r['ms_per_step'], 'kernel ms', r['roofline'].get('kernel_ms'), 'per_cu', r['roofline'].get('per_cu_stream'))"
done; done; done
