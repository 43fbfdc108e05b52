# A/B of build flags on the default and B=256 benches (same box, back to back, twice)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do
for flags in "$@"; do
bash seeme_amd/csrc/build.sh $flags > gpurun_out/build_var.log 2>&1 || { tail -5 gpurun_out/build_var.log; exit 1; }
python bench.py --no-cpu-baseline > gpurun_out/b32.json 2>/dev/null; python bench.py --no-cpu-baseline --batch 256 > gpurun_out/b256.json 2>/dev/null
python -c "
import json
r=json.load(open('gpurun_out/b32.json')); q=json.load(open('gpurun_out/b256.json'))
print('flags [$flags]  B32', r['value'], r['ms_per_step'], ' B256', q['value'], q['ms_per_step'])
"
done
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
