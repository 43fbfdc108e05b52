cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -rf gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python scripts/train_breakdown.py vae16 > gpurun_out/prof_train.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/prof_train/*/*kernel_stats.csv')[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < 14: print(r['Name'][:60], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'pct', r['Percentage'])
PY
