cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -rf gpurun_out/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -- python bench.py --steps 5 --warmup 1 --batch ${1:-256} --no-cpu-baseline > gpurun_out/prof_b.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/prof_b/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'][:26]
    if k.startswith('k_') or 'den' in k:
        g = (int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))
        agg[(k, g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v)//6, 'calls/pass', round(sum(v)/len(v),1), 'us avg', round(sum(v)/6,1), 'us/pass'); tot += sum(v)/6
print('total kernel us/pass', round(tot,1))
PY
