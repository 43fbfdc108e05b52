# per-kernel durations of the PointNet forward for build variants given as arguments (quoted flag strings)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for flags in "$@"; do
bash seeme_amd/csrc/build.sh $flags > gpurun_out/build_var.log 2>&1 || { tail -5 gpurun_out/build_var.log; exit 1; }
rm -rf gpurun_out/prof_pn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pn -- python scripts/pn_times.py > gpurun_out/prof_pn.log 2>&1
echo "== flags: $flags   $(grep 'forward ms' gpurun_out/prof_pn.log)"
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/prof_pn/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'][:40]
    if 'k_pn' in k or 'k_lin' in k:
        g = (int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))
        agg[(k, g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(' ', k, len(v), 'calls', round(sum(v)/len(v),1), 'us avg', round(sum(v)/13,1), 'us/forward')
PY
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
