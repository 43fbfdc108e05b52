# HBM-side traffic of the PointNet block kernels (separate PMC passes; FETCH_SIZE / WRITE_SIZE in KiB, gfx950 x2 on FETCH_SIZE)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_pn_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_pn_$c -- python scripts/pn_times.py > gpurun_out/pmc_pn_$c.log 2>&1
done
python - <<PY
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_pn_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"][:34]; acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        out.setdefault(k, {})[c] = (acc[k] / n[k], n[k])
for k, d in out.items():
    if "k_pn" in k:
        fe = d.get("FETCH_SIZE", (0, 0))[0] * 1024 * 2
        wr = d.get("WRITE_SIZE", (0, 0))[0] * 1024
        print(f"{k:36s} per launch: fetch {fe/1e6:10.2f} MB (corrected x2)  write {wr/1e6:8.2f} MB  launches {d.get('FETCH_SIZE',(0,0))[1]}")
PY
