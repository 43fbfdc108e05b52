# HBM-side traffic of the dominant kernel (separate PMC passes, MI355X_MICROARCH.md "HBM" section):
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> x2.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
W=${1:-bf16}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --weights $W --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"][:28]; acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        out.setdefault(k, {})[c] = (acc[k] / n[k], n[k])
for k, d in out.items():
    if k.startswith("k_") or "den" in k:
        fe = d.get("FETCH_SIZE", (0, 0))[0] * 1024 * 2     # KiB -> B, gfx950 x2 correction
        wr = d.get("WRITE_SIZE", (0, 0))[0] * 1024
        print(f"{k:30s} per launch: fetch {fe/1e6:10.2f} MB (corrected x2)  write {wr/1e6:8.2f} MB  launches {d.get('FETCH_SIZE',(0,0))[1]}")
PY
