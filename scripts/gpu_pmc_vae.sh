# where the waves of the VAE tile kernels spend their cycles (SQ counters, quad-cycles; separate PMC-only pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -rf gpurun_out/pmc_vae
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_vae -- python bench.py --steps 2 --warmup 1 --batch ${1:-256} --no-cpu-baseline > gpurun_out/pmc_vae.log 2>&1
python - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_vae/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    acc[r["Kernel_Name"][:34]][r["Counter_Name"]] += float(r["Counter_Value"])
names = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"]
for k, d in acc.items():
    w = d.get("SQ_WAVE_CYCLES", 0)
    if w > 0 and (k.startswith("k_") or "den" in k):
        print(f"{k:36s}", " ".join(f"{n[3:]}={100*d.get(n,0)/w:5.1f}%" for n in names))
PY
