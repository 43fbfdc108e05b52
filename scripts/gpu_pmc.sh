cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
W=${1:-bf16}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$W -- python bench.py --steps 2 --warmup 1 --weights $W --no-cpu-baseline > gpurun_out/pmc_$W.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_$W/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in acc.items():
        if "den_sample" in k or "k_linear" in k or "attn" in k or "ffn" in k:
            print(k, {c: round(v/1e6, 2) for c, v in d.items()})
PY
