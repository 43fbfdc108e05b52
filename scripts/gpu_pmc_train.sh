cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -rf gpurun_out/pmc_train
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_train -- python scripts/train_step_bench.py --steps 1 --scene-precision bf16 > gpurun_out/pmc_train.log 2>&1
tail -2 gpurun_out/pmc_train.log
python - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_train/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:34]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "pn_block" in k:
            print(k, {c: round(v / n[(k, c)] / 1e6, 2) for c, v in d.items()})
PY
