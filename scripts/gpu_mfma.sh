# Matrix-core utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs) (the gfx94x MfmaUtil
# formula; ROCm 7.2 ships no derived counters for gfx950).  Separate PMC-only passes, program directly after `--`.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -rf gpurun_out/pmc_mfma_bench gpurun_out/pmc_mfma_train
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_bench -- python bench.py --steps 2 --warmup 1 --batch ${1:-256} --no-cpu-baseline > gpurun_out/pmc_mfma_bench.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_train -- python scripts/train_breakdown.py vae16 > gpurun_out/pmc_mfma_train.log 2>&1
python - <<PY
import csv, glob, collections, json
res = {}
for tag in ("bench", "train"):
    fs = glob.glob(f"gpurun_out/pmc_mfma_{tag}/*/*counter_collection.csv")
    if not fs:
        print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
        if busy > 0 and act > 0:
            util = busy / (act / 8 * 256 * 4)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs (see profiles/r01_e_mfma_util.json)
            res[f"{tag}:{k}"] = {"mfma_busy_cycles": busy, "gui_active_cycles": act, "launches": n[k], "mfma_util": round(util, 4)}
            print(f"{tag:6s} {k:42s} launches {n[k]:4d}  MFMA util {100 * util:6.2f} %")
json.dump(res, open("gpurun_out/mfma_util.json", "w"), indent=1)
PY
