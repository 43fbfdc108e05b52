# GPU box: steady-state kernel stats of the sampling pass (B=32 and B=256) and matrix-core utilisation of the VAE kernels at B=256 / 512.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && T=${1:-r02c} && mkdir -p gpurun_out/$T && O=gpurun_out/$T
for B in 32 256; do
  rm -rf gpurun_out/kt_s
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_s -- python bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline --no-parity-check > $O/kt_sample_B$B.log 2>&1
  python scripts/steady_stats.py kt_s k_den_sample 3 10 $O/kernel_stats_sample_steady_B$B.csv
done
rm -rf gpurun_out/kt_s
for B in 256 512; do
  rm -rf gpurun_out/pmc_mfma_bench
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_bench -- python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-parity-check > $O/pmc_mfma_B$B.log 2>&1
  python - $B $O <<PY
import csv, glob, collections, json, sys
B, O = sys.argv[1], sys.argv[2]
fs = glob.glob("gpurun_out/pmc_mfma_bench/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
res = {}
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
    if busy > 0 and act > 0:
        util = busy / (act / 8 * 256 * 4)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs (profiles/r01_e_mfma_util.json)
        res[k] = {"mfma_busy_cycles": busy, "gui_active_cycles": act, "launches": n[k], "mfma_util": round(util, 4)}
        print(f"B={B} {k:42s} launches {n[k]:4d}  MFMA util {100 * util:6.2f} %")
json.dump(res, open(f"{O}/mfma_util_B{B}.json", "w"), indent=1)
PY
done
rm -rf gpurun_out/pmc_mfma_bench
