# PointNet block kernels: matrix-core busy fraction, clock during the kernel, and where the waves' cycles go
# (separate PMC-only passes, program directly after `--`)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -rf gpurun_out/pmc_pnb1 gpurun_out/pmc_pnb2 gpurun_out/pmc_pnb3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_pnb1 -- python scripts/pn_bench.py > gpurun_out/pmc_pnb1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_pnb2 -- python scripts/pn_bench.py > gpurun_out/pmc_pnb2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_pnb3 -- python scripts/pn_bench.py > gpurun_out/pmc_pnb3.log 2>&1
python - <<PY
import csv, glob, collections
def load(tag):
    fs = glob.glob(f"gpurun_out/{tag}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    if not fs: print(tag, "no counters"); return acc, n
    first = None
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:30]
        if "k_pn" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        first = first or r["Counter_Name"]
        if r["Counter_Name"] == first: n[k] += 1
    return acc, n
a, n = load("pmc_pnb1")
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_pnb1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pn" in r["Kernel_Name"]: dur[r["Kernel_Name"][:30]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in a.items():
    act = d["GRBM_GUI_ACTIVE"] / 8 / max(n[k], 1)
    us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    print(f"{k:32s} launches {n[k]} cycles/launch {act:.0f} us/launch(pmc run) {us:.1f} clock GHz {act / us / 1e3 if us else 0:.2f} MFMA util {100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %")
a, n = load("pmc_pnb2")
names = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"]
for k, d in a.items():
    w = d.get("SQ_WAVE_CYCLES", 0)
    if w: print(f"{k:32s}", " ".join(f"{x[3:]}={100 * d.get(x, 0) / w:5.1f}%" for x in names))
a, n = load("pmc_pnb3")
for k, d in a.items():
    print(f"{k:32s}", " ".join(f"{x[3:]}={v / max(n[k], 1):.3g}" for x, v in d.items()))
PY
