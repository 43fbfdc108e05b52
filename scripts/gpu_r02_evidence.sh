# Round-2 evidence run (one gpurun call): tests, bench lines, steady-state kernel stats (rocprofv3 --kernel-trace), the
# per-CU stream probe, PointNet PMC counters.  Summaries land in gpurun_out/r02/ (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && T=${1:-r02} && mkdir -p gpurun_out/$T && O=gpurun_out/$T
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --weights fp32 --vae fp32 --no-cpu-baseline > $O/bench_fp32.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 256 --no-cpu-baseline > $O/bench_B256.json 2>/dev/null
timeout -k 10 300 python bench.py --mode train --steps 20 > $O/bench_train_scene.json 2>/dev/null
timeout -k 10 300 python bench.py --mode train --train-config gimo --steps 20 > $O/bench_train_gimo.json 2>/dev/null
SEEME_BENCH_BACKEND=gloo SEEME_BENCH_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_gloo2.json 2>/dev/null
SEEME_BENCH_BACKEND=gloo SEEME_BENCH_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --mode train --steps 5 > $O/bench_train_gloo2.json 2>/dev/null
bash probes/build.sh > $O/probes_build.log 2>&1 && timeout -k 10 300 python scripts/stream_probe3.py > $O/stream_probe3.txt 2>&1
rm -rf gpurun_out/kt_s gpurun_out/kt_t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check > $O/kt_sample.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_t -- python bench.py --mode train --steps 10 --warmup 3 > $O/kt_train.log 2>&1
python - <<PY
import csv, glob, collections, json
def steady(tag, marker, skip, steps, out):
    fs = glob.glob(f"gpurun_out/{tag}/*/*kernel_trace.csv")
    if not fs: print(tag, "no trace"); return
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    # steady state = from the first launch after the warm-up passes' last marker kernel to the last marker kernel of the timed passes
    per_pass = len(marks) // (skip + steps) if marks else 0
    if not per_pass: print(tag, "marker not found"); return
    lo, hi = marks[skip * per_pass - 1] + 1, marks[-1]
    # extend to the end of the last pass: everything up to the next host gap is the decode of that pass -- keep through the last kernel before teardown
    agg = collections.defaultdict(lambda: [0, 0.0])
    t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["End_Timestamp"])
    for r in rows[lo:hi + 1]:
        k = r["Kernel_Name"].split("(")[0][:80]
        agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "calls_per_step", "total_us", "avg_us", "pct_of_kernel_time"])
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, v[0], round(v[0] / steps, 2), round(v[1], 1), round(v[1] / v[0], 2), round(100 * v[1] / tot, 2)])
        w.writerow(["# steady-state window", f"{steps} steps", f"launches/step {sum(v[0] for v in agg.values()) / steps:.1f}", f"kernel time/step {tot / steps:.1f} us", f"wall/step {(t1 - t0) / 1e3 / steps:.1f} us", ""])
    print(tag, "launches/step", round(sum(v[0] for v in agg.values()) / steps, 1), "kernel us/step", round(tot / steps, 1), "wall us/step", round((t1 - t0) / 1e3 / steps, 1))
steady("kt_s", "k_den_sample", 3, 10, "gpurun_out/$T/kernel_stats_sample_steady.csv")
steady("kt_t", "k_adamw", 3, 10, "gpurun_out/$T/kernel_stats_train_steady.csv")
PY
rm -rf gpurun_out/kt_s gpurun_out/kt_t
timeout -k 10 400 bash scripts/gpu_pmc_pn2.sh > $O/pmc_pointnet_v2.log 2>&1; rm -rf gpurun_out/pmc_pnb1 gpurun_out/pmc_pnb2 gpurun_out/pmc_pnb3 gpurun_out/pmc_pnb*.log
tail -4 $O/pytest_gpu.log; cat $O/bench_default.json | cut -c1-400; cat $O/bench_train_scene.json | cut -c1-300
