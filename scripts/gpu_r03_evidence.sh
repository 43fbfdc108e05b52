# Round-3 evidence run (one gpurun call): gpu tests, bench lines, rocprofv3 --kernel-trace --stats of the default command, steady-state
# kernel stats, PMC traffic of the sampling kernel (separate passes) for both cluster placements.  Summaries land in gpurun_out/<tag>/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && T=${1:-r03} && mkdir -p gpurun_out/$T && O=gpurun_out/$T
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --cluster 0 --no-cpu-baseline --no-parity-check > $O/bench_one_cu_per_sample.json 2>/dev/null
timeout -k 10 300 python bench.py --weights fp32 --vae fp32 --no-cpu-baseline --no-parity-check > $O/bench_fp32.json 2>/dev/null
timeout -k 10 300 python bench.py --weights bf16 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 64 --no-cpu-baseline --no-parity-check > $O/bench_B64.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 128 --no-cpu-baseline --no-parity-check > $O/bench_B128.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 256 --no-cpu-baseline --no-parity-check > $O/bench_B256.json 2>/dev/null
timeout -k 10 300 python bench.py --batch 512 --no-cpu-baseline --no-parity-check > $O/bench_B512.json 2>/dev/null
SEEME_DEN_CLUSTER_MS=0 timeout -k 10 300 python bench.py --batch 128 --no-cpu-baseline --no-parity-check > $O/bench_B128_no_ms.json 2>/dev/null
SEEME_DEN_CLUSTER_MS=0 timeout -k 10 300 python bench.py --batch 256 --no-cpu-baseline --no-parity-check > $O/bench_B256_no_ms.json 2>/dev/null
SEEME_DEN_CLUSTER_MS=0 timeout -k 10 300 python bench.py --batch 512 --no-cpu-baseline --no-parity-check > $O/bench_B512_no_ms.json 2>/dev/null
SEEME_DEN_CLUSTER_MS=0 timeout -k 10 300 python bench.py --scheduler ddpm --batch 512 --graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check > $O/bench_ddpm1000_B512_no_ms.json 2>/dev/null
timeout -k 10 300 python bench.py --scheduler ddpm --batch 512 --graph --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check > $O/bench_ddpm1000_B512.json 2>/dev/null
timeout -k 10 300 python bench.py --mode train --steps 20 > $O/bench_train_scene.json 2>/dev/null
timeout -k 10 300 python bench.py --mode train --train-config gimo --steps 20 > $O/bench_train_gimo.json 2>/dev/null
SEEME_BENCH_BACKEND=gloo SEEME_BENCH_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-parity-check > $O/bench_gloo2.json 2>/dev/null
SEEME_BENCH_BACKEND=gloo SEEME_BENCH_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --mode train --steps 5 > $O/bench_train_gloo2.json 2>/dev/null
# rocprofv3 --kernel-trace --stats of the default command (whole process)
rm -rf gpurun_out/rp_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_stats -- python bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2> $O/rp_stats.err
cp $(ls gpurun_out/rp_stats/*/*kernel_stats.csv | head -1) $O/rocprof_kernel_stats_default.csv; rm -rf gpurun_out/rp_stats
# steady-state window of the sampling pass and of the training step
rm -rf gpurun_out/kt_s gpurun_out/kt_t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check > $O/kt_sample.log 2>&1
python scripts/steady_stats.py kt_s k_den_cluster 3 10 $O/kernel_stats_sample_steady.csv
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_t -- python bench.py --mode train --steps 10 --warmup 3 > $O/kt_train.log 2>&1
python scripts/steady_stats.py kt_t k_adamw 3 10 $O/kernel_stats_train_steady.csv
rm -rf gpurun_out/kt_s gpurun_out/kt_t
# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the sampling kernel: both cluster placements, and one CU per sample
for cfg in "place1:SEEME_DEN_CLUSTER_PLACE=1" "place0:SEEME_DEN_CLUSTER_PLACE=0" "onecu:SEEME_DEN_CLUSTER=0"; do
  tag=${cfg%%:*}; kv=${cfg#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    export ${kv?}
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check > gpurun_out/pmc_$c.log 2>&1
    unset ${kv%%=*}
  done
  python - "$tag" <<'PY' >> $O/traffic_sampling_fp16.txt
import csv, glob, collections, sys
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"][:40]; acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        out.setdefault(k, {})[c] = (acc[k] / n[k], n[k])
for k, d in out.items():
    if "k_den_" in k:
        fe = d.get("FETCH_SIZE", (0, 0))[0] * 1024 * 2     # KiB -> B, gfx950 x2 wide-read correction
        wr = d.get("WRITE_SIZE", (0, 0))[0] * 1024
        print(f"{sys.argv[1]:8s} {k:42s} per launch: fetch {fe/1e6:10.2f} MB (x2 corrected)  write {wr/1e6:8.2f} MB  total {(fe+wr)/1e6:10.2f} MB  launches {d.get('FETCH_SIZE',(0,0))[1]}")
PY
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
done
tail -3 $O/pytest_gpu.log; cut -c1-330 $O/bench_default.json; cat $O/traffic_sampling_fp16.txt; cut -c1-200 $O/bench_train_scene.json; head -5 $O/rocprof_kernel_stats_default.csv
