# per-kernel times (rocprofv3 --kernel-trace --stats) of scripts/pn_bench.py for prebuilt library variants.  usage: gpu_pn2_libs_prof.sh <name>...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for n in "$@"; do
export SEEME_HIP_LIB=$PWD/probes/variants/libseeme_$n.so
rm -rf gpurun_out/pnprof_$n
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pnprof_$n -o p -- python3 scripts/pn_bench.py > gpurun_out/pnprof_$n.log 2>&1 || { tail -5 gpurun_out/pnprof_$n.log; exit 1; }
f=$(find gpurun_out/pnprof_$n -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv for $n"; exit 1; }
echo "== $n" | tee -a gpurun_out/pn2_libs_prof.txt; grep -E "k_pn_block2|Name" "$f" < /dev/null | cut -d, -f1-8 | tee -a gpurun_out/pn2_libs_prof.txt
done
