# A/B of prebuilt library variants (probes/variants/libseeme_<name>.so, built here with SEEME_BUILD_OUT=... build.sh <flags>) on the PointNet
# bf16 encode (B=64 x 20000 points): scripts/pn_bench.py per variant, twice.   usage: gpu_pn2_libs.sh <name>...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do
for n in "$@"; do
echo "variant $n $(SEEME_HIP_LIB=$PWD/probes/variants/libseeme_$n.so timeout -k 10 120 python scripts/pn_bench.py 2>/dev/null)" | tee -a gpurun_out/pn2_libs.txt
done
done
