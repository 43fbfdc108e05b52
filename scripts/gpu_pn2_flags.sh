# A/B of build flags on the PointNet bf16 encode (B=64 x 20000 points): scripts/pn_bench.py per flag set, twice.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do
for flags in "$@"; do
bash seeme_amd/csrc/build.sh $flags > gpurun_out/build_var.log 2>&1 || { tail -5 gpurun_out/build_var.log; exit 1; }
echo "flags [$flags] $(timeout -k 10 120 python scripts/pn_bench.py 2>/dev/null)"
done
done
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
