# Timing-only ablations of k_layer_h (cycle stamps of one workgroup, 4-wave instantiation): full kernel, no B-fragment re-fills,
# no re-fills and no A-fragment reads; then the 64-row instantiation (full).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for flags in "" "-DH16_DBG_NOREFILL" "-DH16_DBG_NOREFILL -DH16_DBG_NOAREAD"; do
echo "== flags [$flags] B=${H16_B:-32}"
bash seeme_amd/csrc/build.sh -DH16_DBG_TIMES -DH16_DBG_KERNEL=5 $flags > gpurun_out/build_dbg.log 2>&1 || { tail -5 gpurun_out/build_dbg.log; exit 1; }
SEEME_LAYER_W8_MAX=0 timeout -k 10 300 python scripts/h16_times.py 5 2>&1 | grep -v "^{" | grep -v amdgpu.ids | tail -14
done
echo "== 64-row instantiation, full, B=${H16_B:-32}"
bash seeme_amd/csrc/build.sh -DH16_DBG_TIMES -DH16_DBG_KERNEL=5 > gpurun_out/build_dbg.log 2>&1
SEEME_LAYER_ROWS=64 timeout -k 10 300 python scripts/h16_times.py 5 2>&1 | grep -v "^{" | grep -v amdgpu.ids | tail -14
bash seeme_amd/csrc/build.sh > /dev/null 2>&1
