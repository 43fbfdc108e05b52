# Training step (cfg3) with the denoiser forward / backward workgroups packed onto 8 / 4 / 2 XCDs.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for rep in 1 2; do for k in 8 0 4 2; do
SEEME_DEN_XCDS=$k python bench.py --mode train --steps 20 > gpurun_out/xcds_t.json 2>/dev/null
python -c "
import json; r=json.load(open('gpurun_out/xcds_t.json')); print('xcds $k (0 = default policy) train step', r['value'], r['ms_per_step'])"
done; done
