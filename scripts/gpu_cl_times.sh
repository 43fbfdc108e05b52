# cycle stamps of k_den_cluster (debug library built HERE into seeme_amd/libseeme_hip_dbg.so with -DDEN_DBG_TIMES) + kernel time
out=gpurun_out/${1:-r3}; mkdir -p $out
export SEEME_HIP_LIB=$PWD/seeme_amd/libseeme_hip_dbg.so
cfgs=${2:-"8:0:0 8:1:0 4:0:0"}
for cfg in $cfgs; do
  IFS=: read c p f <<< "$cfg"
  SEEME_DEN_CLUSTER=$c SEEME_DEN_CLUSTER_PLACE=$p SEEME_DEN_CLUSTER_FLAGS=$f python scripts/cl_times.py fp16 ${3:-32} 2>&1 | grep -v amdgpu.ids
done > $out/times.txt
python - <<'PY' >> $out/times.txt
import os, sys, json, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_parity import make_den, _sched
dev = torch.device("cuda:0")
den = make_den(dev, weight_dtype="fp16"); sch = _sched(); sch.set_timesteps(50)
torch.manual_seed(5); lat = torch.randn(32, 1, 256, device=dev); cond = torch.randn(32, 1, 256, device=dev)
os.environ["SEEME_DEN_CLUSTER"] = "0"; base = den.sample_loop(lat, cond, sch)
for c, p in ((8, 0), (8, 1), (4, 0), (4, 1)):
    os.environ["SEEME_DEN_CLUSTER"] = str(c); os.environ["SEEME_DEN_CLUSTER_PLACE"] = str(p)
    z = den.sample_loop(lat, cond, sch); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        den.sample_loop(lat, cond, sch, events=ev); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    print(json.dumps({"C": c, "place": p, "ms": round(float(np.median(ts)), 4), "vs_onecu": float((z - base).abs().max() / base.abs().max()), "status": den.cluster_status()}))
PY
cat $out/times.txt
