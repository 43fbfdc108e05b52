"""Timing: 50-step DDIM launch with TWO condition tokens (scene + interactee, the shipped config_mld_egobody.yaml) per batch size."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_parity import make_den, _sched
dev = torch.device("cuda:0")
den = make_den(dev, cond=("text", "scene", "interactee"), weight_dtype=os.environ.get("WD", "fp16"))
sch = _sched(); sch.set_timesteps(50)
torch.manual_seed(3)
for B in [int(x) for x in (sys.argv[1:] or ["32", "64", "128"])]:
    lat = torch.randn(B, 1, 256, device=dev); cond = torch.randn(B, 2, 256, device=dev)
    outs = {}
    for cl in (0, "auto_no_ms", "auto"):
        den.cluster = "auto" if cl == "auto_no_ms" else cl
        den.cluster_ms = cl != "auto_no_ms"
        for _ in range(2):
            den.sample_loop(lat, cond, sch)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            outs[cl] = den.sample_loop(lat, cond, sch, events=ev); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
        print(json.dumps({"N": 2, "B": B, "cluster": cl, "plan": den._cluster_plan(B, 2, False, False) if cl != 0 else [0, 1],
                          "ms": round(float(np.median(ts)), 4), "status": den.cluster_status(),
                          "vs_one_cu": float((outs[cl] - outs[0]).abs().max() / outs[0].abs().max()) if 0 in outs else None}), flush=True)
