"""Timing only: 50-step DDIM launch of the large-batch cluster kernel per batch size, for the placement / flags in the environment."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_parity import make_den, _sched
dev = torch.device("cuda:0")
den = make_den(dev, weight_dtype=os.environ.get("WD", "fp16"))
sch = _sched(); sch.set_timesteps(50)
torch.manual_seed(7)
lat = torch.randn(512, 1, 256, device=dev); cond = torch.randn(512, 1, 256, device=dev)
for B in [int(x) for x in (sys.argv[1:] or ["64", "128", "256", "512"])]:
    l, c = lat[:B].contiguous(), cond[:B].contiguous()
    for _ in range(2):
        den.sample_loop(l, c, sch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        den.sample_loop(l, c, sch, events=ev)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(json.dumps({"B": B, "plan": den._cluster_plan(B, 1, False, False), "place": os.environ.get("SEEME_DEN_CLUSTER_PLACE"), "flags": os.environ.get("SEEME_DEN_CLUSTER_FLAGS"),
                      "wd": os.environ.get("WD", "fp16"), "lib": os.path.basename(os.environ.get("SEEME_HIP_LIB", "")), "ms": round(float(np.median(ts)), 4), "us_per_step": round(float(np.median(ts)) * 20, 2),
                      "status": den.cluster_status()}), flush=True)
