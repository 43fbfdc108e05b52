"""GPU check of k_den_cluster with TWO condition tokens (ca_block query / proj_out stages, third exchange per layer): the reference
fixture denoiser_N2.npz (fp32), and the one-CU kernel on the same weight image dtype for every C / placement; timing at B = 32."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden, rel_err
from test_gpu_parity import make_den, _sched
dev = torch.device("cuda:0")
g = load_golden("denoiser_N2.npz")
ok_all = True
for wd in ("fp32", "fp16", "bf16"):
    den = make_den(dev, cond=("text", "scene", "interactee"), weight_dtype=wd)
    s, c = torch.from_numpy(g["sample"]).to(dev), torch.from_numpy(g["cond"]).to(dev)
    sch = _sched(); sch.set_timesteps(50)
    torch.manual_seed(3)
    lat, cond = torch.randn(32, 1, 256, device=dev), torch.randn(32, 2, 256, device=dev)
    den.cluster = 0
    base = den.sample_loop(lat, cond, sch)
    for Cc in (8, 4, 2):
        den.cluster = Cc
        errs = {}
        for t in (981, 501, 1):
            y = den(sample=s, timestep=torch.tensor(t), encoder_hidden_states=c)[0]
            errs[t] = rel_err(y.cpu().numpy(), g[f"out_t{t}"])
        yv = den(sample=s, timestep=torch.from_numpy(g["tvec"]).to(dev), encoder_hidden_states=c)[0]
        errs["tvec"] = rel_err(yv.cpu().numpy(), g["out_tvec"])
        for place in (0, 1):
            den.cluster_placement = place
            z = den.sample_loop(lat, cond, sch); z2 = den.sample_loop(lat, cond, sch)
            torch.cuda.synchronize()
            st = den.cluster_status()
            rec = {"wd": wd, "C": Cc, "place": place, "fwd_vs_reference": {str(k): float(v) for k, v in errs.items()},
                   "ddim50_vs_onecu": rel_err(z.cpu().numpy(), base.cpu().numpy()), "bit_identical_rerun": bool(torch.equal(z, z2)), "status": st}
            print(json.dumps(rec), flush=True)
            if st[0] or not rec["bit_identical_rerun"] or (wd == "fp32" and (max(errs.values()) > 1e-4 or rec["ddim50_vs_onecu"] > 1e-5)):
                ok_all = False
    if wd == "fp16":
        for Cc in (0, 8, 4):
            den.cluster = Cc; den.cluster_placement = 1
            for _ in range(3): den.sample_loop(lat, cond, sch)
            ts = []
            for _ in range(10):
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                den.sample_loop(lat, cond, sch, events=ev); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
            print(json.dumps({"time_ddim50_B32_N2": wd, "C": Cc, "ms": round(float(np.median(ts)), 4)}), flush=True)
print("ALL_OK" if ok_all else "FAILURES")
