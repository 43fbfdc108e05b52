"""GPU: stage-1 (VAE) training step at B=64, T=196 (config_vae_egobody): ms per step with the hand-written HIP backward and
with the PyTorch-autograd twin, phases from HIP events, and the top kernels of the HIP step."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda", 0)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res = {}
for hip in (True, False):
    cfg = parse_config(os.path.join(repo, "configs", "config_vae_egobody.yaml"))
    cfg.TRAIN.HIP_VAE_BACKWARD = hip
    dm = SyntheticEgoDataModule(nfeats=75, T=196, device=dev)
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae)
    model = model.to(dev).train()
    batch = dm.batch(B, idx=0)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    def step():
        e = [ev() for _ in range(3)]
        e[0].record()
        loss = model.training_step(batch)
        e[1].record()
        model.optimizer_step(loss)
        e[2].record()
        return e
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    evs = [step() for _ in range(10)]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    fwd = sum(e[0].elapsed_time(e[1]) for e in evs) / 10
    bwd = sum(e[1].elapsed_time(e[2]) for e in evs) / 10
    res["hip" if hip else "autograd_twin"] = {"ms_per_step": round(wall, 3), "forward_and_losses_ms": round(fwd, 3), "backward_allreduce_adamw_ms": round(bwd, 3)}
    if hip:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        agg = collections.defaultdict(lambda: [0, 0.0])
        for e in prof.events():
            if e.device_type.name != "CPU":
                agg[e.name[:60]][0] += 1; agg[e.name[:60]][1] += e.device_time
        tot = sum(v[1] for v in agg.values())
        res["hip"]["kernel_ms"] = round(tot / 1e3, 3)
        res["hip"]["launches"] = sum(v[0] for v in agg.values())
        res["hip"]["top_kernels"] = [[k, v[0], round(v[1] / 1e3, 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]]
    del model
print(json.dumps({"workload": f"config_vae_egobody stage-1 step, B={B}, T=196", **res}, indent=1))
