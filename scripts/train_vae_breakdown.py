"""Debug: stage-1 (VAE) training step at B=64, T=196: device time of forward / backward / optimizer, and the top kernels."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda", 0)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = parse_config(os.path.join(repo, "configs", "config_vae_egobody.yaml"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dm = SyntheticEgoDataModule(nfeats=75, T=196, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
load_recipe_(model.vae)
model = model.to(dev).train()
batch = dm.batch(B, idx=0)
model.configure_optimizers()
ev = lambda: torch.cuda.Event(enable_timing=True)
def step(rec=None):
    e = [ev() for _ in range(4)]
    e[0].record()
    rs = model.train_vae_forward(batch)
    loss = model.losses["train"].update(rs)
    e[1].record()
    model.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    e[2].record()
    model.optimizer_update()
    e[3].record()
    torch.cuda.synchronize()
    return {"fwd_ms": round(e[0].elapsed_time(e[1]), 2), "backward_ms": round(e[1].elapsed_time(e[2]), 2), "adamw_ms": round(e[2].elapsed_time(e[3]), 2)}
for it in range(5):
    r = step()
print(json.dumps(r))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU":
        agg[e.name[:70]][0] += 1; agg[e.name[:70]][1] += e.device_time
tot = sum(v[1] for v in agg.values())
print("device kernel time ms", round(tot / 1e3, 2), "kernels", sum(v[0] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{v[1] / 1e3:8.3f} ms {v[0]:5d}x  {k}")
