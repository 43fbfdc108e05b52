"""Debug: where the stage-2 step spends its time (device events between phases)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda", 0)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = parse_config(os.path.join(repo, "configs", "config_mld_scene.yaml"))
if "twin" in sys.argv[1:]:
    cfg.TRAIN.HIP_BACKWARD = False
if "vae16" in sys.argv[1:]:
    cfg.TRAIN.FROZEN_VAE_PRECISION = "fp16"
dm = SyntheticEgoDataModule(nfeats=75, T=196, n_points=20000, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
load_recipe_(model.vae), load_recipe_(model.denoiser), load_recipe_(model.proscene.scene_enc)
model.proscene.scene_enc.precision = "bf16"
model = model.to(dev).train()
batch = dm.batch(64, idx=0, with_scene=True)
model.configure_optimizers()
ev = lambda: torch.cuda.Event(enable_timing=True)
import time
for it in range(6):
    e = [ev() for _ in range(5)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e[0].record()
    rs = model.train_diffusion_forward(batch)
    loss = model.losses["train"].update(rs)
    e[1].record()
    model.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    e[2].record()
    model.optimizer_update()
    e[3].record()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if it >= 3:
        print(json.dumps({"fwd_ms": round(e[0].elapsed_time(e[1]), 2), "backward_ms": round(e[1].elapsed_time(e[2]), 2),
                          "adamw_ms": round(e[2].elapsed_time(e[3]), 2), "wall_ms": round((t1 - t0) * 1e3, 2)}))

if "prof" in sys.argv[1:]:
    import collections
    from torch.profiler import profile, ProfilerActivity
    rs = model.train_diffusion_forward(batch)
    loss = model.losses["train"].update(rs)
    model.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        model.optimizer_update()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type.name != "CPU":
            agg[e.name[:80]][0] += 1; agg[e.name[:80]][1] += e.device_time
    print("optimizer.step device kernels:", sum(v[0] for v in agg.values()), "total us", round(sum(v[1] for v in agg.values()), 1))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {v[1]:8.1f} us {v[0]:4d}x {k}")
    print("trainable tensors", len(model.trainable_parameters()), "elements", sum(p.numel() for p in model.trainable_parameters()))

if "proffwd" in sys.argv[1:]:
    import collections
    from torch.profiler import profile, ProfilerActivity
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        rs = model.train_diffusion_forward(batch)
        loss = model.losses["train"].update(rs)
        torch.cuda.synchronize()
    fw = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type.name != "CPU":
            fw[e.name[:70]][0] += 1; fw[e.name[:70]][1] += e.device_time
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        model.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
    print("forward device kernels:", sum(v[0] for v in fw.values()), "total us", round(sum(v[1] for v in fw.values()), 1))
    for k, v in sorted(fw.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {v[1]:8.1f} us {v[0]:4d}x {k}")
    print("backward:")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type.name != "CPU":
            agg[e.name[:70]][0] += 1; agg[e.name[:70]][1] += e.device_time
    print("backward device kernels:", sum(v[0] for v in agg.values()), "total us", round(sum(v[1] for v in agg.values()), 1))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"  {v[1]:8.1f} us {v[0]:4d}x {k}")

if "lines" in sys.argv[1:]:
    import collections
    from torch.profiler import profile, ProfilerActivity
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        rs = model.train_diffusion_forward(batch)
        loss = model.losses["train"].update(rs)
        model.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        model.optimizer_update()
        torch.cuda.synchronize()
    # CPU-side op events carry the Python stack; attribute every launch (cuda_time of the op) to the innermost repo frame
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type.name == "CPU" and e.stack and e.device_time > 0 and not e.cpu_children:
            fr = [s for s in e.stack if "/seeme_amd/" in s or "train_breakdown" in s]
            key = fr[0].split("/")[-1][:70] if fr else "(autograd engine / other)"
            agg[key][0] += 1; agg[key][1] += e.device_time
    print("leaf ops with device time, by innermost repo frame:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  {v[1]:8.1f} us {v[0]:4d} ops  {k}")
