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
    model.optimizer.step()
    e[3].record()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if it >= 3:
        print(json.dumps({"fwd_ms": round(e[0].elapsed_time(e[1]), 2), "backward_ms": round(e[1].elapsed_time(e[2]), 2),
                          "adamw_ms": round(e[2].elapsed_time(e[3]), 2), "wall_ms": round((t1 - t0) * 1e3, 2)}))
