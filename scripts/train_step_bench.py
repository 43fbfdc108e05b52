"""Stage-2 training step on one GPU (BASELINE config 3 shape: scene + interactee conditions, B=64, T=196,
20 000 scene points): frozen PointNet + two VAE encodes on the HIP path, denoiser fwd+bwd through the autograd
twin, flat-bucket gradient all-reduce (no-op at world size 1), AdamW.  Prints per-phase device times."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
from seeme_amd.weights_recipe import load_recipe_
from seeme_amd import distributed as D

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=20000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--scene-precision", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--graph", action="store_true", help="capture the whole step as one hipGraph and time replays")
args = ap.parse_args()
rank, ws, local = D.init_from_env()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = parse_config(os.path.join(repo, "configs", "config_mld_scene.yaml"))
dm = SyntheticEgoDataModule(nfeats=75, T=196, n_points=args.points, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
load_recipe_(model.vae), load_recipe_(model.denoiser), load_recipe_(model.proscene.scene_enc)
model.proscene.scene_enc.precision = args.scene_precision
model = model.to(dev).train()
D.broadcast_parameters(model)
batch = dm.batch(args.batch, idx=rank, with_scene=True)
ev = lambda: torch.cuda.Event(enable_timing=True)
if args.graph:
    replay = model.capture_training_step(batch)
    torch.cuda.synchronize()
    for it in range(args.steps + 1):
        e0, e1 = ev(), ev()
        e0.record()
        loss = replay()
        e1.record()
        torch.cuda.synchronize()
        if it and rank == 0:
            print(json.dumps({"step": it, "loss": round(float(loss.detach()), 5), "graph_replay_ms": round(e0.elapsed_time(e1), 2), "B": args.batch,
                              "points": args.points, "world": ws, "scene_precision": args.scene_precision}))
    sys.exit(0)
for it in range(args.steps + 1):
    e = [ev() for _ in range(4)]
    e[0].record()
    scene = model._scene_token(batch[4])
    e[1].record()
    rs = model.train_diffusion_forward(batch)
    loss = model.losses["train"].update(rs)
    e[2].record()
    model.optimizer_step(loss)
    e[3].record()
    torch.cuda.synchronize()
    if it and rank == 0:
        print(json.dumps({"step": it, "loss": round(float(loss), 5), "pointnet_ms": round(e[0].elapsed_time(e[1]), 2),
                          "forward_ms(incl. pointnet again)": round(e[1].elapsed_time(e[2]), 2),
                          "bwd+allreduce+adamw_ms": round(e[2].elapsed_time(e[3]), 2), "B": args.batch,
                          "points": args.points, "world": ws, "scene_precision": args.scene_precision}))
