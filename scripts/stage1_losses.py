"""Debug: stage-1 (VAE) training on a fixed batch with a fixed objective: loss trajectory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
dev = torch.device("cuda", 0)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = parse_config(os.path.join(repo, "configs", "config_vae_egobody.yaml"))
cfg.TRAIN.OPTIM.LR = float(os.environ.get("LR", 1e-3))
torch.manual_seed(1234)
dm = SyntheticEgoDataModule(nfeats=75, T=24, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234)).to(dev).train()
batch = dm.batch(4, idx=0)
out = []
for it in range(60):
    torch.manual_seed(99)
    loss = model.training_step(batch)
    model.optimizer_step(loss)
    out.append(round(float(loss), 4))
print(out)
print({k: round(v, 4) for k, v in model.losses["train"].compute().items()})
