"""GPU debug: which stage-1 parameters end a training step without a finite gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = parse_config(os.path.join(REPO, "configs", "config_vae_egobody.yaml"))
cfg.TRAIN.OPTIM.LR = 1e-4
torch.manual_seed(1234)
dm = SyntheticEgoDataModule(nfeats=75, T=24, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234)).to(dev).train()
batch = dm.batch(4, idx=0)
for it in range(3):
    torch.manual_seed(99)
    loss = model.training_step(batch)
    model.optimizer_step(loss)
    bad = [(n, None if p.grad is None else bool(torch.isfinite(p.grad).all())) for n, p in model.vae.named_parameters()
           if p.grad is None or not torch.isfinite(p.grad).all()]
    print(it, float(loss), "used", len(model._used_params), "bad:", bad)
print("requires_grad False:", [n for n, p in model.vae.named_parameters() if not p.requires_grad])
