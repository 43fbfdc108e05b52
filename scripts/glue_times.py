"""GPU debug: time of each launch of the stage-2 glue (warm, and cold after a 2 GB stream through the caches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule
from seeme_amd.smpl import SMPL
from seeme_amd import _lib as L
import ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = parse_config(os.path.join(REPO, "configs", "config_mld_scene.yaml"))
dm = SyntheticEgoDataModule(nfeats=75, T=196, n_points=2000, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234)).to(dev).train()
B = 64
batch = dm.batch(B, idx=0, with_scene=True)
for _ in range(3):
    loss = model.training_step(batch)
    model.optimizer_step(loss)
glue = model._glue
plan = glue.plans[(B, True, True)]
big = torch.empty(512 * 1024 * 1024, device=dev, dtype=torch.float32)
st = L.current_stream()
def t(fn, cold):
    ts = []
    for _ in range(8):
        if cold:
            big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
items = [("l0", plan.g_l0.launch), ("l1", plan.g_l1.launch), ("l2", plan.g_l2.launch), ("b0", plan.g_b0.launch), ("b2", plan.g_b2.launch),
         ("b3", plan.g_b3.launch), ("mid", lambda: L.check(L.lib().seeme_glue_mid(C.byref(plan.mid), st))),
         ("ln", lambda: L.check(L.lib().seeme_glue_ln(plan.cond.data_ptr(), plan.xhat.data_ptr(), plan.rstd.data_ptr(), plan.M, st))),
         ("empty-ish k_fill", lambda: plan.rstd.zero_())]
for name, fn in items:
    g = getattr(plan, "g_" + name, None)
    print(f"{name:18s} warm {t(fn, False):7.1f} us   cold {t(fn, True):7.1f} us" + (f"   problems {g.n} tiles {g.tiles}" if g else ""))
