"""PointNet bf16 encode: time and error of the block-kernel generation selected by SEEME_PN_V2 (1 default, 0 = first generation)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seeme_amd.respointnet import ResnetPointnet
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda", 0)
B, P = int(os.environ.get("B", 64)), int(os.environ.get("P", 20000))
pn = load_recipe_(ResnetPointnet(512, 256)).to(dev).eval()
g = torch.Generator().manual_seed(3)
pts = (torch.rand(B, P, 3, generator=g) * 6 - 3).to(dev)
with torch.no_grad():
    ref = pn(pts[:4])                      # fp32 path
    pn.precision = "bf16"
    got = pn(pts[:4])
    err = float((got - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        pn(pts)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); pn(pts); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
flops = B * P * (2 * 3 * 512 + (2 * 512 * 256 + 2 * 256 * 256 + 2 * 4 * 256) + 3 * (3 * 2 * 256 * 256))
print(json.dumps({"v2": os.environ.get("SEEME_PN_V2", "1"), "B": B, "P": P, "ms": round(ms, 4), "tflops_executed": round(flops / ms / 1e9, 1),
                  "rel_err_vs_fp32_B4": err}))
