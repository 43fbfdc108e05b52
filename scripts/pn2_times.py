"""Debug build (-DP2_DBG_TIMES [-DP2_DBG_FIRST=1]): cycles per step of one workgroup of the second-generation PointNet block
kernel: wait at the barrier, and from the barrier to the end of the step, per wave."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seeme_amd import _lib as L
from seeme_amd.respointnet import ResnetPointnet
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda", 0)
pn = load_recipe_(ResnetPointnet(512, 256, precision="bf16")).to(dev).eval()
pts = (torch.rand(64, 20000, 3) * 6 - 3).to(dev)
with torch.no_grad():
    for _ in range(3):
        pn(pts)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 24 * 6))()
f = L.lib().seeme_debug_pn2_times
f.restype = C.c_int
assert f(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(8, 24, 6)
print("step: [barrier wait | barrier->end] per wave (cycles); step length = wave 0's barrier exit to its next barrier exit")
for s in range(24):
    wait = t[:, s, 1] - t[:, s, 0]
    work = t[:, s, 2] - t[:, s, 1]
    nxt = (t[0, s + 1, 1] - t[0, s, 1]) if s + 1 < 24 else -1
    print(f"s={s:2d} len {nxt:5d}  wait {' '.join(f'{int(v):5d}' for v in wait)}   work {' '.join(f'{int(v):5d}' for v in work)}")
print("tile total (step 0 barrier exit -> step 23 end), wave 0:", int(t[0, 23, 2] - t[0, 0, 1]))
for s in (15, 23):
    print(f"epilogue of step {s}: MFMA section {[int(v) for v in t[:, s, 3] - t[:, s, 1]]}  pack + stores issued {[int(v) for v in t[:, s, 4] - t[:, s, 3]]}  "
          f"max-pool {[int(v) for v in t[:, s, 2] - t[:, s, 4]]}")
