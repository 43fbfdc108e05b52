"""PointNet forward alone (B=64 x 20 000 points, bf16 blocks): ms per forward, and with a -DPN_DBG_TIMES build the
cycle stamps of one workgroup of a middle block at its phase boundaries."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seeme_amd import _lib as L
from seeme_amd.respointnet import ResnetPointnet
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda", 0)
net = ResnetPointnet(precision="bf16")
load_recipe_(net)
net = net.to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(1)
pts = torch.randn(64, 20000, 3, generator=g).to(dev)
with torch.no_grad():
    for _ in range(3):
        net(pts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net(pts)
    e1.record()
    torch.cuda.synchronize()
print("pointnet forward ms", round(e0.elapsed_time(e1) / 10, 3))
lib = L.lib()
if hasattr(lib, "seeme_debug_pn_times"):
    f = lib.seeme_debug_pn_times
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 32)()
    L.check(f(buf, 32))
    tt = np.array(buf[16:20], dtype=np.float64)
    print("cycles per tile between tiles 3/13/23/33:", np.diff(tt) / 10)
    t = np.array(buf[:10], dtype=np.float64)
    names = ["land tile in LDS + barrier", "fc_0 gemm", "barrier", "hidden write + barrier", "fc_1 gemm", "shortcut gemm",
             "issue next tile", "epilogue from registers", "barrier"]
    d = np.diff(t)
    print("total cycles", t[-1] - t[0], " epilogue split: rows+stores", buf[10] - buf[7], "max reduce + smax", buf[8] - buf[10])
    for i, x in enumerate(d):
        print(f"  {names[i]}: {x:.0f}")
