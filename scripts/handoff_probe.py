"""GPU probe: one-way latency of a CU-to-CU hand-off of a 256-float vector through L2 (flag + data), same XCD and across XCDs."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd import _lib as L
from probes import probe_lib
lib = probe_lib.lib()
f = lib.seeme_debug_handoff
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
iters = 2000
for name, a, b in (("same XCD (blocks 0 and 8)", 0, 8), ("neighbouring XCDs (blocks 0 and 1)", 0, 1), ("XCD 0 and XCD 4 (blocks 0 and 4)", 0, 4)):
    for floats in (1, 256):
        ts = []
        for rep in range(4):
            flags = torch.zeros(128, dtype=torch.int32, device=dev)
            buf = torch.zeros(2048, device=dev)
            out = torch.zeros(2, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            probe_lib.check(f(flags.data_ptr(), buf.data_ptr(), a, b, iters, floats, 16, out.data_ptr(), L.current_stream()))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us = sorted(ts)[1]
        print(f"{name:38s} {floats:3d} floats: {us / (2 * iters):6.2f} us per one-way hand-off  ({iters} round trips in {us:8.1f} us)")
print("sampling kernel: 32-47 dependent GEMVs per step x 50 steps; a split of one sample over k CUs needs one exchange per GEMV")
