"""GPU probe: how barriers / epilogues / chunk depth affect the per-CU weight-stream rate (debug)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd import _lib as L
from probes import probe_lib
lib = probe_lib.lib()
f = lib.seeme_debug_stream_gemv
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
nbytes = 21 * 1024 * 1024
src = torch.randn(nbytes // 4, device=dev)
reps, blocks = 10, 32
out = torch.zeros(blocks * 512, device=dev)
for ch in (4, 8, 16):
    for per_gemv, epi in ((0, 0), (8, 0), (4, 0), (2, 0), (4, 8), (4, 32), (2, 32)):
        for _ in range(2):
            probe_lib.check(f(src.data_ptr(), nbytes, reps, ch, per_gemv, epi, blocks, out.data_ptr(), L.current_stream()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            probe_lib.check(f(src.data_ptr(), nbytes, reps, ch, per_gemv, epi, blocks, out.data_ptr(), L.current_stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"CH={ch:2d} barrier every {per_gemv} chunks, epilogue {epi:2d} wave-reductions: {nbytes * reps / ms / 1e6:7.1f} GB/s per CU")
