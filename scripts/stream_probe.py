"""GPU probe: per-CU weight-stream bandwidth, register loads vs per-wave LDS-DMA ring (debug)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd import _lib as L
from probes import probe_lib
lib = probe_lib.lib()
f = lib.seeme_debug_stream
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
nbytes = 21 * 1024 * 1024
src = torch.randn(nbytes // 4, device=dev)
reps = 10
# expected per-thread sums: thread (wave w, lane l) sums float4 pieces j*8+w, lane l
v = src.view(-1, 8, 64, 4)           # [piece j][wave][lane][4]
expect = (v.double().sum(dim=(0, 3)) * reps).reshape(-1).float()
for blocks in (1, 32, 256):
    for mode, ring in ((0, 0), (1, 8), (1, 16)):
        out = torch.zeros(blocks * 512, device=dev)
        for _ in range(2):
            probe_lib.check(f(src.data_ptr(), nbytes, reps, mode, ring, blocks, out.data_ptr(), L.current_stream()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 3
        for _ in range(n):
            probe_lib.check(f(src.data_ptr(), nbytes, reps, mode, ring, blocks, out.data_ptr(), L.current_stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        err = (out.view(blocks, 512) - expect[None]).abs().max().item() / expect.abs().max().item()
        print(f"blocks={blocks:4d} mode={'regs16' if mode == 0 else 'ring%d' % ring:7s} {ms:8.3f} ms  "
              f"{nbytes * reps / ms / 1e6:8.1f} GB/s per CU   rel err {err:.2e}")
