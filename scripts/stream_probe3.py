"""GPU probe: register ring depth vs per-CU weight-stream rate with GEMV-like barriers/epilogues (debug)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd import _lib as L
from probes import probe_lib
lib = probe_lib.lib()
f = lib.seeme_debug_stream_rr
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
nbytes = 11 * 1024 * 1024
src = torch.randn(nbytes // 4, device=dev)
nchunks = 12 * 170 * 10          # ~10 passes over the image (64 KiB per chunk per workgroup)
for blocks in (32,):
    out = torch.zeros(blocks * 512, device=dev)
    for ring in (2, 3, 4, 6):
        for per_gemv, epi in ((0, 0), (2, 0), (2, 8), (2, 24), (4, 8), (4, 24), (8, 8), (8, 24)):
            for _ in range(2):
                probe_lib.check(f(src.data_ptr(), nbytes, nchunks, ring, per_gemv, epi, blocks, out.data_ptr(), L.current_stream()))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                probe_lib.check(f(src.data_ptr(), nbytes, nchunks, ring, per_gemv, epi, blocks, out.data_ptr(), L.current_stream()))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"blocks={blocks:3d} R={ring} barrier every {per_gemv} chunks, epilogue {epi:2d} wave-reductions: "
                  f"{nchunks * 65536 / ms / 1e6:7.1f} GB/s per CU  ({ms * 1e3 / nchunks:5.2f} us/chunk)")

g = lib.seeme_debug_stream_spec
g.restype = C.c_int
g.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
blocks = 32
out = torch.zeros(blocks * 512, device=dev)
for c in (2, 4, 8):
    ngemv = nchunks // c
    for epi in (0, 8, 24):
        for _ in range(2):
            probe_lib.check(g(src.data_ptr(), nbytes, ngemv, c, 0, epi, blocks, out.data_ptr(), L.current_stream()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            probe_lib.check(g(src.data_ptr(), nbytes, ngemv, c, 0, epi, blocks, out.data_ptr(), L.current_stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"role-specialised R=4: {c} chunks per GEMV, epilogue {epi:2d} reductions on wave 0: "
              f"{ngemv * c * 65536 / ms / 1e6:7.1f} GB/s per CU  ({ms * 1e3 / ngemv:5.2f} us/GEMV)")
