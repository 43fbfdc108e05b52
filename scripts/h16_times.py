"""Debug (build with -DH16_DBG_TIMES -DH16_DBG_KERNEL=k): cycle stamps of one workgroup of a fp16 VAE kernel at its phase
boundaries during a default bench pass (the last launch of the kernel in the pass is reported)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seeme_amd import _lib as L
import bench
names = {1: ["Q tile -> LDS", "Q K^T + scores", "softmax", "P V", "out_proj", "residual + LN + store"],
         2: ["stage A tile", "GEMM", "acc -> LDS + barrier", "epilogue"],
         3: ["stage rows (+ cross-attention vector, LN)", "GEMM1 + hidden -> LDS", "GEMM2 + tile -> LDS", "residual + LN + store"],
         5: ["Q tile -> LDS", "Q K^T + scores -> LDS", "softmax", "P V -> O operand", "out_proj -> tile (+ FFN weight requests)", "residual + LN1 (+ cross vector, LN)",
             "FFN GEMM1 + GELU -> hidden", "FFN GEMM2 -> tile", "residual + LN2 + store", "skip linear (if any)", "next layer's q | k | V^T"],
         4: ["stage 128 rows", "GEMM", "barrier + acc -> LDS", "stores (q|k rows / V transposed)"]}
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity-check"] + (["--batch", os.environ["H16_B"]] if "H16_B" in os.environ else [])
bench.main()
f = L.lib().seeme_debug_h16_times
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
L.check(f(buf, 32))
n = len(names[k]) + 1
t = np.array(buf[:n], dtype=np.float64)
print("kernel", k, "total", t[-1] - t[0])
for nm, d in zip(names[k], np.diff(t)):
    print(f"  {nm}: {d:.0f}")
if k == 5:
    t2 = np.array(buf[16:21], dtype=np.float64)
    print("  LN1 phase detail (cycles since the phase's barrier):", [int(x - t[5]) for x in t2],
          "= loads issued, tile + residual in registers, LN of 8 rows done, 2nd LN done, operand rows stored")
    t3 = np.array(buf[23:26], dtype=np.float64)
    print("  out_proj phase detail (cycles since the phase's barrier):", [int(x - t[4]) for x in t3], "= GEMM done, tile stored, FFN requests issued; the rest is the barrier")
