"""Debug (build with -DDEN_DBG_TIMES): cycle stamps of workgroup 0 at every barrier of DDIM step 2.
Stamps alternate barrier A (partial sums published) / barrier B (next input published) per GEMV stage."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seeme_amd import _lib as L
import bench
lib = L.lib()
w = sys.argv[1] if len(sys.argv) > 1 else "fp16"
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--weights", w, "--no-cpu-baseline"]
bench.main()
f = lib.seeme_debug_den_times
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 512)()
L.check(f(buf, 512))
n = int(buf[0])
t = np.array(buf[1:1 + n], dtype=np.float64)
d = np.diff(t)
print("stamps", n, "total cycles", t[-1] - t[0])
print("A->B (epilogue phase) cycles:", " ".join(f"{x:.0f}" for x in d[0::2]))
print("B->A (consume phase)  cycles:", " ".join(f"{x:.0f}" for x in d[1::2]))
print("sum A->B", d[0::2].sum(), "sum B->A", d[1::2].sum())
