"""Debug (library built with -DDEN_DBG_TIMES, SEEME_HIP_LIB pointing at it): cycle stamps of workgroup 0's epilogue wave at every
barrier / finished sweep of DDIM step 2 of k_den_cluster.  Per layer: A (in_proj' MFMAs + lane epilogue) | X1 (scores, publish,
sweep) | attention + norm1 | B | C + X2 (MFMAs, publish, sweep) | norm2 | D | E | E epilogue | F  (+ scheduler after layer 4)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from seeme_amd import _lib as L
from test_gpu_parity import make_den, _sched
dev = torch.device("cuda:0")
wd = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
den = make_den(dev, weight_dtype=wd)
sch = _sched(); sch.set_timesteps(50)
torch.manual_seed(5)
lat = torch.randn(B, 1, 256, device=dev); cond = torch.randn(B, 1, 256, device=dev)
for _ in range(3):
    den.sample_loop(lat, cond, sch)
torch.cuda.synchronize()
f = L.lib().seeme_debug_den_times
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 512)()
L.check(f(buf, 512))
n = int(buf[0])
t = np.array(buf[1:1 + n], dtype=np.float64)
d = np.diff(t)
names = ["X1", "attn+ln1", "B", "C+X2", "ln2", "D", "E", "E-epi", "F", "A"]
print(json.dumps({"wd": wd, "B": B, "cluster": os.environ.get("SEEME_DEN_CLUSTER"), "place": os.environ.get("SEEME_DEN_CLUSTER_PLACE"),
                  "flags": os.environ.get("SEEME_DEN_CLUSTER_FLAGS"), "stamps": n, "step_cycles": float(t[-1] - t[0])}))
# stamp 0 = after barrier 1 of layer 0; 10 stamps per layer (+1 after the last layer: scheduler)
per = {k: [] for k in names + ["sched"]}
i = 0
for l in range(5):
    for k in names[:9]:
        if i < len(d): per[k].append(d[i]); i += 1
    if l == 4:
        if i < len(d): per["sched"].append(d[i]); i += 1
    else:
        if i < len(d): per["A"].append(d[i]); i += 1
for k, v in per.items():
    print(f"{k:9s} " + " ".join(f"{x:6.0f}" for x in v) + f"   | sum {sum(v):7.0f}")
