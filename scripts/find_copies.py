"""Debug: which host lines of one sampling pass issue device copies (hipMemcpy* / blit kernels)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
# reuse bench's model construction by running it once, then profile further passes through its globals
state = {}
orig = bench.one_pass
def spy(*a, **k):
    state["args"] = (a, k)
    return orig(*a, **k)
bench.one_pass = spy
bench.main()
a, k = state["args"]
k = dict(k); k.pop("ev", None)
a = a[:6]
for _ in range(2):
    orig(*a)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        orig(*a)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "copy_" in n and e.device_type.name == "CPU" and False:
        st = [s for s in (e.stack or []) if "/root/repo" in s or "seeme_amd" in s or "bench.py" in s]
        cnt[(n, tuple(st[:3]))] += 1
for (n, st), c in cnt.most_common(40):
    print(c / 3, n, " <- ", " | ".join(x.split("/")[-1] for x in st))
names = collections.Counter(e.name for e in prof.events() if e.device_type.name != "CPU")
print({k: v / 3 for k, v in names.most_common(12)})
