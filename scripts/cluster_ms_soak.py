"""Soak of the large-batch cluster kernel: random batch sizes 33..512, 3-8 steps, DDIM or DDPM with injected noise; every result must be
bit-identical to k_den_cluster with the same C on the same samples, no cluster may give up, and the run is repeated on a second denoiser
object with its own buffers.  One progress line per 10 cases (stdout is a file under gpurun_out/)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_parity import make_den, _sched

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
NTOK = 2 if os.environ.get("N2") else 1          # N2=1: two condition tokens (scene + interactee)
CONDS = ("text", "scene", "interactee") if NTOK == 2 else ("text", "interactee")
rng = np.random.default_rng(2026)
dens = [make_den(dev, cond=CONDS, weight_dtype=os.environ.get("WD", "fp16")), make_den(dev, cond=CONDS, weight_dtype=os.environ.get("WD", "fp16"))]
ref = make_den(dev, cond=CONDS, weight_dtype=os.environ.get("WD", "fp16"))
ref.cluster_ms = False
bad = 0
t0 = time.time()
for it in range(n_cases):
    B = int(rng.integers(33, (257 if os.environ.get("WD", "fp16") != "fp16" else (385 if NTOK == 2 else 513))))
    steps = int(rng.integers(3, 9))
    kind = "ddpm" if rng.random() < 0.4 else "ddim"
    sch = _sched(kind); sch.set_timesteps(1000 if kind == "ddpm" else 50); sch.timesteps = sch.timesteps[:steps]
    g = torch.Generator(device="cpu").manual_seed(1000 + it)
    lat = torch.randn(B, 1, 256, generator=g).to(dev); cond = torch.randn(B, NTOK, 256, generator=g).to(dev)
    noise = torch.randn(steps, B, 256, generator=g).to(dev) if kind == "ddpm" else None
    den = dens[it % 2]
    Cc, spc = den._cluster_plan(B, NTOK, False, False)
    z = den.sample_loop(lat, cond, sch, step_noise=noise)
    torch.cuda.synchronize()
    st = den.cluster_status()
    ref.cluster, ref.cluster_placement = Cc, 1
    ch = 256 // Cc
    r = torch.cat([ref.sample_loop(lat[i:i + ch].contiguous(), cond[i:i + ch].contiguous(), sch,
                                   step_noise=None if noise is None else noise[:, i:i + ch].contiguous()) for i in range(0, B, ch)], 1)
    torch.cuda.synchronize()
    ok = bool(torch.equal(z, r)) and st[0] == 0
    if not ok:
        bad += 1
        print(json.dumps({"case": it, "B": B, "steps": steps, "kind": kind, "plan": [Cc, spc], "status": st, "max_abs_diff": float((z - r).abs().max())}), flush=True)
    if it % 10 == 9:
        print(json.dumps({"done": it + 1, "bad": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
print("SOAK_OK" if bad == 0 else f"SOAK_FAILURES {bad}", flush=True)
