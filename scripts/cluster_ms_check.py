"""GPU bring-up of k_den_cluster_ms (a cluster of C CUs owns up to 8 samples): results against the one-CU-per-sample kernel on the
same fp16 image for several batch sizes (ragged last clusters included), DDIM and DDPM, then timings of the 50-step launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from conftest import rel_err
from test_gpu_parity import make_den, _sched

dev = torch.device("cuda:0")
den = make_den(dev, weight_dtype="fp16")
sch = _sched(); sch.set_timesteps(50)
ok_all = True


def run(B, ms, cluster="auto", noise=None, sched=sch):
    os.environ["SEEME_DEN_CLUSTER_MS"] = "1" if ms else "0"
    os.environ["SEEME_DEN_CLUSTER"] = str(cluster)
    out = den.sample_loop(lat[:B].contiguous(), cond[:B].contiguous(), sched, step_noise=noise)
    torch.cuda.synchronize()
    return out, den.cluster_status()


def timed(B, ms, cluster="auto", reps=10):
    os.environ["SEEME_DEN_CLUSTER_MS"] = "1" if ms else "0"
    os.environ["SEEME_DEN_CLUSTER"] = str(cluster)
    l, c = lat[:B].contiguous(), cond[:B].contiguous()
    for _ in range(2):
        den.sample_loop(l, c, sch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        den.sample_loop(l, c, sch, events=ev)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    return float(np.median(ts))


torch.manual_seed(7)
lat = torch.randn(512, 1, 256, device=dev); cond = torch.randn(512, 1, 256, device=dev)
# the same image, the same order of every addition: the one-sample cluster kernel with the same C on the same samples, 32 at a time
for B, Cc in ((100, 4), (256, 4), (512, 4)):
    z, st = run(B, True)
    os.environ["SEEME_DEN_CLUSTER_MS"] = "0"; os.environ["SEEME_DEN_CLUSTER"] = str(Cc)
    chunk = 256 // Cc
    ref = torch.cat([den.sample_loop(lat[i:min(i + chunk, B)].contiguous(), cond[i:min(i + chunk, B)].contiguous(), sch) for i in range(0, B, chunk)], 1)
    torch.cuda.synchronize()
    rec = {"test": "vs_one_sample_cluster_same_C", "B": B, "C": Cc, "plan": den._cluster_plan(B, 1, False, False), "max_abs_diff": float((z - ref).abs().max()),
           "bit_identical": bool(torch.equal(z, ref)), "status": st}
    print(json.dumps(rec), flush=True)
    if not rec["bit_identical"]:
        ok_all = False
for B in (65, 100, 128, 256, 257, 300, 512):
    base, _ = run(B, False, 0)
    z, st = run(B, True)
    z2, _ = run(B, True)
    plan = den._cluster_plan(B, 1, False, False)
    rec = {"test": "ddim50", "B": B, "plan": plan, "vs_onecu": rel_err(z.cpu().numpy(), base.cpu().numpy()),
           "worst_sample": float(((z - base).abs().amax(-1) / base.abs().amax()).max()), "rerun_bit_identical": bool(torch.equal(z, z2)), "status": st}
    print(json.dumps(rec), flush=True)
    if st[0] or not rec["rerun_bit_identical"] or rec["vs_onecu"] > 5e-4 or plan[1] < 2:
        ok_all = False
# DDPM with step noise, short
schp = _sched("ddpm"); schp.set_timesteps(1000); schp.timesteps = schp.timesteps[:40]
for B in (130, 512):
    noise = torch.randn(40, B, 256, device=dev)
    bp, _ = run(B, False, 0, noise=noise, sched=schp)
    zp, stp = run(B, True, noise=noise, sched=schp)
    e = rel_err(zp.cpu().numpy(), bp.cpu().numpy())
    print(json.dumps({"test": "ddpm40", "B": B, "vs_onecu": e, "status": stp}), flush=True)
    if stp[0] or e > 5e-4:
        ok_all = False
for B in (128, 192, 256, 384, 512):
    t0 = timed(B, False, 0)
    t1 = timed(B, False)            # one sample per cluster where it fits (B <= 128), else the one-CU kernel
    t2 = timed(B, True)
    print(json.dumps({"test": "time_ddim50", "B": B, "onecu_ms": round(t0, 4), "cluster1_ms": round(t1, 4), "cluster_ms_ms": round(t2, 4),
                      "plan": den._cluster_plan(B, 1, False, False), "us_per_step_ms": round(t2 * 20, 2)}), flush=True)
print("ALL_OK" if ok_all else "FAILURES", flush=True)
