"""GPU bring-up of k_den_cluster (one sample split over C CUs): results against the one-CU-per-sample kernel and the reference
fixtures, every weight dtype / cluster size / placement / store flavour, then timings of the 50-step DDIM launch at B = 32."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from conftest import load_golden, rel_err
from test_gpu_parity import make_den, _sched

dev = torch.device("cuda:0")
quick = "--quick" in sys.argv


def run(den, lat, cond, sch, cluster, place=0, flags=0, noise=None):
    os.environ["SEEME_DEN_CLUSTER"] = str(cluster)
    os.environ["SEEME_DEN_CLUSTER_PLACE"] = str(place)
    os.environ["SEEME_DEN_CLUSTER_FLAGS"] = str(flags)
    out = den.sample_loop(lat, cond, sch, step_noise=noise)
    torch.cuda.synchronize()
    st = den.cluster_status() if cluster else (0, 0)
    return out, st


def timed(den, lat, cond, sch, cluster, place=0, flags=0, reps=20):
    os.environ["SEEME_DEN_CLUSTER"] = str(cluster)
    os.environ["SEEME_DEN_CLUSTER_PLACE"] = str(place)
    os.environ["SEEME_DEN_CLUSTER_FLAGS"] = str(flags)
    for _ in range(3):
        den.sample_loop(lat, cond, sch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        den.sample_loop(lat, cond, sch, events=ev)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    return float(np.median(ts))


ok_all = True
g1 = load_golden("denoiser_N1.npz")
g50 = load_golden("ddim50_N1_B3.npz")
for wd in ("fp32", "fp16", "bf16"):
    den = make_den(dev, weight_dtype=wd)
    sch = _sched(); sch.set_timesteps(50)
    # one forward against the reference fixture (per-sample and scalar timesteps)
    s, c = torch.from_numpy(g1["sample"]).to(dev), torch.from_numpy(g1["cond"]).to(dev)
    for Cc in (8, 4, 2):
        os.environ["SEEME_DEN_CLUSTER"] = str(Cc)
        y = den(sample=s, timestep=torch.tensor(501), encoder_hidden_states=c)[0]
        yv = den(sample=s, timestep=torch.from_numpy(g1["tvec"]).to(dev), encoder_hidden_states=c)[0]
        torch.cuda.synchronize()
        e, ev_ = rel_err(y.cpu().numpy(), g1["out_t501"]), rel_err(yv.cpu().numpy(), g1["out_tvec"])
        print(json.dumps({"test": "forward_vs_reference", "wd": wd, "C": Cc, "rel_err_t501": e, "rel_err_tvec": ev_, "status": den.cluster_status()}), flush=True)
        if wd == "fp32" and (e > 1e-4 or ev_ > 1e-4):
            ok_all = False
    # 50-step loop against the reference fixture and against the one-CU kernel
    lat3, cond3 = torch.from_numpy(g50["latents"]).to(dev), torch.from_numpy(g50["cond_bf"]).to(dev)
    base3, _ = run(den, lat3, cond3, sch, 0)
    torch.manual_seed(5)
    lat = torch.randn(32, 1, 256, device=dev); cond = torch.randn(32, 1, 256, device=dev)
    base, _ = run(den, lat, cond, sch, 0)
    for Cc in (8, 4, 2):
        for place in (0, 1):
            for flags in (0, 1):
                if quick and (place, flags) != (0, 0):
                    continue
                z3, st3 = run(den, lat3, cond3, sch, Cc, place, flags)
                z, st = run(den, lat, cond, sch, Cc, place, flags)
                z2, _ = run(den, lat, cond, sch, Cc, place, flags)
                rec = {"test": "ddim50", "wd": wd, "C": Cc, "place": place, "flags": flags,
                       "B3_vs_reference": rel_err(z3.cpu().numpy(), g50["out"]), "B3_onecu_vs_reference": rel_err(base3.cpu().numpy(), g50["out"]),
                       "B32_vs_onecu": rel_err(z.cpu().numpy(), base.cpu().numpy()), "rerun_bit_identical": bool(torch.equal(z, z2)),
                       "status_B3": st3, "status_B32": st}
                print(json.dumps(rec), flush=True)
                if st[0] or st3[0] or not rec["rerun_bit_identical"] or (wd == "fp32" and rec["B3_vs_reference"] > 5e-4):
                    ok_all = False
    # timings at B = 32
    if wd != "bf16" or not quick:
        t0 = timed(den, lat, cond, sch, 0)
        for Cc in (8, 4, 2):
            for place in (0, 1):
                for flags in (0, 1):
                    if (place == 1 and flags == 0) and False:
                        continue
                    t = timed(den, lat, cond, sch, Cc, place, flags)
                    print(json.dumps({"test": "time_ddim50_B32", "wd": wd, "C": Cc, "place": place, "flags": flags, "ms": round(t, 4),
                                      "onecu_ms": round(t0, 4), "us_per_step": round(t * 20, 2)}), flush=True)
    # DDPM with step noise, short
    schp = _sched("ddpm"); schp.set_timesteps(1000); schp.timesteps = schp.timesteps[:40]
    noise = torch.randn(40, 32, 256, device=dev)
    bp, _ = run(den, lat, cond, schp, 0, noise=noise)
    zp, stp = run(den, lat, cond, schp, 8, noise=noise)
    print(json.dumps({"test": "ddpm40", "wd": wd, "C": 8, "vs_onecu": rel_err(zp.cpu().numpy(), bp.cpu().numpy()), "status": stp}), flush=True)
    # ragged batch (clusters beyond B exit at once)
    z5, st5 = run(den, lat[:5].contiguous(), cond[:5].contiguous(), sch, 8)
    print(json.dumps({"test": "B5", "wd": wd, "vs_first5": rel_err(z5.cpu().numpy(), base[:, :5].cpu().numpy()), "status": st5}), flush=True)
print("ALL_OK" if ok_all else "FAILURES", flush=True)
