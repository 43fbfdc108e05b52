#!/usr/bin/env python3
"""bench.py -- sampled sequences / second on the motion-latent-diffusion hot path.

One "step" = one pass of the path over one batch of synthetic input, per GPU:
    interactee motion [B,196,132] --MldVae.encode--> condition token
    latents [B,1,256] --50-step DDIM loop through MldDenoiser (one persistent kernel)--> z
    z --MldVae.decode--> motion [B,196,132]
(BASELINE.json configs[1]: config_mld_egobody.yaml interactee-only denoiser, 50 DDIM steps, B=32.)
All inputs are resident in HBM before the timed region.  Weights are random-init by the seeded
recipe (no checkpoint exists offline); data is synthetic N(0,1) of the named shape.

    python bench.py --gpus N --steps K --warmup W
N>1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when WORLD_SIZE
is not set -- this process starts the N ranks itself as child processes BEFORE anything touches the GPU and relays
rank 0's line.  `--mode train` times the stage-2 training step instead (BASELINE configs[2] at N=1: scene + interactee,
B=64/GPU, 20 000-point scenes; configs[3] at N>1: config_mld_gimo, scene only), with the gradient all-reduce over RCCL.

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` (dominant kernel,
timed with events on its own stream) and `cpu_baseline` (the PyTorch-CPU oracle on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

T_FRAMES, NFEATS, DDIM_STEPS = 196, 132, 50
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def ablation():
    return types.SimpleNamespace(MLP_DIST=False, PE_TYPE="mld", SKIP_CONNECT=True, VAE_TYPE="actor",
                                 DIFF_PE_TYPE="mld", MD_TRANS=True)


def build_models(dev, weight_dtype, vae_precision="fp32"):
    from seeme_amd.mld_denoiser import MldDenoiser
    from seeme_amd.mld_vae import MldVae
    from seeme_amd.schedulers import DDIMScheduler
    from seeme_amd.weights_recipe import load_recipe_
    vae = load_recipe_(MldVae(ablation(), nfeats=NFEATS, latent_dim=[1, 256], arch="encoder_decoder",
                             precision=vae_precision)).to(dev).eval()
    den = load_recipe_(MldDenoiser(ablation(), nfeats=NFEATS, condition=["text", "interactee"], latent_dim=[1, 256],
                                   ff_size=128, num_layers=5, num_heads=1, weight_dtype=weight_dtype)).to(dev).eval()
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sch.set_timesteps(DDIM_STEPS)
    return vae, den, sch


def one_pass(vae, den, sch, motion, latents, lengths, ev=None):
    dist = vae.encode_dist(motion, lengths)                 # [2,B,256]; mu is the condition (ego_eval, mld.py:1271-1295)
    cond = dist[0].unsqueeze(1)                             # [B,1,256] batch-first, N = 1 token
    z = den.sample_loop(latents, cond, sch, events=ev)   # event pair brackets only the persistent kernel
    return vae.decode(z, lengths)


REF_LIVE_PARAMS = 7_644_288   # the denoiser's 7.64 M live parameters (mem_pos and unused pe rows excluded)


def den_algorithmic_bytes(den, B, N, steps):
    """SURVEY.md section 8(d) per-unit figure x units of one launch: per DDIM step the reference graph's live
    weights once for the whole batch (15.3 MB at 16 bit, 30.6 MB fp32) + B x (2+N) x 256 activations."""
    esize = 4 if den.weight_dtype == "fp32" else 2
    return steps * (REF_LIVE_PARAMS * esize + B * (2 + N) * 256 * esize)


def den_executed_bytes(den, B, N, steps):
    """Bytes the sampling kernel actually streams per launch and workgroup chain: the matrices its program
    keeps after the exact reductions (token-0 pruning, folded out_proj, tabulated ca term), the per-layer
    vectors and the table rows, once per step; per sample the condition tables and the latent."""
    esize = 4 if den.weight_dtype == "fp32" else 2
    D, FS, FF = 256, 1024, den.ff_size
    per_layer = 3 * D * D + 2 * FS * D + 2 * FF * D + D * D          # in_proj, linear1/2, ffn.linear1/2, ffn proj_out
    if den.num_heads != 1:
        per_layer += D * D                                            # out_proj (not folded)
    if N > 1:
        per_layer += 2 * D * D                                        # ca query + proj_out
    mats = 5 * per_layer + 2 * 2 * D * D                              # + two skip linears
    per_step = mats * esize + 5 * (6272 + 1536 + (256 if N == 1 else 0)) * 4
    per_sample = N * 5120 * 4 + 2 * 256 * 4
    return steps * per_step + B * per_sample


def den_cluster_bytes_per_cu(den, Cc, steps):
    """Bytes ONE CU of a cluster streams per launch of k_den_cluster (csrc/den_cluster.inc.hip): its slice of in_proj' (+ the folded
    skip linear in layers 3, 4), linear1, linear2 and the whole of ffn.linear1 / linear2 / proj_out (replicated), plus the per-layer
    vectors and table rows, once per step."""
    esize = 4 if den.weight_dtype == "fp32" else 2
    D, FS, FF = 256, 1024, den.ff_size
    S = D // Cc
    a_plain, a_skip = 3 * S * D, 4 * S * 2 * D                         # q | k | v' rows; + y rows, K = 512
    split = 2 * FS * D // Cc                                           # linear1 rows + linear2 columns of this CU
    repl = 2 * FF * D + D * D
    mats = 3 * (a_plain + split + repl) + 2 * (a_skip + split + repl)
    return steps * (mats * esize + 5 * (6272 + 1536 + 1024 + 256) * 4)


def measured_traffic(weights, B, steps, Cc=0):
    """HBM-side bytes per launch of the sampling kernel from the committed rocprofv3 PMC passes (scripts/gpu_traffic.sh;
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), if this exact configuration was profiled: a constant read from
    profiles/traffic.json, NOT observed in this run (PMC counters need the profiler around the process)."""
    path = os.path.join(REPO, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"{weights}_B{B}_steps{steps}" + (f"_C{Cc}" if Cc else ""))
    except OSError:
        return None


def cpu_baseline(B, budget_s=12.0):
    """The CPU oracle (oracle/mld_oracle_torch.py: a port of the reference CPU path on PyTorch-CPU fp32 -- the baseline
    SURVEY.md section 8(d) names -- pinned to the reference by tests/golden) on this host, same pass as the GPU leg:
    VAE encode -> 50-step DDIM -> VAE decode at the bench batch.  The thread count is picked by a short trial
    (more threads than the box's CPU share only add contention) and reported as `cores`."""
    from oracle import mld_oracle_torch as OT
    from seeme_amd import shapes
    from seeme_amd.weights_recipe import recipe_state_dict
    Pv, Pd = OT.to_torch(recipe_state_dict(shapes.vae_shapes(NFEATS))), OT.to_torch(recipe_state_dict(shapes.denoiser_shapes()))
    rng = np.random.Generator(np.random.PCG64(1234))
    Bc = min(B, 32)
    motion = torch.from_numpy(rng.standard_normal((Bc, T_FRAMES, NFEATS)).astype(np.float32))
    lat = torch.from_numpy(rng.standard_normal((Bc, 1, 256)).astype(np.float32))
    lengths = [T_FRAMES] * Bc

    def one():
        mu, _ = OT.vae_encode(Pv, motion, lengths)
        z = OT.diffusion_reverse(Pd, mu.permute(1, 0, 2), lat, DDIM_STEPS)
        OT.vae_decode(Pv, z, lengths)

    prev = torch.get_num_threads()
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    trials = {}

    def short():                                          # trial workload: the encode + 5 of the 50 steps
        mu, _ = OT.vae_encode(Pv, motion, lengths)
        OT.diffusion_reverse(Pd, mu.permute(1, 0, 2), lat, 5)

    for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(th)
        short()                                           # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        short()
        trials[th] = time.perf_counter() - t0
        if trials[th] > 1.2 * min(trials.values()):       # past the box's CPU share more threads only add contention
            break
    best = min(trials, key=trials.get)
    torch.set_num_threads(best)
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += Bc
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    # BASELINE configs[0] (the reference's own CPU-runnable case): VAE reconstruct, B = 4 random T = 196 sequences
    m4, l4 = motion[:4], [T_FRAMES] * 4

    def rec():
        mu4, _ = OT.vae_encode(Pv, m4, l4)
        OT.vae_decode(Pv, mu4, l4)

    rec()
    n1, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0:
        rec()
        n1 += 4
    cfg1 = {"value": round(n1 / (time.perf_counter() - t1), 2), "unit": "seqs/s",
            "sample": f"{n1} sequences: config_vae_egobody VAE reconstruct (encode -> decode(mu)), passes of B=4, T=196, F={NFEATS}, {best} threads"}
    torch.set_num_threads(prev)
    return {"value": round(n / dt, 3), "unit": "seqs/s", "cores": best, "kind": "port", "cfg1_vae_reconstruct": cfg1,
            "sample": f"{n} sequences (passes of B={Bc}, T=196, 50 DDIM steps) in {dt:.1f}s, PyTorch-CPU fp32 oracle, "
                      f"{best} threads of {ncpu} usable logical CPUs (trial s: "
                      + ", ".join(f"{k}t {v:.2f}" for k, v in trials.items()) + ")"}


def mpjpe_check(dev, weights, vae_prec, B):
    """north_star's accuracy gate for the throughput mode: config_mld_egobody (interactee-only, nfeats 75, so that the
    decoded features are SMPL parameters), B sequences of T=196, the same inputs / initial latents / condition noise
    through (a) the fp32 parity path -- pinned to the oracle within 1e-5 mm by tests/test_gpu_parity.py -- and (b) the
    precision mode this run is timed in; MPJPE of each against the synthetic ground truth (metrics/compute.py alignment),
    their difference (the gate: 1e-3 mm) and the mean joint-to-joint distance between the two outputs."""
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule, EgoMetrics
    from seeme_amd.smpl import SMPL
    from seeme_amd.weights_recipe import load_recipe_
    cfg = parse_config(os.path.join(REPO, "configs", "config_mld_egobody.yaml"))
    dm = SyntheticEgoDataModule(nfeats=75, T=T_FRAMES, device=dev)
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser)
    model = model.to(dev).eval()
    NB = 8                                              # evaluated set: NB batches of B sequences
    per_seq = {"fp32": [], "mode": []}
    per_batch, j2j = [], []
    with torch.no_grad():
        for it in range(NB):
            batch = dm.batch(B, idx=1 + it)
            g = torch.Generator().manual_seed(5 + it)
            lat, eps = torch.randn(B, 1, 256, generator=g).to(dev), torch.randn(1, B, 256, generator=g).to(dev)
            out = {}
            for tag, (wd, vp) in (("fp32", ("fp32", "fp32")), ("mode", (weights, vae_prec))):
                model.denoiser.weight_dtype, model.vae.precision = wd, vp
                rs = model.ego_eval(batch, latents=lat, cond_noise=eps)
                m = EgoMetrics.per_sequence(rs["joints_rst"], rs["joints_ref"], rs["lengths"])["MPJPE"].double()
                per_seq[tag].append(m)
                out[tag] = (m.mean().item(), rs["joints_rst"])
            per_batch.append(abs(out["mode"][0] - out["fp32"][0]))
            j2j.append(float((out["mode"][1] - out["fp32"][1]).norm(dim=-1).mean() * 1000.0))
    m32, mm = torch.cat(per_seq["fp32"]).mean().item(), torch.cat(per_seq["mode"]).mean().item()
    return {"mpjpe_delta_mm": round(abs(mm - m32), 7), "gate_mm": 1e-3,
            "mpjpe_fp32_path_mm": round(m32, 6), "mpjpe_this_mode_mm": round(mm, 6),
            "per_batch_delta_mm": {"mean": round(sum(per_batch) / NB, 7), "max": round(max(per_batch), 7)},
            "joint_to_joint_mm": round(sum(j2j) / NB, 4),
            "on": f"config_mld_egobody, {NB} batches of B={B} (= {NB * B} sequences), T=196, nfeats=75, synthetic SMPL, same inputs / "
                  "latents / condition noise; MPJPE = mean over the evaluated sequences, as the reference's metric is (compute.py:"
                  "488-580); reference = the fp32 HIP path (pinned to the CPU oracle by the gpu tests).  The shift of ONE batch is "
                  "rounding noise of the 16-bit operands (two equally valid fp16 schedules of the same VAE differ as much): see "
                  "per_batch_delta_mm"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which has not
    touched the GPU (no HIP call, no torch.cuda.is_available()) and never will; rank 0 prints the JSON line on the
    inherited stdout.  Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    GRACE_S = 10.0          # a failed rank's siblings get this long to finish on their own (their own error message, their own
    failed_at = None        # exit code) before they are terminated: in a collective they would otherwise wait for ever
    try:
        while any(p.poll() is None for p in procs):
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = time.perf_counter()
            if failed_at is not None and time.perf_counter() - failed_at > GRACE_S:
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            rc = max(rc, abs(p.wait()))
    return rc


PN_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak ~2.5 PFLOP/s


def pointnet_flops(B, P):
    """(executed, reference-graph) FLOPs of one PointNet encode.  Executed: fc_pos_0 3->512, block_0 = fc_0 512->256 +
    fc_1 256->256 (+ the folded 3->256 shortcut), blocks 1-3 = fc_0 / shortcut on the 256 per-point features (the pooled
    half is a per-scene vector) + fc_1.  Reference graph: SURVEY section 6, 52.49 GF per 20 000 points."""
    per_point = 2 * 3 * 512 + (2 * 512 * 256 + 2 * 256 * 256 + 2 * 4 * 256) + 3 * (3 * 2 * 256 * 256)
    return float(B) * P * per_point, float(B) * P * 52.49e9 / 20000.0


def train_mode(args, dev, rank, world, backend, dist_on):
    """One step = stage-2 training step on one batch per GPU: frozen PointNet + frozen VAE encodes (HIP), denoiser
    forward + hand-written backward (HIP), one in-place all-reduce of the flat gradient buffer, one-launch AdamW."""
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule
    from seeme_amd.smpl import SMPL
    from seeme_amd.weights_recipe import load_recipe_
    from seeme_amd import distributed as D
    which = args.train_config or ("scene" if world == 1 else "gimo")
    cfgfile = {"scene": "config_mld_scene.yaml", "gimo": "config_mld_gimo.yaml", "egobody": "config_mld_egobody.yaml",
               "vae": "config_vae_egobody.yaml"}[which]
    stage1 = which == "vae"
    cfg = parse_config(os.path.join(REPO, "configs", cfgfile))
    cfg.TRAIN.FROZEN_VAE_PRECISION = args.vae
    cfg.TRAIN.SCENE_PRECISION = "bf16"
    nfeats = 69 if cfg.DATASET_NAME == "gimo" else 75
    B, P = args.batch, args.points
    dm = SyntheticEgoDataModule(nfeats=nfeats, T=T_FRAMES, n_points=P, seed=1234, device=dev)
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser)
    with_scene = "scene" in cfg.model.condition and not stage1
    if with_scene:
        load_recipe_(model.proscene.scene_enc)
    model = model.to(dev).train()
    D.broadcast_parameters(model)
    model.configure_optimizers()
    batches = [dm.batch(B, idx=2 * rank + i, with_scene=with_scene) for i in range(2)]     # resident in HBM
    ev = lambda: torch.cuda.Event(enable_timing=True)
    pn_ev = []
    if args.graph and stage1:
        raise SystemExit("--graph: stage-2 training only")
    if args.graph:
        replay = model.capture_training_step(batches[0])
        step = lambda i, e=None: replay(batches[i % 2])
    else:
        def step(i, e=None):
            if e is not None and with_scene:
                model.proscene.scene_enc.timing_events = (e[4], e[5])
            loss = model.training_step(batches[i % 2], i)
            model.optimizer_step(loss, events=e[:4] if e is not None else None)
            return loss
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [[ev() for _ in range(6)] for _ in range(args.steps)]
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i, evs[i])
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(loss.detach()).all()
    if with_scene:
        model.proscene.scene_enc.timing_events = None
    res = {
        "metric": ("stage-1 (VAE) training seqs/sec (T=196, B=%d/GPU)" % B) if stage1 else
                  "stage-2 training seqs/sec (T=196, B=%d/GPU%s)" % (B, ", %d-point scenes" % P if with_scene else ""),
        "value": round(world * B * args.steps / dt, 2), "unit": "seqs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (VAE + SMPL forward / backward on fp32 MFMA, AdamW)" if stage1 else
                 f"f32 (denoiser forward/backward, AdamW; frozen encoders: bf16 PointNet MFMA operands, {args.vae} VAE operands)",
        "data": "synthetic",
        "config": {"workload": f"{cfgfile}: stage-{1 if stage1 else 2} training step, condition {list(cfg.model.condition)}, B={B}/GPU, T=196, "
                               f"nfeats={nfeats}" + (f", {P}-point scenes" if with_scene else "") + ", random-init recipe weights",
                   "batch_per_gpu": B, "global_batch": B * world, "seq_len": T_FRAMES,
                   "parallelism": f"dp{world} (one in-place all-reduce of the flat fp32 gradient buffer per step)"
                                  + (", hipGraph replay" if args.graph else "")},
        "rccl_ranks": torch.distributed.get_world_size() if dist_on else 1,
        "collective_backend": (backend if dist_on else None),
        "grad_bucket_bytes": (int(model.grad_bucket().flat.numel() * 4) if model.grad_bucket() is not None else None),
    }
    gb = model.grad_bucket()
    if gb is not None and gb.early_span is not None:
        # exchanged asynchronously from the middle of the backward (every chain matrix but in_proj), the rest after it
        eb = int((gb.early_span[1] - gb.early_span[0]) * 4)
        res["grad_exchange"] = {"overlapped_with_backward_bytes": eb if (gb.overlap and world > 1) else 0,
                                "after_backward_bytes": int(gb.flat.numel() * 4) - (eb if (gb.overlap and world > 1) else 0),
                                "note": "phases_ms.allreduce is what the step still waits for after the backward"}
    if not args.graph:
        ph = lambda a, b: round(float(np.mean([e[a].elapsed_time(e[b]) for e in evs])), 4)
        res["phases_ms"] = {"backward": ph(0, 1), "allreduce": ph(1, 2), "adamw": ph(2, 3)}
        if stage1:
            # GEMM work of the step: per sequence and layer QKV + Q K^T + P V + out_proj + FFN, 5 encoder layers on T + 2 tokens and 5
            # decoder layers on T, skip linears, embedding / final projection; forward + data gradient + weight gradient = 3 x forward
            def fwd(S):
                return 5 * (2 * S * 256 * 768 + 2 * 2 * S * S * 256 + 2 * S * 256 * 256 + 2 * 2 * S * 256 * 128) + 2 * 2 * S * 512 * 256
            flops = 3.0 * B * (fwd(T_FRAMES + 2) + fwd(T_FRAMES) + 2 * 2 * T_FRAMES * nfeats * 256)
            ms = dt / args.steps * 1e3
            ach = flops / (ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "mfma", "kernel": "k_gg (grouped fp32 GEMM on v_mfma_f32_32x32x2_f32): every GEMM of the VAE forward / "
                               "backward; priced over the WHOLE step", "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s",
                               "frac": round(ach / 157.3, 4), "traffic": None, "ms_per_launch": round(ms, 4), "gemm_flops_per_step": flops}
        if with_scene:
            pn_ms = float(np.mean([e[4].elapsed_time(e[5]) for e in evs]))
            exe, refg = pointnet_flops(B, P)
            ach = exe / (pn_ms * 1e-3) / 1e12
            res["phases_ms"]["pointnet"] = round(pn_ms, 4)
            res["roofline"] = {"bound": "mfma", "kernel": "seeme_pointnet_encode_bf16 (4 x k_pn_block + pooled-vector maps), "
                               "the dominant part of the step", "achieved": round(ach, 2), "peak": PN_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / PN_PEAK_TFLOPS, 4), "traffic": None,
                               "ms_per_launch": round(pn_ms, 4), "executed_flops_per_launch": exe,
                               "reference_graph_flops_per_launch": refg,
                               "work_equivalent_tflops": round(refg / (pn_ms * 1e-3) / 1e12, 2)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU per pass (BASELINE configs[1]: 32)")
    ap.add_argument("--weights", default="fp16", choices=["fp32", "bf16", "fp16"], help="denoiser weight image dtype")
    ap.add_argument("--vae", default="fp16", choices=["fp32", "fp16"], help="VAE MFMA operand type (fp32 = exact parity path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the MPJPE-vs-fp32-path check and the fp32 timing")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the passes are dealt over (each with its own model instance); 1 = the named configuration, "
                         ">1 = several B-sized batches in flight on one GPU (a B=32 pass occupies 32 of 256 CUs)")
    ap.add_argument("--graph", action="store_true",
                    help="capture each stream's pass (47 launches) as one hipGraph and time replays: the serving configuration, "
                         "host cost per pass ~15 us instead of ~2.5 ms of Python")
    ap.add_argument("--mode", default="sample", choices=["sample", "train"],
                    help="sample = the headline metric (BASELINE configs[1]); train = the stage-2 training step (configs[2] / configs[3])")
    ap.add_argument("--train-config", default=None, choices=["scene", "gimo", "egobody", "vae"],
                    help="--mode train: scene = scene + interactee (configs[2], default at 1 GPU), gimo = config_mld_gimo scene-only "
                         "(configs[3], default at N > 1), egobody = interactee only")
    ap.add_argument("--points", type=int, default=20000, help="--mode train: points per scene cloud")
    ap.add_argument("--cluster", default=None, choices=["auto", "0", "2", "4", "8"],
                    help="CUs per sample in the sampling kernel (default: the model's policy, 8 / 4 / 2 while B x C <= 256; 0 = one CU per sample)")
    ap.add_argument("--scheduler", default="ddim", choices=["ddim", "ddpm"], help="ddpm = 1000-step ancestral sampling (BASELINE configs[4])")
    args = ap.parse_args()
    if args.mode == "train" and args.batch == 32 and "--batch" not in sys.argv:
        args.batch = 64                                     # BASELINE configs[2] / [3]: 64 sequences per GPU

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:    # no launcher: start the ranks ourselves, GPU untouched here
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs an MI355X (no CPU fallback on the product path) [rank {rank} of {world}]")
    # rehearsal on a one-GPU box: SEEME_BENCH_BACKEND=gloo SEEME_BENCH_DEVICE=0 puts every rank on one card
    backend = os.environ.get("SEEME_BENCH_BACKEND", "nccl")
    if "SEEME_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SEEME_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.mode == "train":
        res = train_mode(args, dev, rank, world, backend, dist_on)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist_on:
            dist.destroy_process_group()
        return

    B = args.batch
    S = max(1, args.streams)
    # (S > 1, batches in flight side by side: the sampling kernel's workgroups are dealt to all XCDs -- pack_xcds = 8 below)
    models = [build_models(dev, args.weights, args.vae) for _ in range(S)]
    if S > 1:       # ... and one CU per sample: a cluster launch wants the whole chip for ONE batch
        for _, d, _ in models:
            d.cluster, d.pack_xcds = 0, 8
    if args.cluster is not None:
        for _, d, _ in models:
            d.cluster = args.cluster if args.cluster == "auto" else int(args.cluster)
    vae, den, sch = models[0]
    streams = [torch.cuda.Stream() for _ in range(S)]     # (side streams only: the legacy default stream serialises with all others)
    n_infer = DDIM_STEPS
    if args.scheduler == "ddpm":
        from seeme_amd.schedulers import DDPMScheduler
        sch = DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                            variance_type="fixed_small", clip_sample=False)
        sch.set_timesteps(1000)
        n_infer = 1000
        models = [(v, d, sch) for v, d, _ in models]
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    motion = torch.randn(B, T_FRAMES, NFEATS, generator=g).to(dev)
    latents = torch.randn(B, 1, 256, generator=g).to(dev)
    lengths = [T_FRAMES] * B

    def run_pass(i, ev=None):
        v, d, sc = models[i % S]
        if S == 1:
            return one_pass(v, d, sc, motion, latents, lengths, ev)
        with torch.cuda.stream(streams[i % S]):
            return one_pass(v, d, sc, motion, latents, lengths, ev)

    for i in range(max(args.warmup, S if S > 1 else 0)):
        run_pass(i)
    torch.cuda.synchronize()
    graphs, gouts, gev = [], [], []
    if args.graph:
        for si in range(S):
            v, d, sc = models[si]
            gr = torch.cuda.CUDAGraph()
            ev2 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            with torch.cuda.graph(gr, stream=streams[si]):
                gouts.append(one_pass(v, d, sc, motion, latents, lengths))
            graphs.append(gr)
        torch.cuda.synchronize()

        def run_pass(i, ev=None):      # noqa: F811  (replay; the kernel-level event pair cannot live inside a captured graph)
            with torch.cuda.stream(streams[i % S]):
                if ev is not None:
                    ev[0].record()
                graphs[i % S].replay()
                if ev is not None:
                    ev[1].record()
            return gouts[i % S]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = run_pass(i, evs[i])
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert os.environ.get("SEEME_DEBUG_NOCHECK") or torch.isfinite(out).all()   # (debug timing builds produce garbage)
    for _, d, _ in models:                                  # no cluster of the sampling kernel may have given up waiting for a peer
        code, _local = d.cluster_status()
        assert code == 0, f"k_den_cluster: exchange timed out (code {code}): the results of this run are invalid"

    # dominant kernel: the persistent DDIM kernel
    loop_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if args.graph:      # events bracket whole replays there: take the kernel time from the eager warm-up instead
        for _ in range(2):  # (the first eager pass after the capture allocates its tables outside the graph's pool: timed on the second)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            one_pass(vae, den, sch, motion, latents, lengths, (e0, e1))
            torch.cuda.synchronize()
        loop_ms = e0.elapsed_time(e1)
    alg_bytes = den_algorithmic_bytes(den, B, 1, n_infer)
    achieved = alg_bytes / (loop_ms * 1e-3) / 1e9
    from seeme_amd.mld_denoiser import _device_cus
    Cc, spc = den._cluster_plan(B, 1, False, False, _device_cus(dev))
    if Cc:          # one sample (or, above B = 64, up to 8) split over Cc CUs: a CU streams its slices + the replicated FFN matrices
        exe_cu = den_cluster_bytes_per_cu(den, Cc, n_infer)
        cus = ((B + spc - 1) // spc + 7) // 8 * 8 * Cc
        exe_bytes = exe_cu * cus
        kname = (f"k_den_cluster (persistent DDIM loop, one sample split over {Cc} CUs)" if spc == 1 else
                 f"k_den_cluster_ms (persistent loop, clusters of {Cc} CUs that own {spc} samples each)")
    else:
        exe_bytes = exe_cu = den_executed_bytes(den, B, 1, n_infer)
        cus = min(B, 256)                               # one workgroup (one CU) per sample chain
        kname = "k_den_sample (persistent DDIM loop, one CU per sample)"
    per_cu = exe_cu / (loop_ms * 1e-3) / 1e9            # every CU streams its bytes itself (from L2 / Infinity Cache)

    if rank == 0:
        res = {
            "metric": "sampled seqs/sec (T=196, 50 DDIM steps)" if args.scheduler == "ddim" else "sampled seqs/sec (T=196, 1000 DDPM steps)",
            "value": round(world * B * args.steps / dt, 2),
            "unit": "seqs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if (args.weights == "fp32" and args.vae == "fp32") else
                     f"f32 accumulate ({args.weights} denoiser weights, {args.vae} VAE MFMA operands)",
            "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if dist_on else 1,
            "collective_backend": (backend if dist_on else None),
            "config": {"workload": f"config_mld_egobody interactee-only: VAE encode -> " + (f"{n_infer}-step DDIM" if args.scheduler == "ddim" else f"{n_infer}-step DDPM (ancestral)")
                                   + f" -> VAE decode, B={B}/GPU, T=196, nfeats=132, random-init recipe weights",
                       "batch_per_gpu": B, "seq_len": T_FRAMES, "ddim_steps": n_infer,
                       "parallelism": f"dp{world} (independent shards, no collective on the data path)"
                                      + (f", {S} batches in flight per GPU" if S > 1 else "") + (", hipGraph replay" if args.graph else "")},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(args.weights, B, n_infer, Cc if spc == 1 else 0) if spc == 1 else None,
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of this configuration; a committed constant, not observed in this run)",
                         "ms_per_launch": round(loop_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                         # what the CUs actually stream (from L2 / Infinity Cache, never HBM: the image is 9-20 MB) and how close
                         # each CU's L1 fill path runs to its measured ceiling of 118 GB/s (scripts/stream_probe3.py)
                         "executed_bytes_per_launch": int(exe_bytes),
                         "per_cu_stream": {"achieved": round(per_cu, 2), "peak": 118.0, "unit": "GB/s per CU",
                                           "frac": round(per_cu / 118.0, 4), "cus_busy": cus}},
        }
        if world == 1 and S == 1 and not args.graph and args.scheduler == "ddim" and not args.no_parity_check:
            res["mpjpe_vs_ref"] = mpjpe_check(dev, args.weights, args.vae, B)
            if not (args.weights == "fp32" and args.vae == "fp32"):      # the parity configuration, timed the same way
                pv, pd, ps = build_models(dev, "fp32", "fp32")
                for i in range(args.warmup):
                    one_pass(pv, pd, ps, motion, latents, lengths)
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for i in range(args.steps):
                    one_pass(pv, pd, ps, motion, latents, lengths)
                torch.cuda.synchronize()
                res["parity_mode"] = {"dtype": "f32", "value": round(B * args.steps / (time.perf_counter() - tp), 2), "unit": "seqs/s"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(B)
            # the GPU side of BASELINE configs[0] (VAE reconstruct, B = 4), fp32 parity path, beside its CPU leg
            pv = build_models(dev, "fp32", "fp32")[0]
            m4, l4 = motion[:4].contiguous(), [T_FRAMES] * 4
            for _ in range(3):
                pv.decode(pv.encode_dist(m4, l4)[0].unsqueeze(0), l4)
            torch.cuda.synchronize()
            tq, nq = time.perf_counter(), 50
            for _ in range(nq):
                pv.decode(pv.encode_dist(m4, l4)[0].unsqueeze(0), l4)
            torch.cuda.synchronize()
            res["cfg1_vae_reconstruct"] = {"value": round(4 * nq / (time.perf_counter() - tq), 1), "unit": "seqs/s", "dtype": "f32",
                                           "on": "config_vae_egobody VAE reconstruct (encode -> decode(mu)), B=4, T=196, F=132, fp32 parity path"}
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
