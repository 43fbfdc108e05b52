"""Debug: which part of the stage-2 training step survives hipGraph capture (each stage in a child process)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAGES = ["api_e2_w1", "seed_api_e2_w1", "pts_api_e2_w1", "seedpts_api_e2_w1"]


def child(stage):
    import torch
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule
    from seeme_amd.smpl import SMPL
    from seeme_amd.weights_recipe import load_recipe_
    dev = torch.device("cuda", 0)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = parse_config(os.path.join(repo, "configs", "config_mld_egobody.yaml"))
    if stage == "fwd_bwd_twin":
        cfg.TRAIN.HIP_BACKWARD = False
    n_points = 2048
    if stage.startswith("seed"):
        torch.manual_seed(7)
        stage = stage.split("_", 1)[1] if not stage.startswith("seedpts") else "pts_" + stage.split("_", 1)[1]
    if stage.startswith("pts"):
        n_points = 384
        stage = stage.split("_", 1)[1]
    dm = SyntheticEgoDataModule(nfeats=75, T=16, n_points=n_points, device=dev, pose_dim=72)
    if stage.startswith("prev"):           # what an earlier test of the same process leaves behind
        import gc
        kind = stage.split("_")[0]
        stage = stage.split("_", 1)[1]
        prev = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
        load_recipe_(prev.vae), load_recipe_(prev.denoiser)
        prev = prev.to(dev).train()
        if kind == "prevrot":
            prev.eval()
            prev.ego_eval(dm.batch(2, idx=6, lengths=[16, 13]))
        else:
            prev.configure_optimizers()
            for _ in range(3):
                prev.optimizer_step(prev.training_step(dm.batch(4, idx=3)))
        torch.cuda.synchronize()
        if kind in ("prevdel", "prevgc", "prevrot"):
            del prev
        if kind == "prevgc":
            gc.collect(); torch.cuda.synchronize()
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser)
    model = model.to(dev).train()
    tb = dm.batch(4, idx=3)
    model.configure_optimizers()
    n_eager, n_warm = 3, 2
    if "_e" in stage:
        n_eager, n_warm = int(stage.split("_e")[1][0]), int(stage.split("_w")[1][0])
    for _ in range(n_eager):
        model.optimizer_step(model.training_step(tb))
    torch.cuda.synchronize()
    if stage.startswith("api"):
        replay = model.capture_training_step(tb, warmup=n_warm)
        torch.cuda.synchronize()
        for _ in range(3):
            out = replay()
        torch.cuda.synchronize()
        print(f"stage {stage}: OK, value {float(out):.6f}", flush=True)
        return
    losses = model.losses["train"]
    fused = model._fused_adamw

    def body():
        if stage == "vae_encode":
            f = model._wearer_features(tb[0].float(), tb[1].float(), 0)
            return model.vae.encode_dist(f, [16] * 4).sum()
        if stage == "forward_nograd":
            with torch.no_grad():
                return losses.update(model.train_diffusion_forward(tb), accumulate=False)
        loss = losses.update(model.train_diffusion_forward(tb), accumulate=False)
        if stage == "forward":
            return loss.detach()
        if stage in ("fwd_bwd", "fwd_bwd_twin"):
            for p in model.trainable_parameters():
                p.grad = None
            loss.backward()
            return loss.detach()
        if stage == "fwd_bwd_bucket":
            model.backward(loss)
            return loss.detach()
        if stage == "adamw_only":
            fused.step(device_step=True)
            return loss.detach()
        model.backward(loss)
        fused.step(device_step=True)
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n_warm):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"stage {stage}: OK, value {float(out):.6f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for s in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "fault" in l.lower()][:3]
            print(f"{s}: rc={r.returncode} {tail} {err}", flush=True)
