"""GPU parity of the ORCHESTRATION (seeme_amd.mld.MLD on the HIP path) against the flow oracle (oracle/mld_flows.py),
with every random draw injected: stage-2 forward with classifier-free masks (a20), stage-1 forward and losses (a21, a24),
ego_eval in its 'vae' / classifier-free / rot6d / GIMO forms (a22, f4), the GIMO configuration (BASELINE configs[3]),
and the training-step plumbing (flat gradient bucket, graph-captured step)."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, rel_err
from oracle import mld_flows as F
from oracle import mld_oracle as O
from seeme_amd import shapes
from seeme_amd.weights_recipe import load_recipe_, recipe_state_dict

pytestmark = pytest.mark.gpu
TOL_F32 = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def _mld(dev, cfg_name, T=16, n_points=384, mutate=None):
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule
    from seeme_amd.smpl import SMPL
    cfg = parse_config(os.path.join(REPO, "configs", cfg_name))
    if mutate:
        mutate(cfg)
    dm = SyntheticEgoDataModule(nfeats=cfg.model.nfeats, T=T, n_points=n_points, device=dev,
                                pose_dim=cfg.model.nfeats - (3 if cfg.TRAIN.ABLATION.PREDICT_TRANSL else 0))
    torch.manual_seed(7)                                    # output_scene keeps torch's init: make it reproducible
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser)
    if hasattr(model, "proscene"):
        load_recipe_(model.proscene.scene_enc)
    return model.to(dev), dm, cfg


def _np(t):
    return t.detach().cpu().numpy()


def _oracle_params(model, nfeats):
    Pv, Pd = recipe_state_dict(shapes.vae_shapes(nfeats)), recipe_state_dict(shapes.denoiser_shapes())
    Ppn = recipe_state_dict(shapes.pointnet_shapes())
    Pos = {k: _np(v) for k, v in model.output_scene.state_dict().items()} if hasattr(model, "output_scene") else None
    return Pv, Pd, Ppn, Pos


def _gen(seed, dev):
    g = torch.Generator().manual_seed(seed)
    return lambda *shape: torch.randn(*shape, generator=g).to(dev), lambda p, *shape: (torch.rand(*shape, generator=g) < p).to(dev)


# ----------------------------------------------------------------------------- a20: stage-2 forward, injected draws
@pytest.mark.parametrize("cfg_name,guidance", [("config_mld_scene.yaml", 7.5), ("config_mld_gimo.yaml", 1.0),
                                               ("config_mld_egobody.yaml", 1.0)])
def test_train_diffusion_forward_vs_oracle(dev, cfg_name, guidance):
    """train_diffusion_forward (mld.py:887-1017 + 582-631): batch unpack by condition, classifier-free INPUT masks on
    scene points and interactee features, the two frozen-VAE samples, condition assembly [z_cond, scene], add_noise and
    the denoiser -- against the oracle chain on identical inputs, masks and noises.  config_mld_gimo = BASELINE
    configs[3]'s model: nfeats 69, scene-only condition (N = 1)."""
    def mut(cfg):
        cfg.model.guidance_scale = guidance
    model, dm, cfg = _mld(dev, cfg_name, mutate=mut)
    model.eval()                                             # (eval: no dropout anywhere; the forward is the training one)
    B, T, nf = 3, 16, cfg.model.nfeats
    with_scene = "scene" in cfg.model.condition
    batch = dm.batch(B, idx=5, with_scene=with_scene)
    assert batch[0].shape[-1] == nf - 3
    rn, rm = _gen(21, dev)
    eps_z, eps_c, noise = rn(1, B, 256), rn(1, B, 256), rn(B, 1, 256)
    ts = torch.tensor([999, 0, 417], device=dev)
    m_scene = rm(0.1, B, dm.n_points, 3) if with_scene else None
    m_int = rm(0.1, B, T, nf)
    with torch.no_grad():
        rs = model.train_diffusion_forward(batch, noise=noise, timesteps=ts, eps=(eps_z, eps_c), masks=(m_scene, m_int))
    Pv, Pd, Ppn, Pos = _oracle_params(model, nf)
    want = F.train_diffusion_forward(
        Pv, Pd, _np(batch[0]), _np(batch[1]), condition=tuple(cfg.model.condition), eps_z=_np(eps_z), eps_c=_np(eps_c),
        noise=_np(noise), timesteps=_np(ts), scene=_np(batch[4]) if with_scene else None, Ppn=Ppn, Pos=Pos,
        mask_scene=_np(m_scene) if with_scene else None, mask_int=_np(m_int), guidance_scale=guidance)
    n_tokens = len([c for c in cfg.model.condition if c in ("scene", "interactee")])
    assert want["cond_emb"].shape[0] == n_tokens
    assert rel_err(_np(rs["noise_pred"]), want["noise_pred"]) < 2 * TOL_F32
    loss = float(model.losses["train"].update(rs, accumulate=False))
    assert abs(loss - F.mld_losses(want, "diffusion", {"LAMBDA_REC": 1.0})["total"]) < 1e-4 * max(1.0, loss)
    if guidance > 1.0:      # the masks matter: without them the prediction changes
        with torch.no_grad():
            rs2 = model.train_diffusion_forward(batch, noise=noise, timesteps=ts, eps=(eps_z, eps_c),
                                                masks=(torch.zeros_like(m_scene), torch.zeros_like(m_int)))
        assert rel_err(_np(rs2["noise_pred"]), want["noise_pred"]) > 1e-3


# ----------------------------------------------------------------------------- a21 / a24: stage-1 forward and losses
@pytest.mark.parametrize("cfg_name", ["config_vae_egobody.yaml", "config_vae_gimo.yaml"])
def test_train_vae_forward_and_losses_vs_oracle(dev, cfg_name):
    """train_vae_forward (mld.py:633-885) and MLDLosses.update (losses/mld.py:113-156) with non-unit lambdas: renormed
    m_ref / m_rst, SMPL joints ([:21] and the reference's global orientation for GIMO), SmoothL1 / KL terms, total --
    on the no-grad HIP path and on the differentiable path used for training."""
    def mut(cfg):
        cfg.LOSS.LAMBDA_KL, cfg.LOSS.LAMBDA_REC, cfg.LOSS.LAMBDA_JOINT, cfg.LOSS.LAMBDA_ROOT = 3e-3, 0.7, 1.9, 0.31
    model, dm, cfg = _mld(dev, cfg_name, mutate=mut)
    B, nf = 3, cfg.model.nfeats
    dataset = cfg.DATASET_NAME
    batch = dm.batch(B, idx=2)
    rn, _ = _gen(4, dev)
    eps = rn(1, B, 256)
    Pv = recipe_state_dict(shapes.vae_shapes(nf))
    smpl = O.make_synthetic_smpl(1234)
    want = F.train_vae_forward(Pv, smpl, _np(batch[0]), _np(batch[1]), _np(batch[2]), _np(dm.mean), _np(dm.std), _np(eps),
                               dataset=dataset)
    lam = {k: float(cfg.LOSS[k]) for k in ("LAMBDA_KL", "LAMBDA_REC", "LAMBDA_JOINT", "LAMBDA_ROOT")}
    wl = F.mld_losses(want, "vae", lam)
    nj = 21 if dataset == "gimo" else 24
    for grad in (False, True):
        model.train(grad)
        model.vae.eval()                                     # dropout off: parity needs the deterministic forward
        with torch.set_grad_enabled(grad):
            rs = model.train_vae_forward(batch, eps=eps)
            assert rs["joints_rst"].shape == (B, 16, nj, 3)
            assert rel_err(_np(rs["m_rst"]), want["m_rst"]) < 2 * TOL_F32 and rel_err(_np(rs["m_ref"]), want["m_ref"]) < 1e-6
            assert rel_err(_np(rs["joints_ref"]), want["joints_ref"]) < TOL_F32
            assert rel_err(_np(rs["joints_rst"]), want["joints_rst"]) < 5 * TOL_F32
            L = model.losses["train"]
            L.reset()
            tot = L.update(rs)
            total = float(tot.detach())
        if grad:       # the feature term must reach the DECODER through the renorm (a HIP op with its own backward)
            ps = [model.vae.final_layer.weight, model.vae.skel_embedding.weight]     # (the HIP backward writes .grad itself)
            for q in model.vae.parameters():
                q.grad = None
            tot.backward()
            assert all(q.grad is not None and torch.isfinite(q.grad).all() and float(q.grad.abs().sum()) > 0 for q in ps)
        got = L.compute()
        for k in ("recons_feature", "recons_joints", "recons_transl", "kl_motion"):
            assert abs(got[k] - wl[k]) < 5e-4 * max(1e-3, abs(wl[k])), (grad, k, got[k], wl[k])
        assert abs(total - wl["total"]) < 5e-4 * abs(wl["total"]), (grad, total, wl["total"])


# ----------------------------------------------------------------------------- ego_eval forms
def test_ego_eval_vae_stage_validation_and_test_steps(dev):
    """STAGE vae (configs/config_vae_egobody.yaml, condition [text]): ego_eval encodes the target motion and decodes it
    (mld.py:1328-1360) -- no condition tokens, no reverse diffusion -- and validation_step / test_step score that."""
    model, dm, cfg = _mld(dev, "config_vae_egobody.yaml")
    model.eval()
    B = 3
    lengths = [16, 11, 16]
    batch = dm.batch(B, idx=9, lengths=lengths)
    rn, _ = _gen(8, dev)
    eps = rn(1, B, 256)
    rs = model.ego_eval(batch, cond_noise=eps)
    Pv = recipe_state_dict(shapes.vae_shapes(75))
    want = F.ego_eval(Pv, None, O.make_synthetic_smpl(1234), _np(batch[0]), _np(batch[1]), _np(batch[2]), lengths,
                      _np(dm.mean), _np(dm.std), stage="vae", eps_c=_np(eps))
    assert rel_err(_np(rs["m_rst"]), want["m_rst"]) < 2 * TOL_F32
    assert rel_err(_np(rs["joints_rst"]), want["joints_rst"]) < 5 * TOL_F32
    model.EgoMetric.reset()
    loss = model.validation_step(batch)                      # 'val': loss of train_vae_forward + metrics of ego_eval
    assert torch.isfinite(loss)
    model.EgoMetric.reset()
    torch.manual_seed(3)
    out = model.test_step(batch)
    assert out.shape == (B, 16, 24, 3) and np.isfinite(model.EgoMetric.compute()["MPJPE"])
    # metrics of the deterministic reconstruction vs the oracle's
    model.sample_mean = True
    model.EgoMetric.reset()
    model.validation_step(batch)
    got = model.EgoMetric.compute()
    wm = F.ego_eval(Pv, None, O.make_synthetic_smpl(1234), _np(batch[0]), _np(batch[1]), _np(batch[2]), lengths,
                    _np(dm.mean), _np(dm.std), stage="vae", eps_c=np.zeros((1, B, 256), np.float32))
    m = O.ego_metrics(wm["joints_rst"].astype(np.float64), wm["joints_ref"].astype(np.float64), wm["orientation_quat_rst"].astype(np.float64),
                      wm["orientation_quat_ref"].astype(np.float64), lengths, "val")
    assert abs(got["MPJPE"] - m["MPJPE"]) < 1e-3 * max(1.0, m["MPJPE"])


def test_ego_eval_cfg_scene_reference_order_vs_oracle(dev):
    """Classifier-free guidance with scene + interactee (mld.py:1144-1158, 1283-1290, 489): interactee tokens
    [uncond, cond], scene tokens [cond, uncond] -- the reference's layout, reproduced by default -- vs the oracle; the
    opt-in TEST.CFG_SCENE_ORDER 'fixed' differs."""
    def mut(cfg):
        cfg.model.guidance_scale = 2.5
        cfg.model.scheduler.num_inference_timesteps = 10
    model, dm, cfg = _mld(dev, "config_mld_scene.yaml", mutate=mut)
    model.eval()
    B, lengths = 2, [16, 16]
    batch = dm.batch(B, idx=4, with_scene=True)
    rn, _ = _gen(15, dev)
    lat, e_c, e_u = rn(B, 1, 256), rn(1, B, 256), rn(1, B, 256)
    rs = model.ego_eval(batch, latents=lat, cond_noise=(e_c, e_u))
    Pv, Pd, Ppn, Pos = _oracle_params(model, 75)
    want = F.ego_eval(Pv, Pd, O.make_synthetic_smpl(1234), _np(batch[0]), _np(batch[1]), _np(batch[2]), lengths, _np(dm.mean),
                      _np(dm.std), condition=("text", "scene", "interactee"), latents=_np(lat), eps_c=_np(e_c), eps_u=_np(e_u),
                      scene=_np(batch[4]), Ppn=Ppn, Pos=Pos, guidance_scale=2.5, steps=10)
    assert rel_err(_np(rs["lat_t"]), want["lat_t"]) < 5 * TOL_F32
    assert rel_err(_np(rs["joints_rst"]), want["joints_rst"]) < 1e-3
    model.cfg_scene_order = "fixed"
    rs_fixed = model.ego_eval(batch, latents=lat, cond_noise=(e_c, e_u))
    assert rel_err(_np(rs_fixed["lat_t"]), want["lat_t"]) > 1e-3


def test_ego_eval_rot6d_vs_oracle(dev):
    """DATA_TYPE rot6d (mld.py:161-163, 1410-1449): 24 x 6 features -> rotation matrices -> SMPL with pose2rot=False, no
    betas / translation; the reference builds that body model in float64, the HIP layer is fp32 (tolerance 1e-4)."""
    def mut(cfg):
        cfg.DATA_TYPE = "rot6d"
        cfg.model.nfeats = 144
        cfg.model.motion_vae.params.nfeats = cfg.model.denoiser.params.nfeats = 144   # (${model.nfeats} is resolved at parse)
        cfg.TRAIN.ABLATION.PREDICT_TRANSL = False
        cfg.model.scheduler.num_inference_timesteps = 5
    model, dm, cfg = _mld(dev, "config_mld_egobody.yaml", mutate=mut)
    model.eval()
    B, lengths = 2, [16, 13]
    batch = dm.batch(B, idx=6, lengths=lengths)
    assert batch[0].shape[-1] == 144
    rn, _ = _gen(2, dev)
    lat, e_c = rn(B, 1, 256), rn(1, B, 256)
    rs = model.ego_eval(batch, latents=lat, cond_noise=e_c)
    Pv, Pd = recipe_state_dict(shapes.vae_shapes(144)), recipe_state_dict(shapes.denoiser_shapes())
    want = F.ego_eval(Pv, Pd, O.make_synthetic_smpl(1234), _np(batch[0]), _np(batch[1]), _np(batch[2]), lengths, _np(dm.mean),
                      _np(dm.std), latents=_np(lat), eps_c=_np(e_c), data_type="rot6d", predict_transl=False, steps=5)
    assert rs["orientation_quat_rst"] is None and rs["joints_rst"].shape == (B, 16, 24, 3)
    assert rel_err(_np(rs["m_rst"]), want["m_rst"]) < 5 * TOL_F32
    assert rel_err(_np(rs["joints_ref"]), want["joints_ref"]) < TOL_F32
    assert rel_err(_np(rs["joints_rst"]), want["joints_rst"]) < 1e-3


def test_gimo_config_eval_and_training_step(dev):
    """config_mld_gimo (BASELINE configs[3]): 69 features (root + 21 joints + translation, padded to SMPL's 23 joints),
    scene-only condition (one token: the tabulated ca_block path of the kernel), ego_eval keeps 24 joints and the
    predicted orientation (mld.py:1655-1745); a few optimiser steps through the flat gradient bucket reduce the loss."""
    def mut(cfg):
        cfg.model.scheduler.num_inference_timesteps = 10
        cfg.TRAIN.OPTIM.LR = 1e-3
    model, dm, cfg = _mld(dev, "config_mld_gimo.yaml", mutate=mut)
    assert model.vae.nfeats == 69 and list(cfg.model.condition) == ["text", "scene"]
    model.eval()
    B, lengths = 3, [16, 16, 9]
    batch = dm.batch(B, idx=1, with_scene=True, lengths=lengths)
    rn, _ = _gen(31, dev)
    lat = rn(B, 1, 256)
    rs = model.ego_eval(batch, latents=lat)
    Pv, Pd, Ppn, Pos = _oracle_params(model, 69)
    want = F.ego_eval(Pv, Pd, O.make_synthetic_smpl(1234), _np(batch[0]), _np(batch[1]), _np(batch[2]), lengths, _np(dm.mean),
                      _np(dm.std), condition=("text", "scene"), latents=_np(lat), scene=_np(batch[4]), Ppn=Ppn, Pos=Pos,
                      dataset="gimo", steps=10)
    assert rs["joints_rst"].shape == (B, 16, 24, 3)
    assert rel_err(_np(rs["lat_t"]), want["lat_t"]) < 5 * TOL_F32
    assert rel_err(_np(rs["joints_rst"]), want["joints_rst"]) < 1e-3
    model.EgoMetric.reset()
    model.test_step(batch)
    assert np.isfinite(model.EgoMetric.compute()["MPJPE"])
    # training: scene-only condition through the HIP backward, gradients in one flat buffer
    model.train()
    tb = dm.batch(4, idx=3, with_scene=True)
    g = torch.Generator().manual_seed(11)
    noise, ts = torch.randn(4, 1, 256, generator=g).to(dev), torch.randint(0, 1000, (4,), generator=g).to(dev)
    eps = (torch.randn(1, 4, 256, generator=g).to(dev), None)
    losses = []
    for _ in range(8):
        loss = model.losses["train"].update(model.train_diffusion_forward(tb, noise=noise, timesteps=ts, eps=eps))
        model.optimizer_step(loss)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    bucket = model.grad_bucket()
    assert bucket.n_pack > 0 and all(p.grad.data_ptr() == bucket.views[id(p)].data_ptr() for p in bucket.params)
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + 4 * bucket.flat.numel()
    off_path = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert off_path == ["denoiser.mem_pos.pe"]                 # never used by trans_enc: no gradient, never decayed
    assert all(lo <= p.grad.data_ptr() < hi for p in model.trainable_parameters() if p.grad is not None)
    assert bucket.flat.numel() == sum(p.numel() for p in model.trainable_parameters() if p.grad is not None)


# ----------------------------------------------------------------------------- training-step plumbing
def test_grad_bucket_step_equals_separate_gradients(dev):
    """optimizer_step (flat bucket: views as .grad, HIP backward writing its block in place, one-launch AdamW) gives the
    same parameters as the plain path (zero_grad, backward into separate tensors, torch.optim.AdamW.step)."""
    res = []
    for bucketed in (True, False):
        model, dm, cfg = _mld(dev, "config_mld_scene.yaml")
        model.train()
        tb = dm.batch(4, idx=3, with_scene=True)
        g = torch.Generator().manual_seed(5)
        noise, ts = torch.randn(4, 1, 256, generator=g).to(dev), torch.randint(0, 1000, (4,), generator=g).to(dev)
        eps = (torch.randn(1, 4, 256, generator=g).to(dev), torch.randn(1, 4, 256, generator=g).to(dev))
        model.configure_optimizers()
        for _ in range(3):
            loss = model.losses["train"].update(model.train_diffusion_forward(tb, noise=noise, timesteps=ts, eps=eps))
            if bucketed:
                model.optimizer_step(loss)
            else:
                model.optimizer.zero_grad(set_to_none=True)
                loss.backward()
                model.optimizer.step()
        res.append({k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad})
    worst = max(rel_err(_np(res[0][k]), _np(res[1][k])) for k in res[0])
    assert worst < 2e-5, worst


def test_capture_training_step_replay_equals_eager(dev):
    """capture_training_step: the stage-2 step as one hipGraph (frozen encoders, denoiser forward + backward, in-graph
    one-launch AdamW whose step count and learning rate live on the device).  An optimiser already exists before the
    capture (the case ADVICE r1 flagged); every replay must advance the step counts on both sides and apply exactly
    the AdamW update of the gradients it left in the flat bucket, including after an LR-scheduler edit."""
    model, dm, cfg = _mld(dev, "config_mld_egobody.yaml")
    model.train()
    tb = dm.batch(4, idx=3)
    model.configure_optimizers()
    loss = model.training_step(tb)
    model.optimizer_step(loss)                                 # eager steps first: optimiser state exists ...
    model.optimizer_step(model.training_step(tb))              # ... and so does the gradient bucket; step = 2
    replay = model.capture_training_step(tb, warmup=1)         # + 1 warm-up step
    torch.cuda.synchronize()
    bucket = model.grad_bucket()
    opt = model.optimizer
    for it in range(3):
        if it == 2:
            opt.param_groups[0]["lr"] = 0.5 * opt.param_groups[0]["lr"]         # what StepLR does at an epoch end
        p0 = {id(p): p.detach().clone() for p in bucket.params}
        m0 = {id(p): opt.state[p]["exp_avg"].clone() for p in bucket.params}
        v0 = {id(p): opt.state[p]["exp_avg_sq"].clone() for p in bucket.params}
        t = float(opt.state[bucket.params[0]]["step"]) + 1.0
        loss = replay()
        torch.cuda.synchronize()
        assert torch.isfinite(loss)
        assert float(opt.state[bucket.params[0]]["step"]) == t == 4.0 + it
        assert float(list(model._fused_adamw._dev_scalars.values())[0][0]) == t
        lr, (b1, b2), eps_, wd = (opt.param_groups[0][k] for k in ("lr", "betas", "eps", "weight_decay"))
        worst = 0.0
        for p in bucket.params:
            gth = bucket.views[id(p)].double()
            m = m0[id(p)].double() * b1 + (1 - b1) * gth
            v = v0[id(p)].double() * b2 + (1 - b2) * gth * gth
            want = p0[id(p)].double() * (1 - lr * wd) - lr / (1 - b1 ** t) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps_)
            worst = max(worst, float((p.detach().double() - want).abs().max() / want.abs().max().clamp_min(1e-12)))
        assert worst < 1e-5, (it, worst)
        assert float(bucket.flat.abs().max()) > 0


# ----------------------------------------------------------------------------- data module (f2) on the device
def test_data_module_feeds_training_and_eval(dev, tmp_path):
    """seeme_amd.data.EgoDataModule (the EgoBody on-disk layout, split resident in HBM) as the `datamodule` of MLD: renorm
    against numpy, a stage-2 training step and a test step on its batches (scene + interactee conditions)."""
    from test_data_module import write_dataset
    from seeme_amd import data as D
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD
    from seeme_amd.smpl import SMPL
    root = str(tmp_path / "egobody")
    write_dataset(root, "egobody", n=9, T=12, P=64, full_every=4)
    dm = D.EgoDataModule(root, "egobody", condition=("text", "scene", "interactee"), motion_length=12, device=dev, scene_root=root)
    assert dm.splits["train"].motion.is_cuda and dm.splits["train"].scene_table.is_cuda
    x = torch.randn(3, 12, 75, device=dev)
    mean, std = np.load(os.path.join(root, "mean.npy")), np.load(os.path.join(root, "std.npy"))
    assert rel_err(_np(dm.renorm(x)), _np(x) * std[0, :75] + mean[0, :75]) < 1e-6
    cfg = parse_config(os.path.join(REPO, "configs", "config_mld_scene.yaml"))
    cfg.model.scheduler.num_inference_timesteps = 5
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser), load_recipe_(model.proscene.scene_enc)
    model = model.to(dev).train()
    losses = []
    for it in range(3):
        batch = dm.batch(4, idx=it, split="train")
        assert batch[0].shape == (4, 12, 2, 72) and batch[4].shape == (4, 64, 3) and batch[5].shape == (4, 1)
        loss = model.training_step(batch)
        model.optimizer_step(loss)
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    model.eval()
    model.EgoMetric.reset()
    for b in dm.iterate("test", 4):
        out = model.test_step(b)
        assert out.shape[1:] == (12, 24, 3)
    assert np.isfinite(model.EgoMetric.compute()["MPJPE"])     # 'test' keeps only plausible sequences (compute.py:488-517):
    model.EgoMetric.reset()                                    # random weights leave none, so count on the 'val' rule
    for b in dm.iterate("test", 4):
        model.validation_step(b)
    got = model.EgoMetric.compute()
    assert np.isfinite(got["MPJPE"]) and got["count_seq"] > 0


def test_data_module_pose_estimation_and_gimo_on_device(dev, tmp_path):
    """(i) TEST.POSE_ESTIMATION_TASK on files: EgoDataModule batches end with the interactee's ground truth (dataset.py:1765-1781) and
    MLD.test_step consumes them (batch[:-3] / batch[-3:], mld.py:1119-1131) -- with and without EgoHMR estimates as the condition.
    (ii) GIMO: the scene vertices live on the device, every batch draws a new 20000-point sample (+ jitter in training,
    dataset.py:2013-2033), and a stage-2 training step runs on those batches."""
    import pickle
    from test_data_module import write_dataset
    from seeme_amd import data as D
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD
    from seeme_amd.smpl import SMPL
    root = str(tmp_path / "egobody")
    items, _ = write_dataset(root, "egobody", n=6, T=12, P=64, full_every=2)
    rng = np.random.default_rng(1)
    pred = {im: {"smpl_parameters": {"global_orient": rng.standard_normal(3), "body_pose": 0.3 * rng.standard_normal(69), "betas": rng.standard_normal(10)}}
            for (sp, _n), it in items.items() if sp == "test" for im in it["recording_utils"]["original_imgname"]}
    with open(os.path.join(root, "interactee_pred_test.pkl"), "wb") as f:
        pickle.dump(pred, f)
    cfg = parse_config(os.path.join(REPO, "configs", "config_mld_scene.yaml"))
    cfg.model.scheduler.num_inference_timesteps = 5
    cfg.TEST.POSE_ESTIMATION_TASK = True
    for with_pred in (False, True):
        dm = D.EgoDataModule(root, "egobody", condition=("text", "scene", "interactee"), motion_length=12, device=dev, scene_root=root,
                             pose_estimation_task=True, interactee_pred=with_pred, splits=("test",))
        model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
        load_recipe_(model.vae), load_recipe_(model.denoiser), load_recipe_(model.proscene.scene_enc)
        model = model.to(dev).eval()
        model.EgoMetric.reset()
        for b in dm.iterate("test", 3):
            assert len(b) == 9 and b[6].shape[1:] == (12, 1, 72) and b[7].shape[1:] == (1, 12, 3)
            out = model.test_step(b)
            assert out.shape[1:] == (12, 24, 3) and torch.isfinite(out).all()
        got = model.EgoMetric.compute()
        assert np.isfinite(got["mpjpe_interactee"]) and got["mpjpe_interactee"] >= 0
    # ---- GIMO
    groot = str(tmp_path / "gimo")
    write_dataset(groot, "gimo", n=6, T=12, P=300, full_every=2)
    dm = D.EgoDataModule(groot, "gimo", condition=("text", "scene"), motion_length=12, device=dev, scene_root=groot, scene_points=2000)
    tr = dm.splits["train"]
    assert tr.scene_flat.is_cuda and tr.scene_table is None
    b1, b2 = dm.batch(4, idx=0, split="train"), dm.batch(4, idx=0, split="train")
    assert b1[4].shape == (4, 2000, 3) and b1[4].is_cuda and not torch.equal(b1[4], b2[4]) and torch.equal(b1[0], b2[0])
    cfg = parse_config(os.path.join(REPO, "configs", "config_mld_gimo.yaml"))
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser), load_recipe_(model.proscene.scene_enc)
    model = model.to(dev).train()
    for it in range(2):
        loss = model.training_step(dm.batch(4, idx=it, split="train"))
        model.optimizer_step(loss)
        assert np.isfinite(float(loss))


@pytest.mark.parametrize("cfg_name,B", [("config_mld_scene.yaml", 5), ("config_mld_gimo.yaml", 3), ("config_mld_egobody.yaml", 70)])
def test_stage2_glue_matches_autograd_path(dev, cfg_name, B):
    """stage2_glue (hand-written HIP forward + backward of rsample / add_noise / time MLP / output_scene / condition and
    time tables, gradients accumulated straight into .grad) against the same step with those parts as torch ops under
    autograd (TRAIN.HIP_GLUE false; itself pinned to the CPU oracle above): loss, noise prediction and EVERY parameter
    gradient, for two tokens (interactee + scene), scene only and interactee only; B = 70 crosses a 64-row tile edge."""
    with_scene = "scene" in cfg_name or "gimo" in cfg_name
    got = []
    for glue in (True, False):
        def mut(cfg):
            cfg.TRAIN.HIP_GLUE = glue
        model, dm, cfg = _mld(dev, cfg_name, mutate=mut)
        model.train()
        tb = dm.batch(B, idx=3, with_scene=with_scene)
        g = torch.Generator().manual_seed(5)
        noise, ts = torch.randn(B, 1, 256, generator=g).to(dev), torch.randint(0, 1000, (B,), generator=g).to(dev)
        eps = (torch.randn(1, B, 256, generator=g).to(dev), torch.randn(1, B, 256, generator=g).to(dev))
        masks = None
        if model.do_classifier_free_guidance:
            ms = (torch.rand(B, dm.n_points, 3, generator=g) < 0.1).to(dev) if with_scene else None
            mi = (torch.rand(B, 16, cfg.model.nfeats, generator=g) < 0.1).to(dev) if "interactee" in model.condition else None
            masks = (ms, mi)
        out = []
        for it in range(2):                                  # twice: the second step runs on cached plans / descriptor tables
            for p in model.parameters():
                p.grad = None
            rs = model.train_diffusion_forward(tb, noise=noise, timesteps=ts, eps=eps, masks=masks)
            loss = model.losses["train"].update(rs)
            loss.backward()
            out.append((float(loss.detach()), rs["noise_pred"].detach().clone(),
                        {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}))
        assert (getattr(model, "_glue", None) is not None) == glue
        got.append(out)
    for it in range(2):
        (l1, n1, g1), (l0, n0, g0) = got[0][it], got[1][it]
        assert abs(l1 - l0) < 1e-5 * abs(l0), (it, l1, l0)
        assert rel_err(_np(n1), _np(n0)) < 1e-5
        assert set(g1) == set(g0), set(g1) ^ set(g0)
        # (a key bias of the linear attention has a mathematically zero gradient -- softmax over the tokens is shift
        # invariant -- so errors are measured against the largest gradient entry of the whole model, not per tensor)
        scale = max(float(v.abs().max()) for v in g0.values())
        errs = sorted(((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * scale), k) for k in g0), reverse=True)
        assert errs[0][0] < 5e-5, (it, errs[:6])


@pytest.mark.parametrize("cfg_name,B,T", [("config_vae_egobody.yaml", 3, 16), ("config_vae_gimo.yaml", 5, 70),
                                          ("config_vae_egobody.yaml", 64, 14)])
def test_vae_hip_backward_matches_autograd(dev, cfg_name, B, T):
    """Stage 1 through vae_train.py (hand-written HIP forward with saves + backward of the VAE encoder / decoder: grouped fp32
    GEMMs, LayerNorm / softmax / GELU backward kernels, weight gradients as split atomic reductions) against the same step on
    the PyTorch-autograd twin (TRAIN.HIP_VAE_BACKWARD false; pinned to the HIP forward and the oracle above): loss terms,
    m_rst, and EVERY parameter gradient, with ragged lengths; T = 70 crosses a 64-row tile edge; B = 64 makes the row counts
    multiples of 128, so the projections run on the specialised seeme_gemm128 kernel (both forms)."""
    got = []
    for hip in (True, False):
        def mut(cfg):
            cfg.TRAIN.HIP_VAE_BACKWARD = hip
        model, dm, cfg = _mld(dev, cfg_name, T=T, mutate=mut)
        model.train()
        model.vae.eval()                                     # dropout off on both sides (the twin's arithmetic)
        tb = dm.batch(B, idx=4)
        g = torch.Generator().manual_seed(11)
        eps = torch.randn(1, B, 256, generator=g).to(dev)
        out = []
        for it in range(2):                                  # twice: the second step runs on the recorded launch lists
            for p in model.parameters():
                p.grad = None
            rs = model.train_vae_forward(tb, eps=eps)
            loss = model.losses["train"].update(rs)
            loss.backward()
            out.append((float(loss.detach()), rs["m_rst"].detach().clone(),
                        {k: v.grad.detach().clone() for k, v in model.vae.named_parameters() if v.grad is not None}))
        assert (getattr(model, "_vae_tr", None) is not None) == hip
        got.append(out)
    for it in range(2):
        (l1, m1, g1), (l0, m0, g0) = got[0][it], got[1][it]
        assert abs(l1 - l0) < 1e-5 * abs(l0), (it, l1, l0)
        assert rel_err(_np(m1), _np(m0)) < 2e-5
        assert set(g0) <= set(g1), set(g0) - set(g1)
        scale = max(float(v.abs().max()) for v in g0.values())
        errs = sorted(((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * scale), k) for k in g0), reverse=True)
        assert errs[0][0] < 1e-4, (it, errs[:6])


def test_vae_hip_training_dropout_matches_twin_with_same_masks(dev):
    """Training mode: vae_train.py applies the reference's dropout sites (attention weights, dropout1/2/3, the FFN's inner
    dropout, the dropped-out single-key cross-attention weight; cross_attention.py:264-273,324-337) with keep-masks it draws
    itself.  With those very masks injected into the autograd twin, outputs and every parameter gradient agree; the masks keep
    ~90 % of the elements; eval mode draws none."""
    from seeme_amd.vae_train import VaeTrainer
    from seeme_amd.vae_autograd import vae_encode_torch, vae_decode_torch
    from test_gpu_parity import make_vae
    B, T = 3, 21
    lengths = [21, 13, 17]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, 75, generator=g).to(dev)
    eps, w1 = torch.randn(1, B, 256, generator=g).to(dev), torch.randn(B, T, 75, generator=g).to(dev)
    w2, w3 = torch.randn(1, B, 256, generator=g).to(dev), torch.randn(1, B, 256, generator=g).to(dev) * 0.1
    vae = make_vae(75, dev).train()
    p = float(vae.encoder.input_blocks[0].self_attn.dropout)
    assert p == 0.1

    def loss_of(mu, logvar, dec):
        z = mu + eps * logvar.exp().pow(0.5)
        feats = dec(z)
        return (feats * w1).sum() + (mu * w2).sum() + (logvar * w3).sum(), feats

    tr = VaeTrainer(vae)
    torch.manual_seed(0)
    mu, logvar = tr.encode(x, lengths)
    loss, feats = loss_of(mu, logvar, lambda z: tr.decode(z, lengths))
    pe, pd = tr.plans[(False, B, T, p)], tr.plans[(True, B, T, p)]
    masks_e = {f"{l}.{k}": sv[k].clone() for l, sv in enumerate(pe.sv) for k in ("mP", "m1", "mh", "m2")}
    masks_d = {f"{l}.{k}": sv[k].clone() for l, sv in enumerate(pd.sv) for k in ("mP", "m1", "mh", "m2", "mw", "mc")}
    keep = float(torch.cat([m.float().flatten() for m in masks_e.values()]).mean())
    assert abs(keep - (1 - p)) < 0.01, keep
    loss.backward()
    g1 = {k: v.grad.detach().clone() for k, v in vae.named_parameters() if v.grad is not None}
    out1 = (mu.detach().clone(), logvar.detach().clone(), feats.detach().clone())
    for q in vae.parameters():
        q.grad = None
    mu0, std0 = vae_encode_torch(vae, x, lengths, masks=masks_e)
    loss0, feats0 = loss_of(mu0, 2 * std0.log(), lambda z: vae_decode_torch(vae, z, lengths, masks=masks_d))
    loss0.backward()
    g0 = {k: v.grad.detach().clone() for k, v in vae.named_parameters() if v.grad is not None}
    assert rel_err(_np(out1[0]), _np(mu0)) < 2e-5 and rel_err(_np(out1[2]), _np(feats0)) < 2e-5
    assert abs(float(loss.detach()) - float(loss0.detach())) < 2e-5 * abs(float(loss0.detach()))
    scale = max(float(v.abs().max()) for v in g0.values())
    errs = sorted(((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * scale), k) for k in g0), reverse=True)
    assert errs[0][0] < 1e-4, errs[:6]
    vae.eval()                                               # eval mode: a plan without masks
    tr.encode(x, lengths)
    assert (False, B, T, 0.0) in tr.plans and not hasattr(tr.plans[(False, B, T, 0.0)], "masks")


@pytest.mark.parametrize("argv,checks", [
    (["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity-check", "--streams", "2", "--graph"],
     {"metric": "sampled seqs/sec (T=196, 50 DDIM steps)", "n_gpus": 1}),
    (["--steps", "2", "--warmup", "1", "--mode", "train", "--train-config", "gimo", "--points", "2048"],
     {"n_gpus": 1}),
    (["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-parity-check", "--scheduler", "ddpm", "--batch", "8"],
     {"n_gpus": 1}),
    (["--steps", "2", "--warmup", "1", "--mode", "train", "--train-config", "vae", "--batch", "8"],
     {"metric": "stage-1 (VAE) training seqs/sec (T=196, B=8/GPU)"}),
])
def test_bench_modes_emit_the_contract_line(dev, argv, checks):
    """bench.py as the driver runs it (a child process, one JSON line on stdout): the --streams / --graph sampling mode (each pass
    replayed as a hipGraph, two batches in flight), the stage-2 training mode on the GIMO configuration, the 1000-step DDPM
    configuration and the stage-1 (VAE) training mode -- every line carries the contract's fields, a roofline object and a positive value."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] == int(argv[1]) and "workload" in d["config"]
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1
    for k, v in checks.items():
        assert d[k] == v, (k, d[k])


def test_launch_mapping_knobs_do_not_change_results(dev, tmp_path):
    """SEEME_DEN_XCDS (which XCDs the sampling kernel's working workgroups sit on; default ceil(chains / 16)), SEEME_LAYER_ROWS
    (32- or 64-row workgroups of the per-layer VAE kernel) and SEEME_LAYER_W8_MAX (up to how many workgroups a launch runs its 32
    rows on 8 waves instead of 4) only move work around: VAE encode / decode and a 5-step DDIM sample on
    fixed inputs (scripts/launch_knobs_probe.py, one child process per setting -- the knobs are read once per process) must be
    bit-identical to the default.  B = 37 gives ragged lengths, an odd chain count and partial XCD rounds."""
    import subprocess
    import sys
    probe = os.path.join(REPO, "scripts", "launch_knobs_probe.py")
    outs = {}
    for tag, env in (("default", {}), ("xcds1", {"SEEME_DEN_XCDS": "1"}), ("xcds5", {"SEEME_DEN_XCDS": "5"}), ("xcds8", {"SEEME_DEN_XCDS": "8"}),
                     ("rows64", {"SEEME_LAYER_ROWS": "64"}), ("waves4", {"SEEME_LAYER_W8_MAX": "0"}), ("unfused", {"SEEME_VAE_FUSED": "0"})):
        f = str(tmp_path / (tag + ".pt"))
        r = subprocess.run([sys.executable, probe, f, "37"], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        outs[tag] = torch.load(f, weights_only=True)
    ref = outs["default"]
    assert all(torch.isfinite(v).all() for v in ref.values())
    for tag, o in outs.items():
        for k in ref:
            if tag == "unfused":      # the three-kernels-per-layer sequence is an independent implementation of the same fp16 path
                if k != "latent":     # (row-major K / V, per-sample first decoder layer): equal to fp16 rounding, not bit for bit
                    assert float((o[k] - ref[k]).abs().max()) < 1e-2 * max(1.0, float(ref[k].abs().max())), (tag, k)
                continue
            assert torch.equal(o[k], ref[k]), (tag, k, float((o[k] - ref[k]).abs().max()))


@pytest.mark.parametrize("mode_args,metric_word", [([], "sampled"), (["--mode", "train", "--batch", "8"], "training")])
def test_bench_two_ranks_rehearsal_on_one_gpu(dev, mode_args, metric_word):
    """`python bench.py --gpus 2` with no launcher in the environment: bench.py starts the two ranks itself (before any GPU call),
    they rendezvous on 127.0.0.1, time the steps between barriers, take the max over ranks and rank 0 prints ONE line with
    n_gpus = 2 and the whole-job value.  On a one-GPU box the ranks share the card over gloo (SEEME_BENCH_BACKEND / _DEVICE);
    with one rank per GPU the same code runs on RCCL.  Training mode: the flat gradient bucket goes through one all-reduce."""
    import json
    import subprocess
    import sys
    env = {**os.environ, "SEEME_BENCH_BACKEND": "gloo", "SEEME_BENCH_DEVICE": "0"}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity-check"] + mode_args
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo" and d["scaling"] == "weak"
    assert metric_word in d["metric"] and d["value"] > 0 and d["steps"] == 2


def test_cfg5_ddpm1000_at_its_own_size(dev):
    """BASELINE configs[4]: 1000-step DDPM (ancestral sampling, configs/modules_novae/scheduler.yaml:16-26 of the reference;
    mld.py:432-511), 4096 sequences over 8 GPUs = 512 per GPU, 16-bit weights, hipGraph-captured.
    (i) bench.py --scheduler ddpm --batch 512 --graph runs the per-GPU share at its own size and prints the contract line;
    (ii) what 1000 CHAINED steps do to the 16-bit weight image: on B = 32 sequences with injected initial latents, condition noise
    and step noise [1000,32,256] the fp16 image's final latent, decoded features and MPJPE against the fp32 image's (measured and
    printed; bounded loosely -- the ancestral noise re-injected at every step dominates the state, so rounding does not build up)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--scheduler", "ddpm", "--batch", "512", "--graph", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-parity-check"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "1000 DDPM steps" in d["metric"] and d["config"]["batch_per_gpu"] == 512 and d["config"]["ddim_steps"] == 1000
    assert d["value"] > 0 and d["roofline"]["ms_per_launch"] > 0
    print("cfg5 per-GPU share:", d["value"], "seqs/s,", d["ms_per_step"], "ms per pass of 512 sequences, sampling kernel", d["roofline"]["ms_per_launch"], "ms")

    from seeme_amd.mld import EgoMetrics
    def mut(cfg):
        cfg.model.scheduler.target = "seeme_amd.schedulers.DDPMScheduler"
        cfg.model.scheduler.num_inference_timesteps = 1000
        cfg.model.scheduler.params = {"num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
                                      "variance_type": "fixed_small", "clip_sample": False}
    model, dm, cfg = _mld(dev, "config_mld_egobody.yaml", T=196, mutate=mut)
    model.eval()
    B = 32
    batch = dm.batch(B, idx=3)
    g = torch.Generator().manual_seed(77)
    lat, eps = torch.randn(B, 1, 256, generator=g).to(dev), torch.randn(1, B, 256, generator=g).to(dev)
    noise = torch.randn(1000, B, 256, generator=g).to(dev)
    out = {}
    for wd in ("fp32", "fp16", "bf16"):
        model.denoiser.weight_dtype = wd
        rs = model.ego_eval(batch, latents=lat, cond_noise=eps, step_noise=noise)
        assert model.denoiser.cluster_status()[0] == 0
        out[wd] = (rs["lat_rst"] if "lat_rst" in rs else None, rs["m_rst"], rs["joints_rst"],
                   EgoMetrics.per_sequence(rs["joints_rst"], rs["joints_ref"], rs["lengths"])["MPJPE"].double().mean().item())
    for wd in ("fp16", "bf16"):
        fe = rel_err(_np(out[wd][1]), _np(out["fp32"][1]))
        j2j = float((out[wd][2] - out["fp32"][2]).norm(dim=-1).mean() * 1000.0)
        print(f"1000 DDPM steps, {wd} image vs fp32 image: decoded features rel err {fe:.3e}, joint-to-joint {j2j:.3f} mm, "
              f"MPJPE {out[wd][3]:.6f} vs {out['fp32'][3]:.6f} mm (delta {abs(out[wd][3] - out['fp32'][3]):.2e})")
        assert np.isfinite(fe) and fe < (2e-2 if wd == "fp16" else 1e-1)
        assert abs(out[wd][3] - out["fp32"][3]) < (1e-2 if wd == "fp16" else 5e-2)
