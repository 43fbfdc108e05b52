"""CPU tests (no GPU): the orchestration oracle (oracle/mld_flows.py) and the host-side restatements of the product
(MLDLosses, schedulers) against fixtures generated from the reference's own code (tests/golden/make_golden.py):

* losses.npz              -- mld/models/losses/mld.py MLDLosses.update with non-unit lambdas;
* schedulers_egohmr.npz   -- EgoHMR/diffusion/gaussian_diffusion.py ddim_sample / p_sample / q_sample on the alpha-bar
                             sequence MLD's scheduler configuration visits (the only scheduler arithmetic the reference
                             tree holds; diffusers itself is absent).
"""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import mld_flows as F
from oracle import mld_oracle as O

LAM_KEYS = ("LAMBDA_KL", "LAMBDA_REC", "LAMBDA_JOINT", "LAMBDA_ROOT")


def _lam(g):
    return dict(zip(LAM_KEYS, (float(v) for v in g["lambdas"])), LAMBDA_PRIOR=0.0, LAMBDA_GEN=1.0, LAMBDA_LATENT=1e-5)


def _rs(g):
    return {"m_ref": g["m_ref"], "m_rst": g["m_rst"], "joints_ref": g["joints_ref"], "joints_rst": g["joints_rst"],
            "dist_m": (g["mu"], g["std"]), "dist_ref": (np.zeros_like(g["mu"]), np.ones_like(g["std"])),
            "noise": g["noise"], "noise_pred": g["noise_pred"], "latent": g["latent"], "pred": g["pred"]}


def test_loss_primitives_match_torch():
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((5, 9)) * 2, rng.standard_normal((5, 9))
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    assert abs(F.smooth_l1(a, b) - float(torch.nn.functional.smooth_l1_loss(ta, tb))) < 1e-12
    assert abs(F.mse(a, b) - float(torch.nn.functional.mse_loss(ta, tb))) < 1e-12
    sq, sp = np.exp(rng.standard_normal((5, 9)) * 0.3), np.exp(rng.standard_normal((5, 9)) * 0.2)
    want = torch.distributions.kl_divergence(torch.distributions.Normal(ta, torch.from_numpy(sq)),
                                             torch.distributions.Normal(tb, torch.from_numpy(sp))).mean()
    assert abs(F.kl_normal(a, sq, b, sp) - float(want)) < 1e-12


def test_oracle_losses_match_reference_fixture():
    g = load_golden("losses.npz")
    lam = _lam(g)
    w = F.loss_weights(lam)
    # the reference's elif chain: 'recons' is tested before 'transl', 'joints' is overridden last (losses/mld.py:80-102)
    assert w["recons_transl"] == lam["LAMBDA_REC"] != lam["LAMBDA_ROOT"]
    assert w["recons_joints"] == lam["LAMBDA_JOINT"] and w["recons_feature"] == lam["LAMBDA_REC"]
    v = F.mld_losses(_rs(g), "vae", lam)
    for k in ("recons_feature", "recons_joints", "recons_transl", "kl_motion", "total"):
        assert abs(v[k] - float(g[f"vae_{k}"])) < 2e-6 * max(1.0, abs(float(g[f"vae_{k}"]))), (k, v[k], float(g[f"vae_{k}"]))
    d = F.mld_losses(_rs(g), "diffusion", lam)
    assert abs(d["inst_loss"] - float(g["diffusion_inst_loss"])) < 2e-6 and abs(d["total"] - float(g["diffusion_total"])) < 2e-6
    x = F.mld_losses(_rs(g), "diffusion", lam, predict_epsilon=False)
    assert abs(x["x_loss"] - float(g["diffusion_x_x_loss"])) < 2e-6 and abs(x["total"] - float(g["diffusion_x_total"])) < 2e-6


@pytest.mark.parametrize("stage,eps_pred", [("vae", True), ("diffusion", True), ("diffusion", False)])
def test_product_losses_match_reference_fixture(stage, eps_pred):
    """seeme_amd.mld.MLDLosses (the product's host code, pure torch) on the same tensors as the reference's MLDLosses."""
    from seeme_amd.config import Config
    from seeme_amd.mld import MLDLosses
    g = load_golden("losses.npz")
    lam = _lam(g)
    cfg = Config(LOSS=Config(**lam), TRAIN=Config(STAGE=stage, ABLATION=Config(PREDICT_EPSILON=eps_pred, PREDICT_TRANSL=True)))
    L = MLDLosses(cfg)
    t = {k: torch.from_numpy(g[k].copy()) for k in ("m_ref", "m_rst", "joints_ref", "joints_rst", "noise", "noise_pred", "latent", "pred")}
    t["dist_m"] = torch.distributions.Normal(torch.from_numpy(g["mu"]), torch.from_numpy(g["std"]))
    t["dist_ref"] = torch.distributions.Normal(torch.zeros(1, 3, 256), torch.ones(1, 3, 256))
    total = float(L.update(t))
    tag = stage + ("" if eps_pred else "_x")
    assert abs(total - float(g[f"{tag}_total"])) < 2e-6 * max(1.0, abs(float(g[f"{tag}_total"])))
    got = L.compute()
    names = {"vae": ("recons_feature", "recons_joints", "recons_transl", "kl_motion"),
             "diffusion": ("inst_loss",), "diffusion_x": ("x_loss",)}[tag]
    for k in names:
        assert abs(got[k] - float(g[f"{tag}_{k}"])) < 2e-6 * max(1.0, abs(float(g[f"{tag}_{k}"]))), k
    assert abs(got["total"] - float(g[f"{tag}_total"])) < 2e-6 * max(1.0, abs(float(g[f"{tag}_total"])))


def test_oracle_schedulers_match_egohmr_fixture():
    """DDIM (eta 0 and 0.5), DDPM ancestral step and add_noise of the oracle vs the reference tree's GaussianDiffusion."""
    g = load_golden("schedulers_egohmr.npz")
    acp = O.alphas_cumprod(O.make_betas())
    x, eps = g["x"], g["eps"]
    for t in g["ddim_t"]:
        for eta in (0.0, 0.5):
            got = O.ddim_step(acp, eps, int(t), x, 50, eta, g[f"ddim_noise_t{t}"])
            assert rel_err(got, g[f"ddim_t{t}_eta{eta}"]) < 2e-6, (int(t), eta)
    for t in g["ddpm_t"]:
        got = O.ddpm_step(acp, eps, int(t), x, g[f"ddpm_noise_t{t}"])
        # alpha-bar is a float32 cumprod here (as in diffusers) and float64 in GaussianDiffusion: 1 - alpha-bar[1] = 1.7e-3
        # carries that difference at the 1e-5 level
        assert rel_err(got, g[f"ddpm_t{t}"]) < 2e-5, int(t)
    assert rel_err(O.ddpm_add_noise(acp, x, eps, g["add_noise_t"]), g["add_noise"]) < 2e-6


def test_product_schedulers_match_egohmr_fixture():
    """seeme_amd.schedulers (the diffusers surface of the product, host side) and its coefficient table -- what the fused
    sampling kernel consumes -- vs the same fixture."""
    from seeme_amd.schedulers import DDIMScheduler, DDPMScheduler
    g = load_golden("schedulers_egohmr.npz")
    x, eps = torch.from_numpy(g["x"]), torch.from_numpy(g["eps"])
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False)
    ddim = DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **kw)
    ddim.set_timesteps(50)
    assert ddim.timesteps.tolist() == list(range(981, 0, -20))
    for eta in (0.0, 0.5):
        tab = ddim.coef_table(eta)
        for t in g["ddim_t"]:
            nz = torch.from_numpy(g[f"ddim_noise_t{t}"])
            got = ddim.step(eps, int(t), x, eta=eta, variance_noise=nz).prev_sample
            assert rel_err(got.numpy(), g[f"ddim_t{t}_eta{eta}"]) < 2e-6
            c = tab[ddim.timesteps.tolist().index(int(t))]          # the kernel's form of the same update
            x0 = (x - c[1] * eps) / c[0]
            assert rel_err((c[2] * x0 + c[3] * eps + c[5] * x + c[4] * nz).numpy(), g[f"ddim_t{t}_eta{eta}"]) < 2e-6
    ddpm = DDPMScheduler(variance_type="fixed_small", **kw)
    ddpm.set_timesteps(1000)
    tab = ddpm.coef_table()
    for t in g["ddpm_t"]:
        nz = torch.from_numpy(g[f"ddpm_noise_t{t}"])
        got = ddpm.step(eps, int(t), x, variance_noise=nz).prev_sample
        assert rel_err(got.numpy(), g[f"ddpm_t{t}"]) < 2e-5
        c = tab[999 - int(t)]
        x0 = (x - c[1] * eps) / c[0]
        assert rel_err((c[2] * x0 + c[3] * eps + c[5] * x + c[4] * nz).numpy(), g[f"ddpm_t{t}"]) < 2e-5
    got = ddpm.add_noise(x, eps, torch.from_numpy(g["add_noise_t"]))
    assert rel_err(got.numpy(), g["add_noise"]) < 2e-6


def test_flow_oracle_shapes_and_gimo_rules():
    """Small end-to-end run of the stage-1 / stage-2 / eval restatements (shapes, the GIMO [:21] + reference-orient rule,
    the classifier-free-guidance batch layout)."""
    from seeme_amd import shapes
    from seeme_amd.weights_recipe import recipe_state_dict
    rng = np.random.default_rng(3)
    B, T = 2, 6
    smpl = O.make_synthetic_smpl(7)
    f32 = lambda *s: rng.standard_normal(s).astype(np.float32)
    for dataset, pose, F_ in (("egobody", 72, 75), ("gimo", 66, 69)):
        Pv = recipe_state_dict(shapes.vae_shapes(F_))
        motion, transl, beta = 0.3 * f32(B, T, 2, pose), f32(B, 2, T, 3), 0.3 * f32(B, 2, T, 10)
        mean, std = 0.1 * f32(1, F_ + 4), (0.5 + rng.random((1, F_ + 4))).astype(np.float32)
        rs = F.train_vae_forward(Pv, smpl, motion, transl, beta, mean, std, f32(1, B, 256), dataset=dataset)
        nj = 21 if dataset == "gimo" else 24
        assert rs["joints_ref"].shape == rs["joints_rst"].shape == (B, T, nj, 3) and rs["m_rst"].shape == (B, T, F_)
        if dataset == "gimo":      # same global orientation on both sides: re-posing the reconstruction with its own differs
            own = F.feats_to_joints(smpl, rs["m_rst"], beta[:, 0], "gimo", True)[:, :, :21]
            assert np.abs(own - rs["joints_rst"]).max() > 1e-4
        v = F.mld_losses(rs, "vae", dict(LAMBDA_KL=1e-4, LAMBDA_REC=1.0, LAMBDA_JOINT=1.0, LAMBDA_ROOT=1.0))
        assert np.isfinite(v["total"]) and v["total"] > 0
    # stage 2, scene + interactee with classifier-free masks
    Pv, Pd = recipe_state_dict(shapes.vae_shapes(75)), recipe_state_dict(shapes.denoiser_shapes())
    Ppn = recipe_state_dict(shapes.pointnet_shapes())
    Pos = {"1.weight": 0.05 * f32(256, 512), "1.bias": 0.05 * f32(256)}
    motion, transl = 0.3 * f32(B, T, 2, 72), f32(B, 2, T, 3)
    scene = rng.uniform(-3, 3, (B, 40, 3)).astype(np.float32)
    out = F.train_diffusion_forward(Pv, Pd, motion, transl, condition=("text", "scene", "interactee"), eps_z=f32(1, B, 256),
                                    eps_c=f32(1, B, 256), noise=f32(B, 1, 256), timesteps=np.array([3, 900]), scene=scene,
                                    Ppn=Ppn, Pos=Pos, mask_scene=rng.random(scene.shape) < 0.1,
                                    mask_int=rng.random((B, T, 75)) < 0.1, guidance_scale=7.5)
    assert out["cond_emb"].shape == (2, B, 256) and out["noise_pred"].shape == (B, 1, 256)
