#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE's own PyTorch modules.

Runs only in the build container (it imports /root/reference, which never travels to the GPU
box).  Weights come from ``seeme_amd.weights_recipe`` (name+shape -> values), inputs from a seeded
numpy PCG64; outputs are whatever the reference modules compute on CPU in fp32, eval mode.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

What is pinned by the reference:  MldVae.encode/decode, MldDenoiser.forward, Timesteps,
ResnetPointnet, geometry2 helpers, MLDLosses (losses.npz) and -- through the OpenAI-style
GaussianDiffusion the tree vendors under EgoHMR/ -- the DDIM / DDPM update formulas for the alpha-bar
sequence MLD's scheduler configuration visits (schedulers_egohmr.npz).  The 50-step loop fixture drives
the reference denoiser with the build's restated DDIM step (diffusers itself is absent).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SEEME_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.modules.setdefault("clip", types.ModuleType("clip"))  # mdiff_transformer.py:10 imports it, never calls it

from seeme_amd.weights_recipe import load_recipe_  # noqa: E402
from oracle import mld_oracle as O  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)
SEED = 1234  # configs/base.yaml:2


def ablation():
    # configs/base.yaml:18-27 (+) config_mld_egobody.yaml:45-50
    return types.SimpleNamespace(MLP_DIST=False, PE_TYPE="mld", SKIP_CONNECT=True, VAE_TYPE="actor",
                                 DIFF_PE_TYPE="mld", MD_TRANS=True)


def rng(tag):
    return np.random.Generator(np.random.PCG64(abs(hash_str(tag)) + SEED))


def hash_str(s):
    import zlib
    return zlib.crc32(s.encode())


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB", {k: v.shape for k, v in arrs.items()})


def make_vae(F):
    from mld.models.architectures.mld_vae import MldVae
    vae = MldVae(ablation(), nfeats=F, latent_dim=[1, 256], ff_size=1024, num_layers=9, num_heads=4,
                 dropout=0.1, arch="encoder_decoder", normalize_before=False, activation="gelu",
                 position_embedding="learned").eval()
    return load_recipe_(vae, SEED)


def make_denoiser(cond):
    from mld.models.architectures.mld_denoiser import MldDenoiser
    den = MldDenoiser(ablation(), nfeats=75, condition=cond, latent_dim=[1, 256], ff_size=128,
                      num_layers=5, num_heads=1, dropout=0.1, normalize_before=False, activation="gelu",
                      flip_sin_to_cos=True, return_intermediate_dec=False, position_embedding="learned",
                      arch="trans_enc", freq_shift=0, guidance_scale=1.0, guidance_uncondp=0.1,
                      text_encoded_dim=256, nclasses=10).eval()
    return load_recipe_(den, SEED)


def vae_case(name, F, T, lengths, with_layers=False):
    vae = make_vae(F)
    B = len(lengths)
    g = rng(name)
    x = g.standard_normal((B, T, F)).astype(np.float32)
    out = {}
    if with_layers:
        hooks = []

        def mk(tag):
            def hook(_m, _i, o):
                out[tag] = o.detach().numpy().copy()  # seq-first [S,B,D]
            return hook
        for i, blk in enumerate(list(vae.encoder.input_blocks) + [vae.encoder.middle_block]
                                + list(vae.encoder.output_blocks)):
            hooks.append(blk.register_forward_hook(mk(f"enc_block{i}")))
        for i, blk in enumerate(list(vae.decoder.input_blocks) + [vae.decoder.middle_block]
                                + list(vae.decoder.output_blocks)):
            hooks.append(blk.register_forward_hook(mk(f"dec_block{i}")))
    _, dist = vae.encode(torch.from_numpy(x), None, list(lengths))
    mu, std = dist.loc, dist.scale
    feats = vae.decode(mu, list(lengths))
    save(name, features=x, lengths=np.asarray(lengths, np.int64), mu=mu.numpy(), std=std.numpy(),
         decoded=feats.numpy(), **out)


def denoiser_case(name, N, B=4):
    cond_names = ["text", "interactee"] if N == 1 else ["text", "scene", "interactee"]
    den = make_denoiser(cond_names)
    g = rng(name)
    sample = g.standard_normal((B, 1, 256)).astype(np.float32)
    cond = g.standard_normal((N, B, 256)).astype(np.float32)  # seq-first, as MldDenoiser takes it
    arrs = dict(sample=sample, cond=cond)
    for t in (981, 501, 1):
        y = den(sample=torch.from_numpy(sample), timestep=torch.tensor(t),
                encoder_hidden_states=torch.from_numpy(cond), lengths=None)[0]
        arrs[f"out_t{t}"] = y.numpy()
    tv = np.array([999, 0, 37, 512][:B], np.int64)
    y = den(sample=torch.from_numpy(sample), timestep=torch.from_numpy(tv),
            encoder_hidden_states=torch.from_numpy(cond), lengths=None)[0]
    arrs["tvec"] = tv
    arrs["out_tvec"] = y.numpy()
    save(name, **arrs)


def loop_case(name, N, B, steps, cfg_scale=1.0):
    """Reference denoiser module inside the restated DDIM loop (mld.py:467-497)."""
    cond_names = ["text", "interactee"] if N == 1 else ["text", "scene", "interactee"]
    den = make_denoiser(cond_names)
    g = rng(name)
    lat = g.standard_normal((B, 1, 256)).astype(np.float32)
    nb = 2 * B if cfg_scale > 1.0 else B
    cond_bf = g.standard_normal((nb, N, 256)).astype(np.float32)  # batch-first, as _diffusion_reverse takes it
    if cfg_scale > 1.0:
        cond_bf[:B] = 0.0  # uncond branch first (mld.py:489, 1283-1290)
    acp = O.alphas_cumprod(O.make_betas())
    x = lat.copy()
    cond_sf = torch.from_numpy(np.ascontiguousarray(cond_bf.transpose(1, 0, 2)))
    for t in O.ddim_timesteps(steps):
        xin = torch.from_numpy(np.concatenate([x, x]) if cfg_scale > 1.0 else x)
        eps = den(sample=xin, timestep=torch.tensor(int(t)), encoder_hidden_states=cond_sf, lengths=None)[0].numpy()
        if cfg_scale > 1.0:
            eu, ec = np.split(eps, 2)
            eps = eu + np.float32(cfg_scale) * (ec - eu)
        x = O.ddim_step(acp, eps, int(t), x, steps, 0.0)
    save(name, latents=lat, cond_bf=cond_bf, steps=np.int64(steps), guidance_scale=np.float32(cfg_scale),
         out=np.ascontiguousarray(x.transpose(1, 0, 2)))


def misc_case():
    from mld.models.architectures.tools.embeddings import Timesteps
    from mld.utils import geometry2 as G
    from EgoHMR.models.respointnet import ResnetPointnet
    g = rng("misc")
    t = np.array([0, 1, 21, 501, 981, 999], np.int64)
    tf = Timesteps(256, True, 0)(torch.from_numpy(t)).numpy()
    aa = (g.standard_normal((16, 3)) * 1.5).astype(np.float32)
    aa[0] = 0.0
    r6 = g.standard_normal((16, 6)).astype(np.float32)
    pn = load_recipe_(ResnetPointnet(512, 256).eval(), SEED)
    # the reference zero-inits fc_1.weight (respointnet.py:86); the recipe makes it non-zero on purpose
    pts = g.uniform(-3, 3, (2, 1024, 3)).astype(np.float32)
    save("misc.npz", t=t, timestep_features=tf, aa=aa,
         aa_to_quat=G.aa_to_quat(torch.from_numpy(aa)).numpy(),
         aa_to_rotmat=G.aa_to_rotmat(torch.from_numpy(aa)).numpy(),
         rot6d=r6, rot6d_to_rotmat=G.rot6d_to_rotmat(torch.from_numpy(r6)).numpy(),
         points=pts, pointnet=pn(torch.from_numpy(pts)).numpy())


def losses_case():
    """The reference's own MLDLosses (mld/models/losses/mld.py:10-188) on fixed tensors with NON-UNIT lambdas.  Its base
    class torchmetrics.Metric is not installed; the six-line stand-in below only provides ``add_state`` (a named
    attribute), every weight, loss function and the whole ``update`` arithmetic are the reference's."""
    tm = types.ModuleType("torchmetrics")

    class Metric(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone() if torch.is_tensor(default) else default)

    tm.Metric = Metric
    sys.modules.setdefault("torchmetrics", tm)
    from mld.models.losses.mld import MLDLosses
    lam = dict(LAMBDA_LATENT=1e-5, LAMBDA_KL=3e-3, LAMBDA_REC=0.7, LAMBDA_JOINT=1.9, LAMBDA_GEN=1.0, LAMBDA_CROSS=1.0,
               LAMBDA_CYCLE=0.0, LAMBDA_PRIOR=0.0, LAMBDA_ROOT=0.31, DIST_SYNC_ON_STEP=False)
    g = rng("losses")
    B, T, F = 3, 7, 75
    arr = lambda *shape, scale=1.0: (g.standard_normal(shape) * scale).astype(np.float32)
    rs = dict(m_ref=arr(B, T, F), m_rst=arr(B, T, F, scale=1.7), joints_ref=arr(B, T, 24, 3), joints_rst=arr(B, T, 24, 3, scale=2.0),
              mu=arr(1, B, 256, scale=0.5), std=np.exp(arr(1, B, 256, scale=0.3)).astype(np.float32),
              noise=arr(B, 1, 256), noise_pred=arr(B, 1, 256, scale=1.2), latent=arr(B, 1, 256), pred=arr(B, 1, 256))
    out = {k: v for k, v in rs.items()}
    out["lambdas"] = np.array([lam[k] for k in ("LAMBDA_KL", "LAMBDA_REC", "LAMBDA_JOINT", "LAMBDA_ROOT")], np.float64)
    for stage, eps_pred in (("vae", True), ("diffusion", True), ("diffusion", False)):
        cfg = types.SimpleNamespace(
            LOSS=types.SimpleNamespace(**lam),
            TRAIN=types.SimpleNamespace(STAGE=stage, ABLATION=types.SimpleNamespace(VAE_TYPE="actor", PREDICT_EPSILON=eps_pred,
                                                                                 PREDICT_TRANSL=True)))
        L = MLDLosses(vae=True, mode="ego", cfg=cfg)
        t = {k: torch.from_numpy(v.copy()) for k, v in rs.items()}
        t["dist_m"] = torch.distributions.Normal(t["mu"], t["std"])
        t["dist_ref"] = torch.distributions.Normal(torch.zeros_like(t["mu"]), torch.ones_like(t["std"]))
        total = L.update(t)
        tag = stage + ("" if eps_pred else "_x")
        out[f"{tag}_total"] = np.float64(total.item())
        for name in L.losses:
            out[f"{tag}_{name}"] = np.float64(float(getattr(L, name)))
        if stage == "vae":       # align_root replaced the dict entries (losses/mld.py:119-121)
            out["aligned_joints_ref"], out["aligned_joints_rst"] = t["joints_ref"].numpy(), t["joints_rst"].numpy()
    save("losses.npz", **out)


def scheduler_case():
    """Scheduler arithmetic the reference tree DOES hold: EgoHMR/diffusion/gaussian_diffusion.py (OpenAI formulation):
    ``ddim_sample`` (:511-557), ``p_sample`` (:298-338) and ``q_sample`` (:189-207).  ``diffusers`` itself is absent
    (SURVEY section 8c), so this is the one executable cross-check of the DDIM / DDPM update formulas: a
    GaussianDiffusion is built on the alpha-bar sequence the MLD scheduler configuration visits
    (configs/modules/scheduler.yaml:1-14: scaled_linear 0.00085..0.012, 50 steps, steps_offset 1, set_alpha_to_one
    False => alpha-bar_prev of the last step is alpha-bar[0]), and its outputs for given (x_t, eps, noise) are stored."""
    sys.path.insert(0, os.path.join(REF, "EgoHMR"))
    from diffusion.gaussian_diffusion import GaussianDiffusion
    acp = np.cumprod(1.0 - (np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2))
    ts = (np.arange(50) * 20 + 1).astype(np.int64)                    # ascending DDIM timesteps 1, 21, ..., 981
    abar = np.concatenate([[acp[0]], acp[ts]])                         # index 0 = the "previous" of t = 1
    betas = 1.0 - abar / np.concatenate([[1.0], abar[:-1]])
    gd = GaussianDiffusion(betas=betas)
    assert np.allclose(gd.alphas_cumprod, abar)
    g = rng("sched")
    B = 4
    x = g.standard_normal((B, 1, 256)).astype(np.float32)
    eps = g.standard_normal((B, 1, 256)).astype(np.float32)

    class EpsModel:                      # the EgoHMR model interface returns pred_x_start; ours predicts epsilon
        def __init__(self, d, eps):
            self.d, self.eps = d, torch.from_numpy(eps)

        def __call__(self, batch, t):
            return {"pred_x_start": self.d._predict_xstart_from_eps(batch["x_t"], t, self.eps)}

    out = dict(x=x, eps=eps, ddim_t=np.array([981, 501, 21, 1], np.int64))
    for t in out["ddim_t"]:
        idx = torch.full((B,), int((t - 1) // 20 + 1), dtype=torch.long)
        for eta in (0.0, 0.5):
            torch.manual_seed(100 + int(t))
            noise = torch.randn(B, 1, 256)                             # ddim_sample draws th.randn_like(x) once
            torch.manual_seed(100 + int(t))
            r = gd.ddim_sample(EpsModel(gd, eps), {}, torch.from_numpy(x), idx, clip_denoised=False, eta=eta)
            out[f"ddim_t{t}_eta{eta}"] = r["sample"].numpy()
            out[f"ddim_noise_t{t}"] = noise.numpy()
    # DDPM ancestral step and forward noising on the full 1000-step schedule (modules_novae/scheduler.yaml:16-26: fixed_small)
    gd2 = GaussianDiffusion(betas=1.0 - acp / np.concatenate([[1.0], acp[:-1]]))
    out["ddpm_t"] = np.array([999, 500, 1, 0], np.int64)
    for t in out["ddpm_t"]:
        idx = torch.full((B,), int(t), dtype=torch.long)
        torch.manual_seed(200 + int(t))
        noise = torch.randn(B, 1, 256)
        torch.manual_seed(200 + int(t))
        r = gd2.p_sample(EpsModel(gd2, eps), {}, torch.from_numpy(x), idx, clip_denoised=False)
        out[f"ddpm_t{t}"] = r["sample"].numpy()
        out[f"ddpm_noise_t{t}"] = noise.numpy()
    tv = np.array([999, 0, 37, 512], np.int64)
    out["add_noise_t"] = tv
    out["add_noise"] = gd2.q_sample(torch.from_numpy(x), torch.from_numpy(tv), noise=torch.from_numpy(eps)).numpy()
    save("schedulers_egohmr.npz", **out)


if __name__ == "__main__":
    vae_case("vae_F132_T24.npz", 132, 24, [24, 17, 9], with_layers=True)
    vae_case("vae_F75_T60.npz", 75, 60, [60, 60])
    vae_case("vae_F132_T196.npz", 132, 196, [196, 150])
    denoiser_case("denoiser_N1.npz", 1)
    denoiser_case("denoiser_N2.npz", 2)
    loop_case("ddim50_N1_B3.npz", 1, 3, 50)
    loop_case("ddim10_N2_B2_cfg.npz", 2, 2, 10, cfg_scale=7.5)
    misc_case()
    losses_case()
    scheduler_case()
