"""CPU oracle for the ORCHESTRATION of the SEE-ME hot path (numpy restatement of ``MLD``'s live flows).

TEST INFRASTRUCTURE ONLY (same rule as ``mld_oracle.py``: nothing under ``seeme_amd/`` imports this).

``mld.models.modeltype.mld`` cannot be imported in the build container (it needs smplx, torchmetrics,
pytorch_lightning, yacs, omegaconf -- SURVEY.md section 8c), so these functions restate its flows from the source
text, line by line, on top of the building blocks of ``mld_oracle.py`` (which ARE pinned by reference-generated
fixtures).  Every random draw of the reference (``rsample`` noise, classifier-free-guidance masks, training noise and
timesteps, initial latents) is an explicit argument, in the order the reference draws them.

Parity status: the arithmetic inside (VAE, denoiser, PointNet, geometry) is pinned through ``mld_oracle``; the losses
use only ``SmoothL1`` / ``MSE`` / ``KL(Normal||Normal)`` whose closed forms are restated here and cross-checked against
``torch.nn.functional`` in ``tests/test_oracle_flows.py``; the flow logic itself (which tensor goes where) is a
restatement with no executable reference to compare against: **parity unpinned** for the orchestration, anchored on
the cited lines.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from . import mld_oracle as O

Array = np.ndarray


# ----------------------------------------------------------------------------- losses (mld/models/losses/mld.py)
def smooth_l1(a: Array, b: Array, beta: float = 1.0) -> float:
    """torch.nn.SmoothL1Loss(reduction='mean'), the 'recons' / 'gen' / 'latent' / 'transl' losses (losses/mld.py:80-97)."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return float(np.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).mean())


def mse(a: Array, b: Array) -> float:
    """nn.MSELoss(reduction='mean'): 'inst', 'x', 'prior' (losses/mld.py:68-76)."""
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float((d * d).mean())


def kl_normal(mu_q: Array, std_q: Array, mu_p: Array, std_p: Array) -> float:
    """KLLoss: torch.distributions.kl_divergence(Normal q, Normal p).mean() (losses/mld.py:178-188)."""
    mu_q, std_q, mu_p, std_p = (np.asarray(x, np.float64) for x in (mu_q, std_q, mu_p, std_p))
    var_ratio = (std_q / std_p) ** 2
    t1 = ((mu_q - mu_p) / std_p) ** 2
    return float((0.5 * (var_ratio + t1 - 1.0 - np.log(var_ratio))).mean())


def loss_weights(lam: Dict[str, float]) -> Dict[str, float]:
    """The weight table built by MLDLosses.__init__ (losses/mld.py:66-102): an if / elif chain on the loss NAME, where
    ``split('_')[0] == 'recons'`` is tested before ``split('_')[-1] == 'transl'`` -- so ``recons_transl`` is weighted by
    LAMBDA_REC and the LAMBDA_ROOT branch is never reached -- and names ending in 'joints' are overridden last."""
    names = ["inst_loss", "x_loss", "prior_loss", "recons_feature", "recons_verts", "recons_joints", "recons_limb",
             "recons_transl", "gen_feature", "gen_joints", "kl_motion"]
    w = {}
    for loss in names:
        head, tail = loss.split("_")[0], loss.split("_")[-1]
        if head == "inst":
            w[loss] = 1.0
        elif head == "x":
            w[loss] = 1.0
        elif head == "prior":
            w[loss] = lam.get("LAMBDA_PRIOR", 0.0)
        if head == "kl":
            if lam.get("LAMBDA_KL", 0.0) != 0.0:
                w[loss] = lam["LAMBDA_KL"]
        elif head == "recons":
            w[loss] = lam["LAMBDA_REC"]
        elif head == "gen":
            w[loss] = lam.get("LAMBDA_GEN", 1.0)
        elif head == "latent":
            w[loss] = lam.get("LAMBDA_LATENT", 1.0)
        elif tail == "transl":
            w[loss] = lam["LAMBDA_ROOT"]
        if tail == "joints":
            w[loss] = lam.get("LAMBDA_JOINT", 1.0)
    return w


def mld_losses(rs: dict, stage: str, lam: Dict[str, float], predict_transl: bool = True,
               predict_epsilon: bool = True) -> Dict[str, float]:
    """MLDLosses.update (losses/mld.py:113-156): returns every unweighted term and the weighted 'total'.
    rs holds numpy arrays; the posterior is given as dist_m = (mu, std), dist_ref = (mu, std)."""
    w = loss_weights(lam)
    out: Dict[str, float] = {}
    total = 0.0
    if stage in ("vae", "vae_diffusion"):
        jr, jp = rs["joints_ref"], rs["joints_rst"]
        if predict_transl:                                      # align_root, :104-111 (the dict entries are replaced)
            pg, pp = jr[:, :, [0]], jp[:, :, [0]]
            jr, jp = jr - pg, jp - pp
            out["recons_transl"] = smooth_l1(pp, pg)
            total += w["recons_transl"] * out["recons_transl"]
        out["recons_feature"] = smooth_l1(rs["m_rst"], rs["m_ref"])
        total += w["recons_feature"] * out["recons_feature"]
        out["recons_joints"] = smooth_l1(jp, jr)
        total += w["recons_joints"] * out["recons_joints"]
        if "kl_motion" in w:                                    # (the reference raises KeyError when LAMBDA_KL == 0)
            out["kl_motion"] = kl_normal(*rs["dist_m"], *rs["dist_ref"])
            total += w["kl_motion"] * out["kl_motion"]
    if stage in ("diffusion", "vae_diffusion"):
        if predict_epsilon:
            out["inst_loss"] = mse(rs["noise_pred"], rs["noise"])
            total += w["inst_loss"] * out["inst_loss"]
        else:
            out["x_loss"] = mse(rs["pred"], rs["latent"])
            total += w["x_loss"] * out["x_loss"]
    out["total"] = total
    return out


# ----------------------------------------------------------------------------- shared pieces
def person_features(motion: Array, transl: Array, idx: int, predict_transl: bool) -> Array:
    """f_ref = cat([feats_ref[:, :, idx, :], transl[:, idx, :, :]], -1)  (mld.py:657-661, 944-948, 1273-1275)."""
    f = motion[:, :, idx, :]
    return np.concatenate([f, transl[:, idx]], axis=-1) if predict_transl else f


def scene_token(Ppn: dict, Pos: dict, scene: Array) -> Array:
    """proscene.encode_scene -> output_scene = ReLU + Linear(512, 256) -> [1,B,256]  (mld.py:257-261, 921-922)."""
    s512 = O.pointnet_forward(Ppn, scene)
    return O.linear(O.relu(s512), Pos["1.weight"], Pos["1.bias"])[None]


def feats_to_joints(smpl: dict, feats: Array, betas: Array, dataset: str, predict_transl: bool,
                    orient: Optional[Array] = None) -> Array:
    """The 'angle' SMPL call of train_vae_forward / ego_eval (mld.py:757-799 egobody, :801-860 gimo): body pose columns
    3:72 (egobody) or 3:66 padded with 6 zeros (gimo), global orientation columns 0:3 (or `orient`), translation = the
    last three columns.  Returns joints [B,T,24,3] (the callers slice [:21] where the reference does)."""
    B, T, _ = feats.shape
    nb = 69 if dataset == "egobody" else 63
    body = feats[:, :, 3:3 + nb].reshape(-1, nb)
    if nb < 69:
        body = np.concatenate([body, np.zeros((body.shape[0], 69 - nb), body.dtype)], axis=1)
    go = (feats[:, :, :3] if orient is None else orient).reshape(-1, 3)
    tr = feats[:, :, -3:].reshape(-1, 3) if predict_transl else None
    j, _ = O.smpl_lbs(smpl, betas.reshape(-1, 10).astype(feats.dtype), go, body, tr, return_verts=False)
    return j.reshape(B, T, 45, 3)[:, :, :24]


def feats_to_joints_rot6d(smpl: dict, feats: Array) -> Array:
    """DATA_TYPE 'rot6d' (mld.py:699-735, 1410-1449): 24 x 6 -> rotation matrices ('prohmr' column order,
    geometry2.py:98-117), body model built in float64 (mld.py:161-163), pose2rot=False, no betas, no translation."""
    B, T, _ = feats.shape
    R = O.rot6d_to_rotmat(feats[..., :144].reshape(-1, 6)).reshape(B * T, 24, 3, 3).astype(np.float64)
    m64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for k, v in smpl.items()}
    j, _ = O.smpl_lbs(m64, np.zeros((B * T, 10), np.float64), R[:, 0:1], R[:, 1:], None, pose2rot=False, return_verts=False)
    return j.reshape(B, T, 45, 3)[:, :, :24]


# ----------------------------------------------------------------------------- stage 1 (mld.py:633-885)
def train_vae_forward(Pv: dict, smpl: dict, motion: Array, transl: Array, beta: Array, mean: Array, std: Array, eps: Array,
                      *, estimate: str = "wearer", dataset: str = "egobody", predict_transl: bool = True,
                      is_vae: bool = True) -> dict:
    idx = 0 if estimate == "wearer" else 1
    B, T = motion.shape[:2]
    lengths = [T] * B                                                        # :652
    f_ref = person_features(motion, transl, idx, predict_transl)
    mu, sd = O.vae_encode(Pv, f_ref, lengths)                                # :656-675
    z = mu + eps * sd                                                        # rsample, mld_vae.py:191-192
    feats_rst = O.vae_decode(Pv, z, lengths)                                 # :676
    m_ref = O.renorm(f_ref, mean, std)                                       # :757
    m_rst = O.renorm(feats_rst, mean, std)                                   # :778 / :830
    b = beta[:, idx]
    nj = 21 if dataset == "gimo" else 24                                     # :826-828, 857-859
    joints_ref = feats_to_joints(smpl, m_ref, b, dataset, predict_transl)[:, :, :nj]
    # gimo: the reconstruction is posed with the REFERENCE's global orientation (:853)
    joints_rst = feats_to_joints(smpl, m_rst, b, dataset, predict_transl,
                                 orient=m_ref[:, :, :3] if dataset == "gimo" else None)[:, :, :nj]
    dist_ref = (np.zeros_like(mu), np.ones_like(sd)) if is_vae else (mu, sd)  # :683-691
    return {"m_ref": m_ref, "m_rst": m_rst, "joints_ref": joints_ref, "joints_rst": joints_rst,
            "dist_m": (mu, sd), "dist_ref": dist_ref, "z": z}


# ----------------------------------------------------------------------------- stage 2 (mld.py:887-1017, 582-631)
def train_diffusion_forward(Pv: dict, Pd: dict, motion: Array, transl: Array, *, condition: Sequence[str],
                            eps_z: Array, noise: Array, timesteps: Array, eps_c: Optional[Array] = None,
                            scene: Optional[Array] = None, Ppn: Optional[dict] = None, Pos: Optional[dict] = None,
                            mask_scene: Optional[Array] = None, mask_int: Optional[Array] = None,
                            estimate: str = "wearer", predict_transl: bool = True, guidance_scale: float = 1.0,
                            nhead: int = 1) -> dict:
    """Draw order of the reference: scene mask (:917-919), target rsample (:944-948), interactee mask (:966-968),
    condition rsample (:969-971), training noise and timesteps (:591-601).  mask_* are the boolean results of
    ``torch.rand_like(x) < guidance_uncondp``; they are only applied when guidance_scale > 1 (do_classifier_free_guidance)."""
    cfg = guidance_scale > 1.0
    B, T = motion.shape[:2]
    lengths = [T] * B
    tok_scene = None
    if "scene" in condition:
        s = scene.astype(np.float32)
        if cfg:
            s = np.where(mask_scene, np.float32(0.0), s)
        tok_scene = scene_token(Ppn, Pos, s)                                 # [1,B,256]
    idx = 0 if estimate == "wearer" else 1
    mu, sd = O.vae_encode(Pv, person_features(motion, transl, idx, predict_transl), lengths)
    z = mu + eps_z * sd
    z_cond = None
    if "interactee" in condition:
        f_int = person_features(motion, transl, 1, predict_transl)
        if cfg:
            f_int = np.where(mask_int, np.float32(0.0), f_int)
        mu_c, sd_c = O.vae_encode(Pv, f_int, lengths)
        z_cond = mu_c + eps_c * sd_c
    toks = [t for t in (z_cond, tok_scene) if t is not None]                 # :991-1013: [z_cond, scene] in this order
    cond_emb = np.concatenate(toks, axis=0)                                  # [N,B,256] seq-first
    latents = np.transpose(z, (1, 0, 2))                                     # :588
    acp = O.alphas_cumprod(O.make_betas())
    noisy = O.ddpm_add_noise(acp, latents, noise, np.asarray(timesteps))     # :604-606
    noise_pred = O.denoiser_forward(Pd, noisy, np.asarray(timesteps), cond_emb, nhead=nhead)   # :608-613
    return {"noise": noise, "noise_pred": noise_pred, "z": z, "cond_emb": cond_emb, "noisy": noisy}


# ----------------------------------------------------------------------------- evaluation (mld.py:1076-1905)
def ego_eval(Pv: dict, Pd: Optional[dict], smpl: dict, motion: Array, transl: Array, beta: Array, lengths: Sequence[int],
             mean: Array, std: Array, *, stage: str = "diffusion", condition: Sequence[str] = ("text", "interactee"),
             latents: Optional[Array] = None, eps_c: Optional[Array] = None, eps_u: Optional[Array] = None,
             scene: Optional[Array] = None, Ppn: Optional[dict] = None, Pos: Optional[dict] = None,
             estimate: str = "wearer", dataset: str = "egobody", data_type: str = "angle", predict_transl: bool = True,
             guidance_scale: float = 1.0, steps: int = 50, pred_global_orient: bool = True, see_future: bool = False,
             nhead: int = 1) -> dict:
    """The live part of ego_eval.  Classifier-free guidance reproduces the reference's batch layout: interactee tokens
    [uncond, cond] (:1283-1290), scene tokens [cond, uncond] (:1144-1158), first half taken as unconditional (:489)."""
    cfg = guidance_scale > 1.0
    lengths = [int(v) for v in lengths]
    idx = 0 if estimate == "wearer" else 1
    if stage in ("diffusion", "vae_diffusion"):
        tok_scene = None
        if "scene" in condition:
            tok_scene = scene_token(Ppn, Pos, scene.astype(np.float32))
            if cfg:
                unc = scene_token(Ppn, Pos, np.zeros_like(scene, dtype=np.float32))
                tok_scene = np.concatenate([tok_scene, unc], axis=1)
        text = None
        if "interactee" in condition:
            f_int = person_features(motion, transl, 1, predict_transl)
            mu, sd = O.vae_encode(Pv, f_int, lengths)
            text = mu + eps_c * sd                                           # vae.encode(...)[0]: a sample, :1280
            if cfg:
                mu_u, sd_u = O.vae_encode(Pv, np.zeros_like(f_int), lengths)
                text = np.concatenate([mu_u + eps_u * sd_u, text], axis=1)
        cond_emb = np.concatenate([t for t in (text, tok_scene) if t is not None], axis=0)
        z = O.diffusion_reverse(Pd, np.transpose(cond_emb, (1, 0, 2)), latents, steps, guidance_scale=guidance_scale,
                                nhead=nhead)
    elif stage == "vae":                                                     # :1328-1352
        mu, sd = O.vae_encode(Pv, person_features(motion, transl, idx, predict_transl), lengths)
        z = mu + eps_c * sd
    else:
        raise ValueError(stage)
    if see_future:
        lengths = [int(v // 2) for v in lengths]                             # :1357-1358
    feats_rst = O.vae_decode(Pv, z, lengths)
    min_len = min(motion.shape[1], feats_rst.shape[1])
    f_ref = person_features(motion[:, :min_len], transl[:, :, :min_len], idx, predict_transl)
    m_ref = O.renorm(f_ref, mean, std)
    m_rst = O.renorm(feats_rst[:, :min_len], mean, std)
    b = beta[:, idx, :min_len]
    if data_type == "rot6d":
        joints_ref, joints_rst = feats_to_joints_rot6d(smpl, m_ref), feats_to_joints_rot6d(smpl, m_rst)
        q_ref = q_rst = None
    else:
        o_rst = m_ref[:, :, :3] if (dataset == "egobody" and not pred_global_orient) else None      # :1497-1501
        joints_ref = feats_to_joints(smpl, m_ref, b, dataset, predict_transl)
        joints_rst = feats_to_joints(smpl, m_rst, b, dataset, predict_transl, orient=o_rst)
        q_ref = O.aa_to_quat(m_ref[:, :, :3].reshape(-1, 3))
        q_rst = O.aa_to_quat((m_rst[:, :, :3] if o_rst is None else o_rst).reshape(-1, 3))
    return {"m_ref": m_ref, "m_rst": m_rst, "joints_ref": joints_ref, "joints_rst": joints_rst, "lat_t": z,
            "orientation_quat_ref": q_ref, "orientation_quat_rst": q_rst, "lengths": lengths}
