"""MLD -- the orchestration class of the path, with the method surface of
``mld.models.modeltype.mld.MLD`` / ``base.BaseModel`` (reference mld.py:88-2130, base.py:38-213) but
without Lightning: ``training_step / validation_step / test_step / allsplit_step``,
``_diffusion_reverse``, ``_diffusion_process``, ``train_diffusion_forward``, ``train_vae_forward``,
``ego_eval``, ``forward`` / ``sample``, ``configure_optimizers``.

What runs where
  * VAE encode/decode, the whole reverse-diffusion loop, the denoiser forward, PointNet, SMPL, rotation
    helpers and renorm run in libseeme_hip.so (HIP kernels for gfx950).
  * Stage-2 *training*: the frozen encoders and the denoiser forward + hand-written backward
    (``denoiser_train.py``: ``k_den_sample`` with saves, ``k_den_bwd``, ``seeme_den_wgrad``) are HIP; the table builders
    around the chain (rsample, add_noise, time MLP, output_scene, condition / time tables) are hand-written too
    (``stage2_glue.py``: one autograd node for the whole step); every gradient lives in one flat buffer
    (``distributed.GradBucket``) that is all-reduced in place and consumed by the one-launch AdamW.  More than one
    attention head falls back to the autograd twin (``denoiser_autograd.py``).
  * Stage-1 (VAE) *training*: hand-written HIP forward-with-saves / backward of the encoder and the decoder
    (``vae_train.py``: grouped fp32 GEMMs + LayerNorm / softmax / GELU / dropout kernels) and of the SMPL joint regressor
    (``smpl._JointsAA``); the differentiable twins (``vae_autograd.py``) are the fallback; evaluation of that stage is HIP.
  * Only the live flows are implemented; the reference's dead code (``forward`` calling the undefined
    ``feats2joints``, t2m_eval, the ``save_for_edo`` debug dump -- SURVEY.md App. D) is not reproduced:
    ``forward``/``sample`` = what ``ego_eval`` really does (condition -> reverse diffusion -> decode).
  * Deliberate, documented deviations from the reference: none by default.  ``TEST.SAMPLE_MEAN``
    (condition on the posterior mean instead of a sample) and ``TEST.CFG_SCENE_ORDER: fixed`` (classifier-free guidance
    with the unconditional scene token in the unconditional half) are opt-in.
"""
from __future__ import annotations

import inspect
import time
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import geometry as G
from .config import instantiate_from_config
from .denoiser_autograd import denoiser_forward_torch
from .respointnet import ResnetPointnet
from .smpl import SMPL


# ----------------------------------------------------------------------------- losses / metrics
_WARNED = set()


def _warn_once(key: str, msg: str) -> None:
    """Fallbacks onto the autograd twins are announced once per process (they used to be silent)."""
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class MLDLosses:
    """mld/models/losses/mld.py:10-176 without torchmetrics: running sums + the weighted total."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.stage = cfg.TRAIN.STAGE
        self.predict_epsilon = cfg.TRAIN.ABLATION.PREDICT_EPSILON
        self.predict_transl = cfg.TRAIN.ABLATION.PREDICT_TRANSL
        if self.stage not in ("vae", "diffusion", "vae_diffusion"):
            raise ValueError(f"Stage {self.stage} not supported")
        self.sums: Dict[str, float] = {}
        self.count = 0
        self.accumulate = True

    def _acc(self, name, val, weight):
        if self.accumulate:    # running sums stay on the device: no host sync per step (compute() converts)
            self.sums[name] = self.sums.get(name, 0.0) + val.detach()
        return weight * val

    @staticmethod
    def align_root(gt, pred):
        pg, pp = gt[:, :, [0]], pred[:, :, [0]]
        return gt - pg, pred - pp, pg, pp

    def update(self, rs, accumulate: bool = True):
        self.accumulate = accumulate
        L = self.cfg.LOSS
        total = 0.0
        mse = nn.functional.mse_loss
        sl1 = nn.functional.smooth_l1_loss
        if self.stage in ("vae", "vae_diffusion"):
            jr, jp = rs["joints_ref"], rs["joints_rst"]
            if self.predict_transl:
                jr, jp, pg, pp = self.align_root(jr, jp)
                # weight: the reference's elif chain tests split('_')[0] == 'recons' before 'transl' (losses/mld.py:80-97),
                # so recons_transl is weighted by LAMBDA_REC and its LAMBDA_ROOT branch is unreachable
                total = total + self._acc("recons_transl", sl1(pp, pg), L.LAMBDA_REC)
            total = total + self._acc("recons_feature", sl1(rs["m_rst"], rs["m_ref"]), L.LAMBDA_REC)
            total = total + self._acc("recons_joints", sl1(jp, jr), L.get("LAMBDA_JOINT", 1.0))
            if L.LAMBDA_KL != 0.0:
                kl = torch.distributions.kl_divergence(rs["dist_m"], rs["dist_ref"]).mean()
                total = total + self._acc("kl_motion", kl, L.LAMBDA_KL)
        if self.stage in ("diffusion", "vae_diffusion"):
            if self.predict_epsilon:
                total = total + self._acc("inst_loss", mse(rs["noise_pred"], rs["noise"]), 1.0)
            else:
                total = total + self._acc("x_loss", mse(rs["pred"], rs["latent"]), 1.0)
        if accumulate:
            self.sums["total"] = self.sums.get("total", 0.0) + total.detach()
            self.count += 1
        return total

    def compute(self):
        return {k: float(v) / max(self.count, 1) for k, v in self.sums.items()}

    def reset(self):
        self.sums, self.count = {}, 0


class EgoMetrics:
    """ComputeMetrics of the reference (metrics/compute.py:349-580,184-232) on the device, without its per-sequence
    `.cpu().numpy()` loops: MPJPE and root error after first-frame head (joint 15) and per-frame pelvis alignment
    (:364-399,470-473), acceleration error (:243-271,474) and head-orientation error ||I - R_gt R_pred^-1||_F
    (:338-346,469), x1000 where the reference does; per-sequence means averaged over the sequences the reference
    counts: on 'test' only those with head error < 0.9, root error < 300 mm and a non-zero acceleration error
    (:488-517), otherwise those with a non-zero acceleration error (:567-576).  Sums are one device tensor, so
    the multi-GPU reduction is one small all-reduce (seeme_amd.distributed.reduce_sums)."""

    NAMES = ("MPJPE", "ROOT_ERROR", "ACCL", "HEAD_ORIENTATION_ERROR", "mpjpe_interactee")

    def __init__(self):
        self.reset()

    def reset(self):
        self._sums = None            # [5 sums, 5 counts] float64 on the device of the first update

    @staticmethod
    def quat_to_rotmat(q):
        """(w,x,y,z) -> [M,3,3], normalising as transformations.quaternion_matrix does (compute.py:292)."""
        n = (q * q).sum(-1, keepdim=True)
        q = q * torch.sqrt(2.0 / n.clamp_min(1e-30))
        w, x, y, z = q.unbind(-1)
        R = torch.stack([1 - y * y - z * z, x * y - z * w, x * z + y * w,
                         x * y + z * w, 1 - x * x - z * z, y * z - x * w,
                         x * z - y * w, y * z + x * w, 1 - x * x - y * y], dim=-1).reshape(-1, 3, 3)
        return torch.where((n < 1e-15)[..., None], torch.eye(3, device=q.device, dtype=q.dtype).expand_as(R), R)

    @staticmethod
    def per_sequence(jts_pred, jts_ref, lengths, quat_pred=None, quat_ref=None):
        B, T = jts_ref.shape[:2]
        dev = jts_ref.device
        ref = jts_ref - jts_ref[:, 0:1, 15:16, :]
        pred = jts_pred - jts_pred[:, 0:1, 15:16, :]
        pelvis_ref, pelvis_pred = ref[:, :, [0]], pred[:, :, [0]]
        ref, pred = ref - pelvis_ref, pred - pelvis_pred
        lens = torch.as_tensor(lengths, device=dev).reshape(B)
        frame = torch.arange(T, device=dev)[None, :]
        mask = (frame < lens[:, None]).to(ref.dtype)
        out = {}
        out["MPJPE"] = ((pred - ref).norm(dim=-1).mean(dim=-1) * mask).sum(1) / lens * 1000.0
        out["ROOT_ERROR"] = ((pelvis_pred - pelvis_ref).norm(dim=-1).squeeze(-1) * mask).sum(1) / lens * 1000.0
        if T >= 3:   # X[t-1] - 2 X[t] + X[t+1] over the valid frames (compute.py:257-271)
            acc = lambda x: x[:, :-2] - 2 * x[:, 1:-1] + x[:, 2:]
            amask = (frame[:, : T - 2] < (lens - 2)[:, None]).to(ref.dtype)
            an = (acc(pred) - acc(ref)).norm(dim=-1).mean(dim=-1)
            out["ACCL"] = (an * amask).sum(1) / (lens - 2).clamp_min(1) * 1000.0
        else:
            out["ACCL"] = torch.zeros(B, device=dev, dtype=ref.dtype)
        if quat_pred is not None and quat_ref is not None:
            Rg = EgoMetrics.quat_to_rotmat(quat_ref.reshape(-1, 4).to(ref.dtype))
            Rp = EgoMetrics.quat_to_rotmat(quat_pred.reshape(-1, 4).to(ref.dtype))
            err = (torch.eye(3, device=dev, dtype=ref.dtype) - Rg @ torch.linalg.inv(Rp)).flatten(1).norm(dim=1).reshape(B, T)
            out["HEAD_ORIENTATION_ERROR"] = (err * mask).sum(1) / lens
        else:
            out["HEAD_ORIENTATION_ERROR"] = torch.zeros(B, device=dev, dtype=ref.dtype)
        return out

    def update(self, jts_pred, jts_ref, lengths, quat_pred=None, quat_ref=None, split: str = "test",
               jts_int=None, jts_int_gt=None):
        m = self.per_sequence(jts_pred, jts_ref, lengths, quat_pred, quat_ref)
        every = torch.ones_like(m["MPJPE"], dtype=torch.bool)
        if jts_int is not None and jts_int_gt is not None:      # POSE_ESTIMATION_TASK: every sequence counts (compute.py:476-481)
            a, g = jts_int - jts_int[:, :, [0]], jts_int_gt - jts_int_gt[:, :, [0]]
            T = a.shape[1]
            lens = torch.as_tensor(lengths, device=a.device).reshape(-1)
            mask = (torch.arange(T, device=a.device)[None, :] < lens[:, None]).to(a.dtype)
            m["mpjpe_interactee"] = ((a - g).norm(dim=-1).mean(dim=-1) * mask).sum(1) / lens * 1000.0
        else:
            m["mpjpe_interactee"] = torch.zeros_like(m["MPJPE"])
            every = ~every
        moving = m["ACCL"] > 0
        have_q = quat_pred is not None and quat_ref is not None
        if split == "test" and have_q:
            keep = moving & (m["HEAD_ORIENTATION_ERROR"] < 0.9) & (m["ROOT_ERROR"] < 300.0)
            cols = (keep, keep, keep, keep, every)
        else:                        # train / val: MPJPE and root error only (compute.py:567-576)
            none = torch.zeros_like(moving)
            cols = (moving, moving, none, none, every) if split != "test" else (moving, moving, moving, none, every)
        vals = torch.stack([(m[k] * c).sum() for k, c in zip(self.NAMES, cols)] + [c.sum().to(m["MPJPE"].dtype) for c in cols]).double()
        self._sums = vals if self._sums is None else self._sums + vals

    def sums(self):
        return torch.zeros(2 * len(self.NAMES), dtype=torch.float64) if self._sums is None else self._sums

    def compute(self, sums=None):
        s = (self.sums() if sums is None else sums).detach().double().cpu()
        n = len(self.NAMES)
        return {k: float(s[i]) / max(float(s[n + i]), 1.0) for i, k in enumerate(self.NAMES)} | {"count_seq": float(s[n])}


class SyntheticEgoDataModule:
    """Batches with the EgoBody/GIMO tuple layout (mld/data/humanml/data/dataset.py:1754-1794, 2479-2509): motion
    [B,T,2,72 | 66], transl [B,2,T,3], beta [B,2,T,10], utils [B,T,6], scene [B,P,3], length [B,1]; mean/std
    for ``renorm`` (mld/data/EgoBody.py:151-157).  Datasets are licence-gated, so this is what tests and
    benchmarks run on."""

    def __init__(self, nfeats=75, T=196, n_points=2048, seed=1234, device="cpu", pose_dim=None):
        self.nfeats, self.T, self.n_points = nfeats, T, n_points
        # per-person pose width of `motion`: EgoBody 72 (24 joints, axis-angle), GIMO 66 (root + 21 joints); the VAE sees
        # pose + 3 translation values = nfeats when TRAIN.ABLATION.PREDICT_TRANSL (mld.py:121-123)
        self.pose_dim = int(pose_dim) if pose_dim is not None else nfeats - 3
        g = torch.Generator().manual_seed(seed)
        self.mean = (0.1 * torch.randn(1, nfeats + 16, generator=g)).to(device)
        self.std = (0.5 + torch.rand(1, nfeats + 16, generator=g)).to(device)
        self.device = device
        self.is_mm = False
        self.seed = seed

    def renorm(self, features):
        return G.renorm(features, self.mean, self.std)

    def batch(self, B, idx=0, with_scene=False, lengths=None, pose_estimation=False, split="train"):
        g = torch.Generator().manual_seed(self.seed * 7919 + idx)
        T = self.T
        motion = 0.5 * torch.randn(B, T, 2, self.pose_dim, generator=g)
        transl = torch.randn(B, 2, T, 3, generator=g)
        beta = 0.5 * torch.randn(B, 2, 1, 10, generator=g).expand(B, 2, T, 10).contiguous()
        utils_ = torch.zeros(B, T, 6)
        length = torch.full((B, 1), T, dtype=torch.long) if lengths is None else torch.as_tensor(lengths).reshape(B, 1)
        dev = self.device
        out = [motion.to(dev), transl.to(dev), beta.to(dev), utils_.to(dev)]
        if with_scene:
            out.append((torch.rand(B, self.n_points, 3, generator=g) * 6 - 3).to(dev))
        out.append(length.to(dev))
        if with_scene and not pose_estimation:
            out.append([])          # img_path / dict_images slot
        if pose_estimation:         # interactee ground truth: motion [B,T,1,72], transl [B,1,T,3], beta [B,T,1,10] (mld.py:1119-1131)
            noise = 0.05 * torch.randn(B, T, self.pose_dim, generator=g)
            out += [(motion[:, :, 1] + noise).unsqueeze(2).to(dev), transl[:, 1:2].clone().to(dev), beta[:, 1].unsqueeze(2).to(dev)]
        return tuple(out)


# ----------------------------------------------------------------------------- the model
class _SceneEncoderHolder(nn.Module):
    """``proscene.scene_enc`` -- only this sub-module of ProHMRScene is on the path (prohmr_scene.py:102-104)."""

    def __init__(self):
        super().__init__()
        self.scene_enc = ResnetPointnet(512, 256)

    def encode_scene(self, scene):
        return self.scene_enc(scene)


class MLD(nn.Module):

    def __init__(self, cfg, datamodule=None, smpl_model: Optional[nn.Module] = None, **kwargs):
        super().__init__()
        self.cfg = cfg
        self.stage = cfg.TRAIN.STAGE
        self.condition = cfg.model.condition
        self.is_vae = cfg.model.vae
        self.predict_epsilon = cfg.TRAIN.ABLATION.PREDICT_EPSILON
        self.name_dataset = cfg.DATASET_NAME
        self.latent_dim = cfg.model.latent_dim
        self.guidance_scale = cfg.model.guidance_scale
        self.guidance_uncodp = cfg.model.guidance_uncondp
        self.datamodule = datamodule
        self.estimate = cfg.ESTIMATE
        self.predict_transl = cfg.TRAIN.ABLATION.PREDICT_TRANSL
        self.data_type = cfg.DATA_TYPE
        self.see_future = cfg.TEST.get("SEE_FUTURE", False)
        self.pred_global_orient = cfg.TEST.get("GLOBAL_ORIENT_PRED", True)           # mld.py:111
        # reference behaviour by default: the interactee condition is a SAMPLE of the posterior (mld.py:1280) and the CFG
        # scene pair is concatenated [cond, uncond] (mld.py:1144-1158) although _diffusion_reverse reads [uncond, cond]
        self.sample_mean = bool(cfg.TEST.get("SAMPLE_MEAN", False))
        self.cfg_scene_order = str(cfg.TEST.get("CFG_SCENE_ORDER", "reference"))
        if self.cfg_scene_order not in ("reference", "fixed"):
            raise ValueError("TEST.CFG_SCENE_ORDER must be 'reference' or 'fixed'")
        self.hip_backward = cfg.TRAIN.get("HIP_BACKWARD", True)   # hand-written backward of the denoiser chain (one head)
        self.hip_vae_backward = cfg.TRAIN.get("HIP_VAE_BACKWARD", True)   # stage 1: hand-written VAE backward (vae_train.py)
        self.hip_glue = cfg.TRAIN.get("HIP_GLUE", True)           # ... and of everything around it (stage2_glue.py); needs HIP_BACKWARD
        self.pose_estimation_task = cfg.TEST.get("POSE_ESTIMATION_TASK", False)      # mld.py:116
        if self.name_dataset == "egobody":                               # mld.py:122-125
            self.nfeats = 75 if self.predict_transl else 72
        elif self.name_dataset == "gimo":
            self.nfeats = 69 if self.predict_transl else 66
        else:
            self.nfeats = cfg.model.nfeats
        if "image" in self.condition:
            raise NotImplementedError("image conditioning (ProHMR ResNet-50 backbone) is outside the accelerated path")

        # SMPL (mld.py:151-153); frozen
        if smpl_model is not None:
            self.smpl_model = smpl_model
        else:
            self.smpl_model = SMPL(model_path=cfg.model.smpl_path, batch_size=cfg.TRAIN.BATCH_SIZE, gender="neutral")
        for p in self.smpl_model.parameters():
            p.requires_grad = False

        self.vae_type = cfg.model.get("vae_type", None) or \
            cfg.model.motion_vae.target.split(".")[-1].lower().replace("vae", "")      # mld.py:174-179
        if "scene" in self.condition:                                     # mld.py:182-207, 257-261
            self.proscene = _SceneEncoderHolder()
            for p in self.proscene.parameters():
                p.requires_grad = False
            # frozen scene encoder: TRAIN.SCENE_PRECISION bf16 runs the fused bf16-MFMA PointNet blocks
            self.proscene.scene_enc.precision = cfg.TRAIN.get("SCENE_PRECISION", self.proscene.scene_enc.precision)
            self.output_scene = nn.Sequential(nn.ReLU(), nn.Linear(512, 256))
        self.vae = instantiate_from_config(cfg.model.motion_vae)          # :264
        if self.stage == "diffusion":                                     # :267-271
            for p in self.vae.parameters():
                p.requires_grad = False
            # the frozen VAE only produces latents here: TRAIN.FROZEN_VAE_PRECISION fp16 runs it on the fp16-MFMA path
            self.vae.precision = cfg.TRAIN.get("FROZEN_VAE_PRECISION", self.vae.precision)
        self.denoiser = instantiate_from_config(cfg.model.denoiser)       # :282
        if not self.predict_epsilon:
            cfg.model.scheduler.params["prediction_type"] = "sample"
            cfg.model.noise_scheduler.params["prediction_type"] = "sample"
        self.scheduler = instantiate_from_config(cfg.model.scheduler)     # :286-287
        self.noise_scheduler = instantiate_from_config(cfg.model.noise_scheduler)
        if cfg.TRAIN.OPTIM.TYPE.lower() != "adamw":
            raise NotImplementedError("Do not support other optimizer for now.")
        self.optimizer = None          # built lazily: parameters must be on the device first
        self.losses = {k: MLDLosses(cfg) for k in ("train", "val", "test")}
        self.EgoMetric = EgoMetrics()
        self.do_classifier_free_guidance = self.guidance_scale > 1.0
        self.renorm = datamodule.renorm if datamodule is not None else (lambda x: x)
        self.times: List[float] = []

    # ------------------------------------------------------------------ optimiser (mld.py:292-299, base.py:157-158)
    def configure_optimizers(self, capturable: bool = False):
        if self.optimizer is not None and capturable and not self.optimizer.param_groups[0].get("capturable", False):
            raise RuntimeError("configure_optimizers(capturable=True): an optimiser built with capturable=False exists "
                               "(capture_training_step does not need a capturable torch AdamW: it uses seeme_adamw_step_dev)")
        if self.optimizer is None:
            params = [p for p in self.parameters() if p.requires_grad]
            self.optimizer = torch.optim.AdamW(params, lr=self.cfg.TRAIN.OPTIM.LR, capturable=capturable)
            self.sch = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=self.cfg.TRAIN.OPTIM.STEP_SIZE,
                                                       gamma=self.cfg.TRAIN.OPTIM.GAMMA)
            # eager steps update all tensors in one launch (seeme_adamw_step) on this optimiser's own state; the
            # hipGraph-captured step (capturable=True) stays on PyTorch's AdamW
            self._fused_adamw = None
            if not capturable and bool(self.cfg.TRAIN.get("FUSED_ADAMW", True)) and params and all(p.is_cuda for p in params):
                from .optim import FusedAdamWStep
                self._fused_adamw = FusedAdamWStep(self.optimizer)
        return {"optimizer": self.optimizer}

    def optimizer_update(self):
        """The AdamW update of the trainable tensors (after backward and the gradient all-reduce)."""
        if getattr(self, "_fused_adamw", None) is not None:
            self._fused_adamw.step()
        else:
            self.optimizer.step()

    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    # ------------------------------------------------------------------ condition assembly
    def _scene_token(self, scene, cfg_mask_train=False, mask=None, code_only=False):
        scene = scene.float()
        if cfg_mask_train and self.do_classifier_free_guidance:           # mld.py:917-919
            if mask is None:
                mask = torch.rand_like(scene) < self.guidance_uncodp
            scene = scene.masked_fill(mask, 0.0)
        s512 = self.proscene.encode_scene(scene)                          # HIP PointNet
        if code_only:
            return s512
        # output_scene = ReLU + Linear(512,256) (trainable, mld.py:257-261): torch op so that autograd sees it
        return self.output_scene(s512).unsqueeze(0)                       # [1,B,256]

    def _wearer_features(self, feats_ref, transl, idx):
        f = feats_ref[:, :, idx, :]
        if self.predict_transl:
            f = torch.cat([f, transl[:, idx, :, :]], dim=-1)
        if f.shape[-1] != self.vae.nfeats:
            raise ValueError(f"motion features are {f.shape[-1]} wide (pose {feats_ref.shape[-1]}"
                             f"{' + 3 translation' if self.predict_transl else ''}) but the VAE was built with nfeats "
                             f"{self.vae.nfeats} (model.nfeats)")
        return f.contiguous()

    def _sample_latent(self, feats, lengths, eps=None, mean=False):
        """vae.encode(...)[0] of the reference (a posterior sample, mld_vae.py:186-193) with the noise injectable:
        eps [1,B,256] replaces the draw; mean=True returns mu.  Returns (z [1,B,256], Normal)."""
        dist = self.vae.encode_dist(feats, lengths)
        mu, std = dist[0:1], dist[1:2].exp().pow(0.5)
        normal = torch.distributions.Normal(mu, std, validate_args=False)
        if mean:
            return mu, normal
        if eps is None:
            eps = torch.empty_like(mu).normal_()
        return mu + eps.to(mu) * std, normal

    # ------------------------------------------------------------------ reverse diffusion (mld.py:432-511)
    def _diffusion_reverse(self, encoder_hidden_states, lengths=None, latents=None, step_noise=None):
        bsz = encoder_hidden_states.shape[0]
        if self.do_classifier_free_guidance:
            bsz = bsz // 2
        if latents is None:
            latents = torch.randn((bsz, self.latent_dim[0], self.latent_dim[-1]), device=encoder_hidden_states.device,
                                  dtype=torch.float)
        latents = latents * self.scheduler.init_noise_sigma
        self.scheduler.set_timesteps(self.cfg.model.scheduler.num_inference_timesteps)
        eta = 0.0
        if "eta" in set(inspect.signature(self.scheduler.step).parameters.keys()):
            eta = self.cfg.model.scheduler.eta
        # the 50 (or 1000) sequential denoiser calls + CFG + scheduler.step are ONE kernel launch
        return self.denoiser.sample_loop(latents, encoder_hidden_states.contiguous(), self.scheduler, eta=eta,
                                         guidance_scale=self.guidance_scale if self.do_classifier_free_guidance else 1.0,
                                         step_noise=step_noise)            # [1,B,256]

    # ------------------------------------------------------------------ forward diffusion (mld.py:582-631)
    def _diffusion_process(self, latents, encoder_hidden_states, lengths=None, noise=None, timesteps=None):
        latents = latents.permute(1, 0, 2)
        if noise is None:
            noise = torch.randn_like(latents)
        bsz = latents.shape[0]
        if timesteps is None:
            timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (bsz,), device=latents.device)
        timesteps = timesteps.long()
        noisy = self.noise_scheduler.add_noise(latents.clone(), noise, timesteps)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.denoiser.parameters()):
            from .denoiser_train import denoiser_forward_hip_train, hip_train_supported
            if self.hip_backward and noisy.is_cuda and hip_train_supported(self.denoiser, encoder_hidden_states.shape[0]):
                noise_pred = denoiser_forward_hip_train(self.denoiser, noisy, timesteps, encoder_hidden_states)  # HIP fwd + bwd
            else:
                if self.hip_backward and noisy.is_cuda:
                    _warn_once("denoiser", "stage-2 training of this denoiser configuration (more than one attention head, or a shape the "
                               "hand-written backward does not cover) runs on the PyTorch-autograd twin (seeme_amd/denoiser_autograd.py): correct, "
                               "several times slower than the HIP forward/backward")
                noise_pred = denoiser_forward_torch(self.denoiser, noisy, timesteps, encoder_hidden_states)  # autograd twin
        else:
            noise_pred = self.denoiser(sample=noisy, timestep=timesteps, encoder_hidden_states=encoder_hidden_states,
                                       lengths=lengths)[0]                                              # HIP
        n_set = {"noise": noise, "noise_prior": 0, "noise_pred": noise_pred, "noise_pred_prior": 0}
        if not self.predict_epsilon:
            n_set["pred"] = noise_pred
            n_set["latent"] = latents
        return n_set

    # ------------------------------------------------------------------ stage-2 training forward (mld.py:887-1017)
    def train_diffusion_forward(self, batch, noise=None, timesteps=None, eps=None, masks=None):
        """Injection points for parity tests (not in the reference), in the order the reference draws: masks =
        (scene mask, interactee mask) -- the boolean results of ``rand_like(x) < guidance_uncondp`` (:917-919, :966-968),
        eps = (target rsample noise, condition rsample noise) [1,B,256] each, then `noise` and `timesteps` (:591-601)."""
        m_scene, m_int = masks if masks is not None else (None, None)
        e_z, e_c = eps if eps is not None else (None, None)
        if "scene" in self.condition:
            feats_ref, transl, beta, utils_, scene, length = batch[:6]
            scene = self._scene_token(scene, cfg_mask_train=True, mask=m_scene, code_only=True)      # [B,512] PointNet code
        else:
            feats_ref, transl, beta, utils_, length = batch[:5]
            scene = None
        feats_ref, transl = feats_ref.float(), transl.float()
        lengths = [feats_ref.shape[1]] * feats_ref.shape[0]
        glue = self._stage2_glue(int(scene is not None) + int("interactee" in self.condition)) if feats_ref.is_cuda else None
        if glue is not None:
            return self._train_diffusion_forward_glue(glue, feats_ref, transl, scene, lengths, noise, timesteps, e_z, e_c, m_int)
        if scene is not None:
            scene = self.output_scene(scene).unsqueeze(0)                     # [1,B,256]
        with torch.no_grad():
            idx = 0 if self.estimate == "wearer" else 1
            f_tgt = self._wearer_features(feats_ref, transl, idx)
            z_cond = None
            B = f_tgt.shape[0]
            shape = (1, B, self.vae.latent_dim)
            draw = lambda e: torch.empty(shape, device=f_tgt.device, dtype=torch.float32).normal_() if e is None else e.to(f_tgt)
            if "interactee" in self.condition:
                # target and interactee go through the frozen VAE as ONE batch of 2B sequences (the encode is a chain of
                # ~20 latency-bound launches); the three random draws keep the order of the two separate encodes:
                # rsample noise of the target, the CFG input mask, rsample noise of the condition (mld.py:944-984)
                eps_z = draw(e_z)
                f_int = self._wearer_features(feats_ref, transl, 1)
                if self.do_classifier_free_guidance:                       # mld.py:966-981
                    mask = (torch.rand_like(f_int) < self.guidance_uncodp) if m_int is None else m_int
                    f_int = f_int.masked_fill(mask, 0.0)
                eps_c = draw(e_c)
                dist = self.vae.encode_dist(torch.cat([f_tgt, f_int], dim=0), lengths + lengths)      # [2, 2B, 256]
                mu, std = dist[0:1], dist[1:2].exp().pow(0.5)              # mld_vae.py:186-190
                z = mu[:, :B] + eps_z * std[:, :B]                         # Normal(mu, std).rsample(), :192
                z_cond = mu[:, B:] + eps_c * std[:, B:]
            else:
                z, _ = self._sample_latent(f_tgt, lengths, e_z)
        if scene is not None and z_cond is not None:
            cond_emb = torch.cat([z_cond, scene], dim=0)                    # :991-993
        elif scene is not None:
            cond_emb = scene
        elif z_cond is not None:
            cond_emb = z_cond
        else:
            raise ValueError("no condition: MldDenoiser needs at least one condition token")
        return {**self._diffusion_process(z, cond_emb, lengths, noise=noise, timesteps=timesteps)}

    def _vae_trainer(self, T: int):
        """The hand-written stage-1 forward / backward (vae_train.VaeTrainer) when this model can take it."""
        if not self.hip_vae_backward:
            return None
        from .vae_train import VaeTrainer
        if not VaeTrainer.supported(self.vae, T):
            _warn_once("vae", "stage-1 training of this VAE configuration (more than one attention head, or more than 512 tokens) runs on the "
                       "PyTorch-autograd twin (seeme_amd/vae_autograd.py): correct, several times slower than the HIP forward/backward")
            return None
        t = getattr(self, "_vae_tr", None)
        if t is None or t.stale():
            t = VaeTrainer(self.vae)
            object.__setattr__(self, "_vae_tr", t)
        return t

    def _stage2_glue(self, n_tokens: int):
        """The hand-written forward/backward around the chain (stage2_glue.Stage2Glue) when this step can take it."""
        if not (self.hip_glue and self.hip_backward and torch.is_grad_enabled() and n_tokens >= 1):
            return None
        from .stage2_glue import Stage2Glue
        if not any(p.requires_grad for p in self.denoiser.parameters()) or not Stage2Glue.supported(self, n_tokens):
            return None
        g = getattr(self, "_glue", None)
        if g is None or g.stale():
            g = Stage2Glue(self)
            object.__setattr__(self, "_glue", g)
        return g

    def _train_diffusion_forward_glue(self, glue, feats_ref, transl, s512, lengths, noise, timesteps, e_z, e_c, m_int):
        """train_diffusion_forward + _diffusion_process with everything between the frozen encoders and the loss in
        stage2_glue's kernels; the random draws keep the reference's order (mld.py:917-919,944-984,591-601)."""
        with torch.no_grad():
            idx = 0 if self.estimate == "wearer" else 1
            f_tgt = self._wearer_features(feats_ref, transl, idx)
            B, dev = f_tgt.shape[0], f_tgt.device
            draw = lambda e: torch.empty(B, 256, device=dev, dtype=torch.float32).normal_() if e is None else e.to(f_tgt).reshape(B, 256)
            eps_z = draw(e_z)
            eps_c = None
            if "interactee" in self.condition:
                f_int = self._wearer_features(feats_ref, transl, 1)
                if self.do_classifier_free_guidance:                       # mld.py:966-981
                    mask = (torch.rand_like(f_int) < self.guidance_uncodp) if m_int is None else m_int
                    f_int = f_int.masked_fill(mask, 0.0)
                eps_c = draw(e_c)
                dist = self.vae.encode_dist(torch.cat([f_tgt, f_int], dim=0), lengths + lengths)      # [2, 2B, 256]
            else:
                dist = self.vae.encode_dist(f_tgt, lengths)
            noise = torch.randn(B, 1, 256, device=dev) if noise is None else noise.to(f_tgt)
            if timesteps is None:
                timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (B,), device=dev)
        noise_pred, latents = glue(dist, eps_z, eps_c, noise, timesteps.long(), s512)
        n_set = {"noise": noise.reshape(B, 1, 256), "noise_prior": 0, "noise_pred": noise_pred, "noise_pred_prior": 0}
        if not self.predict_epsilon:
            n_set["pred"] = noise_pred
            n_set["latent"] = latents
        return n_set

    # ------------------------------------------------------------------ stage-1 forward (mld.py:633-885)
    def train_vae_forward(self, batch, eps=None):
        """encode -> rsample -> decode -> renorm -> SMPL joints of reference and reconstruction.  m_ref / m_rst are the
        RENORMED features, as in the reference (mld.py:757,778,871-878).  GIMO keeps the first 21 joints and poses the
        reconstruction with the reference's global orientation (:826-828,852-859).  eps [1,B,256]: injected rsample noise."""
        feats_ref, transl, beta = batch[0].float(), batch[1].float(), batch[2].float()
        lengths = [feats_ref.shape[1]] * feats_ref.shape[0]
        idx = 0 if self.estimate == "wearer" else 1
        f_ref = self._wearer_features(feats_ref, transl, idx)
        gimo = self.name_dataset == "gimo" and self.data_type == "angle"
        nj = 21 if gimo else 24
        m_ref = self.renorm(f_ref)
        ref_orient = m_ref[:, :, :3] if gimo else None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.vae.parameters()):
            # stage-1 training: differentiable twins of the VAE and the SMPL joint regressor (PyTorch-ROCm autograd)
            tr = self._vae_trainer(f_ref.shape[1]) if f_ref.is_cuda else None
            if tr is not None:        # hand-written HIP forward-with-saves / backward of the encoder and the decoder (vae_train.py)
                mu, logvar = tr.encode(f_ref, lengths)
                std = logvar.exp().pow(0.5)
                dec = tr.decode
            else:                     # differentiable twins on PyTorch autograd
                from .vae_autograd import vae_encode_torch, vae_decode_torch
                mu, std = vae_encode_torch(self.vae, f_ref, lengths)
                dec = lambda zz, ll: vae_decode_torch(self.vae, zz, ll)
            dist_m = torch.distributions.Normal(mu, std, validate_args=False)
            z = dist_m.rsample() if eps is None else mu + eps.to(mu) * std
            m_rst = self.renorm(dec(z, lengths))                           # differentiable (geometry._Renorm)
            with torch.no_grad():
                joints_ref = self._feats_to_joints(m_ref, beta[:, idx])[:, :, :nj]
            joints_rst = self._feats_to_joints_torch(m_rst, beta[:, idx], orient=ref_orient)[:, :, :nj]
        else:
            z, dist_m = self._sample_latent(f_ref, lengths, eps)
            m_rst = self.renorm(self.vae.decode(z, lengths))
            joints_ref = self._feats_to_joints(m_ref, beta[:, idx])[:, :, :nj]
            joints_rst = self._feats_to_joints(m_rst, beta[:, idx], orient=ref_orient)[:, :, :nj]
        if self.is_vae:                                                    # mld.py:683-691
            dist_ref = torch.distributions.Normal(torch.zeros_like(dist_m.loc), torch.ones_like(dist_m.scale))
        else:
            dist_ref = dist_m
        return {"m_ref": m_ref, "m_rst": m_rst, "joints_ref": joints_ref, "joints_rst": joints_rst,
                "dist_m": dist_m, "dist_ref": dist_ref, "lat_m": z.permute(1, 0, 2)}

    # ------------------------------------------------------------------ features -> SMPL joints
    def _feats_to_joints(self, feats, betas, want_vertices=False, orient=None):
        """feats [B,T,F] (renormed).  'angle': [global_orient 3 | body_pose 21..23 joints | transl 3];
        'rot6d': 24 x 6 (the reference builds that body model in float64, mld.py:161-163).  orient [B,T,3]: global
        orientation to pose with instead of the features' own.  Returns joints [B,T,24,3] (and vertices)."""
        B, T, F = feats.shape
        if self.data_type == "rot6d":                                      # mld.py:1410-1449
            R = G.rot6d_to_rotmat(feats[..., :144].reshape(-1, 6).contiguous()).reshape(B * T, 24, 3, 3)
            zb = torch.zeros(B * T, 10, device=feats.device, dtype=torch.float32)     # create_beta=False: betas are not passed
            out = self.smpl_model(betas=zb, body_pose=R[:, 1:], global_orient=R[:, 0:1],
                                  pose2rot=False, return_verts=want_vertices, transl=None)
        else:
            nb = 69 if self.name_dataset == "egobody" else 63
            body = feats[:, :, 3:3 + nb].reshape(-1, nb).float()
            if nb < 69:                                                    # GIMO pads 21 -> 23 joints (mld.py:807-813)
                body = torch.cat([body, torch.zeros(body.shape[0], 69 - nb, device=body.device)], dim=1)
            go = (feats[:, :, :3] if orient is None else orient).reshape(-1, 3).float()
            tr = feats[:, :, -3:].reshape(-1, 3).float().contiguous() if self.predict_transl else None
            out = self.smpl_model(betas=betas.reshape(-1, 10).float(), body_pose=body.contiguous(),
                                  global_orient=go.contiguous(), transl=tr, pose2rot=True, return_verts=want_vertices)
        joints = out.joints.reshape(B, T, 45, 3)[:, :, :24]                # mld.py:771-773
        if want_vertices:
            return joints, out.vertices.reshape(B, T, -1, 3)
        return joints

    def _feats_to_joints_torch(self, feats, betas, orient=None):
        """Differentiable version of _feats_to_joints for stage-1 training ('angle' data: axis-angle pose)."""
        from .vae_autograd import smpl_joints_torch
        if self.data_type != "angle":
            raise NotImplementedError("stage-1 training twin: DATA_TYPE 'angle'")
        B, T, _ = feats.shape
        nb = 69 if self.name_dataset == "egobody" else 63
        pose = feats[:, :, :3 + nb]
        if orient is not None:
            pose = torch.cat([orient.to(pose), pose[:, :, 3:]], dim=-1)
        pose = pose.reshape(B * T, 3 + nb)
        if nb < 69:                                                        # GIMO pads 21 -> 23 joints (mld.py:807-813)
            pose = torch.cat([pose, torch.zeros(B * T, 69 - nb, device=pose.device, dtype=pose.dtype)], dim=1)
        tr = feats[:, :, -3:].reshape(B * T, 3) if self.predict_transl else None
        if self.hip_vae_backward and pose.is_cuda:           # HIP forward + hand-written backward (smpl._JointsAA)
            from .smpl import smpl_joints_hip
            return smpl_joints_hip(self.smpl_model, betas.reshape(-1, 10).float(), pose.float(), tr).reshape(B, T, 24, 3)
        return smpl_joints_torch(self.smpl_model, betas.reshape(-1, 10).float(), pose.float(), tr).reshape(B, T, 24, 3)

    # ------------------------------------------------------------------ evaluation (mld.py:1076-1905, live part)
    @torch.no_grad()
    def ego_eval(self, batch, latents=None, want_vertices=False, cond_noise=None, step_noise=None):
        """Injection points for parity tests (not in the reference): latents [B,1,256] initial noise; cond_noise = eps
        [1,B,256] of the condition sample (stage 'vae': of the target's sample), or a pair (eps_cond, eps_uncond) with
        classifier-free guidance; step_noise for DDPM."""
        int_gt = None
        if self.pose_estimation_task:       # batch ends with the interactee's ground truth (mld.py:1119-1131)
            batch, int_gt = tuple(batch[:-3]), tuple(t.float() for t in batch[-3:])
        eps_c, eps_u = cond_noise if isinstance(cond_noise, (tuple, list)) else (cond_noise, None)
        if "scene" in self.condition:
            feats_ref, transl, beta, utils_, scene, length = batch[:6]
            scene_tok = None
            if self.stage != "vae":
                scene_tok = self._scene_token(scene)
                if self.do_classifier_free_guidance:                       # zero-scene branch (:1144-1158)
                    unc = self._scene_token(torch.zeros_like(scene))
                    # the reference concatenates [scene, scene_uncond] while _diffusion_reverse takes the FIRST half as
                    # unconditional (:489): reproduced by default, TEST.CFG_SCENE_ORDER 'fixed' puts uncond first
                    scene_tok = torch.cat([scene_tok, unc] if self.cfg_scene_order == "reference" else [unc, scene_tok], dim=1)
        else:
            feats_ref, transl, beta, utils_, length = batch[:5]
            scene_tok = None
        feats_ref, transl, beta = feats_ref.float(), transl.float(), beta.float()
        lengths = length.long().reshape(-1).tolist()
        idx_ref = 0 if self.estimate == "wearer" else 1
        start = time.time()
        if self.stage in ("diffusion", "vae_diffusion"):
            text_emb = None
            if "interactee" in self.condition:                             # :1271-1295
                f_int = self._wearer_features(feats_ref, transl, 1)
                text_emb, _ = self._sample_latent(f_int, lengths, eps_c, mean=self.sample_mean)
                if self.do_classifier_free_guidance:
                    unc, _ = self._sample_latent(torch.zeros_like(f_int), lengths, eps_u, mean=self.sample_mean)
                    text_emb = torch.cat([unc, text_emb], dim=1)
            toks = [t for t in (text_emb, scene_tok) if t is not None]
            if not toks:
                raise ValueError("no condition tokens")
            cond_emb = torch.cat(toks, dim=0)                               # [N, B or 2B, 256]
            z = self._diffusion_reverse(cond_emb.permute(1, 0, 2), lengths, latents=latents, step_noise=step_noise)
        elif self.stage == "vae":                                          # :1328-1352: reconstruction of the target itself
            z, _ = self._sample_latent(self._wearer_features(feats_ref, transl, idx_ref), lengths, eps_c, mean=self.sample_mean)
        else:
            raise ValueError(f"Not support this stage {self.stage}!")
        if self.see_future:
            lengths = [int(i // 2) for i in lengths]
        feats_rst = self.vae.decode(z, lengths)
        self.times.append(time.time() - start)                              # mld.py:1367-1368
        min_len = min(feats_ref.shape[1], feats_rst.shape[1])
        f_ref = self._wearer_features(feats_ref[:, :min_len], transl[:, :, :min_len], idx_ref)
        f_ref, f_rst = self.renorm(f_ref), self.renorm(feats_rst[:, :min_len].contiguous())
        b_ref = beta[:, idx_ref, :min_len]
        # egobody: TEST.GLOBAL_ORIENT_PRED False poses the prediction with the reference orientation (:1497-1501)
        o_rst = f_ref[:, :, :3] if (self.data_type == "angle" and self.name_dataset == "egobody" and not self.pred_global_orient) else None
        out_ref = self._feats_to_joints(f_ref, b_ref, want_vertices)
        out_rst = self._feats_to_joints(f_rst, b_ref, want_vertices, orient=o_rst)
        joints_ref, joints_rst = (out_ref[0], out_rst[0]) if want_vertices else (out_ref, out_rst)
        f_int_r = self.renorm(self._wearer_features(feats_ref[:, :min_len], transl[:, :, :min_len], 1))
        joints_int = self._feats_to_joints(f_int_r, beta[:, 1, :min_len])
        joints_int_gt = None
        if int_gt is not None:              # mld.py:1843-1866: SMPL joints of the interactee's ground-truth motion
            g_motion, g_transl, _g_beta = int_gt
            f_gt = g_motion[:, :min_len, 0]
            if self.predict_transl:
                f_gt = torch.cat([f_gt, g_transl[:, 0, :min_len]], dim=-1)
            joints_int_gt = self._feats_to_joints(self.renorm(f_gt.contiguous()), beta[:, 1, :min_len])
        if self.data_type == "angle":
            quat = lambda f, o=None: G.aa_to_quat((f[:, :, :3] if o is None else o).reshape(-1, 3).contiguous())
        else:
            quat = lambda f, o=None: None
        rs = {"m_ref": f_ref, "m_rst": f_rst, "joints_ref": joints_ref, "joints_rst": joints_rst,
              "orientation_quat_rst": quat(f_rst, o_rst), "orientation_quat_ref": quat(f_ref),
              "root_interactee": joints_int[:, :, 0], "joints_interactee": joints_int,
              "orientation_quat_int": quat(f_int_r), "joints_interactee_gt": joints_int_gt, "lengths": lengths,
              "list_names": {}, "lat_t": z}
        if want_vertices:
            rs["vertices_ref"], rs["vertices_rst"] = out_ref[1], out_rst[1]
        return rs

    def forward(self, batch, **kw):
        return self.ego_eval(batch, **kw)

    sample = forward

    # ------------------------------------------------------------------ step dispatch (mld.py:2037-2130, base.py:38-53)
    def allsplit_step(self, split: str, batch, batch_idx=0):
        loss = None
        if split in ("train", "val"):
            if self.stage == "vae":
                rs_set = self.train_vae_forward(batch)
            elif self.stage == "diffusion":
                rs_set = self.train_diffusion_forward(batch)
            else:
                raise ValueError(f"Not support this stage {self.stage}!")
            loss = self.losses[split].update(rs_set)
        if split in ("val", "test"):
            rs_set = self.ego_eval(batch)
            self.EgoMetric.update(rs_set["joints_rst"], rs_set["joints_ref"], rs_set["lengths"],
                                  rs_set.get("orientation_quat_rst"), rs_set.get("orientation_quat_ref"), split=split,
                                  jts_int=rs_set.get("joints_interactee"), jts_int_gt=rs_set.get("joints_interactee_gt"))
        if split == "test":
            return rs_set["joints_rst"]
        return loss

    def training_step(self, batch, batch_idx=0):
        return self.allsplit_step("train", batch, batch_idx)

    def validation_step(self, batch, batch_idx=0):
        return self.allsplit_step("val", batch, batch_idx)

    def test_step(self, batch, batch_idx=0):
        return self.allsplit_step("test", batch, batch_idx)

    def capture_training_step(self, batch, warmup: int = 3):
        """One stage-2 training step (frozen HIP encoders, denoiser forward + backward, AdamW) captured as ONE
        hipGraph on static input buffers -- the launch-bound chain of small kernels replays without host work.
        Returns replay(new_batch=None) -> loss tensor (static).  The AdamW update in the graph is the one-launch
        ``seeme_adamw_step_dev`` on the existing optimiser's state (step count and learning rate live on the device, so
        replays advance them and an LR-scheduler edit reaches the graph); with data parallelism the graph ends after
        backward and the gradient all-reduce + optimiser step stay eager after the replay."""
        from . import distributed as D
        world = D.world()[1]
        self.configure_optimizers()
        fused = getattr(self, "_fused_adamw", None)
        if fused is None:
            raise NotImplementedError("capture_training_step needs the one-launch AdamW (TRAIN.FUSED_ADAMW, ROCm parameters): "
                                      "torch.optim.AdamW built with capturable=False cannot be stepped inside a capture")
        if self.stage != "diffusion":
            raise NotImplementedError("capture_training_step: stage 'diffusion'")
        static = [b.clone() if torch.is_tensor(b) else b for b in batch]
        losses = self.losses["train"]

        def step():
            loss = losses.update(self.train_diffusion_forward(static), accumulate=False)
            self.backward(loss)
            if world == 1:
                fused.step(device_step=True)
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # real steps: the first discovers the parameters on the path, the next ones create the gradient bucket and the
            # optimiser's pointer tables -- all of which must exist before the capture
            for _ in range(max(3 if self.grad_bucket() is None else 1, warmup)):
                step()
                if world > 1:
                    b = self.grad_bucket()
                    D.allreduce_gradients(self._used_params) if b is None else b.allreduce()
                    self.optimizer_update()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()

        def replay(new_batch=None):
            if new_batch is not None:
                for dst, src in zip(static, new_batch):
                    if torch.is_tensor(dst):
                        dst.copy_(src)
            if world == 1:
                fused.sync_lr()
            graph.replay()
            if world > 1:
                self.grad_bucket().allreduce()
                self.optimizer_update()
            else:
                fused.note_replay()
            return static_loss

        return replay

    def grad_bucket(self):
        """The flat gradient buffer (distributed.GradBucket) of the parameters that RECEIVE gradients -- known after the
        first backward, which runs on separate tensors (a parameter off the path, e.g. denoiser.mem_pos.pe, keeps
        ``.grad`` None for ever, so AdamW never decays it, as in the reference) -- laid out around the HIP backward's own
        gradient block when stage 2 trains through it; rebuilt when that block is re-created.  None before that first step."""
        from . import distributed as D
        used = getattr(self, "_used_params", None)
        if used is None:
            return None
        pack = getattr(self.denoiser, "_train_pack", None) if self.stage == "diffusion" and self.hip_backward else None
        b = getattr(self, "_bucket", None)
        if b is None or not b.matches(pack):
            b = D.GradBucket(used, pack)
            self._bucket = b
        return b

    def backward(self, loss):
        """loss.backward() into the flat bucket; the very first call discovers which parameters are on the path.
        Returns the bucket (None on the discovery step)."""
        b = self.grad_bucket()
        if b is None:
            for p in self.trainable_parameters():
                p.grad = None
            loss.backward()
            self._used_params = [p for p in self.trainable_parameters() if p.grad is not None]
        else:
            b.prepare()
            loss.backward()
        # the caller usually keeps `loss` (logging): cut it loose from the autograd graph, or the graph's AccumulateGrad nodes --
        # created on THIS stream -- stay alive and are reused by a later backward on another stream (a hipGraph capture runs on
        # a side stream: the engine then synchronises with the non-capturing stream and the capture dies in hipStreamEndCapture)
        if loss.grad_fn is not None and not torch.cuda.is_current_stream_capturing():
            loss.detach_()
        return b

    def optimizer_step(self, loss, events=None):
        """backward -> (data-parallel) gradient all-reduce -> AdamW step; what Lightning + DDP do around
        training_step in the reference (train.py:127-149).  Every gradient lives in one flat buffer, so the exchange is
        ONE all_reduce of it in place.  `events`: optional 4 torch.cuda.Event (start, after backward, after all-reduce,
        after AdamW) recorded on the current stream."""
        from . import distributed as D
        self.configure_optimizers()
        if events is not None:
            events[0].record()
        bucket = self.backward(loss)
        if events is not None:
            events[1].record()
        if bucket is None:
            D.allreduce_gradients(self._used_params)
        else:
            bucket.allreduce()
        if events is not None:
            events[2].record()
        self.optimizer_update()
        if events is not None:
            events[3].record()
