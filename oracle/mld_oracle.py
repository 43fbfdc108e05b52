"""CPU oracle for the SEE-ME motion-latent-diffusion hot path (numpy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``seeme_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported baseline.

Every function restates one piece of the reference (file:line cited, paths
relative to the upstream repository root) in plain numpy on state-dict
arrays ``P[name]`` with the reference's parameter names.

Pinning: the transformer pieces (VAE encode/decode, denoiser forward,
PointNet, geometry helpers, timestep embedding) are checked against outputs of
the reference's own PyTorch modules, generated in the build container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``.
The DDIM/DDPM arithmetic (``diffusers``, unpinned upstream) and SMPL LBS
(``smplx==0.1.28`` + SMPL_NEUTRAL.pkl) live in third-party packages that are
absent from the reference tree and from this image: for those two pieces
**parity is unpinned** -- they restate the published algorithms and are
anchored on the reference's call sites (mld/models/modeltype/mld.py:456-497,
598-606, 151-163, 764-770) and on self-consistency properties only.

Layouts: this file is batch-first internally ([B, S, D]); the wrappers at the
bottom accept/return the reference's layouts.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # scipy is in the image; erf is the only thing needed from it
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

Array = np.ndarray
Params = Dict[str, Array]

# --------------------------------------------------------------------------
# primitives (torch.nn semantics)
# --------------------------------------------------------------------------


def linear(x: Array, w: Array, b: Optional[Array] = None) -> Array:
    """torch.nn.functional.linear: x @ w.T + b."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def layer_norm(x: Array, w: Array, b: Array, eps: float = 1e-5) -> Array:
    """torch.nn.LayerNorm over the last axis (biased variance, eps inside sqrt)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w + b


def gelu(x: Array) -> Array:
    """Exact (erf) GELU, torch default (cross_attention.py:430-431)."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def relu(x: Array) -> Array:
    return np.maximum(x, 0)


def silu(x: Array) -> Array:
    return x / (1.0 + np.exp(-x))


def softmax(x: Array, axis: int) -> Array:
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def lengths_to_mask(lengths: Sequence[int], max_len: Optional[int] = None) -> Array:
    """mld/utils/temos_utils.py:10-17 -> bool [B, max_len], True = valid frame."""
    lengths = np.asarray(lengths, dtype=np.int64)
    max_len = int(max_len) if max_len else int(lengths.max())
    return np.arange(max_len)[None, :] < lengths[:, None]


def mha(q_in: Array, k_in: Array, v_in: Array, P: Params, pre: str, nhead: int,
        key_padding_mask: Optional[Array] = None) -> Array:
    """torch.nn.MultiheadAttention forward (eval), batch-first here.

    q_in [B,Sq,D], k_in/v_in [B,Sk,D]; key_padding_mask bool [B,Sk], True = ignore
    (cross_attention.py:264,286-289).
    """
    w, b = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    D = q_in.shape[-1]
    q = linear(q_in, w[:D], b[:D])
    k = linear(k_in, w[D:2 * D], b[D:2 * D])
    v = linear(v_in, w[2 * D:], b[2 * D:])
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    hd = D // nhead
    q = q.reshape(B, Sq, nhead, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, Sk, nhead, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, Sk, nhead, hd).transpose(0, 2, 1, 3)
    s = (q @ k.transpose(0, 1, 3, 2)) * q.dtype.type(1.0 / math.sqrt(hd))
    if key_padding_mask is not None:
        s = np.where(key_padding_mask[:, None, None, :], -np.inf, s).astype(q.dtype)
    p = softmax(s, axis=-1)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, Sq, D)
    return linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


_ACT = {"gelu": gelu, "relu": relu}

# --------------------------------------------------------------------------
# transformer blocks (mld/models/operator/cross_attention.py)
# --------------------------------------------------------------------------


def encoder_layer_post(x: Array, P: Params, pre: str, nhead: int, act: str,
                       key_padding_mask: Optional[Array]) -> Array:
    """TransformerEncoderLayer.forward_post, cross_attention.py:281-294
    (same body as mdiff_transformer.py:54-67)."""
    x = x + mha(x, x, x, P, pre + "self_attn.", nhead, key_padding_mask)
    x = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    h = _ACT[act](linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"]))
    x = x + linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"])


def decoder_layer_post(x: Array, mem: Array, P: Params, pre: str, nhead: int, act: str,
                       tgt_key_padding_mask: Optional[Array]) -> Array:
    """TransformerDecoderLayer.forward_post, cross_attention.py:345-367."""
    x = x + mha(x, x, x, P, pre + "self_attn.", nhead, tgt_key_padding_mask)
    x = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    x = x + mha(x, mem, mem, P, pre + "multihead_attn.", nhead, None)
    x = layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
    h = _ACT[act](linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"]))
    x = x + linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return layer_norm(x, P[pre + "norm3.weight"], P[pre + "norm3.bias"])


def _skip_stack(x: Array, P: Params, pre: str, num_layers: int, layer_fn) -> Array:
    """SkipTransformerEncoder/Decoder.forward, cross_attention.py:46-65,118-147."""
    nb = (num_layers - 1) // 2
    xs = []
    for i in range(nb):
        x = layer_fn(x, f"{pre}input_blocks.{i}.")
        xs.append(x)
    x = layer_fn(x, f"{pre}middle_block.")
    for i in range(nb):
        x = np.concatenate([x, xs.pop()], axis=-1)
        x = linear(x, P[f"{pre}linear_blocks.{i}.weight"], P[f"{pre}linear_blocks.{i}.bias"])
        x = layer_fn(x, f"{pre}output_blocks.{i}.")
    return layer_norm(x, P[pre + "norm.weight"], P[pre + "norm.bias"])


# --------------------------------------------------------------------------
# MldVae (mld/models/architectures/mld_vae.py)
# --------------------------------------------------------------------------

VAE_LAYERS, VAE_HEADS, VAE_ACT = 5, 1, "gelu"  # hard-coded at mld_vae.py:51-53


def vae_encode(P: Params, features: Array, lengths: Sequence[int]) -> Tuple[Array, Array]:
    """MldVae.encode, mld_vae.py:128-193 (MLP_DIST False, PE_TYPE 'mld').

    features [B,T,F] -> (mu [1,B,256], std [1,B,256]); the reference then draws
    ``Normal(mu, std).rsample()`` (RNG, not compared).
    """
    B, T, _ = features.shape
    mask = lengths_to_mask(lengths)                                  # :143
    assert mask.shape[1] == T, "reference requires max(lengths) == nframes"
    x = linear(features, P["skel_embedding.weight"], P["skel_embedding.bias"])   # :147
    tok = np.broadcast_to(P["global_motion_token"][None], (B,) + P["global_motion_token"].shape)  # :154
    aug = np.concatenate([np.ones((B, tok.shape[1]), bool), mask], axis=1)         # :157-160
    xseq = np.concatenate([tok, x], axis=1)                                        # :164
    xseq = xseq + P["query_pos_encoder.pe"][: xseq.shape[1], 0][None]              # :171
    kpm = ~aug
    out = _skip_stack(xseq, P, "encoder.", VAE_LAYERS,
                      lambda h, pre: encoder_layer_post(h, P, pre, VAE_HEADS, VAE_ACT, kpm))
    mu, logvar = out[:, 0], out[:, 1]                                              # :186-187
    std = np.power(np.exp(logvar), x.dtype.type(0.5))                              # :190
    return mu[None], std[None]


def vae_decode(P: Params, z: Array, lengths: Sequence[int]) -> Array:
    """MldVae.decode, arch 'encoder_decoder', mld_vae.py:195-256.  z [1,B,256] -> [B,T,F]."""
    mask = lengths_to_mask(lengths)
    B, T = mask.shape
    D = z.shape[-1]
    q = np.zeros((B, T, D), z.dtype) + P["query_pos_decoder.pe"][:T, 0][None]      # :198,232
    mem = np.transpose(z, (1, 0, 2))                                               # [B,1,D]
    kpm = ~mask
    out = _skip_stack(q, P, "decoder.", VAE_LAYERS,
                      lambda h, pre: decoder_layer_post(h, mem, P, pre, VAE_HEADS, VAE_ACT, kpm))
    return linear(out, P["final_layer.weight"], P["final_layer.bias"])            # :251 (pad not zeroed, :253)


# --------------------------------------------------------------------------
# MldDenoiser (mld_denoiser.py, mdiff_transformer.py, tools/embeddings.py)
# --------------------------------------------------------------------------


def timestep_features(t: Array, dim: int = 256, flip_sin_to_cos: bool = True,
                      freq_shift: float = 0.0, max_period: float = 10000.0,
                      dtype=np.float32) -> Array:
    """get_timestep_embedding, tools/embeddings.py:245-285.  t [B] -> [B, dim]."""
    half = dim // 2
    expo = (-math.log(max_period) * np.arange(half, dtype=np.float32)).astype(np.float32)
    expo = expo / np.float32(half - freq_shift)
    freq = np.exp(expo).astype(dtype)
    arg = np.asarray(t).astype(dtype)[:, None] * freq[None, :]
    emb = np.concatenate([np.sin(arg), np.cos(arg)], axis=-1)
    if flip_sin_to_cos:
        emb = np.concatenate([emb[:, half:], emb[:, :half]], axis=-1)
    if dim % 2 == 1:
        emb = np.pad(emb, ((0, 0), (0, 1)))
    return emb.astype(dtype)


def time_embedding(P: Params, feat: Array, pre: str = "time_embedding.") -> Array:
    """TimestepEmbedding.forward, tools/embeddings.py:298-305."""
    h = silu(linear(feat, P[pre + "linear_1.weight"], P[pre + "linear_1.bias"]))
    return linear(h, P[pre + "linear_2.weight"], P[pre + "linear_2.bias"])


def stylization(P: Params, pre: str, h: Array, emb: Array) -> Array:
    """StylizationBlock.forward, mdiff_transformer.py:152-163.  h [B,T,D], emb [B,D]."""
    eo = linear(silu(emb), P[pre + "emb_layers.1.weight"], P[pre + "emb_layers.1.bias"])[:, None, :]
    D = h.shape[-1]
    scale, shift = eo[..., :D], eo[..., D:]
    h = layer_norm(h, P[pre + "norm.weight"], P[pre + "norm.bias"]) * (1 + scale) + shift
    return linear(silu(h), P[pre + "out_layers.2.weight"], P[pre + "out_layers.2.bias"])


def linear_cross_attention(P: Params, pre: str, x: Array, xf: Array, emb: Array, nhead: int) -> Array:
    """LinearTemporalCrossAttention.forward, mdiff_transformer.py:219-239.
    x [B,T,D], xf [B,N,D], emb [B,D]."""
    B, T, D = x.shape
    N = xf.shape[1]
    H = nhead
    q = linear(layer_norm(x, P[pre + "norm.weight"], P[pre + "norm.bias"]),
               P[pre + "query.weight"], P[pre + "query.bias"])
    xfn = layer_norm(xf, P[pre + "text_norm.weight"], P[pre + "text_norm.bias"])
    k = linear(xfn, P[pre + "key.weight"], P[pre + "key.bias"])
    q = softmax(q.reshape(B, T, H, -1), axis=-1)
    k = softmax(k.reshape(B, N, H, -1), axis=1)
    v = linear(xfn, P[pre + "value.weight"], P[pre + "value.bias"]).reshape(B, N, H, -1)
    att = np.einsum("bnhd,bnhl->bhdl", k, v)
    y = np.einsum("bnhd,bhdl->bnhl", q, att).reshape(B, T, D)
    return x + stylization(P, pre + "proj_out.", y, emb)


def ffn_stylized(P: Params, pre: str, x: Array, emb: Array) -> Array:
    """FFN.forward, mdiff_transformer.py:251-254."""
    y = linear(gelu(linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"])),
               P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return x + stylization(P, pre + "proj_out.", y, emb)


def md_layer(P: Params, pre: str, x: Array, xf: Array, emb: Array, nhead: int) -> Array:
    """LinearTemporalDiffusionTransformerDecoderLayer.forward, mdiff_transformer.py:286-304.
    x [B,L,D] (L latent tokens), xf [B,N,D], emb [B,D] (batch-first here)."""
    L = x.shape[1]
    seq = np.concatenate([x, xf, emb[:, None, :]], axis=1)                        # :295
    seq = encoder_layer_post(seq, P, pre + "sa_block.", nhead, "relu", None)       # :296 (ff 1024 relu :279)
    x = seq[:, :L]                                                                 # :297
    x = linear_cross_attention(P, pre + "ca_block.", x, xf, emb, nhead)            # :300
    return ffn_stylized(P, pre + "ffn.", x, emb)                                   # :301


def denoiser_forward(P: Params, sample: Array, timestep, cond: Array,
                     num_layers: int = 5, nhead: int = 1,
                     flip_sin_to_cos: bool = True, freq_shift: float = 0.0) -> Array:
    """MldDenoiser.forward, mld_denoiser.py:151-244 (arch trans_enc, MD_TRANS, SKIP_CONNECT,
    DIFF_PE_TYPE 'mld', text_encoded_dim == latent_dim).

    sample [B,L,D]; timestep scalar or [B]; cond **seq-first** [N,B,D] as the reference takes it.
    Returns [B,L,D].
    """
    B, L, D = sample.shape
    t = np.broadcast_to(np.asarray(timestep), (B,))                                # :167
    emb = time_embedding(P, timestep_features(t, D, flip_sin_to_cos, freq_shift, dtype=sample.dtype))  # :168-171
    xf = np.transpose(cond, (1, 0, 2))
    x = sample + P["query_pos.pe"][:L, 0][None]                                    # :210
    out = _skip_stack(x, P, "encoder.", num_layers,
                      lambda h, pre: md_layer(P, pre, h, xf, emb, nhead))          # :212
    return out[:, :L]                                                              # :222,242


# --------------------------------------------------------------------------
# schedulers (diffusers DDIM / DDPM restated) -- PARITY UNPINNED
# --------------------------------------------------------------------------


def make_betas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
               beta_schedule="scaled_linear") -> Array:
    if beta_schedule == "linear":
        return np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
    if beta_schedule == "scaled_linear":
        return (np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                            dtype=np.float32) ** 2).astype(np.float32)
    raise NotImplementedError(beta_schedule)


def alphas_cumprod(betas: Array) -> Array:
    return np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)


def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000, steps_offset: int = 1) -> Array:
    ratio = num_train_timesteps // num_inference_steps
    return ((np.arange(num_inference_steps) * ratio).round()[::-1].astype(np.int64) + steps_offset)


def ddim_step(acp: Array, eps: Array, t: int, x: Array, num_inference_steps: int, eta: float = 0.0,
              noise: Optional[Array] = None, set_alpha_to_one: bool = False,
              num_train_timesteps: int = 1000, prediction_type: str = "epsilon") -> Array:
    """DDIMScheduler.step (call site mld.py:495-497; params configs/modules/scheduler.yaml:1-14)."""
    f = x.dtype.type
    t_prev = t - num_train_timesteps // num_inference_steps
    a_t = f(acp[t])
    a_prev = f(acp[t_prev]) if t_prev >= 0 else (f(1.0) if set_alpha_to_one else f(acp[0]))
    b_t = f(1) - a_t
    if prediction_type == "epsilon":
        x0 = (x - np.sqrt(b_t) * eps) / np.sqrt(a_t)
    else:  # "sample"
        x0 = eps
        eps = (x - np.sqrt(a_t) * x0) / np.sqrt(b_t)
    var = (f(1) - a_prev) / (f(1) - a_t) * (f(1) - a_t / a_prev)
    std = f(eta) * np.sqrt(var)
    dirn = np.sqrt(f(1) - a_prev - std * std) * eps
    prev = np.sqrt(a_prev) * x0 + dirn
    if eta > 0:
        prev = prev + std * noise
    return prev.astype(x.dtype)


def ddpm_add_noise(acp: Array, x0: Array, noise: Array, t: Array) -> Array:
    """DDPMScheduler.add_noise (call site mld.py:604-606). t [B] int."""
    f = x0.dtype
    a = np.sqrt(acp[t]).astype(f)
    s = np.sqrt(1.0 - acp[t]).astype(f)
    shp = (-1,) + (1,) * (x0.ndim - 1)
    return a.reshape(shp) * x0 + s.reshape(shp) * noise


def ddpm_step(acp: Array, eps: Array, t: int, x: Array, noise: Optional[Array] = None,
              clip_sample: bool = False) -> Array:
    """DDPMScheduler.step, variance_type fixed_small, epsilon prediction, 1000 inference steps
    (configs/modules_novae/scheduler.yaml:16-26)."""
    f = x.dtype.type
    a_t = f(acp[t])
    a_prev = f(acp[t - 1]) if t > 0 else f(1.0)
    b_t, b_prev = f(1) - a_t, f(1) - a_prev
    cur_a = a_t / a_prev
    cur_b = f(1) - cur_a
    x0 = (x - np.sqrt(b_t) * eps) / np.sqrt(a_t)
    if clip_sample:
        x0 = np.clip(x0, -1, 1)
    c0 = np.sqrt(a_prev) * cur_b / b_t
    c1 = np.sqrt(cur_a) * b_prev / b_t
    mean = c0 * x0 + c1 * x
    if t > 0:
        var = max(b_prev / b_t * cur_b, f(1e-20))
        mean = mean + np.sqrt(f(var)) * noise
    return mean.astype(x.dtype)


def diffusion_reverse(P: Params, cond_bf: Array, latents: Array, num_inference_steps: int = 50,
                      eta: float = 0.0, guidance_scale: float = 1.0, scheduler: str = "ddim",
                      step_noise: Optional[Array] = None, **den_kw) -> Array:
    """MLD._diffusion_reverse, mld.py:432-511, with the RNG draws injected.

    cond_bf batch-first [B or 2B, N, D] (uncond first when CFG, :489); latents [B,1,D] initial noise
    (already multiplied by init_noise_sigma = 1).  Returns [1,B,D].
    """
    acp = alphas_cumprod(make_betas())
    cfg = guidance_scale > 1.0
    if scheduler == "ddim":
        ts = ddim_timesteps(num_inference_steps)
    else:
        ts = np.arange(num_inference_steps)[::-1].astype(np.int64)
    x = latents
    cond_sf = np.transpose(cond_bf, (1, 0, 2))
    for i, t in enumerate(ts):
        xin = np.concatenate([x, x], axis=0) if cfg else x
        eps = denoiser_forward(P, xin, int(t), cond_sf, **den_kw)
        if cfg:
            e_u, e_c = np.split(eps, 2, axis=0)
            eps = e_u + x.dtype.type(guidance_scale) * (e_c - e_u)
        nz = None if step_noise is None else step_noise[i]
        if scheduler == "ddim":
            x = ddim_step(acp, eps, int(t), x, num_inference_steps, eta, nz)
        else:
            x = ddpm_step(acp, eps, int(t), x, nz)
    return np.transpose(x, (1, 0, 2))


# --------------------------------------------------------------------------
# geometry helpers (mld/utils/geometry2.py)
# --------------------------------------------------------------------------


def aa_to_quat(theta: Array) -> Array:
    """geometry2.py:33-54.  [M,3] -> [M,4] (w,x,y,z)."""
    norm = np.linalg.norm(theta + theta.dtype.type(1e-8), axis=1, keepdims=True)
    n = theta / norm
    half = norm * theta.dtype.type(0.5)
    return np.concatenate([np.cos(half), np.sin(half) * n], axis=1)


def quat_to_rotmat(quat: Array) -> Array:
    """geometry2.py:74-95."""
    q = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    R = np.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                  2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                  2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], axis=1)
    return R.reshape(-1, 3, 3)


def aa_to_rotmat(theta: Array) -> Array:
    """geometry2.py:56-72."""
    return quat_to_rotmat(aa_to_quat(theta))


def rot6d_to_rotmat(x: Array, rot6d_mode: str = "prohmr") -> Array:
    """geometry2.py:98-117 (F.normalize eps 1e-12)."""
    if rot6d_mode == "prohmr":
        x = x.reshape(-1, 2, 3).transpose(0, 2, 1)
    else:
        x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]

    def _nrm(v):
        return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), v.dtype.type(1e-12))

    b1 = _nrm(a1)
    b2 = _nrm(a2 - (b1 * a2).sum(axis=1, keepdims=True) * b1)
    b3 = np.cross(b1, b2)
    return np.stack([b1, b2, b3], axis=-1)


# --------------------------------------------------------------------------
# SMPL linear blend skinning (smplx 0.1.28 lbs restated) -- PARITY UNPINNED
# --------------------------------------------------------------------------

SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                         20, 21], dtype=np.int64)
# smplx vertex_ids['smpl'] in VertexJointSelector order: face (nose, reye, leye, rear, lear),
# feet (LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel), finger tips (l: thumb..pinky, r: thumb..pinky)
SMPL_EXTRA_VERTEX_IDS = np.array([332, 6260, 2800, 4071, 583,
                                  3216, 3226, 3387, 6617, 6624, 6787,
                                  2746, 2319, 2445, 2556, 2673,
                                  6191, 5782, 5905, 6016, 6133], dtype=np.int64)


def rodrigues(rot_vecs: Array) -> Array:
    """smplx.lbs.batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1-cos) K^2."""
    f = rot_vecs.dtype.type
    angle = np.linalg.norm(rot_vecs + f(1e-8), axis=1, keepdims=True)
    d = rot_vecs / angle
    c, s = np.cos(angle)[:, :, None], np.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = np.zeros_like(rx)
    K = np.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], axis=1).reshape(-1, 3, 3)
    eye = np.eye(3, dtype=rot_vecs.dtype)[None]
    return eye + s * K + (f(1) - c) * (K @ K)


def smpl_lbs(model: Dict[str, Array], betas: Array, global_orient: Array, body_pose: Array,
             transl: Optional[Array] = None, pose2rot: bool = True,
             return_verts: bool = True) -> Tuple[Array, Optional[Array]]:
    """smplx.SMPL.forward -> (joints [M,45,3], vertices [M,6890,3]).

    model: v_template [V,3], shapedirs [V,3,10], posedirs [207,V*3], J_regressor [24,V],
    lbs_weights [V,24], parents [24].  body_pose [M,69] (+global_orient [M,3]) axis-angle when
    pose2rot, else rotation matrices [M,23,3,3] / [M,1,3,3].
    """
    M = betas.shape[0]
    dt = betas.dtype
    vt, sd, pd = model["v_template"].astype(dt), model["shapedirs"].astype(dt), model["posedirs"].astype(dt)
    Jr, W = model["J_regressor"].astype(dt), model["lbs_weights"].astype(dt)
    parents = model.get("parents", SMPL_PARENTS)
    V = vt.shape[0]
    if pose2rot:
        full = np.concatenate([global_orient.reshape(M, 3), body_pose.reshape(M, -1)], axis=1)
        R = rodrigues(full.reshape(-1, 3)).reshape(M, -1, 3, 3)
    else:
        R = np.concatenate([global_orient.reshape(M, 1, 3, 3), body_pose.reshape(M, -1, 3, 3)], axis=1)
    J_n = R.shape[1]
    v_shaped = vt[None] + np.einsum("bl,mkl->bmk", betas, sd)
    J = np.einsum("bik,ji->bjk", v_shaped, Jr)
    feat = (R[:, 1:] - np.eye(3, dtype=dt)).reshape(M, -1)
    v_posed = v_shaped + (feat @ pd).reshape(M, V, 3)
    # kinematic chain (smplx.lbs.batch_rigid_transform)
    rel = J.copy()
    rel[:, 1:] -= J[:, parents[1:]]
    Tm = np.zeros((M, J_n, 4, 4), dt)
    Tm[:, :, :3, :3] = R
    Tm[:, :, :3, 3] = rel
    Tm[:, :, 3, 3] = 1
    G = [Tm[:, 0]]
    for i in range(1, J_n):
        G.append(G[parents[i]] @ Tm[:, i])
    G = np.stack(G, axis=1)
    joints = G[:, :, :3, 3].copy()
    Jh = np.concatenate([J, np.zeros((M, J_n, 1), dt)], axis=2)[..., None]
    A = G.copy()
    A[:, :, :, 3:4] -= G @ Jh
    verts = None
    need = np.arange(V) if return_verts else SMPL_EXTRA_VERTEX_IDS
    Tv = np.einsum("vj,bjk->bvk", W[need], A.reshape(M, J_n, 16)).reshape(M, -1, 4, 4)
    vp = v_posed[:, need]
    vh = np.concatenate([vp, np.ones((M, vp.shape[1], 1), dt)], axis=2)[..., None]
    vsel = (Tv @ vh)[:, :, :3, 0]
    if return_verts:
        verts = vsel
        extra = verts[:, SMPL_EXTRA_VERTEX_IDS]
    else:
        extra = vsel
    joints = np.concatenate([joints, extra], axis=1)
    if transl is not None:
        joints = joints + transl[:, None, :]
        if verts is not None:
            verts = verts + transl[:, None, :]
    return joints, verts


def make_synthetic_smpl(seed: int = 1234, V: int = 6890, dtype=np.float32) -> Dict[str, Array]:
    """Seeded SMPL-shaped model (true parents / extra-joint ids; random geometry).  No SMPL file is
    available offline."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vt = (rng.standard_normal((V, 3)) * [0.25, 0.6, 0.15]).astype(dtype)
    sd = (rng.standard_normal((V, 3, 10)) * 0.01).astype(dtype)
    pd = (rng.standard_normal((207, V * 3)) * 0.002).astype(dtype)
    Jr = rng.random((24, V)) ** 8
    Jr = (Jr / Jr.sum(axis=1, keepdims=True)).astype(dtype)
    W = rng.random((V, 24)) ** 12
    W = (W / W.sum(axis=1, keepdims=True)).astype(dtype)
    return dict(v_template=vt, shapedirs=sd, posedirs=pd, J_regressor=Jr, lbs_weights=W,
                parents=SMPL_PARENTS.copy())


# --------------------------------------------------------------------------
# ResnetPointnet (EgoHMR/models/respointnet.py)
# --------------------------------------------------------------------------


def _resblock(P: Params, pre: str, x: Array) -> Array:
    """ResnetBlockFC.forward, respointnet.py:88-97."""
    net = linear(relu(x), P[pre + "fc_0.weight"], P[pre + "fc_0.bias"])
    dx = linear(relu(net), P[pre + "fc_1.weight"], P[pre + "fc_1.bias"])
    xs = linear(x, P[pre + "shortcut.weight"]) if (pre + "shortcut.weight") in P else x
    return xs + dx


def pointnet_forward(P: Params, p: Array) -> Array:
    """ResnetPointnet.forward, respointnet.py:33-59.  p [B,n_pts,3] -> [B,out_dim]."""
    net = linear(p, P["fc_pos_0.weight"], P["fc_pos_0.bias"])
    net = _resblock(P, "block_0.", net)
    for i in (1, 2, 3):
        pooled = np.broadcast_to(net.max(axis=1, keepdims=True), net.shape)
        net = _resblock(P, f"block_{i}.", np.concatenate([net, pooled], axis=2))
    net = net.max(axis=1)
    return linear(relu(net), P["fc_c.weight"], P["fc_c.bias"])


# --------------------------------------------------------------------------
# metrics / losses (definitions only)
# --------------------------------------------------------------------------


def mpjpe_mm(pred: Array, gt: Array) -> float:
    """Mean per-joint position error x1000 (metrics/compute.py:472-473 core definition,
    without the alignment steps): joints [..., J, 3] in metres."""
    return float(np.linalg.norm(pred - gt, axis=-1).mean() * 1000.0)


def _quat_matrix(q: Array) -> Array:
    """transformations.quaternion_matrix (w,x,y,z), the function behind get_root_matrix, compute.py:286-295."""
    q = np.array(q, dtype=np.float64)
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(3)
    q *= np.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([[1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0]],
                     [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0]],
                     [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2]]])


def ego_metrics(jts_pred: Array, jts_ref: Array, quat_pred: Array, quat_ref: Array, lengths, split: str = "test") -> dict:
    """ComputeMetrics.update + compute, metrics/compute.py:349-580,184-232, as the reference's per-sequence loops:
    first-frame head alignment (:364-373), align_root (:399), per sequence MPJPE / root error (:470-473), acceleration
    error (:243-271,474), head-orientation error (:338-346,469) and the split-dependent inclusion rule (:488-517,
    :567-576).  jts [B,T,24,3] metres; quats [B*T,4] (w,x,y,z)."""
    B, T = jts_ref.shape[:2]
    ref = jts_ref - jts_ref[:, 0:1, 15:16, :]
    pred = jts_pred - jts_pred[:, 0:1, 15:16, :]
    pel_g, pel_p = ref[:, :, [0]], pred[:, :, [0]]
    ref, pred = ref - pel_g, pred - pel_p
    qg, qp = quat_ref.reshape(B, T, 4), quat_pred.reshape(B, T, 4)
    sums = {k: 0.0 for k in ("MPJPE", "ROOT_ERROR", "ACCL", "HEAD_ORIENTATION_ERROR")}
    cnt = dict.fromkeys(sums, 0)
    for b in range(B):
        L = int(lengths[b])
        g, p = ref[b, :L].reshape(-1, 24, 3), pred[b, :L].reshape(-1, 24, 3)
        head = 0.0
        for t in range(L):
            head += np.linalg.norm(np.identity(3) - _quat_matrix(qg[b, t]) @ np.linalg.inv(_quat_matrix(qp[b, t])), "fro")
        head /= L
        root = np.linalg.norm(pel_g[b, :L].reshape(-1, 3) - pel_p[b, :L].reshape(-1, 3), axis=1).mean() * 1000
        mp = np.linalg.norm(p - g, axis=-1).mean() * 1000
        ag, ap = g[:-2] - 2 * g[1:-1] + g[2:], p[:-2] - 2 * p[1:-1] + p[2:]
        accl = np.mean(np.linalg.norm(ap - ag, axis=2), axis=1)
        if not np.mean(accl) > 0:
            continue
        if split == "test":
            if head < 0.9 and root < 300:
                for k, v in (("MPJPE", mp), ("ROOT_ERROR", root), ("ACCL", np.mean(accl) * 1000), ("HEAD_ORIENTATION_ERROR", head)):
                    sums[k] += v
                    cnt[k] += 1
        else:
            for k, v in (("MPJPE", mp), ("ROOT_ERROR", root)):
                sums[k] += v
                cnt[k] += 1
    return {k: sums[k] / max(cnt[k], 1) for k in sums} | {"count_seq": float(cnt["MPJPE"])}


def renorm(x: Array, mean: Array, std: Array) -> Array:
    """EgoBodyDataModule.renorm, mld/data/EgoBody.py:151-157."""
    n = x.shape[-1]
    return x * std[..., :n] + mean[..., :n]


def cast_params(P: Params, dtype) -> Params:
    return {k: (v.astype(dtype) if np.issubdtype(v.dtype, np.floating) else v) for k, v in P.items()}
