"""CPU oracle, PyTorch-on-CPU edition: the same restatement as ``mld_oracle.py`` (numpy), statement for statement,
on ``torch`` CPU tensors -- the "pure-torch fp32 restatement run on the node's host cores" that SURVEY.md section 8(d)
names as the CPU baseline of the sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``seeme_amd/`` may import this module; only ``tests/`` and the
``cpu_baseline`` leg of ``bench.py`` use it, as the checker / reported baseline.  It covers exactly what that leg
times: VAE encode, the denoiser forward, the DDIM/DDPM reverse loop and VAE decode.

Pinning: checked against the same reference-generated fixtures as the numpy oracle (``tests/golden/*.npz``, made by
``tests/golden/make_golden.py`` from the reference's own PyTorch modules) and against the numpy oracle itself in
``tests/test_oracle_golden.py``.  The scheduler arithmetic (``diffusers``, unpinned upstream) is shared with
``mld_oracle.py`` and stays **parity unpinned** as documented there.

Reference lines are cited per function (paths relative to the upstream repository root).  Batch-first inside.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import mld_oracle as _np_oracle

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def to_torch(P_np: Dict[str, np.ndarray]) -> Params:
    """State-dict arrays (numpy, reference key names) -> CPU tensors."""
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in P_np.items()}


def _ln(x: Tensor, P: Params, pre: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], 1e-5)


def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, P: Params, pre: str, nhead: int,
        key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """torch.nn.MultiheadAttention forward (eval), batch-first; key_padding_mask bool [B,Sk], True = ignore
    (cross_attention.py:264,286-289)."""
    w, b = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    D = q_in.shape[-1]
    q = F.linear(q_in, w[:D], b[:D])
    k = F.linear(k_in, w[D:2 * D], b[D:2 * D])
    v = F.linear(v_in, w[2 * D:], b[2 * D:])
    B, Sq, _ = q.shape
    hd = D // nhead
    q = q.view(B, Sq, nhead, hd).transpose(1, 2)
    k = k.view(B, -1, nhead, hd).transpose(1, 2)
    v = v.view(B, -1, nhead, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Sq, D)
    return F.linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


_ACT = {"gelu": F.gelu, "relu": F.relu}


def encoder_layer_post(x: Tensor, P: Params, pre: str, nhead: int, act: str, kpm: Optional[Tensor]) -> Tensor:
    """TransformerEncoderLayer.forward_post, cross_attention.py:281-294 (same body as mdiff_transformer.py:54-67)."""
    x = _ln(x + mha(x, x, x, P, pre + "self_attn.", nhead, kpm), P, pre + "norm1.")
    h = _ACT[act](F.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"]))
    return _ln(x + F.linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"]), P, pre + "norm2.")


def decoder_layer_post(x: Tensor, mem: Tensor, P: Params, pre: str, nhead: int, act: str, kpm: Optional[Tensor]) -> Tensor:
    """TransformerDecoderLayer.forward_post, cross_attention.py:345-367."""
    x = _ln(x + mha(x, x, x, P, pre + "self_attn.", nhead, kpm), P, pre + "norm1.")
    x = _ln(x + mha(x, mem, mem, P, pre + "multihead_attn.", nhead, None), P, pre + "norm2.")
    h = _ACT[act](F.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"]))
    return _ln(x + F.linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"]), P, pre + "norm3.")


def _skip_stack(x: Tensor, P: Params, pre: str, num_layers: int, layer_fn) -> Tensor:
    """SkipTransformerEncoder/Decoder.forward, cross_attention.py:46-65,118-147."""
    nb = (num_layers - 1) // 2
    xs = []
    for i in range(nb):
        x = layer_fn(x, f"{pre}input_blocks.{i}.")
        xs.append(x)
    x = layer_fn(x, f"{pre}middle_block.")
    for i in range(nb):
        x = F.linear(torch.cat([x, xs.pop()], dim=-1), P[f"{pre}linear_blocks.{i}.weight"], P[f"{pre}linear_blocks.{i}.bias"])
        x = layer_fn(x, f"{pre}output_blocks.{i}.")
    return _ln(x, P, pre + "norm.")


VAE_LAYERS, VAE_HEADS, VAE_ACT = _np_oracle.VAE_LAYERS, _np_oracle.VAE_HEADS, _np_oracle.VAE_ACT   # mld_vae.py:51-53


def _mask(lengths: Sequence[int]) -> Tensor:
    """lengths_to_mask, temos_utils.py:10-17."""
    return torch.arange(int(max(lengths)))[None, :] < torch.as_tensor(list(lengths))[:, None]


@torch.no_grad()
def vae_encode(P: Params, features: Tensor, lengths: Sequence[int]) -> Tuple[Tensor, Tensor]:
    """MldVae.encode, mld_vae.py:128-193.  features [B,T,F] -> (mu [1,B,256], std [1,B,256])."""
    B = features.shape[0]
    mask = _mask(lengths)                                                          # :143
    x = F.linear(features, P["skel_embedding.weight"], P["skel_embedding.bias"])   # :147
    tok = P["global_motion_token"][None].expand(B, -1, -1)                         # :154
    aug = torch.cat([torch.ones(B, tok.shape[1], dtype=torch.bool), mask], dim=1)  # :157-160
    xseq = torch.cat([tok, x], dim=1)                                              # :164
    xseq = xseq + P["query_pos_encoder.pe"][: xseq.shape[1], 0][None]              # :171
    kpm = ~aug
    out = _skip_stack(xseq, P, "encoder.", VAE_LAYERS,
                      lambda h, pre: encoder_layer_post(h, P, pre, VAE_HEADS, VAE_ACT, kpm))
    mu, logvar = out[:, 0], out[:, 1]                                              # :186-187
    return mu[None], logvar.exp().pow(0.5)[None]                                   # :190


@torch.no_grad()
def vae_decode(P: Params, z: Tensor, lengths: Sequence[int]) -> Tensor:
    """MldVae.decode, arch 'encoder_decoder', mld_vae.py:195-256.  z [1,B,256] -> [B,T,F]."""
    mask = _mask(lengths)
    B, T = mask.shape
    q = P["query_pos_decoder.pe"][:T, 0][None].expand(B, -1, -1)                   # :198,232
    mem = z.permute(1, 0, 2)
    kpm = ~mask
    out = _skip_stack(q, P, "decoder.", VAE_LAYERS,
                      lambda h, pre: decoder_layer_post(h, mem, P, pre, VAE_HEADS, VAE_ACT, kpm))
    return F.linear(out, P["final_layer.weight"], P["final_layer.bias"])          # :251 (pad not zeroed, :253)


def time_embedding(P: Params, feat: Tensor, pre: str = "time_embedding.") -> Tensor:
    """TimestepEmbedding.forward, tools/embeddings.py:298-305."""
    return F.linear(F.silu(F.linear(feat, P[pre + "linear_1.weight"], P[pre + "linear_1.bias"])),
                    P[pre + "linear_2.weight"], P[pre + "linear_2.bias"])


def stylization(P: Params, pre: str, h: Tensor, emb: Tensor) -> Tensor:
    """StylizationBlock.forward, mdiff_transformer.py:152-163.  h [B,T,D], emb [B,D]."""
    eo = F.linear(F.silu(emb), P[pre + "emb_layers.1.weight"], P[pre + "emb_layers.1.bias"])[:, None, :]
    D = h.shape[-1]
    h = _ln(h, P, pre + "norm.") * (1 + eo[..., :D]) + eo[..., D:]
    return F.linear(F.silu(h), P[pre + "out_layers.2.weight"], P[pre + "out_layers.2.bias"])


def linear_cross_attention(P: Params, pre: str, x: Tensor, xf: Tensor, emb: Tensor, nhead: int) -> Tensor:
    """LinearTemporalCrossAttention.forward, mdiff_transformer.py:219-239.  x [B,T,D], xf [B,N,D], emb [B,D]."""
    B, T, D = x.shape
    N = xf.shape[1]
    q = F.linear(_ln(x, P, pre + "norm."), P[pre + "query.weight"], P[pre + "query.bias"])
    xfn = _ln(xf, P, pre + "text_norm.")
    k = F.linear(xfn, P[pre + "key.weight"], P[pre + "key.bias"])
    q = torch.softmax(q.view(B, T, nhead, -1), dim=-1)
    k = torch.softmax(k.view(B, N, nhead, -1), dim=1)
    v = F.linear(xfn, P[pre + "value.weight"], P[pre + "value.bias"]).view(B, N, nhead, -1)
    att = torch.einsum("bnhd,bnhl->bhdl", k, v)
    y = torch.einsum("bnhd,bhdl->bnhl", q, att).reshape(B, T, D)
    return x + stylization(P, pre + "proj_out.", y, emb)


def ffn_stylized(P: Params, pre: str, x: Tensor, emb: Tensor) -> Tensor:
    """FFN.forward, mdiff_transformer.py:251-254."""
    y = F.linear(F.gelu(F.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"])),
                 P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return x + stylization(P, pre + "proj_out.", y, emb)


def md_layer(P: Params, pre: str, x: Tensor, xf: Tensor, emb: Tensor, nhead: int) -> Tensor:
    """LinearTemporalDiffusionTransformerDecoderLayer.forward, mdiff_transformer.py:286-304."""
    L = x.shape[1]
    seq = torch.cat([x, xf, emb[:, None, :]], dim=1)                               # :295
    seq = encoder_layer_post(seq, P, pre + "sa_block.", nhead, "relu", None)       # :296 (ff 1024 relu :279)
    x = seq[:, :L]                                                                 # :297
    x = linear_cross_attention(P, pre + "ca_block.", x, xf, emb, nhead)            # :300
    return ffn_stylized(P, pre + "ffn.", x, emb)                                   # :301


@torch.no_grad()
def denoiser_forward(P: Params, sample: Tensor, timestep, cond: Tensor, num_layers: int = 5, nhead: int = 1,
                     flip_sin_to_cos: bool = True, freq_shift: float = 0.0) -> Tensor:
    """MldDenoiser.forward, mld_denoiser.py:151-244 (arch trans_enc, MD_TRANS, SKIP_CONNECT, DIFF_PE_TYPE 'mld').
    sample [B,L,D]; timestep scalar or [B]; cond seq-first [N,B,D].  Returns [B,L,D]."""
    B, L, D = sample.shape
    t = np.broadcast_to(np.asarray(timestep), (B,))                                # :167
    feat = torch.from_numpy(_np_oracle.timestep_features(t, D, flip_sin_to_cos, freq_shift, dtype=np.float32))
    emb = time_embedding(P, feat.to(sample.dtype))                                 # :168-171
    xf = cond.permute(1, 0, 2)
    x = sample + P["query_pos.pe"][:L, 0][None]                                    # :210
    out = _skip_stack(x, P, "encoder.", num_layers, lambda h, pre: md_layer(P, pre, h, xf, emb, nhead))   # :212
    return out[:, :L]                                                              # :222,242


@torch.no_grad()
def diffusion_reverse(P: Params, cond_bf: Tensor, latents: Tensor, num_inference_steps: int = 50, eta: float = 0.0,
                      guidance_scale: float = 1.0, scheduler: str = "ddim", step_noise: Optional[Tensor] = None,
                      **den_kw) -> Tensor:
    """MLD._diffusion_reverse, mld.py:432-511, RNG draws injected; scheduler arithmetic = mld_oracle.ddim_step /
    ddpm_step on the same float32 coefficients (evaluated through numpy views of the tensors' storage).
    cond_bf [B or 2B,N,D] (uncond first when CFG); latents [B,1,D].  Returns [1,B,D]."""
    acp = _np_oracle.alphas_cumprod(_np_oracle.make_betas())
    cfg = guidance_scale > 1.0
    ts = _np_oracle.ddim_timesteps(num_inference_steps) if scheduler == "ddim" else np.arange(num_inference_steps)[::-1].astype(np.int64)
    x = latents
    cond_sf = cond_bf.permute(1, 0, 2)
    for i, t in enumerate(ts):
        xin = torch.cat([x, x], dim=0) if cfg else x
        eps = denoiser_forward(P, xin, int(t), cond_sf, **den_kw)
        if cfg:
            e_u, e_c = eps.chunk(2, dim=0)
            eps = e_u + guidance_scale * (e_c - e_u)
        nz = None if step_noise is None else step_noise[i].numpy()
        if scheduler == "ddim":
            x = torch.from_numpy(_np_oracle.ddim_step(acp, eps.numpy(), int(t), x.numpy(), num_inference_steps, eta, nz))
        else:
            x = torch.from_numpy(_np_oracle.ddpm_step(acp, eps.numpy(), int(t), x.numpy(), nz))
    return x.permute(1, 0, 2)
