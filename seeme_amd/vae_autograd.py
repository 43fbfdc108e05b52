"""Differentiable twins for STAGE-1 training (``TRAIN.STAGE: vae``): the motion VAE and the SMPL joint regressor in
plain PyTorch ops on the same parameters / buffers as the HIP modules, so that ``MLD.train_vae_forward`` has a backward
(PyTorch-ROCm autograd on the device).  Since round 2 the training path is the hand-written HIP forward / backward of
``vae_train.py`` and ``smpl._JointsAA``; these twins are its fallback (shapes the kernels do not take,
``TRAIN.HIP_VAE_BACKWARD: false``) and its test oracle.  Checked against the HIP forward in
tests/test_gpu_parity.py::test_vae_autograd_twin_matches_hip.

Reference: mld_vae.py:128-256, cross_attention.py:41-147,281-367 (dropout at the reference's sites in training mode);
smplx lbs: shape blend -> joint regression -> Rodrigues -> pose blend is irrelevant for joints ->
kinematic chain -> posed joints + translation.
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn.functional as F


class Dropper:
    """The dropout of the reference's layers (cross_attention.py:264-273,324-337).  ``masks``: injected keep-masks by site name
    (tests: the very masks the HIP path drew); otherwise F.dropout with probability p when ``training``; identity in eval."""

    def __init__(self, p: float = 0.0, training: bool = False, masks=None):
        self.p, self.training, self.masks = float(p), bool(training), masks

    def __call__(self, x, site: str):
        if self.masks is not None:
            m = self.masks[site].reshape(x.shape).to(x.dtype)
            return x * m * (1.0 / (1.0 - self.p))
        if self.training and self.p > 0:
            return F.dropout(x, self.p, True)
        return x


_NO_DROP = Dropper()


def _mha(lp, q_in, k_in, v_in, nhead: int, key_padding_mask=None, drop=_NO_DROP, site=""):
    """nn.MultiheadAttention arithmetic, batch-first [B,S,D]; key_padding_mask bool [B,Sk], True = ignore."""
    w, b = lp.in_proj_weight, lp.in_proj_bias
    D = q_in.shape[-1]
    q = F.linear(q_in, w[:D], b[:D])
    k = F.linear(k_in, w[D:2 * D], b[D:2 * D])
    v = F.linear(v_in, w[2 * D:], b[2 * D:])
    B, Sq, _ = q.shape
    hd = D // nhead
    q = q.view(B, Sq, nhead, hd).transpose(1, 2)
    k = k.view(B, -1, nhead, hd).transpose(1, 2)
    v = v.view(B, -1, nhead, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = (drop(torch.softmax(s, dim=-1), site) @ v).transpose(1, 2).reshape(B, Sq, D)
    return F.linear(o, lp.out_proj.weight, lp.out_proj.bias)


def _ln(x, n):
    return F.layer_norm(x, (x.shape[-1],), n.weight, n.bias)


def _enc_layer(lp, x, nhead, kpm, drop=_NO_DROP, l=0):               # forward_post, cross_attention.py:281-294
    x = _ln(x + drop(_mha(lp.self_attn, x, x, x, nhead, kpm, drop, f"{l}.mP"), f"{l}.m1"), lp.norm1)
    h = drop(F.gelu(F.linear(x, lp.linear1.weight, lp.linear1.bias)), f"{l}.mh")
    return _ln(x + drop(F.linear(h, lp.linear2.weight, lp.linear2.bias), f"{l}.m2"), lp.norm2)


def _dec_layer(lp, x, mem, nhead, kpm, drop=_NO_DROP, l=0):          # forward_post, cross_attention.py:345-367
    x = _ln(x + drop(_mha(lp.self_attn, x, x, x, nhead, kpm, drop, f"{l}.mP"), f"{l}.m1"), lp.norm1)
    x = _ln(x + drop(_mha(lp.multihead_attn, x, mem, mem, nhead, None, drop, f"{l}.mw"), f"{l}.mc"), lp.norm2)
    h = drop(F.gelu(F.linear(x, lp.linear1.weight, lp.linear1.bias)), f"{l}.mh")
    return _ln(x + drop(F.linear(h, lp.linear2.weight, lp.linear2.bias), f"{l}.m2"), lp.norm3)


def _skip_stack(stack, x, layer_fn):                                  # cross_attention.py:46-65,118-147; layer_fn(lp, x, index)
    nb = len(stack.input_blocks)
    xs = []
    for i in range(nb):
        x = layer_fn(stack.input_blocks[i], x, i)
        xs.append(x)
    x = layer_fn(stack.middle_block, x, nb)
    for i in range(nb):
        x = F.linear(torch.cat([x, xs.pop()], dim=-1), stack.linear_blocks[i].weight, stack.linear_blocks[i].bias)
        x = layer_fn(stack.output_blocks[i], x, nb + 1 + i)
    return _ln(x, stack.norm)


def _dropper(vae, masks):
    p = float(vae.encoder.input_blocks[0].self_attn.dropout)
    return Dropper(p, vae.training, masks)


def _mask(lengths: List[int], device):
    lens = torch.as_tensor(lengths, device=device)
    return torch.arange(int(max(lengths)), device=device)[None, :] < lens[:, None]


def vae_encode_torch(vae, features: torch.Tensor, lengths: List[int], masks=None):
    """features [B,T,F] -> (mu [1,B,256], std [1,B,256]) (mld_vae.py:128-193).  Dropout as the module's mode says; `masks`:
    injected keep-masks {"<layer>.<site>": tensor} (tests)."""
    B = features.shape[0]
    nhead = vae.encoder.input_blocks[0].self_attn.num_heads
    mask = _mask(lengths, features.device)
    x = F.linear(features, vae.skel_embedding.weight, vae.skel_embedding.bias)
    tok = vae.global_motion_token[None].expand(B, -1, -1)
    aug = torch.cat([torch.ones(B, tok.shape[1], dtype=torch.bool, device=x.device), mask], dim=1)
    xseq = torch.cat([tok, x], dim=1)
    xseq = xseq + vae.query_pos_encoder.pe[: xseq.shape[1], 0][None]
    drop = _dropper(vae, masks)
    out = _skip_stack(vae.encoder, xseq, lambda lp, h, l: _enc_layer(lp, h, nhead, ~aug, drop, l))
    mu, logvar = out[:, 0], out[:, 1]
    return mu[None], logvar.exp().pow(0.5)[None]


def vae_decode_torch(vae, z: torch.Tensor, lengths: List[int], masks=None):
    """z [1,B,256] -> feats [B,T,F] (mld_vae.py:195-256, arch encoder_decoder; padded frames not zeroed, :253)."""
    nhead = vae.decoder.input_blocks[0].self_attn.num_heads
    mask = _mask(lengths, z.device)
    B, T = mask.shape
    q = vae.query_pos_decoder.pe[:T, 0][None].expand(B, -1, -1)
    mem = z.permute(1, 0, 2)
    drop = _dropper(vae, masks)
    out = _skip_stack(vae.decoder, q, lambda lp, h, l: _dec_layer(lp, h, mem, nhead, ~mask, drop, l))
    return F.linear(out, vae.final_layer.weight, vae.final_layer.bias)


def _rodrigues(aa: torch.Tensor) -> torch.Tensor:
    """Axis-angle [M,3] -> rotation matrices [M,3,3] (smplx batch_rodrigues: angle = ||aa + 1e-8||)."""
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    d = aa / angle
    c, s = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(-1, 3, 3)
    eye = torch.eye(3, device=aa.device, dtype=aa.dtype)[None]
    return eye + s * K + (1 - c) * (K @ K)


def smpl_joints_torch(smpl, betas: torch.Tensor, pose_aa: torch.Tensor, transl: torch.Tensor = None) -> torch.Tensor:
    """The 24 posed SMPL joints [M,24,3], differentiable w.r.t. pose and translation (betas too).
    pose_aa [M,72] = global_orient | body_pose."""
    M = pose_aa.shape[0]
    # rest joints = J_regressor (v_template + shapedirs betas), with the regressor folded in
    cache = getattr(smpl, "_joint_fold", None)
    if cache is None or cache[0].device != betas.device:
        cache = (smpl.J_regressor @ smpl.v_template, torch.einsum("jv,vkl->jkl", smpl.J_regressor, smpl.shapedirs))
        smpl._joint_fold = cache
    J = cache[0][None] + torch.einsum("jkl,bl->bjk", cache[1], betas)                # [M,24,3]
    R = _rodrigues(pose_aa.reshape(-1, 3)).view(M, 24, 3, 3)
    parents = smpl.parents.tolist()
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    Tm = torch.cat([torch.cat([R, rel[..., None]], dim=-1),
                    torch.tensor([0.0, 0.0, 0.0, 1.0], device=J.device, dtype=J.dtype).expand(M, 24, 1, 4)], dim=-2)   # [M,24,4,4]
    chain = [Tm[:, 0]]
    for i in range(1, 24):
        chain.append(chain[parents[i]] @ Tm[:, i])
    posed = torch.stack(chain, dim=1)[:, :, :3, 3]
    return posed if transl is None else posed + transl[:, None, :]
