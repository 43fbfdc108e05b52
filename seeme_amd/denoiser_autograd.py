"""Differentiable twin of the denoiser forward: the fallback of the stage-2 training step for shapes the hand-written HIP
backward does not take (more than one head, TRAIN.HIP_BACKWARD false) and its test oracle (PyTorch-ROCm autograd).

Same parameters (it reads them from the MldDenoiser instance), same arithmetic as the HIP kernel: the
token-0 pruning of SURVEY.md App. E3 (only the latent token needs Q / out_proj / FFN of the sa_block) and
the linear cross-attention in its ``sum_n (q.k_n) v_n`` form (E2).  Checked against the HIP forward in
tests/test_gpu_parity.py::test_autograd_twin_matches_hip.  Inference never routes through here.

Reference: mld_denoiser.py:151-244, mdiff_transformer.py:152-163, 219-254, 286-304.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .mld_denoiser import timestep_features


# byte offsets of a layer's dropout sites in the HIP path's mask block (csrc/den_train.h DM_*), and their widths
_SITES = {"P": (0, 8), "1": (8, 256), "H": (264, 1024), "2": (1288, 256), "C": (1544, 256), "F": (1800, 128), "O": (1928, 256)}
_DM_LAYER = 2192


class _Drop:
    """Dropout at the MD layer's nn.Dropout sites (mdiff_transformer.py:137-165,241-254; cross_attention.py:264-273): with
    `masks` (uint8 [B, 5 * 2192], the HIP path's own block) those very keep-masks, else F.dropout in training mode."""

    def __init__(self, p, training, masks):
        self.p, self.training, self.masks, self.layer = float(p), bool(training), masks, 0

    def __call__(self, x, site):
        if self.masks is not None:
            o, w = _SITES[site]
            n = x.shape[-1]
            m = self.masks[:, self.layer * _DM_LAYER + o: self.layer * _DM_LAYER + o + n].to(x.dtype)
            return x * m.reshape(x.shape) * (1.0 / (1.0 - self.p))
        if self.training and self.p > 0:
            return F.dropout(x, self.p, True)
        return x


def _stylization(p, h, emb, drop, site):
    eo = F.linear(F.silu(emb), p.emb_layers[1].weight, p.emb_layers[1].bias).unsqueeze(1)
    scale, shift = torch.chunk(eo, 2, dim=2)
    h = F.layer_norm(h, (h.shape[-1],), p.norm.weight, p.norm.bias) * (1 + scale) + shift
    return F.linear(drop(F.silu(h), site), p.out_layers[2].weight, p.out_layers[2].bias)


def _layer(blk, x, xf, emb, H, drop):
    """x [B,1,D] latent token, xf [B,N,D], emb [B,D]."""
    B, _, D = x.shape
    sa, ca, ffn = blk.sa_block, blk.ca_block, blk.ffn
    w, b = sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias
    seq = torch.cat([x, xf, emb.unsqueeze(1)], dim=1)                     # [B,S,D]  (mdiff_transformer.py:295)
    q = F.linear(x, w[:D], b[:D])                                          # only token 0 is kept (:297)
    k = F.linear(seq, w[D:2 * D], b[D:2 * D])
    v = F.linear(seq, w[2 * D:], b[2 * D:])
    hd = D // H
    qh = q.view(B, 1, H, hd).transpose(1, 2)
    kh = k.view(B, -1, H, hd).transpose(1, 2)
    vh = v.view(B, -1, H, hd).transpose(1, 2)
    pw = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1)          # [B,H,1,S]: token 0's row
    att = drop(pw, "P") @ vh                                                        # (injected masks: one head)
    att = att.transpose(1, 2).reshape(B, 1, D)
    x = F.layer_norm(x + drop(F.linear(att, sa.self_attn.out_proj.weight, sa.self_attn.out_proj.bias), "1"), (D,),
                     sa.norm1.weight, sa.norm1.bias)
    h = drop(F.relu(F.linear(x, sa.linear1.weight, sa.linear1.bias)), "H")
    x = F.layer_norm(x + drop(F.linear(h, sa.linear2.weight, sa.linear2.bias), "2"), (D,), sa.norm2.weight, sa.norm2.bias)
    # linear cross-attention (:219-239)
    qc = F.linear(F.layer_norm(x, (D,), ca.norm.weight, ca.norm.bias), ca.query.weight, ca.query.bias)
    xfn = F.layer_norm(xf, (D,), ca.text_norm.weight, ca.text_norm.bias)
    kc = F.linear(xfn, ca.key.weight, ca.key.bias)
    vc = F.linear(xfn, ca.value.weight, ca.value.bias)
    N = xf.shape[1]
    qc = torch.softmax(qc.view(B, 1, H, hd), dim=-1)
    kc = torch.softmax(kc.view(B, N, H, hd), dim=1)
    vc = vc.view(B, N, H, hd)
    attention = torch.einsum("bnhd,bnhl->bhdl", kc, vc)
    y = torch.einsum("bnhd,bhdl->bnhl", qc, attention).reshape(B, 1, D)
    x = x + _stylization(ca.proj_out, y, emb, drop, "C")
    y = F.linear(drop(F.gelu(F.linear(x, ffn.linear1.weight, ffn.linear1.bias)), "F"), ffn.linear2.weight, ffn.linear2.bias)
    return x + _stylization(ffn.proj_out, y, emb, drop, "O")


def denoiser_forward_torch(den, sample, timestep, encoder_hidden_states, masks=None):
    """sample [B,1,256]; timestep 0-d or [B]; encoder_hidden_states [N,B,256] (seq-first) -> [B,1,256].  Dropout as the module's
    mode says (training: the reference's sites); `masks`: the HIP path's keep-mask block, injected (tests)."""
    B = sample.shape[0]
    t = torch.as_tensor(timestep, device=sample.device)
    t = t.expand(B) if t.dim() == 0 else t
    feat = timestep_features(t, den.text_encoded_dim, den.flip_sin_to_cos, den.freq_shift).to(sample.dtype)
    te = den.time_embedding
    emb = F.linear(F.silu(F.linear(feat, te.linear_1.weight, te.linear_1.bias)), te.linear_2.weight, te.linear_2.bias)
    xf = encoder_hidden_states.permute(1, 0, 2)
    x = sample + den.query_pos.pe[:1, 0][None]
    enc = den.encoder
    blocks = enc.blocks()
    xs = []
    nb = (len(blocks) - 1) // 2
    drop = _Drop(blocks[0].sa_block.self_attn.dropout, den.training, masks)

    def layer(i, x):
        drop.layer = i
        return _layer(blocks[i], x, xf, emb, den.num_heads, drop)
    for i in range(nb):
        x = layer(i, x)
        xs.append(x)
    x = layer(nb, x)
    for i in range(nb):
        x = torch.cat([x, xs.pop()], dim=-1)
        x = F.linear(x, enc.linear_blocks[i].weight, enc.linear_blocks[i].bias)
        x = layer(nb + 1 + i, x)
    return F.layer_norm(x, (x.shape[-1],), enc.norm.weight, enc.norm.bias)
