"""MldVae -- drop-in for ``mld.models.architectures.mld_vae.MldVae`` (reference mld_vae.py:33-256).

Same constructor arguments, same ``encode`` / ``decode`` signatures and tensor layouts, same
``state_dict`` keys and shapes so reference checkpoints load with
``strict=True``.  The parameters live in ordinary ``torch.nn`` containers; the arithmetic does NOT:
``encode``/``decode`` hand device pointers to libseeme_hip.so (seeme_vae_encode / seeme_vae_decode).

Select it from the unchanged YAML by pointing ``target:`` at
``seeme_amd.mld_vae.MldVae`` (configs/modules/motion_vae.yaml:3 in the reference).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L


class _PositionEmbeddingLearned1D(nn.Module):
    """Parameter holder for position_encoding.py:138-159 (pe [500,1,256], uniform(0,1) init)."""

    def __init__(self, d_model: int, max_len: int = 500):
        super().__init__()
        self.pe = nn.Parameter(torch.zeros(max_len, 1, d_model))
        nn.init.uniform_(self.pe)


class _EncoderLayerParams(nn.Module):
    """Parameters of TransformerEncoderLayer (cross_attention.py:258-276)."""

    def __init__(self, d_model, nhead, ff, dropout):
        super().__init__()
        self.d_model = d_model
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, ff)
        self.linear2 = nn.Linear(ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _DecoderLayerParams(nn.Module):
    """Parameters of TransformerDecoderLayer (cross_attention.py:319-340)."""

    def __init__(self, d_model, nhead, ff, dropout):
        super().__init__()
        self.d_model = d_model
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, ff)
        self.linear2 = nn.Linear(ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)


class _SkipStackParams(nn.Module):
    """Parameters of SkipTransformerEncoder/Decoder (cross_attention.py:18-39, 88-109)."""

    def __init__(self, make_layer, num_layers: int, d_model: int):
        super().__init__()
        assert num_layers % 2 == 1
        nb = (num_layers - 1) // 2
        self.d_model = d_model
        self.num_layers = num_layers
        self.input_blocks = nn.ModuleList([make_layer() for _ in range(nb)])
        self.middle_block = make_layer()
        self.output_blocks = nn.ModuleList([make_layer() for _ in range(nb)])
        self.linear_blocks = nn.ModuleList([nn.Linear(2 * d_model, d_model) for _ in range(nb)])
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():  # _reset_parameters, cross_attention.py:36-39
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def blocks(self):
        return list(self.input_blocks) + [self.middle_block] + list(self.output_blocks)


def _param_fingerprint(module: nn.Module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


class MldVae(nn.Module):

    def __init__(self,
                 ablation,
                 nfeats: int,
                 latent_dim: list = [1, 256],
                 ff_size: int = 1024,
                 num_layers: int = 9,
                 num_heads: int = 4,
                 dropout: float = 0.1,
                 arch: str = "all_encoder",
                 normalize_before: bool = False,
                 activation: str = "gelu",
                 position_embedding: str = "learned",
                 precision: str = "fp32",
                 **kwargs) -> None:
        super().__init__()
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' (fp32-exact MFMA, parity path) or 'fp16' (fp16 MFMA operands)")
        self.precision = precision
        # The reference overrides the configured sizes (mld_vae.py:51-53); checkpoints depend on it.
        num_layers, num_heads, ff_size = 5, 1, 128
        self.latent_size = latent_dim[0]
        self.latent_dim = latent_dim[-1]
        self.arch = arch
        self.mlp_dist = ablation.MLP_DIST
        self.pe_type = ablation.PE_TYPE
        self.nfeats = nfeats
        self.ff_size = ff_size
        if self.pe_type != "mld":
            if self.pe_type == "actor":
                raise NotImplementedError("PE_TYPE 'actor' is outside the accelerated path (all SEE-ME configs use 'mld')")
            raise ValueError("Not Support PE type")               # mld_vae.py:74
        if position_embedding not in ("v3", "learned"):
            raise NotImplementedError("only the learned 1D position embedding is on the accelerated path")
        if self.arch != "encoder_decoder":
            if self.arch == "all_encoder":
                raise NotImplementedError("arch 'all_encoder' is outside the accelerated path (motion_vae.yaml uses encoder_decoder)")
            raise ValueError("Not support architecture!")          # mld_vae.py:105
        if self.mlp_dist:
            raise NotImplementedError("MLP_DIST=True is outside the accelerated path")
        if normalize_before or activation != "gelu" or self.latent_dim != 256 or self.latent_size != 1:
            raise NotImplementedError("accelerated path: post-norm, gelu, latent_dim [1,256]")

        d = self.latent_dim
        self.query_pos_encoder = _PositionEmbeddingLearned1D(d)
        self.query_pos_decoder = _PositionEmbeddingLearned1D(d)
        self.encoder = _SkipStackParams(lambda: _EncoderLayerParams(d, num_heads, ff_size, dropout), num_layers, d)
        self.decoder = _SkipStackParams(lambda: _DecoderLayerParams(d, num_heads, ff_size, dropout), num_layers, d)
        self.global_motion_token = nn.Parameter(torch.randn(self.latent_size * 2, d))
        self.skel_embedding = nn.Linear(nfeats, d)
        self.final_layer = nn.Linear(d, nfeats)

        self._wcache = None      # (fingerprint, VaeWeights struct, keep-alive tensors)
        self._ws = None          # workspace tensor (grow only)

    # ------------------------------------------------------------------ weight image
    def _weights(self) -> L.VaeWeights:
        fpnt = (_param_fingerprint(self), self.precision)
        if self._wcache is not None and self._wcache[0] == fpnt:
            return self._wcache[1]
        for p in self.parameters():
            L.require_cuda(p, "MldVae parameter")
            if not p.is_contiguous():
                raise L.SeemeError("MldVae parameters must be contiguous")
        keep = []
        w = L.VaeWeights()
        w.nfeats, w.ff = self.nfeats, self.ff_size
        w.token = L.ptr(self.global_motion_token)
        w.pe_enc = L.ptr(self.query_pos_encoder.pe)
        w.pe_dec = L.ptr(self.query_pos_decoder.pe)
        F = self.nfeats
        Fp = (F + 15) // 16 * 16
        with torch.no_grad():  # zero-padded copy so that the MFMA k-loop can run in blocks of 16
            embw = torch.zeros(self.latent_dim, Fp, device=self.skel_embedding.weight.device, dtype=torch.float32)
            embw[:, :F] = self.skel_embedding.weight
        keep.append(embw)
        w.emb_w, w.emb_ldw, w.emb_b = L.ptr(embw), Fp, L.ptr(self.skel_embedding.bias)
        w.fin_w, w.fin_b = L.ptr(self.final_layer.weight), L.ptr(self.final_layer.bias)
        for stack, dst, dec in ((self.encoder, w.enc, False), (self.decoder, w.dec, True)):
            for i, blk in enumerate(stack.blocks()):
                ly = dst.layer[i]
                ly.in_w, ly.in_b = L.ptr(blk.self_attn.in_proj_weight), L.ptr(blk.self_attn.in_proj_bias)
                ly.out_w, ly.out_b = L.ptr(blk.self_attn.out_proj.weight), L.ptr(blk.self_attn.out_proj.bias)
                ly.l1_w, ly.l1_b = L.ptr(blk.linear1.weight), L.ptr(blk.linear1.bias)
                ly.l2_w, ly.l2_b = L.ptr(blk.linear2.weight), L.ptr(blk.linear2.bias)
                ly.n1_w, ly.n1_b = L.ptr(blk.norm1.weight), L.ptr(blk.norm1.bias)
                ly.n2_w, ly.n2_b = L.ptr(blk.norm2.weight), L.ptr(blk.norm2.bias)
                if dec:
                    ly.ca_in_w, ly.ca_in_b = L.ptr(blk.multihead_attn.in_proj_weight), L.ptr(blk.multihead_attn.in_proj_bias)
                    ly.ca_out_w, ly.ca_out_b = L.ptr(blk.multihead_attn.out_proj.weight), L.ptr(blk.multihead_attn.out_proj.bias)
                    ly.n3_w, ly.n3_b = L.ptr(blk.norm3.weight), L.ptr(blk.norm3.bias)
            for i in range(2):
                dst.skip_w[i] = L.ptr(stack.linear_blocks[i].weight)
                dst.skip_b[i] = L.ptr(stack.linear_blocks[i].bias)
            dst.norm_w, dst.norm_b = L.ptr(stack.norm.weight), L.ptr(stack.norm.bias)
        with torch.no_grad():   # decoder cross-attention to one memory token, folded
            D = self.latent_dim
            fw, fb = [], []
            for blk in self.decoder.blocks():
                wv, bv = blk.multihead_attn.in_proj_weight[2 * D:], blk.multihead_attn.in_proj_bias[2 * D:]
                wo, bo = blk.multihead_attn.out_proj.weight, blk.multihead_attn.out_proj.bias
                fw.append(wo @ wv)
                fb.append(wo @ bv + bo)
            fold_w, fold_b = torch.cat(fw).contiguous(), torch.cat(fb).contiguous()
        keep += [fold_w, fold_b]
        w.ca_fold_w, w.ca_fold_b = fold_w.data_ptr(), fold_b.data_ptr()
        w.h16 = 0
        if self.precision == "fp16":
            h = L.VaeWeightsH()
            pk = lambda W: (keep.append(L.pack_mfma16(W, torch.float16)), keep[-1].data_ptr())[1]
            for stack, dst in ((self.encoder, h.enc), (self.decoder, h.dec)):
                for i, blk in enumerate(stack.blocks()):
                    dst[i].in_w, dst[i].out_w = pk(blk.self_attn.in_proj_weight), pk(blk.self_attn.out_proj.weight)
                    dst[i].l1_w, dst[i].l2_w = pk(blk.linear1.weight), pk(blk.linear2.weight)
            for i in range(2):
                h.enc_skip[i] = pk(self.encoder.linear_blocks[i].weight)
                h.dec_skip[i] = pk(self.decoder.linear_blocks[i].weight)
            h.emb_w, h.fin_w, h.ca_fold_w = pk(self.skel_embedding.weight), pk(self.final_layer.weight), pk(fold_w)
            keep.append(h)
            w.h16 = C.addressof(h)
        self._wcache = (fpnt, w, keep)
        return w

    def _workspace(self, B: int, T: int, device) -> torch.Tensor:
        need = L.lib().seeme_vae_workspace_bytes(B, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    @staticmethod
    def _lengths_tensor(lengths, device) -> torch.Tensor:
        # cached per (lengths, device): no host-to-device copy in steady state (and none inside a hipGraph capture)
        key = (tuple(int(v) for v in lengths), str(device))
        t = MldVae._LEN_CACHE.get(key)
        if t is None:
            if len(MldVae._LEN_CACHE) > 64:
                MldVae._LEN_CACHE.clear()
            t = torch.as_tensor(list(key[0]), dtype=torch.int32).to(device)
            MldVae._LEN_CACHE[key] = t
        return t

    _LEN_CACHE: dict = {}

    # ------------------------------------------------------------------ reference API
    def forward(self, features, lengths: Optional[List[int]] = None):
        z, dist = self.encode(features, None, lengths)
        return self.decode(z, lengths), z, dist

    def encode_dist(self, features: torch.Tensor, lengths: Optional[List[int]] = None) -> torch.Tensor:
        """Posterior parameters as one tensor [2,B,256] (row 0 = mu, row 1 = logvar): the reference's
        ``dist`` variable before the split (mld_vae.py:172-173,186-187)."""
        L.require_cuda(features, "features")
        B, T, F = features.shape
        if F != self.nfeats:
            raise ValueError(f"features last dim {F} != nfeats {self.nfeats}")
        if lengths is None:
            lengths = [T] * B                                        # mld_vae.py:134-135
        if len(lengths) != B or max(lengths) != T or min(lengths) < 1:
            raise ValueError("lengths must have B entries with 1 <= len and max(lengths) == nframes "
                             "(the reference builds its mask from max(lengths), temos_utils.py:10-17)")
        features = features.contiguous()
        dist = torch.empty(2, B, self.latent_dim, device=features.device, dtype=torch.float32)
        lens = self._lengths_tensor(lengths, features.device)
        ws = self._workspace(B, T, features.device)
        w = self._weights()
        rc = L.lib().seeme_vae_encode(C.byref(w), features.data_ptr(), lens.data_ptr(), B, T,
                                      dist.data_ptr(), dist[1].data_ptr(), ws.data_ptr(), ws.numel(),
                                      L.current_stream())
        L.check(rc, "seeme_vae_encode")
        return dist

    def encode(self, features: torch.Tensor, images: Optional[torch.Tensor] = None,
               lengths: Optional[List[int]] = None):
        dist = self.encode_dist(features, lengths)
        mu, logvar = dist[0:1], dist[1:2]
        std = logvar.exp().pow(0.5)                                   # mld_vae.py:190
        normal = torch.distributions.Normal(mu, std, validate_args=False)   # (validation would force a host sync)
        latent = normal.rsample()                                     # :192 (RNG stays in torch)
        return latent, normal

    def decode(self, z: torch.Tensor, lengths: List[int]):
        L.require_cuda(z, "z")
        if z.dim() != 3 or z.shape[0] != 1 or z.shape[2] != self.latent_dim:
            raise ValueError("z must be [1, B, 256]")
        B = z.shape[1]
        if len(lengths) != B:
            raise ValueError("len(lengths) must equal the batch size")
        T = int(max(lengths))
        z2 = z.reshape(B, self.latent_dim).contiguous()
        feats = torch.empty(B, T, self.nfeats, device=z.device, dtype=torch.float32)
        lens = self._lengths_tensor(lengths, z.device)
        ws = self._workspace(B, T, z.device)
        w = self._weights()
        rc = L.lib().seeme_vae_decode(C.byref(w), z2.data_ptr(), lens.data_ptr(), B, T, feats.data_ptr(),
                                      ws.data_ptr(), ws.numel(), L.current_stream())
        L.check(rc, "seeme_vae_decode")
        return feats
