"""MldDenoiser -- drop-in for ``mld.models.architectures.mld_denoiser.MldDenoiser``
(reference mld_denoiser.py:18-244, layer mdiff_transformer.py:257-304).

Same constructor arguments, ``forward(sample, timestep, encoder_hidden_states, lengths)`` signature
and layouts (condition **seq-first** [N,B,256]), same ``state_dict`` keys.
The arithmetic runs in libseeme_hip.so; on top of the reference surface there is ``sample_loop``:
the whole reverse-diffusion loop of MLD._diffusion_reverse (mld.py:467-497) as ONE kernel launch.

``target: seeme_amd.mld_denoiser.MldDenoiser`` in configs/modules/denoiser.yaml selects it.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from .mld_vae import _PositionEmbeddingLearned1D, _EncoderLayerParams, _SkipStackParams, _param_fingerprint

_LAYER_FIELDS = ["skip", "inp", "outp", "l1", "l2", "caq", "cao", "f1", "f2", "fo",
                 "skip_b", "in_b", "out_b", "n1w", "n1b", "l1b", "l2b", "n2w", "n2b", "cnw", "cnb", "caq_b",
                 "csnw", "csnb", "cao_b", "f1b", "f2b", "fsnw", "fsnb", "fo_b"]


_CUS = {}


def _device_cus(device) -> int:
    """Compute units of `device` (cached): a cluster launch needs all its workgroups resident at once."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _CUS:
        _CUS[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _CUS[idx]


def timestep_features(timesteps: torch.Tensor, dim: int = 256, flip_sin_to_cos: bool = True,
                      freq_shift: float = 0.0, max_period: int = 10000) -> torch.Tensor:
    """Sinusoidal timestep features, tools/embeddings.py:245-285 (host-side plumbing: a [rows,256]
    table; evaluated with torch on whatever device `timesteps` lives on)."""
    assert timesteps.dim() == 1
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1, 0, 0))
    return emb


def cluster_pack_stage(Wsub: torch.Tensor, TPW: int, KL: int, UL: int) -> torch.Tensor:
    """One GEMV stage of the cluster weight image (csrc/den_cluster.inc.hip): [8 TPW x 16 outputs, K] -> [units, wave 8, UL, lane 64,
    KL / 4].  Wave w owns the 16-output tiles w TPW .. w TPW + TPW - 1; its wave-loads run k-block major, tile inner (load j =
    k-block j // TPW, tile j % TPW), UL loads per unit; lane 16 g + r of a load holds W[16 T + r][KL kb + (KL/4) g .. + KL/4 - 1]:
    the B operand of v_mfma_f32_16x16x32 (16-bit, KL = 32) or of four v_mfma_f32_16x16x4 (fp32, KL = 16)."""
    EPL = KL // 4
    NT, K = Wsub.shape[0] // 16, Wsub.shape[1]
    assert NT == 8 * TPW and K % KL == 0 and (K // KL * TPW) % UL == 0
    KB = K // KL
    t = Wsub.reshape(NT, 16, KB, 4, EPL).permute(0, 2, 3, 1, 4).reshape(8, TPW, KB, 64, EPL)
    t = t.permute(0, 2, 1, 3, 4).reshape(8, (KB * TPW) // UL, UL, 64, EPL)
    return t.permute(1, 0, 2, 3, 4)


class _TimestepEmbeddingParams(nn.Module):
    def __init__(self, channel, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(channel, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class _StylizationParams(nn.Module):
    """mdiff_transformer.py:137-150 (indices 1 / 2 of the Sequentials carry the Linear layers)."""

    def __init__(self, d, t):
        super().__init__()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(t, 2 * d))
        self.norm = nn.LayerNorm(d)
        self.out_layers = nn.Sequential(nn.SiLU(), nn.Dropout(p=0.0), nn.Linear(d, d))


class _CrossAttnParams(nn.Module):
    def __init__(self, d, t):
        super().__init__()
        self.norm = nn.LayerNorm(d)
        self.text_norm = nn.LayerNorm(d)
        self.query = nn.Linear(d, d)
        self.key = nn.Linear(d, d)
        self.value = nn.Linear(d, d)
        self.proj_out = _StylizationParams(d, t)


class _FFNParams(nn.Module):
    def __init__(self, d, ff, t):
        super().__init__()
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.proj_out = _StylizationParams(d, t)


class _MDLayerParams(nn.Module):
    """LinearTemporalDiffusionTransformerDecoderLayer (mdiff_transformer.py:257-284): registration
    order ca_block, ffn, sa_block as in the reference; sa_block ff is hard-coded 1024/relu (:279)."""

    def __init__(self, d, ff, nhead, dropout):
        super().__init__()
        self.d_model = d
        self.ca_block = _CrossAttnParams(d, d)
        self.ffn = _FFNParams(d, ff, d)
        self.sa_block = _EncoderLayerParams(d, nhead, 1024, dropout)


class MldDenoiser(nn.Module):

    def __init__(self,
                 ablation,
                 nfeats: int = 72,
                 condition: str = "text",
                 latent_dim: list = [1, 256],
                 ff_size: int = 128,
                 num_layers: int = 6,
                 num_heads: int = 4,
                 dropout: float = 0.1,
                 normalize_before: bool = False,
                 activation: str = "gelu",
                 flip_sin_to_cos: bool = True,
                 return_intermediate_dec: bool = False,
                 position_embedding: str = "learned",
                 arch: str = "trans_enc",
                 freq_shift: int = 0,
                 guidance_scale: float = 7.5,
                 guidance_uncondp: float = 0.1,
                 text_encoded_dim: int = 256,
                 nclasses: int = 10,
                 weight_dtype: str = "fp32",
                 cluster="auto",
                 **kwargs) -> None:
        super().__init__()
        self.latent_dim = latent_dim[-1]
        self.text_encoded_dim = text_encoded_dim
        self.condition = condition
        self.arch = arch
        self.pe_type = ablation.DIFF_PE_TYPE
        self.MD_trans = ablation.MD_TRANS
        self.flip_sin_to_cos, self.freq_shift = flip_sin_to_cos, freq_shift
        self.num_heads, self.ff_size, self.num_layers = num_heads, ff_size, num_layers
        if "text" not in self.condition:
            raise TypeError(f"condition type {self.condition} not supported")      # mld_denoiser.py:190
        if self.pe_type != "mld":
            if self.pe_type == "actor":
                raise NotImplementedError("DIFF_PE_TYPE 'actor' is outside the accelerated path")
            raise ValueError("Not Support PE type")                                # :96
        if arch != "trans_enc":
            if arch == "trans_dec":
                raise NotImplementedError("arch 'trans_dec' is outside the accelerated path")
            raise ValueError(f"Not supported architechure{self.arch}!")            # :149
        if ablation.VAE_TYPE == "no" or not ablation.SKIP_CONNECT or not self.MD_trans:
            raise NotImplementedError("accelerated path: VAE latent + SKIP_CONNECT + MD_TRANS (all SEE-ME configs)")
        if text_encoded_dim != self.latent_dim or self.latent_dim != 256 or latent_dim[0] != 1 or num_layers != 5:
            raise NotImplementedError("accelerated path: latent_dim [1,256], text_encoded_dim 256, 5 layers")
        if num_heads not in (1, 2, 4):
            raise NotImplementedError("accelerated path: num_heads in {1,2,4}")
        if weight_dtype not in ("fp32", "bf16", "fp16"):
            raise ValueError("weight_dtype must be 'fp32', 'bf16' or 'fp16'")
        self.weight_dtype = weight_dtype
        # CUs per sample in sample_loop / forward: "auto" (8 / 4 / 2 while B x C <= 256 -- one head, at most two condition tokens, no CFG pair --, i.e. batches that leave most of the chip
        # idle; one CU per sample beyond), 0 (never) or 2 / 4 / 8.  SEEME_DEN_CLUSTER overrides.  Speed only.
        if cluster not in ("auto", 0, 2, 4, 8):
            raise ValueError("cluster must be 'auto', 0, 2, 4 or 8")
        self.cluster = cluster
        # 1: a cluster's C workgroups sit on C different XCDs (consecutive blockIdx), so each XCD's L2 only ever sees ONE slice of the
        # weights (2.3 MB at C = 8: resident) and nothing is re-fetched from the Infinity Cache; 0: one XCD per cluster (granules stay in
        # that L2, but every L2 streams the whole 11 MB per step).  Same speed at B = 32 (bench 11.77 k vs 11.79 k seqs/s).
        self.cluster_placement = 1
        self.cluster_flags = 0           # bit 0: write-through granule stores always
        self.cluster_ms = True           # batches above 32: clusters that own up to 8 samples (k_den_cluster_ms)
        self.pack_xcds = "auto"          # one-CU-per-sample launches: XCDs the working workgroups sit on (_lib.default_xcds); 8 = dealt out

        d = self.latent_dim
        self.time_embedding = _TimestepEmbeddingParams(text_encoded_dim, d)
        self.query_pos = _PositionEmbeddingLearned1D(d)
        self.mem_pos = _PositionEmbeddingLearned1D(d)    # unused in trans_enc, kept for state_dict parity
        self.encoder = _SkipStackParams(lambda: _MDLayerParams(d, ff_size, num_heads, dropout), num_layers, d)

        self._wcache = None
        self._ws = None
        self._table_cache = {}
        self._ccache = {}
        self._xchg = None

    # ------------------------------------------------------------------ weight image
    def _layout(self):
        n = 5 * len(_LAYER_FIELDS) + 5
        buf = (C.c_int64 * n)()
        L.check(L.lib().seeme_den_layout(1024, self.ff_size, buf, n), "seeme_den_layout")
        vals = list(buf)
        layers = [dict(zip(_LAYER_FIELDS, vals[i * 30:(i + 1) * 30])) for i in range(5)]
        pe0, fnw, fnb, wg_total, vp_total = vals[150:155]
        self._layout_vals = vals
        return layers, pe0, fnw, fnb, wg_total, vp_total

    def _weights(self):
        fpnt = (_param_fingerprint(self), self.weight_dtype)
        if self._wcache is not None and self._wcache[0] == fpnt:
            return self._wcache[1]
        for p in self.parameters():
            L.require_cuda(p, "MldDenoiser parameter")
        dev = self.query_pos.pe.device
        layers, pe0, fnw, fnb, wg_total, vp_total = self._layout()
        bf16 = self.weight_dtype != "fp32"          # any 16-bit image: 8 weights per 16-byte vector
        KV = 8 if bf16 else 4
        wdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[self.weight_dtype]
        wg = torch.zeros(wg_total, dtype=wdt, device=dev)
        vp = torch.zeros(vp_total, dtype=torch.float32, device=dev)

        def put_w(off, W):
            N, K = W.shape
            if bf16:
                # matrix-core stream (den_kernels.hip, GS::TW / TG / KB): wave w owns output tiles w*TW .. w*TW+TW-1 (16
                # outputs each) for all k, in groups of TG tiles; within a group k-block major, then tile; lane =
                # 16*(k-group) + output row holds 8 consecutive k -- the v_mfma_f32_16x16x32 operand layout
                TW, KB = N // 128, K // 32
                TG = 4 if TW % 4 == 0 else (3 if TW % 3 == 0 else (2 if TW % 2 == 0 else 1))
                g = W.detach().reshape(8, TW // TG, TG, 16, KB, 4, 8).permute(0, 1, 4, 2, 5, 3, 6).reshape(-1)
            else:
                # vector-ALU stream: [K/KV][N][KV] 16-byte vectors
                g = W.detach().reshape(N, K // KV, KV).permute(1, 0, 2).reshape(-1)
            wg[off:off + g.numel()] = g.to(wg.dtype)

        def put_v(off, v):
            vp[off:off + v.numel()] = v.detach().reshape(-1)

        blocks = self.encoder.blocks()
        # One attention head: fold out_proj into the value projection (softmax weights sum to 1, so the bias
        # folds too): in_proj V rows <- W_o W_v, bias <- W_o b_v + b_o; the out_proj GEMV disappears and the
        # K|V tables of the condition / time tokens carry W_o v directly (den_kernels.hip, V_FOLD).
        fold = self.num_heads == 1

        def sa_in_proj(sa):
            Wi, bi = sa.self_attn.in_proj_weight.detach(), sa.self_attn.in_proj_bias.detach()
            if not fold:
                return Wi, bi
            Wo, bo = sa.self_attn.out_proj.weight.detach().double(), sa.self_attn.out_proj.bias.detach().double()
            Wu = (Wo @ Wi[512:].double()).float()
            bu = (Wo @ bi[512:].double() + bo).float()
            return torch.cat([Wi[:512], Wu]), torch.cat([bi[:512], bu])

        with torch.no_grad():
            in_proj = [sa_in_proj(b.sa_block) for b in blocks]
            put_v(pe0, self.query_pos.pe[0, 0])
            put_v(fnw, self.encoder.norm.weight)
            put_v(fnb, self.encoder.norm.bias)
            for l, (blk, o) in enumerate(zip(blocks, layers)):
                sa, ca, ffn = blk.sa_block, blk.ca_block, blk.ffn
                if l >= 3:
                    put_w(o["skip"], self.encoder.linear_blocks[l - 3].weight)
                    put_v(o["skip_b"], self.encoder.linear_blocks[l - 3].bias)
                put_w(o["inp"], in_proj[l][0]); put_v(o["in_b"], in_proj[l][1])
                if not fold:
                    put_w(o["outp"], sa.self_attn.out_proj.weight); put_v(o["out_b"], sa.self_attn.out_proj.bias)
                put_w(o["l1"], sa.linear1.weight); put_v(o["l1b"], sa.linear1.bias)
                put_w(o["l2"], sa.linear2.weight); put_v(o["l2b"], sa.linear2.bias)
                put_v(o["n1w"], sa.norm1.weight); put_v(o["n1b"], sa.norm1.bias)
                put_v(o["n2w"], sa.norm2.weight); put_v(o["n2b"], sa.norm2.bias)
                put_v(o["cnw"], ca.norm.weight); put_v(o["cnb"], ca.norm.bias)
                put_w(o["caq"], ca.query.weight); put_v(o["caq_b"], ca.query.bias)
                put_v(o["csnw"], ca.proj_out.norm.weight); put_v(o["csnb"], ca.proj_out.norm.bias)
                put_w(o["cao"], ca.proj_out.out_layers[2].weight); put_v(o["cao_b"], ca.proj_out.out_layers[2].bias)
                put_w(o["f1"], ffn.linear1.weight); put_v(o["f1b"], ffn.linear1.bias)
                put_w(o["f2"], ffn.linear2.weight); put_v(o["f2b"], ffn.linear2.bias)
                put_v(o["fsnw"], ffn.proj_out.norm.weight); put_v(o["fsnb"], ffn.proj_out.norm.bias)
                put_w(o["fo"], ffn.proj_out.out_layers[2].weight); put_v(o["fo_b"], ffn.proj_out.out_layers[2].bias)
            kv_w = torch.cat([in_proj[l][0][256:] for l in range(len(blocks))]).contiguous()
            kv_b = torch.cat([in_proj[l][1][256:] for l in range(len(blocks))]).contiguous()
            st_w = torch.cat([torch.cat([b.ca_block.proj_out.emb_layers[1].weight, b.ffn.proj_out.emb_layers[1].weight])
                              for b in blocks]).contiguous()
            st_b = torch.cat([torch.cat([b.ca_block.proj_out.emb_layers[1].bias, b.ffn.proj_out.emb_layers[1].bias])
                              for b in blocks]).contiguous()
            ca_w = [torch.cat([b.ca_block.key.weight, b.ca_block.value.weight]).contiguous() for b in blocks]
            ca_b = [torch.cat([b.ca_block.key.bias, b.ca_block.value.bias]).contiguous() for b in blocks]
            # text_norm affine folded into key|value: W diag(g), W beta + b
            cf_w = torch.cat([ca_w[l] * blk.ca_block.text_norm.weight[None, :] for l, blk in enumerate(blocks)]).contiguous()
            cf_b = torch.cat([ca_w[l] @ blk.ca_block.text_norm.bias + ca_b[l] for l, blk in enumerate(blocks)]).contiguous()
            ones, zeros = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        lay_dev = torch.tensor(self._layout_vals, dtype=torch.int64, device=dev)
        w = L.DenoiserWeights()
        w.wg, w.wdtype, w.vp, w.layout = wg.data_ptr(), {"fp32": 0, "bf16": 1, "fp16": 2}[self.weight_dtype], vp.data_ptr(), lay_dev.data_ptr()
        w.nhead, w.ff_sa, w.ff = self.num_heads, 1024, self.ff_size
        w.kv_cat_w, w.kv_cat_b, w.style_cat_w, w.style_cat_b = kv_w.data_ptr(), kv_b.data_ptr(), st_w.data_ptr(), st_b.data_ptr()
        te = self.time_embedding
        w.time_w1, w.time_b1 = L.ptr(te.linear_1.weight), L.ptr(te.linear_1.bias)
        w.time_w2, w.time_b2 = L.ptr(te.linear_2.weight), L.ptr(te.linear_2.bias)
        for l, b in enumerate(blocks):
            w.ca_kv_w[l], w.ca_kv_b[l] = ca_w[l].data_ptr(), ca_b[l].data_ptr()
            w.ca_tn_w[l], w.ca_tn_b[l] = L.ptr(b.ca_block.text_norm.weight), L.ptr(b.ca_block.text_norm.bias)
        with torch.no_grad():   # ca_block.proj_out of the five layers, stacked for the batched table builder
            pn_w = torch.stack([b.ca_block.proj_out.norm.weight for b in blocks]).contiguous()
            pn_b = torch.stack([b.ca_block.proj_out.norm.bias for b in blocks]).contiguous()
            po_w = torch.stack([b.ca_block.proj_out.out_layers[2].weight for b in blocks]).contiguous()
            po_b = torch.stack([b.ca_block.proj_out.out_layers[2].bias for b in blocks]).contiguous()
        w.ca_pn_w, w.ca_pn_b, w.ca_po_w, w.ca_po_b = pn_w.data_ptr(), pn_b.data_ptr(), po_w.data_ptr(), po_b.data_ptr()
        w.sa_fold = int(fold)
        w.ca_fold_w, w.ca_fold_b, w.ln_ones, w.ln_zeros = cf_w.data_ptr(), cf_b.data_ptr(), ones.data_ptr(), zeros.data_ptr()
        self._ccache = {}
        self._fold_parts = (in_proj, vp, layers)
        self._wcache = (fpnt, w, (wg, vp, kv_w, kv_b, st_w, st_b, ca_w, ca_b, lay_dev, cf_w, cf_b, ones, zeros, pn_w, pn_b, po_w, po_b))
        self._table_cache = {}
        return w

    # ------------------------------------------------------------------ cluster image (csrc/den_cluster.inc.hip)
    def _cluster_weights(self, Cc: int, query: bool = False):
        """Weight image for one sample split over Cc CUs: [layer][CU][unit][wave 8][load][lane 64][16 B] in order of use.
        Stage A = rows c S .. of q | k | W_o v (| y) of [W_in' W_s ; W_s] (layers 3, 4: the skip linear folded in, acting on
        cat[x, skip]; cross_attention.py:77-79) or of W_in' (layers 0-2); B = linear1 rows of the CU's hidden units; C = the
        matching columns of linear2; D, E, F = ffn.linear1 / linear2 / proj_out whole (replicated).  `query` (several condition tokens):
        between C and D, G = the rows of ca_block.query of the CU's dims and H = ca_block.proj_out.out_layers whole."""
        self._weights()
        if (Cc, query) in self._ccache:
            return self._ccache[(Cc, query)]
        in_proj, vp, layers = self._fold_parts
        dev = vp.device
        code = {"fp32": 0, "bf16": 1, "fp16": 2}[self.weight_dtype]
        wdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[self.weight_dtype]
        lo = (C.c_int64 * 14)()
        L.check(L.lib().seeme_den_cluster_layout(Cc, code, int(query), lo, 14), "seeme_den_cluster_layout")
        NU, UB, total, UL, KL = (int(v) for v in lo[:5])
        U_D, U_G, U_H = int(lo[9]), int(lo[12]), int(lo[13])
        EPL = KL // 4                                   # elements per lane per wave-load (16 B)
        S, NB, TA, TB = 256 // Cc, 1024 // Cc, 256 // Cc // 32, 8 // Cc

        def pack_stage(Wsub, TPW):
            return cluster_pack_stage(Wsub, TPW, KL, UL)

        blocks = self.encoder.blocks()
        img = torch.zeros(5, Cc, NU, 8, UL, 64, EPL, dtype=wdt, device=dev)
        vpc = vp.clone()
        with torch.no_grad():
            for l, (blk, o) in enumerate(zip(blocks, layers)):
                Wq, bq = in_proj[l][0].double(), in_proj[l][1].double()
                if l >= 3:
                    Ws = self.encoder.linear_blocks[l - 3].weight.detach().double()
                    bs = self.encoder.linear_blocks[l - 3].bias.detach().double()
                    Wa = torch.cat([Wq @ Ws, Ws])                                  # [1024, 512] on cat[x, skip]
                    vpc[o["in_b"]:o["in_b"] + 768] = (Wq @ bs + bq).float()
                else:
                    Wa = torch.cat([Wq, torch.zeros(256, 256, dtype=torch.float64, device=dev)])   # no y rows
                W1, W2 = blk.sa_block.linear1.weight.detach(), blk.sa_block.linear2.weight.detach()
                Wf1, Wf2 = blk.ffn.linear1.weight.detach(), blk.ffn.linear2.weight.detach()
                Wfo = blk.ffn.proj_out.out_layers[2].weight.detach()
                for c in range(Cc):
                    rows = torch.cat([torch.arange(p * 256 + c * S, p * 256 + (c + 1) * S, device=dev) for p in range(4)])
                    ua = pack_stage(Wa[rows].to(wdt), TA)              # TA units (K = 256) or 2 TA (K = 512)
                    img[l, c, 0:ua.shape[0]] = ua
                    img[l, c, 2 * TA:2 * TA + TB] = pack_stage(W1[c * NB:(c + 1) * NB].to(wdt), TB)
                    img[l, c, 2 * TA + TB:2 * TA + 2 * TB] = pack_stage(W2[:, c * NB:(c + 1) * NB].to(wdt), 2)
                    if query:
                        Wq_ca = blk.ca_block.query.weight.detach()[c * S:(c + 1) * S]          # S / 16 tiles, one per wave; the other waves idle
                        Wq_pad = torch.zeros(128, 256, dtype=Wq_ca.dtype, device=dev)
                        Wq_pad[:S] = Wq_ca
                        img[l, c, U_G:U_G + 1] = pack_stage(Wq_pad.to(wdt), 1)
                        img[l, c, U_H:U_H + 2] = pack_stage(blk.ca_block.proj_out.out_layers[2].weight.detach().to(wdt), 2)
                    u0 = U_D
                    img[l, c, u0:u0 + 1] = pack_stage(Wf1.to(wdt), 1)
                    img[l, c, u0 + 1:u0 + 2] = pack_stage(Wf2.to(wdt), 2)
                    img[l, c, u0 + 2:u0 + 4] = pack_stage(Wfo.to(wdt), 2)
        assert img.numel() * img.element_size() == total
        self._ccache[(Cc, query)] = (img, vpc, code)
        return self._ccache[(Cc, query)]

    def _cluster_size(self, B: int, N: int, cfg: bool, cus: int = 256) -> int:
        """CUs per sample for this launch (0: the one-CU-per-sample kernel).  `cus`: compute units of the device -- every
        workgroup of a cluster launch must be resident at once (one per CU)."""
        want = os.environ.get("SEEME_DEN_CLUSTER")
        want = self.cluster if want is None else (want if want == "auto" else int(want))
        if want == 0 or N > 2 or cfg or self.num_heads != 1:
            return 0
        Bp = (B + 7) // 8 * 8
        cus = min(int(cus), 256)
        if want == "auto":
            for Cc in (8, 4, 2):
                if Bp * Cc <= cus:
                    return Cc
            return 0
        return want if Bp * want <= cus else 0

    def _cluster_plan(self, B: int, N: int, cfg: bool, per_sample: bool, cus: int = 256):
        """(CUs per cluster, samples per cluster) of this launch; (0, 1) = the one-CU-per-sample kernel.  Up to 64 samples a cluster owns
        ONE sample (k_den_cluster, windowed schedules: 8 CUs up to B = 32, 4 up to 64); above that -- 16-bit image, one or two condition tokens,
        one table row per step -- the large-batch form k_den_cluster_ms: 64 clusters of 4 CUs that own ceil(B / 64) <= 8 samples each
        (B <= 512; with two condition tokens up to 6 samples, B <= 384: 5.14 -> 4.09 ms at B = 128, 5.35 -> 4.54 at 256, none at 512).  Measured at 50 DDIM steps (profiles/r03_g_cluster_ms.txt, r03_k_c4_windows.txt): B = 64 2.83 ms (8 CUs x 2 samples
        3.09), B = 128 3.44 against 4.30 (2 CUs per sample) / 4.46 (one), B = 256 3.66 against 4.70, B = 512 4.37 against 4.92;
        8 CUs x 4 or 8 samples is slower: the exchange volume of a CU grows with C x samples."""
        Cc = self._cluster_size(B, N, cfg, cus)
        want = os.environ.get("SEEME_DEN_CLUSTER")
        want = self.cluster if want is None else (want if want == "auto" else int(want))
        ms = os.environ.get("SEEME_DEN_CLUSTER_MS", "1") != "0" and self.cluster_ms
        if (want != "auto" or not ms or B <= 32 or N not in (1, 2) or cfg or per_sample or self.num_heads != 1
                or self.weight_dtype not in ("fp16", "bf16")):
            return Cc, 1
        forced = os.environ.get("SEEME_DEN_CLUSTER_MS_PLAN")           # "C,samples" (samples >= 2): measurement only
        if forced:
            C4, spc = (int(x) for x in forced.split(","))
            return C4, spc
        cus = min(int(cus), 256)
        if B <= 64 and Cc == 4:
            return Cc, 1
        ncl = (cus // 4) // 8 * 8
        most = 4 if self.weight_dtype == "bf16" else (8 if N == 1 else 6)   # bf16: four A rows per sample; two condition tokens: no gain left at 7, 8
        if ncl >= 8 and -(-B // ncl) <= most:
            return 4, max(2, -(-B // ncl))
        return Cc, 1

    def cluster_status(self):
        """(give-up code, clusters that ran with L2-local granule stores) of the last cluster launch; synchronises."""
        if self._xchg is None:
            return (0, 0)
        hdr = self._xchg[:8].cpu().view(torch.int32)
        return int(hdr[0]), int(hdr[1])

    def _workspace(self, rows: int, device):
        need = L.lib().seeme_denoiser_workspace_bytes(rows, 0, 0)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    # ------------------------------------------------------------------ tables
    def time_tables(self, tfeat: torch.Tensor) -> torch.Tensor:
        """[rows,256] sinusoidal features -> [rows, 7680] (K|V of the time token, AdaLN scale|shift)."""
        L.require_cuda(tfeat, "tfeat")
        tfeat = tfeat.contiguous()
        rows = tfeat.shape[0]
        ttab = torch.empty(rows, L.TROW, device=tfeat.device, dtype=torch.float32)
        ws = self._workspace(rows, tfeat.device)
        w = self._weights()
        L.check(L.lib().seeme_denoiser_time_tables(C.byref(w), tfeat.data_ptr(), rows, ttab.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), L.current_stream()),
                "seeme_denoiser_time_tables")
        return ttab

    def cond_tables(self, cond_bf: torch.Tensor) -> torch.Tensor:
        """[Bc,N,256] batch-first condition tokens -> [Bc,N,5120] step-invariant K/V tables."""
        L.require_cuda(cond_bf, "condition")
        cond_bf = cond_bf.contiguous()
        Bc, N, _ = cond_bf.shape
        ctab = torch.empty(Bc, N, L.CROW, device=cond_bf.device, dtype=torch.float32)
        w = self._weights()
        L.check(L.lib().seeme_denoiser_cond_tables(C.byref(w), cond_bf.data_ptr(), Bc, N, ctab.data_ptr(), 0, 0,
                                                    L.current_stream()), "seeme_denoiser_cond_tables")
        return ctab

    def ca_tables(self, ctab: torch.Tensor, ttab: torch.Tensor, trow: torch.Tensor, per_sample: bool) -> torch.Tensor:
        """ONE condition token: the ca_block term of every (sample, table row, layer), [Bc,R,5,256] (it does not
        depend on the latent -- mdiff_transformer.py:231-237 with a single key; include/seeme_hip.h)."""
        Bc = ctab.shape[0]
        R = 1 if per_sample else trow.numel()
        catab = torch.empty(Bc, R, 5, self.latent_dim, device=ctab.device, dtype=torch.float32)
        ws = torch.empty(5 * Bc * R * 256, device=ctab.device, dtype=torch.float32)
        w = self._weights()
        L.check(L.lib().seeme_denoiser_ca_tables(C.byref(w), ctab.data_ptr(), ttab.data_ptr(), trow.data_ptr(), int(per_sample),
                                                  trow.numel(), Bc, catab.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                  L.current_stream()), "seeme_denoiser_ca_tables")
        return catab

    def _launch(self, latents2d, ctab, ttab, trow, per_sample, steps, sched, coef, noise, cfg, guidance):
        B, N = latents2d.shape[0], ctab.shape[1]
        out = torch.empty(B, self.latent_dim, device=latents2d.device, dtype=torch.float32)
        a = L.SampleArgs()
        a.B, a.N, a.steps, a.sched, a.cfg, a.guidance_scale = B, N, steps, sched, int(cfg), float(guidance)
        a.latents, a.ctab, a.ttab, a.trow, a.trow_per_sample = latents2d.data_ptr(), ctab.data_ptr(), ttab.data_ptr(), trow.data_ptr(), int(per_sample)
        a.coef, a.noise, a.out = L.ptr(coef), L.ptr(noise), out.data_ptr()
        catab = self.ca_tables(ctab, ttab, trow, per_sample) if N == 1 else None
        a.catab = L.ptr(catab)
        a.xcds = L.default_xcds((B + 1) // 2 if (B > 256 and not per_sample and not cfg) else B, self.pack_xcds)
        w = self._weights()
        Cc, spc = self._cluster_plan(B, N, bool(cfg), bool(per_sample), _device_cus(latents2d.device))
        if Cc:
            img, vpc, code = self._cluster_weights(Cc, N > 1)
            need = (L.lib().seeme_den_cluster_ms_xchg_bytes(B, Cc, spc) if spc > 1 else L.lib().seeme_den_cluster_xchg_bytes(B, Cc))
            if self._xchg is None or self._xchg.numel() < need or self._xchg.device != latents2d.device:
                self._xchg = torch.zeros(need, dtype=torch.uint8, device=latents2d.device)
            cl = L.DenCluster()
            cl.wgc, cl.wdtype, cl.vpc, cl.C = img.data_ptr(), code, vpc.data_ptr(), Cc
            cl.placement = int(os.environ.get("SEEME_DEN_CLUSTER_PLACE", self.cluster_placement))
            cl.flags = int(os.environ.get("SEEME_DEN_CLUSTER_FLAGS", self.cluster_flags))
            cl.xchg, cl.xchg_bytes = self._xchg.data_ptr(), self._xchg.numel()
            cl.query = int(N > 1)
            cl.samples = spc
            L.check(L.lib().seeme_denoiser_sample_cluster(C.byref(w), C.byref(cl), C.byref(a), L.current_stream()),
                    "seeme_denoiser_sample_cluster")
            return out
        L.check(L.lib().seeme_denoiser_sample(C.byref(w), C.byref(a), L.current_stream()), "seeme_denoiser_sample")
        return out

    # ------------------------------------------------------------------ reference API
    def forward(self, sample, timestep, encoder_hidden_states, lengths=None, **kwargs):
        """sample [B,1,256]; timestep 0-d or [B]; encoder_hidden_states [N,B,256] -> ([B,1,256],)."""
        L.require_cuda(sample, "sample")
        L.require_cuda(encoder_hidden_states, "encoder_hidden_states")
        B = sample.shape[0]
        if sample.shape[1] != 1:
            raise NotImplementedError("accelerated path: one latent token (latent_dim [1,256])")
        t = torch.as_tensor(timestep)
        per_sample = t.dim() > 0 and t.numel() > 1
        t1 = t.reshape(-1) if per_sample else t.reshape(1)            # mld_denoiser.py:167 (expand)
        tfeat = timestep_features(t1.to(sample.device), self.text_encoded_dim, self.flip_sin_to_cos, self.freq_shift)
        ttab = self.time_tables(tfeat.to(torch.float32))
        trow = (torch.arange(B, dtype=torch.int32, device=sample.device) if per_sample
                else torch.zeros(max(B, 1), dtype=torch.int32, device=sample.device))
        ctab = self.cond_tables(encoder_hidden_states.permute(1, 0, 2))
        out = self._launch(sample.reshape(B, -1).contiguous(), ctab, ttab, trow, True, 1, L.SCHED_NONE,
                           None, None, False, 1.0)
        return (out.reshape(B, 1, self.latent_dim),)

    # ------------------------------------------------------------------ fused loop
    def sample_loop(self, latents: torch.Tensor, cond_bf: torch.Tensor, scheduler, eta: float = 0.0,
                    guidance_scale: float = 1.0, step_noise: Optional[torch.Tensor] = None,
                    events=None) -> torch.Tensor:
        """MLD._diffusion_reverse (mld.py:467-497) fused.  latents [B,1,256] (already scaled by
        init_noise_sigma); cond_bf batch-first [B or 2B,N,256] (uncond first when guidance_scale > 1);
        scheduler: seeme_amd.schedulers.* after set_timesteps().  Returns [1,B,256]."""
        L.require_cuda(latents, "latents")
        B = latents.shape[0]
        cfg = guidance_scale > 1.0
        if cond_bf.shape[0] != (2 * B if cfg else B):
            raise ValueError("condition batch must be B (or 2B with classifier-free guidance)")
        dev = latents.device
        key = (tuple(scheduler.timesteps.tolist()), float(eta), type(scheduler).__name__)
        cached = self._table_cache.get(key)
        self._weights()
        if cached is None:
            tfeat = timestep_features(scheduler.timesteps.cpu(), self.text_encoded_dim, self.flip_sin_to_cos,
                                      self.freq_shift).to(dev)
            coef = scheduler.coef_table(eta).to(dev)
            trow = torch.arange(len(scheduler.timesteps), dtype=torch.int32, device=dev)
            # the time tables are a function of (weights, timestep schedule) only -- like the packed weight image
            # they are built once per scheduler configuration (_weights() drops the cache when a parameter changes)
            cached = (tfeat, coef, trow, self.time_tables(tfeat))
            self._table_cache = {key: cached}
        tfeat, coef, trow, ttab = cached
        ctab = self.cond_tables(cond_bf)
        steps = len(trow)
        noise = None
        if scheduler.needs_noise(eta):
            if step_noise is None:
                step_noise = torch.randn(steps, B, self.latent_dim, device=dev, dtype=torch.float32)
            noise = step_noise.reshape(steps, B, self.latent_dim).contiguous()
        sched = L.SCHED_DDIM if type(scheduler).__name__.startswith("DDIM") else L.SCHED_DDPM
        lat2 = latents.reshape(B, -1).contiguous()
        if events is not None:     # (start, end) torch.cuda.Event pair bracketing only the persistent kernel
            events[0].record()
        out = self._launch(lat2, ctab, ttab, trow, False, steps, sched, coef, noise, cfg, guidance_scale)
        if events is not None:
            events[1].record()
        return out.reshape(1, B, self.latent_dim)
