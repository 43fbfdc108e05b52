"""Stage-2 training of the denoiser on the HIP path: forward + hand-written backward of the token-0 chain.

``MLD._diffusion_process`` (mld/models/modeltype/mld.py:582-631) calls :func:`denoiser_forward_hip_train` when the
denoiser has one attention head (all shipped ``config_mld_*.yaml``); other shapes keep the PyTorch autograd twin
(``denoiser_autograd.py``).  Division of labour:

* the persistent kernels do what is a dependent chain per sample: the denoiser forward (the sampling kernel, one step,
  per-sample timesteps, unfolded fp32 weights, intermediates saved) and its backward (``k_den_bwd``,
  ``seeme_amd/csrc/den_train.inc.hip``), one workgroup per sample;
* what is a reduction over the batch is a handful of batched GEMMs / sums on the buffers the backward kernel writes
  (x and dy of every linear, LayerNorm parameter terms): ``dW = sum_b dy_b x_b^T``;
* the condition / time tables (K|V of the condition and time tokens, linear-attention keys | values, AdaLN rows,
  timestep embedding MLP) are built with ~10 differentiable torch ops, and autograd carries the table gradients
  returned by the kernel back into their parameters (``in_proj`` rows 256.. are shared between the chain and the
  tables; autograd adds the two contributions).

The weight images (forward and transposed layouts) are refreshed from the parameter tensors by one pack kernel per step.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.nn.functional as F

from . import _lib as L
from .mld_denoiser import _LAYER_FIELDS, timestep_features

# (forward matrix id in DenLayerOff order, offset of X / dY in the backward buffer, in-features K, out-features N)
_MATS = ("skip", "inp", "outp", "l1", "l2", "caq", "cao", "f1", "f2", "fo")


def _train_layout():
    n = 54
    buf = (C.c_int64 * n)()
    L.check(L.lib().seeme_den_train_layout(buf, n), "seeme_den_train_layout")
    v = list(buf)
    return {"bwd_total": v[50], "DT_TOTAL": v[51], "DB_TOTAL": v[52], "DB_LAYER": v[53]}


# offsets inside one layer block of the backward buffer (seeme_amd/csrc/den_train.h)
_DB = dict(X_INP=0, X_OUTP=256, X_L1=512, X_L2=768, X_CAQ=1792, X_CAO=2048, X_F1=2304, X_F2=2560, X_FO=2688, X_SKIP=2944,
           Y_INP=3456, Y_OUTP=4224, Y_L1=4480, Y_L2=5504, Y_CAQ=5760, Y_CAO=6016, Y_F1=6272, Y_F2=6400, Y_FO=6656,
           Y_SKIP=6912, LN=7168)
# name -> (X offset, K, dY offset, N)
_LIN = {"inp": ("X_INP", 256, "Y_INP", 768), "outp": ("X_OUTP", 256, "Y_OUTP", 256), "l1": ("X_L1", 256, "Y_L1", 1024),
        "l2": ("X_L2", 1024, "Y_L2", 256), "caq": ("X_CAQ", 256, "Y_CAQ", 256), "cao": ("X_CAO", 256, "Y_CAO", 256),
        "f1": ("X_F1", 256, "Y_F1", 128), "f2": ("X_F2", 128, "Y_F2", 256), "fo": ("X_FO", 256, "Y_FO", 256),
        "skip": ("X_SKIP", 512, "Y_SKIP", 256)}


def _layer_params(den, l):
    """The chain's parameters of layer l, keyed like den_layout.h."""
    blk = den.encoder.blocks()[l]
    sa, ca, ffn = blk.sa_block, blk.ca_block, blk.ffn
    p = {"inp": (sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias, "in_b"),
         "outp": (sa.self_attn.out_proj.weight, sa.self_attn.out_proj.bias, "out_b"),
         "l1": (sa.linear1.weight, sa.linear1.bias, "l1b"), "l2": (sa.linear2.weight, sa.linear2.bias, "l2b"),
         "caq": (ca.query.weight, ca.query.bias, "caq_b"),
         "cao": (ca.proj_out.out_layers[2].weight, ca.proj_out.out_layers[2].bias, "cao_b"),
         "f1": (ffn.linear1.weight, ffn.linear1.bias, "f1b"), "f2": (ffn.linear2.weight, ffn.linear2.bias, "f2b"),
         "fo": (ffn.proj_out.out_layers[2].weight, ffn.proj_out.out_layers[2].bias, "fo_b")}
    if l >= 3:
        lin = den.encoder.linear_blocks[l - 3]
        p["skip"] = (lin.weight, lin.bias, "skip_b")
    norms = [(sa.norm1, "n1w", "n1b"), (sa.norm2, "n2w", "n2b"), (ca.norm, "cnw", "cnb"),
             (ca.proj_out.norm, "csnw", "csnb"), (ffn.proj_out.norm, "fsnw", "fsnb")]
    return p, norms


class TrainPack:
    """Device images of the chain's parameters for one MldDenoiser (forward GEMV layout, transposed layout, vectors),
    refreshed by ``seeme_den_train_pack`` before every forward."""

    def __init__(self, den):
        if den.num_heads != 1:
            raise NotImplementedError("HIP training path: one attention head")
        self.den = den
        dev = den.query_pos.pe.device
        layers, pe0, fnw, fnb, wg_total, vp_total = den._layout()
        lay = _train_layout()
        self.lay = lay
        self.img_f = torch.zeros(wg_total, device=dev, dtype=torch.float32)
        self.img_b = torch.zeros(lay["bwd_total"], device=dev, dtype=torch.float32)
        self.vp = torch.zeros(vp_total, device=dev, dtype=torch.float32)
        mats: List[int] = []
        vecs = [(den.query_pos.pe, 256, pe0), (den.encoder.norm.weight, 256, fnw), (den.encoder.norm.bias, 256, fnb)]
        self.params: List[torch.nn.Parameter] = [den.query_pos.pe, den.encoder.norm.weight, den.encoder.norm.bias]
        self.index = {}            # (layer, name, 'w' | 'b' | 'nw' | 'nb') -> position in self.params
        for l in range(5):
            p, norms = _layer_params(den, l)
            for name in _MATS:
                if name in p:
                    w, b, bname = p[name]
                    mats.append(w.data_ptr())
                    vecs.append((b, b.numel(), layers[l][bname]))
                    self.index[(l, name, "w")] = len(self.params); self.params.append(w)
                    self.index[(l, name, "b")] = len(self.params); self.params.append(b)
                else:
                    mats.append(0)
            for i, (nm, wn, bn) in enumerate(norms):
                vecs.append((nm.weight, 256, layers[l][wn]))
                vecs.append((nm.bias, 256, layers[l][bn]))
                self.index[(l, i, "nw")] = len(self.params); self.params.append(nm.weight)
                self.index[(l, i, "nb")] = len(self.params); self.params.append(nm.bias)
        for t in self.params:
            L.require_cuda(t, "MldDenoiser parameter")
        self._ptrs = (C.c_void_p * len(mats))(*mats)
        self._vsrc = (C.c_void_p * len(vecs))(*[v[0].data_ptr() for v in vecs])
        self._vn = (C.c_int * len(vecs))(*[int(v[1]) for v in vecs])
        self._vdst = (C.c_int64 * len(vecs))(*[int(v[2]) for v in vecs])
        self._nvec = len(vecs)
        self._ptr_key = tuple(t.data_ptr() for t in self.params)
        w = L.DenoiserWeights()
        lay_dev = torch.tensor(den._layout_vals, dtype=torch.int64, device=dev)
        w.wg, w.wdtype, w.vp, w.layout = self.img_f.data_ptr(), 0, self.vp.data_ptr(), lay_dev.data_ptr()
        w.nhead, w.ff_sa, w.ff, w.sa_fold = 1, 1024, den.ff_size, 0
        self.w, self._keep = w, lay_dev
        self._build_flat_grads(dev)

    def _build_flat_grads(self, dev):
        """One flat buffer for every chain-parameter gradient: the five layers' matrices of one kind are a contiguous
        [L,N,K] block (target of one batched GEMM), every bias / LayerNorm gradient is one gather from the column sums
        of the backward buffer.  ``.grad`` of the parameters become views of it: ~13 device ops per backward instead
        of one autograd accumulation per parameter, and contiguous gradients for the optimiser."""
        DBL = self.lay["DB_LAYER"]
        fin = 5 * DBL
        self.mat_blocks = []                       # (name, layers, offset, N, K)
        self.grad_views = [None] * len(self.params)
        off = 0
        spans = []
        for name, (xo, K, yo, Nn) in _LIN.items():
            ls = [l for l in range(5) if (l, name, "w") in self.index]
            self.mat_blocks.append((name, ls, off, Nn, K))
            for i, l in enumerate(ls):
                spans.append((self.index[(l, name, "w")], off + i * Nn * K, (Nn, K)))
            off += len(ls) * Nn * K
        gather = []
        vec0 = off

        def vec(pos, src, n):
            nonlocal off
            spans.append((pos, off, (n,)))
            gather.extend(range(src, src + n))
            off += n

        for l in range(5):
            for name, (xo, K, yo, Nn) in _LIN.items():
                if (l, name, "b") in self.index:
                    vec(self.index[(l, name, "b")], l * DBL + _DB[yo], Nn)
            for i in range(5):
                o = l * DBL + _DB["LN"] + i * 512
                vec(self.index[(l, i, "nw")], o, 256)
                vec(self.index[(l, i, "nb")], o + 256, 256)
        vec(1, fin, 256)                           # encoder.norm.weight / bias
        vec(2, fin + 256, 256)
        self.gflat_numel, self._vec0, self._spans = off, vec0, spans
        # [end of the in_proj block, start of the vectors): no other part of the backward adds to these gradients (in_proj's K | V rows
        # and bias also collect the condition / time-table contributions) -- distributed.GradBucket exchanges it early
        name0, ls0, off0, Nn0, K0 = self.mat_blocks[0]
        assert name0 == "inp" and off0 == 0
        self.early_span = (len(ls0) * Nn0 * K0, vec0)
        self.bucket = None
        self.gather_idx = torch.tensor(gather, device=dev, dtype=torch.int64)
        self.dx0_span = (fin + 512, fin + 768)
        self.bound, self._attached = False, False
        self._attach(torch.zeros(off, device=dev, dtype=torch.float32), torch.zeros_like(self.params[0]))
        # tiles of seeme_den_wgrad: {int x_col, y_col, ldo, nn, kk, pad; int64 out_off} per 32 x 256 block of every matrix
        tiles = []
        for name, ls, moff, Nn, K in self.mat_blocks:
            xo, _, yo, _ = _LIN[name]
            for i, l in enumerate(ls):
                for n0 in range(0, Nn, 32):
                    for k0 in range(0, K, 256):
                        o = moff + (i * Nn + n0) * K + k0
                        tiles.append([l * DBL + _DB[xo] + k0, l * DBL + _DB[yo] + n0, K, min(32, Nn - n0), min(256, K - k0), 0,
                                      o & 0xFFFFFFFF if o < 2 ** 31 else o - 2 ** 32, o >> 32])
        self._wgrad_tiles = torch.tensor(tiles, dtype=torch.int32).to(dev)
        self._n_wgrad_tiles = len(tiles)

    def _attach(self, storage: torch.Tensor, dpe: torch.Tensor):
        """Point the gradient views at `storage` (gflat_numel floats) and `dpe` (query_pos.pe: only row 0 is on the path)."""
        self.gflat = storage
        self.vec_region = storage[self._vec0:self.gflat_numel]
        for pos, o, shape in self._spans:
            n = 1
            for d in shape:
                n *= d
            self.grad_views[pos] = storage[o:o + n].view(*shape)
        self.dpe = dpe
        self.grad_views[0] = dpe

    def bind(self, storage: torch.Tensor, dpe: torch.Tensor):
        """Write the chain's gradients straight into a caller-owned block (distributed.GradBucket): the parameters'
        ``.grad`` ARE these views, so reduce_into_grads overwrites in place and never touches ``.grad``."""
        assert storage.numel() == self.gflat_numel and storage.is_contiguous()
        self._attach(storage, dpe if dpe is not None else torch.zeros_like(self.params[0]))
        self.bound = True

    def reduce_into_grads(self, gout: torch.Tensor):
        """dW = sum_b dy_b x_b^T (one batched GEMM per matrix kind), bias / LayerNorm gradients = column sums; then
        hand the views to ``.grad`` (accumulating where a gradient already exists, e.g. the in_proj rows the tables share)."""
        B = gout.shape[0]
        DBL = self.lay["DB_LAYER"]
        if self.bound and gout.is_cuda:
            L.check(L.lib().seeme_den_wgrad(gout.data_ptr(), gout.shape[1], B, self._wgrad_tiles.data_ptr(), self._n_wgrad_tiles,
                                            self.gflat.data_ptr(), L.current_stream()), "seeme_den_wgrad")
            L.check(L.lib().seeme_den_vecgrad(gout.data_ptr(), gout.shape[1], B, self.gather_idx.data_ptr(), self.gather_idx.numel(),
                                              self.vec_region.data_ptr(), self.dx0_span[0], self.dpe.data_ptr(), L.current_stream()),
                    "seeme_den_vecgrad")
            if self._attached:                                 # GradBucket.prepare() made .grad these very views
                self._attached = False
                if self.bucket is not None:
                    self.bucket.allreduce_early()              # the matrices other than in_proj are final: exchange them now
                return
            for p, v in zip(self.params, self.grad_views):     # a backward outside the bucket's step: same contract as below
                if p.requires_grad:
                    if p.grad is None:
                        p.grad = v
                    elif p.grad.data_ptr() != v.data_ptr():
                        p.grad.add_(v)
            return
        live = any(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.grad_views))
        flat = torch.empty_like(self.gflat) if live else self.gflat   # a previous backward's gradients are still in use
        colsum = gout.sum(0)
        G = gout[:, : 5 * DBL].view(B, 5, DBL)
        if gout.is_cuda:
            L.check(L.lib().seeme_den_wgrad(gout.data_ptr(), gout.shape[1], B, self._wgrad_tiles.data_ptr(), self._n_wgrad_tiles,
                                            flat.data_ptr(), L.current_stream()), "seeme_den_wgrad")
        else:                                                                         # (CPU: shape tests of the host logic)
            for name, ls, off, Nn, K in self.mat_blocks:
                xo, _, yo, _ = _LIN[name]
                Y = G[:, ls[0]:ls[-1] + 1, _DB[yo]:_DB[yo] + Nn].permute(1, 2, 0)    # [L,N,B]
                X = G[:, ls[0]:ls[-1] + 1, _DB[xo]:_DB[xo] + K].permute(1, 0, 2)     # [L,B,K]
                torch.bmm(Y, X, out=flat[off:off + len(ls) * Nn * K].view(len(ls), Nn, K))
        v0 = self.vec_region.data_ptr() - self.gflat.data_ptr()
        flat[v0 // 4:].copy_(colsum.index_select(0, self.gather_idx))
        dpe = self.dpe if not live else torch.zeros_like(self.dpe)
        dpe[0, 0].copy_(colsum[self.dx0_span[0]:self.dx0_span[1]])
        base = self.gflat.data_ptr()
        for p, v in zip(self.params, self.grad_views):
            if not p.requires_grad:
                continue
            g = v if flat is self.gflat else (dpe if v is self.dpe else flat[(v.data_ptr() - base) // 4:(v.data_ptr() - base) // 4 + v.numel()].view_as(v))
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)

    def stale(self) -> bool:
        return tuple(t.data_ptr() for t in self.params) != self._ptr_key

    def refresh(self):
        L.check(L.lib().seeme_den_train_pack(self._ptrs, self.img_f.data_ptr(), self.img_b.data_ptr(), self._vsrc, self._vn,
                                             self._vdst, self._nvec, self.vp.data_ptr(), L.current_stream()), "seeme_den_train_pack")


DROP_BYTES = 10960          # SEEME_DEN_DROP_BYTES: 5 layers x 2192 mask bytes per sample (csrc/den_train.h DM_*)


def draw_dropout_masks(den, B: int, out: torch.Tensor = None):
    """Keep-masks of the MD layers' dropout sites for one training forward (uint8 [B, DROP_BYTES], 1 = keep), or None in eval mode
    / with p = 0.  The reference applies nn.Dropout(p) at these sites whenever the module is in training mode
    (mdiff_transformer.py:137-165,241-254; cross_attention.py:264-273).  Returns (masks, 1 / (1 - p))."""
    p = float(den.encoder.blocks()[0].sa_block.self_attn.dropout)
    if not den.training or p <= 0.0:
        return None, 1.0
    m = out if out is not None else torch.empty(B, DROP_BYTES, dtype=torch.uint8, device=den.query_pos.pe.device)
    m.bernoulli_(1.0 - p)
    return m, 1.0 / (1.0 - p)


def _tables(den, cond_sf: torch.Tensor, emb: torch.Tensor):
    """Differentiable table builders: ctab [B,N,5120], ttab [B,7680] in the layouts of include/seeme_hip.h."""
    blocks = den.encoder.blocks()
    cond = cond_sf.permute(1, 0, 2)                                                  # [B,N,256]
    kv_w = torch.cat([b.sa_block.self_attn.in_proj_weight[256:] for b in blocks])    # [2560,256]: K|V per layer
    kv_b = torch.cat([b.sa_block.self_attn.in_proj_bias[256:] for b in blocks])
    sa_c = F.linear(cond, kv_w, kv_b)                                                # [B,N,2560]
    # text_norm of the five layers shares its statistics (only the affine part is per layer), and the five key|value
    # projections are one batched GEMM: per layer they were 5 + 10 small GEMM launches of ~60 us each in forward + backward
    Bc, Nc = cond.shape[0], cond.shape[1]
    xhat = F.layer_norm(cond, (256,))                                                # [B,N,256]
    tn_w = torch.stack([b.ca_block.text_norm.weight for b in blocks])                # [5,256]
    tn_b = torch.stack([b.ca_block.text_norm.bias for b in blocks])
    xl = (xhat.reshape(1, Bc * Nc, 256) * tn_w[:, None, :] + tn_b[:, None, :])       # [5,B*N,256]
    # (flat concatenations: one copy kernel per torch.cat call, nested cats were 33 launches per step)
    ca_w = torch.cat([w for b in blocks for w in (b.ca_block.key.weight, b.ca_block.value.weight)]).view(5, 512, 256)
    ca_b = torch.cat([w for b in blocks for w in (b.ca_block.key.bias, b.ca_block.value.bias)]).view(5, 512)
    ca_c = torch.baddbmm(ca_b[:, None, :], xl, ca_w.transpose(1, 2))                 # [5,B*N,512]
    ca_c = ca_c.permute(1, 0, 2).reshape(Bc, Nc, 5 * 512)
    ctab = torch.cat([sa_c, ca_c], dim=-1).contiguous()
    st_w = torch.cat([w for b in blocks for w in (b.ca_block.proj_out.emb_layers[1].weight, b.ffn.proj_out.emb_layers[1].weight)])
    st_b = torch.cat([w for b in blocks for w in (b.ca_block.proj_out.emb_layers[1].bias, b.ffn.proj_out.emb_layers[1].bias)])
    ttab = torch.cat([F.linear(emb, kv_w, kv_b), F.linear(F.silu(emb), st_w, st_b)], dim=-1).contiguous()
    return ctab, ttab


class _Chain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack: TrainPack, noisy, ctab, ttab):
        B, N = noisy.shape[0], ctab.shape[1]
        dev = noisy.device
        pack.refresh()
        lay = pack.lay
        save = torch.empty(B, lay["DT_TOTAL"], device=dev, dtype=torch.float32)
        out = torch.empty(B, 256, device=dev, dtype=torch.float32)
        trow = torch.arange(B, device=dev, dtype=torch.int32)
        a = L.SampleArgs()
        a.B, a.N, a.steps, a.sched, a.cfg, a.guidance_scale = B, N, 1, L.SCHED_NONE, 0, 1.0
        lat = noisy.contiguous()
        a.latents, a.ctab, a.ttab, a.trow, a.trow_per_sample = lat.data_ptr(), ctab.data_ptr(), ttab.data_ptr(), trow.data_ptr(), 1
        a.coef, a.noise, a.out, a.catab = 0, 0, out.data_ptr(), 0
        a.save, a.force_query = save.data_ptr(), 1
        masks, scale = draw_dropout_masks(pack.den, B)
        a.drop, a.drop_scale = L.ptr(masks), scale
        a.xcds = L.default_xcds(B)
        L.check(L.lib().seeme_denoiser_sample(C.byref(pack.w), C.byref(a), L.current_stream()), "seeme_denoiser_sample")
        ctx.pack, ctx.N, ctx.masks, ctx.drop_scale = pack, N, masks, scale
        pack.last_masks = masks
        ctx.save_for_backward(save, ctab, ttab, trow)
        return out

    @staticmethod
    def backward(ctx, dout):
        pack = ctx.pack
        save, ctab, ttab, trow = ctx.saved_tensors
        B, N, dev = save.shape[0], ctx.N, save.device
        lay = pack.lay
        gout = torch.zeros(B, lay["DB_TOTAL"], device=dev, dtype=torch.float32)
        dctab, dttab = torch.empty_like(ctab), torch.empty_like(ttab)
        L.check(L.lib().seeme_denoiser_backward_drop(C.byref(pack.w), pack.img_b.data_ptr(), B, N, save.data_ptr(), ctab.data_ptr(),
                                                      ttab.data_ptr(), trow.data_ptr(), dout.contiguous().data_ptr(), gout.data_ptr(),
                                                      dctab.data_ptr(), dttab.data_ptr(), L.ptr(ctx.masks), ctx.drop_scale,
                                                      L.default_xcds(B), L.current_stream()), "seeme_denoiser_backward")
        # the chain's own parameters do not travel through autograd: their gradients are reduced straight into .grad
        pack.reduce_into_grads(gout)
        fin = 5 * lay["DB_LAYER"]
        dnoisy = gout[:, fin + 512:fin + 768] if ctx.needs_input_grad[1] else None
        return None, dnoisy, dctab, dttab


def hip_train_supported(den, n_tokens: int) -> bool:
    return den.num_heads == 1 and 1 <= n_tokens <= 4 and den.ff_size == 128 and den.query_pos.pe.is_cuda


def denoiser_forward_hip_train(den, sample: torch.Tensor, timesteps: torch.Tensor, cond_sf: torch.Tensor) -> torch.Tensor:
    """sample [B,1,256]; timesteps [B]; cond_sf [N,B,256] (seq-first) -> noise prediction [B,1,256], differentiable
    w.r.t. every denoiser parameter and the condition tokens."""
    pack = getattr(den, "_train_pack", None)
    if pack is None or pack.stale():
        pack = TrainPack(den)
        den._train_pack = pack
    B = sample.shape[0]
    tfeat = timestep_features(timesteps.to(sample.device), den.text_encoded_dim, den.flip_sin_to_cos, den.freq_shift).to(torch.float32)
    te = den.time_embedding                                                 # TimestepEmbedding: Linear -> SiLU -> Linear
    emb = F.linear(F.silu(F.linear(tfeat, te.linear_1.weight, te.linear_1.bias)), te.linear_2.weight, te.linear_2.bias)   # [B,256]
    ctab, ttab = _tables(den, cond_sf, emb)
    out = _Chain.apply(pack, sample.reshape(B, 256), ctab, ttab)
    return out.reshape(B, 1, 256)
