"""Stage-2 training step around the denoiser chain as hand-written HIP, forward and backward.

What ``MLD.train_diffusion_forward`` / ``_diffusion_process`` (mld/models/modeltype/mld.py:582-631,887-1017) do between the
frozen encoders and the loss -- posterior rsample, ``scheduler.add_noise``, timestep features + TimestepEmbedding MLP,
``output_scene``, ``MldDenoiser.forward``'s condition / time tables, the token-0 chain -- runs here as

    forward : k_glue_rows -> k_gg (time MLP layer 1, output_scene) -> k_glue_ln -> k_gg (time MLP layer 2)
              -> k_gg (all tables) -> k_den_sample (chain, intermediates saved)
    backward: k_den_bwd -> seeme_den_wgrad (chain weights) -> k_gg (data gradients of the tables) -> k_glue_mid
              -> k_gg (time MLP) -> k_gg (every weight / bias gradient of the tables, time MLP and output_scene)

with every parameter gradient accumulated straight into ``.grad`` (the views of ``distributed.GradBucket`` once it exists; the
weight-gradient descriptor table is rebuilt when those tensors move, so keep them -- ``MLD.optimizer_step`` does):
autograd sees ONE node (:class:`_Stage2`) instead of ~70 and launches ~12 kernels instead of ~170.  The torch-autograd
table builders (``denoiser_train._tables``) stay as the fallback for shapes this path does not take, and are the oracle
of ``tests/test_gpu_flows.py::test_stage2_glue_matches_autograd_path``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import torch

from . import _lib as L
from .denoiser_train import DROP_BYTES, TrainPack, draw_dropout_masks, hip_train_supported


def _tiles(M, N):
    return ((M + 63) // 64) * ((N + 63) // 64), (N + 63) // 64


class _Group:
    """A launch of seeme_grouped_gemm: the descriptor table on the device."""

    def __init__(self, probs: List[L.GemmProblem], dev):
        t0 = 0
        for p in probs:
            n, tn = _tiles(p.M, p.N)
            n *= max(1, p.nbatch)
            p.tile0, p.tiles_n = t0, tn
            if p.colsum and (p.a_rs != 1 or p.a_pro):
                raise ValueError("colsum needs an i-contiguous A without a prologue")
            t0 += n
        arr = (L.GemmProblem * len(probs))(*probs)
        assert C.sizeof(L.GemmProblem) == L.lib().seeme_gemm_problem_bytes(), "GemmProblem layout mismatch with the library"
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = host.to(dev)
        self.n, self.tiles = len(probs), t0

    def launch(self):
        L.check(L.lib().seeme_grouped_gemm(self.dev.data_ptr(), self.n, self.tiles, L.current_stream()), "seeme_grouped_gemm")


def _prob(a, b, seg_len, a_ks, b_ks, a_rs, b_cs, c, ldc, M, N, *, a_pro=0, a_p=(0, 0), b_pro=0, b_p=(0, 0), bias=0, epi=0,
          e0=0, e_ld=0, accumulate=0, colsum=0, nbatch=1, bstrides=(0, 0, 0), alpha=1.0, addend=0, add_ld=0) -> L.GemmProblem:
    p = L.GemmProblem()
    n = len(a)
    for s in range(n):
        p.a[s], p.b[s], p.seg_len[s], p.a_ks[s], p.b_ks[s] = a[s], b[s], seg_len[s], a_ks[s], b_ks[s]
    p.nseg, p.a_rs, p.b_cs, p.c, p.ldc, p.M, p.N = n, a_rs, b_cs, c, ldc, M, N
    p.a_pro, p.a_p0, p.a_p1, p.b_pro, p.b_p0, p.b_p1 = a_pro, a_p[0], a_p[1], b_pro, b_p[0], b_p[1]
    p.bias, p.epi, p.e0, p.e_ld, p.accumulate, p.colsum = bias, epi, e0, e_ld, accumulate, colsum
    p.nbatch, p.a_bstride, p.b_bstride, p.c_bstride, p.alpha = nbatch, bstrides[0], bstrides[1], bstrides[2], alpha
    p.addend, p.add_ld = addend, add_ld
    return p


def _fwd(x, ldx, w, K, bias, c, ldc, M, N, **kw):
    """C[M,N] = pro(X[M,K]) W[N,K]^T + bias."""
    return _prob([x], [w], [K], [1], [1], ldx, K, c, ldc, M, N, bias=bias, **kw)


def _dgrad(dy, ldy, ws, seg, c, M, Kin, **kw):
    """C[M,Kin] = sum_s dY[:, s*seg:(s+1)*seg] W_s[seg,Kin]."""
    return _prob([dy + 4 * seg * s for s in range(len(ws))], ws, [seg] * len(ws), [1] * len(ws), [Kin] * len(ws), ldy, 1,
                 c, Kin, M, Kin, **kw)


def _wgrad(srcs, g, Nout, Kin, gbias, **kw):
    """g[Nout,Kin] += sum over sources (dY ptr, ldy, X ptr, ldx, rows) of dY^T X; gbias[Nout] += column sums of dY."""
    return _prob([s[0] for s in srcs], [s[2] for s in srcs], [s[4] for s in srcs], [s[1] for s in srcs], [s[3] for s in srcs],
                 1, 1, g, Kin, Nout, Kin, accumulate=1, colsum=gbias, **kw)


class _Plan:
    """Buffers and descriptor tables for one (B, token layout)."""

    def __init__(self, glue: "Stage2Glue", B: int, has_int: bool, has_scene: bool):
        den, dev = glue.den, glue.dev
        self.B, self.has_int, self.has_scene = B, has_int, has_scene
        N = int(has_int) + int(has_scene)
        self.N, self.M = N, B * N
        M = self.M
        self.slot_c, self.slot_s = 0, int(has_int)
        lay = glue.pack.lay
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.cond, self.tfeat, self.pre1, self.emb = z(B, N, 256), z(B, 256), z(B, 256), z(B, 256)
        self.noisy = z(B, 256)
        self.xhat, self.rstd = z(M, 256), z(M)
        self.ctab, self.ttab = z(B, N, 5120), z(B, 7680)
        self.s512 = z(B, 512) if has_scene else None
        self.save = z(B, lay["DT_TOTAL"])
        self.trow = torch.arange(B, device=dev, dtype=torch.int32)
        self.gout = z(B, lay["DB_TOTAL"])
        self.dctab, self.dttab = z(B, N, 5120), z(B, 7680)
        self.dxl, self.dcs, self.dea, self.deb = z(5, M, 256), z(5, M, 256), z(5, B, 256), z(10, B, 256)
        self.dcond, self.demb, self.dpre1 = z(M, 256), z(B, 256), z(B, 256)
        self.busy = False
        self._bwd_key = None
        self.drop = torch.zeros(B, DROP_BYTES, dtype=torch.uint8, device=dev)      # dropout keep-masks of the chain (training mode)
        self.masks, self.drop_scale = None, 1.0
        blocks = den.encoder.blocks()
        te = den.time_embedding
        P = lambda t: t.data_ptr()
        kvw = [P(b.sa_block.self_attn.in_proj_weight) + 4 * 256 * 256 for b in blocks]
        kvb = [P(b.sa_block.self_attn.in_proj_bias) + 4 * 256 for b in blocks]
        l0 = [_fwd(P(self.tfeat), 256, P(te.linear_1.weight), 256, P(te.linear_1.bias), P(self.pre1), 256, B, 256)]
        if has_scene:
            lin = glue.output_scene[1]
            l0.append(_fwd(P(self.s512), 512, P(lin.weight), 512, P(lin.bias), P(self.cond) + 4 * 256 * self.slot_s, N * 256, B, 256,
                           a_pro=2))
        self.g_l0 = _Group(l0, dev)
        self.g_l1 = _Group([_fwd(P(self.pre1), 256, P(te.linear_2.weight), 256, P(te.linear_2.bias), P(self.emb), 256, B, 256, a_pro=1)], dev)
        l2 = []
        for l, b in enumerate(blocks):
            ca = b.ca_block
            l2.append(_fwd(P(self.cond), 256, kvw[l], 256, kvb[l], P(self.ctab) + 4 * 512 * l, 5120, M, 512))
            for h, lin in enumerate((ca.key, ca.value)):
                l2.append(_fwd(P(self.xhat), 256, P(lin.weight), 256, P(lin.bias), P(self.ctab) + 4 * (2560 + 512 * l + 256 * h), 5120, M, 256,
                               a_pro=3, a_p=(P(ca.text_norm.weight), P(ca.text_norm.bias))))
            l2.append(_fwd(P(self.emb), 256, kvw[l], 256, kvb[l], P(self.ttab) + 4 * 512 * l, 7680, B, 512))
            for h, lin in enumerate((ca.proj_out.emb_layers[1], b.ffn.proj_out.emb_layers[1])):
                l2.append(_fwd(P(self.emb), 256, P(lin.weight), 256, P(lin.bias), P(self.ttab) + 4 * (2560 + 512 * (2 * l + h)), 7680, B, 512,
                               a_pro=1))
        self.g_l2 = _Group(l2, dev)
        # data gradients of the tables
        b0 = []
        for l, b in enumerate(blocks):
            ca = b.ca_block
            b0.append(_dgrad(P(self.dctab) + 4 * (2560 + 512 * l), 5120, [P(ca.key.weight), P(ca.value.weight)], 256,
                             P(self.dxl) + 4 * l * M * 256, M, 256))
            b0.append(_dgrad(P(self.dctab) + 4 * 512 * l, 5120, [kvw[l]], 512, P(self.dcs) + 4 * l * M * 256, M, 256))
            b0.append(_dgrad(P(self.dttab) + 4 * 512 * l, 7680, [kvw[l]], 512, P(self.dea) + 4 * l * B * 256, B, 256))
            for h, lin in enumerate((ca.proj_out.emb_layers[1], b.ffn.proj_out.emb_layers[1])):
                i = 2 * l + h
                b0.append(_dgrad(P(self.dttab) + 4 * (2560 + 512 * i), 7680, [P(lin.weight)], 512, P(self.deb) + 4 * i * B * 256, B, 256))
        self.g_b0 = _Group(b0, dev)
        self.g_b2 = _Group([_dgrad(P(self.demb), 256, [P(te.linear_2.weight)], 256, P(self.dpre1), B, 256, epi=1, e0=P(self.pre1), e_ld=256)], dev)
        self.g_b3 = None
        self.mid = None

    def bind_grads(self, glue: "Stage2Glue"):
        """(Re)build the launches that write parameter gradients when the ``.grad`` tensors moved."""
        params = glue.params
        for p in params:
            if p.grad is None:           # first step, before the bucket exists: what AccumulateGrad would allocate
                p.grad = torch.zeros_like(p)
        key = tuple(p.grad.data_ptr() for p in params)
        if key == self._bwd_key:
            return
        self._bwd_key = key
        den, B, M, N = glue.den, self.B, self.M, self.N
        P = lambda t: t.data_ptr()
        G = lambda t: t.grad.data_ptr()
        te = den.time_embedding
        b3 = []
        mid = L.GlueMid()
        for l, b in enumerate(den.encoder.blocks()):
            ca, sa = b.ca_block, b.sa_block.self_attn
            b3.append(_wgrad([(P(self.dctab) + 4 * 512 * l, 5120, P(self.cond), 256, M), (P(self.dttab) + 4 * 512 * l, 7680, P(self.emb), 256, B)],
                             G(sa.in_proj_weight) + 4 * 256 * 256, 512, 256, G(sa.in_proj_bias) + 4 * 256))
            for h, lin in enumerate((ca.key, ca.value)):
                b3.append(_wgrad([(P(self.dctab) + 4 * (2560 + 512 * l + 256 * h), 5120, P(self.xhat), 256, M)], G(lin.weight), 256, 256,
                                 G(lin.bias), b_pro=3, b_p=(P(ca.text_norm.weight), P(ca.text_norm.bias))))
            for h, lin in enumerate((ca.proj_out.emb_layers[1], b.ffn.proj_out.emb_layers[1])):
                b3.append(_wgrad([(P(self.dttab) + 4 * (2560 + 512 * (2 * l + h)), 7680, P(self.emb), 256, B)], G(lin.weight), 512, 256,
                                 G(lin.bias), b_pro=1))
            mid.tn_w[l], mid.g_tn_w[l], mid.g_tn_b[l] = P(ca.text_norm.weight), G(ca.text_norm.weight), G(ca.text_norm.bias)
        b3.append(_wgrad([(P(self.demb), 256, P(self.pre1), 256, B)], G(te.linear_2.weight), 256, 256, G(te.linear_2.bias), b_pro=1))
        b3.append(_wgrad([(P(self.dpre1), 256, P(self.tfeat), 256, B)], G(te.linear_1.weight), 256, 256, G(te.linear_1.bias)))
        if self.has_scene:
            lin = glue.output_scene[1]
            b3.append(_wgrad([(P(self.dcond) + 4 * 256 * self.slot_s, N * 256, P(self.s512), 512, B)], G(lin.weight), 256, 512, G(lin.bias),
                             b_pro=2))
        self.g_b3 = _Group(b3, glue.dev)
        mid.M, mid.B = M, B
        mid.dxl, mid.dcs, mid.xhat, mid.rstd, mid.dcond = P(self.dxl), P(self.dcs), P(self.xhat), P(self.rstd), P(self.dcond)
        mid.dea, mid.deb, mid.emb, mid.demb = P(self.dea), P(self.deb), P(self.emb), P(self.demb)
        self.mid = mid


class _Stage2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, glue, hook, dist, eps_z, eps_c, noise, timesteps, s512):
        plan = glue._forward(dist, eps_z, eps_c, noise, timesteps, s512)
        ctx.glue, ctx.plan = glue, plan
        out, latents = plan.out, plan.latents
        ctx.mark_non_differentiable(latents)
        return out, latents

    @staticmethod
    def backward(ctx, dout, _dlat):
        ctx.glue._backward(ctx.plan, dout)
        return (None,) * 8


class Stage2Glue:
    """Owner of the plans of one MLD model (denoiser + optional output_scene + noise scheduler)."""

    def __init__(self, mld):
        den = mld.denoiser
        self.den = den
        self.dev = den.query_pos.pe.device
        self.output_scene = getattr(mld, "output_scene", None)
        self.scheduler = mld.noise_scheduler
        self.acp = self.scheduler.alphas_cumprod.to(self.dev, torch.float32).contiguous()
        half = den.text_encoded_dim // 2
        ex = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=self.dev) / (half - den.freq_shift)
        self.freq = torch.exp(ex).contiguous()           # the values timestep_features multiplies t with (embeddings.py:262-268)
        self.flip = int(bool(den.flip_sin_to_cos))
        self.plans = {}
        self._pack()
        blocks = den.encoder.blocks()
        te = den.time_embedding
        ps = [te.linear_1.weight, te.linear_1.bias, te.linear_2.weight, te.linear_2.bias]
        for b in blocks:
            ca = b.ca_block
            ps += [b.sa_block.self_attn.in_proj_weight, b.sa_block.self_attn.in_proj_bias, ca.key.weight, ca.key.bias, ca.value.weight,
                   ca.value.bias, ca.text_norm.weight, ca.text_norm.bias, ca.proj_out.emb_layers[1].weight, ca.proj_out.emb_layers[1].bias,
                   b.ffn.proj_out.emb_layers[1].weight, b.ffn.proj_out.emb_layers[1].bias]
        if self.output_scene is not None:
            ps += [self.output_scene[1].weight, self.output_scene[1].bias]
        self.params = ps
        self._param_key = tuple(p.data_ptr() for p in ps)

    def _pack(self):
        pack = getattr(self.den, "_train_pack", None)
        if pack is None or pack.stale():
            pack = TrainPack(self.den)
            self.den._train_pack = pack
            self.plans = {}
        self.pack = pack

    @staticmethod
    def supported(mld, n_tokens: int) -> bool:
        den = mld.denoiser
        if not hip_train_supported(den, n_tokens) or den.text_encoded_dim != 256:
            return False
        ps = list(den.parameters()) + (list(mld.output_scene.parameters()) if getattr(mld, "output_scene", None) is not None else [])
        return all(p.requires_grad and p.dtype == torch.float32 for p in ps)

    def stale(self) -> bool:
        return tuple(p.data_ptr() for p in self.params) != self._param_key

    def __call__(self, dist, eps_z, eps_c, noise, timesteps, s512):
        """dist [2,R,256] (R = B, or 2B with the condition motion in rows B..); eps_z / eps_c / noise [B,256]-shaped; timesteps
        [B] int64; s512 [B,512] PointNet code or None.  Returns (noise_pred [B,1,256] differentiable, latents [B,1,256])."""
        out, lat = _Stage2.apply(self, self.params[0], dist, eps_z, eps_c, noise, timesteps, s512)
        B = out.shape[0]
        return out.view(B, 1, 256), lat.view(B, 1, 256)

    # ------------------------------------------------------------------ forward / backward bodies
    def _forward(self, dist, eps_z, eps_c, noise, timesteps, s512) -> _Plan:
        self._pack()
        B = noise.reshape(-1, 256).shape[0]
        has_int, has_scene = eps_c is not None, s512 is not None
        key = (B, has_int, has_scene)
        plan = self.plans.get(key)
        if plan is None or plan.busy:        # busy: a second forward before the first one's backward (keep its saved tensors intact)
            plan = _Plan(self, B, has_int, has_scene)
            self.plans[key] = plan
        dev = self.dev
        st = L.current_stream()
        for name, t in (("dist", dist), ("eps_z", eps_z), ("noise", noise)):
            L.require_cuda(t, name)
        dist = dist.contiguous().float()
        if dist.shape[0] != 2 or dist.shape[1] < (2 * B if has_int else B) or dist.shape[2] != 256:
            raise ValueError(f"dist is {tuple(dist.shape)}: expected [2, {'2B' if has_int else 'B'}, 256]")
        eps_z, noise = eps_z.contiguous().float(), noise.contiguous().float()
        timesteps = timesteps.to(dev, torch.int64).contiguous()
        plan.out = torch.empty(B, 256, device=dev, dtype=torch.float32)
        plan.latents = torch.empty(B, 256, device=dev, dtype=torch.float32)
        if has_scene:
            plan.s512.copy_(s512.reshape(B, 512))
        a = L.GlueRows()
        a.B, a.N, a.dist, a.dist_rows, a.eps_z = B, plan.N, dist.data_ptr(), dist.shape[1], eps_z.data_ptr()
        if has_int:
            eps_c = eps_c.contiguous().float()
            a.eps_c, a.slot_c, a.cond = eps_c.data_ptr(), plan.slot_c, plan.cond.data_ptr()
        a.noise, a.timesteps, a.acp, a.freq, a.flip_sin_to_cos = noise.data_ptr(), timesteps.data_ptr(), self.acp.data_ptr(), self.freq.data_ptr(), self.flip
        a.latents, a.noisy, a.tfeat = plan.latents.data_ptr(), plan.noisy.data_ptr(), plan.tfeat.data_ptr()
        L.check(L.lib().seeme_glue_rows(C.byref(a), st), "seeme_glue_rows")
        plan.g_l0.launch()
        L.check(L.lib().seeme_glue_ln(plan.cond.data_ptr(), plan.xhat.data_ptr(), plan.rstd.data_ptr(), plan.M, st), "seeme_glue_ln")
        plan.g_l1.launch()
        plan.g_l2.launch()
        pack = self.pack
        pack.refresh()
        s = L.SampleArgs()
        s.B, s.N, s.steps, s.sched, s.cfg, s.guidance_scale = B, plan.N, 1, L.SCHED_NONE, 0, 1.0
        s.latents, s.ctab, s.ttab, s.trow, s.trow_per_sample = plan.noisy.data_ptr(), plan.ctab.data_ptr(), plan.ttab.data_ptr(), plan.trow.data_ptr(), 1
        s.coef, s.noise, s.out, s.catab = 0, 0, plan.out.data_ptr(), 0
        s.save, s.force_query = plan.save.data_ptr(), 1
        plan.masks, plan.drop_scale = draw_dropout_masks(self.den, B, out=plan.drop)
        s.drop, s.drop_scale = L.ptr(plan.masks), plan.drop_scale
        s.xcds = L.default_xcds(B)
        L.check(L.lib().seeme_denoiser_sample(C.byref(pack.w), C.byref(s), st), "seeme_denoiser_sample")
        plan.busy = True
        plan._keep = (dist, eps_z, eps_c, noise, timesteps)
        return plan

    def _backward(self, plan: _Plan, dout: torch.Tensor):
        pack, st = self.pack, L.current_stream()
        B, N = plan.B, plan.N
        plan.gout.zero_()
        L.check(L.lib().seeme_denoiser_backward_drop(C.byref(pack.w), pack.img_b.data_ptr(), B, N, plan.save.data_ptr(), plan.ctab.data_ptr(),
                                                      plan.ttab.data_ptr(), plan.trow.data_ptr(), dout.contiguous().data_ptr(),
                                                      plan.gout.data_ptr(), plan.dctab.data_ptr(), plan.dttab.data_ptr(),
                                                      L.ptr(plan.masks), plan.drop_scale, L.default_xcds(B), st), "seeme_denoiser_backward")
        pack.reduce_into_grads(plan.gout)            # chain weights (overwrites its block of the bucket) -- BEFORE the accumulations below
        plan.bind_grads(self)
        plan.g_b0.launch()
        L.check(L.lib().seeme_glue_mid(C.byref(plan.mid), st), "seeme_glue_mid")
        plan.g_b2.launch()
        plan.g_b3.launch()
        plan.busy = False
        plan._keep = None
