"""Stage-1 (VAE) training through hand-written HIP: forward with saved intermediates and backward of
``MldVae.encode`` / ``decode`` (reference: mld_vae.py:128-256; skip stacks and post-norm layers of
cross_attention.py:41-147,281-367), i.e. what ``loss.backward()`` walks for ``train_vae_forward`` (mld.py:633-885).

Two autograd nodes, :class:`_Encode` (features -> dist [2,B,256]) and :class:`_Decode` (z -> features), so that the
reparameterisation, the KL / reconstruction losses and the SMPL joint regressor stay ordinary torch code between and after
them.  Inside a node everything is launches of ``libseeme_hip.so``:

* every GEMM -- projections, per-sequence ``Q K^T`` and ``P V`` (batched problems), their data gradients, and all weight
  gradients as split reductions over the sequences (atomic accumulation) with the bias gradient as a free column sum --
  is a problem of the grouped fp32 GEMM ``seeme_grouped_gemm`` (descriptor tables built once per batch shape);
* ``seeme_vt_add_ln`` / ``seeme_vt_ln_bwd`` (residual + LayerNorm and its backward), ``seeme_vt_softmax_fwd/bwd`` (masked
  softmax over the keys), ``seeme_vt_gelu`` (exact GELU and its derivative), ``seeme_vt_seq_sum`` (the cross-attention
  vector of the single latent token is one row per sequence).

Parameter gradients are accumulated straight into ``.grad`` (the views of ``distributed.GradBucket`` once it exists).  The
backward's launch lists hold the addresses of those ``.grad`` tensors: ``MLD.optimizer_step`` keeps them stable (the bucket's
views); a loop that drops them every step (``zero_grad(set_to_none=True)``) stays correct but re-records ~130 descriptor tables
per step -- zero them in place instead.
In training mode the reference's dropout sites are applied (attention weights of ``nn.MultiheadAttention``, ``dropout1/2/3``, the
FFN's inner dropout; cross_attention.py:264-273,324-337) with keep-masks drawn by one ``bernoulli_`` per stack and step; in eval
mode the arithmetic is that of the HIP inference path.  The autograd twin (``vae_autograd.py``) stays as the fallback (more than
one head, S > 512) and as the oracle of ``tests/test_gpu_flows.py::test_vae_hip_backward_matches_autograd`` (with the same masks).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List

import torch

from . import _lib as L
from .stage2_glue import _Group, _prob


def P(t) -> int:
    return t.data_ptr()


def _gemm_fwd(xs, ldx, w, ldw, Ks, bias, c, ldc, M, N, **kw):
    """C[M,N] = sum_s X_s[M,K_s] W[:, off_s : off_s+K_s]^T + bias  (W row stride ldw; the column blocks follow each other)."""
    offs, o = [], 0
    for k in Ks:
        offs.append(o)
        o += k
    return _prob(list(xs), [w + 4 * o for o in offs], list(Ks), [1] * len(xs), [1] * len(xs), ldx, ldw, c, ldc, M, N, bias=bias, **kw)


def _gemm_dgrad(dy, ldy, w, ldw, Nout, Kin, c, ldc, M, **kw):
    """C[M,Kin] = dY[M,Nout] W[Nout, 0:Kin]  (W row stride ldw)."""
    return _prob([dy], [w], [Nout], [1], [ldw], ldy, 1, c, ldc, M, Kin, **kw)


def _gemm_wgrad(dy, ldy, x, ldx, rows, nbatch, g, ldg, Nout, Kin, gbias=0):
    """g[Nout,Kin] += sum over nbatch chunks of `rows` rows of dY^T X (atomic); gbias[Nout] += column sums of dY."""
    return _prob([dy], [x], [rows], [ldy], [ldx], 1, 1, g, ldg, Nout, Kin, accumulate=2, colsum=gbias, nbatch=nbatch,
                 bstrides=(rows * ldy, rows * ldx, 0))


class _Ops:
    """A recorded sequence of launches."""

    def __init__(self, dev):
        self.dev, self.ops, self._pending = dev, [], []

    def gemm(self, *probs):
        g = _Group(list(probs), self.dev)
        self.ops.append(g.launch)

    def plain(self, A, lda, B, ldb, nt, Cp, ldc, M, N, K, bias=0, addend=0, add_ld=0):
        """C[M,N] = A[M,K] B + bias + addend, B = W[N,K] (nt: y = x W^T) or W[K,N] (data gradient): the specialised 128 x 128 kernel
        when the shape allows (M, N multiples of 128, K of 32), else a problem of the grouped GEMM."""
        if M % 128 == 0 and N % 128 == 0 and K % 32 == 0 and lda % 4 == 0 and ldb % 4 == 0 and A % 16 == 0 and B % 16 == 0 \
                and os.environ.get("SEEME_GEMM128", "1") != "0":
            self.ops.append(lambda: L.check(L.lib().seeme_gemm128(A, lda, B, ldb, 1 if nt else 0, Cp, ldc, M, N, K, bias, addend, add_ld,
                                                                   L.current_stream()), "seeme_gemm128"))
        elif nt:
            self.gemm(_gemm_fwd([A], lda, B, ldb, [K], bias, Cp, ldc, M, N, addend=addend, add_ld=add_ld))
        else:
            self.gemm(_gemm_dgrad(A, lda, B, ldb, K, N, Cp, ldc, M, bias=bias, addend=addend, add_ld=add_ld))

    def wgrad(self, dy, ldy, x, ldx, S, B, g, ldg, Nout, Kin, gbias=0):
        """g[Nout,Kin] += dY^T X over the B * S token rows, gbias += column sums of dY: the specialised kernel when the output is
        a multiple of 128 x 128, else a split reduction of the grouped GEMM (one member per sequence)."""
        if Nout % 128 == 0 and Kin % 128 == 0 and ldy % 4 == 0 and ldx % 4 == 0 and dy % 16 == 0 and x % 16 == 0 \
                and os.environ.get("SEEME_GEMM128", "1") != "0":
            self.ops.append(lambda: L.check(L.lib().seeme_wgrad128(dy, ldy, x, ldx, B * S, Nout, Kin, g, ldg, gbias, L.current_stream()),
                                            "seeme_wgrad128"))
        else:
            self.gemm(_gemm_wgrad(dy, ldy, x, ldx, S, B, g, ldg, Nout, Kin, gbias))

    def add_ln(self, sub, res, norm, y, xhat, rstd, M, sub_seq_rows=0):
        a = L.VtLn()
        a.sub, a.res, a.gamma, a.beta, a.y, a.xhat, a.rstd = sub, res, P(norm.weight), P(norm.bias), y, xhat, rstd
        a.M, a.sub_seq_rows, a.eps = M, sub_seq_rows, 1e-5
        self.ops.append(lambda a=a: L.check(L.lib().seeme_vt_add_ln(C.byref(a), L.current_stream()), "seeme_vt_add_ln"))

    def ln_bwd(self, dy, dy2, xhat, rstd, norm, dpre, M):
        a = L.VtLnBwd()
        a.dy, a.dy2, a.xhat, a.rstd, a.gamma, a.dpre = dy, dy2, xhat, rstd, P(norm.weight), dpre
        a.dgamma, a.dbeta, a.M, a.accumulate = P(norm.weight.grad), P(norm.bias.grad), M, 0
        self.ops.append(lambda a=a: L.check(L.lib().seeme_vt_ln_bwd(C.byref(a), L.current_stream()), "seeme_vt_ln_bwd"))

    def dropout(self, x, mask, scale, out, n):
        self.ops.append(lambda: L.check(L.lib().seeme_vt_dropout(x, mask, scale, out, n, L.current_stream()), "seeme_vt_dropout"))

    def call(self, fn):
        self.ops.append(fn)

    def run(self):
        for f in self.ops:
            f()


class _StackPlan:
    """Buffers and recorded launches of one skip stack (encoder or decoder) for a batch shape."""

    def __init__(self, vae, dec: bool, B: int, T: int, drop: float = 0.0):
        self.vae, self.dec, self.B, self.T = vae, dec, B, T
        self.drop, self.dscale = float(drop), 1.0 / (1.0 - float(drop))
        self.S = S = T if dec else T + 2
        self.M = M = B * S
        self.F = vae.nfeats
        dev = vae.final_layer.weight.device
        self.dev = dev
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        stack = vae.decoder if dec else vae.encoder
        self.stack = stack
        self.layers = [stack.input_blocks[0], stack.input_blocks[1], stack.middle_block, stack.output_blocks[0], stack.output_blocks[1]]
        self.lins = [stack.linear_blocks[0], stack.linear_blocks[1]]
        self.lengths = torch.zeros(B, dtype=torch.int32, device=dev)
        self.x0 = z(M, 256)
        self.sv = []                              # per layer saved tensors
        for _ in range(5):
            d = dict(qkv=z(M, 768), P=z(B, S, S), O=z(M, 256), x1=z(M, 256), xh1=z(M, 256), rs1=z(M), hpre=z(M, 128), h=z(M, 128),
                     x2=z(M, 256), xh2=z(M, 256), rs2=z(M))
            if dec:
                d.update(vc=z(B, 256), cv=z(B, 256), x1b=z(M, 256), xh1b=z(M, 256), rs1b=z(M))
            self.sv.append(d)
        if self.drop > 0:      # keep-masks of every dropout site of the stack, drawn by ONE bernoulli_ per forward
            r16 = lambda n: (n + 15) // 16 * 16
            sizes = [("mP", B * S * S), ("m1", M * 256), ("mh", M * 128), ("m2", M * 256)] + ([("mw", B * S), ("mc", M * 256)] if dec else [])
            per = sum(r16(n) for _, n in sizes)
            self.masks = torch.zeros(5 * per, dtype=torch.uint8, device=dev)
            for l, d in enumerate(self.sv):
                o = l * per
                for name, n in sizes:
                    d[name] = self.masks[o:o + n]
                    o += r16(n)
                d["Pd"] = z(B, S, S)
            self.Gm = z(M, 256)
        self.xs = [z(M, 256), z(M, 256)]          # outputs of the two skip linears
        self.yn, self.xhn, self.rsn = z(M, 256), z(M, 256), z(M)
        self.tmp = z(M, 256)                      # forward temporary (attention / FFN output before the LayerNorm)
        if dec:
            self.zb, self.feats, self.dz = z(B, 256), z(M, self.F), z(B, 256)
            self.dfeats, self.dcv, self.dvc = z(M, self.F), z(B, 256), z(B, 256)
            self.G1b = z(M, 256)
        else:
            self.feat_in = z(B * T, self.F)
        # backward scratch
        self.dyn = z(M, 256)
        self.G2, self.dh, self.dhpre, self.DX1, self.G1, self.dO = z(M, 256), z(M, 128), z(M, 128), z(M, 256), z(M, 256), z(M, 256)
        self.dP, self.dqkv = z(B, S, S), z(M, 768)
        self.DXa, self.DXb, self.SK = z(M, 256), z(M, 256), [z(M, 256), z(M, 256)]
        self.fwd = self._record_forward()
        self.bwd, self._bwd_key = None, None
        self.busy = False

    # ------------------------------------------------------------------ parameters of this stack
    def params(self) -> List[torch.nn.Parameter]:
        v = self.vae
        ps = list(self.stack.parameters())
        if self.dec:
            ps += [v.query_pos_decoder.pe, v.final_layer.weight, v.final_layer.bias]
        else:
            ps += [v.query_pos_encoder.pe, v.global_motion_token, v.skel_embedding.weight, v.skel_embedding.bias]
        return ps

    # ------------------------------------------------------------------ forward
    def _layer_fwd(self, ops: _Ops, l: int, x_in: int):
        lp, sv, B, S, M = self.layers[l], self.sv[l], self.B, self.S, self.M
        sa = lp.self_attn
        ops.plain(x_in, 256, P(sa.in_proj_weight), 256, True, P(sv["qkv"]), 768, M, 768, 256, bias=P(sa.in_proj_bias))
        q = P(sv["qkv"])
        ops.gemm(_prob([q], [q + 4 * 256], [256], [1], [1], 768, 768, P(sv["P"]), S, S, S, nbatch=B, bstrides=(S * 768, S * 768, S * S)))
        n_prefix = 0 if self.dec else 2
        ops.call(lambda p=P(sv["P"]): L.check(L.lib().seeme_vt_softmax_fwd(p, P(self.lengths), B, S, n_prefix, 1.0 / 16.0, L.current_stream()),
                                              "seeme_vt_softmax_fwd"))
        dr, ds = self.drop > 0, self.dscale
        pv = P(sv["P"])
        if dr:                                                   # attention-weight dropout of nn.MultiheadAttention
            ops.dropout(P(sv["P"]), P(sv["mP"]), ds, P(sv["Pd"]), B * S * S)
            pv = P(sv["Pd"])
        ops.gemm(_prob([pv], [q + 4 * 512], [S], [1], [768], S, 1, P(sv["O"]), 256, S, 256, nbatch=B,
                       bstrides=(S * S, S * 768, S * 256)))
        ops.plain(P(sv["O"]), 256, P(sa.out_proj.weight), 256, True, P(self.tmp), 256, M, 256, 256, bias=P(sa.out_proj.bias))
        if dr:
            ops.dropout(P(self.tmp), P(sv["m1"]), ds, P(self.tmp), M * 256)           # dropout1
        ops.add_ln(P(self.tmp), x_in, lp.norm1, P(sv["x1"]), P(sv["xh1"]), P(sv["rs1"]), M)
        x1 = P(sv["x1"])
        n_ffn = lp.norm2
        if self.dec:
            ca = lp.multihead_attn
            ops.gemm(_gemm_fwd([P(self.zb)], 256, P(ca.in_proj_weight) + 4 * 512 * 256, 256, [256], P(ca.in_proj_bias) + 4 * 512,
                               P(sv["vc"]), 256, B, 256))
            if dr:       # cv = W_o v without the bias; per row (w[b,s] cv[b] + b_o) * dropout2 mask
                ops.gemm(_gemm_fwd([P(sv["vc"])], 256, P(ca.out_proj.weight), 256, [256], 0, P(sv["cv"]), 256, B, 256))
                ops.call(lambda: L.check(L.lib().seeme_vt_cross_rows(P(sv["cv"]), P(ca.out_proj.bias), P(sv["mw"]), P(sv["mc"]), ds, B, S,
                                                                      P(self.tmp), L.current_stream()), "seeme_vt_cross_rows"))
                ops.add_ln(P(self.tmp), x1, lp.norm2, P(sv["x1b"]), P(sv["xh1b"]), P(sv["rs1b"]), M)
            else:
                ops.gemm(_gemm_fwd([P(sv["vc"])], 256, P(ca.out_proj.weight), 256, [256], P(ca.out_proj.bias), P(sv["cv"]), 256, B, 256))
                ops.add_ln(P(sv["cv"]), x1, lp.norm2, P(sv["x1b"]), P(sv["xh1b"]), P(sv["rs1b"]), M, sub_seq_rows=S)
            x1 = P(sv["x1b"])
            n_ffn = lp.norm3
        ops.plain(x1, 256, P(lp.linear1.weight), 256, True, P(sv["hpre"]), 128, M, 128, 256, bias=P(lp.linear1.bias))
        ops.call(lambda a=P(sv["hpre"]), o=P(sv["h"]): L.check(L.lib().seeme_vt_gelu(a, 0, o, M * 128, L.current_stream()), "seeme_vt_gelu"))
        if dr:
            ops.dropout(P(sv["h"]), P(sv["mh"]), ds, P(sv["h"]), M * 128)               # dropout inside the FFN
        ops.plain(P(sv["h"]), 128, P(lp.linear2.weight), 128, True, P(self.tmp), 256, M, 256, 128, bias=P(lp.linear2.bias))
        if dr:
            ops.dropout(P(self.tmp), P(sv["m2"]), ds, P(self.tmp), M * 256)           # dropout2 (decoder: dropout3)
        ops.add_ln(P(self.tmp), x1, n_ffn, P(sv["x2"]), P(sv["xh2"]), P(sv["rs2"]), M)
        return P(sv["x2"])

    def _record_forward(self) -> _Ops:
        ops = _Ops(self.dev)
        v, B, S, T, M, F = self.vae, self.B, self.S, self.T, self.M, self.F
        if self.drop > 0:
            ops.call(lambda: self.masks.bernoulli_(1.0 - self.drop))
        if not self.dec:
            emb = v.skel_embedding
            ops.gemm(_prob([P(self.feat_in)], [P(emb.weight)], [F], [1], [1], F, F, P(self.x0) + 4 * 2 * 256, 256, T, 256, bias=P(emb.bias),
                           nbatch=B, bstrides=(T * F, 0, S * 256)))

            def assemble():
                x = self.x0.view(B, S, 256)
                x[:, :2] = v.global_motion_token[None]
                x += v.query_pos_encoder.pe[:S, 0][None]
            ops.call(assemble)
        else:
            ops.call(lambda: self.x0.view(B, S, 256).copy_(v.query_pos_decoder.pe[:S, 0][None].expand(B, S, 256)))
        x = P(self.x0)
        outs = []
        x = self._layer_fwd(ops, 0, x); outs.append(x)
        x = self._layer_fwd(ops, 1, x); outs.append(x)
        x = self._layer_fwd(ops, 2, x)
        for i in range(2):
            lin = self.lins[i]
            ops.gemm(_gemm_fwd([x, outs.pop()], 256, P(lin.weight), 512, [256, 256], P(lin.bias), P(self.xs[i]), 256, M, 256))
            x = self._layer_fwd(ops, 3 + i, P(self.xs[i]))
        ops.add_ln(x, 0, self.stack.norm, P(self.yn), P(self.xhn), P(self.rsn), M)
        if self.dec:
            fl = v.final_layer
            ops.gemm(_gemm_fwd([P(self.yn)], 256, P(fl.weight), 256, [256], P(fl.bias), P(self.feats), F, M, F))
        return ops

    # ------------------------------------------------------------------ backward
    def _layer_bwd(self, ops: _Ops, l: int, x_in: int, dy: int, dy2: int, out: int):
        """dy (+ dy2) = gradient w.r.t. the layer output; writes the gradient w.r.t. the layer input to `out`."""
        lp, sv, B, S, M = self.layers[l], self.sv[l], self.B, self.S, self.M
        sa = lp.self_attn
        G = lambda t: t.grad.data_ptr()
        n_ffn = lp.norm3 if self.dec else lp.norm2
        x1 = P(sv["x1b"]) if self.dec else P(sv["x1"])
        dr, ds = self.drop > 0, self.dscale
        ops.ln_bwd(dy, dy2, P(sv["xh2"]), P(sv["rs2"]), n_ffn, P(self.G2), M)
        g2 = P(self.G2)                                          # gradient of the FFN output: behind its dropout mask
        if dr:
            ops.dropout(P(self.G2), P(sv["m2"]), ds, P(self.Gm), M * 256)
            g2 = P(self.Gm)
        ops.plain(g2, 256, P(lp.linear2.weight), 128, False, P(self.dh), 128, M, 128, 256)
        ops.wgrad(g2, 256, P(sv["h"]), 128, S, B, G(lp.linear2.weight), 128, 256, 128, G(lp.linear2.bias))
        if dr:
            ops.dropout(P(self.dh), P(sv["mh"]), ds, P(self.dh), M * 128)
        ops.call(lambda: L.check(L.lib().seeme_vt_gelu(P(sv["hpre"]), P(self.dh), P(self.dhpre), M * 128, L.current_stream()), "seeme_vt_gelu"))
        ops.plain(P(self.dhpre), 128, P(lp.linear1.weight), 256, False, P(self.DX1), 256, M, 256, 128, addend=P(self.G2), add_ld=256)
        ops.wgrad(P(self.dhpre), 128, x1, 256, S, B, G(lp.linear1.weight), 256, 128, 256, G(lp.linear1.bias))
        d_x1 = P(self.DX1)
        if self.dec:
            ca = lp.multihead_attn
            ops.ln_bwd(P(self.DX1), 0, P(sv["xh1b"]), P(sv["rs1b"]), lp.norm2, P(self.G1b), M)
            if dr:
                ops.dropout(P(self.G1b), P(sv["mc"]), ds, P(self.Gm), M * 256)

                def bias_grad(ca=ca):
                    ca.out_proj.bias.grad += self.Gm.sum(0)
                ops.call(bias_grad)
                ops.call(lambda: L.check(L.lib().seeme_vt_seq_sum(P(self.Gm), P(self.dcv), B, S, 0, P(sv["mw"]), ds, L.current_stream()),
                                         "seeme_vt_seq_sum"))
            else:
                ops.call(lambda: L.check(L.lib().seeme_vt_seq_sum(P(self.G1b), P(self.dcv), B, S, 0, 0, 1.0, L.current_stream()), "seeme_vt_seq_sum"))
            ops.gemm(_gemm_dgrad(P(self.dcv), 256, P(ca.out_proj.weight), 256, 256, 256, P(self.dvc), 256, B))
            ops.gemm(_gemm_dgrad(P(self.dvc), 256, P(ca.in_proj_weight) + 4 * 512 * 256, 256, 256, 256, P(self.dz), 256, B, accumulate=1),
                     _gemm_wgrad(P(self.dcv), 256, P(sv["vc"]), 256, B, 1, G(ca.out_proj.weight), 256, 256, 256, 0 if dr else G(ca.out_proj.bias)),
                     _gemm_wgrad(P(self.dvc), 256, P(self.zb), 256, B, 1, G(ca.in_proj_weight) + 4 * 512 * 256, 256, 256, 256,
                                 G(ca.in_proj_bias) + 4 * 512))
            d_x1 = P(self.G1b)
        ops.ln_bwd(d_x1, 0, P(sv["xh1"]), P(sv["rs1"]), lp.norm1, P(self.G1), M)
        g1 = P(self.G1)                                          # gradient of the attention output: behind dropout1
        if dr:
            ops.dropout(P(self.G1), P(sv["m1"]), ds, P(self.Gm), M * 256)
            g1 = P(self.Gm)
        ops.plain(g1, 256, P(sa.out_proj.weight), 256, False, P(self.dO), 256, M, 256, 256)
        ops.wgrad(g1, 256, P(sv["O"]), 256, S, B, G(sa.out_proj.weight), 256, 256, 256, G(sa.out_proj.bias))
        q, dq = P(sv["qkv"]), P(self.dqkv)
        pv = P(sv["Pd"]) if dr else P(sv["P"])
        ops.gemm(_prob([P(self.dO)], [q + 4 * 512], [256], [1], [1], 256, 768, P(self.dP), S, S, S, nbatch=B, bstrides=(S * 256, S * 768, S * S)),
                 _prob([pv], [P(self.dO)], [S], [S], [256], 1, 1, dq + 4 * 512, 768, S, 256, nbatch=B, bstrides=(S * S, S * 256, S * 768)))
        if dr:
            ops.dropout(P(self.dP), P(sv["mP"]), ds, P(self.dP), B * S * S)
        ops.call(lambda: L.check(L.lib().seeme_vt_softmax_bwd(P(self.dP), P(sv["P"]), B * S, S, 1.0 / 16.0, L.current_stream()),
                                 "seeme_vt_softmax_bwd"))
        ops.gemm(_prob([P(self.dP)], [q + 4 * 256], [S], [1], [768], S, 1, dq, 768, S, 256, nbatch=B, bstrides=(S * S, S * 768, S * 768)),
                 _prob([P(self.dP)], [q], [S], [S], [768], 1, 1, dq + 4 * 256, 768, S, 256, nbatch=B, bstrides=(S * S, S * 768, S * 768)))
        ops.plain(dq, 768, P(sa.in_proj_weight), 256, False, out, 256, M, 256, 768, addend=P(self.G1), add_ld=256)
        ops.wgrad(dq, 768, x_in, 256, S, B, G(sa.in_proj_weight), 256, 768, 256, G(sa.in_proj_bias))

    def record_backward(self):
        params = [p for p in self.params() if p.requires_grad]
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        key = tuple(p.grad.data_ptr() for p in params)
        if key == self._bwd_key:
            return
        self._bwd_key = key
        ops = _Ops(self.dev)
        v, B, S, T, M, F = self.vae, self.B, self.S, self.T, self.M, self.F
        G = lambda t: t.grad.data_ptr()
        if self.dec:
            fl = v.final_layer
            ops.call(lambda: self.dz.zero_())
            ops.gemm(_gemm_dgrad(P(self.dfeats), F, P(fl.weight), 256, F, 256, P(self.dyn), 256, M),
                     _gemm_wgrad(P(self.dfeats), F, P(self.yn), 256, S, B, G(fl.weight), 256, F, 256, G(fl.bias)))
        A, Bb = P(self.DXa), P(self.DXb)
        x2 = [P(sv["x2"]) for sv in self.sv]
        ops.ln_bwd(P(self.dyn), 0, P(self.xhn), P(self.rsn), self.stack.norm, A, M)
        self._layer_bwd(ops, 4, P(self.xs[1]), A, 0, Bb)                       # Bb = d xs[1]
        lin = self.lins[1]                                                      # xs[1] = W [x2[3] | x2[0]] + b
        ops.gemm(_gemm_dgrad(Bb, 256, P(lin.weight), 512, 256, 256, A, 256, M),
                 _gemm_dgrad(Bb, 256, P(lin.weight) + 4 * 256, 512, 256, 256, P(self.SK[0]), 256, M),
                 )
        ops.wgrad(Bb, 256, x2[3], 256, S, B, G(lin.weight), 512, 256, 256, G(lin.bias))
        ops.wgrad(Bb, 256, x2[0], 256, S, B, G(lin.weight) + 4 * 256, 512, 256, 256)
        self._layer_bwd(ops, 3, P(self.xs[0]), A, 0, Bb)                       # Bb = d xs[0]
        lin = self.lins[0]                                                      # xs[0] = W [x2[2] | x2[1]] + b
        ops.gemm(_gemm_dgrad(Bb, 256, P(lin.weight), 512, 256, 256, A, 256, M),
                 _gemm_dgrad(Bb, 256, P(lin.weight) + 4 * 256, 512, 256, 256, P(self.SK[1]), 256, M),
                 )
        ops.wgrad(Bb, 256, x2[2], 256, S, B, G(lin.weight), 512, 256, 256, G(lin.bias))
        ops.wgrad(Bb, 256, x2[1], 256, S, B, G(lin.weight) + 4 * 256, 512, 256, 256)
        self._layer_bwd(ops, 2, x2[1], A, 0, Bb)                                # Bb = d x2[1] (+ SK[1])
        self._layer_bwd(ops, 1, x2[0], Bb, P(self.SK[1]), A)                    # A = d x2[0] (+ SK[0])
        self._layer_bwd(ops, 0, P(self.x0), A, P(self.SK[0]), Bb)               # Bb = d x0
        if self.dec:
            def pe_grad():
                v.query_pos_decoder.pe.grad[:S, 0] += self.DXb.view(B, S, 256).sum(0)
            ops.call(pe_grad)
        else:
            emb = v.skel_embedding
            ops.gemm(_prob([Bb + 4 * 2 * 256], [P(self.feat_in)], [T], [256], [F], 1, 1, G(emb.weight), F, 256, F, accumulate=2,
                           colsum=G(emb.bias), nbatch=B, bstrides=(S * 256, T * F, 0)))

            def in_grads():
                d = self.DXb.view(B, S, 256)
                v.query_pos_encoder.pe.grad[:S, 0] += d.sum(0)
                v.global_motion_token.grad += d[:, :2].sum(0)
            ops.call(in_grads)
        self.bwd = ops


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tr, hook, features, lengths_t):
        plan = tr.plan(False, features.shape[0], features.shape[1])
        B, T = plan.B, plan.T
        plan.feat_in.copy_(features.reshape(B * T, plan.F))
        plan.lengths.copy_(lengths_t)
        plan.fwd.run()
        plan.busy = True
        ctx.plan = plan
        return plan.yn.view(B, plan.S, 256)[:, :2].permute(1, 0, 2).contiguous()      # [2,B,256]: mu | logvar

    @staticmethod
    def backward(ctx, ddist):
        plan = ctx.plan
        plan.record_backward()
        plan.dyn.zero_()
        plan.dyn.view(plan.B, plan.S, 256)[:, :2] = ddist.permute(1, 0, 2)
        plan.bwd.run()
        plan.busy = False
        return None, None, None, None


class _Decode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tr, hook, z, lengths_t):
        B = z.shape[1]
        plan = tr.plan(True, B, tr._T)
        plan.zb.copy_(z.reshape(B, 256))
        plan.lengths.copy_(lengths_t)
        plan.fwd.run()
        plan.busy = True
        ctx.plan = plan
        return plan.feats.view(B, plan.T, plan.F).clone()

    @staticmethod
    def backward(ctx, dfeats):
        plan = ctx.plan
        plan.record_backward()
        plan.dfeats.copy_(dfeats.reshape(plan.M, plan.F))
        plan.bwd.run()
        plan.busy = False
        return None, None, plan.dz.view(1, plan.B, 256).clone(), None


class VaeTrainer:
    """Plans of one MldVae; ``encode(features, lengths) -> (mu, logvar)`` and ``decode(z, lengths) -> feats``, differentiable."""

    def __init__(self, vae):
        self.vae = vae
        self.plans = {}
        self._len_cache = {}
        self._T = 0
        self._key = tuple(p.data_ptr() for p in vae.parameters())

    @staticmethod
    def supported(vae, T: int) -> bool:
        ok = vae.encoder.input_blocks[0].self_attn.num_heads == 1 and T + 2 <= 512 and vae.final_layer.weight.is_cuda
        return ok and all(p.requires_grad and p.dtype == torch.float32 for p in vae.parameters())

    def stale(self) -> bool:
        return tuple(p.data_ptr() for p in self.vae.parameters()) != self._key

    def plan(self, dec: bool, B: int, T: int) -> _StackPlan:
        # training mode: the reference's dropout sites (nn.MultiheadAttention weights, dropout1/2/3, the FFN's inner dropout;
        # cross_attention.py:264-273,324-337) with p of the module; eval mode: none
        drop = float(self.vae.encoder.input_blocks[0].self_attn.dropout) if self.vae.training else 0.0
        k = (dec, B, T, drop)
        p = self.plans.get(k)
        if p is None or p.busy:           # busy: a second forward before the first one's backward keeps that one's saved tensors
            if len(self.plans) >= 8:      # a plan holds every saved activation of its shape (B=64, T=196: ~1.5 GB): keep the recent ones
                for old in [q for q, v in self.plans.items() if not v.busy][: len(self.plans) - 7]:
                    del self.plans[old]
            p = _StackPlan(self.vae, dec, B, T, drop)
            self.plans.pop(k, None)
            self.plans[k] = p
        return p

    def _lengths(self, lengths, dev):
        # (cached per value: the list -> device tensor conversion is a synchronous host-to-device copy, twice per step otherwise)
        key = (tuple(int(v) for v in lengths), str(dev))
        t = self._len_cache.get(key)
        if t is None:
            if len(self._len_cache) > 64:
                self._len_cache.clear()
            t = torch.as_tensor(key[0], dtype=torch.int32, device=dev)
            self._len_cache[key] = t
        return t

    def encode(self, features: torch.Tensor, lengths: List[int]):
        """features [B,T,F] -> (mu [1,B,256], logvar [1,B,256])."""
        L.require_cuda(features, "features")
        dist = _Encode.apply(self, self.vae.final_layer.weight, features.contiguous().float(), self._lengths(lengths, features.device))
        return dist[0:1], dist[1:2]

    def decode(self, z: torch.Tensor, lengths: List[int]) -> torch.Tensor:
        """z [1,B,256] -> feats [B,T,F] (padded frames not zeroed, as the twin)."""
        self._T = int(max(lengths))
        return _Decode.apply(self, self.vae.final_layer.weight, z.contiguous().float(), self._lengths(lengths, z.device))
