"""Rotation-representation helpers with the signatures of ``mld/utils/geometry2.py`` (:33-117) and the
dataset ``renorm`` (mld/data/EgoBody.py:151-157), running in libseeme_hip.so."""
from __future__ import annotations

import torch

from . import _lib as L


def _run(op: int, x: torch.Tensor, in_w: int, out_shape):
    L.require_cuda(x, "input")
    x2 = x.reshape(-1, in_w).contiguous()
    M = x2.shape[0]
    out = torch.empty((M,) + out_shape, device=x.device, dtype=torch.float32)
    L.check(L.lib().seeme_geometry(op, x2.data_ptr(), out.data_ptr(), M, L.current_stream()), "seeme_geometry")
    return out


def aa_to_quat(theta: torch.Tensor) -> torch.Tensor:
    """[M,3] axis-angle -> [M,4] quaternion (w,x,y,z)."""
    return _run(L.GEO_AA_TO_QUAT, theta, 3, (4,))


def aa_to_rotmat(theta: torch.Tensor) -> torch.Tensor:
    return _run(L.GEO_AA_TO_ROTMAT, theta, 3, (3, 3))


def quat_to_rotmat(quat: torch.Tensor) -> torch.Tensor:
    return _run(L.GEO_QUAT_TO_ROTMAT, quat, 4, (3, 3))


def rot6d_to_rotmat(x: torch.Tensor, rot6d_mode: str = "prohmr") -> torch.Tensor:
    if rot6d_mode not in ("prohmr", "diffusion"):
        raise ValueError(rot6d_mode)
    return _run(L.GEO_ROT6D_PROHMR if rot6d_mode == "prohmr" else L.GEO_ROT6D_DIFFUSION, x, 6, (3, 3))


def _renorm_launch(f2: torch.Tensor, m: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(f2)
    L.check(L.lib().seeme_renorm(f2.data_ptr(), m.data_ptr(), s.data_ptr(), out.data_ptr(), f2.shape[0], f2.shape[1],
                                  L.current_stream()), "seeme_renorm")
    return out


class _Renorm(torch.autograd.Function):
    """y = x * std + mean through k_renorm; dL/dx = dL/dy * std is the same kernel with a zero mean (stage-1 training
    differentiates through the renormed reconstruction, mld.py:757-778)."""

    @staticmethod
    def forward(ctx, f2, m, s):
        ctx.save_for_backward(s)
        return _renorm_launch(f2, m, s)

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return _renorm_launch(g.contiguous().float(), torch.zeros_like(s), s), None, None


def renorm(features: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """features [..., F] * std[..., :F] + mean[..., :F]."""
    L.require_cuda(features, "features")
    F = features.shape[-1]
    f2 = features.reshape(-1, F).contiguous()
    m = mean.reshape(-1)[:F].to(features.device, torch.float32).contiguous()
    s = std.reshape(-1)[:F].to(features.device, torch.float32).contiguous()
    out = _Renorm.apply(f2, m, s) if (f2.requires_grad and torch.is_grad_enabled()) else _renorm_launch(f2, m, s)
    return out.reshape(features.shape)
