"""ctypes binding of libseeme_hip.so (the C-ABI declared in include/seeme_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails the caller gets an
exception.  The structures below mirror the header field by field.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEEME_HIP_LIB: another build of the same library (A/B timing of kernel variants, debug builds with cycle stamps)
LIB_PATH = os.environ.get("SEEME_HIP_LIB") or os.path.join(_HERE, "libseeme_hip.so")

NLAYERS = 5
TROW = 7680
CROW = 5120
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3
SCHED_NONE, SCHED_DDIM, SCHED_DDPM = 0, 1, 2

fp = C.c_void_p  # device pointers travel as integers


class SeemeError(RuntimeError):
    pass


class LinearArgs(C.Structure):
    _fields_ = [("A", fp), ("lda", C.c_int), ("A2", fp), ("lda2", C.c_int), ("K1", C.c_int),
                ("W", fp), ("ldw", C.c_int), ("bias", fp), ("res", fp), ("ldr", C.c_int),
                ("ln_w", fp), ("ln_b", fp), ("pre_ln_w", fp), ("pre_ln_b", fp),
                ("Y", fp), ("ldy", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("pre_act", C.c_int), ("act", C.c_int), ("eps", C.c_float)]


class XfLayer(C.Structure):
    _fields_ = [(n, fp) for n in ("in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b",
                                  "n1_w", "n1_b", "n2_w", "n2_b",
                                  "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b", "n3_w", "n3_b")]


class SkipStack(C.Structure):
    _fields_ = [("layer", XfLayer * NLAYERS), ("skip_w", fp * 2), ("skip_b", fp * 2),
                ("norm_w", fp), ("norm_b", fp)]


class XfLayerH(C.Structure):
    _fields_ = [(n, fp) for n in ("in_w", "out_w", "l1_w", "l2_w")]


class VaeWeightsH(C.Structure):
    _fields_ = [("enc", XfLayerH * NLAYERS), ("dec", XfLayerH * NLAYERS), ("enc_skip", fp * 2), ("dec_skip", fp * 2),
                ("emb_w", fp), ("fin_w", fp), ("ca_fold_w", fp)]


def pack_mfma16(W, dtype):
    """[N,K] matrix -> 16-bit copy in MFMA fragment order [N/16][K/32][kq=4][r=16][8] (N padded to 16, K to 32):
    every wave-load of a B fragment is then 1 KiB contiguous."""
    import torch
    N, K = W.shape
    Np, Kp = (N + 15) // 16 * 16, (K + 31) // 32 * 32
    Wz = torch.zeros(Np, Kp, device=W.device, dtype=dtype)
    Wz[:N, :K] = W.detach().to(dtype)
    return Wz.view(Np // 16, 16, Kp // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


class VaeWeights(C.Structure):
    _fields_ = [("nfeats", C.c_int), ("ff", C.c_int), ("token", fp), ("pe_enc", fp), ("pe_dec", fp),
                ("emb_w", fp), ("emb_ldw", C.c_int), ("emb_b", fp), ("fin_w", fp), ("fin_b", fp),
                ("enc", SkipStack), ("dec", SkipStack), ("ca_fold_w", fp), ("ca_fold_b", fp), ("h16", fp)]


class DenoiserWeights(C.Structure):
    _fields_ = [("wg", fp), ("wdtype", C.c_int), ("vp", fp), ("layout", fp), ("nhead", C.c_int), ("ff_sa", C.c_int),
                ("ff", C.c_int), ("kv_cat_w", fp), ("kv_cat_b", fp), ("style_cat_w", fp), ("style_cat_b", fp),
                ("time_w1", fp), ("time_b1", fp), ("time_w2", fp), ("time_b2", fp),
                ("ca_kv_w", fp * NLAYERS), ("ca_kv_b", fp * NLAYERS),
                ("ca_tn_w", fp * NLAYERS), ("ca_tn_b", fp * NLAYERS),
                ("ca_fold_w", fp), ("ca_fold_b", fp), ("ln_ones", fp), ("ln_zeros", fp), ("sa_fold", C.c_int),
                ("ca_pn_w", fp), ("ca_pn_b", fp), ("ca_po_w", fp), ("ca_po_b", fp)]


class SampleArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("steps", C.c_int), ("sched", C.c_int), ("cfg", C.c_int),
                ("guidance_scale", C.c_float), ("latents", fp), ("ctab", fp), ("ttab", fp), ("trow", fp),
                ("trow_per_sample", C.c_int), ("coef", fp), ("noise", fp), ("out", fp), ("catab", fp), ("save", fp), ("force_query", C.c_int), ("drop", fp), ("drop_scale", C.c_float), ("xcds", C.c_int)]


class DenCluster(C.Structure):
    _fields_ = [("wgc", fp), ("wdtype", C.c_int), ("vpc", fp), ("C", C.c_int), ("placement", C.c_int), ("flags", C.c_int),
                ("xchg", fp), ("xchg_bytes", C.c_size_t), ("query", C.c_int), ("samples", C.c_int)]


class SmplModel(C.Structure):
    _fields_ = [("V", C.c_int), ("v_template", fp), ("blend_w", fp), ("lbs_weights", fp), ("J_template", fp),
                ("J_shapedirs", fp), ("parents", fp), ("ex_template", fp), ("ex_shapedirs", fp),
                ("ex_posedirs", fp), ("ex_weights", fp)]


class PointnetWeights(C.Structure):
    _fields_ = [("out_dim", C.c_int), ("pos_w", fp), ("pos_b", fp), ("fc0_w", fp * 4), ("fc0_b", fp * 4),
                ("fc1_w", fp * 4), ("fc1_b", fp * 4), ("sc_w", fp * 4), ("fcc_w", fp), ("fcc_b", fp)]


class PointnetBf16(C.Structure):
    _fields_ = [("fc0", fp * 4), ("fc1", fp * 4), ("sc", fp * 4), ("posf", fp), ("sc3", fp), ("stream", fp * 4), ("sc3f", fp)]


class GlueRows(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("dist", fp), ("dist_rows", C.c_int), ("eps_z", fp), ("eps_c", fp),
                ("slot_c", C.c_int), ("cond", fp), ("noise", fp), ("timesteps", fp), ("acp", fp), ("freq", fp),
                ("flip_sin_to_cos", C.c_int), ("latents", fp), ("noisy", fp), ("tfeat", fp)]


class GemmProblem(C.Structure):
    _fields_ = [("a", fp * 10), ("b", fp * 10), ("seg_len", C.c_int * 10), ("a_ks", C.c_long * 10),
                ("b_ks", C.c_long * 10), ("nseg", C.c_int), ("a_rs", C.c_long), ("b_cs", C.c_long), ("c", fp), ("ldc", C.c_long),
                ("M", C.c_int), ("N", C.c_int), ("a_pro", C.c_int), ("a_p0", fp), ("a_p1", fp), ("b_pro", C.c_int),
                ("b_p0", fp), ("b_p1", fp), ("bias", fp), ("epi", C.c_int), ("e0", fp), ("e_ld", C.c_long),
                ("accumulate", C.c_int), ("colsum", fp), ("tile0", C.c_int), ("tiles_n", C.c_int), ("nbatch", C.c_int),
                ("a_bstride", C.c_long), ("b_bstride", C.c_long), ("c_bstride", C.c_long), ("alpha", C.c_float), ("addend", fp),
                ("add_ld", C.c_long)]


class VtLn(C.Structure):
    _fields_ = [("sub", fp), ("res", fp), ("gamma", fp), ("beta", fp), ("y", fp), ("xhat", fp), ("rstd", fp), ("M", C.c_long),
                ("sub_seq_rows", C.c_int), ("eps", C.c_float)]


class VtLnBwd(C.Structure):
    _fields_ = [("dy", fp), ("xhat", fp), ("rstd", fp), ("gamma", fp), ("dpre", fp), ("dgamma", fp), ("dbeta", fp), ("M", C.c_long),
                ("accumulate", C.c_int), ("dy2", fp)]


class GlueMid(C.Structure):
    _fields_ = [("M", C.c_int), ("B", C.c_int), ("dxl", fp), ("dcs", fp), ("xhat", fp), ("rstd", fp), ("tn_w", fp * 5),
                ("g_tn_w", fp * 5), ("g_tn_b", fp * 5), ("dcond", fp), ("dea", fp), ("deb", fp), ("emb", fp), ("demb", fp)]


GEO_AA_TO_QUAT, GEO_AA_TO_ROTMAT, GEO_QUAT_TO_ROTMAT, GEO_ROT6D_PROHMR, GEO_ROT6D_DIFFUSION = range(5)

# name -> (restype, argtypes); every symbol of include/seeme_hip.h
_SIGNATURES = {
    "seeme_smpl_joints_backward": (C.c_int, [C.POINTER(SmplModel), fp, fp, fp, C.c_int, fp, fp, C.c_int, fp]),
    "seeme_vt_add_ln": (C.c_int, [C.POINTER(VtLn), fp]),
    "seeme_vt_ln_bwd": (C.c_int, [C.POINTER(VtLnBwd), fp]),
    "seeme_vt_softmax_fwd": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, fp]),
    "seeme_vt_softmax_bwd": (C.c_int, [fp, fp, C.c_long, C.c_int, C.c_float, fp]),
    "seeme_vt_gelu": (C.c_int, [fp, fp, fp, C.c_long, fp]),
    "seeme_vt_seq_sum": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, fp, C.c_float, fp]),
    "seeme_vt_dropout": (C.c_int, [fp, fp, C.c_float, fp, C.c_long, fp]),
    "seeme_vt_cross_rows": (C.c_int, [fp, fp, fp, fp, C.c_float, C.c_int, C.c_int, fp, fp]),
    "seeme_den_vecgrad": (C.c_int, [fp, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, fp, fp]),
    "seeme_glue_rows": (C.c_int, [C.POINTER(GlueRows), fp]),
    "seeme_glue_ln": (C.c_int, [fp, fp, fp, C.c_int, fp]),
    "seeme_glue_mid": (C.c_int, [C.POINTER(GlueMid), fp]),
    "seeme_grouped_gemm": (C.c_int, [fp, C.c_int, C.c_int, fp]),
    "seeme_gemm_problem_bytes": (C.c_int, []),
    "seeme_wgrad128": (C.c_int, [fp, C.c_long, fp, C.c_long, C.c_int, C.c_int, C.c_int, fp, C.c_long, fp, fp]),
    "seeme_gemm128": (C.c_int, [fp, C.c_long, fp, C.c_long, C.c_int, fp, C.c_long, C.c_int, C.c_int, C.c_int, fp, fp, C.c_long, fp]),
    "seeme_version": (C.c_int, []),
    "seeme_last_error": (C.c_char_p, []),
    "seeme_linear": (C.c_int, [C.POINTER(LinearArgs), fp]),
    "seeme_vae_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "seeme_vae_encode": (C.c_int, [C.POINTER(VaeWeights), fp, fp, C.c_int, C.c_int, fp, fp, fp, C.c_size_t, fp]),
    "seeme_vae_decode": (C.c_int, [C.POINTER(VaeWeights), fp, fp, C.c_int, C.c_int, fp, fp, C.c_size_t, fp]),
    "seeme_denoiser_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "seeme_denoiser_time_tables": (C.c_int, [C.POINTER(DenoiserWeights), fp, C.c_int, fp, fp, C.c_size_t, fp]),
    "seeme_denoiser_cond_tables": (C.c_int, [C.POINTER(DenoiserWeights), fp, C.c_int, C.c_int, fp, fp, C.c_size_t, fp]),
    "seeme_denoiser_ca_tables": (C.c_int, [C.POINTER(DenoiserWeights), fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_size_t, fp]),
    "seeme_denoiser_sample": (C.c_int, [C.POINTER(DenoiserWeights), C.POINTER(SampleArgs), fp]),
    "seeme_denoiser_sample_cluster": (C.c_int, [C.POINTER(DenoiserWeights), C.POINTER(DenCluster), C.POINTER(SampleArgs), fp]),
    "seeme_den_cluster_xchg_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "seeme_den_cluster_ms_xchg_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "seeme_den_cluster_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "seeme_den_train_layout": (C.c_int, [C.POINTER(C.c_int64), C.c_int]),
    "seeme_den_train_pack": (C.c_int, [fp, fp, fp, fp, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_int, fp, fp]),
    "seeme_denoiser_backward": (C.c_int, [C.POINTER(DenoiserWeights), fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp]),
    "seeme_denoiser_backward_drop": (C.c_int, [C.POINTER(DenoiserWeights), fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_float, C.c_int, fp]),
    "seeme_den_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "seeme_den_wgrad": (C.c_int, [fp, C.c_int, C.c_int, fp, C.c_int, fp, fp]),
    "seeme_adamw_step": (C.c_int, [fp, C.c_int, fp, fp, fp, fp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, fp]),
    "seeme_adamw_step_dev": (C.c_int, [fp, C.c_int, fp, fp, fp, fp, fp, C.c_double, C.c_double, C.c_double, C.c_double, fp]),
    "seeme_geometry": (C.c_int, [C.c_int, fp, fp, C.c_int, fp]),
    "seeme_renorm": (C.c_int, [fp, fp, fp, fp, C.c_long, C.c_int, fp]),
    "seeme_pointnet_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "seeme_pointnet_encode": (C.c_int, [C.POINTER(PointnetWeights), fp, C.c_int, C.c_int, fp, fp, C.c_size_t, fp]),
    "seeme_pointnet_bf16_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "seeme_pointnet_encode_bf16": (C.c_int, [C.POINTER(PointnetWeights), C.POINTER(PointnetBf16), fp, C.c_int, C.c_int, fp, fp,
                                             C.c_size_t, fp]),
    "seeme_smpl_workspace_bytes": (C.c_size_t, [C.c_int]),
    "seeme_smpl_lbs": (C.c_int, [C.POINTER(SmplModel), fp, fp, C.c_int, fp, C.c_int, fp, fp, fp, C.c_size_t, fp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raises SeemeError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeemeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or seeme_amd/csrc/build.sh.  There is no CPU / PyTorch fallback for this path.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise SeemeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().seeme_last_error().decode("utf-8", "replace")
        raise SeemeError(f"{what or 'seeme_hip'} failed (rc={rc}): {msg}")


def default_xcds(chains: int, want="auto") -> int:
    """How many XCDs the working workgroups of a one-CU-per-chain launch sit on (SeemeSampleArgs.xcds): `want` = "auto" packs ~16
    chains per XCD -- right for a launch that has the chip to itself (one stream); callers that keep several launches in flight
    pass 8 (dealt over all XCDs).  SEEME_DEN_XCDS overrides (timing experiments, tests/test_gpu_flows.py)."""
    e = os.environ.get("SEEME_DEN_XCDS")
    if e is not None:
        want = int(e)
    if want == "auto":
        want = (chains + 15) // 16
    return min(8, max(1, int(want)))


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t, name: str):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SeemeError(f"{name} must be a tensor on the ROCm device (cuda:N); this path has no CPU fallback")
    if t.dtype != torch.float32:
        raise SeemeError(f"{name} must be float32, got {t.dtype}")
