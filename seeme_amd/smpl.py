"""SMPL body model layer with the ``smplx.SMPL`` call surface the reference uses
(mld/models/modeltype/mld.py:151-153: ``smplx.SMPL(model_path=..., gender="neutral", batch_size=...)``
then ``smpl_model(betas=, body_pose=, global_orient=, transl=, pose2rot=...)`` -> ``.joints`` [M,45,3],
``.vertices`` [M,6890,3], e.g. :764-773).

``smplx==0.1.28`` and SMPL_NEUTRAL.pkl are absent offline: the LBS algorithm is restated
(App. C) and runs in libseeme_hip.so (seeme_smpl_lbs).  Buffer / parameter names follow smplx so that the
``smpl_model.*`` entries of a reference checkpoint load.  ``SMPL.synthetic()`` builds a seeded SMPL-shaped
model (true kinematic tree and extra-joint vertex ids, random geometry) for tests and benchmarks.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
# smplx vertex_ids['smpl'], VertexJointSelector order: nose, reye, leye, rear, lear, LBigToe, LSmallToe, LHeel,
# RBigToe, RSmallToe, RHeel, l{thumb,index,middle,ring,pinky}, r{thumb,index,middle,ring,pinky}
SMPL_EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                         2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]


def synthetic_model_arrays(seed: int = 1234, V: int = 6890):
    """Same recipe as oracle.mld_oracle.make_synthetic_smpl (kept separate: the product never imports oracle/)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vt = (rng.standard_normal((V, 3)) * [0.25, 0.6, 0.15]).astype(np.float32)
    sd = (rng.standard_normal((V, 3, 10)) * 0.01).astype(np.float32)
    pd = (rng.standard_normal((207, V * 3)) * 0.002).astype(np.float32)
    Jr = rng.random((24, V)) ** 8
    Jr = (Jr / Jr.sum(axis=1, keepdims=True)).astype(np.float32)
    W = rng.random((V, 24)) ** 12
    W = (W / W.sum(axis=1, keepdims=True)).astype(np.float32)
    return dict(v_template=vt, shapedirs=sd, posedirs=pd, J_regressor=Jr, lbs_weights=W,
                parents=np.asarray(SMPL_PARENTS, np.int64), faces=np.zeros((13776, 3), np.int64))


class _ChumpyStub:
    """Placeholder for ``chumpy.ch.Ch`` objects of the original SMPL pickles: keeps the state dict, exposes the array."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    def array(self):
        return np.asarray(self.__dict__.get("x", self.__dict__.get("r")))


class _RestrictedUnpickler(pickle.Unpickler):
    """An SMPL model file is arrays, a sparse matrix and strings.  Only the reconstructors of those are allowed
    (numpy arrays / dtypes / scalars, scipy.sparse matrices, chumpy arrays as inert stubs); any other global in the file
    -- i.e. anything that could run code on load -- is refused."""

    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("collections", "OrderedDict"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
        ("_codecs", "encode"),                    # how protocol-2 pickles carry the bytes of an array (pure string -> bytes)
        ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor"), ("__builtin__", "object"), ("builtins", "object"),
        ("scipy.sparse.csc", "csc_matrix"), ("scipy.sparse._csc", "csc_matrix"),
        ("scipy.sparse.csr", "csr_matrix"), ("scipy.sparse._csr", "csr_matrix"),
        ("scipy.sparse.coo", "coo_matrix"), ("scipy.sparse._coo", "coo_matrix"),
    }

    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChumpyStub
        if (module, name) in self._ALLOWED:
            import importlib
            return getattr(importlib.import_module(module), name)
        raise pickle.UnpicklingError(f"SMPL model file refers to {module}.{name}: only numpy / scipy.sparse / chumpy "
                                     "array reconstructors are accepted (convert the file to .npz otherwise)")


def _load_model_file(path: str):
    if os.path.isdir(path):
        for cand in ("SMPL_NEUTRAL.npz", "SMPL_NEUTRAL.pkl"):
            if os.path.exists(os.path.join(path, cand)):
                path = os.path.join(path, cand)
                break
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            d = {k: z[k] for k in z.files}
    else:  # user-supplied SMPL pickle (licence-gated download, prepare/download_smpl_model.sh)
        with open(path, "rb") as f:
            d = _RestrictedUnpickler(f, encoding="latin1").load()
        d = {k: (v.array() if isinstance(v, _ChumpyStub) else v) for k, v in d.items()}
    out = {}
    out["v_template"] = np.asarray(d["v_template"], np.float32)
    out["shapedirs"] = np.asarray(d["shapedirs"], np.float32)[:, :, :10]
    pd = np.asarray(d["posedirs"], np.float32)
    out["posedirs"] = pd.reshape(-1, pd.shape[-1]).T if pd.ndim == 3 else pd      # smplx: [207, V*3]
    jr = d["J_regressor"]
    out["J_regressor"] = np.asarray(jr.todense() if hasattr(jr, "todense") else jr, np.float32)
    out["lbs_weights"] = np.asarray(d["weights"] if "weights" in d else d["lbs_weights"], np.float32)
    kt = np.asarray(d["kintree_table"])[0].astype(np.int64) if "kintree_table" in d else np.asarray(SMPL_PARENTS)
    kt[0] = -1
    out["parents"] = kt
    out["faces"] = np.asarray(d.get("f", np.zeros((13776, 3))), np.int64)
    return out


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path: Optional[str] = None, gender: str = "neutral", batch_size: int = 1,
                 create_betas=True, create_global_orient=True, create_body_pose=True, create_transl=True,
                 model_arrays: Optional[dict] = None, dtype=torch.float32, **kwargs):
        super().__init__()
        if model_arrays is None:
            if model_path is None or not os.path.exists(model_path):
                raise FileNotFoundError(f"SMPL model file not found: {model_path!r} (use SMPL.synthetic() for tests)")
            model_arrays = _load_model_file(model_path)
        if dtype != torch.float32:
            raise NotImplementedError("accelerated SMPL layer is fp32")
        self.batch_size = batch_size
        m = model_arrays
        self.register_buffer("faces_tensor", torch.as_tensor(m["faces"], dtype=torch.long))
        self.register_buffer("v_template", torch.as_tensor(m["v_template"], dtype=dtype))
        self.register_buffer("shapedirs", torch.as_tensor(m["shapedirs"], dtype=dtype))
        self.register_buffer("J_regressor", torch.as_tensor(m["J_regressor"], dtype=dtype))
        self.register_buffer("posedirs", torch.as_tensor(m["posedirs"], dtype=dtype))
        self.register_buffer("parents", torch.as_tensor(m["parents"], dtype=torch.long))
        self.register_buffer("lbs_weights", torch.as_tensor(m["lbs_weights"], dtype=dtype))
        V = self.v_template.shape[0]
        ids = torch.as_tensor([i % V for i in SMPL_EXTRA_VERTEX_IDS], dtype=torch.long)
        self.vertex_joint_selector = nn.Module()
        self.vertex_joint_selector.register_buffer("extra_joints_idxs", ids)
        if create_betas:
            self.betas = nn.Parameter(torch.zeros(batch_size, 10, dtype=dtype))
        if create_global_orient:
            self.global_orient = nn.Parameter(torch.zeros(batch_size, 3, dtype=dtype))
        if create_body_pose:
            self.body_pose = nn.Parameter(torch.zeros(batch_size, 69, dtype=dtype))
        if create_transl:
            self.transl = nn.Parameter(torch.zeros(batch_size, 3, dtype=dtype))
        self._derived = None
        self._ws = None

    @classmethod
    def synthetic(cls, seed: int = 1234, V: int = 6890, batch_size: int = 1):
        return cls(model_arrays=synthetic_model_arrays(seed, V), batch_size=batch_size)

    # ------------------------------------------------------------------ derived model (cached)
    def _model(self) -> L.SmplModel:
        bufs = (self.v_template, self.shapedirs, self.posedirs, self.J_regressor, self.lbs_weights, self.parents)
        key = tuple((b.data_ptr(), b._version) for b in bufs)
        if self._derived is not None and self._derived[0] == key:
            return self._derived[1]
        L.require_cuda(self.v_template, "SMPL buffers")
        V = self.v_template.shape[0]
        with torch.no_grad():
            blend = torch.zeros(V * 3, 224, device=self.v_template.device, dtype=torch.float32)
            blend[:, :10] = self.shapedirs.reshape(V * 3, 10)
            blend[:, 10:217] = self.posedirs.t()
            J_t = (self.J_regressor @ self.v_template).contiguous()                                  # [24,3]
            J_s = torch.einsum("jv,vcl->jcl", self.J_regressor, self.shapedirs).contiguous()        # [24,3,10]
            ids = self.vertex_joint_selector.extra_joints_idxs
            ex_t = self.v_template[ids].contiguous()
            ex_s = self.shapedirs[ids].reshape(63, 10).contiguous()
            ex_p = self.posedirs.reshape(207, V, 3)[:, ids].reshape(207, 63).contiguous()
            ex_w = self.lbs_weights[ids].contiguous()
            par = self.parents.to(torch.int32).contiguous()
            vt = self.v_template.contiguous()
            lw = self.lbs_weights.contiguous()
        m = L.SmplModel()
        m.V = V
        m.v_template, m.blend_w, m.lbs_weights = vt.data_ptr(), blend.data_ptr(), lw.data_ptr()
        m.J_template, m.J_shapedirs, m.parents = J_t.data_ptr(), J_s.data_ptr(), par.data_ptr()
        m.ex_template, m.ex_shapedirs, m.ex_posedirs, m.ex_weights = ex_t.data_ptr(), ex_s.data_ptr(), ex_p.data_ptr(), ex_w.data_ptr()
        self._derived = (key, m, (blend, J_t, J_s, ex_t, ex_s, ex_p, ex_w, par, vt, lw))
        return m

    # ------------------------------------------------------------------ smplx.SMPL.forward
    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, return_verts=True,
                return_full_pose=False, pose2rot: bool = True, **kwargs):
        global_orient = global_orient if global_orient is not None else self.global_orient
        body_pose = body_pose if body_pose is not None else self.body_pose
        betas = betas if betas is not None else self.betas
        if transl is None and hasattr(self, "transl"):
            transl = self.transl
        M = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != M:
            betas = betas.expand(M, -1)
        L.require_cuda(betas, "betas")
        if pose2rot:
            pose = torch.cat([global_orient.reshape(M, 3), body_pose.reshape(M, -1)], dim=1)
            if pose.shape[1] != 72:
                raise ValueError("body_pose must hold 23 joints (69 values); pad GIMO's 21 joints as mld.py:807-813 does")
        else:
            pose = torch.cat([global_orient.reshape(M, 1, 9), body_pose.reshape(M, -1, 9)], dim=1)
            if pose.shape[1] != 24:
                raise ValueError("rotation-matrix pose must hold 24 joints")
        pose = pose.to(torch.float32).contiguous()
        betas = betas.to(torch.float32).contiguous()
        if transl is not None and transl.shape[0] != M:      # the module's own parameter [batch_size, 3] (smplx broadcasts it)
            transl = transl.reshape(-1, 3).expand(M, 3)
        tr = None if transl is None else transl.to(torch.float32).reshape(M, 3).contiguous()
        dev = betas.device
        joints = torch.empty(M, 45, 3, device=dev, dtype=torch.float32)
        model = self._model()
        verts = None
        ws_ptr, ws_n = 0, 0
        if return_verts:
            verts = torch.empty(M, model.V, 3, device=dev, dtype=torch.float32)
            need = L.lib().seeme_smpl_workspace_bytes(M)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            ws_ptr, ws_n = self._ws.data_ptr(), self._ws.numel()
        rc = L.lib().seeme_smpl_lbs(C.byref(model), betas.data_ptr(), pose.data_ptr(), 0 if pose2rot else 1,
                                    L.ptr(tr), M, joints.data_ptr(), L.ptr(verts), ws_ptr, ws_n, L.current_stream())
        L.check(rc, "seeme_smpl_lbs")
        return SimpleNamespace(vertices=verts, joints=joints, betas=betas, body_pose=body_pose,
                               global_orient=global_orient, full_pose=pose if return_full_pose else None)


class _JointsAA(torch.autograd.Function):
    """The 24 posed joints as a differentiable function of the axis-angle pose [M,72] and the translation [M,3]: forward =
    the joints-only launch of ``seeme_smpl_lbs``, backward = ``seeme_smpl_joints_backward`` (hand-written; replaces autograd
    through smplx's lbs for the joints loss of ``train_vae_forward``, mld.py:764-773,871-878).  betas get no gradient."""

    @staticmethod
    def forward(ctx, smpl, betas, pose, transl):
        M = pose.shape[0]
        pose = pose.to(torch.float32).contiguous()
        betas = betas.to(torch.float32).contiguous()
        tr = None if transl is None else transl.to(torch.float32).contiguous()
        joints = torch.empty(M, 45, 3, device=pose.device, dtype=torch.float32)
        model = smpl._model()
        L.check(L.lib().seeme_smpl_lbs(C.byref(model), betas.data_ptr(), pose.data_ptr(), 0, L.ptr(tr), M, joints.data_ptr(), 0, 0, 0,
                                       L.current_stream()), "seeme_smpl_lbs")
        ctx.smpl, ctx.has_tr = smpl, tr is not None
        ctx.save_for_backward(betas, pose)
        return joints[:, :24]

    @staticmethod
    def backward(ctx, dj):
        betas, pose = ctx.saved_tensors
        M = pose.shape[0]
        dj = dj.to(torch.float32).contiguous()
        dpose = torch.empty_like(pose)
        dtr = torch.empty(M, 3, device=pose.device, dtype=torch.float32) if ctx.has_tr else None
        model = ctx.smpl._model()
        L.check(L.lib().seeme_smpl_joints_backward(C.byref(model), betas.data_ptr(), pose.data_ptr(), dj.data_ptr(), 24, dpose.data_ptr(),
                                                    L.ptr(dtr), M, L.current_stream()), "seeme_smpl_joints_backward")
        return None, None, dpose, dtr


def smpl_joints_hip(smpl, betas: torch.Tensor, pose_aa: torch.Tensor, transl: torch.Tensor = None) -> torch.Tensor:
    """[M,24,3] posed joints, differentiable w.r.t. pose_aa [M,72] and transl [M,3] (HIP forward and backward)."""
    L.require_cuda(pose_aa, "pose_aa")
    return _JointsAA.apply(smpl, betas, pose_aa, transl)
