"""Batch producers for the path: the EgoBody / GIMO on-disk layout of the reference, read once, kept RESIDENT (HBM or pinned
host memory) and served as the reference's batch tuples.

Reference: ``mld/data/humanml/data/dataset.py`` -- ``EgoBodyData3`` (:1055-1794) and ``GimoData`` (:1797-2509) -- wrapped by the
Lightning data modules ``mld/data/EgoBody.py`` / ``mld/data/Gimo.py`` (mean / std, ``renorm`` :151-157 / :139-145) and torch's
default collate.  There every ``__getitem__`` re-does, in numpy on DataLoader workers, the same work for the same sequence:
zero-pad to ``motion_length``, normalise pose / global orientation / translation with slices of ``mean.npy`` / ``std.npy``,
look the scene cloud up and move it into the camera frame.  None of it depends on the epoch, so here it is done ONCE at load
time, vectorised, and the whole split then lives in device memory (a split is a few thousand 60-frame sequences: megabytes;
the scene clouds are a table of [n_scenes, 20000, 3] -- 288 GB of HBM hold any of it); a batch is an index-select on the
device (plus the per-sequence rigid transform of its scene cloud), i.e. no host work, no PCIe traffic and no worker
processes in the training loop.  ``storage="pinned"`` keeps the split in pinned host memory instead and copies batches
asynchronously.

On-disk layout (what the reference reads; the datasets themselves are licence-gated and absent):

    <root>/mean.npy, <root>/std.npy                 [1, >= numdims + 3]: global orientation | body pose | ... translation
    <root>/<split>/<name>.npy                       one pickled dict per sequence (``np.save(path, dict)``):
        video [L] frame names, recording_utils {original_imgname [L], fx, cx, cy [L], center [L,2], scale [L]},
        wearer / interactee {global_orient [L,1,3], transl [L,1,3], body_pose [L,1,69 | 63], betas [L,1,10]}
    EgoBody scenes (``condition`` contains 'scene'), under <scene_root>:
        map_dict_<split>.pkl {image name -> scene key}, pcd_verts_dict_<split>.pkl {scene key -> [P,3] cloud, kinect frame},
        transf_matrices_all_seqs.pkl {sequence -> {trans_kinect2holo [4,4], trans_world2pv {timestamp -> [4,4]}}}
    GIMO scenes: <scene_root>/<scene>/scene_obj/{transform_norm.txt, scene_points.npy}; the reference reads the vertices of
        ``scene_downsampled.ply`` through trimesh (absent here) -- a one-off conversion of ALL mesh vertices to ``scene_points.npy``
        replaces that dependency.  As in the reference (dataset.py:2013-2033) every item draws 20000 vertices with replacement
        from its scene on every access, and the train split adds N(0, 0.01) jitter: done on the device from a seedable generator
        (or from injected draws, for the tests).
    Optional <root>/interactee_pred_<split>.pkl {image name -> {"smpl_parameters": {global_orient, body_pose, betas}}}: EgoHMR
        estimates that replace the interactee's pose as the CONDITION (dataset.py:1215-1223, 1300-1321; translation stays the file's).

Files are read with loaders that execute nothing: ``np.load(allow_pickle=False)`` for arrays, and for the pickled ``.npy`` /
``.pkl`` containers an unpickler that only admits numpy array reconstruction and plain containers.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import geometry as G

ADD_TRANS = np.array([[1.0, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])      # dataset.py:1195-1197


# ----------------------------------------------------------------------------- code-free readers
class _ArraysOnlyUnpickler(pickle.Unpickler):
    """dicts / lists / strings / numbers are pickle opcodes; the only globals a sequence file or a scene table needs are the
    numpy array / dtype / scalar reconstructors.  Everything else -- the way a pickle runs code -- is refused."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"), ("collections", "OrderedDict"),
                ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            import importlib
            return getattr(importlib.import_module(module), name)
        raise pickle.UnpicklingError(f"data file refers to {module}.{name}: only numpy arrays and plain containers are accepted")


def load_pickled(path: str):
    """A ``.pkl`` or an object-array ``.npy`` (``np.save(path, dict)``) without executing anything from it."""
    with open(path, "rb") as f:
        if path.endswith(".npy"):
            version = np.lib.format.read_magic(f)
            np.lib.format._check_version(version)
            shape, _fortran, dtype = np.lib.format._read_array_header(f, version)
            if not dtype.hasobject:
                f.seek(0)
                return np.load(f, allow_pickle=False)
            obj = _ArraysOnlyUnpickler(f, encoding="latin1").load()
            return obj.item() if isinstance(obj, np.ndarray) and obj.shape == () else obj
        return _ArraysOnlyUnpickler(f, encoding="latin1").load()


# ----------------------------------------------------------------------------- one split, resident
class EgoSequenceSplit:
    """One split of EgoBody (``EgoBodyData3``) or GIMO (``GimoData``), preprocessed once into stacked tensors."""

    def __init__(self, root: str, split: str, dataset: str = "egobody", condition: Sequence[str] = ("text", "interactee"),
                 motion_length: int = 60, data_type: str = "angle", predict_transl: bool = True,
                 pose_estimation_task: bool = False, scene_root: Optional[str] = None, max_items: Optional[int] = None,
                 interactee_pred: Optional[str] = None, scene_points: int = 20000):
        if data_type != "angle":
            raise NotImplementedError("data module: DATA_TYPE 'angle' (the rot6d variant re-encodes the same files)")
        self.dataset, self.split, self.condition = dataset, split, tuple(condition)
        self.motion_length, self.predict_transl = int(motion_length), bool(predict_transl)
        self.pose_estimation_task = bool(pose_estimation_task)
        self.numdims, self.go_dims = (72, 3) if dataset == "egobody" else (66, 3)          # dataset.py:1087-1088, 1828-1829
        self.pose_dim = self.numdims - self.go_dims                                          # 69 | 63
        self.mean = np.load(os.path.join(root, "mean.npy"), allow_pickle=False).astype(np.float32)
        self.std = np.load(os.path.join(root, "std.npy"), allow_pickle=False).astype(np.float32)
        d = os.path.join(root, "test" if (dataset == "gimo" and split == "val") else split)   # GIMO has no val split (:1842-1843)
        names = sorted(n for n in os.listdir(d) if n.endswith(".npy"))
        if max_items is not None:
            names = names[:max_items]
        T = self.motion_length
        N = len(names)
        if N == 0:
            raise FileNotFoundError(f"no sequence files under {d}")
        motion = np.zeros((N, T, 2, self.numdims), np.float32)
        transl = np.zeros((N, 2, T, 3), np.float32)
        beta = np.zeros((N, 2, T, 10), np.float32)
        utils = np.zeros((N, T, 6), np.float32)
        length = np.zeros((N, 1), np.int32)
        self.names, self.images = names, []
        self.scene_points = int(scene_points)
        pred = load_pickled(interactee_pred) if interactee_pred else None            # EgoHMR estimates (dataset.py:1215-1223)
        pe = self.pose_estimation_task
        # POSE_ESTIMATION_TASK: the file's own interactee is the ground truth the estimate is scored against (dataset.py:1255-1256,
        # 1333-1342); without `interactee_pred` it coincides with the condition
        pe_motion = np.zeros((N, T, 1, self.numdims), np.float32) if pe else None
        pe_beta = np.zeros((N, T, 1, 10), np.float32) if pe else None
        first_image: List[str] = []
        m, s = self.mean[0], self.std[0]
        t_lo = self.numdims if dataset == "egobody" else m.shape[0] - 3                    # EgoBody: [numdims, +3); GIMO: the last three (:1607-1612, 2360-2364)
        for i, name in enumerate(names):
            it = load_pickled(os.path.join(d, name))
            ru = it["recording_utils"]
            L = len(it["video"])
            if L > T:
                raise ValueError(f"{name}: {L} frames > motion_length {T}")
            length[i, 0] = L
            self.images.append([str(x) for x in (ru["original_imgname"] if dataset == "egobody" else it["video"])])
            first_image.append(self.images[-1][0])
            for p, who in enumerate(("wearer", "interactee")):
                sp = it[who]
                go = np.zeros((T, 3), np.float32)
                bp = np.zeros((T, self.pose_dim), np.float32)
                tr = np.zeros((T, 3), np.float32)
                go[:L] = np.asarray(sp["global_orient"], np.float32).reshape(L, 3)
                bp[:L] = np.asarray(sp["body_pose"], np.float32).reshape(L, -1)[:, : self.pose_dim]
                tr[:L] = np.asarray(sp["transl"], np.float32).reshape(L, 3)
                bt = np.asarray(sp["betas"], np.float32).reshape(L, 10)
                if p == 1 and pe:
                    pe_motion[i, :, 0, : self.go_dims] = (go - m[: self.go_dims]) / s[: self.go_dims]
                    pe_motion[i, :, 0, self.go_dims:] = (bp - m[self.go_dims: self.numdims]) / s[self.go_dims: self.numdims]
                    pe_beta[i, :L, 0] = bt
                if p == 1 and pred is not None:                                                # estimates, per image (:1300-1321)
                    est = [pred[str(im)]["smpl_parameters"] for im in self.images[-1][:L]]
                    go[:L] = np.stack([np.asarray(e["global_orient"], np.float32).reshape(3) for e in est])
                    bp[:L] = np.stack([np.asarray(e["body_pose"], np.float32).reshape(-1)[: self.pose_dim] for e in est])
                    bt = np.stack([np.asarray(e["betas"], np.float32).reshape(10) for e in est])
                # zero padding comes BEFORE the normalisation, as in the reference (:1524-1546): padded frames are -mean/std
                motion[i, :, p, : self.go_dims] = (go - m[: self.go_dims]) / s[: self.go_dims]
                motion[i, :, p, self.go_dims:] = (bp - m[self.go_dims: self.numdims]) / s[self.go_dims: self.numdims]
                if self.predict_transl:
                    tr = (tr - m[t_lo: t_lo + 3]) / s[t_lo: t_lo + 3]
                transl[i, p] = tr
                beta[i, p, :L] = bt
            cols = [np.asarray(ru[k], np.float32).reshape(L, -1) for k in ("fx", "cx", "cy", "center", "scale")]
            utils[i, :L] = np.concatenate(cols, axis=1)                                        # [L, 1+1+1+2+1] (:1584-1586)
        self.motion, self.transl, self.beta = torch.from_numpy(motion), torch.from_numpy(transl), torch.from_numpy(beta)
        self.utils, self.length = torch.from_numpy(utils), torch.from_numpy(length)
        self.pe_motion = torch.from_numpy(pe_motion) if pe else None
        self.pe_beta = torch.from_numpy(pe_beta) if pe else None
        self.scene_table = self.scene_index = self.scene_xform = self.scene_flat = self.scene_off = self.scene_cnt = None
        if "scene" in self.condition:
            self._load_scenes(scene_root or root, first_image)

    def _load_scenes(self, scene_root: str, first_image: List[str]):
        N = len(first_image)
        xform = np.zeros((N, 4, 4), np.float32)
        keys: Dict[str, int] = {}
        clouds: List[np.ndarray] = []
        idx = np.zeros(N, np.int64)
        if self.dataset == "egobody":                                                          # dataset.py:1198-1214, 1265-1286
            smap = load_pickled(os.path.join(scene_root, f"map_dict_{self.split}.pkl"))
            verts = load_pickled(os.path.join(scene_root, f"pcd_verts_dict_{self.split}.pkl"))
            tm = load_pickled(os.path.join(scene_root, "transf_matrices_all_seqs.pkl"))
            for i, img in enumerate(first_image):
                seq, ts = img.split("/")[1], img.split("/")[4].split("_")[0]
                k2h = np.asarray(tm[seq]["trans_kinect2holo"], np.float32)
                h2pv = np.asarray(tm[seq]["trans_world2pv"][str(ts)], np.float32)
                xform[i] = (ADD_TRANS @ (h2pv @ k2h)).astype(np.float32)
                key = smap[img]
                if key not in keys:
                    keys[key] = len(clouds)
                    clouds.append(np.asarray(verts[key], np.float32))
                idx[i] = keys[key]
        else:                                                                                  # GIMO: dataset.py:1989-2033
            scale = 1.03
            for i, img in enumerate(first_image):
                key = img.split("/")[-4]
                if key not in keys:
                    base = os.path.join(scene_root, key, "scene_obj")
                    keys[key] = len(clouds)
                    clouds.append(np.load(os.path.join(base, "scene_points.npy"), allow_pickle=False).astype(np.float32) / scale)
                tn = np.loadtxt(os.path.join(scene_root, key, "scene_obj", "transform_norm.txt")).reshape(4, 4).astype(np.float32)
                tn[:3, 3] /= scale
                xform[i], idx[i] = tn, keys[key]
        self.scene_index, self.scene_xform = torch.from_numpy(idx), torch.from_numpy(xform)
        if self.dataset == "egobody":                # one fixed cloud per scene key, used as it is (dataset.py:1265-1286)
            P = min(c.shape[0] for c in clouds)
            self.scene_table = torch.from_numpy(np.stack([c[:P] for c in clouds]))          # [S,P,3]
        else:                                        # GIMO: ALL vertices of every scene; items sample from them on every access
            cnt = np.array([c.shape[0] for c in clouds], np.int64)
            self.scene_flat = torch.from_numpy(np.concatenate(clouds, axis=0))              # [sum V,3]
            self.scene_cnt = torch.from_numpy(cnt)
            self.scene_off = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64))

    def __len__(self):
        return self.motion.shape[0]

    def to(self, device, pinned: bool = False):
        for k in ("motion", "transl", "beta", "utils", "length", "scene_table", "scene_index", "scene_xform", "scene_flat", "scene_off",
                  "scene_cnt", "pe_motion", "pe_beta"):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, t.pin_memory() if pinned else t.to(device))
        return self

    @property
    def has_scene(self) -> bool:
        return self.scene_table is not None or self.scene_flat is not None

    def scenes(self, index: torch.Tensor, generator: Optional[torch.Generator] = None, draws=None) -> torch.Tensor:
        """The scene clouds of a batch of items, in the frame the reference hands to the model.
        EgoBody: points_coord_trans(cloud, M) = cloud @ M[:3,:3]^T + M[:3,3] (utils_egobody/geometry.py:324-328).
        GIMO (dataset.py:2013-2033): per item `scene_points` vertices drawn with replacement from ALL vertices of its scene
        (np.random.choice there; here floor(u V) from `generator` on the device), the norm transform, and in the train split
        N(0, 0.01) jitter.  `draws` = (u [B,P] in [0,1), noise [B,P,3] standard normal or None) injects the random numbers."""
        M = self.scene_xform.index_select(0, index)
        if self.scene_table is not None:
            pts = self.scene_table.index_select(0, self.scene_index.index_select(0, index))
            return torch.baddbmm(M[:, None, :3, 3], pts, M[:, :3, :3].transpose(1, 2))
        dev, B, P = self.scene_flat.device, index.shape[0], self.scene_points
        sc = self.scene_index.index_select(0, index)
        cnt, off = self.scene_cnt.index_select(0, sc), self.scene_off.index_select(0, sc)
        u = draws[0].to(dev) if draws is not None else torch.rand(B, P, device=dev, generator=generator)
        pick = torch.minimum((u.double() * cnt[:, None].double()).long(), cnt[:, None] - 1) + off[:, None]
        pts = self.scene_flat.index_select(0, pick.reshape(-1)).reshape(B, P, 3)
        pts = torch.baddbmm(M[:, None, :3, 3], pts, M[:, :3, :3].transpose(1, 2))
        if self.split == "train":
            noise = draws[1] if draws is not None else torch.randn(B, P, 3, device=dev, generator=generator)
            if noise is not None:
                pts = pts + 0.01 * noise.to(dev)
        return pts

    def item(self, i: int):
        """The tuple ``__getitem__`` returns in the reference (:1754-1794, 2479-2509), for one sequence."""
        ix = torch.tensor([i], device=self.motion.device)
        out = [self.motion[i], self.transl[i], self.beta[i], self.utils[i]]
        if self.has_scene:
            out.append(self.scenes(ix)[0])
        out.append(self.length[i])
        if self.pose_estimation_task:            # (motion_interacte_pe_gt, interactee_transl_pe_gt, interactee_beta_pe_gt), :1765-1781
            out += [self.pe_motion[i], self.transl[i, 1:2], self.pe_beta[i]]
        elif self.has_scene:
            out.append(self.images[i])
        return tuple(out)


class EgoDataModule:
    """``mld/data/EgoBody.py:EgoBodyDataModule`` / ``mld/data/Gimo.py:GimoDataModule`` without Lightning: ``renorm``, ``nfeats``,
    ``njoints`` and batch iterators in place of the DataLoaders.  ``batch(B, idx, ...)`` has the signature of
    ``SyntheticEgoDataModule.batch`` so that ``seeme_amd.cli`` runs on either."""

    def __init__(self, root: str, dataset: str = "egobody", condition: Sequence[str] = ("text", "interactee"),
                 motion_length: int = 60, predict_transl: bool = True, device="cuda", storage: str = "device",
                 scene_root: Optional[str] = None, pose_estimation_task: bool = False, splits: Sequence[str] = ("train", "val", "test"),
                 max_items: Optional[int] = None, interactee_pred: bool = False, scene_points: int = 20000, seed: int = 1234):
        if storage not in ("device", "pinned"):
            raise ValueError("storage: 'device' (split resident in HBM) or 'pinned' (pinned host memory, async copies)")
        self.name, self.device, self.storage = dataset, torch.device(device), storage
        self.njoints = 23 if dataset == "egobody" else 21
        self.numdims = (75 if predict_transl else 72) if dataset == "egobody" else (69 if predict_transl else 66)   # EgoBody.py:128, Gimo.py:119
        self.nfeats = self.numdims
        self.is_mm = False
        self.splits: Dict[str, EgoSequenceSplit] = {}
        for sp in splits:
            d = os.path.join(root, "test" if (dataset == "gimo" and sp == "val") else sp)
            if os.path.isdir(d):
                pred = os.path.join(root, f"interactee_pred_{sp}.pkl") if interactee_pred else None
                s = EgoSequenceSplit(root, sp, dataset, condition, motion_length, "angle", predict_transl, pose_estimation_task,
                                     scene_root, max_items, pred, scene_points)
                self.splits[sp] = s.to(self.device, pinned=(storage == "pinned"))
        if not self.splits:
            raise FileNotFoundError(f"no split directory under {root}")
        any_split = next(iter(self.splits.values()))
        self.mean = torch.from_numpy(any_split.mean).to(self.device)
        self.std = torch.from_numpy(any_split.std).to(self.device)
        self.with_scene = any_split.has_scene
        self.pose_estimation_task = pose_estimation_task
        self.generator = torch.Generator(device=self.device if storage == "device" else "cpu").manual_seed(int(seed))   # scene sampling / jitter

    def renorm(self, features):
        """features * std[0, :numdims] + mean[0, :numdims]  (EgoBody.py:151-157, Gimo.py:139-145).  For GIMO the translation
        statistics are the LAST three entries of mean / std (dataset.py:2360-2364), for EgoBody entries [72, 75)."""
        if self.name == "gimo" and self.mean.shape[1] != self.numdims:
            idx = list(range(self.numdims - 3)) + list(range(self.mean.shape[1] - 3, self.mean.shape[1]))
            return G.renorm(features, self.mean[:, idx].contiguous(), self.std[:, idx].contiguous())
        return G.renorm(features, self.mean, self.std)

    def collate(self, split: str, index: torch.Tensor):
        """default_collate of the reference's items, from the resident tensors: (motion [B,T,2,P], transl [B,2,T,3],
        beta [B,2,T,10], utils [B,T,6], [scene [B,P,3]], length [B,1], [image names])."""
        s = self.splits[split]
        ix = index.to(s.motion.device)
        sel = lambda t: t.index_select(0, ix)
        out = [sel(s.motion), sel(s.transl), sel(s.beta), sel(s.utils)]
        if s.has_scene:
            out.append(s.scenes(ix, generator=self.generator))
        out.append(sel(s.length))
        if self.pose_estimation_task:            # the interactee's ground truth closes the tuple (dataset.py:1765-1781; MLD.ego_eval batch[-3:])
            out += [sel(s.pe_motion), sel(s.transl)[:, 1:2].contiguous(), sel(s.pe_beta)]
        if self.storage == "pinned":
            out = [t.pin_memory().to(self.device, non_blocking=True) for t in out]
        if s.has_scene and not self.pose_estimation_task:
            out.append([s.images[int(i)] for i in index.tolist()])
        return tuple(out)

    def iterate(self, split: str, batch_size: int, shuffle: bool = False, seed: int = 0, epoch: int = 0, rank: int = 0, world: int = 1,
                drop_last: bool = False):
        """One epoch of batches for this rank: the permutation depends on (seed, epoch) only, ranks take strided slices of it
        (what DistributedSampler does for Lightning's DDP, train.py:127-149)."""
        n = len(self.splits[split])
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed * 100003 + epoch)) if shuffle else torch.arange(n)
        order = order[rank::world]
        for lo in range(0, len(order), batch_size):
            ix = order[lo:lo + batch_size]
            if drop_last and len(ix) < batch_size:
                break
            yield self.collate(split, ix)

    def batch(self, B, idx=0, with_scene=None, lengths=None, pose_estimation=False, split: str = "train"):
        """Batch number `idx` of an endless shuffled stream over `split` (the interface seeme_amd.cli trains / tests on)."""
        sp = split if split in self.splits else next(iter(self.splits))
        n = len(self.splits[sp])
        per_epoch = max(1, n // B)
        epoch, k = divmod(int(idx) % (10 ** 9), per_epoch)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(1234 * 100003 + epoch))
        ix = order[k * B:(k + 1) * B]
        if len(ix) < B:
            ix = torch.cat([ix, order[: B - len(ix)]])
        return self.collate(sp, ix)
