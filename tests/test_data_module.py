"""Batch producers (seeme_amd/data.py) against a per-item restatement of the reference's ``EgoBodyData3.__getitem__``
(mld/data/humanml/data/dataset.py:1245-1794) / ``GimoData.__getitem__`` (:1968-2509) and torch's default collate, on files
this test writes in the reference's on-disk layout (the datasets themselves are licence-gated).  CPU-only (storage on the
CPU device); the GPU variant in test_gpu_flows.py also covers ``renorm`` and a training step on these batches."""
import os
import pickle

import numpy as np
import pytest
import torch

from seeme_amd import data as D


def write_dataset(root, dataset="egobody", n=7, T=12, P=50, seed=0, with_scene=True, full_every=None):
    """A tiny dataset in the reference's layout.  Sequences have ragged lengths <= T (every `full_every`-th one is T long:
    the VAE mask is built from max(lengths), so -- as in the reference -- a batch needs one full-length member)."""
    rng = np.random.default_rng(seed)
    pose = 69 if dataset == "egobody" else 63
    nd = 72 if dataset == "egobody" else 66
    os.makedirs(root, exist_ok=True)
    np.save(os.path.join(root, "mean.npy"), rng.standard_normal((1, nd + 3 + (4 if dataset == "gimo" else 0))).astype(np.float32) * 0.1)
    np.save(os.path.join(root, "std.npy"), (0.5 + rng.random((1, nd + 3 + (4 if dataset == "gimo" else 0)))).astype(np.float32))
    items = {}
    smap, verts, tm = {}, {}, {}
    for split in ("train", "test"):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        for i in range(n):
            L = int(rng.integers(T // 2, T + 1))
            if full_every and i % full_every == 0:
                L = T
            seq = f"recording_{split}_{i // 2}"
            ts = [1000 + 7 * i + k for k in range(L)]
            if dataset == "egobody":
                imgs = [f"egocentric_color/{seq}/2021-09-07-{i}/PV/{t}_frame_{k:05d}.jpg" for k, t in enumerate(ts)]
                video = [f"frame_{k}" for k in range(L)]
            else:
                imgs = [f"x/{k}" for k in range(L)]
                video = [f"gimo/scene_{i % 3}/seq{i}/cam/frame_{k:05d}.jpg" for k in range(L)]      # scene = split('/')[-4]
            person = lambda: {"global_orient": rng.standard_normal((L, 1, 3)), "transl": rng.standard_normal((L, 1, 3)),
                              "body_pose": rng.standard_normal((L, 1, pose)) * 0.3, "betas": np.repeat(rng.standard_normal((1, 1, 10)), L, 0)}
            it = {"video": video, "recording_utils": {"original_imgname": imgs, "fx": list(rng.random(L)), "cx": list(rng.random(L)),
                                                       "cy": list(rng.random(L)), "center": rng.random((L, 2)), "scale": list(rng.random(L))},
                  "wearer": person(), "interactee": person()}
            name = f"seq_{i:03d}.npy"
            np.save(os.path.join(root, split, name), it, allow_pickle=True)       # what the reference's preprocessing writes
            items[(split, name)] = it
            if with_scene and dataset == "egobody":
                key = f"scene_{i % 3}"
                smap[imgs[0]] = key
                verts.setdefault(key, rng.uniform(-3, 3, (P, 3)))
                e = tm.setdefault(seq, {"trans_kinect2holo": np.eye(4) + 0.1 * rng.standard_normal((4, 4)), "trans_world2pv": {}})
                e["trans_world2pv"][str(ts[0])] = np.eye(4) + 0.1 * rng.standard_normal((4, 4))
            if with_scene and dataset == "gimo":
                base = os.path.join(root, f"scene_{i % 3}", "scene_obj")
                os.makedirs(base, exist_ok=True)
                if not os.path.exists(os.path.join(base, "scene_points.npy")):
                    np.save(os.path.join(base, "scene_points.npy"), rng.uniform(-3, 3, (P + 5 * (i % 3), 3)).astype(np.float32))   # ragged vertex counts
                    np.savetxt(os.path.join(base, "transform_norm.txt"), (np.eye(4) + 0.1 * rng.standard_normal((4, 4))).reshape(-1))
        if with_scene and dataset == "egobody":
            for fn, obj in ((f"map_dict_{split}.pkl", smap), (f"pcd_verts_dict_{split}.pkl", verts)):
                with open(os.path.join(root, fn), "wb") as f:
                    pickle.dump(obj, f)
    if with_scene and dataset == "egobody":
        with open(os.path.join(root, "transf_matrices_all_seqs.pkl"), "wb") as f:
            pickle.dump(tm, f)
    return items, (smap, verts, tm)


def reference_item(it, mean, std, T, dataset, scene=None):
    """__getitem__ of the reference, line by line in numpy (dataset.py:1320-1643 / 2150-2396), 'angle' data, predict_transl."""
    nd, go = (72, 3) if dataset == "egobody" else (66, 3)
    L = len(it["video"])
    pad = T - L
    out_m, out_t, out_b = [], [], []
    for who in ("wearer", "interactee"):
        sp = it[who]
        bp = np.concatenate([np.asarray(sp["body_pose"]), np.zeros((pad, 1, nd - go))], 0).reshape(T, -1)
        bp = (bp - mean[0, go:nd]) / std[0, go:nd]
        g = np.concatenate([np.asarray(sp["global_orient"]), np.zeros((pad, 1, 3))], 0).reshape(T, 3)
        g = (g - mean[0, :go]) / std[0, :go]
        out_m.append(np.concatenate([g, bp], -1))
        tr = np.concatenate([np.asarray(sp["transl"]), np.zeros((pad, 1, 3))], 0).reshape(T, 3)
        lo = nd if dataset == "egobody" else mean.shape[1] - 3
        out_t.append((tr - mean[0, lo:lo + 3]) / std[0, lo:lo + 3])
        out_b.append(np.concatenate([np.asarray(sp["betas"]), np.zeros((pad, 1, 10))], 0).reshape(T, 10))
    ru = it["recording_utils"]
    utils = np.concatenate([np.asarray(ru[k]).reshape(L, -1) for k in ("fx", "cx", "cy", "center", "scale")], 1)
    utils = np.concatenate([utils, np.zeros((pad, 6))], 0)
    res = {"motion": np.stack(out_m, 1), "transl": np.stack(out_t, 0), "beta": np.stack(out_b, 0), "utils": utils, "length": L}
    if scene is not None:
        smap, verts, tm = scene
        img = ru["original_imgname"][0]
        seq, ts = img.split("/")[1], img.split("/")[4].split("_")[0]
        M = D.ADD_TRANS @ (np.asarray(tm[seq]["trans_world2pv"][ts]) @ np.asarray(tm[seq]["trans_kinect2holo"]))
        v = verts[smap[img]]
        res["scene"] = v.dot(M[:3, :3].T) + M[:3, 3][None]
    return res


@pytest.mark.parametrize("dataset", ["egobody", "gimo"])
def test_split_matches_reference_items(tmp_path, dataset):
    root = str(tmp_path / dataset)
    items, scene = write_dataset(root, dataset, with_scene=True)
    dm = D.EgoDataModule(root, dataset, condition=("text", "scene", "interactee"), motion_length=12, device="cpu", scene_root=root)
    # (GIMO has no val split: the reference reads the test files for it, dataset.py:1842-1843)
    assert dm.nfeats == (75 if dataset == "egobody" else 69) and set(dm.splits) == ({"train", "test"} if dataset == "egobody" else {"train", "val", "test"})
    mean, std = np.load(os.path.join(root, "mean.npy")), np.load(os.path.join(root, "std.npy"))
    for split in ("train", "test"):
        s = dm.splits[split]
        for i, name in enumerate(s.names):
            want = reference_item(items[(split, name)], mean, std, 12, dataset, scene if dataset == "egobody" else None)
            got = s.item(i)
            np.testing.assert_allclose(got[0].numpy(), want["motion"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(got[1].numpy(), want["transl"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(got[2].numpy(), want["beta"], rtol=1e-6)
            np.testing.assert_allclose(got[3].numpy(), want["utils"], rtol=1e-6)
            assert int(got[5]) == want["length"] and got[4].shape[1] == 3
            if dataset == "egobody":
                np.testing.assert_allclose(got[4].numpy(), want["scene"], rtol=1e-4, atol=1e-5)
                assert got[6] == items[(split, name)]["recording_utils"]["original_imgname"]


def test_batches_are_collated_items_and_shards_are_disjoint(tmp_path):
    root = str(tmp_path / "egobody")
    write_dataset(root, "egobody", n=9)
    dm = D.EgoDataModule(root, "egobody", condition=("text", "scene"), motion_length=12, device="cpu", scene_root=root)
    s = dm.splits["train"]
    seen = []
    for rank in range(2):
        for b in dm.iterate("train", 2, shuffle=True, seed=3, epoch=1, rank=rank, world=2):
            motion, transl, beta, utils, scene, length, names = b
            assert motion.shape[1:] == (12, 2, 72) and transl.shape[1:] == (2, 12, 3) and beta.shape[1:] == (2, 12, 10)
            assert utils.shape[1:] == (12, 6) and scene.shape[1:] == (50, 3) and length.shape[1:] == (1,)
            for j in range(motion.shape[0]):                 # every row is one item of the split, unchanged
                hit = [i for i in range(len(s)) if torch.equal(s.motion[i], motion[j])]
                assert len(hit) == 1 and torch.equal(s.item(hit[0])[4], scene[j]) and names[j] == s.images[hit[0]]
                seen.append(hit[0])
    assert sorted(seen) == list(range(9))                    # both ranks together: every sequence exactly once
    a = [b[0] for b in dm.iterate("train", 4, shuffle=True, seed=3, epoch=0)]
    c = [b[0] for b in dm.iterate("train", 4, shuffle=True, seed=3, epoch=1)]
    assert not all(torch.equal(x, y) for x, y in zip(a, c))   # a new permutation per epoch
    b0 = dm.batch(4, idx=0, split="train")
    assert b0[0].shape == (4, 12, 2, 72) and len(b0) == 7


def test_sequence_files_are_read_without_executing_them(tmp_path):
    root = str(tmp_path / "e")
    write_dataset(root, "egobody", n=2, with_scene=False)

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > " + str(tmp_path / "pwned"),))

    np.save(os.path.join(root, "train", "zzz_evil.npy"), {"video": [Evil()]}, allow_pickle=True)
    with pytest.raises(pickle.UnpicklingError):
        D.EgoSequenceSplit(root, "train", "egobody", ("text", "interactee"), 12)
    assert not (tmp_path / "pwned").exists()


def test_gimo_scene_clouds_are_resampled_per_access_and_jittered_in_training(tmp_path):
    """GimoData.__getitem__ (dataset.py:2013-2033): 20000 vertices drawn WITH replacement from all vertices of the item's scene on
    every access, / 1.03, the norm transform, and -- train split only -- N(0, 0.01) jitter.  The module does it on the device for a
    whole batch; with the random numbers injected it must equal the per-item restatement, for scenes of different vertex counts."""
    root = str(tmp_path / "gimo")
    items, _ = write_dataset(root, "gimo", n=7, with_scene=True)
    Pn = 64
    dm = D.EgoDataModule(root, "gimo", condition=("text", "scene"), motion_length=12, device="cpu", scene_root=root, scene_points=Pn)
    rng = np.random.default_rng(5)
    for split in ("train", "test"):
        s = dm.splits[split]
        ix = torch.arange(len(s))
        u = torch.from_numpy(rng.random((len(s), Pn)).astype(np.float32))
        nz = torch.from_numpy(rng.standard_normal((len(s), Pn, 3)).astype(np.float32))
        got = s.scenes(ix, draws=(u, nz)).numpy()
        for i, name in enumerate(s.names):
            scene = items[(split, name)]["video"][0].split("/")[-4]
            verts = np.load(os.path.join(root, scene, "scene_obj", "scene_points.npy")).astype(np.float64)
            tn = np.loadtxt(os.path.join(root, scene, "scene_obj", "transform_norm.txt")).reshape(4, 4)
            tn[:3, 3] /= 1.03
            pick = np.minimum((u[i].numpy().astype(np.float64) * len(verts)).astype(np.int64), len(verts) - 1)
            pts = verts[pick] * (1 / 1.03)
            pts = (tn[:3, :3] @ pts.T + tn[:3, 3:]).T
            if split == "train":
                pts = pts + 0.01 * nz[i].numpy()
            np.testing.assert_allclose(got[i], pts, rtol=1e-4, atol=1e-5)
        # the jitter is a train-split thing; every vertex of the scene can be drawn
        if split == "test":
            assert np.array_equal(got, s.scenes(ix, draws=(u, None)).numpy())
    b1, b2 = dm.collate("train", torch.arange(3)), dm.collate("train", torch.arange(3))
    assert b1[4].shape == (3, Pn, 3) and not torch.equal(b1[4], b2[4])          # a new draw per access
    dm2 = D.EgoDataModule(root, "gimo", condition=("text", "scene"), motion_length=12, device="cpu", scene_root=root, scene_points=Pn)
    assert torch.equal(dm2.collate("train", torch.arange(3))[4], b1[4])         # ... reproducible from the seed


@pytest.mark.parametrize("with_pred", [False, True])
def test_pose_estimation_batches_end_with_the_interactee_ground_truth(tmp_path, with_pred):
    """TEST.POSE_ESTIMATION_TASK (dataset.py:1255-1256, 1333-1342, 1765-1781): the tuple ends with the interactee of the FILE --
    motion [B,T,1,72] (normalised), transl [B,1,T,3] (normalised), betas -- in place of the image names; with EgoHMR estimates
    (`interactee_pred`, :1300-1321) the condition slot carries the estimates (translation stays the file's) and the ground truth does not."""
    root = str(tmp_path / "egobody")
    items, scene = write_dataset(root, "egobody", n=5, with_scene=True)
    mean, std = np.load(os.path.join(root, "mean.npy")), np.load(os.path.join(root, "std.npy"))
    rng = np.random.default_rng(9)
    if with_pred:
        for split in ("train", "test"):
            pred = {}
            for (sp, name), it in items.items():
                if sp == split:
                    for im in it["recording_utils"]["original_imgname"]:
                        pred[im] = {"smpl_parameters": {"global_orient": rng.standard_normal(3), "body_pose": rng.standard_normal(69) * 0.3,
                                                        "betas": rng.standard_normal(10)}}
            with open(os.path.join(root, f"interactee_pred_{split}.pkl"), "wb") as f:
                pickle.dump(pred, f)
            preds = pred if split == "test" else None
    dm = D.EgoDataModule(root, "egobody", condition=("text", "scene", "interactee"), motion_length=12, device="cpu", scene_root=root,
                         pose_estimation_task=True, interactee_pred=with_pred)
    s = dm.splits["test"]
    b = dm.collate("test", torch.arange(len(s)))
    assert len(b) == 9 and all(torch.is_tensor(t) for t in b)
    motion, transl, beta, utils, sc, length, g_motion, g_transl, g_beta = b
    assert g_motion.shape == (len(s), 12, 1, 72) and g_transl.shape == (len(s), 1, 12, 3) and g_beta.shape == (len(s), 12, 1, 10)
    for i, name in enumerate(s.names):
        want = reference_item(items[("test", name)], mean, std, 12, "egobody")
        np.testing.assert_allclose(g_motion[i, :, 0].numpy(), want["motion"][:, 1], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g_transl[i, 0].numpy(), want["transl"][1], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g_beta[i, :, 0].numpy(), want["beta"][1], rtol=1e-6)
        np.testing.assert_allclose(transl[i].numpy(), want["transl"], rtol=1e-5, atol=1e-6)           # translation: never from the estimates
        np.testing.assert_allclose(motion[i, :, 0].numpy(), want["motion"][:, 0], rtol=1e-5, atol=1e-6)
        if not with_pred:
            np.testing.assert_allclose(motion[i, :, 1].numpy(), want["motion"][:, 1], rtol=1e-5, atol=1e-6)
        else:
            L = int(length[i])
            imgs = items[("test", name)]["recording_utils"]["original_imgname"]
            est = np.stack([np.concatenate([preds[im]["smpl_parameters"]["global_orient"], preds[im]["smpl_parameters"]["body_pose"]]) for im in imgs])
            np.testing.assert_allclose(motion[i, :L, 1].numpy(), (est - mean[0, :72]) / std[0, :72], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(beta[i, 1, :L].numpy(), np.stack([preds[im]["smpl_parameters"]["betas"] for im in imgs]), rtol=1e-5)
    assert len(s.item(0)) == 9
