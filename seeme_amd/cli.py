"""Lightning-free equivalents of the reference's ``train.py`` / ``test.py`` for the accelerated path.

What the reference delegates to ``pl.Trainer`` (train.py:127-149, test.py:86-134) is done here directly:
one process per GPU (``torch.distributed`` / RCCL, launched by ``torchrun``), parameter broadcast, one
flat-bucket gradient all-reduce per step, checkpoints in Lightning's ``{"state_dict": ...}`` layout under
``<FOLDER>/<model_type>/<NAME>/checkpoints/epoch=<n>.ckpt`` (train.py:114-123), resume from the newest
``epoch=*.ckpt`` of ``TRAIN.RESUME`` (train.py:26-53), strict load of the ``vae.*`` sub-dict for stage 2
(train.py:155-167), strict full load for testing (test.py:111-113), metrics summed over ranks and written
to ``metrics_<time>.json`` (test.py:136-152).

The argument surface is the reference's (mld/config.py:35-65: --cfg --cfg_assets --batch_size --device
--nodebug --dir) plus loop bounds for the synthetic data module: the EgoBody / GIMO datasets are
licence-gated: ``--data_root`` points ``seeme_amd.data.EgoDataModule`` at a directory in the reference's on-disk layout
(split resident in HBM); without it batches come from ``SyntheticEgoDataModule``.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import re
import time
from typing import Dict, List, Optional

import torch

from . import distributed as D
from .config import parse_config

CKPT_RE = re.compile(r"^epoch=(\d+)\.ckpt$")


# ----------------------------------------------------------------------------- arguments / folders
def build_parser(phase: str) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog=f"seeme_amd {phase}")
    g = p.add_argument_group("reference options (mld/config.py:35-65)")
    g.add_argument("--cfg", type=str, default="./configs/config_mld_egobody.yaml", help="config file")
    g.add_argument("--cfg_assets", type=str, default=None, help="config file for asset paths")
    g.add_argument("--batch_size", type=int, help="batch size per GPU")
    g.add_argument("--device", type=int, nargs="+", help="accepted for compatibility; ranks come from torchrun")
    g.add_argument("--nodebug", action="store_true", help="debug or not")
    g.add_argument("--dir", type=str, help="evaluate existing npys (not supported on this path)")
    s = p.add_argument_group("loop bounds of the synthetic data module")
    s.add_argument("--epochs", type=int, default=None, help="override TRAIN.END_EPOCH")
    s.add_argument("--iters_per_epoch", type=int, default=8, help="batches per epoch and rank")
    s.add_argument("--test_batches", type=int, default=4, help="batches per replication and rank")
    s.add_argument("--data_root", type=str, default=None,
                   help="directory in the reference's EgoBody / GIMO on-disk layout (seeme_amd/data.py); default: synthetic batches")
    s.add_argument("--scene_root", type=str, default=None, help="scene tables of --data_root (default: the same directory)")
    s.add_argument("--storage", type=str, default="device", choices=["device", "pinned"], help="where --data_root's splits live")
    s.add_argument("--scene_points", type=int, default=20000)
    s.add_argument("--frames", type=int, default=196)
    s.add_argument("--folder", type=str, default=None, help="override FOLDER (experiment root)")
    s.add_argument("--checkpoint", type=str, default=None, help="override TEST.CHECKPOINTS")
    return p


def load_cfg(args, phase: str):
    cfg = parse_config(args.cfg, cfg_assets=args.cfg_assets, batch_size=args.batch_size, phase=phase)
    if phase == "train":
        cfg.DEBUG = (not args.nodebug) if args.nodebug else cfg.get("DEBUG", False)
        if cfg.DEBUG:
            cfg.NAME = "debug--" + str(cfg.get("NAME", "exp"))       # mld/config.py:190-193
    if args.folder:
        cfg.FOLDER = args.folder
    cfg.setdefault("FOLDER", "./experiments")
    cfg.setdefault("NAME", "exp")
    cfg.setdefault("TIME", time.strftime("%Y-%m-%d-%H-%M-%S"))
    cfg.FOLDER_EXP = os.path.join(cfg.FOLDER, str(cfg.model.get("model_type", "mld")), str(cfg.NAME))
    if args.dir:
        raise NotImplementedError("--dir (evaluate stored npys) is outside the accelerated path")
    return cfg


def make_logger(cfg, phase: str, rank: int) -> logging.Logger:
    log = logging.getLogger(f"seeme_amd.{phase}")
    log.setLevel(logging.INFO if rank == 0 else logging.WARNING)
    log.handlers.clear()
    fmt = logging.Formatter("%(asctime)s %(message)s")
    h = logging.StreamHandler()
    h.setFormatter(fmt)
    log.addHandler(h)
    if rank == 0:
        os.makedirs(cfg.FOLDER_EXP, exist_ok=True)
        fh = logging.FileHandler(os.path.join(cfg.FOLDER_EXP, f"log_{phase}_{cfg.TIME}.log"))
        fh.setFormatter(fmt)
        log.addHandler(fh)
    return log


# ----------------------------------------------------------------------------- checkpoints (Lightning layout)
def save_checkpoint(path: str, model, epoch: int, global_step: int) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    obj = {"epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "seeme-amd",
           "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
           "optimizer_states": [model.optimizer.state_dict()] if model.optimizer is not None else []}
    tmp = path + ".tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def read_checkpoint(path: str) -> Dict:
    """Tensor-only load: nothing in the file is executed."""
    return torch.load(path, map_location="cpu", weights_only=True)


def newest_checkpoint(resume_dir: str) -> Optional[str]:
    d = os.path.join(resume_dir, "checkpoints")
    if not os.path.isdir(d):
        raise ValueError("Resume path is not right.")                  # train.py:56
    best = None
    for fn in os.listdir(d):
        m = CKPT_RE.match(fn)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.join(d, fn))
    return best[1] if best else None


def load_pretrained_vae(model, path: str) -> int:
    """Stage 2 starts from a stage-1 checkpoint: the ``vae.*`` entries, strictly (train.py:155-167)."""
    sd = read_checkpoint(path)["state_dict"]
    sub = {k[len("vae."):]: v for k, v in sd.items() if k.split(".")[0] == "vae"}
    model.vae.load_state_dict(sub, strict=True)
    return len(sub)


# ----------------------------------------------------------------------------- model / data
def build(cfg, dev, args, datamodule=None, smpl_model=None):
    from .mld import MLD, SyntheticEgoDataModule
    from .smpl import SMPL
    nfeats = 75 if cfg.DATASET_NAME == "egobody" else (69 if cfg.DATASET_NAME == "gimo" else cfg.model.nfeats)
    if datamodule is None and getattr(args, "data_root", None):
        from .data import EgoDataModule
        datamodule = EgoDataModule(args.data_root, cfg.DATASET_NAME, tuple(cfg.model.condition), motion_length=int(cfg.MOTION_LENGTH),
                                   predict_transl=bool(cfg.TRAIN.ABLATION.PREDICT_TRANSL), device=dev, storage=args.storage,
                                   scene_root=args.scene_root, pose_estimation_task=bool(cfg.TEST.get("POSE_ESTIMATION_TASK", False)),
                                   interactee_pred=bool(cfg.TEST.get("INTERACTEE_PRED", False)),      # get_data.py:196
                                   seed=int(cfg.SEED_VALUE))
    dm = datamodule or SyntheticEgoDataModule(nfeats=nfeats, T=args.frames, n_points=args.scene_points,
                                               seed=int(cfg.SEED_VALUE), device=dev)
    if smpl_model is None and not os.path.exists(str(cfg.model.smpl_path)):
        smpl_model = SMPL.synthetic(int(cfg.SEED_VALUE))              # SMPL_NEUTRAL.pkl is licence-gated
    model = MLD(cfg, dm, smpl_model=smpl_model)
    return model, dm


def _with_scene(cfg) -> bool:
    return "scene" in cfg.model.condition


# ----------------------------------------------------------------------------- train
def train_main(argv: Optional[List[str]] = None, datamodule=None, smpl_model=None) -> Dict:
    args = build_parser("train").parse_args(argv)
    rank, ws, local = D.init_from_env()
    cfg = load_cfg(args, "train")
    log = make_logger(cfg, "train", rank)
    if not torch.cuda.is_available():
        raise SystemExit("training runs on the HIP path: an MI355X is required (no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(int(cfg.SEED_VALUE) + rank)

    model, dm = build(cfg, dev, args, datamodule, smpl_model)
    start_epoch, global_step = 0, 0
    if cfg.TRAIN.get("PRETRAINED_VAE"):
        n = load_pretrained_vae(model, cfg.TRAIN.PRETRAINED_VAE)
        log.info("Loading pretrain vae from %s (%d tensors, strict)", cfg.TRAIN.PRETRAINED_VAE, n)
    resume_ckpt = newest_checkpoint(cfg.TRAIN.RESUME) if cfg.TRAIN.get("RESUME") else None
    pre = resume_ckpt or cfg.TRAIN.get("PRETRAINED")
    ck = None
    if pre:
        ck = read_checkpoint(pre)
        sd = {k: v for k, v in ck["state_dict"].items() if k != "denoiser.sequence_pos_encoding.pe"}   # train.py:177-180
        missing, unexpected = model.load_state_dict(sd, strict=False)
        log.info("Loading pretrain model from %s (missing %d, unexpected %d)", pre, len(missing), len(unexpected))
    model = model.to(dev).train()
    D.broadcast_parameters(model)
    model.configure_optimizers()
    if resume_ckpt and ck is not None:
        start_epoch, global_step = int(ck.get("epoch", -1)) + 1, int(ck.get("global_step", 0))
        if ck.get("optimizer_states"):
            model.optimizer.load_state_dict(ck["optimizer_states"][0])
        log.info("Resuming after epoch %d (step %d)", start_epoch - 1, global_step)

    B = int(cfg.TRAIN.BATCH_SIZE)
    end_epoch = int(args.epochs if args.epochs is not None else cfg.TRAIN.END_EPOCH)
    save_every = max(1, int((cfg.get("LOGGER") or {}).get("SACE_CHECKPOINT_EPOCH", 1)))
    ckpt_dir = os.path.join(cfg.FOLDER_EXP, "checkpoints")
    log.info("stage %s, conditions %s, batch %d per GPU x %d GPU(s), epochs %d..%d", cfg.TRAIN.STAGE,
             list(cfg.model.condition), B, ws, start_epoch, end_epoch - 1)
    last = {}
    for epoch in range(start_epoch, end_epoch):
        model.losses["train"].reset()
        t0 = time.perf_counter()
        if hasattr(dm, "iterate"):      # files: one pass over the train split per epoch, a new permutation each, ranks take disjoint
            # strided shares (what the reference's DataLoader + DistributedSampler do, train.py:127-149); drop_last keeps the ranks in step
            batches = dm.iterate("train", B, shuffle=True, seed=int(cfg.SEED_VALUE), epoch=epoch, rank=rank, world=ws, drop_last=True)
        else:                           # synthetic stream: `iters_per_epoch` fresh batches
            batches = (dm.batch(B, idx=(epoch * args.iters_per_epoch + it) * ws + rank, with_scene=_with_scene(cfg))
                       for it in range(args.iters_per_epoch))
        n_it = 0
        for it, batch in enumerate(batches):
            loss = model.training_step(batch, it)
            model.optimizer_step(loss)
            global_step += 1
            n_it += 1
        if getattr(model, "sch", None) is not None:
            model.sch.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sums = model.losses["train"].compute()
        last = {"epoch": epoch, "step": global_step, "seqs_per_s": round(ws * B * n_it / dt, 1),
                **{k: round(v, 6) for k, v in sums.items()}}
        log.info("epoch %d: %s", epoch, json.dumps(last))
        if rank == 0 and ((epoch + 1) % save_every == 0 or epoch + 1 == end_epoch):
            path = os.path.join(ckpt_dir, f"epoch={epoch}.ckpt")
            save_checkpoint(path, model, epoch, global_step)
            log.info("checkpoint %s", path)
    if D.is_dist():
        torch.distributed.barrier()
    log.info("The checkpoints are stored in %s", ckpt_dir)
    log.info("Training ends!")
    return {"folder": cfg.FOLDER_EXP, "checkpoints": ckpt_dir, **last}


# ----------------------------------------------------------------------------- test
def test_main(argv: Optional[List[str]] = None, datamodule=None, smpl_model=None) -> Dict:
    args = build_parser("test").parse_args(argv)
    rank, ws, local = D.init_from_env()
    cfg = load_cfg(args, "test")
    log = make_logger(cfg, "test", rank)
    if not torch.cuda.is_available():
        raise SystemExit("testing runs on the HIP path: an MI355X is required (no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(int(cfg.SEED_VALUE) + rank)
    model, dm = build(cfg, dev, args, datamodule, smpl_model)
    ckpt = args.checkpoint or cfg.TEST.get("CHECKPOINTS")
    if not ckpt:
        raise ValueError("TEST.CHECKPOINTS (or --checkpoint) is required")
    log.info("Loading checkpoints from %s", ckpt)
    model.load_state_dict(read_checkpoint(ckpt)["state_dict"])        # strict, test.py:111-113
    model = model.to(dev).eval()
    B = int(cfg.TEST.BATCH_SIZE if args.batch_size is None else args.batch_size)
    all_metrics: Dict[str, List[float]] = {}
    for rep in range(int(cfg.TEST.REPLICATION_TIMES)):
        model.EgoMetric.reset()
        t0 = time.perf_counter()
        with torch.no_grad():
            if hasattr(dm, "iterate"):  # files: ONE pass over the test split, every sequence exactly once over the ranks (test.py:115-133)
                batches = dm.iterate("test", B, rank=rank, world=ws)
            else:
                batches = (dm.batch(B, idx=10_000_000 + it * ws + rank, with_scene=_with_scene(cfg), split="test")
                           for it in range(args.test_batches))
            n_seq = 0
            for it, batch in enumerate(batches):
                model.test_step(batch, it)
                n_seq += int(batch[0].shape[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sums = D.reduce_sums(model.EgoMetric.sums().to(dev)).cpu()
        metrics = model.EgoMetric.compute(sums)
        metrics["seqs_per_s"] = ws * n_seq / dt
        log.info("Replication %d: %s", rep, json.dumps({k: round(v, 4) for k, v in metrics.items()}))
        for k, v in metrics.items():
            all_metrics.setdefault(k, []).append(float(v))
    out = {}
    for k, v in all_metrics.items():
        t = torch.tensor(v, dtype=torch.float64)
        out[f"Metrics/{k}/mean"] = float(t.mean())
        out[f"Metrics/{k}/min"], out[f"Metrics/{k}/max"] = float(t.min()), float(t.max())
        out[f"Metrics/{k}/conf_interval"] = float(1.96 * t.std(unbiased=False) / max(len(v), 1) ** 0.5)
        out[f"Metrics/{k}"] = v
    if rank == 0:
        os.makedirs(cfg.FOLDER_EXP, exist_ok=True)
        metric_file = os.path.join(cfg.FOLDER_EXP, f"metrics_{cfg.TIME}.json")
        with open(metric_file, "w", encoding="utf-8") as f:
            json.dump(out, f, indent=4)
        log.info("Testing done, the metrics are saved to %s", metric_file)
        out["file"] = metric_file
    return out
