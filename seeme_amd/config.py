"""Config loading with the reference's surface (mld/config.py:8-32, 152-156), without OmegaConf.

* ``parse_config(cfg, cfg_assets, configs_dir)`` reproduces the merge order
  base.yaml <- experiment yaml <- every ``configs/<model.target>/*.yaml`` merged into ``cfg.model``
  <- assets.yaml, then resolves ``${a.b}`` interpolations (whole-node references such as
  ``ablation: ${TRAIN.ABLATION}`` in configs/modules/denoiser.yaml:22).
* ``instantiate_from_config(node)`` = ``import_module(target).cls(**params)``.  ``target`` strings of
  the reference's hot-path classes are redirected to this package (TARGET_MAP), so the reference's
  YAML files work UNCHANGED; writing ``target: seeme_amd.mld_vae.MldVae`` explicitly works too.
"""
from __future__ import annotations

import copy
import importlib
import os
import re
from typing import Any, Optional

import yaml

# reference class -> drop-in in this package
TARGET_MAP = {
    "mld.models.architectures.mld_vae.MldVae": "seeme_amd.mld_vae.MldVae",
    "mld.models.architectures.mld_denoiser.MldDenoiser": "seeme_amd.mld_denoiser.MldDenoiser",
    "diffusers.DDIMScheduler": "seeme_amd.schedulers.DDIMScheduler",
    "diffusers.DDPMScheduler": "seeme_amd.schedulers.DDPMScheduler",
    "mld.models.modeltype.mld.MLD": "seeme_amd.mld.MLD",
    "EgoHMR.models.respointnet.ResnetPointnet": "seeme_amd.respointnet.ResnetPointnet",
    "smplx.SMPL": "seeme_amd.smpl.SMPL",
}

_FLOAT_RE = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)$")   # PyYAML reads 1e-4 as a string
_INTERP_RE = re.compile(r"\$\{([^}]+)\}")


class Config(dict):
    """dict with attribute access, the slice of OmegaConf's DictConfig the path uses."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, str) and _FLOAT_RE.match(x):
        return float(x)
    return x


def load_yaml(path: str) -> Config:
    with open(path, "r") as f:
        return _wrap(yaml.safe_load(f) or {})


def merge(dst: Config, src: dict) -> Config:
    """Deep merge ``src`` into ``dst`` (OmegaConf.merge semantics for dict nodes; lists replace)."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _select(root: Config, dotted: str):
    node: Any = root
    for part in dotted.split("."):
        node = node[part]
    return node


_ROOT = object()


def resolve(root: Config, node: Any = _ROOT, _depth: int = 0):
    """Resolve ``${a.b.c}`` references against ``root`` (in place for containers)."""
    if _depth > 20:
        raise ValueError("interpolation cycle")
    node = root if node is _ROOT else node
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = resolve(root, node[k], _depth)
        return node
    if isinstance(node, list):
        return [resolve(root, v, _depth) for v in node]
    if isinstance(node, str):
        m = _INTERP_RE.fullmatch(node.strip())
        # OmegaConf resolves lazily (on access); references to keys that do not exist in the merged tree
        # (unused evaluator modules of the reference) are therefore left as they are.
        try:
            if m:  # whole-node reference: may be a dict / list / scalar
                return resolve(root, copy.deepcopy(_select(root, m.group(1))), _depth + 1)
            if "${" in node:
                return _INTERP_RE.sub(lambda mm: str(resolve(root, _select(root, mm.group(1)), _depth + 1)), node)
        except (KeyError, TypeError):
            return node
    return node


def parse_config(cfg: str, cfg_assets: Optional[str] = None, configs_dir: Optional[str] = None,
                 batch_size: Optional[int] = None, device=None, phase: str = "train") -> Config:
    """base <- experiment <- modules/* (into .model) <- assets, interpolations resolved (config.py:152-156)."""
    configs_dir = configs_dir or os.path.dirname(os.path.abspath(cfg))
    out = load_yaml(os.path.join(configs_dir, "base.yaml"))
    merge(out, load_yaml(cfg))
    mod_dir = os.path.join(configs_dir, out.model.get("target", "modules"))
    if os.path.isdir(mod_dir):
        for fn in sorted(os.listdir(mod_dir)):
            if fn.endswith(".yaml"):
                merge(out.model, load_yaml(os.path.join(mod_dir, fn)))
    if cfg_assets is None:
        cand = os.path.join(configs_dir, "assets.yaml")
        cfg_assets = cand if os.path.exists(cand) else None
    if cfg_assets:
        merge(out, load_yaml(cfg_assets))
    if batch_size:
        out.TRAIN.BATCH_SIZE = batch_size
    if device is not None:
        out.DEVICE = device
    if phase == "test":                      # config.py:167-170
        out.DEBUG = False
        out.DEVICE = [0]
    return resolve(out)


def get_obj_from_str(string: str):
    string = TARGET_MAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))
