"""Deterministic weight recipe ("weights by recipe, I/O by fixture").

No checkpoint, dataset or SMPL file exists offline, so every
parity fixture and every benchmark uses weights generated here from a seed.
The recipe only looks at parameter *names and shapes*, which are identical in
the reference modules and in this package, so the same
state-dict can be loaded into the reference module (fixture generation, in the
build container only) and into ours (everywhere).

The values are chosen so that every branch of the path is numerically alive
(non-zero biases, non-unit LayerNorm gains, non-zero ``zero_module`` layers;
the reference re-initialises those with xavier anyway, see
mld/models/operator/cross_attention.py:36-39).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np


def _key_seed(seed: int, key: str) -> int:
    # Stable per-key stream: independent of dict order and of other keys.
    return (int(seed) * 1000003 + zlib.crc32(key.encode("utf-8"))) & 0x7FFFFFFF


def recipe_tensor(key: str, shape: Tuple[int, ...], seed: int = 1234) -> np.ndarray:
    """float32 array for one parameter, determined by (seed, key, shape)."""
    rng = np.random.Generator(np.random.PCG64(_key_seed(seed, key)))
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "pe":  # PositionEmbeddingLearned1D: uniform(0,1) (position_encoding.py:150-151)
        out = rng.random(shape)
    elif key.endswith("global_motion_token"):
        out = rng.standard_normal(shape)
    elif len(shape) >= 2:
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        # xavier-uniform-like scale (cross_attention.py:36-39), gaussian draw
        std = np.sqrt(2.0 / (fan_in + fan_out))
        out = rng.standard_normal(shape) * std
    elif "norm" in key and leaf == "weight":
        out = 1.0 + 0.1 * rng.standard_normal(shape)
    else:  # biases and LayerNorm biases
        out = 0.05 * rng.standard_normal(shape)
    return np.ascontiguousarray(out, dtype=np.float32)


def recipe_state_dict(shapes: Mapping[str, Iterable[int]], seed: int = 1234) -> Dict[str, np.ndarray]:
    """Fill every entry of ``{name: shape}`` by the recipe."""
    return {k: recipe_tensor(k, tuple(v), seed) for k, v in sorted(shapes.items())}


def load_recipe_(module, seed: int = 1234):
    """In-place: overwrite every entry of ``module.state_dict()`` (torch) by the recipe."""
    import torch

    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            new[k] = v
            continue
        new[k] = torch.from_numpy(recipe_tensor(k, tuple(v.shape), seed)).to(dtype=v.dtype)
    module.load_state_dict(new, strict=True)
    return module
