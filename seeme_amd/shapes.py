"""Parameter names and shapes of the modules on the path, independent of torch.

Used by the weight recipe, by tests and by ``bench.py`` to build random-init weights of the
reference architecture without importing the reference.
"""
from __future__ import annotations

from typing import Dict, Tuple

D = 256
Shapes = Dict[str, Tuple[int, ...]]


def _ln(pre: str, out: Shapes, d: int = D):
    out[pre + "weight"] = (d,)
    out[pre + "bias"] = (d,)


def _lin(pre: str, out: Shapes, n: int, k: int, bias: bool = True):
    out[pre + "weight"] = (n, k)
    if bias:
        out[pre + "bias"] = (n,)


def _mha(pre: str, out: Shapes):
    out[pre + "in_proj_weight"] = (3 * D, D)
    out[pre + "in_proj_bias"] = (3 * D,)
    _lin(pre + "out_proj.", out, D, D)


_BLOCKS = ("input_blocks.0.", "input_blocks.1.", "middle_block.", "output_blocks.0.", "output_blocks.1.")


def vae_shapes(nfeats: int, ff: int = 128) -> Shapes:
    """MldVae (mld_vae.py:35-116): 5 layers / 1 head / ff 128 hard-coded (:51-53)."""
    s: Shapes = {"global_motion_token": (2, D),
                 "query_pos_encoder.pe": (500, 1, D), "query_pos_decoder.pe": (500, 1, D)}
    for stack, dec in (("encoder.", False), ("decoder.", True)):
        for b in _BLOCKS:
            p = stack + b
            _mha(p + "self_attn.", s)
            if dec:
                _mha(p + "multihead_attn.", s)
            _lin(p + "linear1.", s, ff, D)
            _lin(p + "linear2.", s, D, ff)
            _ln(p + "norm1.", s)
            _ln(p + "norm2.", s)
            if dec:
                _ln(p + "norm3.", s)
        for i in (0, 1):
            _lin(f"{stack}linear_blocks.{i}.", s, D, 2 * D)
        _ln(stack + "norm.", s)
    _lin("skel_embedding.", s, D, nfeats)
    _lin("final_layer.", s, nfeats, D)
    return s


def _styl(pre: str, out: Shapes):
    _lin(pre + "emb_layers.1.", out, 2 * D, D)
    _ln(pre + "norm.", out)
    _lin(pre + "out_layers.2.", out, D, D)


def denoiser_shapes(ff: int = 128, sa_ff: int = 1024) -> Shapes:
    """MldDenoiser (mld_denoiser.py:20-149) with MD_TRANS layers (mdiff_transformer.py:257-284)."""
    s: Shapes = {"query_pos.pe": (500, 1, D), "mem_pos.pe": (500, 1, D)}
    _lin("time_embedding.linear_1.", s, D, D)
    _lin("time_embedding.linear_2.", s, D, D)
    for b in _BLOCKS:
        p = "encoder." + b
        _mha(p + "sa_block.self_attn.", s)
        _lin(p + "sa_block.linear1.", s, sa_ff, D)
        _lin(p + "sa_block.linear2.", s, D, sa_ff)
        _ln(p + "sa_block.norm1.", s)
        _ln(p + "sa_block.norm2.", s)
        _ln(p + "ca_block.norm.", s)
        _ln(p + "ca_block.text_norm.", s)
        for n in ("query.", "key.", "value."):
            _lin(p + "ca_block." + n, s, D, D)
        _styl(p + "ca_block.proj_out.", s)
        _lin(p + "ffn.linear1.", s, ff, D)
        _lin(p + "ffn.linear2.", s, D, ff)
        _styl(p + "ffn.proj_out.", s)
    for i in (0, 1):
        _lin(f"encoder.linear_blocks.{i}.", s, D, 2 * D)
    _ln("encoder.norm.", s)
    return s


def pointnet_shapes(out_dim: int = 512, hidden: int = 256) -> Shapes:
    """ResnetPointnet (EgoHMR/models/respointnet.py:13-27)."""
    s: Shapes = {}
    _lin("fc_pos_0.", s, 2 * hidden, 3)
    for i in range(4):
        _lin(f"block_{i}.fc_0.", s, hidden, 2 * hidden)
        _lin(f"block_{i}.fc_1.", s, hidden, hidden)
        _lin(f"block_{i}.shortcut.", s, hidden, 2 * hidden, bias=False)
    _lin("fc_c.", s, out_dim, hidden)
    return s
