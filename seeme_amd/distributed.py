"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for CPU tests).  The reference gets all of this implicitly from Lightning's
``strategy="ddp"`` (train.py:127-131); the path needs exactly two things:

* sampling: sequences are independent -> contiguous shards per rank, NO collective on the data path;
  only the few metric sums are reduced at the end (C2);
* training: one mean all-reduce of the trainable gradients per step (C1: denoiser 7.90 M + output_scene
  0.13 M parameters = 32 MB fp32) as ONE flat bucket, so RCCL sees a single large message.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torch.distributed.run)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=ws)
    return rank, ws, local


def shard_range(n: int, rank: int = None, ws: int = None) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of `n` independent sequences for this rank (ranks differ by <= 1)."""
    if rank is None:
        rank, ws = world()
    base, rem = divmod(n, ws)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """Every trainable gradient of a model in ONE flat fp32 buffer whose views are the parameters' ``.grad``
    (what DDP calls ``gradient_as_bucket_view``): autograd accumulates into the views in place, the hand-written
    denoiser backward writes its region directly (``TrainPack.bind``), and the data-parallel exchange is a single
    ``all_reduce`` of the buffer -- no concatenation before it, no copy back after it (C1: 32 MB for stage 2).

    Layout: [the chain's gradient block of ``pack`` (denoiser_train.TrainPack), if any | every other trainable
    parameter in ``parameters`` order].  ``prepare()`` before each backward re-attaches the views and zeroes the part
    autograd accumulates into; gradient accumulation over several backwards is not what this object is for.

    Overlap with the backward (what DDP's bucketed hooks give the reference, train.py:127-131): ``early_span`` is the part of the
    chain block that is FINAL as soon as the chain's weight-gradient kernels have run -- every matrix except in_proj (whose K | V
    rows also collect the condition / time-table contributions later in the backward) -- 18 of the 31.6 MB.  ``allreduce_early()``
    (called by TrainPack.reduce_into_grads) starts its exchange asynchronously right there, on the collective's own stream, while the
    rest of the backward runs; ``allreduce()`` exchanges what is left and waits for both."""

    def __init__(self, params: Iterable[torch.nn.Parameter], pack=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev = self.params[0].device
        self.pack = pack
        in_pack = {}
        n_pack = 0
        if pack is not None:
            n_pack = pack.gflat_numel
            in_pack = {id(p): i for i, p in enumerate(pack.params)}
        off = n_pack
        spans = []
        for p in self.params:
            i = in_pack.get(id(p))
            if i is not None and i != 0:          # position 0 is query_pos.pe: a whole [500,1,256] tensor, kept outside
                continue
            spans.append((p, off))
            off += p.numel()
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.n_pack = n_pack
        self.views = {}
        for p, o in spans:
            self.views[id(p)] = self.flat[o:o + p.numel()].view_as(p)
        if pack is not None:
            pack.bind(self.flat[:n_pack], self.views.get(id(pack.params[0])))
            for i, p in enumerate(pack.params):
                if i != 0 and p.requires_grad:
                    self.views[id(p)] = pack.grad_views[i]
        self._pack_key = None if pack is None else id(pack)
        self.early_span = None
        self._early = None
        self.overlap = os.environ.get("SEEME_GRAD_OVERLAP", "1") != "0"
        if pack is not None and getattr(pack, "early_span", None) is not None:
            self.early_span = tuple(int(v) for v in pack.early_span)
            pack.bucket = self

    def matches(self, pack) -> bool:
        return self._pack_key == (None if pack is None else id(pack))

    def prepare(self):
        """Before backward: ``.grad`` = the views; zero what autograd adds into (the chain block is overwritten)."""
        assert self._early is None, "GradBucket: the previous step's early exchange was never completed (allreduce() not called)"
        self.flat[self.n_pack:].zero_()
        for p in self.params:
            v = self.views[id(p)]
            if p.grad is not v:
                p.grad = v
        if self.pack is not None:
            self.pack._attached = True

    def allreduce_early(self, average: bool = True) -> int:
        """Start the mean all-reduce of ``early_span`` now, asynchronously (it is final; see the class docstring).  Returns the
        number of elements in flight (0: single process, no early span, or overlap switched off)."""
        rank, ws = world()
        if ws == 1 or self.early_span is None or not self.overlap or self._early is not None:
            return 0
        if self.flat.is_cuda and torch.cuda.is_current_stream_capturing():
            return 0
        lo, hi = self.early_span
        view = self.flat[lo:hi]
        avg_op = average and dist.get_backend() == "nccl"
        work = dist.all_reduce(view, op=dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM, async_op=True)
        self._early = (work, view, average and not avg_op)
        return int(view.numel())

    def allreduce(self, average: bool = True) -> int:
        rank, ws = world()
        if ws > 1:
            n = int(self.flat.numel())
            spans = [(0, n)] if self._early is None else [(0, self.early_span[0]), (self.early_span[1], n)]
            avg_op = average and dist.get_backend() == "nccl"
            for a, b in spans:
                if b > a:
                    dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM)
                    if average and not avg_op:
                        self.flat[a:b].div_(ws)
            if self._early is not None:
                work, view, div = self._early
                work.wait()                               # (NCCL: the current stream waits for the collective's stream; the host does not)
                if div:
                    view.div_(ws)
                self._early = None
        return int(self.flat.numel())


def allreduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True) -> int:
    """Mean all-reduce of every .grad as one flat bucket; parameters without a gradient (e.g. mem_pos.pe,
    which trans_enc never touches) contribute zeros so that all ranks agree on the layout.
    Returns the number of elements reduced.  (Generic form for gradients that live in separate tensors: it has to
    gather them and scatter the result.  The training loop keeps them in a GradBucket instead and reduces in place.)"""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    rank, ws = world()
    if ws == 1:
        return sum(p.numel() for p in params)
    dev, dt = params[0].device, torch.float32
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= ws
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return int(flat.numel())


def reduce_sums(t: torch.Tensor) -> torch.Tensor:
    """Sum a small tensor of metric / loss accumulators over ranks (dist_reduce_fx="sum" in the reference's
    torchmetrics states, metrics/compute.py:106-178)."""
    if not is_dist():
        return t
    backend = dist.get_backend()
    x = t.clone().to("cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return x.to(t.device)


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank `src`'s weights (DDP does this at construction)."""
    if not is_dist():
        return
    for p in module.parameters():
        dist.broadcast(p.data, src)
    for b in module.buffers():
        dist.broadcast(b.data, src)
