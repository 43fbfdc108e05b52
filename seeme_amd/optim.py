"""One-launch AdamW step for the trainable tensors of ``MLD`` (``seeme_adamw_step``, csrc/misc_kernels.hip).

The reference trains with ``torch.optim.AdamW`` built in ``BaseModel.configure_optimizers`` (mld/models/modeltype/base.py)
and stepped by Lightning after ``training_step`` (train.py:127-149).  The ``torch.optim.AdamW`` object is kept -- it owns
the hyper-parameters the LR scheduler edits and the ``state_dict`` layout that goes into the checkpoints -- and only its
``step()`` is replaced: same arithmetic (decoupled weight decay, bias corrections, ``amsgrad`` off) on the same state
tensors (``exp_avg``, ``exp_avg_sq``, ``step``), in one kernel over all tensors instead of 27 multi-tensor launches.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib as L

_CHUNK = 16384      # elements per workgroup


class FusedAdamWStep:
    def __init__(self, optimizer: torch.optim.AdamW):
        for g in optimizer.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                raise NotImplementedError("FusedAdamWStep: amsgrad / maximize / capturable / differentiable AdamW")
        self.opt = optimizer
        self._key = None
        self._shared = {}
        self._dev_scalars, self._dev_lr = {}, {}

    def _tables(self, params: List[torch.nn.Parameter]):
        """Static device tables (chunks, parameter / moment pointers); rebuilt when the tensors behind them change."""
        key = tuple((p.data_ptr(), self.opt.state[p]["exp_avg"].data_ptr(), self.opt.state[p]["exp_avg_sq"].data_ptr()) for p in params)
        if key == self._key:
            return
        dev = params[0].device
        chunks = []
        for t, p in enumerate(params):
            n = p.numel()
            for off in range(0, n, _CHUNK):
                chunks.append((t, min(_CHUNK, n - off), off))
        assert all(off < 2 ** 31 for _, _, off in chunks)
        # {int tensor, int count, int64 offset}, little-endian: the offset's high word stays zero
        ch = torch.tensor([[t, cnt, off, 0] for t, cnt, off in chunks], dtype=torch.int32)
        self._chunks = ch.to(dev)
        self._n_chunks = len(chunks)
        ptr = lambda xs: torch.tensor(xs, dtype=torch.int64).to(dev)
        self._p = ptr([k[0] for k in key])
        self._m = ptr([k[1] for k in key])
        self._v = ptr([k[2] for k in key])
        self._key = key

    def _grad_pointers(self, grads, dev):
        """Table of gradient base pointers.  With the gradients in a GradBucket the addresses never change and the table
        is uploaded once; otherwise autograd re-allocates most of them every step: pinned host buffers, two in rotation,
        copied asynchronously -- a pageable copy would stall the host on the whole backward."""
        n = len(grads)
        key = tuple(g.data_ptr() for g in grads)
        if getattr(self, "_gp_key", None) == key:
            return self._gp_dev
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedAdamWStep: gradient addresses changed inside a graph capture (run warm-up steps first)")
        if getattr(self, "_gp_host", None) is None or self._gp_host[0].numel() != n:
            self._gp_host = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(2)]
            self._gp_ev = [None, None]
            self._gp_dev = torch.empty(n, dtype=torch.int64, device=dev)
            self._gp_i = 0
        i = self._gp_i
        self._gp_i ^= 1
        if self._gp_ev[i] is not None:
            self._gp_ev[i].synchronize()                  # the copy that last read this host buffer has run
        self._gp_host[i].copy_(torch.tensor(key, dtype=torch.int64))
        self._gp_dev.copy_(self._gp_host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._gp_ev[i] = ev
        self._gp_key = key
        return self._gp_dev

    def _device_scalars(self, group, shared, dev):
        """{step, lr} on the device for the graph-capturable launch; the host copies stay authoritative for checkpoints."""
        sc = self._dev_scalars.get(id(group))
        if sc is None:
            sc = torch.tensor([float(shared), float(group["lr"])], dtype=torch.float32, device=dev)
            self._dev_scalars[id(group)] = sc
            self._dev_lr[id(group)] = float(group["lr"])
        elif self._dev_lr[id(group)] != float(group["lr"]) and not torch.cuda.is_current_stream_capturing():
            sc[1].fill_(float(group["lr"]))               # the LR scheduler edited the group
            self._dev_lr[id(group)] = float(group["lr"])
        return sc

    def _refresh_device_step(self, group, shared):
        """The device-side {step, lr} tensor is never dropped once it exists: a hipGraph captured by MLD.capture_training_step has
        its ADDRESS baked in, and a freed tensor's memory may be handed to someone else while replays still add to it.  After a
        host-side step (or load_state_dict) the count is rewritten in place instead."""
        sc = self._dev_scalars.get(id(group))
        if sc is not None and not torch.cuda.is_current_stream_capturing():
            sc[0].fill_(float(shared))

    def sync_lr(self):
        """Push a learning rate the LR scheduler edited to the device-side scalars (before replaying a captured step)."""
        for group in self.opt.param_groups:
            k = id(group)
            if k in self._dev_scalars and self._dev_lr[k] != float(group["lr"]):
                self._dev_scalars[k][1].fill_(float(group["lr"]))
                self._dev_lr[k] = float(group["lr"])

    def note_replay(self):
        """A captured step was replayed: advance the host-side step counts (the device-side count advanced in the graph)."""
        for group in self.opt.param_groups:
            shared = self._shared.get(id(group))
            if shared is not None:
                shared.add_(1.0)

    @torch.no_grad()
    def step(self, device_step: bool = False):
        """device_step: step count and learning rate are read from device memory (seeme_adamw_step_dev), which makes the
        launch capturable in a hipGraph; inside a capture the host-side step count is left to note_replay()."""
        capturing = torch.cuda.is_current_stream_capturing()
        for group in self.opt.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                L.require_cuda(p, "AdamW parameter")
                if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32:
                    raise NotImplementedError("FusedAdamWStep: contiguous fp32 parameters and gradients")
                st = self.opt.state[p]
                if len(st) == 0:                          # same lazy state as torch.optim.AdamW._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # one shared step tensor per group: 204 scalar increments per step cost more host time than the kernel
            # (a tensor of this object's own: state handed over by load_state_dict may alias another optimiser's)
            shared = self._shared.get(id(group))
            if shared is None or any(self.opt.state[p]["step"] is not shared for p in params):
                steps = {float(self.opt.state[p]["step"]) for p in params}
                if len(steps) != 1:
                    raise NotImplementedError("FusedAdamWStep: parameters of a group must share their step count")
                shared = torch.tensor(steps.pop(), dtype=torch.float32)
                self._shared[id(group)] = shared
                for p in params:
                    self.opt.state[p]["step"] = shared
                self._refresh_device_step(group, shared)
            self._tables(params)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
            gp = self._grad_pointers(grads, params[0].device)
            b1, b2 = group["betas"]
            if device_step:
                sc = self._device_scalars(group, shared, params[0].device)
                sc[0:1].add_(1.0)
                if not capturing:
                    shared.add_(1.0)
                L.check(L.lib().seeme_adamw_step_dev(self._chunks.data_ptr(), self._n_chunks, self._p.data_ptr(), gp.data_ptr(),
                                                     self._m.data_ptr(), self._v.data_ptr(), sc.data_ptr(), float(b1), float(b2),
                                                     float(group["eps"]), float(group["weight_decay"]), L.current_stream()),
                        "seeme_adamw_step_dev")
            else:
                step = float(shared) + 1.0
                shared.fill_(step)
                self._refresh_device_step(group, shared)  # the device-side count (if any) follows the host's
                L.check(L.lib().seeme_adamw_step(self._chunks.data_ptr(), self._n_chunks, self._p.data_ptr(), gp.data_ptr(),
                                                 self._m.data_ptr(), self._v.data_ptr(), float(group["lr"]), float(b1), float(b2),
                                                 float(group["eps"]), float(group["weight_decay"]), float(step),
                                                 L.current_stream()), "seeme_adamw_step")
            self._keep = (grads, gp)                      # alive until the next step (the launch is asynchronous)
        self.opt._opt_called = True                       # what the LR scheduler's wrapper of optimizer.step() records
