"""AdamW over ONE flat parameter buffer.

``torch.optim.AdamW(model.parameters(), fused=True)`` - what the reference's training loop builds
(alignn/train.py:175-186 through ``setup_optimizer``) - walks ~100 parameter tensors in five ``multi_tensor_apply``
launches of ~44 us each at the benchmark model (16 MB of parameters): 0.22 ms of a 17.5 ms step for 112 MB of traffic
that a single elementwise pass moves in ~25 us.  ``FlatAdamW`` re-homes the parameters as views of one contiguous
fp32 buffer (names, shapes and ``state_dict`` are untouched; the four gate / update weights of every
``EdgeGatedGraphConv`` stay adjacent, in the order its fused node projection wants them, and the module adopts the
flat slice as that fused buffer), gathers the gradients with one batched copy and runs torch's own fused AdamW kernel
on ONE tensor.  Same arithmetic per element, same step count: the parameters after a step are bit-identical to the
per-tensor optimizer's (tests/test_gpu_round2.py).

Parameters that never receive a gradient (``grad is None`` after backward: e.g. the edge norm of the last convolution,
whose output is dead) are left out of the buffer - torch's AdamW skips them, so they must not see weight decay either.
The layout is therefore fixed at the first ``step()``.  Learning-rate schedules: assign ``opt.lr`` (or use
``opt.param_groups`` once the first step has run).
"""
from __future__ import annotations

from typing import List

import torch


def _fused_groups(module: torch.nn.Module):
    """[(owner, [weights in fused order], [biases in fused order])] for every module that keeps a fused weight buffer."""
    out = []
    for m in module.modules():
        f = getattr(m, "_fused_parameter_groups", None)
        if f is not None:
            out.append((m,) + tuple(f()))
    return out


class FlatAdamW:
    def __init__(self, module: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        self.module = module
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.flat = None  # nn.Parameter over the flat buffer (its .grad is the flat gradient)
        self.live: List[torch.nn.Parameter] = []
        self.inner = None

    # ---- the pieces of torch.optim.Optimizer the training loops use
    @property
    def param_groups(self):
        return self.inner.param_groups if self.inner is not None else [dict(self.defaults, params=[])]

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        self.defaults["lr"] = value
        if self.inner is not None:
            self.inner.param_groups[0]["lr"] = value

    def zero_grad(self, set_to_none: bool = True):
        for p in self.module.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _build(self):
        params = [p for p in self.module.parameters() if p.requires_grad]
        live_ids = {id(p) for p in params if p.grad is not None}
        if not live_ids:
            raise RuntimeError("FlatAdamW.step() before any backward(): no parameter has a gradient")
        order, seen, adopt = [], set(), []
        for owner, ws, bs in _fused_groups(self.module):  # fused groups first, each contiguous and in its own order
            if all(id(p) in live_ids for p in list(ws) + list(bs)):
                adopt.append((owner, ws, bs))
                for p in list(ws) + list(bs):
                    order.append(p)
                    seen.add(id(p))
        order += [p for p in params if id(p) in live_ids and id(p) not in seen]
        p0 = order[0]
        if any(p.dtype != p0.dtype or p.device != p0.device for p in order):
            raise ValueError("FlatAdamW needs all parameters on one device in one dtype")
        total = sum(p.numel() for p in order)
        flat = torch.empty(total, dtype=p0.dtype, device=p0.device)
        off = 0
        with torch.no_grad():
            for p in order:
                view = flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                off += p.numel()
        for owner, ws, bs in adopt:
            owner._adopt_fused_buffers()
        self.live = order
        self.flat = torch.nn.Parameter(flat, requires_grad=True)
        self.flat.grad = torch.zeros_like(flat)
        self.inner = torch.optim.AdamW([self.flat], fused=flat.is_cuda, **self.defaults)

    @torch.no_grad()
    def step(self):
        if self.inner is None:
            self._build()
        grads = []
        for p in self.live:
            if p.grad is None:
                raise RuntimeError("a parameter that had a gradient at the first step() has none now: FlatAdamW's layout is "
                                   "fixed at the first step")
            grads.append(p.grad.reshape(-1))
        torch.cat(grads, out=self.flat.grad)  # one batched copy
        self.inner.step()

    def state_dict(self):
        return {"inner": self.inner.state_dict() if self.inner is not None else None, "defaults": dict(self.defaults)}
