"""AdamW over flat parameter buffers - a ``torch.optim.Optimizer`` the reference's training loop can use as it is.

``torch.optim.AdamW(group_decay(model), ...)`` - what the reference builds (alignn/train.py:209-210 through
``alignn/utils.py:77-108``: two parameter groups, no weight decay on names containing "bias" / "bn" / "norm") - walks
~100 parameter tensors in five ``multi_tensor_apply`` launches of ~44 us each at the benchmark model (16 MB of
parameters): 0.22 ms of a 17.5 ms step for 112 MB of traffic that a single elementwise pass moves in ~25 us.
``FlatAdamW`` re-homes the parameters of every group as views of ONE contiguous fp32 buffer per group (names, shapes
and ``state_dict`` are untouched; the four gate / update weights - and biases - of every ``EdgeGatedGraphConv`` stay
adjacent, in the order its fused node projection wants them, and the module adopts the flat slice as that fused buffer),
gathers the gradients with one batched copy per group and runs torch's own fused AdamW kernel on one tensor per group
(the groups are ranges of ONE buffer, so a data-parallel run all-reduces everything with one collective:
``average_gradients=True``).
Same arithmetic per element, same step count: the parameters after a step are bit-identical to the per-tensor
optimizer's (tests/test_optim.py, tests/test_gpu_round2.py).

It IS a ``torch.optim.Optimizer``: ``param_groups`` are real and persistent from construction (learning-rate schedulers
- the reference's default ``OneCycleLR``, alignn/train.py:217-226 - read and write ``lr`` / ``betas`` there, and every
``step()`` copies those hyper-parameters onto the flat groups), ``state_dict`` / ``load_state_dict`` round-trip the
moments and the layout, ``zero_grad`` is the base class's.

Parameters that never receive a gradient (``grad is None`` after backward: e.g. the edge norm of the last convolution,
whose output is dead) are left out of the buffers - torch's AdamW skips them, so they must not see weight decay either.
The layout is therefore fixed at the first ``step()`` (or by ``load_state_dict``).  Every ``step()`` checks that each
parameter still IS its slice of the flat buffer; something that re-homed one in between (``.to()``,
``load_state_dict(assign=True)``, a module re-fusing its weights) is undone - value kept, storage moved back - instead
of silently training a buffer nobody reads.
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional

import torch

_HYPER = ("lr", "betas", "eps", "weight_decay")


def group_decay(model: torch.nn.Module):
    """The reference's parameter groups (alignn/utils.py:77-92): no weight decay on parameters whose NAME contains
    "bias", "bn" or "norm" (note: the BatchNorm of an ``MLPLayer`` is called ``layer.1`` and is therefore decayed -
    reproduced as it is)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        (no_decay if ("bias" in name or "bn" in name or "norm" in name) else decay).append(p)
    return [{"params": decay}, {"params": no_decay, "weight_decay": 0}]


def _fused_groups(module: torch.nn.Module):
    """[(owner, [weights in fused order], [biases in fused order])] for every module that keeps a fused weight buffer."""
    out = []
    for m in module.modules():
        f = getattr(m, "_fused_parameter_groups", None)
        if f is not None:
            out.append((m,) + tuple(f()))
    return out


def _gradient_runs(module: torch.nn.Module):
    """Parameter lists a module wants adjacent (in that order) because its backward writes their gradients as one block."""
    out = []
    for m in module.modules():
        f = getattr(m, "_gradient_runs", None)
        if f is not None:
            out += [list(r) for r in f()]
    return out


LAYOUT_VERSION = 2  # order of the parameters inside a group's flat range (2: + gradient runs); checked by load_state_dict


class FlatAdamW(torch.optim.Optimizer):
    """``FlatAdamW(model)`` (one group, decay everywhere), ``FlatAdamW(group_decay(model), module=model)`` (the
    reference's two groups) or any iterable of parameters / group dicts.  ``module``: where to look for layers that keep
    fused weight buffers (``EdgeGatedGraphConv``); defaults to the module passed as ``params``."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 module: Optional[torch.nn.Module] = None, average_gradients: bool = False, process_group=None):
        """``average_gradients``: data-parallel training without a DDP wrapper - ``step()`` all-reduces the ONE flat
        gradient buffer over ``process_group`` (default group if None) and divides by the world size before the update:
        the packed buffer the optimizer needs anyway IS the collective's buffer (one batched copy per step instead of
        the two of FlatGradSync + optimizer).  The parameters are then marked as "gradient read after backward()"
        (``ops.GRAD_READ_AFTER_BACKWARD``), like FlatGradSync's."""
        if isinstance(params, torch.nn.Module):
            module = params if module is None else module
            params = [p for p in params.parameters() if p.requires_grad]
        self.module = module
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.average_gradients, self.process_group = average_gradients, process_group
        self.last_allreduce_events = None  # (start, end) HIP events around the last all-reduce (bench.py reads them)
        if average_gradients:
            from .ops import GRAD_READ_AFTER_BACKWARD

            for g in self.param_groups:
                for p in g["params"]:
                    setattr(p, GRAD_READ_AFTER_BACKWARD, True)
        self._flat_all = self._grad_all = None
        self.sink_in_flight = False  # a backward of the current step has written into _grad_all in place (cmodel.Binding.sink_plan)
        self._flat: List[Optional[torch.nn.Parameter]] = []  # per group (None: no live parameter in it)
        self._live: List[List[torch.nn.Parameter]] = []
        self._live_idx: List[List[int]] = []
        self._offsets: List[List[int]] = []
        self._inner: Optional[torch.optim.AdamW] = None
        self._adopted = []
        self._warned = False

    # ---- conveniences kept from the round-2 class
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g["lr"] = value

    @property
    def flat(self):
        """The flat parameter buffer (all groups, each a contiguous 64-element-aligned range of it)."""
        return self._flat_all

    @property
    def flat_grad(self):
        """The flat gradient buffer ``step()`` packs (and, with ``average_gradients``, all-reduces)."""
        return self._grad_all

    @property
    def flat_buffers(self):
        return [f for f in self._flat if f is not None]

    @staticmethod
    def _group_order(members, fused, runs):
        """Order of a group's live parameters inside its flat range: fused runs first (each contiguous and in its owner's
        order) - a run is the part of an owner's weight (or bias) list that lives in THIS group; the owner adopts the flat
        slices only if BOTH its lists are complete -, then the (dbeta | dgamma) pairs the whole-model backward writes as one
        block (layout 2; ``runs == []`` gives layout 1), then the rest in group order."""
        ids = {id(p) for p in members}
        order, seen = [], set()
        for owner, ws, bs in fused:
            for run in (ws, bs):
                if all(id(p) in ids for p in run) and not any(id(p) in seen for p in run):
                    for p in run:
                        order.append(p)
                        seen.add(id(p))
        for run in runs:
            if all(id(p) in ids for p in run) and not any(id(p) in seen for p in run):
                for p in run:
                    order.append(p)
                    seen.add(id(p))
        order += [p for p in members if id(p) not in seen]
        return order

    def _migrate_layout_1(self, inner):
        """A state dict written before the gradient runs were laid out first (layout 1): the moments are per FLAT buffer, so
        they are permuted parameter by parameter from the old order (the same rule without the runs) into this one."""
        fused = _fused_groups(self.module) if self.module is not None else []
        inner = {"state": {k: dict(v) for k, v in inner["state"].items()}, "param_groups": inner["param_groups"]}
        k = 0
        for g, idx, new_order, new_offs in zip(self.param_groups, self._live_idx, self._live, self._offsets):
            if not idx:
                continue
            st = inner["state"].get(k)
            k += 1
            if st is None:
                continue
            pos = {id(p): i for i, p in enumerate(g["params"])}
            old_order = self._group_order(sorted(new_order, key=lambda p: pos[id(p)]), fused, [])
            old_off, off = {}, 0
            for p in old_order:
                old_off[id(p)] = off
                off += p.numel()
            for name in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                t = st.get(name)
                if t is None or not torch.is_tensor(t) or t.numel() != off:
                    continue
                out = torch.empty_like(t)
                for p, o in zip(new_order, new_offs):
                    out[o:o + p.numel()] = t[old_off[id(p)]:old_off[id(p)] + p.numel()]
                st[name] = out
        return inner

    # ---- layout
    def _build(self, live_idx: Optional[List[List[int]]] = None):
        """Fix the layout: ``live_idx`` (from a checkpoint) or, by default, the parameters that hold a gradient now."""
        if live_idx is None:
            live_idx = [[i for i, p in enumerate(g["params"]) if p.requires_grad and p.grad is not None]
                        for g in self.param_groups]
            if not any(live_idx):
                raise RuntimeError("FlatAdamW.step() before any backward(): no parameter has a gradient")
        fused = _fused_groups(self.module) if self.module is not None else []
        runs = _gradient_runs(self.module) if self.module is not None else []
        self._flat, self._live, self._live_idx, self._offsets, self._adopted = [], [], [], [], []
        flats = []
        # ONE buffer for all groups (each group a contiguous range of it: one all-reduce covers everything), ranges
        # rounded up to 64 elements so that every group's slice starts 256-byte aligned
        sizes = [sum(g["params"][i].numel() for i in idx) for g, idx in zip(self.param_groups, live_idx)]
        starts, total = [], 0
        for n in sizes:
            starts.append(total)
            total += (n + 63) // 64 * 64
        first = next(g["params"][idx[0]] for g, idx in zip(self.param_groups, live_idx) if idx)
        self._flat_all = torch.zeros(total, dtype=first.dtype, device=first.device)
        self._grad_all = torch.zeros_like(self._flat_all)
        for gi, (g, idx) in enumerate(zip(self.param_groups, live_idx)):
            members = [g["params"][i] for i in idx]
            if not members:
                self._flat.append(None)
                self._live.append([])
                self._live_idx.append([])
                self._offsets.append([])
                continue
            order = self._group_order(members, fused, runs)
            if any(p.dtype != first.dtype or p.device != first.device for p in order):
                raise ValueError("FlatAdamW needs all parameters on one device in one dtype")
            flat = self._flat_all[starts[gi]:starts[gi] + sizes[gi]]
            offs, off = [], 0
            with torch.no_grad():
                for p in order:
                    view = flat[off:off + p.numel()].view(p.shape)
                    view.copy_(p.data)
                    p.data = view
                    offs.append(off)
                    off += p.numel()
            fp = torch.nn.Parameter(flat, requires_grad=True)
            fp.grad = self._grad_all[starts[gi]:starts[gi] + sizes[gi]]
            pos = {id(p): i for i, p in enumerate(g["params"])}
            self._flat.append(fp)
            self._live.append(order)
            self._live_idx.append([pos[id(p)] for p in order])
            self._offsets.append(offs)
            flats.append((fp, g))
        everything = {id(p) for lst in self._live for p in lst}
        for owner, ws, bs in fused:
            if all(id(p) in everything for p in list(ws) + list(bs)):
                try:
                    owner._adopt_fused_buffers()
                    self._adopted.append(owner)
                    continue
                except ValueError:
                    pass
            # a member is frozen / unused / in another dtype: this layer must not re-fuse (that would take its live
            # parameters out of our buffers) - it concatenates per forward instead
            pin = getattr(owner, "_pin_unfused", None)
            if pin is not None:
                pin()
        self._inner = torch.optim.AdamW([dict(params=[fp], **{k: g[k] for k in _HYPER}) for fp, g in flats],
                                        fused=flats[0][0].is_cuda)
        # where every live parameter's gradient lives in the packed buffer: a backward that can write whole blocks
        # (alignn_amd/cmodel.py: the whole-model C calls) writes THERE, and step() has nothing left to gather
        self._slot = {}
        for fp, members, offs in zip(self._flat, self._live, self._offsets):
            if fp is not None:
                for p, off in zip(members, offs):
                    self._slot[id(p)] = fp.grad[off:off + p.numel()].view(p.shape)
        self._register_sink()

    def _register_sink(self):
        if self.module is None:
            return
        try:
            import weakref

            from . import cmodel

            cmodel.model_cache(self.module)["grad_sink"] = weakref.ref(self)
        except Exception:  # (a module that cannot be weakly referenced / hashed: the gather in step() stays)
            pass

    def gradient_slot(self, p):
        """The view of the packed gradient buffer that belongs to parameter ``p`` (None: not a live parameter / not built)."""
        return getattr(self, "_slot", {}).get(id(p))

    def _check_aliasing(self):
        """Every live parameter must still be its slice of the flat buffer; put back the ones that are not."""
        moved = 0
        for fp, members, offs in zip(self._flat, self._live, self._offsets):
            if fp is None:
                continue
            base, es = fp.data_ptr(), fp.element_size()
            for p, off in zip(members, offs):
                if p.data_ptr() != base + off * es or p.device != fp.device or p.dtype != fp.dtype:
                    view = fp.data[off:off + p.numel()].view(p.shape)
                    view.copy_(p.data)
                    p.data = view
                    moved += 1
        if moved:
            for owner in self._adopted:
                owner._adopt_fused_buffers()
            if not self._warned:
                self._warned = True
                warnings.warn(f"FlatAdamW: {moved} parameter(s) had been moved out of the flat buffer since the last step "
                              "(.to(), load_state_dict(assign=True), a re-fused layer?); moved back, values kept")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._inner is None:
            self._build()
        self._check_aliasing()
        k = 0
        for g, fp, members in zip(self.param_groups, self._flat, self._live):
            if fp is None:
                continue
            grads, src, dst = [], [], []
            for p in members:
                if p.grad is None:
                    raise RuntimeError("a parameter that had a gradient at the first step() has none now: FlatAdamW's "
                                       "layout is fixed at the first step")
                grads.append(p.grad.reshape(-1))
                slot = self._slot[id(p)]
                if p.grad.data_ptr() != slot.data_ptr():
                    src.append(p.grad)
                    dst.append(slot)
            if len(src) == len(members):
                torch.cat(grads, out=fp.grad)  # one batched copy
            elif src:  # the backward wrote most gradients in place: copy the few it could not
                torch._foreach_copy_(dst, src)
            ig = self._inner.param_groups[k]
            for h in _HYPER:  # what a scheduler wrote into OUR groups since the last step
                ig[h] = g[h]
            k += 1
        if self.average_gradients:
            self._all_reduce()
        self._inner.step()
        self.sink_in_flight = False  # (the packed gradient buffer may be written in place by the next backward again)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        self.sink_in_flight = False

    def _all_reduce(self):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.process_group)
        if world == 1:
            return
        buf = self._grad_all
        timed = buf.is_cuda and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.process_group)
        buf.div_(world)
        if timed:
            e1.record()
            self.last_allreduce_events = (e0, e1)

    # ---- checkpoints
    def state_dict(self):
        return {
            "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
            "live_idx": [list(ix) for ix in self._live_idx] if self._inner is not None else None,
            "layout": LAYOUT_VERSION,
            "inner": self._inner.state_dict() if self._inner is not None else None,
        }

    def load_state_dict(self, state):
        if "state" in state and "inner" not in state:
            raise ValueError("this is a plain torch optimizer state dict ({'state', 'param_groups'}); FlatAdamW keeps its "
                             "moments per FLAT buffer: load a FlatAdamW.state_dict(), or load into torch.optim.AdamW")
        if "inner" not in state or "live_idx" not in state:
            raise ValueError("not a FlatAdamW state dict (needs 'param_groups', 'live_idx', 'inner')")
        if len(state["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, sg in zip(self.param_groups, state["param_groups"]):
            g.update({k: v for k, v in sg.items() if k != "params"})  # (never the saved index lists: 'params' stay the tensors)
        if state.get("inner") is None:
            return
        layout = state.get("layout", 1)
        if layout not in (1, LAYOUT_VERSION):
            raise ValueError(f"FlatAdamW state dict with parameter layout {layout}: this version reads layouts 1 and {LAYOUT_VERSION}")
        if [list(ix) for ix in state["live_idx"]] != [list(ix) for ix in self._live_idx] or self._inner is None:
            self._build(state["live_idx"])  # (compared by CONTENT: an equal-length but different live set is another layout)
        inner = state["inner"]
        if layout == 1 and self.module is not None and _gradient_runs(self.module):
            inner = self._migrate_layout_1(inner)  # (checkpoints from before the gradient runs were placed first)
        self._inner.load_state_dict(inner)
