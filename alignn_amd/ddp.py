"""Data-parallel gradient exchange for the ALIGNN hot path: one flat fp32 bucket, one all-reduce.

The reference wraps the model in ``torch.nn.parallel.DistributedDataParallel`` (alignn/train.py:207,
backend string "nccl" at alignn/train_alignn.py:37 - RCCL on ROCm).  The whole model is 4.03 M fp32
parameters = 16.1 MB, so instead of DDP's 25 MB-capped reverse-order buckets we keep every gradient
as a view into ONE contiguous buffer and issue a single ``all_reduce`` per step: on the 8-GPU xGMI
ring that moves 2*(7/8)*16.1 MB per GPU (~0.2 ms at 153 GB/s per link) against >10 ms of compute,
so there is nothing to gain from finer buckets, and one large message is what the point-to-point
links like.  BatchNorm statistics stay per rank, as in the reference (plain BatchNorm1d under DDP).

Works with any backend (``gloo`` on CPU for the tests, ``nccl``==RCCL on the GPUs).

Sharding is by graph (a batch is a disjoint union - no bond crosses crystals), so there is no data-path collective.
A step costs in proportion to the number of bond PAIRS (rows of the line graph), not to the number of crystals:
``triplet_count`` + ``shard_by_cost`` hand each rank an equal share of that.
"""

from __future__ import annotations

from typing import Iterable, List, Sequence

import numpy as np
import torch
import torch.distributed as dist


class FlatGradSync:
    """``zero_grad()`` -> backward -> ``sync()``: gradients are packed into one flat buffer with a single
    batched copy, all-reduced once, and handed back to the parameters as views (no unpack copy).
    Parameters that autograd never touched keep ``grad is None`` (the reference's optimizer skips them:
    e.g. ``bn_edges`` of the last layer, whose edge output is dead)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        # this class reads ``.grad`` only in sync(), after backward() has returned and joined the side stream: tell
        # alignn_amd.ops that the weight gradients of these parameters may stay on it until then even though a
        # process group exists (ops._deferred_join_is_safe; under DistributedDataParallel they may not)
        from .ops import GRAD_READ_AFTER_BACKWARD

        for p in self.params:
            setattr(p, GRAD_READ_AFTER_BACKWARD, True)
        self.group = process_group
        self.flat = None
        self.views = None
        self.used = None

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def sync(self):
        """Average gradients over ranks with one collective.  Call after backward()."""
        world = self._world()
        if world == 1:
            return
        if self.used is None:
            # the set of parameters that receive gradients is a property of the model, not of the batch;
            # agree on it once (MAX over ranks) so every rank packs the same layout
            flags = torch.tensor([p.grad is not None for p in self.params], dtype=torch.int32, device=self.params[0].device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            self.used = [bool(f) for f in flags.tolist()]
            live = [p for p, u in zip(self.params, self.used) if u]
            total = sum(p.numel() for p in live)
            self.flat = torch.zeros(total, dtype=live[0].dtype, device=live[0].device)
            self.views, off = [], 0
            for p in live:
                self.views.append(self.flat[off : off + p.numel()].view_as(p))
                off += p.numel()
            self.live = live
        grads = [
            (p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.live
        ]
        torch.cat(grads, out=self.flat)  # one batched copy into the bucket
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)
        for p, v in zip(self.live, self.views):
            p.grad = v


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """Rank ``src``'s parameters and buffers to everyone (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)


def triplet_count(u, v, num_nodes: int) -> int:
    """Rows of L(g) of one crystal from its bond list alone: pairs (e1, e2) with dst(e1) == src(e2), e1 != e2
    (``g.line_graph`` semantics, alignn/graphs.py:588) - what the step time is proportional to."""
    u = np.asarray(u, dtype=np.int64)
    v = np.asarray(v, dtype=np.int64)
    indeg = np.bincount(v, minlength=num_nodes)
    return int(indeg[u].sum() - np.count_nonzero(u == v))


def shard_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Split items (crystals of a global batch) into ``world`` shards of near-equal total cost: longest-processing-
    time-first greedy (largest item to the currently lightest shard; ties -> lower rank, so every rank computes the
    same partition without communicating).  Returns the item indices of each rank, ascending."""
    if world < 1:
        raise ValueError("world must be >= 1")
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += costs[i]
    return [sorted(s) for s in shards]
