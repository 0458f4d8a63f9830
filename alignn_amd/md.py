"""Energies / forces / stresses for molecular dynamics with the model's launches replayed from a hipGraph.

``alignn/ff/calculators.py:280-291`` rebuilds the graph of the SAME atoms at every ASE step and calls the model on it.
For a cell of a few hundred atoms the ~400 launches of ``ALIGNNAtomWise`` (eval mode, forces by one fused backward w.r.t.
the bond vectors) are launch-bound: ~4 ms of host time for ~1.5 ms of GPU work.  The library only enqueues on the
current stream and never allocates or synchronises, so the whole evaluation can be captured once per batch SHAPE and
replayed: every kernel argument is a pointer into the batch's tensors or a size derived from (N, E, T), never a value read
back from the device.  While the neighbour lists keep their sizes (a solid between rearrangements) consecutive MD steps
have the same shape; ``GraphedForceField`` keeps one captured graph per shape (a few), copies the new batch's index and
feature arrays into that graph's static batch and replays it.  A batch of a new shape is evaluated eagerly once more
while its graph is captured.

The results are those of ``model(batch)`` - same kernels, same order (tests/test_gpu_round3.py).
"""

from __future__ import annotations

import collections
from typing import Dict, Optional

import torch

from . import ops
from .graph import CSRGraph, GraphBatch

_CSR_TENSORS = ("seg_ptr", "seg_node", "src", "dst", "out_ptr", "out_slot", "perm", "inv", "grp_seg_ptr", "grp_src_ptr",
                "seg_rank")
_BATCH_TENSORS = ("graph_ptr", "atom_features", "r", "h", "volume", "extra_features", "r_from_positions")


def _tensors(b: GraphBatch):
    """(name, tensor) of every tensor a forward can read from the batch, in a fixed order."""
    for tag, csr in (("g", b.g), ("lg", b.lg)):
        if csr is None:
            continue
        for k in _CSR_TENSORS:
            t = getattr(csr, k)
            if t is not None:
                yield f"{tag}.{k}", t
    for k in _BATCH_TENSORS:
        t = getattr(b, k)
        if t is not None:
            yield k, t


def signature(b: GraphBatch):
    """What a captured evaluation is specific to: every size a launch derives a grid / a loop bound / a kernel choice from."""
    sizes = [b.batch_size, b.g.n_nodes, b.g.n_edges, b.g.dense_max_src > 0]
    if b.lg is not None:
        sizes += [b.lg.n_nodes, b.lg.n_edges, b.lg.dense_max_src > 0]
    return tuple(sizes) + tuple((k, tuple(t.shape), t.dtype) for k, t in _tensors(b))


def _clone_csr(c: Optional[CSRGraph], shared: Dict[int, torch.Tensor]) -> Optional[CSRGraph]:
    if c is None:
        return None
    kw = {}
    for f in c.__dataclass_fields__:
        v = getattr(c, f)
        if isinstance(v, torch.Tensor):
            # L(g)'s block pointers ARE g's out_ptr / seg_ptr (graph.line_graph_of): keep them one tensor in the copy too
            if id(v) not in shared:
                shared[id(v)] = v.clone()
            v = shared[id(v)]
        kw[f] = v
    return CSRGraph(**kw)


def clone_batch(b: GraphBatch) -> GraphBatch:
    shared: Dict[int, torch.Tensor] = {}
    out = GraphBatch(g=_clone_csr(b.g, shared), lg=_clone_csr(b.lg, shared), graph_ptr=b.graph_ptr.clone(),
                     batch_size=b.batch_size)
    for k in _BATCH_TENSORS[1:]:
        t = getattr(b, k)
        if t is not None:
            setattr(out, k, t.clone())
    return out


class _Captured:
    def __init__(self, model, batch: GraphBatch, warmup: int):
        self.batch = clone_batch(batch)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # allocator pools, lazy kernel attributes, weight images
            for _ in range(warmup):
                model(self.batch)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        ops.reset_amax_arena()  # the arena's zero-fill must be a node of the graph
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = model(self.batch)
        ops.reset_amax_arena()
        self.static = list(_tensors(self.batch))

    def run(self, batch: GraphBatch):
        done = set()
        for (k, dst), (k2, src) in zip(self.static, _tensors(batch)):
            assert k == k2 and dst.shape == src.shape, (k, k2)
            if id(dst) not in done:  # (shared tensors once)
                done.add(id(dst))
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.out


class GraphedForceField:
    """``ff = GraphedForceField(model); res = ff(batch)`` - ``model`` an ``ALIGNNAtomWise`` (or any module of this package that
    maps a ``GraphBatch`` to a dict of tensors) in eval mode with frozen weights; ``batch`` e.g. from
    ``neighbors.crystal_batch``.  Returns the model's output dict; its tensors are the captured graph's static outputs and
    are overwritten by the next call with a batch of the same shape (``clone=True``: fresh copies).

    ``max_graphs`` shapes are kept (least recently used first out); ``stats`` counts replays / captures."""

    def __init__(self, model, max_graphs: int = 4, warmup: int = 2, clone: bool = False):
        if model.training:
            raise ValueError("GraphedForceField replays an evaluation with frozen weights: call model.eval() first")
        if warmup < 1:  # (the first evaluation makes one-time host-to-device copies - weight descriptor tables - that must
            raise ValueError("GraphedForceField needs warmup >= 1")  # not land inside the capture)
        self.model, self.max_graphs, self.warmup, self.clone = model, max_graphs, warmup, clone
        self.graphs: "collections.OrderedDict[tuple, _Captured]" = collections.OrderedDict()
        self.stats = {"replayed": 0, "captured": 0}

    def __call__(self, batch: GraphBatch):
        if batch.device.type != "cuda":
            return self.model(batch)
        # whether forces / stresses are part of the evaluation depends on the grad mode at CAPTURE time (ALIGNNAtomWise
        # takes them by differentiating the energy): a graph captured under no_grad must not serve a grad-enabled call
        key = signature(batch) + (torch.is_grad_enabled(),)
        cap = self.graphs.get(key)
        if cap is None:
            cap = _Captured(self.model, batch, self.warmup)  # (its capture evaluated this very batch)
            self.graphs[key] = cap
            while len(self.graphs) > self.max_graphs:
                self.graphs.popitem(last=False)
            self.stats["captured"] += 1
            # the capture only RECORDS: run it once for this batch's values
            out = cap.run(batch)
        else:
            self.graphs.move_to_end(key)
            self.stats["replayed"] += 1
            out = cap.run(batch)
        if self.clone:
            return {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
        return out
