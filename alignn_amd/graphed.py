"""Whole training step as one hipGraph.

A step of the fused path is ~600 kernel launches.  At the benchmark batch (64 crystals) the GPU needs ~22 ms for
them and the host ~12 ms to enqueue them, so the launch stream stays ahead; at small batches (8 crystals: ~3 ms
of GPU work) the step is launch-bound.  The library never allocates, frees or synchronises and launches only on
the current stream, so a step can be captured once and replayed: ``GraphedTrainStep`` captures forward + loss +
backward (+ the side-stream weight gradients, joined inside the capture) + the optimizer step for a FIXED batch
structure (same GraphBatch object; feature/target values may change between replays via ``copy_``).

This is the standard ``torch.cuda.CUDAGraph`` whole-network capture (hipGraph on ROCm); the point here is only
that the ctypes-launched kernels are capture-safe.
"""

from __future__ import annotations

import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, batch, target, optimizer, loss_fn=torch.nn.functional.l1_loss, warmup: int = 3):
        self.model, self.batch, self.optimizer = model, batch, optimizer
        # ``target`` may be a tensor or a tuple of tensors (energy, forces, stresses for the force-field head);
        # ``loss_fn(model_output, target)`` gets it back in the same shape
        self.target = tuple(t.clone() for t in target) if isinstance(target, (tuple, list)) else target.clone()
        self.loss = None
        # warm-up on a side stream (allocator pools, lazy kernel attributes, the side stream itself)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                loss_fn(model(batch), self.target).backward()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        ops.reset_amax_arena()  # the arena's zero-fill must be a node of the graph, not something done before it
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):  # (other threads, e.g. RCCL's watchdog, may poll events)
            self.loss = loss_fn(model(batch), self.target)
            self.loss.backward()
            optimizer.step()
        ops.reset_amax_arena()  # ... and eager code after the capture must not draw slots from the graph's pool

    def set_target(self, target):
        if isinstance(self.target, tuple):
            for dst, src in zip(self.target, target):
                dst.copy_(src)
        else:
            self.target.copy_(target)

    def __call__(self):
        """Replay one training step; returns the (static) loss tensor of that step."""
        self.graph.replay()
        return self.loss
