"""Round-3 GPU tests: lane-T tensors reaching plain torch consumers, registry hit counters, composite entry points,
float64 gradient goldens, BatchNorm statistics with a pivot, the device neighbour-list kernel, non-float32 modules."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():  # collected on the CPU box too (then deselected by -m "not gpu")
    DEV = torch.device("cpu")
else:
    DEV = torch.device("cuda:0")

from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402


# ---------------------------------------------------------------------------------------------
# lane-T tensors that reach a consumer which is not lane-aware
# ---------------------------------------------------------------------------------------------
def _atomwise(seed=0, **kw):
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    torch.manual_seed(seed)
    cfg = dict(name="alignn_atomwise", alignn_layers=1, gcn_layers=1, hidden_features=64, embedding_features=32,
               atom_input_features=92, calculate_gradient=False)
    cfg.update(kw)
    return ALIGNNAtomWise(ALIGNNAtomWiseConfig(**cfg)).to(DEV).train()


def _steps(model, batch, target, n=3, lanes=None, min_rows=None):
    prev = (ops._LANE["enabled"], ops._LANE["min_rows"])
    if lanes is not None:
        ops._LANE["enabled"], ops._LANE["min_rows"] = lanes, min_rows
    try:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
        losses = []
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            out = model(batch)
            loss = torch.nn.functional.l1_loss(out["out"] if isinstance(out, dict) else out, target)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return losses, {k: p.detach().clone() for k, p in model.named_parameters()}
    finally:
        ops._LANE["enabled"], ops._LANE["min_rows"] = prev


def test_lane_T_output_multiplied_by_the_cutoff_envelope_has_its_event():
    """``y = edge_embedding(d) * c_off`` (alignn_atomwise.py:446-451 with multiply_cutoff): with every layer forced onto
    lane T the multiply - a plain torch kernel on the caller's stream - must wait for the embedding.  Same bits as one
    stream, run to run."""
    raw = make_batch(6, 24, seed0=5)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(6, generator=torch.Generator().manual_seed(3)).to(DEV)
    kw = dict(use_cutoff_function=True, multiply_cutoff=True, inner_cutoff=6.0)
    ref_l, ref_p = _steps(_atomwise(4, **kw), batch, target, lanes="0", min_rows=1)
    for trial in range(3):
        l, p = _steps(_atomwise(4, **kw), batch, target, lanes="1", min_rows=1)
        for a, b in zip(ref_l, l):
            assert torch.equal(a, b), trial
        for k in ref_p:
            assert torch.equal(ref_p[k], p[k]), (trial, k)


def test_lane_T_output_under_capture_with_the_cutoff_multiply():
    """The same model captured into a hipGraph ("auto": lanes only while capturing): replays equal the eager trajectory."""
    from alignn_amd.graphed import GraphedTrainStep

    prev = ops._LANE["min_rows"]
    ops._LANE["min_rows"] = 1
    try:
        raw = make_batch(4, 20, seed0=9)
        batch = GraphBatch.from_raw(raw, device=DEV)
        target = torch.tensor([0.3, -0.2, 0.5, 0.1], device=DEV)
        kw = dict(use_cutoff_function=True, multiply_cutoff=True, inner_cutoff=6.0)

        def fresh():
            m = _atomwise(2, **kw)
            return m, torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)

        loss_fn = lambda o, t: torch.nn.functional.l1_loss(o["out"], t)  # noqa: E731
        m, o = fresh()
        eager = []
        for _ in range(6):
            o.zero_grad(set_to_none=True)
            loss = loss_fn(m(batch), target)
            loss.backward()
            o.step()
            eager.append(loss.detach().clone())
        m2, o2 = fresh()
        step = GraphedTrainStep(m2, batch, target, o2, loss_fn=loss_fn, warmup=3)
        graphed = [step().detach().clone() for _ in range(3)]
        for a, b in zip(eager[3:], graphed):
            assert torch.equal(a, b)
    finally:
        ops._LANE["min_rows"] = prev


def test_non_canonical_graph_conv_on_a_lane():
    """Stand-alone EdgeGatedGraphConv on a DGL-like graph (edge rows permuted into slot order by a torch gather) inside
    lanes(): the gather and the un-permute of the output are plain torch consumers of lane-T tensors."""
    from alignn_amd.alignn import EdgeGatedGraphConv, MLPLayer
    from alignn_amd.graph import build_csr

    torch.manual_seed(1)
    raw = make_batch(3, 18, seed0=21)
    u, v = torch.from_numpy(raw.u).to(DEV), torch.from_numpy(raw.v).to(DEV)

    class G:  # the DGL surface _as_csr touches
        def edges(self):
            return u, v

        def num_nodes(self):
            return raw.num_nodes

    emb = MLPLayer(16, 64).to(DEV).train()
    conv = EdgeGatedGraphConv(64, 64).to(DEV).train()
    x = torch.randn(raw.num_nodes, 64, device=DEV)
    e = torch.randn(raw.num_edges, 16, device=DEV)

    def run(mode):
        prev = (ops._LANE["enabled"], ops._LANE["min_rows"])
        ops._LANE["enabled"], ops._LANE["min_rows"] = mode, 1
        try:
            with ops.lanes(DEV):
                xo, yo = conv(G(), x, emb(e))
                out = (xo.sum() + (yo * yo).sum())
            torch.cuda.synchronize()
            return xo.clone(), yo.clone(), out.clone()
        finally:
            ops._LANE["enabled"], ops._LANE["min_rows"] = prev

    ref = run("0")
    for _ in range(3):
        got = run("1")
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
