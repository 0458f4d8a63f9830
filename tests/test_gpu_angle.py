"""csrc/angle.hip: the bond-angle embedding (RBF -> MLPLayer -> MLPLayer, alignn/models/alignn.py:215-222) computed without
its [T, bins] / [T, 64] / [T, 256] intermediates, against the same chain of torch modules in float64: the output, the
running statistics and the gradients of all eight parameter tensors."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from alignn_amd.alignn import MLPLayer, RBFExpansion  # noqa: E402
from alignn_amd.angle import AngleEmbedding  # noqa: E402

DEV = "cuda"


def _modules(seed, bins=40):
    torch.manual_seed(seed)
    rbf = RBFExpansion(vmin=-1.0, vmax=1.0, bins=bins)
    l1, l2 = MLPLayer(bins, 64), MLPLayer(64, 256)
    with torch.no_grad():  # non-trivial affine parameters and running statistics
        for m in (l1, l2):
            m.layer[1].weight.uniform_(0.5, 1.5)
            m.layer[1].bias.uniform_(-0.5, 0.5)
    return rbf, l1, l2


def _reference(rbf, l1, l2, h, gz):
    """float64 torch: z, parameter gradients, running statistics after one training-mode forward"""
    import copy

    r64 = torch.exp(-rbf.gamma * (h.double()[:, None] - rbf.centers.double()[None, :]) ** 2)
    m1, m2 = copy.deepcopy(l1.layer).double(), copy.deepcopy(l2.layer).double()
    m1.train(), m2.train()
    z = m2(m1(r64))
    z.backward(gz.double())
    grads = [[m[0].weight.grad, m[0].bias.grad, m[1].bias.grad, m[1].weight.grad] for m in (m1, m2)]
    stats = [[m[1].running_mean, m[1].running_var] for m in (m1, m2)]
    return z.detach(), grads, stats


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("rows,bins,seed", [(1000, 40, 0), (128 * 37 + 5, 40, 1), (31, 40, 2), (200_003, 40, 3), (4097, 24, 4),
                                            (676_200, 40, 5)])
def test_forward_and_parameter_gradients_against_float64(rows, bins, seed):
    rbf, l1, l2 = _modules(seed, bins)
    g = torch.Generator().manual_seed(seed)
    h = torch.rand(rows, generator=g) * 2 - 1  # cosines
    gz = torch.randn(rows, 256, generator=g) * (torch.rand(256, generator=g) + 0.1) + 0.05
    z_ref, grads_ref, stats_ref = _reference(rbf, l1, l2, h, gz)
    rbf, l1, l2 = rbf.to(DEV), l1.to(DEV).train(), l2.to(DEV).train()
    emb = AngleEmbedding(rbf.centers, rbf.gamma, (l1, l2))
    z, amax = emb.forward(h.to(DEV))
    (gW1, gb1, red1), (gW2, gb2, red2) = emb.backward(gz.to(DEV))
    torch.cuda.synchronize()
    z, z_ref = z.cpu(), z_ref
    # activations: 1e-4 rel is the north-star bar; float32 arithmetic on float64's problem lands two orders below it
    assert _rel(z, z_ref) < 2e-6, _rel(z, z_ref)
    assert abs(float(amax) - float(z.abs().max())) == 0.0
    for m, (rm, rv) in zip((l1, l2), stats_ref):
        assert _rel(m.layer[1].running_mean.cpu(), rm) < 1e-5
        assert _rel(m.layer[1].running_var.cpu(), rv) < 1e-5
    got = [[gW1, gb1, red1[:64], red1[64:]], [gW2, gb2, red2[:256], red2[256:]]]
    names = ["weight", "bias (analytically 0)", "norm.bias", "norm.weight"]
    for layer, (gl, rl) in enumerate(zip(got, grads_ref)):
        for name, a, b in zip(names, gl, rl):
            a = a.cpu().reshape(b.shape)
            if name.startswith("bias"):  # Linear bias in front of BatchNorm: exact gradient 0, both sides hold rounding noise
                scale = float(rl[0].abs().max())
                assert float(a.abs().max()) < 1e-4 * max(scale, 1.0), (layer, name, float(a.abs().max()))
                continue
            assert _rel(a, b) < 2e-5, (layer, name, _rel(a, b))


def test_two_runs_are_bit_identical():
    rbf, l1, l2 = _modules(7)
    g = torch.Generator().manual_seed(7)
    h = (torch.rand(50_000, generator=g) * 2 - 1).to(DEV)
    gz = torch.randn(50_000, 256, generator=g).to(DEV)
    rbf, l1, l2 = rbf.to(DEV), l1.to(DEV).train(), l2.to(DEV).train()
    outs = []
    for _ in range(2):
        emb = AngleEmbedding(rbf.centers, rbf.gamma, (l1, l2))
        z, _ = emb.forward(h)
        grads = emb.backward(gz)
        torch.cuda.synchronize()
        outs.append([z.clone()] + [t.clone() for g3 in grads for t in g3])
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_a_training_step_with_the_fused_embedding_matches_the_chain_of_layers():
    """ALIGNN.forward takes csrc/angle.hip for its angle embedding (ops.ANGLE_FUSED): one optimizer-free training step on
    both code paths (per-operator and whole-model C path) against the same step with the embedding as a chain of layers -
    predictions, every gradient and the running statistics agree to float32 rounding; the two code paths agree bit for bit."""
    from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, cmodel, ops
    from alignn_amd.synthetic import make_batch

    raw = make_batch(16, 60, seed0=17)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(16, generator=torch.Generator().manual_seed(1)).to(DEV)

    def step(fused, use_c):
        ops.ANGLE_FUSED = fused
        prev = cmodel.ENABLED
        cmodel.ENABLED = use_c
        try:
            torch.manual_seed(0)
            m = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
            pred = m(batch)
            torch.nn.functional.l1_loss(pred, target).backward()
            torch.cuda.synchronize()
            out = {"pred": pred.detach().clone()}
            out.update({"g." + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
            out.update({"s." + k: v.clone() for k, v in m.state_dict().items()})
            return out
        finally:
            ops.ANGLE_FUSED, cmodel.ENABLED = True, prev

    fused_c, fused_ops, chain = step(True, True), step(True, False), step(False, True)
    assert fused_c.keys() == fused_ops.keys() == chain.keys()
    for k in fused_c:
        assert torch.equal(fused_c[k], fused_ops[k]), k
    gmax = max(float(v.abs().max()) for k, v in chain.items() if k.startswith("g."))
    for k, b in chain.items():
        a = fused_c[k]
        if b.dtype.is_floating_point:
            tol = 2e-5 * (gmax if k.startswith("g.") else max(float(b.abs().max()), 1e-6))
            assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)
        else:
            assert torch.equal(a, b), k


def test_evaluation_mode_without_autograd_takes_the_one_pass_embedding():
    """model.eval() under no_grad: alignn_angle_embed_infer (BatchNorm from the running statistics, one pass) in the per-operator
    path and in the whole-model C path - bit-identical to each other, equal to the chain of layers to float32 rounding, and
    the running statistics untouched."""
    from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, cmodel, ops
    from alignn_amd.synthetic import make_batch

    raw = make_batch(16, 60, seed0=23)
    batch = GraphBatch.from_raw(raw, device=DEV)
    torch.manual_seed(0)
    m = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV)
    m.train()
    for _ in range(2):  # running statistics away from their initial values
        m(batch)
    m.eval()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    outs = {}
    with torch.no_grad():
        for name, fused, use_c in (("c", True, True), ("ops", True, False), ("chain", False, True)):
            ops.ANGLE_FUSED, prev = fused, cmodel.ENABLED
            cmodel.ENABLED = use_c
            try:
                outs[name] = m(batch).clone()
            finally:
                ops.ANGLE_FUSED, cmodel.ENABLED = True, prev
    torch.cuda.synchronize()
    assert torch.equal(outs["c"], outs["ops"])
    scale = float(outs["chain"].abs().max())
    assert float((outs["c"] - outs["chain"]).abs().max()) <= 2e-5 * scale
    after = m.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
