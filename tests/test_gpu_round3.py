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


# ---------------------------------------------------------------------------------------------
# the identity-keyed side-band registries: hits and misses are counted, expensive misses warn
# ---------------------------------------------------------------------------------------------
def _default_step_stats(wrap_ddp):
    import socket
    import warnings

    import torch.distributed as dist

    torch.manual_seed(0)
    model = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
    raw = make_batch(16, 60, seed0=77)  # E = 12.6 k, T = 169 k rows: every split-product path of the benchmark is taken
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(16, generator=torch.Generator().manual_seed(2)).to(DEV)
    net = model
    if wrap_ddp:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        net = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)  # alignn/train.py:207
    from alignn_amd import cmodel

    try:
        with warnings.catch_warnings(), cmodel.disabled():  # (the registries belong to the per-operator path)
            warnings.simplefilter("error", RuntimeWarning)  # a fall-back to bf16x6 / separate reductions would raise here
            torch.nn.functional.l1_loss(net(batch), target).backward()  # first step: lazy initialisations
            ops.reset_registry_stats()
            torch.nn.functional.l1_loss(net(batch), target).backward()
        torch.cuda.synchronize()
        return dict(ops.REGISTRY_STATS), dict(ops.BNRED_STATS)
    finally:
        if wrap_ddp:
            dist.destroy_process_group()


def test_registry_hits_of_a_default_config_step_plain_and_under_ddp():
    plain, bn_plain = _default_step_stats(False)
    print("registry stats, plain:", plain, bn_plain)
    # nothing silently dropped to a slow path
    assert plain["bf16x6_fallback"] == 0 and plain["separate_bn_reduce"] == 0 and plain["wimg_miss"] == 0
    # (amax_miss counts tall tensors that arrive without a tracked max|.| - the raw inputs and the RBF expansions, whose
    # projections are not split-product shapes; a miss that COSTS something shows up as bf16x6_fallback above)
    # measured on MI355X at this batch (16 x 60 atoms: only the T-row products are split-product shapes):
    # amax_hit 18, wimg_hit 4 (the W^T images of the four line-graph edge gates), fused = used = 4 with the angle embedding
    # as a chain of layers; with csrc/angle.hip (the default) the embedding's own projections and its BatchNorm-backward
    # sums leave the registries: 3 fused reductions (the line-graph convolutions that feed another one)
    assert plain["amax_hit"] >= 13, plain
    assert plain["wimg_hit"] >= 4
    assert bn_plain["fused"] == bn_plain["used"] == plain["pre_red_hit"] and bn_plain["fused"] >= 3
    ddp, bn_ddp = _default_step_stats(True)
    print("registry stats, DDP-wrapped:", ddp, bn_ddp)
    assert ddp == plain and bn_ddp == bn_plain  # the reducer's hooks and bucket views cost no registry hit


def test_a_cloned_gradient_is_counted_and_warned_about():
    """A hook that clones the gradient between two layers breaks the identity the pre-reduced BatchNorm-backward sums
    travel under: the step stays correct (same gradients up to summation order), the miss is counted and warns once."""
    import warnings

    from alignn_amd.alignn import MLPLayer

    torch.manual_seed(0)
    a, b = MLPLayer(64, 256).to(DEV).train(), MLPLayer(256, 256).to(DEV).train()
    x = torch.randn(40000, 64, device=DEV)

    def run(clone):
        for m in (a, b):
            m.zero_grad(set_to_none=True)
        h = a(x)
        if clone:
            h.register_hook(lambda g: g.clone())
        b(h).square().mean().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]

    ops._WARNED.discard("pre_red")
    ref = run(False)
    ops.reset_registry_stats()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = run(True)
    if ops.BNRED_STATS["fused"]:  # (the product took the f16x3 kernel with the fused sums: then the clone must be noticed)
        assert ops.REGISTRY_STATS["separate_bn_reduce"] == 1
        assert any("BatchNorm-backward sums" in str(m.message) for m in w)
    scale = max(float(p.abs().max()) for p in ref)  # (the Linear biases in front of BatchNorm hold only rounding noise)
    for p, q in zip(ref, got):
        assert float((p - q).abs().max()) <= 1e-5 * scale


# ---------------------------------------------------------------------------------------------
# non-float32 modules: plain torch on the device (alignn_amd/torch_path.py)
# ---------------------------------------------------------------------------------------------
@pytest.mark.filterwarnings("ignore:alignn_amd. torch.float64 tensors run on plain torch")
def test_float64_model_on_the_gpu_reproduces_the_float64_reference():
    from oracle import alignn_oracle as O
    from tests.helpers import load_golden

    z = load_golden("full_cfg1.npz")
    raw = make_batch(8, 60)
    seed = int(z["seed"])
    model = ALIGNN(ALIGNNConfig(name="alignn"))
    model.load_state_dict(O.perturbed_norm_state_dict(O.init_state_dict(seed=seed), seed=seed + 1))
    model = model.double().to(DEV).train()
    batch = GraphBatch.from_raw(raw, device=DEV, dtype=torch.float64)
    pred = model(batch)
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]).double().to(DEV))
    loss.backward()
    assert float((pred.detach().cpu() - torch.from_numpy(z["pred64"])).abs().max()) < 1e-10
    assert abs(loss.item() - float(z["loss64"])) < 1e-12
    # EVERY parameter gradient the reference's float64 backward stored (512 strided samples + 4 moments each), not one:
    # 1e-9 of the gradient's own scale, with a floor of 1e-12 of the model's largest gradient for the analytically-zero ones
    # (Linear biases in front of a norm hold rounding noise on both sides)
    named = dict(model.named_parameters())
    keys = [k for k in z if k.startswith("grad64.")]
    assert len(keys) > 100
    gmax = max(float(np.abs(z[k][:-4]).max()) for k in keys)
    worst = 0.0
    for k in keys:
        g = O.full_size_sample(named[k[len("grad64."):]].grad, 512)
        ref = z[k]
        err = np.abs(g[:-4] - ref[:-4]).max() / max(np.abs(ref[:-4]).max(), 1e-3 * gmax)
        worst = max(worst, err)
        assert err < 1e-9, (k, err)
    print(f"float64 torch path vs the reference's float64 backward: {len(keys)} gradients, worst {worst:.2e}")


# ---------------------------------------------------------------------------------------------
# composite entry points: one C call per convolution forward / backward - same launches, same bits
# ---------------------------------------------------------------------------------------------
def _train_state(mk_model, batch, target, composite, steps=2, use_cmodel=False):
    """``use_cmodel``: the whole-model C entry points (alignn_amd/cmodel.py - the default training path since round 4);
    False = the per-operator path these A/B tests are about."""
    from alignn_amd import cmodel

    prev, prev_c = ops.COMPOSITE, cmodel.ENABLED
    ops.COMPOSITE = composite
    cmodel.ENABLED = use_cmodel
    for k in ops.COMPOSITE_STATS:
        ops.COMPOSITE_STATS[k] = 0
    for k in cmodel.STATS:
        cmodel.STATS[k] = 0
    ops.DW_STATS["fused"] = 0
    try:
        model = mk_model()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            pred = model(batch)
            torch.nn.functional.l1_loss(pred, target).backward()
            opt.step()
        torch.cuda.synchronize()
        out = {"pred": pred.detach().clone()}
        out.update({"g." + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        out.update({"s." + k: v.clone() for k, v in model.state_dict().items()})
        return out, dict(ops.COMPOSITE_STATS)
    finally:
        ops.COMPOSITE = prev
        cmodel.ENABLED = prev_c


@pytest.mark.parametrize("case", ["small_fp32_kernels", "default_16x60_mixed", "default_48x60_split_products", "no_residual_dims"])
def test_composite_entry_points_are_bit_identical_to_the_per_kernel_path(case):
    if case == "small_fp32_kernels":  # H = 64, 5 x 16 atoms: exact-fp32 projections, generic + line-graph reverse kernels
        raw = make_batch(5, 16, seed0=78)

        def mk():
            torch.manual_seed(21)
            return ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=64,
                                       embedding_features=32)).to(DEV).train()
    elif case == "no_residual_dims":  # hidden 96: not a multiple of 64 lanes x 4, odd tile counts
        raw = make_batch(3, 14, seed0=5)

        def mk():
            torch.manual_seed(3)
            return ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=1, gcn_layers=2, hidden_features=96,
                                       embedding_features=48)).to(DEV).train()
    else:
        raw = make_batch(16 if "16x60" in case else 48, 60, seed0=77)

        def mk():
            torch.manual_seed(0)
            return ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(raw.batch_size, generator=torch.Generator().manual_seed(2)).to(DEV)
    a, stats = _train_state(mk, batch, target, True)
    b, stats_off = _train_state(mk, batch, target, False)
    print(case, "composite calls:", stats)
    n_conv = 2 * (len(mk().alignn_layers) * 2 + len(mk().gcn_layers))  # convolutions x steps
    # (a line-graph convolution whose edge-gate projection takes input gradient + weight gradient in one pass - csrc/gemm_dw.hip,
    # H = 256 and >= ops.DW_MIN_ROWS edge rows - runs its backward on the per-kernel path: the composite declines)
    n_dw = ops.DW_STATS["fused"]
    assert n_dw == (2 * len(mk().alignn_layers) if ("default" in case and batch.lg.n_edges >= ops.DW_MIN_ROWS) else 0), n_dw
    assert stats["fwd"] == n_conv and stats["bwd"] == n_conv - n_dw and stats["wgrad"] == n_conv - n_dw, stats
    assert stats_off == {"fwd": 0, "bwd": 0, "wgrad": 0}
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), (case, k)
    # ... and the whole-model entry points (one C call per forward / backward: csrc/model.hip) issue the same launches again
    from alignn_amd import cmodel

    c, stats_c = _train_state(mk, batch, target, True, use_cmodel=True)
    assert stats_c == {"fwd": 0, "bwd": 0, "wgrad": 0} and cmodel.STATS["fwd"] == 2 and cmodel.STATS["bwd"] == 2, cmodel.STATS
    assert a.keys() == c.keys(), set(a) ^ set(c)
    for k in a:
        assert torch.equal(a[k], c[k]), (case, "cmodel", k)


def test_batched_weight_preparation_gives_the_same_bits():
    """ops.WeightPrep (all slice images of the model in one call per step) vs two launches per weight: identical training
    state, and the per-weight slicing launches are gone from the step."""
    raw = make_batch(48, 60, seed0=77)
    batch = GraphBatch.from_raw(raw, device=DEV)
    target = torch.randn(48, generator=torch.Generator().manual_seed(2)).to(DEV)

    def mk():
        torch.manual_seed(0)
        return ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()

    def run(flag):
        prev = ops.BATCHED_WEIGHT_PREP
        ops.BATCHED_WEIGHT_PREP = flag
        try:
            return _train_state(mk, batch, target, True, steps=3)[0]
        finally:
            ops.BATCHED_WEIGHT_PREP = prev

    ops.WEIGHT_PREP_STATS["runs"] = ops.WEIGHT_PREP_STATS["weights"] = 0
    a = run(True)
    assert ops.WEIGHT_PREP_STATS["runs"] == 3 and ops.WEIGHT_PREP_STATS["weights"] == 3 * 26, ops.WEIGHT_PREP_STATS
    b = run(False)
    assert ops.WEIGHT_PREP_STATS["runs"] == 3
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("composite", [True, False])
@pytest.mark.parametrize("loader", [False, True])
def test_segment_ordered_destination_table_gives_the_same_bits(composite, loader):
    """Line-graph edge-gate projection: the destination term read from a segment-ordered copy of P's Bd block
    (alignn_gather_rows_ld + alignn_gemm_nt_f16x3_gather2, ops.segment_ordered_bd) instead of P[dst[e]] - the same addends
    in the same order, so every bit of the training state agrees.  Both ways a batch gets its line graph are covered."""
    raw = make_batch(48, 60, seed0=77)
    if loader:
        from alignn_amd.loader import pack_raw, stage
        batch, _ = stage(pack_raw(raw), DEV)
    else:
        batch = GraphBatch.from_raw(raw, device=DEV)
    lg = batch.lg
    sp = lg.seg_ptr.long()
    assert lg.seg_rank is not None and lg.seg_rank.dtype == torch.int32
    assert torch.equal(lg.seg_rank.long(), torch.repeat_interleave(torch.arange(lg.n_nodes, device=DEV), sp[1:] - sp[:-1]))
    assert torch.equal(lg.seg_node[lg.seg_rank.long()], lg.dst)  # (what makes the two readings the same values)
    target = torch.randn(48, generator=torch.Generator().manual_seed(2)).to(DEV)

    def mk():
        torch.manual_seed(0)
        return ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()

    def run(flag):
        prev = ops.BD_SEGMENT_TABLE
        ops.BD_SEGMENT_TABLE = flag
        try:
            return _train_state(mk, batch, target, composite)[0]
        finally:
            ops.BD_SEGMENT_TABLE = prev

    ops.BD_TABLE_STATS["used"] = 0
    a = run(True)
    assert ops.BD_TABLE_STATS["used"] == 2 * 4, ops.BD_TABLE_STATS  # steps x line-graph convolutions
    b = run(False)
    assert ops.BD_TABLE_STATS["used"] == 2 * 4
    for k in a:
        assert torch.equal(a[k], b[k]), k


# ---------------------------------------------------------------------------------------------
# BatchNorm statistics as Welford slabs: well-conditioned when |mean| >> std
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,F", [(37, 64), (3840, 256), (50712, 256), (200003, 64)])
@pytest.mark.parametrize("offset", [0.0, 300.0])
def test_welford_column_statistics_are_well_conditioned(rows, F, offset):
    """column mean / rstd of X = offset + small noise.  With offset = 300 and std = 1e-2, E[x^2] - mean^2 in float32 has no
    correct digit (mean^2 / var = 9e8); the Welford slabs give rstd to 1e-4 of the float64 value."""
    from alignn_amd import _lib
    from alignn_amd._lib import check, ptr, stream

    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(rows + F)
    noise = torch.randn(rows, F, generator=g, dtype=torch.float64) * (1e-2 if offset else 1.0)
    X64 = offset * (1.0 + 0.01 * torch.arange(F, dtype=torch.float64)) + noise  # a different mean per column
    X = X64.float().to(DEV)
    X64 = X.double().cpu()  # (statistics of what the kernel really sees)
    gamma, beta = torch.ones(F, device=DEV), torch.zeros(F, device=DEV)
    rm, rv = torch.zeros(F, device=DEV), torch.ones(F, device=DEV)
    slabs = lib.alignn_col_stats_slabs(rows)
    partial = torch.empty(slabs * (3 * F + 1), device=DEV)
    stat = torch.empty(4, F, device=DEV)
    check(lib.alignn_col_stats_welford(ptr(X), F, rows, F, ptr(partial), stream()), "col_stats_welford")
    check(lib.alignn_bn_finalize_welford(ptr(partial), slabs, rows, F, ptr(gamma), ptr(beta), 1e-5, 0.1, ptr(rm), ptr(rv),
                                         ptr(stat), stream()), "bn_finalize_welford")
    torch.cuda.synchronize()
    assert float(partial[slabs * 3 * F:].sum()) == rows  # the slabs' counts
    mean64, var64 = X64.mean(0), X64.var(0, unbiased=False)
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    assert float((stat[0].double().cpu() - mean64).abs().max()) <= 2e-7 * float(mean64.abs().max() + 1.0)
    assert float(((stat[1].double().cpu() - rstd64) / rstd64).abs().max()) < 1e-4
    unb = X64.var(0, unbiased=True) if rows > 1 else var64
    assert float(((rv.double().cpu() - (0.9 + 0.1 * unb)) / (0.9 + 0.1 * unb)).abs().max()) < 1e-5


@pytest.mark.parametrize("ratio", [0.0, 3.0, 30.0])
def test_projection_epilogue_statistics_at_the_conditioning_the_model_sees(ratio):
    """The STATS epilogue of the split-product projections (alignn_gemm_nt_f16x3_stats: BatchNorm statistics of the product
    without another pass over it) keeps plain sum / sum-of-squares slabs (float32 inside a 64-row strip, float64 across
    strips) - unlike the pivot slabs of the statistics pass and the gate passes above.  What that costs, measured against
    float64 statistics of the kernel's OWN float32 output: |mean| / std = 0 .. 30 covers every norm input of the model
    (the projections' outputs are centred by construction: |mean| / std < 3 at the benchmark, profiles/r03_parity_full_size
    .txt); a pre-activation with |mean| >> std in front of such a layer is the one place where the pivot pass
    (ops.STATS_FUSED = False) is the better choice, and the bound asserted here says by how much."""
    rows, K, N = 50712, 64, 256
    g = torch.Generator().manual_seed(int(ratio) + 5)
    a = torch.randn(rows, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)  # unit-variance outputs
    b = (ratio * (1.0 + 0.1 * torch.rand(N, generator=g))).to(DEV)
    out, partial, slabs = ops.gemm_nt_f16x3_stats(a, ops.absmax(a), ops.split_f16x2(w), b)
    gamma, beta = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    stat = ops._bn_finalize(partial, slabs, rows, gamma, beta, None, None, False)
    torch.cuda.synchronize()
    o64 = out.double()
    mean64, var64 = o64.mean(0), o64.var(0, unbiased=False)
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    e_mean = float((stat[0].double() - mean64).abs().max() / (mean64.abs().max() + 1.0))
    e_rstd = float(((stat[1].double() - rstd64) / rstd64).abs().max())
    print(f"|mean|/std = {ratio}: mean {e_mean:.2e}, rstd {e_rstd:.2e} (relative)")
    assert e_mean < 1e-6
    assert e_rstd < {0.0: 1e-5, 3.0: 3e-5, 30.0: 1e-3}[ratio]


def test_conv_with_nearly_constant_node_features_matches_float64():
    """Node features 1 + 1e-2 * noise: |mean| / std = 100 on every norm input, i.e. sum x^2 - (sum x)^2 / n would lose four
    of float32's seven digits.  Output of the float32 kernels against the float64 torch path of the SAME module.  (The
    reference's own set-up, x = ones exactly, is not resolvable in float32 at all: there the atoms' pre-activations differ
    in the 7th digit - the DATA is below float32's resolution before any statistic is taken.)"""
    import copy
    import warnings

    from alignn_amd.alignn import EdgeGatedGraphConv
    from alignn_amd.graph import build_csr

    torch.manual_seed(0)
    raw = make_batch(2, 30, seed0=3)
    u, v = torch.from_numpy(raw.u).to(DEV), torch.from_numpy(raw.v).to(DEV)
    csr = build_csr(u, v, raw.num_nodes)
    conv = EdgeGatedGraphConv(64, 64).to(DEV).train()
    conv64 = copy.deepcopy(conv).double()
    x = 1.0 + 1e-2 * torch.randn(raw.num_nodes, 64, device=DEV)
    y = 2.0 + 1e-2 * torch.randn(raw.num_edges, 64, device=DEV)
    xo, yo = conv(csr, x, y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        xo64, yo64 = conv64(csr, x.double(), y.double())
    for a, b in ((xo, xo64), (yo, yo64)):
        assert float((a.double() - b).abs().max()) < 2e-4 * float(b.abs().max())


# ---------------------------------------------------------------------------------------------
# BatchNorm backward of the node norm with the quotient's adjoints in the same pass
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,F", [(1, 64), (3840, 256), (50712, 256), (140001, 256)])
def test_norm_backward_with_node_adjoints_is_the_two_pass_result_bit_for_bit(rows, F):
    from alignn_amd import _lib
    from alignn_amd._lib import check, ptr

    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(rows)
    gy = torch.randn(rows, F, device=DEV, generator=g)
    x = torch.randn(rows, F, device=DEV, generator=g) * 2 + 0.5
    s0 = torch.rand(rows, F, device=DEV, generator=g) * 8
    s0[::7] = 0.0  # (isolated nodes: S0 = 0, the epsilon alone in the denominator)
    hh = torch.randn(rows, F, device=DEV, generator=g)
    gamma = torch.rand(F, device=DEV, generator=g) + 0.5
    stat = torch.stack([x.mean(0), 1 / (x.var(0, unbiased=False) + 1e-5).sqrt(), gamma / (x.var(0, unbiased=False) + 1e-5).sqrt(),
                        torch.randn(F, device=DEV, generator=g)]).contiguous()
    red = torch.randn(2, F, device=DEV, generator=g)
    st = torch.cuda.current_stream().cuda_stream

    GP_a, GP_b = torch.zeros(rows, 4 * F, device=DEV), torch.zeros(rows, 4 * F, device=DEV)
    am_a, am_b = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    gxa, gxb = GP_a[:, 3 * F:], GP_b[:, 3 * F:]
    gs1a, gs0a, gs1b, gs0b = (torch.empty(rows, F, device=DEV) for _ in range(4))
    check(lib.alignn_bn_silu_bwd_apply(ptr(gy), F, ptr(x), F, ptr(stat), ptr(gamma), ptr(red), 0, ptr(gxa), 4 * F, rows, F,
                                       ptr(am_a), st), "apply")
    check(lib.alignn_egc_node_bwd(ptr(gxa), 4 * F, ptr(s0), ptr(hh), ptr(gs1a), ptr(gs0a), rows, F, st), "node_bwd")
    check(lib.alignn_bn_silu_bwd_apply_node(ptr(gy), F, ptr(x), F, ptr(stat), ptr(gamma), ptr(red), 0, ptr(gxb), 4 * F, rows, F,
                                            ptr(am_b), ptr(s0), ptr(hh), ptr(gs1b), ptr(gs0b), st), "apply_node")
    torch.cuda.synchronize()
    assert torch.equal(GP_a, GP_b) and torch.equal(am_a, am_b)
    assert torch.equal(gs1a, gs1b) and torch.equal(gs0a, gs0b)
    # and against float64
    xc = x.double() - stat[0].double()
    z = xc * stat[2].double() + stat[3].double()
    sg = torch.sigmoid(z)
    gz = gy.double() * (sg * (1 + z * (1 - sg)))
    ref = stat[2].double() * (gz - (red[0].double() + xc * stat[1].double() * red[1].double()) / rows)
    assert float((gxb.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    ref1 = ref / (s0.double() + 1e-6)
    assert float((gs1b.double() - ref1).abs().max()) < 2e-5 * float(ref1.abs().max())


# ---------------------------------------------------------------------------------------------
# MD: the force-field evaluation replayed from a hipGraph per batch shape (alignn_amd/md.py)
# ---------------------------------------------------------------------------------------------
def test_graphed_force_field_replays_equal_the_eager_evaluation():
    from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, neighbors
    from alignn_amd.md import GraphedForceField, signature
    from alignn_amd.synthetic import make_crystal

    n = 64
    lat, frac, _ = make_crystal(n, 4321)
    lat_d, frac_d = torch.from_numpy(lat).to(DEV), torch.from_numpy(frac).to(DEV)
    feats = torch.randn(n, 92, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    torch.manual_seed(0)
    model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=64,
                                                 embedding_features=32, atom_input_features=92, calculate_gradient=True,
                                                 stresswise_weight=0.05)).to(DEV).eval()
    ff = GraphedForceField(model, clone=True)
    gen = torch.Generator(device=DEV).manual_seed(7)
    sigs = set()
    for step in range(8):
        # small displacements: the neighbour lists mostly keep their sizes, the bond vectors change every step
        f = frac_d + (1e-5 * step) * torch.randn(frac_d.shape, device=DEV, dtype=frac_d.dtype, generator=gen)
        batch = neighbors.crystal_batch([lat_d], [f], atom_features=[feats])
        sigs.add(signature(batch))
        ref = model(batch)
        got = ff(batch)
        for k in ("out", "grad", "stresses"):
            assert got[k].shape == ref[k].shape, k
            scale = float(ref[k].abs().max())
            assert float((got[k] - ref[k]).abs().max()) <= 1e-6 * scale, (step, k, float((got[k] - ref[k]).abs().max()), scale)
    assert ff.stats["captured"] == len(sigs) <= 4, (ff.stats, len(sigs))
    assert ff.stats["replayed"] == 8 - len(sigs) and ff.stats["replayed"] >= 4, ff.stats
    with pytest.raises(ValueError):
        GraphedForceField(model.train())


def test_gather_projection_with_and_without_the_segment_table_against_float64():
    """The T-row edge-gate projection with DGL's u_add_v in its epilogue (alignn_gemm_nt_f16x3_gather / _gather2, persistent
    kernel): destination term from P or from the segment-ordered table - same bits -, with and without the BatchNorm
    column sums, against float64."""
    raw = make_batch(48, 60, seed0=3)
    lg = GraphBatch.from_raw(raw, device=DEV).lg
    T, E, H = lg.n_edges, lg.n_nodes, 256
    assert T // 128 >= 1024  # (long enough for the persistent kernels)
    g = torch.Generator(device=DEV).manual_seed(0)
    y = torch.randn(T, H, device=DEV, generator=g)
    P = torch.randn(E, 4 * H, device=DEV, generator=g)
    w = torch.randn(H, H, device=DEV, generator=g) / 16
    bias = torch.randn(H, device=DEV, generator=g)
    wh, am = ops.split_f16x2(w), ops.absmax(y)
    bd2 = ops.segment_ordered_bd(P, lg, H)

    def run(stats, table):
        r = ops.gemm_nt_f16x3_gather(y, am, wh, bias, P, lg.src, lg.dst, want_stats=stats, bd2=bd2 if table else None,
                                     rank=lg.seg_rank if table else None)
        torch.cuda.synchronize()
        return (r[0], r[1][:r[2]]) if stats else (r, None)

    ref = y.double() @ w.double().t() + bias.double() + P[lg.src.long(), :H].double() + P[lg.dst.long(), H:2 * H].double()
    for stats in (True, False):
        a, pa = run(stats, True)
        b, pb = run(stats, False)
        assert torch.equal(a, b), stats
        assert float((b.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
        if stats:
            assert torch.equal(pa, pb)
            assert float((pa[:, 0].double().sum(0) - ref.sum(0)).abs().max()) < 1e-5 * float(ref.sum(0).abs().max() + T ** 0.5)
