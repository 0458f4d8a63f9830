"""alignn_amd.optim.FlatAdamW: AdamW on one flat buffer = the per-tensor optimizer, bit for bit."""
import copy

import pytest
import torch

from alignn_amd.optim import FlatAdamW


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 4)
        self.dead = torch.nn.Linear(3, 3)  # never used: grad stays None, torch's AdamW skips it (no weight decay either)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def test_flat_adamw_equals_per_tensor_adamw_on_cpu():
    torch.manual_seed(0)
    m1 = _Net()
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-2, weight_decay=0.05)
    o2 = FlatAdamW(m2, lr=1e-2, weight_decay=0.05)
    for i in range(5):
        x = torch.randn(5, 8)
        if i == 3:  # a schedule: assign the learning rate
            o1.param_groups[0]["lr"] = 3e-3
            o2.lr = 3e-3
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()
    for p, q in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(p, q)
    assert m2.a.weight.data_ptr() == o2.flat.data_ptr()  # parameters are views of the flat buffer
    # `dead` left out (the buffer itself is padded to a multiple of 64 elements per group)
    assert sum(f.numel() for f in o2.flat_buffers) == sum(p.numel() for p in list(m2.a.parameters()) + list(m2.b.parameters()))
    assert o2.flat.numel() % 64 == 0
    assert set(m2.state_dict()) == set(m1.state_dict())


def test_flat_adamw_needs_a_backward_first():
    with pytest.raises(RuntimeError):
        FlatAdamW(_Net()).step()


@pytest.mark.gpu
def test_flat_adamw_on_the_model_is_bit_identical_and_keeps_the_fused_projection_buffers():
    from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch
    from alignn_amd.synthetic import make_batch

    dev = torch.device("cuda", 0)
    batch = GraphBatch.from_raw(make_batch(4, 12, seed0=5), device=dev)
    target = torch.randn(4, generator=torch.Generator().manual_seed(3)).to(dev)
    torch.manual_seed(0)
    cfg = ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=2, hidden_features=64, embedding_features=32)
    m1 = ALIGNN(cfg).to(dev).train()
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-3, fused=True)
    o2 = FlatAdamW(m2, lr=1e-3)
    for _ in range(3):
        losses = []
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            loss = torch.nn.functional.l1_loss(m(batch), target)
            loss.backward()
            o.step()
            losses.append(float(loss))
        assert losses[0] == losses[1]
    torch.cuda.synchronize()
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert torch.equal(p, q), n
    lo, hi = o2.flat.data_ptr(), o2.flat.data_ptr() + o2.flat.numel() * 4
    conv = m2.alignn_layers[0].node_update
    wcat, bcat = conv._fused_node_projection()
    assert lo <= wcat.data_ptr() < hi and lo <= bcat.data_ptr() < hi  # the module adopted the flat slices: no re-fusing
    assert conv.src_gate.weight.data_ptr() == wcat.data_ptr()
    sd = m2.state_dict()
    m3 = ALIGNN(cfg).to(dev)
    m3.load_state_dict(sd)  # a checkpoint of the re-homed model loads like any other
    assert torch.equal(m3.eval()(batch), m2.eval()(batch))


# ---------------------------------------------------------------------------------------------
# round 3: a real torch.optim.Optimizer: schedulers, the reference's parameter groups, checkpoints, aliasing
# ---------------------------------------------------------------------------------------------
class _NamedNet(torch.nn.Module):
    """Names as in the reference model: ``bn_nodes`` / ``*.bias`` go to the no-decay group of ``group_decay``."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 12)
        self.bn_nodes = torch.nn.BatchNorm1d(12)
        self.out = torch.nn.Linear(12, 3)

    def forward(self, x):
        return self.out(torch.nn.functional.silu(self.bn_nodes(self.lin(x))))


def test_group_decay_is_the_reference_rule():
    from alignn_amd.optim import group_decay

    m = _NamedNet()
    groups = group_decay(m)
    names = {id(p): n for n, p in m.named_parameters()}
    assert sorted(names[id(p)] for p in groups[0]["params"]) == ["lin.weight", "out.weight"]
    assert sorted(names[id(p)] for p in groups[1]["params"]) == ["bn_nodes.bias", "bn_nodes.weight", "lin.bias", "out.bias"]
    assert groups[1]["weight_decay"] == 0 and "weight_decay" not in groups[0]


def test_flat_adamw_is_an_optimizer_onecycle_and_two_groups_match_torch():
    """The reference's loop: ``AdamW(group_decay(net))`` + ``OneCycleLR`` (alignn/train.py:209-226).  The scheduler
    writes lr AND betas into param_groups every step; both optimizers must end bit-identical."""
    from alignn_amd.optim import group_decay

    torch.manual_seed(1)
    m1 = _NamedNet()
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.AdamW(group_decay(m1), lr=1e-2, weight_decay=0.1)
    o2 = FlatAdamW(group_decay(m2), lr=1e-2, weight_decay=0.1, module=m2)
    assert isinstance(o2, torch.optim.Optimizer)
    s1 = torch.optim.lr_scheduler.OneCycleLR(o1, max_lr=1e-2, epochs=2, steps_per_epoch=4)
    s2 = torch.optim.lr_scheduler.OneCycleLR(o2, max_lr=1e-2, epochs=2, steps_per_epoch=4)
    for _ in range(8):
        x = torch.randn(7, 6)
        for m, o, s in ((m1, o1, s1), (m2, o2, s2)):
            o.zero_grad()
            m(x).square().mean().backward()
            o.step()
            s.step()
    assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"]
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert torch.equal(p, q), n
    assert len(o2.flat_buffers) == 2 and o2.param_groups[1]["weight_decay"] == 0


def test_flat_adamw_checkpoint_round_trip():
    torch.manual_seed(2)
    m1 = _NamedNet()
    o1 = FlatAdamW(m1, lr=5e-3)
    xs = [torch.randn(5, 6) for _ in range(6)]

    def run(m, o, batch):
        o.zero_grad()
        m(batch).square().mean().backward()
        o.step()

    for x in xs[:3]:
        run(m1, o1, x)
    ck_model, ck_opt = copy.deepcopy(m1.state_dict()), copy.deepcopy(o1.state_dict())
    for x in xs[3:]:
        run(m1, o1, x)
    # resume in a fresh process: new model + new optimizer, NO backward before load_state_dict
    m2 = _NamedNet()
    m2.load_state_dict(ck_model)
    o2 = FlatAdamW(m2, lr=1.0)  # (the checkpoint's hyper-parameters win)
    o2.load_state_dict(ck_opt)
    assert o2.param_groups[0]["lr"] == 5e-3
    for x in xs[3:]:
        run(m2, o2, x)
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert torch.equal(p, q), n


def test_flat_adamw_puts_rehomed_parameters_back():
    """Something moves a parameter out of the flat buffer between steps (``load_state_dict(assign=True)``, ``.to()``): the
    next step must train THAT value, not a stale slice."""
    torch.manual_seed(3)
    m1 = _NamedNet()
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-2)
    o2 = FlatAdamW(m2, lr=1e-2)
    xs = [torch.randn(5, 6) for _ in range(4)]
    for i, x in enumerate(xs):
        if i == 2:
            with torch.no_grad():
                new = torch.randn_like(m1.lin.weight)
                m1.lin.weight.copy_(new)
                m2.lin.weight.data = new.clone()  # re-homed: no longer a view of the flat buffer
            assert m2.lin.weight.data_ptr() != o2.flat.data_ptr()
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x).square().mean().backward()
            if i == 2 and o is o2:
                with pytest.warns(UserWarning, match="moved out of the flat buffer"):
                    o.step()
            else:
                o.step()
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert torch.equal(p, q), n
    lo = o2.flat.data_ptr()
    assert lo <= m2.lin.weight.data_ptr() < lo + o2.flat.numel() * 4


def test_flat_adamw_refuses_a_plain_torch_state_dict_and_compares_layouts_by_content():
    """a standard torch AdamW state dict must not silently replace the parameter lists with index lists; a saved
    layout with the same NUMBER of live parameters but other members is another layout."""
    import copy

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(8, 2))
    ref = torch.optim.AdamW(net.parameters(), lr=1e-3)
    net(torch.randn(3, 4)).sum().backward()
    ref.step()
    opt = FlatAdamW(net, lr=1e-3)
    opt.step()
    with pytest.raises(ValueError):
        opt.load_state_dict(ref.state_dict())
    assert all(isinstance(p, torch.nn.Parameter) for g in opt.param_groups for p in g["params"])
    sd = copy.deepcopy(opt.state_dict())
    with pytest.raises(ValueError):
        opt.load_state_dict({"param_groups": sd["param_groups"]})
    opt.load_state_dict(sd)  # round trip still works
    assert all(isinstance(p, torch.nn.Parameter) for g in opt.param_groups for p in g["params"])


class _RunNet(torch.nn.Module):
    """A module that declares gradient runs (as the LayerNorm / BatchNorm layers of the models do): layout 2 places them first."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.norm = torch.nn.LayerNorm(16)
        self.b = torch.nn.Linear(16, 4)

    def _gradient_runs(self):
        return [[self.norm.bias, self.norm.weight]]

    def forward(self, x):
        return self.b(torch.tanh(self.norm(self.a(x))))


def test_flat_adamw_reads_a_layout_1_state_dict_and_permutes_the_moments(monkeypatch):
    """checkpoints written before the gradient runs were placed first (no 'layout' key = layout 1) load - the moments
    are permuted parameter by parameter - and training continues exactly as the old optimizer would have."""
    import alignn_amd.optim as O

    torch.manual_seed(0)
    m_old = _RunNet()
    m_new = copy.deepcopy(m_old)
    xs = [torch.randn(5, 8) for _ in range(5)]
    with monkeypatch.context() as mp:  # the old layout: the same rule without the runs
        mp.setattr(O, "_gradient_runs", lambda module: [])
        o_old = FlatAdamW(m_old, lr=1e-2, weight_decay=0.05)
        for x in xs[:3]:
            o_old.zero_grad()
            m_old(x).square().mean().backward()
            o_old.step()
        sd = copy.deepcopy(o_old.state_dict())
        sd.pop("layout")
        weights = copy.deepcopy(m_old.state_dict())
        for x in xs[3:]:
            o_old.zero_grad()
            m_old(x).square().mean().backward()
            o_old.step()
    m_new.load_state_dict(weights)
    o_new = FlatAdamW(m_new, lr=1e-2, weight_decay=0.05)
    m_new(xs[0]).square().mean().backward()  # (fixes the live set)
    o_new.load_state_dict(sd)
    assert [id(p) for p in o_new._live[0]][:2] == [id(m_new.norm.bias), id(m_new.norm.weight)]  # layout 2: the run comes first
    assert [id(p) for p in o_old._live[0]][:2] != [id(m_old.norm.bias), id(m_old.norm.weight)]
    m_new.load_state_dict(weights)
    for x in xs[3:]:
        o_new.zero_grad()
        m_new(x).square().mean().backward()
        o_new.step()
    for (k, p), q in zip(m_old.named_parameters(), m_new.parameters()):
        assert torch.equal(p, q), k
    with pytest.raises(ValueError):
        o_new.load_state_dict(dict(sd, layout=7))
