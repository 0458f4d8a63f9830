"""Pin the CPU oracle (oracle/alignn_oracle.py) to the reference's own model code.

The golden files were produced by /root/reference/alignn/models/alignn.py run unmodified on the
DGL shim (oracle/make_golden.py).  CPU-only; runs everywhere.
"""

import numpy as np
import torch

from oracle import alignn_oracle as O
from tests.helpers import load_golden, raw_from_golden, rel_err, sample, state_dict_from_golden

TOL = 2e-5  # fp32 summation-order noise between index_add orders; north_star's bar is 1e-4


def _run(z, train, dtype=torch.float32):
    raw = raw_from_golden(z)
    g = O.TorchGraph(raw)
    p = O.as_params(state_dict_from_golden(z), dtype)
    stats, rec = {}, {}
    pred = O.alignn_forward(p, g, int(z["cfg.alignn_layers"]), int(z["cfg.gcn_layers"]), training=train, stats=stats, record=rec)
    return p, pred, stats, rec


def test_tiny_train_forward_backward_stats():
    z = load_golden("alignn_tiny_train.npz")
    p, pred, stats, rec = _run(z, True)
    assert rel_err(pred, z["pred"]) < TOL
    for i in range(2):
        assert rel_err(rec[f"alignn.{i}.x"], z[f"act.alignn_layers.{i}.node_update.x_out"]) < TOL
        assert rel_err(rec[f"alignn.{i}.y"], z[f"act.alignn_layers.{i}.edge_update.x_out"]) < TOL
        assert rel_err(rec[f"alignn.{i}.z"], z[f"act.alignn_layers.{i}.edge_update.y_out"]) < TOL
        assert rel_err(rec[f"gcn.{i}.x"], z[f"act.gcn_layers.{i}.x_out"]) < TOL
        assert rel_err(rec[f"gcn.{i}.y"], z[f"act.gcn_layers.{i}.y_out"]) < TOL
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]))
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    checked = 0
    for k, t in p.items():
        if not t.requires_grad:
            continue
        if k in nograd:
            assert t.grad is None or float(t.grad.abs().max()) == 0.0
            continue
        assert rel_err(t.grad, z["grad." + k], floor=gfloor) < 2e-4, k
        checked += 1
    assert checked > 50
    after = O.running_stats_after_step(p, stats)
    for k, v in after.items():
        assert rel_err(v, z["sd_after." + k]) < TOL, k


def test_tiny_eval_uses_running_stats():
    z = load_golden("alignn_tiny_eval.npz")
    with torch.no_grad():
        _, pred, _, rec = _run(z, False)
    assert rel_err(pred, z["pred"]) < TOL
    assert rel_err(rec["gcn.1.x"], z["act.gcn_layers.1.x_out"]) < TOL


def test_default_config_train():
    z = load_golden("alignn_default_train.npz")
    raw = raw_from_golden(z)
    p = O.as_params(O.init_state_dict(seed=0))
    stats, rec = {}, {}
    pred = O.alignn_forward(p, O.TorchGraph(raw), 4, 4, True, stats, rec)
    assert rel_err(pred, z["pred"]) < 1e-4
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]))
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    for i in range(4):
        assert rel_err(sample(rec[f"alignn.{i}.z"]), z[f"act.alignn_layers.{i}.edge_update.y_out"]) < 1e-4
        assert rel_err(sample(rec[f"gcn.{i}.x"]), z[f"act.gcn_layers.{i}.x_out"]) < 1e-4
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v[:-3]).max()) for k, v in z.items() if k.startswith("grad."))
    for k, t in p.items():
        if t.requires_grad and k not in nograd:
            assert rel_err(sample(t.grad)[:-3], z["grad." + k][:-3], floor=gfloor) < 1e-3, k
    after = O.running_stats_after_step(p, stats)
    for k, v in after.items():
        assert rel_err(v, z["sd_after." + k]) < 1e-4, k


def test_conv_f64_isolated_node_multi_edges():
    z = load_golden("conv_f64.npz")
    dt = torch.float64
    p = {"c." + k[3:]: torch.from_numpy(v).clone().requires_grad_(v.dtype == np.float64 and "running" not in k)
         for k, v in z.items() if k.startswith("sd.")}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = torch.from_numpy(z["y"]).requires_grad_(True)
    xo, yo = O.edge_gated_conv(p, "c", torch.from_numpy(z["u"]), torch.from_numpy(z["v"]), x, y, True)
    assert xo.dtype == dt
    assert rel_err(xo, z["x_out"]) < 1e-12 and rel_err(yo, z["y_out"]) < 1e-12
    ((xo * torch.from_numpy(z["wx"])).sum() + (yo * torch.from_numpy(z["wy"])).sum()).backward()
    assert rel_err(x.grad, z["gx"]) < 1e-11 and rel_err(y.grad, z["gy"]) < 1e-11
    for k, v in z.items():
        if k.startswith("grad."):
            assert rel_err(p["c." + k[5:]].grad, v, floor=1e-3) < 1e-11, k


def test_rbf_gamma_matches_reference_constants():
    # edge gamma 9.875, angle gamma 19.5 (probed from the reference class)
    _, g_e = O.rbf_expand(torch.zeros(1), 0.0, 8.0, 80)
    _, g_a = O.rbf_expand(torch.zeros(1), -1.0, 1.0, 40)
    assert abs(g_e - 9.875) < 1e-4 and abs(g_a - 19.5) < 1e-4


def test_atomwise_layernorm_flavour_energy_path():
    """ALIGNNAtomWise (LayerNorm, lg_on_fly cosines) vs the reference's own class."""
    z = load_golden("atomwise_tiny_train.npz")
    raw = raw_from_golden(z)
    p = O.as_params(state_dict_from_golden(z))
    rec = {}
    pred = O.alignn_atomwise_forward(p, O.TorchGraph(raw), 2, 2, True, rec)
    assert rel_err(pred, z["pred"]) < TOL
    for i in range(2):
        assert rel_err(rec[f"alignn.{i}.z"], z[f"act.alignn_layers.{i}.edge_update.y_out"]) < TOL
        assert rel_err(rec[f"gcn.{i}.x"], z[f"act.gcn_layers.{i}.x_out"]) < TOL
    loss = torch.nn.functional.l1_loss(pred, torch.from_numpy(z["target"]))
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, t in p.items():
        if t.requires_grad and k not in nograd:
            assert rel_err(t.grad, z["grad." + k], floor=gfloor) < 2e-4, k
            n += 1
    assert n > 40


def test_atomwise_force_stress_head_second_order():
    """Oracle FF path (forces by create_graph autograd, stress, penalty) vs the reference's own class,
    including the second-order parameter gradients of an energy+force+stress loss."""
    z = load_golden("atomwise_ff_tiny.npz")
    raw = raw_from_golden(z)
    p = O.as_params(state_dict_from_golden(z))
    out, forces, stresses = O.alignn_atomwise_forward(
        p, O.TorchGraph(raw), 2, 2, True, calculate_gradient=True, stress=True,
        volume=torch.from_numpy(z["volume"]), batch_num_edges=torch.from_numpy(raw.batch_num_edges))
    assert rel_err(out, z["pred"]) < TOL
    assert rel_err(forces, z["forces"]) < 1e-4
    assert rel_err(stresses, z["stresses"]) < 1e-4
    L = torch.nn.functional.l1_loss
    loss = (L(out, torch.from_numpy(z["t_energy"])) + L(forces, torch.from_numpy(z["t_forces"]))
            + 0.05 * L(stresses, torch.from_numpy(z["t_stress"])))
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    nograd = set(z["nograd"].tolist())
    gfloor = 1e-2 * max(float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad."))
    n = 0
    for k, t in p.items():
        if t.requires_grad and k not in nograd:
            assert rel_err(t.grad, z["grad." + k], floor=gfloor) < 1e-3, k
            n += 1
    assert n > 40
