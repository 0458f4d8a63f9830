"""Generate the BASELINE-size goldens (tests/golden/full_*.npz) by running the REFERENCE's own model classes.

Run in the authoring container only (needs /root/reference, ~50 GB of RAM, ~10 minutes):

    python oracle/make_golden_full.py [cfg2] [cfg5] [cfg4] [cfg1] [cfg2q]

Test infrastructure: imported by nothing in the product path.  Same method as oracle/make_golden.py (the
reference's ``alignn/models/*.py`` unmodified on ``oracle/shims``), at the EXACT sizes BASELINE.json quotes:

* cfg2  ``ALIGNN`` default config, ``make_batch(64, 60)``              (N=3 840, E=50 712, T=676 200)
* cfg1  the same model on ``make_batch(8, 60)``                        (the reference's CPU plumbing case; + float64 fwd+bwd)
* cfg2q the same model on ``make_batch(16, 60)`` = the first 16 crystals of cfg2, with a float64 forward AND backward
* cfg5  ``ALIGNN`` default config, ``make_batch(256, (9, 27), kind="molecule")``
* cfg4  ``ALIGNNAtomWise`` 4+4 / H=256 / forces + stresses on ``make_batch(16, 200)``.  The force head keeps the
  double-backward graph of a 200-atom crystal alive (~4 GB per crystal), so the batch is evaluated in four chunks
  of four crystals - exact for this model: LayerNorm has no batch statistic, crystals of a batch do not interact,
  and the loss terms are means over the FULL batch (chunk losses use ``reduction="sum"`` over the full-batch
  denominators, parameter gradients accumulate over the chunks).

The inputs are NOT stored (make_batch regenerates them bit-for-bit from the seeds; ``in.sig`` pins that), the
parameters come from ``oracle.alignn_oracle.init_state_dict`` (seeded generator) and big tensors are stored as
strided samples + moments, so the fixtures stay small.
"""

from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import dgl  # noqa: E402,F401  (the shim)
from alignn.models.alignn import ALIGNN, ALIGNNConfig, EdgeGatedGraphConv  # noqa: E402  (the reference)

from alignn_amd.synthetic import make_batch, batch_raw, _one  # noqa: E402
from oracle.alignn_oracle import full_size_sample, init_state_dict, input_signature, perturbed_norm_state_dict  # noqa: E402
from oracle.make_golden import to_dgl  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _hook(model, conv_cls, store):
    for name, mod in model.named_modules():
        if isinstance(mod, conv_cls):

            def hook(_m, _inp, out, name=name):
                store["act." + name + ".x_out"] = full_size_sample(out[0])
                store["act." + name + ".y_out"] = full_size_sample(out[1])

            mod.register_forward_hook(hook)


def case_alignn(tag, raw, seed, grad64=False):
    """One training step (forward, L1 loss, backward) of the reference's ALIGNN, default config, BatchNorm train mode."""
    t0 = time.time()
    model = ALIGNN(ALIGNNConfig(name="alignn"))
    model.load_state_dict(perturbed_norm_state_dict(init_state_dict(seed=seed), seed=seed + 1))
    model.train()
    g, lg, lat = to_dgl(raw)
    B = raw.batch_size
    out = {"in.sig": input_signature(raw), "seed": seed}
    _hook(model, EdgeGatedGraphConv, out)
    target = torch.randn(B, generator=torch.Generator().manual_seed(1))
    pred = model((g, lg, lat))
    loss = torch.nn.functional.l1_loss(pred, target)
    loss.backward()
    out["target"], out["pred"], out["loss"] = target.numpy(), pred.detach().numpy(), loss.item()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad." + k] = full_size_sample(p.grad)
    out["nograd"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
    for k, v in model.state_dict().items():
        if "running" in k:
            out["sd_after." + k] = v.numpy().copy()
    # the same forward in float64 (no autograd: train-mode BatchNorm statistics and the prediction only): separates our
    # rounding from the reference's own float32 rounding - its BatchNorm sums 676 k rows per feature in float32
    m64 = ALIGNN(ALIGNNConfig(name="alignn")).double()
    m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in
                         perturbed_norm_state_dict(init_state_dict(seed=seed), seed=seed + 1).items()})
    m64.train()
    g64, lg64, lat64 = to_dgl(raw)
    for d in (g64.ndata, g64.edata, lg64.edata):
        for k in list(d.keys()):
            if d[k].is_floating_point():
                d[k] = d[k].double()
    if grad64:
        # ... and, where the RAM allows it (cfg 1, and cfg 2's first 16 crystals), the float64 BACKWARD as well: parameter
        # gradients free of the float32 reference's own T-row summation error
        pred64 = m64((g64, lg64, lat64.double()))
        loss64 = torch.nn.functional.l1_loss(pred64, target.double())
        loss64.backward()
        out["pred64"], out["loss64"] = pred64.detach().numpy(), loss64.item()
        for k, p in m64.named_parameters():
            if p.grad is not None:
                out["grad64." + k] = full_size_sample(p.grad)
    else:
        with torch.no_grad():
            out["pred64"] = m64((g64, lg64, lat64.double())).numpy()
    for k, v in m64.state_dict().items():
        if "running" in k:
            out["sd_after64." + k] = v.numpy().copy()
    del m64
    np.savez_compressed(os.path.join(OUT, f"full_{tag}.npz"), **out)
    print(f"{tag}: N={raw.num_nodes} E={raw.num_edges} T={raw.num_triplets} loss {loss.item():.6f} "
          f"pred[:3] {pred.detach().numpy()[:3]} ({time.time() - t0:.0f} s)", flush=True)


def case_cfg4(B=16, atoms=200, chunk=4, seed=40):
    from alignn.models.alignn_atomwise import ALIGNNAtomWise, ALIGNNAtomWiseConfig

    t0 = time.time()
    cfg = ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=4, gcn_layers=4, hidden_features=256,
                               atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05)
    model = ALIGNNAtomWise(cfg)
    sd = perturbed_norm_state_dict(init_state_dict(seed=seed), seed=seed + 1)
    sd = {k: v for k, v in sd.items() if "running" not in k and "tracked" not in k}
    model.load_state_dict(sd)
    model.train()
    graphs = [_one(atoms, 1234 + i, "crystal", 92) for i in range(B)]
    full = batch_raw(graphs)
    N = full.num_nodes
    te = torch.randn(B, generator=torch.Generator().manual_seed(1))
    tf = torch.randn(N, 3, generator=torch.Generator().manual_seed(7))
    ts = torch.randn(B, 3, 3, generator=torch.Generator().manual_seed(8))
    out = {"in.sig": input_signature(full), "seed": seed, "t_energy": te.numpy(), "t_forces": tf.numpy(), "t_stress": ts.numpy()}
    preds, forces, stresses, loss_total, off = [], [], [], 0.0, 0
    for c in range(0, B, chunk):
        raw = batch_raw(graphs[c:c + chunk])
        g, lg, lat = to_dgl(raw)
        vol = np.abs(np.linalg.det(raw.lattice.astype(np.float64))).astype(np.float32)
        g.ndata["V"] = torch.from_numpy(np.repeat(vol, raw.batch_num_nodes))
        res = model((g, lg, lat))
        n = raw.num_nodes
        L = torch.nn.functional.l1_loss
        loss = (L(res["out"], te[c:c + chunk], reduction="sum") / B
                + L(res["grad"], tf[off:off + n], reduction="sum") / (N * 3)
                + L(res["stresses"], ts[c:c + chunk], reduction="sum") / (B * 9))
        loss.backward()  # accumulates into .grad over the chunks
        loss_total += loss.item()
        preds.append(res["out"].detach().numpy().copy())
        forces.append(res["grad"].detach().numpy().copy())
        stresses.append(res["stresses"].detach().numpy().copy())
        off += n
        del res, loss, g, lg
        print(f"cfg4 chunk {c // chunk}: {time.time() - t0:.0f} s", flush=True)
    out["pred"], out["forces"], out["stresses"] = np.concatenate(preds), np.concatenate(forces), np.concatenate(stresses)
    out["loss"] = loss_total
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad." + k] = full_size_sample(p.grad)
    out["nograd"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
    np.savez_compressed(os.path.join(OUT, "full_cfg4.npz"), **out)
    print(f"cfg4: N={N} E={full.num_edges} T={full.num_triplets} loss {loss_total:.6f} E[:3] {out['pred'][:3]} "
          f"|F|max {np.abs(out['forces']).max():.4f} ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg5", "cfg4"]
    torch.set_num_threads(os.cpu_count() or 8)
    if "cfg1" in which:
        case_alignn("cfg1", make_batch(8, 60), seed=10, grad64=True)
    if "cfg2q" in which:  # the first 16 crystals of the cfg-2 batch, with the float64 backward (the full 64 need > 62 GB)
        case_alignn("cfg2q", make_batch(16, 60), seed=20, grad64=True)
    if "cfg2" in which or "cfg2g64" in which:  # cfg2g64: + the float64 backward (peak RSS ~ see DESIGN section 2)
        case_alignn("cfg2", make_batch(64, 60), seed=20, grad64="cfg2g64" in which)
    if "cfg5" in which:
        case_alignn("cfg5", make_batch(256, (9, 27), kind="molecule"), seed=50)
    if "cfg4" in which:
        case_cfg4()
