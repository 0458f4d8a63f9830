"""Restatement of the part of ``alignn.train.train_dgl`` that DRIVES the hot path: the per-batch loop of the
``"alignn_" in config.model.name`` branch (alignn/train.py:238-625) - parameter groups (``group_decay``,
alignn/utils.py:77-90), ``setup_optimizer`` (AdamW, :93-108), the OneCycle / constant schedule (:209-226), L1 losses on
energy, forces and stresses with the model's weights (:291-387), ``loss.backward(); optimizer.step()``, the validation
pass (:425-546), ``history_train`` / ``history_val`` as the reference writes them to JSON.

Test infrastructure (the caller side of the drop-in boundary, SURVEY.md section 8 "next"): the reference's own
``train_dgl`` cannot travel to the GPU box, so oracle/make_golden_train.py runs it here - unmodified, on oracle/shims,
with the reference's model class - stores its histories as a golden, and checks that THIS loop reproduces them with the
same model (pinned).  The GPU test then runs this loop with ``alignn_amd.ALIGNNAtomWise`` passed as ``model=`` would be.
"""

from __future__ import annotations

import numpy as np
import torch
from torch import nn


def group_decay(model):
    """alignn/utils.py:77-90"""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        (no_decay if ("bias" in name or "bn" in name or "norm" in name) else decay).append(p)
    return [{"params": decay}, {"params": no_decay, "weight_decay": 0}]


def _losses(result, dats, cfg, criterion, device, stress_targets):
    m = cfg["model"]
    loss1 = m.get("graphwise_weight", 1.0) * criterion(result["out"], dats[-1].to(device))  # train.py:291-299
    loss3 = loss4 = 0
    if m.get("calculate_gradient", True):  # :322-333
        loss3 = m.get("gradwise_weight", 1.0) * criterion(result["grad"].to(device), dats[0].ndata["atomwise_grad"].to(device))
    if m.get("stresswise_weight", 0.0) != 0:  # :334-358 (one [3,3] target per crystal: the first atom's copy)
        loss4 = m["stresswise_weight"] * criterion(result["stresses"].to(device), stress_targets(dats[0]).to(device))
    return loss1, loss3, loss4


def train_atomwise(net, train_loader, val_loader, cfg, device, stress_targets):
    """-> (history_train, history_val): per epoch [total, energy, atomwise(0), forces, stress, additional(0)]"""
    criterion = nn.L1Loss()
    net.to(device)
    # upstream builds the optimizer TWICE (train.py:207-208 and again :241-242 inside this branch) and attaches the
    # scheduler to the FIRST one (:209-226): the optimizer that takes the steps never sees the OneCycle schedule and runs
    # at the constant ``learning_rate``.  Restated as is.
    first = torch.optim.AdamW(group_decay(net), lr=cfg["learning_rate"], weight_decay=cfg.get("weight_decay", 0))
    if cfg.get("scheduler", "onecycle") == "onecycle":
        scheduler = torch.optim.lr_scheduler.OneCycleLR(first, max_lr=cfg["learning_rate"], epochs=cfg["epochs"],
                                                        steps_per_epoch=len(train_loader), pct_start=0.3)
    else:
        scheduler = torch.optim.lr_scheduler.LambdaLR(first, lambda epoch: 1.0)
    optimizer = torch.optim.AdamW(group_decay(net), lr=cfg["learning_rate"], weight_decay=cfg.get("weight_decay", 0))
    history_train, history_val = [], []
    f = lambda v: v.item() if torch.is_tensor(v) else float(v)  # noqa: E731
    for _ in range(cfg["epochs"]):
        run = np.zeros(6)
        for dats in train_loader:
            optimizer.zero_grad()
            result = net([dats[0].to(device), dats[1].to(device), dats[2].to(device)])
            l1, l3, l4 = _losses(result, dats, cfg, criterion, device, stress_targets)
            loss = l1 + 0 + l3 + l4 + 0
            loss.backward()
            optimizer.step()
            run += [f(loss), f(l1), 0.0, f(l3), f(l4), 0.0]
        scheduler.step()  # (once per EPOCH, as the reference does: train.py:399)
        history_train.append(run.tolist())
        val = np.zeros(6)
        for dats in val_loader:
            optimizer.zero_grad()
            result = net([dats[0].to(device), dats[1].to(device), dats[2].to(device)])
            l1, l3, l4 = _losses(result, dats, cfg, criterion, device, stress_targets)
            val += [f(l1 + l3 + l4), f(l1), 0.0, f(l3), f(l4), 0.0]
        history_val.append(val.tolist())
    return history_train, history_val
