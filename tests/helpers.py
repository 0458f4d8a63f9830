"""Shared helpers for the parity tests."""

import os

import numpy as np
import torch

from alignn_amd.synthetic import RawGraph

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def raw_from_golden(z, prefix="in."):
    return RawGraph(**{k: z[prefix + k] for k in (
        "u v r atom_features lg_u lg_v h batch_num_nodes batch_num_edges batch_num_triplets lattice".split())})


def state_dict_from_golden(z, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(prefix)}


def rel_err(a, b, floor=1e-30):
    """max |a-b| / max(|b|_inf, floor): the 'within 1e-4 rel' of BASELINE.json's north_star,
    measured against the tensor's scale (elementwise relative error is meaningless at zero crossings).
    ``floor`` is for tensors that are mathematically zero (e.g. the gradient of a bias that feeds
    straight into BatchNorm), where both sides hold only rounding noise."""
    a = torch.as_tensor(a).detach().double().reshape(-1).cpu()
    b = torch.as_tensor(b).detach().double().reshape(-1).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


def sample(t, k=97):
    f = torch.as_tensor(t).detach().reshape(-1).double().cpu()
    idx = torch.linspace(0, f.numel() - 1, min(k, f.numel())).long()
    return np.concatenate([f[idx].numpy(), [f.mean().item(), f.abs().mean().item(), f.norm().item()]])
