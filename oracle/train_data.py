"""The tiny seeded force-field dataset + configuration shared by oracle/make_golden_train.py (reference run) and the GPU
test: DGL-shaped batches ``(g, lg, lattice, target)`` with ``g.ndata`` = atom_features, V, atomwise_grad (force targets),
stresses (the crystal's [3,3] target copied to every atom) and ``g.edata['r']`` - what the reference's collate function
(alignn/lmdb_dataset.py:87-108 / graphs.py collate_line_graph) hands to alignn/train.py.  Test infrastructure."""

from types import SimpleNamespace

import numpy as np
import torch

from alignn_amd.synthetic import batch_raw, _one

MODEL_KW = dict(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=64, embedding_features=32,
                atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05, graphwise_weight=1.0,
                gradwise_weight=1.0)
TRAIN_CFG = dict(dataset="user_data", target="target", epochs=2, batch_size=2, learning_rate=1e-3, weight_decay=1e-5,
                 scheduler="onecycle", optimizer="adamw", random_seed=123, compute_line_graph=True, num_workers=0,
                 use_lmdb=False, write_predictions=False)


class _Loader(list):
    def __init__(self, batches, ids):
        super().__init__(batches)
        self.dataset = SimpleNamespace(ids=ids)


def _batch(dgl, seeds, sizes):
    raw = batch_raw([_one(n, s, "crystal", 92) for s, n in zip(seeds, sizes)])
    t = torch.from_numpy
    g = dgl.graph((t(raw.u), t(raw.v)), num_nodes=raw.num_nodes)
    g._bnn, g._bne = t(raw.batch_num_nodes), t(raw.batch_num_edges)
    gen = torch.Generator().manual_seed(1000 + seeds[0])
    vol = np.abs(np.linalg.det(raw.lattice.astype(np.float64))).astype(np.float32)
    g.ndata["atom_features"] = t(raw.atom_features)
    g.ndata["V"] = t(np.repeat(vol, raw.batch_num_nodes))
    g.ndata["atomwise_grad"] = 0.1 * torch.randn(raw.num_nodes, 3, generator=gen)
    st = 0.5 * torch.randn(len(sizes), 3, 3, generator=gen)
    g.ndata["stresses"] = torch.repeat_interleave(st, t(raw.batch_num_nodes), dim=0)
    g.edata["r"] = t(raw.r)
    lg = dgl.graph((t(raw.lg_u), t(raw.lg_v)), num_nodes=raw.num_edges)
    lg._bnn, lg._bne = t(raw.batch_num_edges), t(raw.batch_num_triplets)
    lg.edata["h"] = t(raw.h)
    target = torch.randn(len(sizes), generator=gen)
    return g, lg, t(raw.lattice), target


def stress_targets(dgl, g):
    """train.py:337-342: torch.stack([gg.ndata['stresses'][0] for gg in dgl.unbatch(g)])"""
    return torch.stack([gg.ndata["stresses"][0] for gg in dgl.unbatch(g)])


def make_loaders(dgl):
    tr = _Loader([_batch(dgl, (3000 + 2 * i, 3001 + 2 * i), (8 + i, 11 - i)) for i in range(4)], [f"tr{i}" for i in range(4)])
    va = _Loader([_batch(dgl, (3100 + 2 * i, 3101 + 2 * i), (9, 7 + i)) for i in range(2)], [f"va{i}" for i in range(2)])
    te = _Loader([_batch(dgl, (3200, 3201), (6, 10))], ["te0"])
    return tr, va, te
