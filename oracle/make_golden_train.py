"""Run the REFERENCE's ``alignn.train.train_dgl`` (unmodified, on oracle/shims, reference ``ALIGNNAtomWise`` passed as
``model=``) for two epochs on a tiny seeded force-field dataset, store ``history_train.json`` / ``history_val.json`` as
tests/golden/train_loop.npz, and check that oracle/train_loop_oracle.py reproduces them with the same model class.

    python oracle/make_golden_train.py        (authoring container only: needs /root/reference)
"""

import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import dgl  # noqa: E402  (shim)
from alignn.models.alignn_atomwise import ALIGNNAtomWise, ALIGNNAtomWiseConfig  # noqa: E402  (reference)
from alignn.train import train_dgl  # noqa: E402  (reference)

from oracle import train_loop_oracle as TL  # noqa: E402
from oracle.train_data import MODEL_KW, TRAIN_CFG, make_loaders, stress_targets  # noqa: E402


def fresh_model():
    torch.manual_seed(5)
    return ALIGNNAtomWise(ALIGNNAtomWiseConfig(**MODEL_KW))


if __name__ == "__main__":
    out = tempfile.mkdtemp()
    cfg = dict(TRAIN_CFG, output_dir=out, model=dict(MODEL_KW))
    tr, va, te = make_loaders(dgl)
    prepare = lambda batch, device=None, non_blocking=False: batch  # noqa: E731
    torch.set_num_threads(8)
    train_dgl(cfg, model=fresh_model(), train_val_test_loaders=[tr, va, te, prepare])
    h_tr = json.load(open(os.path.join(out, "history_train.json")))
    h_va = json.load(open(os.path.join(out, "history_val.json")))
    print("reference history_train", h_tr)
    print("reference history_val  ", h_va)
    tr, va, te = make_loaders(dgl)
    o_tr, o_va = TL.train_atomwise(fresh_model(), tr, va, cfg, "cpu", lambda g: stress_targets(dgl, g))
    d = max(np.abs(np.array(o_tr) - np.array(h_tr)).max(), np.abs(np.array(o_va) - np.array(h_va)).max())
    print("restated loop vs reference train_dgl: max |difference| =", d)
    assert d < 1e-6, "oracle/train_loop_oracle.py does not reproduce the reference's loop"
    sd = {k: v.numpy() for k, v in fresh_model().state_dict().items()}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_loop.npz"), history_train=np.array(h_tr),
                        history_val=np.array(h_va), **{"sd." + k: v for k, v in sd.items()})
