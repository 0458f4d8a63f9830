"""CPU baseline bookkeeping: one training step (fwd + bwd + AdamW) of the default ALIGNN on the
benchmark batch (64 x 60-atom crystals), (a) with the REFERENCE's own alignn/models/alignn.py on the torch-only DGL
shim and (b) with the travelling port (oracle/alignn_oracle.py) that bench.py's cpu_baseline uses on the GPU box, on
the same host.  Authoring container only (needs /root/reference).  -> profiles/r02_cpu_reference_vs_port.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from alignn.models.alignn import ALIGNN, ALIGNNConfig  # noqa: E402  (reference)

from alignn_amd.synthetic import make_batch  # noqa: E402
from oracle import alignn_oracle as O  # noqa: E402
from oracle.make_golden import to_dgl  # noqa: E402

if __name__ == "__main__":
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    raw = make_batch(64, 60)
    target = torch.randn(64, generator=torch.Generator().manual_seed(1))
    out = {"host_cpus": cores, "batch": {"graphs": 64, "N": raw.num_nodes, "E": raw.num_edges, "T": raw.num_triplets}}
    # (a) the reference's class
    model = ALIGNN(ALIGNNConfig(name="alignn")).train()
    model.load_state_dict(O.init_state_dict(seed=0))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    ts = []
    for i in range(3):
        g, lg, lat = to_dgl(raw)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.l1_loss(model((g, lg, lat)), target).backward()
        opt.step()
        ts.append(time.perf_counter() - t0)
    out["reference_model_code_on_torch_cpu_dgl_shim"] = {"seconds_per_step": [round(t, 2) for t in ts], "graphs_per_s": round(64 / min(ts[1:]), 3)}
    del model, opt
    # (b) the port
    p = O.as_params(O.init_state_dict(seed=0))
    opt = torch.optim.AdamW([t for t in p.values() if t.requires_grad], lr=1e-3)
    g = O.TorchGraph(raw)
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.l1_loss(O.alignn_forward(p, g, 4, 4, True), target).backward()
        opt.step()
        ts.append(time.perf_counter() - t0)
    out["oracle_port"] = {"seconds_per_step": [round(t, 2) for t in ts], "graphs_per_s": round(64 / min(ts[1:]), 3)}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_cpu_reference_vs_port.json"), "w"), indent=1)
    print(json.dumps(out))
