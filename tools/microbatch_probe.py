"""Force training (BASELINE configs[3]: 16 x 200 atoms) as ONE batch against the same 16 crystals as `parts` micro-batches whose
forwards / backwards run on their own streams and bindings (cmodel.slot): the LayerNorm model couples nothing across crystals, so
the loss and every gradient are the same numbers (up to summation order); what changes is that the bond-row chains of one part
run under the triplet-row kernels of the other.   python tools/microbatch_probe.py [parts=2] [B=16] [atoms=200] [steps=10]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch, cmodel  # noqa: E402
from alignn_amd.optim import FlatAdamW, group_decay  # noqa: E402
from alignn_amd.synthetic import batch_raw, make_graphs  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
atoms = int(sys.argv[3]) if len(sys.argv) > 3 else 200
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda", 0)
graphs = make_graphs(B, atoms, seed0=1234)
torch.manual_seed(0)
model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=4, gcn_layers=4, hidden_features=256,
                                            atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05)).to(dev).train()
l1 = torch.nn.functional.l1_loss
gen = torch.Generator().manual_seed(1)
n_nodes = sum(g.num_nodes for g in graphs)
t_e, t_f, t_s = torch.randn(B, generator=gen).to(dev), torch.randn(n_nodes, 3, generator=gen).to(dev), torch.randn(B, 3, 3, generator=gen).to(dev)


def build(k):
    per = B // k
    out = []
    n0 = 0
    for i in range(k):
        gs = graphs[i * per:(i + 1) * per]
        nn = sum(g.num_nodes for g in gs)
        out.append((GraphBatch.from_raw(batch_raw(gs), device=dev), t_e[i * per:(i + 1) * per], t_f[n0:n0 + nn], t_s[i * per:(i + 1) * per]))
        n0 += nn
    return out


def run(k):
    sub = build(k)
    opt = FlatAdamW(group_decay(model), lr=1e-3, weight_decay=1e-2, module=model)
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]

    def step():
        for p in model.parameters():
            p.grad = None
        cur = torch.cuda.current_stream()
        losses = []
        for i, (b, te, tf, ts) in enumerate(sub):
            s = streams[i] if k > 1 else cur
            if k > 1:
                s.wait_stream(cur)
            with cmodel.slot(i), torch.cuda.stream(s):
                o = model(b)
                losses.append((l1(o["out"], te) + l1(o["grad"], tf) + l1(o["stresses"], ts)) / k)
        if k > 1:
            for s in streams:
                cur.wait_stream(s)
        loss = losses[0]
        for x in losses[1:]:
            loss = loss + x
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    return dt / steps * 1e3, t_enq / steps * 1e3, float(loss), g


state = {k: v.clone() for k, v in model.state_dict().items()}
res = {}
for k in sorted({1, parts}):
    model.load_state_dict(state)
    res[k] = run(k)
    print(f"parts={k}: {res[k][0]:.3f} ms/step (host enqueue {res[k][1]:.2f} ms), loss {res[k][2]:.6f}", flush=True)
