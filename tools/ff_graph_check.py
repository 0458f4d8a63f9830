"""ALIGNN-FF training step (BASELINE configs[3]: 16 x 200-atom crystals, energy + forces + stress loss, the loss
differentiates through the forces): eager vs one hipGraph.  The composed double-backward is ~2000 small launches and
host-bound when run eagerly."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch
from alignn_amd.graphed import GraphedTrainStep
from alignn_amd.synthetic import make_batch

B, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 200)
dev = "cuda"
raw = make_batch(B, n)
batch = GraphBatch.from_raw(raw, device=dev)
gen = torch.Generator().manual_seed(1)
target = (torch.randn(B, generator=gen).to(dev), torch.randn(raw.num_nodes, 3, generator=gen).to(dev),
          torch.randn(B, 3, 3, generator=gen).to(dev))
l1 = torch.nn.functional.l1_loss
loss_fn = lambda o, t: l1(o["out"], t[0]) + l1(o["grad"], t[1]) + l1(o["stresses"], t[2])

def fresh():
    torch.manual_seed(0)
    m = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=4, gcn_layers=4, hidden_features=256,
                                             atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05)).to(dev).train()
    return m, torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)

m, o = fresh()
def eager():
    o.zero_grad(set_to_none=True)
    l = loss_fn(m(batch), target); l.backward(); o.step(); return l
for _ in range(2): eager()
torch.cuda.synchronize(); t = time.perf_counter(); le = [eager() for _ in range(5)]
torch.cuda.synchronize(); te = (time.perf_counter() - t) / 5
m2, o2 = fresh()
g = GraphedTrainStep(m2, batch, target, o2, loss_fn=loss_fn, warmup=2)
torch.cuda.synchronize(); t = time.perf_counter(); lg = [g().clone() for _ in range(5)]
torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 5
print(f"FF B={B} x {n} atoms (N={raw.num_nodes} E={raw.num_edges} T={raw.num_triplets}): eager {te*1e3:.1f} ms/step ({B/te:.1f} graphs/s), "
      f"hipGraph {tg*1e3:.1f} ms/step ({B/tg:.1f} graphs/s)")
print("loss eager", [round(float(x), 5) for x in le]); print("loss graph", [round(float(x), 5) for x in lg])
print("peak memory GB", torch.cuda.max_memory_allocated() / 1e9)

# ---- inference (energies + forces + stresses, no training): composed path (train() forward) vs fused path (eval())
def t_inf(mod, n=5):
    for _ in range(2): mod(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = mod(batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out
tc, oc = t_inf(m.train())
tf, of = t_inf(m.eval())
err = float((oc["grad"].detach() - of["grad"]).abs().max() / oc["grad"].detach().abs().max())
print(f"FF inference (E, F, stress): composed {tc*1e3:.1f} ms ({raw.num_nodes/tc/1e3:.1f} k atoms/s), fused eval() {tf*1e3:.1f} ms "
      f"({raw.num_nodes/tf/1e3:.1f} k atoms/s), max rel force difference {err:.2e}")
