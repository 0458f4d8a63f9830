"""Capture a training step as a hipGraph and compare with eager: same loss trajectory, steps/s at small batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch
from alignn_amd.graphed import GraphedTrainStep
from alignn_amd.synthetic import make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
raw = make_batch(B, 60)
batch = GraphBatch.from_raw(raw, device=dev)
target = torch.randn(B, generator=torch.Generator().manual_seed(1)).to(dev)

def fresh():
    torch.manual_seed(0)
    m = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
    o = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)
    return m, o

m, o = fresh()
losses_e = []
def eager():
    o.zero_grad(set_to_none=True)
    l = torch.nn.functional.l1_loss(m(batch), target); l.backward(); o.step(); return l
for _ in range(3): eager()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): losses_e.append(eager())
torch.cuda.synchronize(); te = (time.perf_counter() - t) / 20
m2, o2 = fresh()
g = GraphedTrainStep(m2, batch, target, o2, warmup=3)
losses_g = []
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): losses_g.append(g().clone())
torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 20
le = torch.stack(losses_e).cpu(); lg = torch.stack(losses_g).cpu()
print(f"B={B}: eager {te*1e3:.2f} ms/step ({B/te:.0f} graphs/s), hipGraph {tg*1e3:.2f} ms/step ({B/tg:.0f} graphs/s)")
print("loss eager ", [round(float(x), 5) for x in le[:5]])
print("loss graph ", [round(float(x), 5) for x in lg[:5]])
print("max |diff| over 20 steps", float((le - lg).abs().max()))
