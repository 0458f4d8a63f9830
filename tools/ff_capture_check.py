"""Force training through the whole-model C calls: eagerly launched vs captured into a hipGraph and replayed, output by output
(debugging aid for tests/test_gpu_cmodel_ff.py::test_one_stream_and_helper_streams_give_the_same_bits_and_capture_replays)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch, cmodel, ops
from alignn_amd.synthetic import make_batch

DEV = "cuda"
l1 = torch.nn.functional.l1_loss


LG_ON_FLY = os.environ.get("LG_ON_FLY", "1") == "1"


def mk():
    torch.manual_seed(6)
    return ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=256,
                                               atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05,
                                               lg_on_fly=LG_ON_FLY)).to(DEV).train()


raw = make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 16, 60, seed0=11)
batch = GraphBatch.from_raw(raw, device=DEV)
g = torch.Generator().manual_seed(6)
te, tf, ts = (torch.randn(raw.batch_size, generator=g).to(DEV), torch.randn(raw.num_nodes, 3, generator=g).to(DEV),
              torch.randn(raw.batch_size, 3, 3, generator=g).to(DEV))


def step(m):
    for p in m.parameters():
        p.grad = None
    o = m(batch)
    loss = l1(o["out"], te) + l1(o["grad"], tf) + l1(o["stresses"], ts)
    loss.backward()
    return o, loss


def snap(m, o, loss):
    torch.cuda.synchronize()
    d = {"out": o["out"].detach().clone(), "F": o["grad"].detach().clone(), "S": o["stresses"].detach().clone(), "loss": loss.detach().clone()}
    d.update({"g." + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    return d


def diff(a, b, tag):
    bad = [(k, float((a[k].double() - b[k].double()).abs().max()), float(b[k].double().abs().max())) for k in a if not torch.equal(a[k], b[k])]
    print(tag, "differing tensors:", len(bad), bad[:6])


for lanes in ("auto", "0"):
    ops._LANE["enabled"] = lanes
    m0 = mk()
    e1 = snap(m0, *step(m0))
    e2 = snap(m0, *step(m0))
    diff(e1, e2, f"[lanes {lanes}] eager twice")
    m = mk()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        w = snap(m, *step(m))
    torch.cuda.current_stream().wait_stream(s)
    diff(e1, w, f"[lanes {lanes}] eager on a side stream")
    for p in m.parameters():
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        o, loss = step(m)
    for i in range(2):
        graph.replay()
        diff(e1, snap(m, o, loss), f"[lanes {lanes}] replay {i}")

# ---- forward only (the force evaluation alone) captured and replayed
ops._LANE["enabled"] = "auto"
m0 = mk()
with torch.enable_grad():
    o = m0(batch)
ref = {"out": o["out"].detach().clone(), "F": o["grad"].detach().clone(), "S": o["stresses"].detach().clone()}
m = mk()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    o = m(batch)
for i in range(3):
    graph.replay()
    torch.cuda.synchronize()
    diff(ref, {"out": o["out"].detach(), "F": o["grad"].detach(), "S": o["stresses"].detach()}, f"[forward only] replay {i}")
m.eval()
ref2 = m(batch)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    o = m(batch)
for i in range(3):
    graph.replay()
    torch.cuda.synchronize()
    diff({"out": ref2["out"], "F": ref2["grad"], "S": ref2["stresses"]}, {"out": o["out"], "F": o["grad"], "S": o["stresses"]}, f"[eval mode, values only] replay {i}")
