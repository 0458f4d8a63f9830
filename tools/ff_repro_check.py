"""Is a force-training step of ALIGNNAtomWise bit-reproducible run to run?  usage: ff_repro_check.py B [c|ops] [lanes 0|auto]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch, cmodel, ops
from alignn_amd.synthetic import make_batch
DEV = "cuda"
l1 = torch.nn.functional.l1_loss
B = int(sys.argv[1]); path = sys.argv[2]; ops._LANE["enabled"] = sys.argv[3]
FF = os.environ.get("FF", "1") == "1"
cmodel.ENABLED = path == "c"
torch.manual_seed(6)
m = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=256, atom_input_features=92,
                                        calculate_gradient=FF, stresswise_weight=0.05 if FF else 0.0)).to(DEV).train()
raw = make_batch(B, 60, seed0=11)
batch = GraphBatch.from_raw(raw, device=DEV)
g = torch.Generator().manual_seed(6)
te, tf, ts = (torch.randn(raw.batch_size, generator=g).to(DEV), torch.randn(raw.num_nodes, 3, generator=g).to(DEV), torch.randn(raw.batch_size, 3, 3, generator=g).to(DEV))
ref = None
for it in range(4):
    for p in m.parameters():
        p.grad = None
    o = m(batch)
    if FF:
        (l1(o["out"], te) + l1(o["grad"], tf) + l1(o["stresses"], ts)).backward()
    else:
        l1(o["out"], te).backward()
    torch.cuda.synchronize()
    cur = {"out": o["out"].detach().clone()}
    if FF:
        cur.update({"F": o["grad"].detach().clone(), "S": o["stresses"].detach().clone()})
    cur.update({"g." + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
    else:
        bad = [(k, float((cur[k].double() - ref[k].double()).abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
        print(f"B={B} FF={int(FF)} path={path} lanes={sys.argv[3]} run {it}: {len(bad)} tensors differ from run 0", bad[:3])
