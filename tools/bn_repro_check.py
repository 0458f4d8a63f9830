"""Is a training step of the BatchNorm model (the headline workload, four streams) bit-reproducible run to run?  usage: bn_repro_check.py B"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, cmodel
from alignn_amd.synthetic import make_batch
DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = ALIGNN(ALIGNNConfig(name="alignn")).to(DEV).train()
raw = make_batch(B, 60)
batch = GraphBatch.from_raw(raw, device=DEV)
target = torch.randn(B, generator=torch.Generator().manual_seed(1)).to(DEV)
ref = None
for it in range(6):
    for p in m.parameters():
        p.grad = None
    sd0 = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    pred = m(batch)
    torch.nn.functional.l1_loss(pred, target).backward()
    torch.cuda.synchronize()
    cur = {"pred": pred.detach().clone()}
    cur.update({"g." + k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    m.load_state_dict(sd0, strict=False)  # (same running statistics going into every run)
    if ref is None:
        ref = cur
    else:
        bad = [(k, float((cur[k].double() - ref[k].double()).abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
        print(f"BatchNorm model B={B} T={raw.num_triplets} run {it}: {len(bad)} tensors differ from run 0", bad[:3])
print("C calls:", cmodel.STATS["fwd"], cmodel.STATS["bwd"])
