"""bench.py's graph capture with a live RCCL process group (world size 1 on one GPU): the communicator's watchdog
thread is running while forward+loss+backward are captured, and an all-reduce runs eagerly between replays."""
import os, sys, subprocess
env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
           HSA_ENABLE_IPC_MODE_LEGACY="0")
code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops
from alignn_amd.synthetic import make_batch
batch = GraphBatch.from_raw(make_batch(8, 20), device="cuda")
torch.manual_seed(0)
model = ALIGNN(ALIGNNConfig(name="alignn", alignn_layers=2, gcn_layers=2)).cuda().train()
target = torch.randn(8, device="cuda")
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
for _ in range(2):
    opt.zero_grad(set_to_none=True); torch.nn.functional.l1_loss(model(batch), target).backward(); opt.step()
torch.cuda.synchronize()
for p in model.parameters(): p.grad = None
ops.reset_amax_arena()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    loss = torch.nn.functional.l1_loss(model(batch), target); loss.backward()
ops.reset_amax_arena()
grads = [p.grad for p in model.parameters()]
for i in range(5):
    g.replay()
    flat = torch.cat([x.reshape(-1) for x in grads if x is not None]); dist.all_reduce(flat)
    for p, x in zip(model.parameters(), grads): p.grad = x
    opt.step()
torch.cuda.synchronize()
print("capture + replay under a live RCCL process group: ok, loss", float(loss))
dist.destroy_process_group()
'''
sys.exit(subprocess.run([sys.executable, "-c", code], env=env).returncode)
