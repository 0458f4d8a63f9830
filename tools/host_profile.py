"""Where does the HOST time of an eager training step go?  cProfile over 5 steps (GPU work is asynchronous)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch
from alignn_amd.synthetic import make_batch
dev = "cuda"
B = int(os.environ.get("B", "64"))
batch = GraphBatch.from_raw(make_batch(B, 60), device=dev)
torch.manual_seed(0)
model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
from alignn_amd.optim import FlatAdamW, group_decay
opt = FlatAdamW(group_decay(model), lr=1e-3, weight_decay=1e-2, module=model)
torch.autograd.set_multithreading_enabled(False)  # backward nodes in this thread: cProfile sees them
target = torch.randn(B, device=dev)
def step():
    opt.zero_grad()
    torch.nn.functional.l1_loss(model(batch), target).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(32); st.sort_stats("cumulative").print_stats(45)
