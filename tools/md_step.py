"""One ALIGNN-FF "MD step" for a 200-atom cell with everything on the device: neighbour search + canonical graph +
line graph (alignn_amd.neighbors) -> energies, forces, stresses (fused eval path).  Compared with building the graph
on the host with the numpy restatement of the reference's builder."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch, neighbors
from alignn_amd.synthetic import make_crystal, _one, batch_raw

dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lat, frac, _ = make_crystal(n, 1234)
feats = torch.randn(n, 92, device=dev)
lat_d, frac_d = torch.from_numpy(lat).to(dev), torch.from_numpy(frac).to(dev)
torch.manual_seed(0)
model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=4, gcn_layers=4, hidden_features=256,
                                             atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05)).to(dev).eval()

def timeit(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k, out

t_knn_hip, _ = timeit(lambda: neighbors.knn_multigraph_batch_hip([lat_d], [frac_d]))
t_knn_torch, _ = timeit(lambda: neighbors.knn_multigraph_batch([lat_d], [frac_d]))
print(f"neighbour list alone: csrc/knn.hip {t_knn_hip*1e3:.2f} ms, torch tensor ops {t_knn_torch*1e3:.2f} ms")
t_build, batch = timeit(lambda: neighbors.crystal_batch([lat_d], [frac_d], atom_features=[feats]))
t_model, res = timeit(lambda: model(batch))
t_step, _ = timeit(lambda: model(neighbors.crystal_batch([lat_d], [frac_d + 1e-4 * torch.randn_like(frac_d)], atom_features=[feats])))
from alignn_amd.md import GraphedForceField
ff = GraphedForceField(model)
def md_step_graphed():
    return ff(neighbors.crystal_batch([lat_d], [frac_d + 1e-6 * torch.randn_like(frac_d)], atom_features=[feats]))
t_gstep, _ = timeit(md_step_graphed, 20)
t_replay, _ = timeit(lambda: ff(batch), 20)
print(f"hipGraph per batch shape (alignn_amd/md.py): model replay {t_replay*1e3:.2f} ms, full MD step {t_gstep*1e3:.2f} ms "
      f"({ff.stats['replayed']} replays, {ff.stats['captured']} captures)")
t0 = time.perf_counter()
for _ in range(3):
    raw = batch_raw([_one(n, 1234, "crystal", 92)])
    b2 = GraphBatch.from_raw(raw, device=dev)
torch.cuda.synchronize(); t_host = (time.perf_counter() - t0) / 3
print(f"{n}-atom cell: E={batch.g.n_edges} T={batch.lg.n_edges} | device graph build {t_build*1e3:.1f} ms, model (E,F,stress) {t_model*1e3:.1f} ms, "
      f"full MD step {t_step*1e3:.1f} ms | host (numpy) graph build + staging {t_host*1e3:.0f} ms")
