"""Which workspace buffer of alignn_ff_eval differs first between an eagerly launched call and a hipGraph replay?
(ALIGNN_AMD_DEBUG_ALLOCS=1; debugging aid for the lane-T race of round 5.)"""
import ctypes as C, os, sys, torch
os.environ["ALIGNN_AMD_DEBUG_ALLOCS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch, cmodel, _lib
from alignn_amd.synthetic import make_batch

DEV = "cuda"


def mk():
    torch.manual_seed(6)
    return ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=256,
                                               atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05,
                                               lg_on_fly=False)).to(DEV).eval()


def allocs():
    buf = (C.c_size_t * 40000)()
    n = _lib.load().alignn_debug_allocs(buf, 20000)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n, 20000))]


raw = make_batch(16, 60, seed0=11)
batch = GraphBatch.from_raw(raw, device=DEV)
m0 = mk()
m0(batch); torch.cuda.synchronize(); allocs()
m0(batch); torch.cuda.synchronize()
la = allocs()
A0 = cmodel.binding_of(m0).arena.clone()
m = mk()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(batch)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize(); allocs()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    o = m(batch)
lb = allocs()
print("allocations:", len(la), len(lb), "same layout:", la == lb, "N E T", raw.num_nodes, raw.num_edges, raw.num_triplets)
E_, T_, N_ = raw.num_edges, raw.num_triplets, raw.num_nodes
def name(nb):
    f = nb / 4
    for lab, rows in (("T", T_), ("E", E_), ("N", N_)):
        for w in (1024, 256, 64, 40, 80, 3, 2, 1):
            if abs(f - rows * w) < 64:
                return f"{lab}x{w}"
    return str(nb)
print(" ".join(f"{i}:{name(nb)}" for i, (off, nb) in enumerate(la)))
for rep in range(3):
    g.replay(); torch.cuda.synchronize()
    A1 = cmodel.binding_of(m).arena
    bad = []
    for i, (off, nb) in enumerate(la):
        a, b = A0[off:off + nb].view(torch.float32), A1[off:off + nb].view(torch.float32)
        if not torch.equal(a, b):
            d = (a.double() - b.double()).abs()
            d = torch.where(torch.isfinite(d), d, torch.zeros_like(d))
            bad.append((i, off, nb, float(d.max()), int((a != b).sum())))
    print(f"replay {rep}: {len(bad)} differing buffers; first:", bad[:4])
    for idx, width in ((113, 1024), (115, 256)):
        off, nb = la[idx]
        a, b = A0[off:off + nb].view(torch.float32).view(-1, width), A1[off:off + nb].view(torch.float32).view(-1, width)
        ne = (a != b)
        rows = ne.any(1).nonzero().flatten()
        cols = ne.any(0).nonzero().flatten()
        print(f"   buffer {idx}: {rows.numel()} rows differ: {rows[:12].tolist()} ... {rows[-4:].tolist()}; columns {int(cols.min())}..{int(cols.max())}")
        if idx == 115 and rep == 0:
            for r in rows[:3].tolist():
                cs = ne[r].nonzero().flatten()
                print("      row", r, "cols", cs[:6].tolist(), "...", cs[-3:].tolist(), "n", cs.numel())
                print("      eager ", [f"{v:.6e}" for v in a[r, cs[:5]].tolist()])
                print("      replay", [f"{v:.6e}" for v in b[r, cs[:5]].tolist()])
                # is the replayed piece some OTHER row of the eager result?
                seg = b[r, 192:256]
                hit = (a[:, 192:256] == seg).all(1).nonzero().flatten()
                print("      replayed cols 192..255 equal eager row(s):", hit[:5].tolist())
