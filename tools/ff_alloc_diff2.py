"""First differing workspace buffer between two eagerly launched alignn_ff_eval calls of the same model (ALIGNN_AMD_DEBUG_ALLOCS=1)."""
import ctypes as C, os, sys, torch
os.environ["ALIGNN_AMD_DEBUG_ALLOCS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, GraphBatch, cmodel, _lib
from alignn_amd.synthetic import make_batch
DEV = "cuda"
B = int(sys.argv[1])
torch.manual_seed(6)
m = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=2, gcn_layers=2, hidden_features=256, atom_input_features=92,
                                        calculate_gradient=True, stresswise_weight=0.05)).to(DEV).eval()
def allocs():
    buf = (C.c_size_t * 40000)()
    n = _lib.load().alignn_debug_allocs(buf, 20000)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n, 20000))]
raw = make_batch(B, 60, seed0=11)
batch = GraphBatch.from_raw(raw, device=DEV)
E_, T_, N_ = raw.num_edges, raw.num_triplets, raw.num_nodes
def name(nb):
    f = nb / 4
    for lab, rows in (("T", T_), ("E", E_), ("N", N_)):
        for w in (1024, 256, 64, 40, 80, 3, 2, 1):
            if abs(f - rows * w) < 64:
                return f"{lab}x{w}"
    return str(nb)
m(batch); torch.cuda.synchronize(); allocs()
m(batch); torch.cuda.synchronize(); la = allocs()
A0 = cmodel.binding_of(m).arena.clone()
print("N E T", N_, E_, T_, "allocs", len(la))
print(" ".join(f"{i}:{name(nb)}" for i, (off, nb) in enumerate(la)))
for rep in range(4):
    m(batch); torch.cuda.synchronize(); allocs()
    A1 = cmodel.binding_of(m).arena
    bad = []
    for i, (off, nb) in enumerate(la):
        a, b = A0[off:off + nb].view(torch.float32), A1[off:off + nb].view(torch.float32)
        if not torch.equal(a, b):
            ne = a != b
            bad.append((i, name(nb), int(ne.sum()), float((a.double() - b.double()).abs().nan_to_num(0).max())))
    print(f"run {rep}: {len(bad)} differing buffers:", bad[:10])
