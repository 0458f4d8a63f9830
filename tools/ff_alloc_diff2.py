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
    if rep == 0 and bad and os.environ.get("FF_DIFF_DETAIL"):
        i, nm, cnt, mx = bad[0]
        off, nb = la[i]
        width = 1024 if nm.endswith("x1024") else 256
        a, b = A0[off:off + nb].view(torch.float32).view(-1, width), A1[off:off + nb].view(torch.float32).view(-1, width)
        ne = a != b
        rows = ne.any(1).nonzero().flatten()
        print(f"   first differing buffer {i} ({nm}): {rows.numel()} rows differ; first rows {rows[:16].tolist()}; row % 4: {sorted(set((rows % 4).tolist()))}; row % 64: {sorted(set((rows % 64).tolist()))[:20]}")
        for r in rows[:4].tolist():
            cs = ne[r].nonzero().flatten()
            print("      row", r, "cols", cs.tolist()[:20], "n", cs.numel())
            print("      good", [f"{v:.6e}" for v in a[r, cs[:6]].tolist()])
            print("      bad ", [f"{v:.6e}" for v in b[r, cs[:6]].tolist()])
            # does the bad value equal the good value of ANOTHER row (same column)?
            for c_ in cs[:3].tolist():
                hit = (a[:, c_] == b[r, c_]).nonzero().flatten()
                hit2 = (a[r, :] == b[r, c_]).nonzero().flatten()
                print(f"        col {c_}: bad value found in good buffer at rows {hit[:5].tolist()} (same column) / cols {hit2[:5].tolist()} (same row)")
    if os.environ.get("FF_DIFF_DETAIL") == "2" and len(la) > 130:
        def buf(A, i, w):
            off, nb = la[i]
            return A[off:off + nb].view(torch.float32).view(-1, w)
        for tag, A in (("reference run", A0), ("this run", A1)):
            o = buf(A, 127, 1024)[:, 768:]
            s0, hh, g1, g0 = buf(A, 24, 256), buf(A, 25, 256), buf(A, 129, 256), buf(A, 130, 256)
            g1_ref = o / (s0 + 1e-6)
            g0_ref = -g1_ref * hh
            bad1 = ((g1 - g1_ref).abs() > 1e-5 * g1_ref.abs() + 1e-12)
            bad0 = ((g0 - g0_ref).abs() > 1e-5 * g0_ref.abs() + 1e-12)
            print(f"   [{tag}] GS1 elements off their recomputation o / (s0 + eps): {int(bad1.sum())}; GS0 off -GS1_ref h: {int(bad0.sum())}; GS0 off -GS1_stored h: {int(((g0 + g1 * hh).abs() > 1e-5 * g0.abs() + 1e-12).sum())}")
            idx = bad1.nonzero()[:6]
            for r, c_ in idx.tolist():
                print(f"      row {r} col {c_}: GS1 stored {float(g1[r, c_]):.6e}  o/(s0+eps) {float(g1_ref[r, c_]):.6e}  -g1_ref*h {float(g0_ref[r, c_]):.6e}  GS0 stored {float(g0[r, c_]):.6e}  o {float(o[r, c_]):.6e} s0 {float(s0[r, c_]):.6e} h {float(hh[r, c_]):.6e}"
                      f"   neighbours .y {float(g1[r, c_ - 1]):.4e}/{float(g1_ref[r, c_ - 1]):.4e} .w {float(g1[r, c_ + 1]):.4e}/{float(g1_ref[r, c_ + 1]):.4e}")
