"""Kernel-level timing of the line-graph reverse passes with the edge LayerNorm inside (csrc/convln.hip) against the separate
kernels they replace, on the line graph of a BASELINE configs[3] batch (16 x 200 atoms), and of the launch variants
(ALIGNN_AMD_LN_REV = "<value><dual>": sources per wave and pass / rows per sub-batch / early loads).

    python tools/ln_rev_time.py [batch] [atoms]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import _lib  # noqa: E402
from alignn_amd.graph import GraphBatch  # noqa: E402
from alignn_amd.ops import ptr  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
A = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
lib = _lib.load()
batch = GraphBatch.from_raw(make_batch(B, A, seed0=99), device=dev)
lg = batch.lg
n, m, H = lg.n_nodes, lg.n_edges, 256
groups = lg.grp_seg_ptr.numel() - 1
print(f"line graph: n={n} m={m} groups={groups} dense_max_src={lg.dense_max_src}")
g = torch.Generator(device=dev).manual_seed(1)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
M, Mt, GY, GYt = R(m, H), R(m, H), R(m, H), R(m, H)
P, Pt = R(n, 4 * H), R(n, 4 * H)
q1, q0, q1t, q0t = R(n, H), R(n, H), R(n, H), R(n, H)
gamma, beta = 1 + 0.1 * R(H), 0.1 * R(H)
mean = M.mean(1)
rstd = 1.0 / torch.sqrt(M.var(1, unbiased=False) + 1e-5)
e_stat = torch.stack([mean, rstd], 1).contiguous()
GM, GMt, GL, GLt = (torch.empty(m, H, device=dev) for _ in range(4))
GP, GPt = torch.empty(n, 4 * H, device=dev), torch.empty(n, 4 * H, device=dev)
gb = torch.empty(groups, H, device=dev)
lnp = torch.empty(groups, 2, H, device=dev)
am = torch.zeros(8, device=dev)
st = torch.cuda.current_stream().cuda_stream
slabs = lib.alignn_dual_slabs(m)
part = torch.empty(slabs, 2, H, device=dev)
vslabs = lib.alignn_ln_slabs(m)
vpart = torch.empty(vslabs, 2, H, device=dev)


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def dual_sep():
    lib.alignn_ln_silu_dual_bwd(ptr(GY), ptr(GYt), H, ptr(M), ptr(Mt), H, ptr(gamma), ptr(beta), ptr(e_stat), ptr(GL), ptr(GLt), H,
                                ptr(part), m, H, ptr(am), st)
    lib.alignn_egc_dual_bwd_lg_dense(ptr(GL), ptr(GLt), ptr(M), ptr(Mt), ptr(P), ptr(Pt), ptr(q1), ptr(q0), ptr(q1t), ptr(q0t), m,
                                     ptr(lg.grp_seg_ptr), ptr(lg.grp_src_ptr), groups, ptr(lg.seg_ptr), ptr(lg.seg_node), H, ptr(GM),
                                     ptr(GMt), ptr(GP), ptr(GPt), ptr(gb), ptr(am[2:]), ptr(am[4:]), st)


def dual_fused():
    rc = lib.alignn_egc_dual_bwd_lg_dense_ln(ptr(GY), ptr(GYt), ptr(M), ptr(Mt), ptr(P), ptr(Pt), ptr(q1), ptr(q0), ptr(q1t), ptr(q0t),
                                             ptr(gamma), ptr(beta), ptr(e_stat), m, ptr(lg.grp_seg_ptr), ptr(lg.grp_src_ptr), groups,
                                             ptr(lg.seg_ptr), ptr(lg.seg_node), H, ptr(GM), ptr(GMt), ptr(GP), ptr(GPt), ptr(gb),
                                             ptr(lnp), ptr(am[2:]), ptr(am[4:]), st)
    assert rc == 0, rc


def val_sep():
    lib.alignn_ln_silu_bwd(ptr(GY), H, ptr(M), H, ptr(gamma), ptr(beta), ptr(e_stat), ptr(GL), H, ptr(vpart), m, H, None, st)
    lib.alignn_egc_bwd_lg_dense(ptr(GL), ptr(M), ptr(P), ptr(q1), ptr(q0), None, None, 0, m, ptr(lg.grp_seg_ptr), ptr(lg.grp_src_ptr),
                                groups, lg.dense_max_src, ptr(lg.seg_ptr), ptr(lg.seg_node), H, ptr(GM), ptr(GP), ptr(gb), ptr(am[2:]),
                                ptr(am[4:]), st)


def val_fused():
    rc = lib.alignn_egc_bwd_lg_dense_ln(ptr(GY), ptr(M), ptr(P), ptr(q1), ptr(q0), ptr(gamma), ptr(beta), ptr(e_stat), m,
                                        ptr(lg.grp_seg_ptr), ptr(lg.grp_src_ptr), groups, lg.dense_max_src, ptr(lg.seg_ptr),
                                        ptr(lg.seg_node), H, ptr(GM), ptr(GP), ptr(gb), ptr(lnp), ptr(am[2:]), ptr(am[4:]), st)
    assert rc == 0, rc


row_bytes = m * H * 4
print(f"value reverse, separate kernels (ln_silu_bwd + egc_bwd_lg_dense<2>): {timed(val_sep):7.1f} us  (6 row passes)")
ref = None
for v, name in ((0, "4 sources/wave, 1 row"), (1, "4 sources/wave, 2 rows"), (2, "2 sources/wave, 2 rows")):
    os.environ["ALIGNN_AMD_LN_REV"] = f"{v}0"
    t = timed(val_fused)
    gm = GM.clone()
    if ref is None:
        val_sep()
        torch.cuda.synchronize()
        print("   max |fused - separate| / max|.|:", float((gm - GM).abs().max() / GM.abs().max()))
    print(f"value reverse, LayerNorm inside, {name}: {t:7.1f} us  = {3 * row_bytes / t / 1e6:5.2f} TB/s over 3 row passes")
    ref = gm
print(f"dual reverse, separate kernels (ln_silu_dual_bwd + egc_dual_bwd_lg_dense): {timed(dual_sep):7.1f} us  (12 row passes)")
dual_sep()
torch.cuda.synchronize()
gm_s, gmt_s = GM.clone(), GMt.clone()
for v, name in ((0, "4 sources/wave, 1 row, loads together"), (1, "4 sources/wave, 2 rows"), (2, "2 sources/wave, 1 row")):
    os.environ["ALIGNN_AMD_LN_REV"] = f"0{v}"
    t = timed(dual_fused)
    e1 = float((GM - gm_s).abs().max() / gm_s.abs().max())
    e2 = float((GMt - gmt_s).abs().max() / gmt_s.abs().max())
    print(f"dual reverse, LayerNorm inside, {name}: {t:7.1f} us  = {6 * row_bytes / t / 1e6:5.2f} TB/s over 6 row passes  (vs separate: {e1:.1e} {e2:.1e})")
