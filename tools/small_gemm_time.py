"""Atom-row projections of a bond-graph convolution (N = 3 840 rows): how long do the exact-fp32 MFMA products take, and
what would a 4-way split of the K = 1024 input gradient buy?  usage: python tools/small_gemm_time.py"""
import sys, torch
sys.path.insert(0, ".")
from alignn_amd import ops
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 3840
x = torch.randn(M, 256, device="cuda"); wcat = torch.randn(1024, 256, device="cuda") / 16; b = torch.randn(1024, device="cuda")
gp = torch.randn(M, 1024, device="cuda"); res = torch.randn(M, 256, device="cuda")
print("forward  P = x Wcat^T        [3840x256 -> 1024]:", round(t(lambda: ops.project(x, wcat, b)), 1), "us")
print("dgrad    gx = GP Wcat + res  [3840x1024 -> 256]:", round(t(lambda: ops._dgrad(gp, wcat, res)), 1), "us  (split reduction:", ops.NN_SPLIT, ")")
w4 = [wcat[i * 256:(i + 1) * 256].contiguous() for i in range(4)]
g4 = [gp[:, i * 256:(i + 1) * 256] for i in range(4)]
print("one quarter  GP_i W_i        [3840x256 -> 256] :", round(t(lambda: ops._dgrad(g4[0], w4[0], None)), 1), "us  (a 4-way split runs four of these side by side + a 4-slab sum)")
streams = [torch.cuda.Stream() for _ in range(4)]
def split():
    cur = torch.cuda.current_stream()
    outs = []
    for i, s_ in enumerate(streams):
        s_.wait_stream(cur)
        with torch.cuda.stream(s_):
            outs.append(ops._dgrad(g4[i], w4[i], None))
    for s_ in streams:
        cur.wait_stream(s_)
    return outs
print("four quarters on four streams                    :", round(t(split), 1), "us")
