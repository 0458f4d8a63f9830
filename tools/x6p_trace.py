"""Phase timing inside the persistent f16x3 NT kernel (second tile of every workgroup): -DX6_TRACE=1 build, s_memtime
stamps of wave 0.  usage: python tools/x6p_trace.py build | python tools/x6p_trace.py [gather|bnred]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_x6p_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DX6_TRACE=" + os.environ.get("X6_TRACE", "1"),
                    os.path.join(ROOT, "alignn_amd", "csrc", "gemm_x6.hip"), "-o", SO] + sys.argv[2:], check=True)
    sys.exit(0)
import numpy as np, torch
M, N, K = 676200, 256, 256
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
am = a.abs().max().reshape(1); wm = w.abs().max().reshape(1)
lib = C.CDLL(SO)
nb = lib.alignn_split_f16x2_bytes; nb.restype = C.c_size_t; nb.argtypes = [C.c_int, C.c_int]
img = torch.empty(nb(N, K), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
sp = lib.alignn_split_f16x2; sp.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
assert sp(w.data_ptr(), K, N, K, 0, wm.data_ptr(), img.data_ptr(), st) == 0
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "plain":
    f = lib.alignn_gemm_nt_f16x3
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    call = lambda: f(a.data_ptr(), K, am.data_ptr(), img.data_ptr(), wm.data_ptr(), None, None, 0, c.data_ptr(), N, M, N, K, st)
elif mode == "bnred":
    res = torch.randn(M, N, device="cuda"); xn = torch.randn(M, N, device="cuda")
    stat = torch.stack([xn.mean(0), torch.rsqrt(xn.var(0, unbiased=False) + 1e-5), torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")]).contiguous()
    part = torch.empty(M // 64 + 3, 2, N, device="cuda")
    f = lib.alignn_gemm_nt_f16x3_bnred
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    call = lambda: f(a.data_ptr(), K, am.data_ptr(), img.data_ptr(), wm.data_ptr(), None, res.data_ptr(), N, c.data_ptr(), N, M, N, K, xn.data_ptr(), N, stat.data_ptr(), part.data_ptr(), st)
else:
    E = 50712
    P = torch.randn(E, 4 * N, device="cuda")
    src = torch.randint(0, E, (M,)).sort().values.int().cuda(); dst = torch.randint(0, E, (M,)).int().cuda()
    part = torch.empty(M // 64 + 3, 2, N, device="cuda")
    f = lib.alignn_gemm_nt_f16x3_gather
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    call = lambda: f(a.data_ptr(), K, am.data_ptr(), img.data_ptr(), wm.data_ptr(), None, c.data_ptr(), N, M, N, K, P.data_ptr(), 4 * N, src.data_ptr(), dst.data_ptr(), part.data_ptr(), st)
for _ in range(20):
    assert call() == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); call(); e.record()
torch.cuda.synchronize()
print(f"{mode}: kernel {s.elapsed_time(e)*1e3:.1f} us")
buf = np.zeros(8192 * 8, dtype=np.uint64)
rd = lib.alignn_x6_trace_read; rd.argtypes = [C.c_void_p, C.c_size_t]
assert rd(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 8)[:512].astype(np.int64)
names = ["k-loop (16 steps)", "wait for the 2 prefetched stages", "barrier before the patches", "epilogue (to last store issued)",
         "epilogue end -> first barrier of next tile", "steps 0,1 of the next tile"]
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"{n:42s} median {np.median(d):9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} cycles")
tot = t[:, 5] - t[:, 0]
print(f"{'tile period':42s} median {np.median(tot):9.0f}  p10 {np.percentile(tot,10):9.0f}  p90 {np.percentile(tot,90):9.0f} cycles")
ph = buf.reshape(8192, 8)[4096:4096 + 512, :5].astype(np.int64)
if ph.any():
    for i, n in enumerate(["vmcnt wait", "barrier", "DMA issue", "LDS reads + high slices (to lgkmcnt 0)", "24 MFMA + low slices (issue)"]):
        print(f"  k-loop share: {n:42s} median {np.median(ph[:, i]):9.0f}  ({100 * np.median(ph[:, i]) / np.median(ph.sum(1)):5.1f} %)")
