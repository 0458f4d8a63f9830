"""Phase timing inside the f16x3 NT kernel (T x 256 x 256): a -DX6_TRACE=1 build stamps s_memtime in wave 0 of every
workgroup at entry, first stage landed, k-step 8, k-loop end, last store issued, stores drained; this prints the
median phase lengths (shader cycles) and the gap between consecutive workgroups in the same CU wave slot.
usage: python tools/x6_trace.py [variant flags...]   (builds tools/_x6_trace.so when flags are given; run on the GPU without)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_x6_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DX6_TRACE=1",
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
f = lib.alignn_gemm_nt_f16x3
f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
for _ in range(3):
    assert f(a.data_ptr(), K, am.data_ptr(), img.data_ptr(), wm.data_ptr(), None, None, 0, c.data_ptr(), N, M, N, K, st) == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); f(a.data_ptr(), K, am.data_ptr(), img.data_ptr(), wm.data_ptr(), None, None, 0, c.data_ptr(), N, M, N, K, st); e.record()
torch.cuda.synchronize()
print(f"kernel {s.elapsed_time(e)*1e3:.1f} us")
buf = np.zeros(8192 * 8, dtype=np.uint64)
rd = lib.alignn_x6_trace_read; rd.argtypes = [C.c_void_p, C.c_size_t]
assert rd(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 8)[: (M + 127) // 128].astype(np.int64)
names = ["entry->stage0 landed", "stage0->k-step 8", "k-step 8->k-loop end", "k-loop end->last store issued", "last store->drained"]
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"{n:32s} median {np.median(d):9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} cycles")
tot = t[:, 5] - t[:, 0]
print(f"{'workgroup lifetime':32s} median {np.median(tot):9.0f}  p10 {np.percentile(tot,10):9.0f}  p90 {np.percentile(tot,90):9.0f} cycles")
hw = t[:, 7]
xcc = hw >> 32; idv = hw & 0xffffffff
key = (xcc << 20) | (idv & 0xffff)  # xcc, se/sh/cu, simd, wave slot of wave 0
gaps = []
span = {}
for k in np.unique(key):
    rows = t[key == k]
    rows = rows[np.argsort(rows[:, 0])]
    gaps += list(rows[1:, 0] - rows[:-1, 5])
    span[k] = (rows[0, 0], rows[-1, 5], len(rows))
gaps = np.array(gaps)
print(f"slots used {len(span)}, workgroups per slot median {np.median([v[2] for v in span.values()]):.0f}")
print(f"{'successor entry - exit (slot)':32s} median {np.median(gaps):9.0f}  p10 {np.percentile(gaps,10):9.0f}  p90 {np.percentile(gaps,90):9.0f} cycles")
first = np.array([v[0] for v in span.values()]); last = np.array([v[1] for v in span.values()])
# counters of different XCDs are not synchronised: spans per XCD
for x in np.unique(xcc):
    sel = [k for k in span if (k >> 20) == x]
    f0 = min(span[k][0] for k in sel); l1 = max(span[k][1] for k in sel)
    print(f"xcc {x}: {len(sel)} slots, first entry -> last exit {l1 - f0} cycles")
