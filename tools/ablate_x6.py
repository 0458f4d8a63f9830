"""Ablation of the bf16x6 GEMM kernel: builds variants with parts stubbed out and times them
(T x 256 x 256).  Results are only meaningful as time differences; stubbed variants compute garbage."""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "alignn_amd", "csrc", "gemm_x6.hip")
OUT = os.path.join(ROOT, "gpurun_out")
VARIANTS = {"warm": [], "base": [], "big": ["-DX6_WM=2", "-DX6_RM=4"], "big_noloads": ["-DX6_WM=2", "-DX6_RM=4", "-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1"], "mid": ["-DX6_WM=2", "-DX6_RM=2"],  "noslice": ["-DX6_ABL_NOSLICE=1"], "nobload": ["-DX6_ABL_NOBLOAD=1"], "noaload": ["-DX6_ABL_NOALOAD=1"],
            "onemfma": ["-DX6_ABL_ONEMFMA=1"], "noloads": ["-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1"],
            "noloads_noslice": ["-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSLICE=1"]}
def build():
    os.makedirs(OUT, exist_ok=True)
    for k, fl in VARIANTS.items():
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", SRC, "-o", os.path.join(ROOT, "tools", f"_x6_{k}.so")] + fl, check=True)
def run():
    M, N, K = 676200, 256, 256
    a = torch.randn(M, K, device="cuda"); w = torch.randn(3, N, K, device="cuda").to(torch.bfloat16); c = torch.empty(M, N, device="cuda")
    for k in VARIANTS:
        lib = C.CDLL(os.path.join(ROOT, "tools", f"_x6_{k}.so"))
        f = lib.alignn_gemm_nt_x6
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(a.data_ptr(), K, w.data_ptr(), None, None, 0, c.data_ptr(), N, M, N, K, st)
        for _ in range(3): call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): call()
        e.record(); torch.cuda.synchronize()
        print(f"{k:18s} {s.elapsed_time(e)/10*1e3:8.1f} us", flush=True)
def run_tn():
    M, N, K = 676200, 256, 256
    g = torch.randn(M, N, device="cuda"); a = torch.randn(M, K, device="cuda"); out = torch.empty(N, K, device="cuda")
    gm = g.abs().max().reshape(1); am = a.abs().max().reshape(1)
    for k in TNV:
        lib = C.CDLL(os.path.join(ROOT, "tools", f"_x6_{k}.so"))
        nbytes = lib.alignn_gemm_tn_x6_workspace; nbytes.restype = C.c_size_t; nbytes.argtypes = [C.c_int64, C.c_int, C.c_int]
        nb = nbytes(M, N, K); ws = torch.empty(nb // 4, device="cuda")
        f = lib.alignn_gemm_tn_x6_partials
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(g.data_ptr(), N, gm.data_ptr(), a.data_ptr(), K, am.data_ptr(), M, N, K, ws.data_ptr(), nb, st)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): call()
        e.record(); torch.cuda.synchronize()
        print(f"TN f16x3 {k:18s} {s.elapsed_time(e)/10*1e3:8.1f} us", flush=True)
TNV = {"warm": [], "base": [], "noload": ["-DX6_ABL_NOALOAD=1"], "onemfma": ["-DX6_ABL_ONEMFMA=1"], "noload_onemfma": ["-DX6_ABL_NOALOAD=1", "-DX6_ABL_ONEMFMA=1"]}
if os.environ.get("ABL_TN") == "1":
    VARIANTS = TNV
    run = run_tn
if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
