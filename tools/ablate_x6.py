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
        src = SRC
        if k.startswith("head"):  # the committed version of the kernel, for same-box A/B against the working tree
            src = os.path.join(ROOT, "alignn_amd", "csrc", "_head_gemm_x6.hip")
            open(src, "w").write(subprocess.run(["git", "show", "HEAD:alignn_amd/csrc/gemm_x6.hip"], cwd=ROOT, check=True, capture_output=True, text=True).stdout)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", os.path.join(ROOT, "tools", f"_x6_{k}.so")] + fl, check=True)
        if src != SRC:
            os.remove(src)
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
        ws.zero_(); assert call() == 0; torch.cuda.synchronize()
        if os.environ.get("ABL_CHECK") == "1":
            if "_tn_ref" not in globals():
                globals()["_tn_ref"] = (k, ws.clone())
            else:
                rk, rw = globals()["_tn_ref"]
                same = torch.equal(ws, rw)
                print(f"   check {k} vs {rk}: {'bit-identical' if same else 'DIFFERENT max|d| = %g' % (ws - rw).abs().max().item()}", flush=True)
def run_f16():
    """f16x3 NT kernel, T x 256 x 256 (the roofline kernel of bench.py)."""
    M, N, K = 676200, 256, 256
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    am = a.abs().max().reshape(1); wm = w.abs().max().reshape(1)
    calls = {}
    st = torch.cuda.current_stream().cuda_stream
    for k in F16V:
        lib = C.CDLL(os.path.join(ROOT, "tools", f"_x6_{k}.so"))
        nb = lib.alignn_split_f16x2_bytes; nb.restype = C.c_size_t; nb.argtypes = [C.c_int, C.c_int]
        img = torch.empty(nb(N, K), dtype=torch.uint8, device="cuda")
        sp = lib.alignn_split_f16x2
        sp.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        assert sp(w.data_ptr(), K, N, K, 0, wm.data_ptr(), img.data_ptr(), st) == 0
        f = lib.alignn_gemm_nt_f16x3
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        calls[k] = (lambda f=f, img=img: f(a.data_ptr(), K, am.data_ptr(), img.data_ptr(), wm.data_ptr(), None, None, 0, c.data_ptr(), N, M, N, K, st))
        for _ in range(3): assert calls[k]() == 0
    torch.cuda.synchronize()
    rounds = int(os.environ.get("ABL_ROUNDS", "1"))
    times = {k: [] for k in calls}
    for _ in range(rounds):  # variants interleaved round by round, so clock / thermal drift hits all of them alike
        for k, call in calls.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): call()
            e.record(); torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 10 * 1e3)
    if os.environ.get("ABL_CHECK") == "1":  # variants that compute the same arithmetic must agree bit for bit
        ref = None
        for rep in range(int(os.environ.get("ABL_CHECK_REPS", "6"))):
            a.normal_(); torch.cuda.synchronize()
            am.copy_(a.abs().max().reshape(1))
            outs = {}
            for k, call in calls.items():
                c.zero_(); assert call() == 0; torch.cuda.synchronize(); outs[k] = c.clone()
            ks = list(outs)
            for k in ks[1:]:
                same = torch.equal(outs[k], outs[ks[0]])
                print(f"check rep {rep}: {k} vs {ks[0]}: {'bit-identical' if same else 'DIFFERENT max|d| = %g' % (outs[k] - outs[ks[0]]).abs().max().item()}", flush=True)
    for k, ts in times.items():
        ts = sorted(ts)
        print(f"NT f16x3 {k:22s} median {ts[len(ts)//2]:8.1f} us   min {ts[0]:8.1f}   max {ts[-1]:8.1f}   ({len(ts)} x 10 launches)", flush=True)
F16V = {"warm": [], "base": [], "nosync": ["-DX6_EPI_NOSYNC=1"], "nolate": ["-DX6_LATE_DMA=0"], "persist": ["-DX6_PERSIST=1"], "head": [], "base2": [], "head2": [], "nostore": ["-DX6_ABL_NOSTORE=1"], "noepi": ["-DX6_ABL_NOSTORE=2"], "noaload": ["-DX6_ABL_NOALOAD=1"],
        "nobload": ["-DX6_ABL_NOBLOAD=1"], "noloads": ["-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1"], "onemfma": ["-DX6_ABL_ONEMFMA=1"],
        "noslice": ["-DX6_ABL_NOSLICE=1"],
        "noloads_noepi": ["-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSTORE=2"],
        "onemfma_noepi": ["-DX6_ABL_ONEMFMA=1", "-DX6_ABL_NOSTORE=2"],
        "noaload_nostore": ["-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSTORE=1"],
        "noslp": ["-fno-slp-vectorize"], "noslp_noloads_noepi": ["-fno-slp-vectorize", "-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSTORE=2"],
        "noslice_noloads_noepi": ["-DX6_ABL_NOSLICE=1", "-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSTORE=2"],
        "kpipe_noslice_noloads_noepi": ["-DX6_KPIPE=1", "-DX6_ABL_NOSLICE=1", "-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSTORE=2"],
        "nopersist": ["-DX6_NO_PERSIST=1"],
        "kpipe": ["-DX6_KPIPE=1"], "kpipe_stage3": ["-DX6_KPIPE=1", "-DX6_NSTAGE=3"], "kpipe_noepi": ["-DX6_KPIPE=1", "-DX6_ABL_NOSTORE=2"],
        "kpipe_noloads_noepi": ["-DX6_KPIPE=1", "-DX6_ABL_NOBLOAD=1", "-DX6_ABL_NOALOAD=1", "-DX6_ABL_NOSTORE=2"],
        "kpipe_stage3_noepi": ["-DX6_KPIPE=1", "-DX6_NSTAGE=3", "-DX6_ABL_NOSTORE=2"],
        "stage3_noepi": ["-DX6_NSTAGE=3", "-DX6_ABL_NOSTORE=2"], "rm1": ["-DX6_FORCE_RM=1"], "stage3": ["-DX6_NSTAGE=3"], "stage3_nostore": ["-DX6_NSTAGE=3", "-DX6_ABL_NOSTORE=1"]}
TNV = {"warm": [], "base": [], "head": [], "pipe": ["-DTN_PIPE=1"], "noload": ["-DX6_ABL_NOALOAD=1"], "onemfma": ["-DX6_ABL_ONEMFMA=1"], "noload_onemfma": ["-DX6_ABL_NOALOAD=1", "-DX6_ABL_ONEMFMA=1"]}
if os.environ.get("ABL_TN") == "1":
    VARIANTS = TNV
    run = run_tn
if os.environ.get("ABL_F16") == "1":
    VARIANTS = F16V
    run = run_f16
if os.environ.get("ABL_ONLY"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["ABL_ONLY"].split(",")}
    F16V = TNV = VARIANTS
if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
