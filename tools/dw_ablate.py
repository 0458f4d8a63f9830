"""Where the time of the fused input-gradient + weight-gradient pass (csrc/gemm_dw.hip) goes: the kernel rebuilt with parts
stubbed out (DW_ABL bits; timing only - results are garbage), with other ring depths (DW_NS), and with phase stamps (DW_TRACE).
    python tools/dw_ablate.py build        (here: cross-compiles tools/_dw_<name>.so, which travel to the GPU box)
    python tools/dw_ablate.py [rows]       (on the GPU: interleaved rounds, medians)"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "alignn_amd", "csrc", "gemm_dw.hip")
VARIANTS = {
    "base": [],
    "noW": ["-DDW_ABL=1"],
    "noE": ["-DDW_ABL=2"],
    "noY": ["-DDW_ABL=32"],
    "noG": ["-DDW_ABL=64"],
    "noDMA": ["-DDW_ABL=99"],
    "noMFMAc": ["-DDW_ABL=4"],
    "noMFMAdw": ["-DDW_ABL=8"],
    "noMFMA": ["-DDW_ABL=12"],
    "noStore": ["-DDW_ABL=16"],
    "noDMA_noMFMA": ["-DDW_ABL=111"],
    "noslp": ["-fno-slp-vectorize"],
    "pingpong": ["-DDW_PP=1"],
    "dma_between": ["-DDW_IL=1"],
    "trace": ["-DDW_TRACE=1"],
}
EXTRA = os.environ.get("DW_VARIANTS")  # e.g. "name:-DX=1 -DY=2;name2:..."
if EXTRA:
    for item in EXTRA.split(";"):
        k, v = item.split(":")
        VARIANTS[k] = v.split()


def so(name):
    return os.path.join(HERE, f"_dw_{name}.so")


def build():
    procs = []
    for name, flags in VARIANTS.items():
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", so(name)] + flags
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE)))
    for name, p in procs:
        err = p.communicate()[1].decode()
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + err[-2000:])


def main(T=676200):
    import torch

    sys.path.insert(0, ROOT)
    from alignn_amd import ops

    H = 256
    g = torch.Generator().manual_seed(0)
    gm = torch.randn(T, H, generator=g).cuda()
    y = torch.randn(T, H, generator=g).cuda()
    w = (torch.randn(H, H, generator=g) / 16).cuda()
    res = torch.randn(T, H, generator=g).cuda()
    xn = (torch.randn(T, H, generator=g) * 1.3 + 0.2).cuda()
    stat = torch.stack([xn.mean(0), torch.rsqrt(xn.var(0) + 1e-5), torch.rsqrt(xn.var(0) + 1e-5), torch.zeros(H).cuda()]).contiguous()
    g_amax, y_amax = ops.absmax(gm), ops.absmax(y)
    wt = ops.split_f16x2(w, True)
    out = torch.empty(T, H, device="cuda")
    dW = torch.empty(H, H, device="cuda")
    libs = {}
    for name in VARIANTS:
        if os.path.exists(so(name)):
            lib = C.CDLL(so(name))
            lib.alignn_gemm_dgrad_wgrad_workspace.restype = C.c_size_t
            lib.alignn_gemm_dgrad_wgrad_workspace.argtypes = [C.c_int64]
            lib.alignn_gemm_dgrad_wgrad_slabs.argtypes = [C.c_int64]
            libs[name] = lib
    nbytes = libs["base"].alignn_gemm_dgrad_wgrad_workspace(T)
    ws = torch.empty(nbytes // 4, device="cuda")
    part = torch.empty(2 * libs["base"].alignn_gemm_dgrad_wgrad_slabs(T), 2, H, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch(lib, addend, bn):
        rc = lib.alignn_gemm_dgrad_wgrad_f16x3(p(gm), C.c_int64(H), p(g_amax), p(y), C.c_int64(H), p(y_amax), p(wt.buf), p(wt.amax),
                                               p(addend), C.c_int64(H), p(out), C.c_int64(H), p(xn if bn else None), C.c_int64(H),
                                               p(stat if bn else None), p(part if bn else None), p(dW), C.c_int64(H), C.c_int64(T),
                                               p(ws), C.c_size_t(nbytes), st)
        assert rc == 0, rc

    def t1(fn, n=5):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3

    for _ in range(20):
        launch(libs["base"], res, True)
    cases = [("plain", None, False), ("addend", res, False), ("bnred", None, True), ("bnred+addend", res, True)]
    names = [n for n in libs if n != "trace"]
    print(f"T = {T}: median of 5 interleaved rounds, us (kernel + the 256-slab sum)")
    print(f"{'variant':16s}" + "".join(f"{c[0]:>14s}" for c in cases))
    res_t = {n: {c[0]: [] for c in cases} for n in names}
    for _ in range(5):
        for n in names:
            for cname, addend, bn in cases:
                res_t[n][cname].append(t1(lambda: launch(libs[n], addend, bn)))
    med = lambda v: sorted(v)[len(v) // 2]
    for n in names:
        print(f"{n:16s}" + "".join(f"{med(res_t[n][c[0]]):14.1f}" for c in cases), flush=True)
    if "trace" in libs:
        import numpy as np

        lib = libs["trace"]
        for cname, addend, bn in cases[:2]:
            launch(lib, addend, bn)
            torch.cuda.synchronize()
            buf = np.zeros(256 * 16, dtype=np.uint64)
            lib.alignn_dw_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
            b = buf.reshape(256, 16).astype(np.int64)
            t = b[:, :6]
            d = np.diff(t, axis=1)
            m = lambda x: f"median {int(np.median(x)):7d}  p10 {int(np.percentile(x, 10)):7d}  p90 {int(np.percentile(x, 90)):7d}"
            print(f"-- {cname}: third tile of every workgroup, shader-clock cycles (s_memtime)")
            for i, lab in enumerate(["wait for the tile's G rows", "slice G in place", "weight gradient (4 Y steps)",
                                     "input gradient (16 W steps)", "epilogue"]):
                print(f"   {lab:32s} {m(d[:, i])}")
            print(f"   {'tile period':32s} {m(t[:, 5] - t[:, 0])}")
            for i, lab in enumerate(["G + Y steps", "W steps", "epilogue steps"]):
                print(f"   vmcnt waits in the {lab:16s} {m(b[:, 6 + i])}    barriers {m(b[:, 9 + i])}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main(*(int(v) for v in sys.argv[1:]))
