"""Run the split-product NT GEMMs (T x 256 x 256) a few times - a minimal target for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ops
M, N, K = 676200, 256, 256
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
ws6, wsh, am = ops.split_bf16x3(w), ops.split_f16x2(w), ops.absmax(a)
for _ in range(5):
    ops.gemm_nt_x6(a, ws6, b, out=out)
    ops.gemm_nt_f16x3(a, am, wsh, b, out=out)
torch.cuda.synchronize()
