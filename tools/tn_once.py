"""Run the split-product TN GEMMs (T x 256 x 256) a few times - a minimal target for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ops
M, N, K = 676200, 256, 256
g = torch.randn(M, N, device="cuda"); a = torch.randn(M, K, device="cuda")
gm, am = ops.absmax(g), ops.absmax(a)
for _ in range(5):
    ops.gemm_tn(g, a)
    ops.gemm_tn(g, a, gm, am)
torch.cuda.synchronize()
