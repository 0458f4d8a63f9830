"""Run the bf16x6 NT GEMM (T x 256 x 256) a few times - a minimal target for rocprofv3 --pmc passes."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from alignn_amd import ops
M, N, K = 676200, 256, 256
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
ws = ops.split_bf16x3(w)
for _ in range(5):
    c = ops.gemm_nt_x6(a, ws)
torch.cuda.synchronize()
