"""Dev tool: a few msmd_gemm_tn launches of one shape (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
sp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, K, device="cuda").bfloat16()
ops.set_tuning(2, sp)
for _ in range(5):
    ops.gemm_tn(a, b, want_colsum=False)
torch.cuda.synchronize()
