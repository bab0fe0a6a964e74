"""Dev tool: run one GEMM shape/variant repeatedly (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops, _lib
lib = _lib.load()
v, M, N, K, act = (int(x) for x in sys.argv[1:6])
lib.msmd_exp_set_tuning(0, v)
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(a, w, bias, None, act, out=out)
torch.cuda.synchronize()
