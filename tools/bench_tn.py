"""Dev tool: msmd_gemm_tn throughput on the training step's weight-gradient shapes vs transposes + NT GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K in [(6400, 768, 768), (6400, 3072, 768), (6400, 768, 3072), (6400, 2304, 768), (3552, 512, 512),
                (3552, 2048, 512), (3520, 1024, 512), (12800, 512, 1536)]:
    a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, K, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    row = f"M={M} N={N} K={K}:"
    for sp in (0, 1, 2, 4, 8, 16):
        ops.set_tuning(2, sp)
        us = t(lambda: ops.gemm_tn(a, b, want_colsum=True))
        row += f"  s{sp} {us:.0f}us {fl / us / 1e6:.0f}TF"
    ops.set_tuning(2, 0)
    def old():
        aT = ops.transpose2d(a, 8); bT = ops.transpose2d(b, 8)
        return ops.gemm(aT, bT, out_dtype=torch.float32)
    us = t(old)
    row += f" | transposes+NT {us:.0f}us"
    print(row, flush=True)
