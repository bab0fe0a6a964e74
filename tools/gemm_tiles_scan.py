"""Dev tool: time of the 128x128 GEMM (variant 17) against the tile count -- shows the wave-quantisation staircase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops, _lib
lib = _lib.load()
V = int(os.environ.get("VARIANT", "17"))
lib.msmd_exp_set_tuning(0, V)
for K in (768, 3072):
    for N in (768,):
        for mt in [int(x) for x in os.environ.get("MT", "21,32,43,50,64,85,86,100,128,171,200").split(",")]:
            M = mt * 128
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda")
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            for _ in range(3):
                ops.gemm(a, w, bias, None, 1, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            R = 30
            e0.record()
            for _ in range(R):
                ops.gemm(a, w, bias, None, 1, out=out)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / R * 1e3
            tiles = mt * (N // 128)
            chk = (out.float() - torch.nn.functional.gelu(a.float() @ w.float().t() + bias)).abs().max().item() if mt <= 50 else float("nan")
            print(f"variant {V} K={K} tiles={tiles:5d} ({tiles / 512:.2f} rounds) maxerr {chk:.3f}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
