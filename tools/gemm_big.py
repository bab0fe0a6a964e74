"""Dev tool: saturated throughput of GEMM kernel variants on a chip-filling problem."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops, _lib
lib = _lib.load()
for (M, N, K) in [(16384, 4096, 3072), (16384, 4096, 768), (32768, 2048, 1024)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = []
    for v in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "17,13,21,22,24,25,26,27").split(",")]:
        lib.msmd_exp_set_tuning(0, v)
        for _ in range(3):
            ops.gemm(a, w, bias, None, 1, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, bias, None, 1, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line.append(f"v{v}:{2.0 * M * N * K / us / 1e6:5.0f}TF")
    print(M, N, K, " ".join(line), flush=True)
lib.msmd_exp_set_tuning(0, 0)
