"""Dev tool: reference point only -- the vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS, no epilogue) on the
path's shapes next to msmd_gemm (bias + GELU fused).  The product never calls the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops
shapes = [(6400, 3072, 768, "ffn1"), (6400, 768, 3072, "ffn2"), (6400, 2304, 768, "qkv"), (6400, 768, 768, "oproj"),
          (3552, 512, 512, "dn512"), (3552, 2048, 512, "dnffn1"), (3552, 512, 2048, "dnffn2"),
          (14208, 512, 512, "samp512"), (14208, 2048, 512, "sampffn1"), (16384, 4096, 3072, "big")]
def timeit(fn):
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
for M, N, K, name in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_lib = timeit(lambda: torch.matmul(a, w.t(), out=out))
    t_lin = timeit(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(a, w, bias.bfloat16())))
    t_our = timeit(lambda: ops.gemm(a, w, bias, None, 1, out=out))
    f = 2.0 * M * N * K / 1e6
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d}  library matmul {t_lib:7.1f} us ({f / t_lib:5.0f} TF)  library linear+gelu {t_lin:7.1f} us"
          f"  msmd_gemm(bias+gelu) {t_our:7.1f} us ({f / t_our:5.0f} TF)", flush=True)
