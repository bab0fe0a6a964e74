"""Dev tool: fused person-token query attention vs the two launches it replaces (sampler shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops
N, T, Tk, H, d = 128, 111, 110, 8, 512
x = torch.randn(N, T, d, device="cuda").bfloat16()
wq = (torch.randn(d, d, device="cuda") / d ** 0.5).bfloat16()
bq = torch.randn(d, device="cuda")
kv = torch.randn(N, Tk, 2 * d, device="cuda").bfloat16()
def fused():
    return ops.person_query_attention(x, wq, bq, kv, H, 0.125)
def two():
    q0 = ops.gemm(x, wq, bq, M=N, K=d, lda=T * d)
    return ops.attention(q0.view(N, 1, d), kv[..., :d], kv[..., d:], H, 0.125)
for name, fn in (("fused", fused), ("two", two)):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 200 * 1e3, "us")
