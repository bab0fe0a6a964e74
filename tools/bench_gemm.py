"""Dev tool: time msmd_gemm kernel variants on the bench workload's shapes (bf16), verifying each against variant 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops, _lib

lib = _lib.load()
ACT = int(os.environ.get("ACT", "1"))
shapes = [  # (M, N, K, lda, rows_per_batch, a_batch_stride, name)
    (205024, 512, 1536, 1024, 6407, 12815 * 512, "conv1"),
    (51232, 512, 1536, 1024, 1601, 3203 * 512, "conv3"),
    (6400, 3072, 768, 768, 0, 0, "ffn1"),
    (6400, 768, 3072, 3072, 0, 0, "ffn2"),
    (6400, 2304, 768, 768, 0, 0, "qkv"),
    (6400, 768, 768, 768, 0, 0, "oproj"),
    (3552, 512, 512, 512, 0, 0, "dn512"),
    (3552, 2048, 512, 512, 0, 0, "dnffn1"),
    (3552, 512, 2048, 2048, 0, 0, "dnffn2"),
    (3552, 1536, 512, 512, 0, 0, "dnqkv"),
]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5, 6, 7, 8]
torch.manual_seed(0)
res = {}
for (M, N, K, lda, rpb, abs_, name) in shapes:
    if rpb:
        nb = M // rpb
        a = torch.randn(nb * (abs_ // 512) * 512 + 4096, device="cuda").to(torch.bfloat16)
    else:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    ref = None
    for v in variants:
        lib.msmd_exp_set_tuning(0, v)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(M=M, K=K, lda=lda, rows_per_batch=rpb, a_batch_stride=abs_) if rpb else {}
        for _ in range(3):
            ops.gemm(a, w, bias, None, ACT, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 20
        e0.record()
        for _ in range(R):
            ops.gemm(a, w, bias, None, ACT, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / R * 1e3
        tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
        if ref is None:
            ref = out.float().clone()
            err = 0.0
        else:
            err = (out.float() - ref).abs().max().item()
        res[(name, v)] = (us, tf, err)
    print(f"{name:8s} M={M:6d} N={N:4d} K={K:4d} | " + " | ".join(
        f"v{v}:{res[(name, v)][1]:6.0f}TF{'' if res[(name, v)][2] < 0.05 else ' ERR%.2g' % res[(name, v)][2]}" for v in variants), flush=True)
lib.msmd_exp_set_tuning(0, 0)
