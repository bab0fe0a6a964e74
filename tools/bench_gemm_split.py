"""Dev tool: time the MSMD_F16X2 (split-pair) GEMM variants on the bench workload's shapes; fp32 kernel alongside.
TF figures are ALGORITHMIC (2 M N K / time): a split product issues 3x that in f16 MFMA FLOPs."""
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
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
OUT_SPLIT = os.environ.get("OUT_SPLIT", "0") == "1"
torch.manual_seed(0)
for (M, N, K, lda, rpb, abs_, name) in shapes:
    if rpb:
        nb = M // rpb
        a32 = torch.randn(nb * (abs_ // 512) + 8, 512, device="cuda")
    else:
        a32 = torch.randn(M, K, device="cuda")
    w32 = torch.randn(N, K, device="cuda") / K ** 0.5
    a, w = ops.to_split(a32), ops.to_split(w32)
    bias = torch.randn(N, device="cuda")
    kw = dict(M=M, K=K, lda=lda, rows_per_batch=rpb, a_batch_stride=abs_) if rpb else {}
    cells = []
    ref = None
    for v in [-1] + variants:
        if v < 0:
            A, W, out = a32, w32, torch.empty(M, N, device="cuda")
        else:
            lib.msmd_set_tuning(3, v)
            A, W = a, w
            out = ops.empty((M, N), "cuda", ops.SPLIT if OUT_SPLIT else torch.float32)
        for _ in range(2):
            ops.gemm(A, W, bias, None, ACT, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 10
        e0.record()
        for _ in range(R):
            ops.gemm(A, W, bias, None, ACT, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / R * 1e3
        tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
        o32 = out.float()
        if ref is None:
            ref = o32.clone(); err = 0.0
        else:
            err = (o32 - ref).abs().max().item()
        cells.append(f"{'f32' if v < 0 else 'v%d' % v}:{tf:5.0f}TF" + ("" if err < 1e-4 else " ERR%.2g" % err))
    print(f"{name:8s} M={M:6d} N={N:4d} K={K:4d} | " + " | ".join(cells), flush=True)
lib.msmd_set_tuning(3, 0)
