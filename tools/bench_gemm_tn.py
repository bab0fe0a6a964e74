"""Weight-gradient GEMM (msmd_gemm_tn + its slab reduction) at the training step's shapes, by contraction split count.
  python tools/bench_gemm_tn.py            RESULT lines: us per call (both launches), TFLOP/s"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import _lib, ops

lib = _lib.load()
p = lambda t: t.data_ptr() if t is not None else None
SHAPES = [(6400, 768, 768), (6400, 2304, 768), (6400, 3072, 768), (6400, 768, 3072), (3520, 512, 512), (3520, 1536, 512),
          (3520, 1024, 512), (3520, 512, 1024), (3520, 2048, 512), (3200, 512, 512)]
if os.environ.get("SHAPES") == "train2":     # both windows in one batch: M = 64 x 200 encoder rows, 64 x 110 / 111 decoder rows
    SHAPES = [(12800, 768, 768), (12800, 2304, 768), (12800, 3072, 768), (12800, 768, 3072), (7040, 512, 512), (7040, 1536, 512),
              (7040, 1024, 512), (7040, 2048, 512), (7040, 512, 2048), (6400, 512, 512)]
ws = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in SHAPES:
    a = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    b = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    c = torch.zeros(N, K, device="cuda")
    cs = torch.zeros(N, device="cuda")
    auto = None
    line = []
    for splits in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        def run():
            _lib.check(lib.msmd_gemm_tn(p(a), p(b), p(c), p(cs), M, N, K, N, K, K, 1, 0, 0, N * K, 0, 0, 1 | (splits << 8), p(ws),
                                        ws.numel(), st), "msmd_gemm_tn")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        line.append(f"{'auto' if splits == 0 else splits}: {us:.1f}")
    print(f"RESULT gemm_tn M={M} N={N} K={K} ({2 * M * N * K / 1e9:.1f} GFLOP) us by splits: " + "  ".join(line), flush=True)
