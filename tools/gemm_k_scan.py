"""Per-launch fixed cost of a GEMM kernel: time against the number of K tiles at fixed M x N (intercept = launch ramp +
prologue + epilogue, slope = one round's K tile).   python tools/gemm_k_scan.py 17,63 [M N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "17,63").split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 6400
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2304
g = torch.Generator(device="cuda").manual_seed(0)
for act, name in ((ops.ACT_NONE, "none"), (ops.ACT_GELU, "gelu")):
    for v in variants:
        pts = []
        for K in (64, 128, 256, 512, 768, 1536, 3072):
            a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda", generator=g)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            graph = torch.cuda.CUDAGraph()
            ops.gemm(a, w, b, None, act, out=out, variant=v)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                for _ in range(20):
                    ops.gemm(a, w, b, None, act, out=out, variant=v)
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            pts.append((K // 64, best))
        slope = (pts[-1][1] - pts[-3][1]) / (pts[-1][0] - pts[-3][0])
        print(f"{M}x{N} act={name} v{v}: " + "  ".join(f"nk={k}:{t:6.1f}" for k, t in pts) + f"   slope {slope:.3f} us/Ktile  intercept {pts[-1][1] - slope * pts[-1][0]:.1f} us", flush=True)
