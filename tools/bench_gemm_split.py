"""Split-pair (MSMD_F16X2) GEMM variant A/B on the parity-grade forward's shapes: per-call variant hints, no library state.
  python tools/bench_gemm_split.py 1,80 [reps]   -> us and TFLOP/s (algorithmic: 2 M N K) per shape, variant and output form
  (fp32 C / split C, GELU epilogue with bias), max |err| of every variant against a float64 product of the same inputs.
SHAPES=encoder|conv|large|decoder picks the list (default: all of the B = 32 x 4 s forward)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

variants = (sys.argv[1] if len(sys.argv) > 1 else "1,80,80b").split(",")      # "80b": W through ops.split_weight (MSMD_GEMM_W_BELOW_32)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
conv = [("conv1", 204768, 512, 1536), ("conv2", 102368, 512, 1536), ("conv3", 51168, 512, 1536), ("conv4", 25568, 512, 1536),
        ("conv5", 12768, 512, 1024), ("conv6", 6400, 512, 1024)]
encoder = [("qkv", 6400, 2304, 768), ("out", 6400, 768, 768), ("ffn1", 6400, 3072, 768), ("ffn2", 6400, 768, 3072)]
decoder = [("d_qkv", 3552, 1536, 512), ("d_out", 3552, 512, 512), ("d_ffn1", 3552, 2048, 512), ("d_ffn2", 3552, 512, 2048),
           ("s_qkv", 21312, 1536, 512), ("s_out", 21312, 512, 512), ("s_ffn1", 21312, 2048, 512), ("s_ffn2", 21312, 512, 2048)]
large = [("hl_qkv", 15968, 3072, 1024), ("hl_out", 15968, 1024, 1024), ("hl_ffn1", 15968, 4096, 1024), ("hl_ffn2", 15968, 1024, 4096),
         ("big", 16384, 4096, 3072)]
shapes = {"conv": conv, "encoder": encoder, "decoder": decoder, "large": large}.get(os.environ.get("SHAPES", ""), conv + encoder + decoder)
if ":" in os.environ.get("SHAPES", ""):        # SHAPES=name:M:N:K,name:M:N:K
    shapes = [(n, int(m), int(nn), int(k)) for n, m, nn, k in (t.split(":") for t in os.environ["SHAPES"].split(","))]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    a32 = torch.randn(M, K, device="cuda", generator=g)
    w32 = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    a, w, wb = ops.to_split(a32), ops.to_split(w32), ops.split_weight(w32)
    rows = torch.arange(0, M, max(1, M // 512), device="cuda")[:512]          # float64 check on a sample of rows
    ref = torch.nn.functional.gelu(a32[rows].double() @ w32.double().T + b.double())
    for out_dtype in (torch.float32, ops.SPLIT):
        line = []
        for vs in variants:
            v, w_ = int(vs.rstrip("b")), (wb if vs.endswith("b") else w)
            out = ops.gemm(a, w_, b, None, ops.ACT_GELU, out_dtype=out_dtype, variant=v)
            torch.cuda.synchronize()
            got = (out.float() if isinstance(out, ops.Split) else out)[rows].double()
            err = (got - ref).abs().max().item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    ops.gemm(a, w_, b, None, ops.ACT_GELU, out=out, variant=v)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            line.append(f"v{vs}: {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:6.1f} TF err {err:.1e}")
        print(f"{name:7s} {M:6d}x{N:4d}x{K:4d} {'f32  ' if out_dtype == torch.float32 else 'split'}  " + "   ".join(line), flush=True)
