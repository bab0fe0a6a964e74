"""Per-call GEMM variant A/B on the path's shapes (product library: variant is an argument, no global switch).
  python tools/bench_gemm_variants.py 17,40 [reps]   -> TFLOP/s per shape and variant (hot caches, back-to-back launches),
  plus a bit-equality check of every variant against the first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

# "17" = variant 17; "17s" = the same with MSMD_GEMM_STAGGER
# "17" = variant 17; "17s" = the same with MSMD_GEMM_STAGGER; "80w" = with MSMD_GEMM_WRITE_THROUGH
variants = [(int(v.rstrip("sw")), ops.GEMM_PAIRED_STORES | (ops.GEMM_STAGGER if v.endswith("s") else 0) | (ops.GEMM_WRITE_THROUGH if v.endswith("w") else 0))
            for v in (sys.argv[1] if len(sys.argv) > 1 else "17,40").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = [("conv1", 205024, 512, 1536), ("conv3", 51232, 512, 1536), ("conv5", 12800, 512, 1024), ("qkv", 6400, 2304, 768),
          ("ffn1", 6400, 3072, 768), ("ffn2", 6400, 768, 3072), ("out", 6400, 768, 768), ("dec_qkv", 21120, 1536, 512),
          ("dec_ffn1", 21120, 2048, 512), ("big", 16384, 4096, 3072)]
if os.environ.get("SHAPES") == "encoder":      # the forward step's encoder layers: M = 32 x 200 rows
    shapes = [("out", 6400, 768, 768), ("ffn2", 6400, 768, 3072), ("ffn1", 6400, 3072, 768), ("qkv", 6400, 2304, 768)]
if os.environ.get("SHAPES") == "sampler":      # the sampler's decoder layers: M = 192 x 111 rows
    shapes = [("sa_out", 21312, 512, 512), ("ffn2", 21312, 512, 2048), ("ffn1", 21312, 2048, 512), ("qkv", 21312, 1536, 512)]
if os.environ.get("SHAPES") == "denoiser":     # the forward step's decoder layers: M = 32 x 111 rows
    shapes = [("sa_out", 3552, 512, 512), ("ffn2", 3552, 512, 2048), ("ffn1", 3552, 2048, 512), ("qkv", 3552, 1536, 512),
              ("kv_all", 3520, 8192, 512), ("md0", 3520, 256, 512)]
if os.environ.get("SHAPES") == "guide":        # the shapes cdna_hip_programming.md quotes its 256^2 8-phase template on (+ ours)
    shapes = [("4k", 4096, 4096, 4096), ("8k", 8192, 8192, 8192), ("big", 16384, 4096, 3072), ("conv1", 205024, 512, 1536),
              ("qkv", 6400, 2304, 768), ("dec_qkv", 21312, 1536, 512), ("dec_ffn2", 21312, 512, 2048)]
if os.environ.get("SHAPES") == "conv":         # the k = 3 / k = 2 layers of the conv stack at B = 32 x 4 s, and HuBERT-large's encoder at B = 32 x 10 s
    shapes = [("conv1", 204768, 512, 1536), ("conv2", 102368, 512, 1536), ("conv3", 51168, 512, 1536), ("conv4", 25568, 512, 1536),
              ("conv5", 12768, 512, 1024), ("hl_qkv", 15968, 3072, 1024), ("hl_out", 15968, 1024, 1024), ("hl_ffn1", 15968, 4096, 1024),
              ("hl_ffn2", 15968, 1024, 4096), ("tr_qkv", 12800, 2304, 768), ("tr_ffn1", 12800, 3072, 768), ("tr_ffn2", 12800, 768, 3072)]
if os.environ.get("SHAPES") == "train":        # the training step's encoder layers with both windows in one batch: M = 64 x 200 rows
    shapes = [("out", 12800, 768, 768), ("ffn2", 12800, 768, 3072), ("ffn1", 12800, 3072, 768), ("qkv", 12800, 2304, 768),
              ("dec_out", 7104, 512, 512), ("dec_ffn1", 7104, 2048, 512), ("dec_ffn2", 7104, 512, 2048), ("dec_qkv", 7104, 1536, 512)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    base, line = None, []
    for v, fl in variants:
        ops.gemm(a, w, b, None, ops.ACT_GELU, out=out, variant=v, flags=fl)
        torch.cuda.synchronize()
        if base is None:
            base = out.clone()
        same = torch.equal(out, base)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                ops.gemm(a, w, b, None, ops.ACT_GELU, out=out, variant=v, flags=fl)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        line.append(f"v{v}{'s' if fl & ops.GEMM_STAGGER else ''}{'w' if fl & ops.GEMM_WRITE_THROUGH else ''}: {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:7.1f} TF{'' if same else ' MISMATCH'}")
    print(f"{name:9s} {M:6d}x{N:4d}x{K:4d}  " + "   ".join(line), flush=True)
