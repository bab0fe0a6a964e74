"""The encoder's positional conv (k = 128, 16 groups of 48 channels) as the batched windowed GEMM the encoder launches:
M = B x T = 6400 rows per group, N = 48, K = 128 x 48 = 6144, batch = 16 -- 60 GFLOP, 10 % of the forward step's FLOPs.
  python tools/bench_posconv.py 9,12[,...]      variants per call (experimental ones need MSMD_LIB=.../libmsmd_hip_exp.so)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,9,12").split(",")]
B, T, d, G, kpos = 32, 200, 768, 16, 128
cg = d // G
g = torch.Generator(device="cuda").manual_seed(0)
DT = os.environ.get("DTYPE", "bf16")
h = torch.randn(B, T, d, device="cuda", generator=g)
w = torch.randn(G, cg, kpos * cg, device="cuda", generator=g) / (kpos * cg) ** 0.5
if DT == "f16x2":        # split storage keeps 32-element blocks whole: 48 channels per group padded to 64
    cgp = 64
    wp = torch.zeros(G, cg, kpos, cgp, device="cuda")
    wp[..., :cg] = w.reshape(G, cg, kpos, cg)
    w = ops.to_split(wp.reshape(G, cg, kpos * cgp).contiguous())
else:
    cgp = cg
    h, w = h.to({"bf16": torch.bfloat16, "fp16": torch.float16}[DT]), w.to({"bf16": torch.bfloat16, "fp16": torch.float16}[DT])
bias = torch.randn(d, device="cuda", generator=g)
xp = ops.group_pad(h, G, kpos // 2, cg_out=cgp, split=DT == "f16x2")
Tp = T + kpos
base = None
for v in variants:
    y = torch.empty_like(h)

    def run():
        ops.gemm(xp, w, bias, h, ops.ACT_GELU, out=y, M=B * T, N=cg, K=kpos * cgp, lda=cgp, rows_per_batch=T,
                 a_batch_stride=G * Tp * cgp, ldw=kpos * cgp, ldc=d, batch=G, strideA=Tp * cgp, strideW=cg * kpos * cgp,
                 strideC=cg, strideBias=cg, strideR=cg, variant=v)
    try:
        run()
    except Exception as e:
        print(f"variant {v}: {e}")
        continue
    torch.cuda.synchronize()
    base = y.clone() if base is None else base
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"variant {v:3d}: {best * 1e3:7.1f} us  {2.0 * B * T * d * kpos * cg / best / 1e9:6.1f} TFLOP/s  "
          f"{'== first' if torch.equal(y, base) else 'max diff %.3g' % float((y.float() - base.float()).abs().max())}", flush=True)
