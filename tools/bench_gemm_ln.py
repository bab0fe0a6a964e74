"""Cost of the LayerNorm-folding GEMM epilogue (msmd_gemm_ln) per feature, on the encoder's shapes (M = 32 x 200 rows; env M, D, F):
plain GEMM + LayerNorm kernel  vs  GEMM with statistics out / residual LN on the fly / operand LN through folded weights.
  python tools/bench_gemm_ln.py [dtype=bf16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
M, D, F = (int(os.environ.get(k, v)) for k, v in (("M", 6400), ("D", 768), ("F", 3072)))
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
a768, a3072, u = rn(M, D).to(dt), rn(M, F).to(dt), rn(M, D).to(dt)
wo, w1, w2, wq = (rn(D, D) / 28).to(dt), (rn(F, D) / 28).to(dt), (rn(D, F) / 55).to(dt), (rn(3 * D, D) / 28).to(dt)
bo, b1, b2, bq = rn(D), rn(F), rn(D), rn(3 * D)
gam, bet = rn(D).abs() + 0.5, rn(D) * 0.1
st = torch.stack([u.float().reshape(M, D // 64, 64).sum(-1), (u.float() ** 2).reshape(M, D // 64, 64).sum(-1)], -1).transpose(0, 1).contiguous()
cs1, csq = w1.float().sum(1).contiguous(), wq.float().sum(1).contiguous()


def timeit(name, fn, n=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) / n * 1e3:8.2f} us", flush=True)


timeit(f"layernorm {M} x {D}", lambda: ops.layernorm(u, gam, bet))
for nm, a, w, b in (("wo  D<-D ", a768, wo, bo), ("w2  D<-F", a3072, w2, b2)):
    timeit(f"{nm} gemm + residual", lambda: ops.gemm(a, w, b, u))
    timeit(f"{nm} gemm_ln + residual (no LN features)", lambda: ops.gemm_ln(a, w, b, u))
    timeit(f"{nm} gemm_ln + residual, stats out", lambda: ops.gemm_ln(a, w, b, u, stats_out=True))
    timeit(f"{nm} gemm_ln + LN(residual)", lambda: ops.gemm_ln(a, w, b, u, r_stats=st, r_gamma=gam, r_beta=bet))
    timeit(f"{nm} gemm_ln + LN(residual), stats out", lambda: ops.gemm_ln(a, w, b, u, r_stats=st, r_gamma=gam, r_beta=bet, stats_out=True))
for nm, w, b, cs, act in (("w1  F<-D", w1, b1, cs1, ops.ACT_GELU), ("qkv 3D<-D", wq, bq, csq, ops.ACT_NONE)):
    timeit(f"{nm} gemm", lambda: ops.gemm(u, w, b, act=act))
    timeit(f"{nm} gemm_ln, LN(operand)", lambda: ops.gemm_ln(u, w, b, act=act, a_stats=st, w_colsum=cs))
