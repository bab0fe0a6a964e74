"""Race screen of the 256 x 256 8-phase kernel (variant 80): random eligible shapes (ragged M, 2 ... 48 K tiles, one tile per
workgroup up to several), every one against the 128 x 128 kernel's bits, repeated, with a second stream keeping HBM and the L2s
busy (copies + GEMMs) so that LDS-DMA arrival times move around.  The hand-offs are ordered by counted waits and barriers, never
by timing: any mismatch here is a schedule bug.   python tools/gemm8_soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = torch.Generator(device="cpu").manual_seed(7)
side = torch.cuda.Stream()
big_a = torch.randn(64 << 20, device="cuda")
big_b = torch.empty_like(big_a)
xa = torch.randn(8192, 1024, device="cuda").to(torch.bfloat16)
xw = torch.randn(1024, 1024, device="cuda").to(torch.bfloat16)
t_end, n_cases, n_runs, bad = time.time() + budget, 0, 0, 0
while time.time() < t_end:
    dtype = (torch.bfloat16, torch.float16)[int(torch.randint(0, 2, (1,), generator=g))]
    M = int(torch.randint(200, 40000, (1,), generator=g))
    N = 256 * int(torch.randint(1, 9, (1,), generator=g))
    K = 64 * int(torch.randint(2, 49, (1,), generator=g))
    a = torch.randn(M, K, generator=g).to("cuda", dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to("cuda", dtype)
    b = torch.randn(N, generator=g).to("cuda")
    r = torch.randn(M, N, generator=g).to("cuda", dtype) if int(torch.randint(0, 2, (1,), generator=g)) else None
    ref = ops.gemm(a, w, b, r, act=ops.ACT_NONE, variant=17, flags=0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):      # noise: HBM streaming + another GEMM's workgroups beside the kernel under test
        for _ in range(6):
            big_b.copy_(big_a)
            ops.gemm(xa, xw, None, None, variant=17)
    outs = [ops.gemm(a, w, b, r, act=ops.ACT_NONE, variant=80, flags=0) for _ in range(12)]
    torch.cuda.synchronize()
    n_cases += 1
    for o in outs:
        n_runs += 1
        if not torch.equal(o, ref):
            bad += 1
            d = (o.float() - ref.float()).abs()
            print(f"MISMATCH M={M} N={N} K={K} {dtype}: {int((d > 0).sum())} elements, max {float(d.max()):.4f}", flush=True)
print(f"{n_cases} shapes, {n_runs} launches of variant 80 compared with variant 17: {bad} mismatching launches")
# LayerNorm forms (row statistics by LDS-DMA, statistics out): every launch must repeat its own first result bit for bit
from msmd_amd import ops as O
ln_runs = ln_bad = 0
t_end = time.time() + budget / 3
while time.time() < t_end:
    dtype = (torch.bfloat16, torch.float16)[int(torch.randint(0, 2, (1,), generator=g))]
    M = 2 * int(torch.randint(150, 12000, (1,), generator=g))
    D = 256 * int(torch.randint(1, 5, (1,), generator=g))
    F = 256 * int(torch.randint(2, 13, (1,), generator=g))
    u0 = (torch.randn(M, D, generator=g) * 2 + 0.3).to("cuda", dtype)
    a2 = torch.randn(M, D, generator=g).to("cuda", dtype)
    w1 = (torch.randn(D, D, generator=g) / D ** 0.5).to("cuda", dtype)
    b1 = torch.randn(D, generator=g).to("cuda")
    g0, be0 = (torch.rand(D, generator=g) + 0.5).to("cuda"), (torch.randn(D, generator=g) * 0.1).to("cuda")
    xs = u0.double().reshape(M, -1, 64)
    st0 = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).transpose(0, 1).float().contiguous()
    w2f, cs2, b2f = ops.fold_layernorm(torch.randn(F, D, generator=g).to("cuda") / D ** 0.5, torch.randn(F, generator=g).to("cuda"),
                                       (torch.rand(D, generator=g) + 0.5).to("cuda"), (torch.randn(D, generator=g) * 0.1).to("cuda"), dtype)
    if ((M + 127) // 128) * (D // 128) < 192:
        continue            # the producer's statistics would be 32-column slabs (64 x 64 tiles): not this kernel's layout
    O.GEMM_LN_TILE = 80
    try:
        first = None
        with torch.cuda.stream(side):
            for _ in range(4):
                big_b.copy_(big_a)
        for _ in range(8):
            c1, s1 = ops.gemm_ln(a2, w1, b1, u0, r_stats=st0, r_gamma=g0, r_beta=be0, stats_out=True)
            f = ops.gemm_ln(c1, w2f, b2f, act=ops.ACT_GELU, a_stats=s1, w_colsum=cs2)
            torch.cuda.synchronize()
            ln_runs += 1
            if first is None:
                first = (c1.clone(), s1.clone(), f.clone())
            elif not (torch.equal(c1, first[0]) and torch.equal(s1, first[1]) and torch.equal(f, first[2])):
                ln_bad += 1
                print(f"LN MISMATCH M={M} D={D} F={F} {dtype}", flush=True)
    finally:
        O.GEMM_LN_TILE = None
print(f"LayerNorm forms: {ln_runs} producer + consumer launch pairs, {ln_bad} not repeating their first result")
bad += ln_bad
sys.exit(1 if bad else 0)
