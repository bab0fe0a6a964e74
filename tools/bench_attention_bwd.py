"""Fused attention backward (msmd_attention_bwd) at the training step's shapes: us per launch, with / without an explicit
mask and attention dropout.   python tools/bench_attention_bwd.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

dev = "cuda"
state = torch.tensor([1234, 0], device=dev, dtype=torch.int64)    # [seed, step], as autograd.TrainNoise.state
for (B, H, Tq, Tk, masked, p) in ((32, 12, 200, 200, False, 0.0), (32, 12, 200, 200, False, 0.1), (64, 8, 110, 110, True, 0.0),
                                  (64, 8, 110, 110, True, 0.1), (64, 8, 110, 100, True, 0.1), (32, 16, 250, 250, False, 0.0)):
    d = 64 * H
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v, do = (torch.randn(B, T, d, device=dev, generator=g).to(torch.bfloat16) for T in (Tq, Tk, Tk, Tq))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    mask = (torch.rand(Tq, Tk, device=dev, generator=g) < 0.3).to(torch.uint8) if masked else None
    if mask is not None:
        mask[:, 0] = 0
    run = lambda: ops.attention_bwd(q, k, v, do, dq, dk, dv, H, 0.125, mask, p, state if p else None, 7)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"RESULT attention_bwd B={B} H={H} Tq={Tq} Tk={Tk} mask={int(masked)} p_drop={p}: {e0.elapsed_time(e1) * 20:.1f} us")
