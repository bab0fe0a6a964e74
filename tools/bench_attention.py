"""Attention launches of the forward step in isolation: encoder (B = 32, 12 heads, T = 200), decoder self (32 x 8 heads,
111 tokens) and masked cross attention (111 x 110).   python tools/bench_attention.py [dtype=bf16]   env MSMD_ATTN_NW (dev)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
g = torch.Generator(device="cuda").manual_seed(0)
for name, B, H, Tq, Tk, masked in (("encoder", 32, 12, 200, 200, False), ("dec self", 32, 8, 111, 111, False),
                                    ("dec cross", 32, 8, 111, 110, True), ("sampler self", 192, 8, 111, 111, False)):
    d = H * 64
    qkv = torch.randn(B, max(Tq, Tk), 3 * d, device="cuda", generator=g).to(dt)
    q, k, v = qkv[:, :Tq, :d], qkv[:, :Tk, d:2 * d], qkv[:, :Tk, 2 * d:]
    mask = None
    if masked:
        mask = torch.ones(Tq, Tk, dtype=torch.bool, device="cuda")
        mask[0] = False
        for t in range(1, Tq):
            mask[t, t - 1] = False
        mask = mask.to(torch.uint8)
    o = ops.attention(q, k, v, H, 0.125, mask=mask)
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.float().view(B, Tq, H, 64).transpose(1, 2), k.float().view(B, Tk, H, 64).transpose(1, 2),
        v.float().view(B, Tk, H, 64).transpose(1, 2), attn_mask=None if mask is None else ~mask.bool(), scale=0.125)
    err = float((o.float().view(B, Tq, H, 64).transpose(1, 2) - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(50):
            ops.attention(q, k, v, H, 0.125, mask=mask)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    print(f"{name:13s} B={B} H={H} {Tq}x{Tk}{' masked' if masked else ''}: {best * 1e3:6.1f} us   max err vs fp32 softmax {err:.4f}", flush=True)
