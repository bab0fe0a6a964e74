"""Dev tool: conv0 + GroupNorm + GELU kernel time at the bench shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops
a = torch.randn(32, 64000, device="cuda"); w = torch.randn(512, 10, device="cuda") * 0.3
g = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda")
for dt in (torch.bfloat16, torch.float32):
    for _ in range(3): y = ops.conv0_gn_gelu(a, w, g, b, 20, 0, dt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): y = ops.conv0_gn_gelu(a, w, g, b, 20, 0, dt)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(dt, f"{us:.0f} us (stats + main), out {y.numel() * y.element_size() / 1e6:.0f} MB -> {y.numel() * y.element_size() / us / 1e6:.2f} TB/s")
