"""Same-process A/B of the decoder's last-layer person-token chain in MSMD.sample's hipGraph loop: (lanes, skip_dead_person_chain)
configurations alternating, 3 rounds; every graph's output against the first one's under one seed (the skip returns the same bits).
env: DTYPE (fp16), T (200), B (64), CONFIGS ("10,11,20,21" = lanes, skip digits)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd.config import synthetic_args
from msmd_amd.model import DiffusionSchedule, get_diffusion_model

T, B = int(os.environ.get("T", "200")), int(os.environ.get("B", "64"))
configs = [(int(c[0]), c[1] == "1") for c in os.environ.get("CONFIGS", "10,11,20,21").split(",")]
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "fp16")), "cuda").eval()
model.diffusion_sched = DiffusionSchedule(T, "cosine").to("cuda")
af = torch.randn(B, 100, 512, device="cuda"); shape = torch.zeros(B, 100, device="cuda"); style = torch.randn(B, 256, device="cuda")
ind = torch.ones(B, 100, device="cuda")
xT = torch.randn(B, 100, 67, device="cuda")
res, outs = {c: [] for c in configs}, {}
for rep in range(3):
    for c in configs:
        model.sampler_lanes, model.denoising_net.skip_dead_person_chain = c
        model.__dict__.pop("_step_graphs", None)
        torch.manual_seed(7)
        x, _, _ = model.sample(af, shape, style, motion_at_T=xT, indicator=ind, cfg_scale=1.15)  # capture + warm-up
        outs[c] = x
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, _, _ = model.sample(af, shape, style, motion_at_T=xT, indicator=ind, cfg_scale=1.15)
        torch.cuda.synchronize(); res[c].append((time.perf_counter() - t0) / T * 1e3)
        assert bool(torch.isfinite(x).all())
base = outs[configs[0]]
for c in configs:
    r = sorted(res[c])
    print(f"lanes {c[0]} skip {int(c[1])}: {r[len(r) // 2]:.3f} ms/step (min {r[0]:.3f})  max |x - x[{configs[0]}]| = "
          f"{float((outs[c] - base).abs().max()):.2e}  B={B} T={T}")
