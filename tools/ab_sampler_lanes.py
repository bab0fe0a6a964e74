"""Same-process A/B of MSMD.sample's hipGraph loop with 1 / 2 / 4 lanes (msmd_amd.sampler.LANES), alternating, 3 rounds.
env: DTYPE (fp16), T (200), B (64), LANES (1,2)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import sampler as smp
from msmd_amd.config import synthetic_args
from msmd_amd.model import DiffusionSchedule, get_diffusion_model

T, B = int(os.environ.get("T", "200")), int(os.environ.get("B", "64"))
lanes = [int(x) for x in os.environ.get("LANES", "1,2").split(",")]
lanes = [(n, False) for n in lanes]
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "fp16")), "cuda").eval()
model.diffusion_sched = DiffusionSchedule(T, "cosine").to("cuda")
af = torch.randn(B, 100, 512, device="cuda"); shape = torch.zeros(B, 100, device="cuda"); style = torch.randn(B, 256, device="cuda")
ind = torch.ones(B, 100, device="cuda")
res = {n: [] for n in lanes}
for rep in range(3):
    for n, f in lanes:
        model.sampler_lanes = n
        model.__dict__.pop("_step_graphs", None)
        x, _, _ = model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)  # capture + warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, _, _ = model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)
        torch.cuda.synchronize(); res[n, f].append((time.perf_counter() - t0) / T * 1e3)
        assert bool(torch.isfinite(x).all())
        used = next(iter(model._step_graphs.values())).lanes
        if rep == 0:
            print(f"lanes asked {n}, used {used}")
for n, f in lanes:
    r = sorted(res[n, f])
    print(f"lanes {n}: {r[len(r) // 2]:.3f} ms/step (min {r[0]:.3f}) B={B} T={T}")
