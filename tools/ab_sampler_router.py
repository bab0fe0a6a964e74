"""Same-process A/B of GEMM tile routing inside the sampler loop (two lanes): settings NAME=PYTHON_EXPRESSION giving
ops.GEMM_ROUTER / ops.GEMM_LN_ROUTER.   env T (200), B (64)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import DiffusionSchedule, get_diffusion_model

T, B = int(os.environ.get("T", "200")), int(os.environ.get("B", "64"))
settings = [a.split("=", 1) for a in sys.argv[1:]] or [["base", "None,None"]]
model = get_diffusion_model(synthetic_args(compute_dtype="fp16"), "cuda").eval()
model.diffusion_sched = DiffusionSchedule(T, "cosine").to("cuda")
af = torch.randn(B, 100, 512, device="cuda"); shape = torch.zeros(B, 100, device="cuda"); style = torch.randn(B, 256, device="cuda")
ind = torch.ones(B, 100, device="cuda")
res = {n: [] for n, _ in settings}
for rep in range(3):
    for name, expr in settings:
        ops.GEMM_ROUTER, ops.GEMM_LN_ROUTER = eval(expr)
        model.__dict__.pop("_step_graphs", None)
        model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, _, _ = model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / T * 1e3)
ops.GEMM_ROUTER = ops.GEMM_LN_ROUTER = None
for name, _ in settings:
    r = sorted(res[name])
    print(f"{name:14s} {r[len(r) // 2]:.3f} ms/step (min {r[0]:.3f})")
