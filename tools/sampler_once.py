"""A few EAGER denoising steps of MSMD.sample on one lane's share of the configs[4] batch (B / 2 = 32 clips x 3 CFG entries = 96
sequences x 111 rows, fp16, injected noise -> no hipGraph): the program behind the sampler's --pmc passes (per-kernel
FETCH_SIZE / WRITE_SIZE, tools/prof_r05.sh) -- counters cannot be attributed to kernels inside a graph replay.
   python3 tools/sampler_once.py [B=32] [T=3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import DiffusionSchedule, get_diffusion_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.GEMM_LN_TILE = 15      # the tile the two-lane step graph routes its LayerNorm-epilogue GEMMs to (msmd_amd/sampler.py)
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "fp16")), "cuda").eval()
model.diffusion_sched = DiffusionSchedule(T, "cosine").to("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
af = torch.randn(B, 100, 512, device="cuda", generator=g)
style = torch.randn(B, 256, device="cuda", generator=g)
shape, ind = torch.zeros(B, 100, device="cuda"), torch.ones(B, 100, device="cuda")
xT = torch.randn(B, 100, 67, device="cuda", generator=g)
noise = {t: torch.randn(B, 100, 67, device="cuda", generator=g) for t in range(2, T + 1)}
for _ in range(2):
    x, _, _ = model.sample(af, shape, style, motion_at_T=xT, indicator=ind, cfg_scale=1.15, noise=noise)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(x).all()))
