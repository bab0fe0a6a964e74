"""Dev tool: forward bench step with / without the diagonal cross-attention fast path, interleaved."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
for _ in range(5): bench.step(model, b)
res = {True: [], False: []}
for rep in range(6):
    for v in (True, False):
        model.denoising_net.diag_fast_path = v
        for _ in range(2): bench.step(model, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): bench.step(model, b)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 15 * 1e3)
for v in (True, False):
    r = sorted(res[v]); print(f"diag_fast_path={v}: median {r[len(r)//2]:.3f} ms/step  min {r[0]:.3f}")
