"""Dev tool: A/B the forward bench step under two msmd_set_tuning settings, interleaved in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd import ops
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2].split(",")]
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
for _ in range(5): bench.step(model, b)
res = {v: [] for v in vals}
for rep in range(6):
    for v in vals:
        ops.set_tuning(key, v)
        for _ in range(2): bench.step(model, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): bench.step(model, b)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 15 * 1e3)
for v in vals:
    r = sorted(res[v]); print(f"tuning[{key}]={v}: median {r[len(r)//2]:.3f} ms/step  min {r[0]:.3f}  all {[round(x,2) for x in res[v]]}")
ops.set_tuning(key, 0)
