"""Dev tool: per-shape time/TFLOPs of every msmd_gemm launch in one bench step (HIP events on the launch stream)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd import ops
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = get_diffusion_model(default_args(compute_dtype=dtype), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
for _ in range(2):
    bench.step(model, b)
torch.cuda.synchronize()
ops.GEMM_TRACE = []
R = 5
for _ in range(R):
    bench.step(model, b)
torch.cuda.synchronize()
tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
agg = collections.OrderedDict()
for (M, N, K, batch, dt, e0, e1) in tr:
    k = (M, N, K, batch)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values()) / R
print(f"total gemm ms/step {tot:.3f}")
for (M, N, K, batch), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    f = 2.0 * M * N * K * batch * n
    print(f"M={M:7d} N={N:5d} K={K:5d} b={batch:2d} calls/step={n // R:3d} ms/step={ms / R:7.3f} ({100 * ms / R / tot:5.1f}%) "
          f"TF={f / (ms * 1e-3) / 1e12:7.1f}")
