"""Dev tool: aten-level op list of one forward bench step (which host-library ops still launch kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
b["time_step"] = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
for _ in range(3): bench.step(model, b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    bench.step(model, b)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
