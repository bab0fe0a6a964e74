"""Dev tool: MSMD.forward bench step eager vs replayed as one hipGraph (static shapes; inputs refreshed by D2D copies)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
b["time_step"] = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
fresh = {k: v.clone() for k, v in b.items() if torch.is_tensor(v)}
for _ in range(3): out_e = bench.step(model, b)
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    bench.step(model, b)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out_g = bench.step(model, b)
def run_graph():
    for k, v in fresh.items():
        b[k].copy_(v, non_blocking=True)
    g.replay()
run_graph(); torch.cuda.synchronize()
print("max diff target", float((out_g[1] - out_e[1]).abs().max()))
for name, fn in (("eager", lambda: bench.step(model, b)), ("graph", run_graph), ("eager", lambda: bench.step(model, b)), ("graph", run_graph)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); print(name, f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step")
