"""Dev tool: A/B msmd_exp_set_tuning settings on the forward bench step, each captured as its own hipGraph (replay timing
is stable to ~0.5 %, unlike eager launches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2].split(",")]
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "bf16")), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
b["time_step"] = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
for _ in range(3): bench.step(model, b)
graphs = {}
for v in vals:
    ops.exp_set_tuning(key, v)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): bench.step(model, b)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): bench.step(model, b)
    graphs[v] = g
ops.exp_set_tuning(key, 0)
res = {v: [] for v in vals}
for rep in range(5):
    for v in vals:
        g = graphs[v]
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): g.replay()
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 30 * 1e3)
for v in vals:
    r = sorted(res[v]); print(f"tuning[{key}]={v}: median {r[len(r)//2]:.3f} ms/step  min {r[0]:.3f}")
