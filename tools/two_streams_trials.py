"""Dev experiment: graph replays only (no per-step copies / checks), two streams; many trials of a few overlapping
replays, both graphs' outputs compared with their one-at-a-time results after each trial."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]
for _ in range(2): bench.step(model, bs[0])
runs, outs = [], []
for b in bs:
    ts = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
    r = model.capture_forward(b["motion"], b["audio"], b["shape"], b["style"], ts, b["indicator"], b["eps"])
    runs.append(r); outs.append(r())          # r() returns the graph's output buffers
torch.cuda.synchronize()
refs = []
for r, o in zip(runs, outs):
    r.graph.replay(); torch.cuda.synchronize(); refs.append(o[1].clone())
s = [torch.cuda.Stream(), torch.cuda.Stream()]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0; busy = 0.0
for trial in range(T):
    for st in s: st.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(3):
        for i in range(2):
            with torch.cuda.stream(s[i]): runs[i].graph.replay()
    torch.cuda.synchronize(); busy += time.perf_counter() - t0
    bad += sum(not torch.equal(outs[i][1], refs[i]) for i in range(2))
print(f"GPU_FLUSH_ON_EXECUTION={os.environ.get('GPU_FLUSH_ON_EXECUTION')}: {busy / (T * 6) * 1e3:.3f} ms per step on two streams; "
      f"{bad} of {2 * T} outputs differ from one-at-a-time results")
