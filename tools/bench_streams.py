"""Dev tool: MSMD.forward at B=32 as k independent sub-batches on k HIP streams (clips are independent units)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model

model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
def split(b, k):
    out = []
    n = 32 // k
    for i in range(k):
        sl = slice(i * n, (i + 1) * n)
        out.append({key: (v[sl] if torch.is_tensor(v) and v.shape[:1] == (32,) else (v[sl] if isinstance(v, list) and len(v) == 32 else v))
                    for key, v in b.items()})
    return out
for k in (1, 2, 4):
    parts = split(b, k)
    streams = [torch.cuda.Stream() for _ in range(k)]
    def run():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for s, p in zip(streams, parts):
            with torch.cuda.stream(s):
                bench.step(model, p)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"k={k}: {dt * 1e3:.2f} ms/step -> {3200 / dt:.0f} frames/s", flush=True)
