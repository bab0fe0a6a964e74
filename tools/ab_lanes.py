"""Same-process A/B of the forward bench step captured with 1 / 2 / 4 lanes (MSMD.capture_forward(lanes=...)): graphs are
replayed alternately (median / min of 7 rounds x 30 replays).  env: DTYPE (bf16), LANES (1,2,4), B (32)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

B = int(os.environ.get("B", "32"))
lanes = [int(x) for x in os.environ.get("LANES", "1,2,4").split(",")]
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "bf16")), "cuda").eval()
b = bench.synth_batch(B, 0, "cuda")
ts = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
runs, outs = {}, {}
for n in lanes:
    try:
        runs[n] = model.capture_forward(b["motion"], b["audio"], b["shape"], b["style"], ts, b["indicator"], b["eps"], lanes=n)
        outs[n] = runs[n]()[1].clone()
    except Exception as e:
        print(f"lanes {n}: {type(e).__name__}: {e}")
res = {n: [] for n in runs}
for rep in range(7):
    for n, r in runs.items():
        for _ in range(3):
            r()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            r()
        torch.cuda.synchronize()
        res[n].append((time.perf_counter() - t0) / 30 * 1e3)
base = outs[lanes[0]]
for n in runs:
    r = sorted(res[n])
    print(f"lanes {n}: drift vs one lane {runs[n].lane_drift:.3g}  median {r[len(r) // 2]:.3f} ms/step  min {r[0]:.3f}   output {'== lanes ' + str(lanes[0]) if torch.equal(outs[n], base) else 'differs: max %.3g' % float((outs[n].float() - base.float()).abs().max())}")
