"""A/B of settings on the forward bench step in ONE process: every setting is captured as its own hipGraph, the graphs
are replayed alternately (median / min of 7 rounds x 30 replays; +-0.3 % repeatability, against +-2 % between boxes).
  python tools/ab_forward.py NAME=PYTHON_STATEMENT ...      statements run with `ops`, `torch`, `os` in scope before the capture
  e.g.  base=pass  'n96=ops.GEMM_ROUTER=lambda M,N,K,b: 37 if N==768 and M==6400 else None'
env: DTYPE (bf16), MSMD_LIB (the experimental library for experimental variants)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

settings = [a.split("=", 1) for a in sys.argv[1:]] or [["base", "pass"]]
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "bf16")), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
b["time_step"] = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
for _ in range(3):
    bench.step(model, b)
graphs, outs = {}, {}
for name, stmt in settings:
    ops.GEMM_ROUTER = None
    exec(stmt, {"ops": ops, "torch": torch, "os": os})
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bench.step(model, b)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs[name] = bench.step(model, b)[1]
    graphs[name] = g
ops.GEMM_ROUTER = None
res = {n: [] for n, _ in settings}
for rep in range(7):
    for name, _ in settings:
        g = graphs[name]
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 30 * 1e3)
base = outs[settings[0][0]]
for name, _ in settings:
    r = sorted(res[name])
    same = torch.equal(outs[name], base)
    print(f"{name:12s} median {r[len(r) // 2]:.3f} ms/step  min {r[0]:.3f}   output {'== first setting' if same else 'differs: max %.3g' % float((outs[name].float() - base.float()).abs().max())}")
