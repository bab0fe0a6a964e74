"""Repeat bench.py's verified two-streams leg: how often does any step differ from its one-at-a-time result?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
a = SimpleNamespace(batch=32, steps=60, warmup=4)
bad = 0; fps = []
for r in range(R):
    d = bench.multi_stream_forward(model, a, 0, torch.device("cuda", 0), 2)
    bad += bool(d["mismatching_streams"]); fps.append(d["frames_per_s"])
print(f"{R} runs x {a.steps + a.warmup} verified steps: runs with a mismatch: {bad}; frames/s min {min(fps):.0f} median {sorted(fps)[len(fps)//2]:.0f} max {max(fps):.0f}")
