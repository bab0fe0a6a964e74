import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch
args = synthetic_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
model = get_diffusion_model(args, "cuda").train()
se = get_style_encoder(args, "vae2").to("cuda").train()
tr = Trainer(args, model, se, use_graph=True)
batch = synthetic_batch(32, 0, "cuda")
tr.capture_all(batch)
for _ in range(4): tr.step(batch, it=1)
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); tr.step(batch, it=1); hs.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per step() call: median {sorted(hs)[10]*1e3:.2f} ms, max {max(hs)*1e3:.2f}; issue loop {(t1-t0)/20*1e3:.2f} ms/step; with final sync {(t2-t0)/20*1e3:.2f} ms/step")
