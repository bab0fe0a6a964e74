"""Dev tool: which ATen ops (host-library glue around the msmd_* kernels) one eager training step issues."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import dp
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch

dev = torch.device("cuda:0")
args = default_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
model = get_diffusion_model(args, dev)
se = get_style_encoder(args, "vae2").to(dev)
model.train(); se.train()
tr = Trainer(args, model, se)
batch = synthetic_batch(32, 0, dev)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
agg = {}
for r in rows:
    if not r.key.startswith("aten::"):
        continue
    stack = [s for s in r.stack if "msmd_amd" in s or "ubisoft" in s][:2]
    k = (r.key, str(r.input_shapes)[:70], " <- ".join(s.split("/")[-1][:60] for s in stack))
    agg[k] = agg.get(k, 0) + r.count
for k, c in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:5d}  {k[0]:28s} {k[1]:70s} {k[2]}")
