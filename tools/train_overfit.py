"""Dev tool: sanity of the whole training path -- N iterations on ONE fixed synthetic batch (train-mode noise on,
hipGraph mode by default): the loss must fall and stay finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch
dev = torch.device("cuda:0")
steps = int(os.environ.get("STEPS", "200"))
args = synthetic_args(compute_dtype="bf16", lr=float(os.environ.get("LR", "1e-4")), warm_iter=20)
model = get_diffusion_model(args, dev); se = get_style_encoder(args, "vae2").to(dev)
model.train(); se.train()
tr = Trainer(args, model, se, use_graph=os.environ.get("GRAPH", "1") == "1")
batch = synthetic_batch(16, 0, dev)
if tr.use_graph:
    tr.capture_all(batch)
hist = []
for it in range(1, steps + 1):
    hist.append(tr.step(batch, it=it)["loss"])
    if it % 20 == 0:
        win = torch.stack(hist[-20:]).float()
        print(f"iter {it:4d}: mean loss of the last 20 = {win.mean().item():.5f}  (finite: {bool(torch.isfinite(win).all())})", flush=True)
