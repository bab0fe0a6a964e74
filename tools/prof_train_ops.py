"""Dev tool: which Python call sites issue the ATen ops of one eager training step (TorchDispatchMode + traceback)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch
args = default_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
model = get_diffusion_model(args, "cuda").train()
se = get_style_encoder(args, "vae2").to("cuda").train()
tr = Trainer(args, model, se, use_graph=False)
batch = synthetic_batch(32, 0, "cuda")
for _ in range(2): tr.step(batch, it=1)
torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.as_strided", "aten.alias", "aten.empty", "aten.split", "aten.unbind", "aten.chunk",
        "aten.is_", "aten.sym_", "aten.size", "aten.stride", "aten.numel", "aten.lift", "aten._local_scalar", "aten.item", "aten.narrow", "aten.unflatten", "aten.flatten")
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "autograd engine / other"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if fr.filename.startswith(ROOT) and "/tools/" not in fr.filename:
                    site = f"{fr.filename.replace(ROOT + '/', '')}:{fr.lineno} {fr.name}"
                    break
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    tr.step(batch, it=1)
torch.cuda.synchronize()
print("dispatched (non-view) ATen ops in one step:", sum(cnt.values()))
agg = collections.Counter()
for (n, s), c in cnt.items(): agg[s] += c
print("--- by call site")
for s, c in agg.most_common(45):
    ops_here = ", ".join(f"{n.replace('aten.', '').replace('.default', '')}x{k}" for (n, ss), k in sorted(cnt.items(), key=lambda kv: -kv[1]) if ss == s)[:150]
    print(f"{c:5d}  {s[:70]:70s} {ops_here}")
