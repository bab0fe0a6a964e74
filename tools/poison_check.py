"""Dev check: does any kernel of the forward path read memory it (or its producer) never wrote?  torch.empty* is patched
to hand out NaN-filled (float) / 0x7f-filled (integer) buffers; the outputs must not change."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = get_diffusion_model(synthetic_args(compute_dtype=dt), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
ref = [x.clone() for x in bench.step(model, b)]
torch.cuda.synchronize()
_empty, _empty_like = torch.empty, torch.empty_like
def poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point(): t.fill_(float("nan"))
        else: t.view(torch.uint8).fill_(0x7f) if t.is_contiguous() else None
    return t
torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
ops.GEMM_TRACE = []
out = bench.step(model, b)
torch.cuda.synchronize()
torch.empty, torch.empty_like = _empty, _empty_like
for i, (o, r) in enumerate(zip(out, ref)):
    o, r = o.float(), r.float()
    same = torch.equal(o, r)
    print(f"output {i} {tuple(o.shape)}: equal {same}; NaNs {int(torch.isnan(o).sum())}" + ("" if same else f"; max diff {float((o - r).abs().nan_to_num(9.0).max()):.3g}"))
