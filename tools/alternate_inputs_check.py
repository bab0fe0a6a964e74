"""Dev check: single stream, two different batches alternated back to back -- any kernel that reads its input before the
producer wrote it would pick up the OTHER batch's values from the re-used address (invisible when one batch repeats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]
fn = lambda i: model.audio_encoder.encode(bs[i]["audio"], 25, frame_num=200, dtype=torch.bfloat16, pad=True).float()
refs = []
for i in range(2):
    torch.cuda.empty_cache()
    o = fn(i); torch.cuda.synchronize(); refs.append(o.clone())
bad = 0
R = 60
for rep in range(R):
    outs = [fn(rep % 2) for _ in range(1)] + [fn((rep + 1) % 2)]
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        i = (rep + k) % 2
        if not torch.equal(o, refs[i]):
            bad += 1
            d = (o - refs[i]).abs(); idx = torch.nonzero(d > 0)
            if bad <= 4: print(f"rep {rep}: batch {i}: {idx.shape[0]} differ, clips {sorted(set(idx[:, 0].tolist()))[:16]}")
print(f"single stream, alternating batches: {bad} of {2 * R} results differ")
