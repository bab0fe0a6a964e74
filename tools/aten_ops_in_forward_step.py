"""Which host-library (ATen) ops does ONE forward bench step issue, by op, shape and calling line of this package?
(The HIP kernels go through ctypes and do not show up: this lists the glue that is NOT ours.)
  python tools/aten_ops_in_forward_step.py [dtype=bf16]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

if os.environ.get("AUDIO_MODEL") == "hubert_large":      # configs[3]: 10 s clips, 250 frames
    import numpy as np
    from msmd_amd import synth
    model = get_diffusion_model(synthetic_args(audio_model="hubert_large", compute_dtype="bf16", n_motions=250), "cuda").eval()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda")
    b = dict(motion=t(synth.normalish("hl_motion", (8, 250, 67))), audio=t(synth.audio_clips(8, 160000, tag="hl_bench")),
             shape=torch.zeros(8, 100, device="cuda"), style=t(synth.normalish("hl_style", (8, 256))),
             time_step=torch.arange(1, 9, device="cuda"), indicator=torch.ones(8, 250, device="cuda"),
             eps=t(synth.normalish("hl_eps", (8, 250, 67))))
else:
    model = get_diffusion_model(synthetic_args(compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16"), "cuda").eval()
    b = bench.synth_batch(32, 0, "cuda")
    b["time_step"] = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
for _ in range(2):
    bench.step(model, b)
torch.cuda.synchronize()
counts = collections.Counter()
VIEWS = ("view", "_unsafe_view", "reshape", "detach", "alias", "t", "transpose", "permute", "expand", "slice", "select", "unsqueeze",
         "squeeze", "as_strided", "split", "unbind", "chunk", "_reshape_alias", "empty", "empty_like", "empty_strided", "new_empty",
         "is_same_size", "sym_size", "stride", "storage_offset", "numel", "_local_scalar_dense", "lift_fresh")


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if name.split(".")[0] in VIEWS:
            return out
        shp = next((tuple(a.shape) for a in args if torch.is_tensor(a)), ())
        where = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "msmd_amd" in fr.filename or fr.filename.endswith("bench.py"):
                where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                break
        counts[(name, shp, where)] += 1
        return out


with Count():
    bench.step(model, b)
torch.cuda.synchronize()
print(f"{sum(counts.values())} data-moving ATen ops in one forward step")
for (n, s, w), c in sorted(counts.items(), key=lambda kv: kv[0][2]):
    print(f"  {w:28s} {n:30s} {str(s):26s} x{c}")
