"""Does a clip's 16-bit result depend on who else is in the batch?  extract_audio_feature and a 3-step eager sample() of clip 0
alone, in a batch of 2 and in a batch of 32 (same clip 0): max |difference| of clip 0's rows against the B = 1 run.
   MSMD_FOLD_LN=0 python tools/batch_invariance.py   -> LayerNorm as kernels of its own (no LayerNorm-form GEMMs)
env: DTYPE (bf16)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import synth
from msmd_amd.config import synthetic_args
from msmd_amd.model import DiffusionSchedule, get_diffusion_model

dt = os.environ.get("DTYPE", "bf16")
model = get_diffusion_model(synthetic_args(compute_dtype=dt), "cuda").eval()
T = 3
model.diffusion_sched = DiffusionSchedule(T, "cosine").to("cuda")
g = torch.Generator(device="cuda").manual_seed(5)
audio = torch.from_numpy(synth.audio_clips(32, 64000)).cuda()
style = torch.randn(32, 256, device="cuda", generator=g)
xT = torch.randn(32, 100, 67, device="cuda", generator=g)
noise = {t: torch.randn(32, 100, 67, device="cuda", generator=g) for t in range(2, T + 1)}
shape, ind = torch.zeros(32, 100, device="cuda"), torch.ones(32, 100, device="cuda")
base_f = base_x = None
for B in (1, 2, 5, 32):
    f = model.extract_audio_feature(audio[:B])
    x, _, _ = model.sample(f, shape[:B], style[:B], motion_at_T=xT[:B], indicator=ind[:B], noise={t: z[:B] for t, z in noise.items()})
    if B == 1:
        base_f, base_x = f[0].clone(), x[0].clone()
    print(f"{dt} B={B:2d}: audio feature clip 0 vs B=1: {float((f[0] - base_f).abs().max()):.3e}   sample() clip 0: {float((x[0] - base_x).abs().max()):.3e}", flush=True)
