"""Dev tool: BASELINE.json configs[3] -- HuBERT-large architecture (24 layers), 10 s clips, bf16, MSMD.forward."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from msmd_amd import synth
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = synthetic_args(audio_model="hubert_large", compute_dtype="bf16", n_motions=250)
model = get_diffusion_model(args, "cuda").eval()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
audio = t(synth.audio_clips(B, 160000, tag="hl_bench"))
motion = t(synth.normalish("hl_motion", (B, 250, 67)))
style = t(synth.normalish("hl_style", (B, 256)))
eps = t(synth.normalish("hl_eps", (B, 250, 67)))
ts = [(37 * i + 11) % 500 + 1 for i in range(B)]
shape = torch.zeros(B, 100, device="cuda"); ind = torch.ones(B, 250, device="cuda")
run = lambda: model(motion, audio, shape, style, time_step=ts, indicator=ind, train_with_CFG=False, eps=eps)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
# encoder FLOPs per 10 s clip: conv 49.1 G + pos-conv 8.4 G + 24 x 13.6 G ~= 384 G
print(f"hubert-large 10 s clips, B={B}: {dt * 1e3:.1f} ms/step -> {B * 250 / dt:.0f} frames/s, "
      f"~{B * 384e9 / dt / 1e12:.0f} TFLOP/s on the encoder; max mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
