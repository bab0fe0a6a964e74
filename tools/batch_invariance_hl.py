"""HuBERT-large encoder (24 layers, 10 s clips): does clip 1's hidden state depend on clip 0 being in the batch?  By stage:
the conv stack, then the transformer on identical conv outputs, then attention alone at T = 500."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import synth, ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.utils.model_common import pad_audio_plan
dt = os.environ.get("DTYPE", "bf16")
model = get_diffusion_model(synthetic_args(compute_dtype=dt, audio_model="hubert_large"), "cuda").eval()
enc = model.audio_encoder
audio = torch.from_numpy(synth.audio_clips(2, 160000, tag="hl_full")).cuda()
cd = torch.bfloat16 if dt == "bf16" else (torch.float16 if dt == "fp16" else torch.float32)
r, rep = pad_audio_plan(audio.shape[1])
x2 = enc.feature_extractor_cl(audio, cd, r, rep)
x1 = enc.feature_extractor_cl(audio[1:], cd, r, rep)
print("conv stack   ", float((x2[1:].float() - x1.float()).abs().max()))
h2 = enc.encode_features(x2, cd)
h1 = enc.encode_features(x2[1:].contiguous(), cd)
print("transformer  ", float((h2[1:].float() - h1.float()).abs().max()), float((h2[1:] != h1).float().mean()))
os.environ["X"] = "1"
ops.FOLD_LN = False
h2 = enc.encode_features(x2, cd)
h1 = enc.encode_features(x2[1:].contiguous(), cd)
print("  FOLD_LN off", float((h2[1:].float() - h1.float()).abs().max()))
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(2, 500, 3 * 1024, device="cuda", generator=g).to(cd)
a2 = ops.attention(q[..., :1024], q[..., 1024:2048], q[..., 2048:], 16, 0.125)
a1 = ops.attention(q[1:, :, :1024], q[1:, :, 1024:2048], q[1:, :, 2048:], 16, 0.125)
print("attention 500", float((a2[1:].float() - a1.float()).abs().max()))
