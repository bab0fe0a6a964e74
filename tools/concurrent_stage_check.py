"""Dev experiment: which stage of MSMD.forward differs when two streams run it concurrently (eager launches)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]
enc = model.audio_encoder
def stage_fe(b): return enc.feature_extractor_cl(b["audio"], torch.bfloat16, 0, 0).float()
def stage_audio(b): return model.extract_audio_feature(b["audio"])
feats = [stage_audio(b).clone() for b in bs]
def stage_full_from_feat(b, i): return model(b["motion"], feats[i], b["shape"], b["style"], time_step=b["time_step"], indicator=b["indicator"], train_with_CFG=False, eps=b["eps"])[1]
def stage_full(b, i): return bench.step(model, b)[1]
stages = {"conv feature extractor": lambda b, i: stage_fe(b), "audio encoder + feature map": lambda b, i: stage_audio(b),
          "denoiser (audio features given)": stage_full_from_feat, "whole forward": stage_full}
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for name, fn in stages.items():
    refs = []
    for i, b in enumerate(bs):
        o = fn(b, i); torch.cuda.synchronize(); refs.append(o.clone())
        o = fn(b, i); torch.cuda.synchronize(); assert torch.equal(o, refs[-1]), name
    bad = 0; worst = 0.0
    for rep in range(12):
        for st in s: st.wait_stream(torch.cuda.current_stream())
        for k in range(2):
            outs = []
            for i in range(2):
                with torch.cuda.stream(s[i]): outs.append(fn(bs[i], i))
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(outs[i], refs[i]):
                bad += 1; worst = max(worst, float((outs[i].float() - refs[i].float()).abs().max()))
    print(f"{name:36s}: {bad} of 24 concurrent results differ (worst {worst:.3g})", flush=True)
