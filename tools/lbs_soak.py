"""Soak test of the skinning kernel: msmd_lbs_skin_v2 (blend on the matrix pipe) against msmd_lbs_skin_bf16x3 (blend on
the vector ALU) on fresh random inputs, many launches, several batch sizes; any element off by more than 1e-5 is reported.
Usage: python tools/lbs_soak.py [iterations]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from msmd_amd import ops, synth
from msmd_amd.utils.flame import FLAME, FLAMEConfig
cfg = SimpleNamespace(**vars(FLAMEConfig)); cfg.asset = synth.flame_asset()
fl = FLAME(cfg).to("cuda")
fl(torch.zeros(2, 100, device="cuda"), torch.zeros(2, 50, device="cuda"), torch.zeros(2, 6, device="cuda"), return_lm2d=False, return_lm3d=False)
c = fl._pack()["lbs"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad_runs = 0
for it in range(N):
    B = (16, 64, 100, 1000, 17, 6400)[it % 6]
    g = torch.Generator(device="cuda").manual_seed(it)
    betas = torch.cat([0.3 * torch.randn(B, 100, device="cuda", generator=g), 0.5 * torch.randn(B, 50, device="cuda", generator=g)], 1)
    pose = 0.3 * torch.randn(B, 15, device="cuda", generator=g)
    coef, coef_hl, A, joints, at = ops.lbs_prepare(betas, pose, c.JS, c.parents, 192, want_split=True, want_blend_tiles=True)
    ref = ops.lbs_skin_bf16x3(coef_hl, A, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
    out = ops.lbs_skin_v2(at, B, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
    d = (out - ref).abs().max().item()
    if not d < 1e-5:
        bad_runs += 1
        print(f"iteration {it} B={B}: max|diff| {d:.3e}")
print(f"{N} launches, {bad_runs} with a mismatch")
