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
# one-subject fast path (FLAME.forward, shape row shared by all frames -> folded template, shape K groups skipped)
bad_uni = 0
for it in range(N // 3):
    B = (16, 64, 100, 1000, 17, 6400)[it % 6]
    g = torch.Generator(device="cuda").manual_seed(10000 + it)
    shape = (0.3 * torch.randn(1, 100, device="cuda", generator=g)).expand(B, -1).contiguous()
    if it % 5 == 4: shape[B // 2, 7] += 0.25          # one frame differs: must fall back to the general path
    exp = 0.5 * torch.randn(B, 50, device="cuda", generator=g); pose = 0.3 * torch.randn(B, 6, device="cuda", generator=g)
    v = fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)[0]
    full_pose = torch.cat([pose[:, :3], torch.zeros(B, 3, device="cuda"), pose[:, 3:], torch.zeros(B, 6, device="cuda")], 1).contiguous()
    coef, coef_hl, A, joints, at = ops.lbs_prepare(torch.cat([shape, exp], 1).contiguous(), full_pose, c.JS, c.parents, 192, want_split=True, want_blend_tiles=True)
    ref = ops.lbs_skin_bf16x3(coef_hl, A, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
    d = (v - ref).abs().max().item()
    if not d < 3e-6:
        bad_uni += 1
        print(f"one-subject iteration {it} B={B}: max|diff| {d:.3e}")
print(f"{N} launches, {bad_runs} with a mismatch; one-subject path: {N // 3} launches, {bad_uni} with a mismatch")
