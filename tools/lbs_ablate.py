"""Dev tool: skinning kernel (msmd_lbs_skin_v2) alone, with the ablation builds of tuning key 8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops, synth
from msmd_amd.utils import lbs as L
from msmd_amd.utils.flame import FLAME, FLAMEConfig
from types import SimpleNamespace
cfg = SimpleNamespace(**vars(FLAMEConfig)); cfg.asset = synth.flame_asset()
fl = FLAME(cfg).to("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
torch.manual_seed(0)
exp = 0.5 * torch.randn(B, 50, device="cuda"); pose = 0.2 * torch.randn(B, 6, device="cuda"); shape = torch.zeros(B, 100, device="cuda")
fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
c = fl._pack()["lbs"]
betas = torch.cat([shape, exp], 1).contiguous()
full_pose = torch.cat([pose[:, :3], torch.zeros(B, 3, device="cuda"), pose[:, 3:], torch.zeros(B, 6, device="cuda")], 1).contiguous()
def timeit(fn, R=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R
tp = timeit(lambda: ops.lbs_prepare(betas, full_pose, c.JS, c.parents, 192, want_split=True, want_blend_tiles=True))
coef, coef_hl, A, joints, at = ops.lbs_prepare(betas, full_pose, c.JS, c.parents, 192, want_split=True, want_blend_tiles=True)
print(f"B={B}: prepare {tp*1e3:.1f} us")
for abl, name in ((0, "full (barrier per 2 tiles)"), (100, "barrier per tile"), (0, "full again"), (100, "per tile again"), (1, "no stores"), (2, "no blend MFMA"), (4, "no blendshape MFMA"), (8, "no DMA"), (7, "no stores/MFMAs"), (15, "nothing")):
    ops.exp_set_tuning(8, abl)
    t = timeit(lambda: ops.lbs_skin_v2(at, B, c.template_planes, c.dirs_hl, c.weight_planes, c.V))
    print(f"  skin_v2 [{name:20s}] {t*1e3:7.1f} us   {B * 60276 / t / 1e6:7.0f} GB/s written")
ops.exp_set_tuning(8, 0)
ops.exp_set_tuning(9, 1)
t = timeit(lambda: ops.lbs_skin_v2(at, B, c.template_planes, c.dirs_hl, c.weight_planes, c.V))
print(f"  skin_v2 [4-wave workgroups, 3 stages] {t*1e3:7.1f} us   {B * 60276 / t / 1e6:7.0f} GB/s written")
ops.exp_set_tuning(9, 0)
ops.exp_set_tuning(8, 0)
for xm in (0, 1, 0, 1):
    ops.exp_set_tuning(10, xm)
    t = timeit(lambda: ops.lbs_skin_v2(at, B, c.template_planes, c.dirs_hl, c.weight_planes, c.V))
    print(f"  skin_v2 [xcd map {'slices dealt one by one' if xm else 'adjacent slices per XCD (default)'}] {t*1e3:7.1f} us")
ops.exp_set_tuning(10, 0)
