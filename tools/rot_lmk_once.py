"""A few launches of the small HBM-bound kernels north_star names beside the skinning (for rocprofv3 --kernel-trace --stats):
rotation conversions on N = 4 194 304 rotations (every utils/rotation_conversions.py entry point that has a kernel), the 68
landmarks + dynamic-contour LUT row on 25 600 frames.   python tools/rot_lmk_once.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace

import torch

from msmd_amd import synth
from msmd_amd.utils import rotation_conversions as rc
from msmd_amd.utils.flame import FLAME, FLAMEConfig

N = 1 << 22
g = torch.Generator(device="cuda").manual_seed(0)
aa = torch.randn(N, 3, device="cuda", generator=g)
for _ in range(3):
    R = rc.axis_angle_to_matrix(aa)
    q = rc.matrix_to_quaternion(R)
    rc.quaternion_to_matrix(q)
    rc.matrix_to_euler_angles(R, "XYZ")
    rc.euler_angles_to_matrix(aa, "XYZ")
    rc.matrix_to_rotation_6d(R)
    rc.rotation_6d_to_matrix(R[:, :2].reshape(N, 6))
    rc.quaternion_to_axis_angle(q)
cfg = SimpleNamespace(**vars(FLAMEConfig))
cfg.asset = synth.flame_asset()
fl = FLAME(cfg).to("cuda")
B = 25600
exp, pose = 0.5 * torch.randn(B, 50, device="cuda", generator=g), 0.2 * torch.randn(B, 6, device="cuda", generator=g)
for _ in range(3):
    fl(torch.zeros(B, 100, device="cuda"), exp, pose, return_lm2d=True, return_lm3d=True)
torch.cuda.synchronize()
print(f"rotations: N = {N} ({N * 12 / 1e6:.0f} MB of axis-angle in, {N * 36 / 1e6:.0f} MB of matrices out per conversion); landmarks: {B} frames")
