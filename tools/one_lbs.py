"""Dev tool: a few FLAME LBS calls (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from msmd_amd import synth
from msmd_amd.utils.flame import FLAME, FLAMEConfig
cfg = SimpleNamespace(**vars(FLAMEConfig)); cfg.asset = synth.flame_asset(); cfg.lbs_precision = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
fl = FLAME(cfg).to("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
exp = 0.5 * torch.randn(B, 50, device="cuda"); pose = 0.2 * torch.randn(B, 6, device="cuda"); shape = torch.zeros(B, 100, device="cuda")
for _ in range(3):
    fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
torch.cuda.synchronize()
