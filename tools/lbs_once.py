"""Dev tool: a handful of skinning launches (for --pmc passes).  python tools/lbs_once.py [frames] [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from msmd_amd import ops, synth
from msmd_amd.utils.flame import FLAME, FLAMEConfig
cfg = SimpleNamespace(**vars(FLAMEConfig)); cfg.asset = synth.flame_asset()
fl = FLAME(cfg).to("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
exp = 0.5 * torch.randn(B, 50, device="cuda"); pose = 0.2 * torch.randn(B, 6, device="cuda"); shape = torch.zeros(B, 100, device="cuda")
for _ in range(R):
    fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
torch.cuda.synchronize()
print("done")
