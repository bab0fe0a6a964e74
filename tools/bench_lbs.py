"""Dev tool: FLAME LBS throughput (frames/s, GB/s of algorithmic traffic) on the synthetic asset."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from msmd_amd import synth
from msmd_amd.utils.flame import FLAME, FLAMEConfig
cfg = SimpleNamespace(**vars(FLAMEConfig)); cfg.asset = synth.flame_asset()
fl = FLAME(cfg).to("cuda")
import msmd_amd.utils.lbs as L
for prec in ("fp32", "bf16x3_valu", "bf16x3"):
  fl.lbs_precision = prec
  for B in (6400, 25600):
      torch.manual_seed(B); exp = (0.5 * torch.randn(B, 50, device="cuda")); pose = 0.2 * torch.randn(B, 6, device="cuda"); shape = torch.zeros(B, 100, device="cuda")
      for _ in range(3):
          v, _, _ = fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      R = 10
      e0.record()
      for _ in range(R):
          v, _, _ = fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
      e1.record(); torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / R
      if prec == "fp32": ref = globals().setdefault("REF", {}); ref[B] = v.clone()
      else: print("   max-abs diff vs fp32 kernel:", (v - REF[B]).abs().max().item())
      print(f"{prec} B={B}: {ms:.3f} ms  {B / ms * 1e3:.0f} frames/s  {B * 60936 / ms / 1e6:.1f} GB/s algorithmic (HBM peak 8000; achievable 6300)")
