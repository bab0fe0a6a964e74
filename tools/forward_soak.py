"""Bit-stability soak of the forward step: N hipGraph replays of one captured step (bf16, then fp16), every output compared
on the device with the first replay's; two graphs on two streams alternate so that launches overlap.
  python tools/forward_soak.py [replays=400]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for dt in ("bf16", "fp16"):
    model = get_diffusion_model(synthetic_args(compute_dtype=dt), "cuda").eval()
    runs = [bench.graphed_step(model, bench.synth_batch(32, i, "cuda")) for i in range(2)]
    refs = []
    for r in runs:
        o = r()
        torch.cuda.synchronize()
        refs.append(o[1].clone())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for it in range(n):
        i = it & 1
        with torch.cuda.stream(streams[i]):
            o = runs[i]()
            bad += (o[1] != refs[i]).any().to(torch.int64)
    torch.cuda.synchronize()
    print(f"RESULT {dt}: {int(bad)} of {n} replays (two graphs alternating on two streams) differ from their first replay", flush=True)
    del model, runs
