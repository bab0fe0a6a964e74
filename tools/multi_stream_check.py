"""Dev experiment: are concurrent hipGraph replays of the forward step bit-equal to serial ones?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(NS)]
for _ in range(2): bench.step(model, bs[0])
runs = []
for b in bs:
    ts = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
    runs.append(model.capture_forward(b["motion"], b["audio"], b["shape"], b["style"], ts, b["indicator"], b["eps"]))
torch.cuda.synchronize()
serial = []
for r in runs:
    o = r(); torch.cuda.synchronize(); serial.append([x.clone() for x in o])
# serial repeat: deterministic?
for i, r in enumerate(runs):
    o = r(); torch.cuda.synchronize()
    print(f"graph {i}: serial repeat equal: {all(torch.equal(a, b) for a, b in zip(o, serial[i]))}; finite {bool(torch.isfinite(o[1]).all())}")
s = [torch.cuda.Stream() for _ in range(NS)]
for trial in range(6):
    for st in s: st.wait_stream(torch.cuda.current_stream())
    for rep in range(3):
        for i in range(NS):
            with torch.cuda.stream(s[i]): runs[i].graph.replay()
    torch.cuda.synchronize()
    bad = []
    for i in range(NS):
        o = runs[i].static and None
    outs = []
    for i in range(NS):
        with torch.cuda.stream(s[i]): outs.append(runs[i]())
    torch.cuda.synchronize()
    for i in range(NS):
        for j, (a, b) in enumerate(zip(outs[i], serial[i])):
            if not torch.equal(a, b):
                bad.append((i, j, float((a.float() - b.float()).abs().max())))
    print(f"trial {trial}: mismatches {bad}")
