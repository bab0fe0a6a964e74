"""Dev experiment: ONE hipGraph holding NS independent forward passes on forked capture streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = get_diffusion_model(synthetic_args(compute_dtype=dt), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(NS)]
for b in bs: b["ts"] = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
def fwd(b):
    return model(b["motion"], b["audio"], b["shape"], b["style"], time_step=b["ts"], indicator=b["indicator"], train_with_CFG=False, eps=b["eps"])
refs = []
for b in bs:
    for _ in range(2): o = fwd(b)
    torch.cuda.synchronize(); refs.append(o[1].clone())
side = [torch.cuda.Stream() for _ in range(NS - 1)]
# warm up on the side streams (allocator) before capture
for st, b in zip(side, bs[1:]):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): fwd(b)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cap = torch.cuda.current_stream()
    outs = [None] * NS
    for i, st in enumerate(side):
        st.wait_stream(cap)
        with torch.cuda.stream(st):
            outs[i + 1] = fwd(bs[i + 1])
    outs[0] = fwd(bs[0])
    for st in side: cap.wait_stream(st)
torch.cuda.synchronize()
bad = 0
for trial in range(30):
    g.replay(); torch.cuda.synchronize()
    bad += sum(not torch.equal(outs[i][1], refs[i]) for i in range(NS))
for _ in range(4): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 30
for _ in range(K): g.replay()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
print(f"{dt}: one graph with {NS} forked forward passes: {ms:.3f} ms per replay = {ms / NS:.3f} ms per 32-clip step; mismatching outputs in 30 replays: {bad}")
