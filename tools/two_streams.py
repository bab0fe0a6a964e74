"""Dev experiment: two independent forward steps in flight (two hipGraphs on two streams) vs one after the other."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
args = synthetic_args(compute_dtype=dt)
model = get_diffusion_model(args, "cuda").eval()
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bs = [bench.synth_batch(32, r, "cuda") for r in range(NS)]
b0 = bs[0]
for _ in range(2): bench.step(model, b0)
runs = []
for b in bs:
    ts = torch.tensor(b["time_step"], device="cuda", dtype=torch.long)
    r = model.capture_forward(b["motion"], b["audio"], b["shape"], b["style"], ts, b["indicator"], b["eps"])
    runs.append(r)
torch.cuda.synchronize()
ref = []
for r in runs:
    r.graph.replay(); torch.cuda.synchronize(); ref.append(r.graph and None)
outs_serial = []
for r in runs:
    o = r(); torch.cuda.synchronize(); outs_serial.append(o[1].clone())
def timed(fn, K=40):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
one = timed(lambda: runs[0].graph.replay())
s = [torch.cuda.Stream() for _ in range(NS)]
state = {"i": 0}
def alt():
    i = state["i"]; state["i"] = (i + 1) % NS
    with torch.cuda.stream(s[i]):
        runs[i].graph.replay()
for st in s: st.wait_stream(torch.cuda.current_stream())
two = timed(alt)
torch.cuda.synchronize()
# correctness of concurrent replays
os_ = []
for i in range(NS):
    with torch.cuda.stream(s[i]): os_.append(runs[i]())
torch.cuda.synchronize()
ok = all(torch.equal(os_[i][1], outs_serial[i]) for i in range(NS))
print(f"{dt}: one stream {one:.3f} ms/step; {NS} graphs on {NS} streams {two:.3f} ms/step ({one / two:.3f}x); concurrent outputs bit-equal to serial: {ok}")
