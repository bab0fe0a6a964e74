"""Dev experiment, part 3: record every op output of extract_audio_feature serially and under 2-stream concurrency;
report the first op whose output differs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd import ops
from msmd_amd.config import default_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(default_args(compute_dtype="bf16"), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]
names = ["gemm", "layernorm", "attention", "conv0_gn_gelu", "interp_linear", "group_pad", "conv1d_cl"]
orig = {n: getattr(ops, n) for n in names}
import threading
rec = {}
def wrap(n):
    f = orig[n]
    def g(*a, **k):
        o = f(*a, **k)
        key = torch.cuda.current_stream().cuda_stream
        if key in rec:
            t = o if torch.is_tensor(o) else o[0]
            if "out" in k and k["out"] is not None: t = k["out"]
            rec[key].append((n, tuple(t.shape), t.detach().clone()))
        return o
    return g
for n in names: setattr(ops, n, wrap(n))
import msmd_amd.utils.wav2vec2 as W
def run(i, stream):
    with torch.cuda.stream(stream):
        rec[stream.cuda_stream] = []
        out = model.extract_audio_feature(bs[i]["audio"])
    return out
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for st in s: st.wait_stream(torch.cuda.current_stream())
serial = []
for i in range(2):
    o = run(i, s[i]); torch.cuda.synchronize(); serial.append((o.clone(), rec[s[i].cuda_stream]))
found = False
for rep in range(12):
    outs = [run(i, s[i]) for i in range(2)]
    torch.cuda.synchronize()
    for i in range(2):
        got = rec[s[i].cuda_stream]
        for k, ((n, shp, t), (n2, shp2, t2)) in enumerate(zip(got, serial[i][1])):
            if not torch.equal(t, t2):
                d = (t.float() - t2.float()).abs()
                idx = torch.nonzero(d > 0)
                print(f"rep {rep} stream {i}: first differing op #{k} {n} {shp}: {idx.shape[0]} elements differ, max {float(d.max()):.4g}; index range {idx.min(0).values.tolist()} .. {idx.max(0).values.tolist()}")
                found = True
                break
    if found and rep >= 3: break
print("ops per call:", len(serial[0][1]), "; any difference found:", found)
