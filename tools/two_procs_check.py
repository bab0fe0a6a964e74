"""Dev experiment: two PROCESSES sharing the GPU, each single-stream -- are results still bit-stable?  Run two copies:
python tools/two_procs_check.py A & python tools/two_procs_check.py B; each computes its reference alone (A first), then both
loop over the encoder at the same time and count mismatches against their own references."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
who = sys.argv[1]
flag = lambda n: f"/tmp/two_procs_{n}"
if who == "B":
    while not os.path.exists(flag("A_ref")): time.sleep(0.05)
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0 if who == "A" else 1, "cuda")
enc = model.audio_encoder
fn = lambda: enc.encode(b["audio"], 25, frame_num=200, dtype=torch.bfloat16, pad=True).float()
ref = fn(); torch.cuda.synchronize(); ref = ref.clone()
open(flag(who + "_ref"), "w").close()
other = "B" if who == "A" else "A"
while not os.path.exists(flag(other + "_ref")): time.sleep(0.05)
bad = 0; R = 300
t0 = time.perf_counter()
for r in range(R):
    o = fn(); o2 = fn()
    torch.cuda.synchronize()
    bad += (not torch.equal(o, ref)) + (not torch.equal(o2, ref))
dt = (time.perf_counter() - t0) / (2 * R) * 1e3
print(f"process {who}: {bad} of {2 * R} results differ from the process's own reference; {dt:.2f} ms per encoder pass", flush=True)
