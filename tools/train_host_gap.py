"""Is the training step's host side on the GPU's critical path?  Times K graph-replay steps (a) as shipped, (b) with the
per-step SpecAugment mask draw replaced by a cached pinned tensor (no host compute), and prints the host time spent INSIDE
each Trainer.step call (a call that takes ~ the step time is blocking somewhere).   python tools/train_host_gap.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = synthetic_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
model = get_diffusion_model(args, "cuda").train()
se = get_style_encoder(args, "vae2").to("cuda").train()
tr = Trainer(args, model, se, use_graph=True)
batch = synthetic_batch(32, 0, "cuda")
tr.capture_all(batch)


def timed(tag):
    for _ in range(3):
        tr.step(batch, it=1)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(K):
        h0 = time.perf_counter()
        tr.step(batch, it=1)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    host.sort()
    print(f"RESULT {tag}: {dt * 1e3:.3f} ms per step; host time inside step(): median {host[len(host) // 2] * 1e3:.2f} ms, "
          f"min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f}", flush=True)


timed("as shipped")
orig = tr._draw_spec_mask
cache = {}
t0 = time.perf_counter()
for _ in range(10):
    orig((32, 200))
print(f"RESULT one SpecAugment mask draw on the host: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")


def cached(shape):
    k = tuple(shape)
    if k not in cache:
        cache[k] = orig(shape)
    return cache[k]


tr._draw_spec_mask = cached
timed("mask draw cached (no host compute)")
tr._draw_spec_mask = orig
timed("as shipped again")

# where does the host wait?  time the graph launch itself and everything else in step()
import torch.cuda.graphs as tg_
acc = {"replay": 0.0, "n": 0}
_replay = torch.cuda.CUDAGraph.replay


def replay(self):
    t = time.perf_counter()
    _replay(self)
    acc["replay"] += time.perf_counter() - t
    acc["n"] += 1


torch.cuda.CUDAGraph.replay = replay
for _ in range(3):
    tr.step(batch, it=1)
torch.cuda.synchronize()
acc.update(replay=0.0, n=0)
t0 = time.perf_counter()
for _ in range(K):
    tr.step(batch, it=1)
host_total = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"RESULT host time per step {host_total / K * 1e3:.2f} ms of which inside CUDAGraph.replay() {acc['replay'] / K * 1e3:.2f} ms ({acc['n'] // K} replays per step)")
# a replay issued on an IDLE device: the pure launch cost of this graph
torch.cuda.synchronize()
t = time.perf_counter()
tr.step(batch, it=1)
print(f"RESULT step() on an idle device returns after {(time.perf_counter() - t) * 1e3:.2f} ms")
torch.cuda.synchronize()

# bare replays: one graph variant K times + the optimizer step, no staging copies, no mask draws -- against the same variant
# through step() (draws pinned by replaying the host choices)
torch.cuda.CUDAGraph.replay = _replay
for key, ent in sorted(tr._graphs.items(), key=str):
    g = ent[3]
    def bare():
        tr.noise_state[1] += 1
        for gseg, _ in g:
            gseg.replay()
        tr._optimizer_step()
    for _ in range(3):
        bare()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        bare()
    torch.cuda.synchronize()
    t_bare = (time.perf_counter() - t0) / K
    tr._host_choices = lambda draws, k=key: ([False, False], [k[1], k[2]])
    for _ in range(3):
        tr.step(batch, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        tr.step(batch, it=1)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / K
    print(f"RESULT variant trunc={key[1:3]}: bare replay + Adam {t_bare * 1e3:.3f} ms, through step() {t_step * 1e3:.3f} ms", flush=True)
