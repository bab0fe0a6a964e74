"""One-GPU rehearsal of the data-parallel step's two-queue regime.

With world_size 1 the gradient all-reduce has nothing to do, so `GradBucketReducer.standin` runs value-preserving work
on the bucket (K passes of `mul_(1.0)`, optionally followed by an unrelated GEMM) on the SIDE stream exactly where the
RCCL all-reduce would run: behind the same event, under the rest of backward.  Every iteration runs forward + backward
TWICE from the same parameters, inputs and injected draws -- once with the side stream idle, once with it busy -- and
compares the two gradient arenas and loss dicts.  float atomics (bias-gradient column sums, loss accumulators) make
last-bit differences legitimate, so the report gives both the exact-equality count and the relative error; a stale
tile (the two-stream GEMM hazard of DESIGN.md 5b) shows up as an O(1e-2 .. 1) relative error in whole buckets.

  python tools/dp_sidestream_check.py [graph|eager] [steps] ; env: B (32), LAYERS ("12,8"), VARIANT (0), FLAGS (0),
  PASSES (10), SIDE_GEMM (0/1), TRAIN_MODE (0/1), QUIET_TWICE (0/1: compare idle vs idle = the atomics' noise floor)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from msmd_amd import autograd as ag
from msmd_amd import ops, synth
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(os.environ.get("B", "32"))
enc_l, dec_l = (int(v) for v in os.environ.get("LAYERS", "12,8").split(","))
passes = int(os.environ.get("PASSES", "10"))
DEV = "cuda"
ops._GEMM_DEFAULT.update(variant=int(os.environ.get("VARIANT", "0")), flags=int(os.environ.get("FLAGS", "0")) << 16)
if mode == "graph":
    os.environ["MSMD_SEGMENT_GRAPHS"] = "1"

args = synthetic_args(compute_dtype="bf16", encoder_layers=enc_l, n_layers=dec_l, lr=2e-5, warm_iter=0,
                    gradient_accumulation_steps=1)
torch.manual_seed(0)
model = get_diffusion_model(args, DEV)
se = get_style_encoder(args, "vae2").to(DEV)
train_mode = os.environ.get("TRAIN_MODE", "0") == "1"
(model.train() if train_mode else model.eval())
(se.train() if train_mode else se.eval())
tr = Trainer(args, model, se, use_graph=(mode == "graph"), bucket_mb=32.0)
red = tr.reducer
side_w = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16) if os.environ.get("SIDE_GEMM", "0") == "1" else None


def busy(view):
    for _ in range(passes):
        view.mul_(1.0)
    if side_w is not None:     # unrelated MFMA work on the second queue (an all-reduce kernel also occupies CUs)
        ops.gemm(side_w, side_w)


def draws_for(it):
    g = np.random.RandomState(1000 + it)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    trunc0 = bool(g.rand() < 0.5)
    return dict(cross=[bool(g.rand() < 0.5), bool(g.rand() < 0.5)],
                end_idx=[t(g.randint(1, 100, size=B)) if trunc0 else None, None],
                t=[g.randint(1, 501, size=B).tolist() for _ in range(2)],
                eps=[t(g.standard_normal((B, 100, 67)).astype(np.float32)) for _ in range(2)],
                style_eps=[t(g.standard_normal((B, 256)).astype(np.float32)) for _ in range(2)],
                cfg_flag=[t(g.rand(B).astype(np.float32)) for _ in range(2)])


def fwd_bwd(batch, draws):
    """Trainer.step up to (not including) the optimizer: forward + backward + bucket hand-over + join."""
    ag.DIRECT_GRAD = tr.direct_grad
    cross, trunc = tr._host_choices(draws)
    ag.TrainNoise.graph_safe = tr.use_graph
    ag.TrainNoise.spec_masks = None
    red.begin_backward()
    tr._stepping = True
    if tr.use_graph:
        red.enabled = False
        out = tr._graph_fwd_bwd(batch, draws, trunc, cross)
    else:
        red.enabled = True
        out = tr._fwd_bwd(batch, draws, trunc, cross)
    red.finish()
    torch.cuda.synchronize()
    return out


quiet_twice = os.environ.get("QUIET_TWICE", "0") == "1"
exact = bad = nonfinite = 0
worst = 0.0
t0 = time.time()
for it in range(1, steps + 1):
    batch = synthetic_batch(B, 0, DEV, it=it % 4)
    draws = draws_for(it)
    tr.noise_state[1] += 1
    rng_state = tr.rng.get_state()
    red.standin = None
    o1 = fwd_bwd(batch, draws)
    g1 = red.arena.clone()
    red.arena.zero_()
    tr.rng.set_state(rng_state)
    red.standin = None if quiet_twice else busy
    o2 = fwd_bwd(batch, draws)
    g2 = red.arena
    if not (np.isfinite(float(o1["loss"])) and np.isfinite(float(o2["loss"])) and bool(torch.isfinite(g1).all())):
        print(f"it {it}: NON-FINITE loss / gradients ({float(o1['loss'])}, {float(o2['loss'])})", flush=True)
        nonfinite += 1
    same = torch.equal(g1, g2)
    exact += int(same)
    if not same:
        rel = []
        for (s, e, _m) in red.buckets:
            d = (g1[s:e] - g2[s:e]).abs().max()
            rel.append(float(d / g1[s:e].abs().max().clamp_min(1e-30)))
        r = max(rel)
        worst = max(worst, r)
        if r > 1e-3:
            bad += 1
            if bad <= 8:
                print(f"it {it}: max relative gradient difference per bucket {['%.2e' % v for v in rel]}; "
                      f"loss {float(o1['loss']):.6f} vs {float(o2['loss']):.6f}", flush=True)
    red.standin = None
    tr._optimizer_step()          # Adam on the second run's gradients: the parameters keep moving
    tr.sched_step += 1
    tr._scheduler_step(it)
print(f"{mode} mode, B={B}, layers {enc_l}+{dec_l}, train_mode={int(train_mode)}, gemm variant {ops._GEMM_DEFAULT['variant']} "
      f"flags {ops._GEMM_DEFAULT['flags'] >> 16}, side work: {'none (idle vs idle)' if quiet_twice else f'{passes} x mul_(1.0) per bucket' + (' + GEMM' if side_w is not None else '')}; "
      f"{len(red.buckets)} buckets, segments {[len(e[3]) for e in tr._graphs.values()] if tr.use_graph else '-'}")
print(f"RESULT steps={steps} non_finite={nonfinite} bit_equal={exact} gross_mismatch(>1e-3)={bad} worst_rel={worst:.3e} "
      f"({(time.time() - t0) / steps * 1e3:.0f} ms per double step)")
