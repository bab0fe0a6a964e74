"""Debug: graph-mode backward with the real one-rank RCCL all-reduce on the side stream vs muted -- which parameters differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MSMD_SEGMENT_GRAPHS"] = "1"
import numpy as np, torch
from msmd_amd import autograd as ag, dp
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch
DEV = "cuda"
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
args = synthetic_args(compute_dtype="bf16", encoder_layers=3, n_layers=2, lr=2e-5, warm_iter=0, gradient_accumulation_steps=1)
B = 8
torch.manual_seed(0)
model = get_diffusion_model(args, DEV).eval(); se = get_style_encoder(args, "vae2").to(DEV).eval()
comm = dp.RcclComm(DEV)
tr = Trainer(args, model, se, use_graph=True, bucket_mb=4.0, comm=comm, exchange_at_world_1=True)
red = tr.reducer
names = {id(p): "se." + n for n, p in se.named_parameters()}; names.update({id(p): n for n, p in model.named_parameters()})
def fwd_bwd(batch, draws):
    ag.DIRECT_GRAD = tr.direct_grad
    cross, trunc = tr._host_choices(draws)
    ag.TrainNoise.graph_safe = True; ag.TrainNoise.spec_masks = None
    red.begin_backward(); tr._stepping = True; red.enabled = False
    out = tr._graph_fwd_bwd(batch, draws, trunc, cross)
    red.finish(); torch.cuda.synchronize()
    return out
mode = sys.argv[1] if len(sys.argv) > 1 else "rccl"
for it in range(1, 8):
    g = np.random.RandomState(100 + it)
    batch = synthetic_batch(B, 0, DEV, it=it % 3)
    draws = dict(cross=[bool(g.rand() < 0.5), False], end_idx=[dev(g.randint(1, 100, size=B)) if it % 2 else None, None],
                 t=[g.randint(1, 501, size=B).tolist() for _ in range(2)],
                 eps=[dev(g.standard_normal((B, 100, 67)).astype(np.float32)) for _ in range(2)],
                 style_eps=[dev(g.standard_normal((B, 256)).astype(np.float32)) for _ in range(2)],
                 cfg_flag=[dev(g.rand(B).astype(np.float32)) for _ in range(2)])
    tr.noise_state[1] += 1
    res = []
    for rep, mute in enumerate((True, False, True, False)):
        red.mute = mute
        red.arena.zero_()
        fwd_bwd(batch, draws)
        res.append(red.arena.clone())
    # the same backward launched eagerly (no graphs) from the same parameters and draws
    red.mute = True
    red.arena.zero_()
    ag.DIRECT_GRAD = tr.direct_grad
    cross, trunc = tr._host_choices(draws)
    ag.TrainNoise.graph_safe = False; ag.TrainNoise.spec_masks = None
    red.begin_backward(); red.enabled = False
    tr._fwd_bwd(batch, draws, trunc, cross)
    red.finish(); torch.cuda.synchronize()
    eager = red.arena.clone()
    b22 = [p for p in red.params if names[id(p)] == "null_audio_feat"][0]
    _, off, n = red.slot[id(b22)]
    print("   null_audio_feat |g| eager %.4e  rep0 %.4e rep1 %.4e rep2 %.4e rep3 %.4e ; max diff eager-rep0 %.3e eager-rep1 %.3e" % (
        float(eager[off:off+n].abs().max()), *[float(r[off:off+n].abs().max()) for r in res],
        float((eager[off:off+n]-res[0][off:off+n]).abs().max()), float((eager[off:off+n]-res[1][off:off+n]).abs().max())))
    print(f"it {it} cross {draws['cross']} trunc {draws['end_idx'][0] is not None}: mute-mute max diff {float((res[0]-res[2]).abs().max()):.3e}  rccl-rccl {float((res[1]-res[3]).abs().max()):.3e}  mute-rccl {float((res[0]-res[1]).abs().max()):.3e}")
    for p in red.params:
        b, off, n = red.slot[id(p)]
        d = float((res[0][off:off+n] - res[1][off:off+n]).abs().max()); m = float(res[0][off:off+n].abs().max())
        if d > 1e-5 * max(m, 1e-30) and d > 0:
            print(f"   bucket {b:2d} {names[id(p)]:60s} max|g| {m:.3e} diff {d:.3e}")
    red.mute = False
    tr._optimizer_step()
