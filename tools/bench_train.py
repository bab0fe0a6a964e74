"""Dev tool: training-step time (BASELINE.json configs[2] per-GPU shape: local batch 32, 2 windows, fwd+bwd+Adam)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("TUNE"):   # e.g. TUNE="7=1,6=28": msmd_exp_set_tuning knobs for A/B runs
    from msmd_amd import ops as _ops
    for kv in os.environ["TUNE"].split(","):
        _ops.exp_set_tuning(*(int(v) for v in kv.split("=")))
from msmd_amd import dp, autograd as ag
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch

rank, local_rank, world = dp.env_rank()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
dp.init("nccl", dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
args = synthetic_args(compute_dtype=dtype, lr=2e-5, warm_iter=5000)
model = get_diffusion_model(args, dev)
se = get_style_encoder(args, "vae2").to(dev)
if os.environ.get('TRAIN_MODE'):
    model.train(); se.train()
else:
    model.eval(); se.eval()
tr = Trainer(args, model, se, use_graph=bool(os.environ.get('GRAPH')))
if os.environ.get('ARENA') == '0':
    tr.weight_arena = None
    ag.CACHE.persistent.clear()
if os.environ.get('DIRECT') is not None:
    tr.direct_grad = bool(int(os.environ['DIRECT']))
batch = synthetic_batch(B, rank, dev)
steps = int(os.environ.get("STEPS", "5"))
if tr.use_graph:
    tr.capture_all(batch)
    torch.cuda.synchronize()
    print('graphs captured; mem', torch.cuda.max_memory_allocated() / 2**30)
if os.environ.get("ISSUE"):
    for _ in range(2):
        tr.step(batch, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tr.step(batch, it=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue {(t1 - t0) / 3 * 1e3:.1f} ms/step, total {(t2 - t0) / 3 * 1e3:.1f} ms/step")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        tr.step(batch, it=1)
    torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
el = dp.timed_steps(lambda: tr.step(batch, it=1), steps, 2, sync=torch.cuda.synchronize, device=dev)
if rank == 0:
    print(f"train step: {el / steps * 1e3:.1f} ms/step at local batch {B} x {world} GPUs ({dtype}, {'train' if model.training else 'eval'} mode{', hipGraph' if tr.use_graph else ''}) -> "
          f"{B * 200 * world * steps / el:.0f} frames/s; max mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
