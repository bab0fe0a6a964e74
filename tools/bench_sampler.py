"""Dev tool: MSMD.sample throughput (T=500, 3 CFG entries) at several batch sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("TUNE"):   # e.g. TUNE="7=1,6=28": msmd_exp_set_tuning knobs for A/B runs
    from msmd_amd import ops as _ops
    for kv in os.environ["TUNE"].split(","):
        _ops.exp_set_tuning(*(int(v) for v in kv.split("=")))
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
model = get_diffusion_model(synthetic_args(compute_dtype=os.environ.get("DTYPE", "bf16")), "cuda").eval()
if os.environ.get("PQA") == "0":
    model.denoising_net.fused_person_query = False
T = int(os.environ.get("T", "500"))
for B in [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,8,64").split(",")]:
    af = torch.randn(B, 100, 512, device="cuda"); shape = torch.zeros(B, 100, device="cuda"); style = torch.randn(B, 256, device="cuda")
    ind = torch.ones(B, 100, device="cuda")
    if T != 500:
        from msmd_amd.model import DiffusionSchedule
        model.diffusion_sched = DiffusionSchedule(T, "cosine").to("cuda")
    model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B}: {dt:.3f} s for {T} steps x 3 entries -> {B * 100 / dt:.1f} frames/s, {dt / T * 1e3:.3f} ms/step, "
          f"{B * 3 * 7.886e9 * T / dt / 1e12:.1f} TFLOP/s")
