#!/usr/bin/env python3
"""Headline benchmark: motion frames/s of the audio -> motion-coefficient forward on MI355X.

Workload (BASELINE.json configs[1]): one step = MSMD.forward on a batch of 32 synthetic 4 s / 16 kHz clips
(raw audio -> wav2vec2-base encoder -> audio_feature_map -> q-sample -> 8-layer denoiser -> heads), bf16
storage with fp32 accumulation, eval-mode arithmetic, random-init (closed-form synthetic) weights.
100 motion frames per clip, so one step produces 3200 frames per GPU.  N > 1: one process per GPU,
clips are independent (no data-path collective), weak scaling.

Prints ONE JSON line (rank 0) with the contract fields plus:
  roofline     -- the dominant kernel (bf16 MFMA GEMM, csrc/gemm.hip, 128x128 tile): algorithmic FLOPs
                  (2*M*N*K per launch, summed over the launches of one step) / their summed durations, measured
                  live with HIP events on the launch stream in a separate traced pass.
  cpu_baseline -- the numpy oracle (a parity-checked port of the reference's CPU path) timed on this host's
                  cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
FLOP_PER_FRAME = 0.6512e9  # MSMD.forward from raw audio = 65.12 GFLOP / 100-frame clip


def synth_batch(B, rank, device):
    """SURVEY.md section 8(d) inputs: z-normalised pseudo-gaussian audio, motion, zero shape, style, ones indicator."""
    from msmd_amd import synth
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    return dict(
        audio=t(synth.audio_clips(B, 64000, tag=f"bench_audio_r{rank}")),
        motion=t(synth.motion_clips(B, tag=f"bench_motion_r{rank}")),
        shape=torch.zeros(B, 100, device=device),
        style=t(synth.normalish(f"bench_style_r{rank}", (B, 256))),
        indicator=torch.ones(B, 100, device=device),
        eps=t(synth.normalish(f"bench_eps_r{rank}", (B, 100, 67))),
        time_step=[(37 * i + 11) % 500 + 1 for i in range(B)],
    )


def step(model, b):
    return model(b["motion"], b["audio"], b["shape"], b["style"], time_step=b["time_step"], indicator=b["indicator"],
                 train_with_CFG=False, eps=b["eps"])


def graphed_step(model, b):
    """The step as ONE hipGraph replay (MSMD.capture_forward): static shapes, so the ~300 launches are captured once
    and re-issued by the GPU's command processor (host-side launch jitter -- 8 ranks share one host in the scaling
    runs -- no longer shows up in the step time).  Every replay runs all kernels on inputs refreshed by
    device-to-device copies into the captured buffers (new data arriving in HBM); capture_forward checks a replay on
    perturbed inputs against the eager forward, bit for bit.
    (Capturing the batch as two concurrent sub-batch branches looked 4 % faster but replayed one branch against stale
    buffers -- a multi-stream capture hazard -- and two independent graphs on two streams gain nothing: not used.)"""
    ts = torch.tensor(b["time_step"], device=b["audio"].device, dtype=torch.long)
    run = model.capture_forward(b["motion"], b["audio"], b["shape"], b["style"], ts, b["indicator"], b["eps"])
    fresh = dict(motion_feat=b["motion"], audio=b["audio"], shape_feat=b["shape"], style_feat=b["style"], time_step=ts,
                 indicator=b["indicator"], eps=b["eps"])
    return lambda: run(**fresh)


def roofline_leg(model, b, steps=3):
    """Per-launch HIP-event timing of every msmd_gemm launch (the dominant kernel family) over `steps` steps."""
    from msmd_amd import ops
    step(model, b)
    torch.cuda.synchronize()
    ops.GEMM_TRACE = []
    for _ in range(steps):
        step(model, b)
    torch.cuda.synchronize()
    trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    # an event pair around NOTHING still measures the record-to-record latency of the queue; subtract it so that the
    # per-launch figure is the kernel's duration (what rocprofv3 --kernel-trace reports in profiles/)
    empt = []
    for _ in range(200):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        a1.record()
        empt.append((a0, a1))
    torch.cuda.synchronize()
    gaps = sorted(x.elapsed_time(y) for x, y in empt)
    overhead = gaps[len(gaps) // 2]
    flops = 0.0
    ms = 0.0
    n_launch = 0
    big_f = big_ms = 0.0
    big_n = 0
    for (M, N, K, batch, dt, e0, e1) in trace:
        f = 2.0 * M * N * K * batch
        d = max(e0.elapsed_time(e1) - overhead, 1e-4)
        flops += f
        ms += d
        n_launch += 1
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
        if N > 64 and tiles >= 192 and K % 64 == 0:  # launches that run the 128x128 LDS-DMA kernel (csrc/gemm.hip)
            big_f += f
            big_ms += d
            big_n += 1
    achieved = big_f / (big_ms * 1e-3) / 1e12 if big_ms > 0 else 0.0
    return dict(bound="mfma", achieved=round(achieved, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=pmc_traffic(),
                kernel="gemm2_kernel<bf16,128,128,4,2,2,pipelined> (csrc/gemm.hip)",
                launches_per_step=big_n // steps, gflop_per_step=round(big_f / steps / 1e9, 1),
                ms_per_step_in_kernel=round(big_ms / steps, 3),
                all_gemm_tflops=round(flops / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0,
                all_gemm_launches_per_step=n_launch // steps, all_gemm_ms_per_step=round(ms / steps, 3),
                event_overhead_us=round(overhead * 1e3, 2))


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in SEPARATE runs of this same command; KB units; FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md section HBM).  PMC counters cannot be read from inside this process, so the
    figure comes from profiles/ (null when the file is absent)."""
    path = os.path.join(ROOT, "profiles", "r01e_pmc_hbm_fetch_write_per_kernel.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r01d_pmc_hbm_fetch_write_per_kernel.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_fetch_write_per_kernel.json")
    try:
        d = json.load(open(path))
        k = [v for name, v in d.items() if "gemm2_kernel" in name and "Li128ELi128ELi4ELi2ELi2E" in name][0]  # same tile / traffic as the pipelined variant
        return round((2.0 * k["fetch_kb_avg"] + k["write_kb_avg"]) * 1024.0)
    except Exception:
        return None


def cpu_baseline_leg(B=12):
    """Numpy oracle (parity-pinned port of the reference CPU path) on this host: MSMD.forward on B clips."""
    from msmd_amd import shapes, synth
    from msmd_amd.config import default_args
    from oracle import diffusion as od
    args = default_args()
    sd = synth.fill_state_dict(shapes.msmd_shapes(args))
    sched = od.diffusion_schedule(500, "cosine")
    audio = synth.audio_clips(B, 64000, tag="bench_audio_r0")
    motion = synth.motion_clips(B, tag="bench_motion_r0")
    style = synth.normalish("bench_style_r0", (B, 256))
    eps = synth.normalish("bench_eps_r0", (B, 100, 67))
    ts = [(37 * i + 11) % 500 + 1 for i in range(B)]
    t0 = time.time()
    od.msmd_forward(sd, sched, motion, audio, np.zeros((B, 100), np.float32), style, ts, eps,
                    indicator=np.ones((B, 100), np.float32))
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return dict(value=round(B * 100 / dt, 1), unit="frames/s", cores=int(cores), kind="port",
                sample=f"oracle.diffusion.msmd_forward (numpy fp32) on {B} clips of the same synthetic workload, "
                       f"{dt:.1f} s wall")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "f16x2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of one hipGraph replay")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from msmd_amd import dp
    dp.init("nccl", device)

    from msmd_amd.config import default_args
    from msmd_amd.model import get_diffusion_model
    if os.environ.get("MSMD_TUNE"):   # developer A/B knob, e.g. MSMD_TUNE="7=1" (msmd_set_tuning key=value pairs)
        from msmd_amd import ops as _ops
        for kv in os.environ["MSMD_TUNE"].split(","):
            _ops.set_tuning(*(int(v) for v in kv.split("=")))
    args = default_args(compute_dtype=a.dtype)
    model = get_diffusion_model(args, device).eval()
    b = synth_batch(a.batch, rank, device)
    for _ in range(2):
        step(model, b)   # lazy packing / allocator warm-up before any capture
    launch = "eager"
    run = lambda: step(model, b)  # noqa: E731
    if not a.eager:
        try:
            run = graphed_step(model, b)
            launch = "one hipGraph replay per step (inputs refreshed by D2D copies)"
        except Exception as e:  # capture unsupported on this stack: keep the host-launched step (same kernels)
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            torch.cuda.synchronize()
    elapsed = dp.timed_steps(run, a.steps, a.warmup, sync=torch.cuda.synchronize, device=device)

    if rank == 0:
        n = max(world, a.gpus) if dist else 1
        frames = a.batch * 100 * a.steps * n
        value = frames / elapsed
        out = {
            "metric": "FLAME frames/sec on 4s@16kHz clips (whole job; MSMD.forward motion-coefficient frames)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: MSMD.forward, batch=32 x 4 s clips per GPU, wav2vec2-base encoder + "
                                   "8-layer motion decoder, eval-mode, synthetic closed-form weights",
                       "batch_per_gpu": a.batch, "clip_seconds": 4, "frames_per_clip": 100,
                       "parallelism": f"dp{n} (independent clips, no collective)",
                       "launch": launch},
            "end_to_end_tflops": round(value * FLOP_PER_FRAME / 1e12 / n, 1),
        }
        if not a.no_roofline:
            out["roofline"] = roofline_leg(model, b)
        if not a.no_cpu_baseline and n == 1:
            out["cpu_baseline"] = cpu_baseline_leg()
        print(json.dumps(out), flush=True)
    if dist:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
