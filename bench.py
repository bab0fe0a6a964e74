#!/usr/bin/env python3
"""Headline benchmark: motion frames/s of the audio -> motion-coefficient hot path on MI355X.

Workload of `value` (BASELINE.json configs[1]): one step = MSMD.forward on a batch of 32 synthetic 4 s / 16 kHz clips
(raw audio -> wav2vec2-base encoder -> audio_feature_map -> q-sample -> 8-layer denoiser -> heads), bf16 storage with
fp32 accumulation, eval-mode arithmetic, closed-form synthetic weights.  100 motion frames per clip, 3200 frames per
GPU and step.  N > 1: one process per GPU (RCCL), clips are independent (no data-path collective), weak scaling.
`python bench.py --gpus N` without WORLD_SIZE in the environment starts the N ranks itself (torch.distributed.run,
from a parent that never touches the GPU) and relays rank 0's line; `n_gpus` is what torch.distributed saw.

One JSON line (rank 0) with the contract fields plus
  max_abs_err_vs_oracle  the headline dtype's output against the CPU restatement of the reference on ALL clips of the
                         batch (the reference's tolerance is 1e-4: bf16 does not meet it, it is the throughput mode);
  parity_mode            the modes that DO meet 1e-4, each timed the same way: "f16x2" (contractions on fp16 split
                         pairs, three MFMAs per k-step: the parity-grade speed mode) and "fp32" (exact-fp32 MFMA), with
                         ms_per_step, frames_per_s, max_abs_err_vs_oracle and the dominant kernel's roofline;
  roofline               dominant kernel of the headline mode: algorithmic FLOPs / launch durations measured live with
                         HIP events on the launch stream; the event pair's overhead is calibrated in-run on a kernel of
                         known duration (msmd_spin_us), NOT assumed; profiles/ holds the rocprofv3 summary of this command;
  legs                   the other BASELINE configs, measured in this run: sampler (B=64, T=500, 3 CFG entries, fp16,
                         hipGraph) + LBS (6400 / 25600 frames, HBM fraction), training step (B=32), HuBERT-large 10 s clips;
  cpu_baseline           oracle/torch_cpu.py (torch-CPU restatement of the reference's fp32 path, pinned to the same
                         reference goldens as the numpy oracle) on this host's cores: median of 5 after 2 warm-ups.
`--mode train` times the configs[2] step instead (local batch 32 x 2 windows: fwd + bwd + bucketed gradient all-reduce +
fused Adam) and adds allreduce_ms / overlap_frac.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (importing torch does not initialise the GPU)

PEAK_MFMA_TFLOPS = 2500.0   # dense bf16 / f16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32 (same table)
PEAK_HBM_GBS = 8000.0
FLOP_PER_FRAME = 0.6512e9   # MSMD.forward from raw audio = 65.12 GFLOP per 100-frame clip
DENOISER_FLOP = 7.886e9     # per call and sequence
TRAIN_FLOP_PER_SAMPLE = 317e9
HUBERT_LARGE_FLOP_PER_CLIP = 384e9
LBS_BYTES_PER_FRAME = 60936


# ----------------------------------------------------------------------------------------------- workload
def synth_batch(B, rank, device):
    """SURVEY.md 8(d) inputs: z-normalised pseudo-gaussian audio, motion, zero shape, style, ones indicator."""
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
    """The step as ONE hipGraph replay (MSMD.capture_forward): static shapes, every replay runs all kernels on the batch
    that sits in the captured input buffers (written there once before the timed region: the contract's "inputs already
    resident in HBM" -- a loader's host-to-device copies would target these buffers); capture_forward checks a replay on
    perturbed inputs against the eager forward, bit for bit, and main() checks a replay of THIS graph against the oracle."""
    ts = torch.tensor(b["time_step"], device=b["audio"].device, dtype=torch.long)
    run = model.capture_forward(b["motion"], b["audio"], b["shape"], b["style"], ts, b["indicator"], b["eps"])
    run(motion_feat=b["motion"], audio=b["audio"], shape_feat=b["shape"], style_feat=b["style"], time_step=ts,
        indicator=b["indicator"], eps=b["eps"])
    return lambda: run()


# ----------------------------------------------------------------------------------------------- roofline
def event_pair_overhead_us():
    """Overhead of ONE HIP event pair around ONE launch, measured on a kernel that reports its own duration."""
    from msmd_amd import _lib
    lib = _lib.load()
    ticks = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    over = []
    for us in (20.0, 60.0) * 12:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.msmd_spin_us(us, ticks.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        over.append(e0.elapsed_time(e1) * 1e3 - float(ticks.item()) / 100.0)
    over.sort()
    return over[len(over) // 2]


def roofline_leg(model, b, mode, steps=3, run_step=None, exclude=None, traffic_key=None):
    """Per-launch HIP-event timing of every msmd_gemm launch of `steps` eager steps; the dominant kernel is the
    128x128-tile kernel of the mode (bf16: gemm2_kernel, f16x2: gemm2s_kernel; fp32: gemm_kernel<float>).
    `run_step` (default: the forward step on batch b) is what gets traced: --mode train passes its eager iteration, the
    sampler leg its eager denoising steps.  `exclude(M, N, K, batch)` drops launches that are not part of a step (the
    sampler's hoisted once-per-call projections); `traffic_key` names the PMC summary to read `traffic` from."""
    from msmd_amd import ops
    step_fn = run_step or (lambda: step(model, b))
    step_fn()
    torch.cuda.synchronize()
    overhead = event_pair_overhead_us() * 1e-3   # ms
    ops.GEMM_TRACE = []
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    flops = ms = big_f = big_ms = tall_f = tall_ms = t256_f = t256_ms = 0.0
    n_launch = big_n = tall_n = t256_n = 0
    kq = 32 if mode == "f16x2" else 64
    for (M, N, K, batch, dt, e0, e1, on_256) in trace:
        if exclude is not None and exclude(M, N, K, batch):
            continue
        f = 2.0 * M * N * K * batch
        d = max(e0.elapsed_time(e1) - overhead, 1e-4)
        flops += f
        ms += d
        n_launch += 1
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
        if on_256:      # the 256 x 256 8-phase kernel (csrc/gemm.hip gemm8_kernel; ops.gemm / gemm_ln record the library's routing)
            t256_f += f
            t256_ms += d
            t256_n += 1
            continue
        if N > 64 and tiles >= 192 and (mode == "fp32" or K % kq == 0):
            # csrc/gemm.hip routes tall 16-bit grids (M >= 16000, >= 400 tiles of 192 x 128) to the 192 x 128 sibling of the
            # 128 x 128 kernel: a different kernel name in the rocprofv3 summary, so it is reported beside, not inside
            if mode in ("bf16", "fp16") and M >= 16000 and ((M + 191) // 192) * ((N + 127) // 128) >= 400:
                tall_f += f
                tall_ms += d
                tall_n += 1
                continue
            big_f += f
            big_ms += d
            big_n += 1
    peak = PEAK_F32_TFLOPS if mode == "fp32" else PEAK_MFMA_TFLOPS
    achieved = big_f / (big_ms * 1e-3) / 1e12 if big_ms > 0 else 0.0
    kernel = {"bf16": "gemm2_kernel<bf16,128,128,4,2,2,pipelined>", "fp16": "gemm2_kernel<f16,128,128,4,2,2,pipelined>",
              "f16x2": "gemm2s_kernel<128,128,4,2,2> (three f16 MFMAs per algorithmic product)",
              "fp32": "gemm_kernel<float,128,128> (v_mfma_f32_16x16x4_f32)"}[mode]
    out = dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
               traffic=pmc_traffic(mode, traffic_key), kernel=kernel + " (csrc/gemm.hip)", launches_per_step=big_n // steps,
               gflop_per_step=round(big_f / steps / 1e9, 1), ms_per_step_in_kernel=round(big_ms / steps, 3),
               avg_launch_us=round(big_ms / max(big_n, 1) * 1e3, 2),
               all_gemm_tflops=round(flops / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0,
               all_gemm_launches_per_step=n_launch // steps, all_gemm_ms_per_step=round(ms / steps, 3),
               event_pair_overhead_us=round(overhead * 1e3, 2),
               note="achieved = algorithmic 2MNK of the dominant kernel's launches / their HIP-event durations minus the "
                    "calibrated event-pair overhead (msmd_spin_us); rocprofv3 summary of this command: profiles/; "
                    "traffic is a constant from this round's --pmc passes in profiles/, not this run")
    if mode == "f16x2":
        out["mfma_issue_frac"] = round(3.0 * achieved / peak, 4)   # the MFMA pipe executes 3 products per algorithmic one
    if t256_n:
        t2 = t256_f / (t256_ms * 1e-3) / 1e12
        out["tile_256x256"] = dict(kernel="gemm8_kernel<%s> (csrc/gemm.hip: 256 x 256 tiles, 8-phase schedule, one workgroup per CU; the conv "
                                          "stack, the encoder QKV projections and every launch whose rounds of 256 tiles fill)" % mode,
                                   launches_per_step=t256_n // steps, achieved=round(t2, 2), frac=round(t2 / peak, 4),
                                   gflop_per_step=round(t256_f / steps / 1e9, 1), ms_per_step_in_kernel=round(t256_ms / steps, 3),
                                   avg_launch_us=round(t256_ms / t256_n * 1e3, 2), traffic=pmc_traffic_256(traffic_key))
        mf = (big_f + tall_f + t256_f) / ((big_ms + tall_ms + t256_ms) * 1e-3) / 1e12
        out["all_mfma_tile_kernels"] = dict(achieved=round(mf, 2), frac=round(mf / peak, 4),
                                            launches_per_step=(big_n + tall_n + t256_n) // steps,
                                            ms_per_step_in_kernel=round((big_ms + tall_ms + t256_ms) / steps, 3))
    if tall_n:
        ta = tall_f / (tall_ms * 1e-3) / 1e12
        fam = (big_f + tall_f) / ((big_ms + tall_ms) * 1e-3) / 1e12
        # both tile shapes of the one kernel template together (what rounds 1-2 reported as the dominant kernel's figure)
        out["both_tiles"] = dict(achieved=round(fam, 2), frac=round(fam / peak, 4), launches_per_step=(big_n + tall_n) // steps,
                                 ms_per_step_in_kernel=round((big_ms + tall_ms) / steps, 3))
        out["tile_192x128"] = dict(kernel=kernel.replace("128,128", "192,128") + " (csrc/gemm.hip; M >= 16000: the conv stack)",
                                   launches_per_step=tall_n // steps, achieved=round(ta, 2), frac=round(ta / peak, 4),
                                   ms_per_step_in_kernel=round(tall_ms / steps, 3), avg_launch_us=round(tall_ms / tall_n * 1e3, 2))
    return out


def pmc_traffic(mode, key=None, tile="128ELi128E"):
    """HBM bytes per launch of the dominant kernel from this round's committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in SEPARATE runs; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  PMC counters
    cannot be read from inside this process: the figure is a constant from profiles/ (null when absent), not this run.
    key: which workload's summary (None = the forward step; "sampler" = profiles/r05_pmc_sampler_fetch_write_per_kernel.json)."""
    names = ("r06_pmc_hbm_fetch_write_per_kernel.json", "r05_pmc_hbm_fetch_write_per_kernel.json", "r04_pmc_hbm_fetch_write_per_kernel.json", "r03_pmc_hbm_fetch_write_per_kernel.json",
             "r02_pmc_hbm_fetch_write_per_kernel.json")
    if key is not None:
        names = (f"r06_pmc_{key}_fetch_write_per_kernel.json", f"r05_pmc_{key}_fetch_write_per_kernel.json")
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
            want = {"bf16": "gemm2_kernel", "fp16": "gemm2_kernel", "f16x2": "gemm2s_kernel", "fp32": "gemm_kernel"}[mode]
            # (the multi-round launches of the same tile run as gemm2p_kernel, its persistent form)
            tiles = (tile, "192ELi128E") if key == "sampler" else (tile,)      # the sampler's family spans both tile shapes
            ks = [v for n, v in d.items() if any(w + "I" in n.replace("<", "I") for w in (want, want.replace("gemm2_", "gemm2p_")))
                  and any(t in n.replace(", ", "ELi").replace("<", "I") for t in tiles)]
            ks = ks or [v for n, v in d.items() if want in n and "128" in n and ("gemm2s" in n) == (mode == "f16x2")]
            if ks:
                # since round 4 the 128 x 128 kernel is one symbol per epilogue family: launch-weighted mean over them
                n = sum(v.get("launches", 0) for v in ks) or 1
                return round(sum(v.get("launches", 0) * (2.0 * v["fetch_kb_avg"] + v["write_kb_avg"]) for v in ks) / n * 1024.0)
        except Exception:
            pass
    return None


def pmc_traffic_256(key=None):
    """HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, launch-weighted over its epilogue forms) of gemm8_kernel from the same
    committed --pmc passes as pmc_traffic; None when the summary has no such kernel."""
    path = next((q for q in (os.path.join(ROOT, "profiles", (f"{r}_pmc_{key}_fetch_write_per_kernel.json" if key else f"{r}_pmc_hbm_fetch_write_per_kernel.json"))
                             for r in ("r06", "r05")) if os.path.exists(q)), "")
    try:
        ks = [v for n, v in json.load(open(path)).items() if "gemm8_kernel" in n]
        n = sum(v.get("launches", 0) for v in ks)
        return round(sum(v.get("launches", 0) * (2.0 * v["fetch_kb_avg"] + v["write_kb_avg"]) for v in ks) / n * 1024.0) if n else None
    except Exception:
        return None


# ----------------------------------------------------------------------------------------------- CPU baseline / oracle
def cpu_baseline_leg(B, want_lbs=True):
    """oracle/torch_cpu.py on this host (reference arithmetic: fp32 torch CPU, eval mode).  Returns (json dict, the
    forward's `target` (B, 110, 67) as the in-run checker of the GPU modes)."""
    from msmd_amd import shapes, synth
    from msmd_amd.config import synthetic_args
    from oracle import diffusion as od, flame as ofl, torch_cpu as tc
    ncpu = os.cpu_count() or 1
    args = synthetic_args()
    sd = tc.to_torch(synth.fill_state_dict(shapes.msmd_shapes(args)))
    sched = od.diffusion_schedule(500, "cosine")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).float()
    audio, motion = t(synth.audio_clips(B, 64000, tag="bench_audio_r0")), t(synth.motion_clips(B, tag="bench_motion_r0"))
    style, eps = t(synth.normalish("bench_style_r0", (B, 256))), t(synth.normalish("bench_eps_r0", (B, 100, 67)))
    ts = [(37 * i + 11) % 500 + 1 for i in range(B)]
    shape, ind = torch.zeros(B, 100), torch.ones(B, 100)

    def med(fn, warm=2, reps=5):
        for _ in range(warm):
            out = fn()
        tt = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            tt.append(time.perf_counter() - t0)
        return sorted(tt)[len(tt) // 2], out
    # torch's intra-op pool does not scale to every core of a 2-socket host on this path (128 threads ran SLOWER than 16
    # on the EPYC 9575F box): scan thread counts on an 8-clip slice and time the batch at the best one
    scan = {}
    for thr in [c for c in (8, 16, 32, 64) if c <= ncpu] or [ncpu]:   # wider pools only ever ran slower (and cost minutes)
        torch.set_num_threads(thr)
        sl = slice(0, min(8, B))
        scan[thr], _ = med(lambda: tc.msmd_forward(sd, sched, motion[sl], audio[sl], shape[sl], style[sl], ts[:sl.stop],
                                                   eps[sl], ind[sl]), 1, 1)
    threads = min(scan, key=scan.get)
    torch.set_num_threads(threads)
    t_fwd, (_, target, _) = med(lambda: tc.msmd_forward(sd, sched, motion, audio, shape, style, ts, eps, ind))
    af = t(synth.normalish("bench_cpu_af", (B, 100, 512)))
    t_den, _ = med(lambda: tc.denoise_step(sd, motion, af, shape, style, 250, ind), 1, 3)
    out = dict(value=round(B * 100 / t_fwd, 1), unit="frames/s", cores=int(threads), kind="port",
               sample=f"oracle/torch_cpu.py msmd_forward (torch fp32 CPU restatement of the reference path, pinned to the "
                      f"reference goldens) on the SAME {B}-clip batch: median of 5 after 2 warm-ups = {t_fwd:.2f} s; "
                      f"host {ncpu} logical CPUs, torch threads {threads} (best of a scan on an 8-clip slice: "
                      + ", ".join(f"{k}: {min(8, B) * 100 / v:.0f} f/s" for k, v in scan.items()) + ")",
               denoise_step_3entries=dict(ms=round(t_den * 1e3, 1), frames_per_s_at_T500=round(B * 100 / (t_den * 500), 2),
                                          sample=f"one sampler loop body on 3 x {B} sequences, median of 3"))
    if want_lbs:
        fl = tc.FlameTorch(ofl.FlameOracle(synth.flame_asset()))
        n = 800
        g = torch.Generator().manual_seed(7)
        ex, po = 0.5 * torch.randn(n, 50, generator=g), 0.2 * torch.randn(n, 6, generator=g)
        t_lbs, _ = med(lambda: fl.forward(torch.zeros(n, 100), ex, po), 1, 3)
        out["lbs"] = dict(frames_per_s=round(n / t_lbs, 1), sample=f"{n} frames, FLAME vertices only, median of 3")
    return out, target.numpy()


def cpu_train_baseline(n=2):
    """Training-step baseline on the host: forward + backward of BOTH windows of `n` samples through oracle/torch_cpu.py
    (the torch-CPU fp32 restatement of the reference modules, autograd by torch) with a plain MSE on the predicted
    sample; the frozen conv feature extractor gets no gradient, as in the reference (model.py:97).  The style encoder,
    the loss terms and Adam (< 2 % of the step's FLOPs) are left out, which flatters the CPU."""
    from msmd_amd import shapes, synth
    from msmd_amd.config import synthetic_args
    from oracle import diffusion as od, torch_cpu as tc
    ncpu = os.cpu_count() or 1
    threads = min(64, ncpu)
    torch.set_num_threads(threads)
    args = synthetic_args()
    sd = tc.to_torch(synth.fill_state_dict(shapes.msmd_shapes(args)))
    for k, v in sd.items():
        if v.is_floating_point() and "feature_extractor" not in k:
            v.requires_grad_(True)
    sched = od.diffusion_schedule(500, "cosine")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).float()
    audio = [t(synth.audio_clips(n, 64000, tag=f"train_r0_it0_a{i}")) for i in range(2)]
    motion = [t(synth.motion_clips(n, tag=f"train_r0_it0_m{i}")) for i in range(2)]
    style, eps = t(synth.normalish("bench_style_r0", (n, 256))), t(synth.normalish("bench_eps_r0", (n, 100, 67)))
    ts = [(37 * i + 11) % 500 + 1 for i in range(n)]
    shape, ind = torch.zeros(n, 100), torch.ones(n, 100)

    fwd = getattr(tc.msmd_forward, "__wrapped__", tc.msmd_forward)   # the oracle's forward without its no_grad wrapper

    def it():
        for v in sd.values():
            v.grad = None
        loss = 0.0
        for w in range(2):
            _, target, _ = fwd(sd, sched, motion[w], audio[w], shape, style, ts, eps, ind)
            loss = loss + ((target[:, -100:] - motion[w]) ** 2).mean()
        loss.backward()
    it()
    tt = []
    for _ in range(3):
        t0 = time.perf_counter()
        it()
        tt.append(time.perf_counter() - t0)
    dt = sorted(tt)[1]
    return dict(value=round(n * 200 / dt, 1), unit="frames/s", cores=int(threads), kind="port",
                sample=f"oracle/torch_cpu.py forward + torch-autograd backward of both windows of {n} samples (conv feature "
                       f"extractor frozen; no style encoder / loss terms / Adam: < 2 % of the FLOPs): median of 3 = {dt:.2f} s; "
                       f"host {ncpu} logical CPUs, {threads} torch threads")


# ----------------------------------------------------------------------------------------------- legs (other configs)
class _LazyNoise:
    """noise[t] for the sampler's injected-noise path: the SAME pseudo-random tensor for a given (seed, t) whoever asks."""

    def __init__(self, shape, device, seed):
        self.shape, self.device, self.seed = shape, device, seed

    def __getitem__(self, t):
        g = torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + int(t))
        return torch.randn(self.shape, generator=g, device=self.device)


def sampler_roofline(model, af, shape, style, ind, xT, device, steps=6):
    """`roofline` block of the sampler's dominant kernel: per-launch HIP-event durations of the GEMMs of `steps` EAGER denoising
    steps on ONE LANE's share of the batch (the hipGraph loop runs msmd_amd.sampler.LANES groups of clips side by side, each
    with the launches timed here: B / lanes clips x 3 CFG entries = 96 sequences x 111 rows at the default).  The hoisted
    once-per-call projections (K / V of the audio memory, N = 1024) are not part of a step and are left out."""
    from msmd_amd import sampler as smp
    from msmd_amd.model import DiffusionSchedule
    B = af.shape[0]
    lanes = getattr(model, "sampler_lanes", smp.LANES)
    while lanes > 1 and (B % lanes or 3 * B // lanes < smp.MIN_LANE_SEQS):
        lanes -= 1
    Bl = B // lanes
    sched = model.diffusion_sched
    model.diffusion_sched = DiffusionSchedule(steps, "cosine").to(device)
    noise = _LazyNoise((Bl, 100, 67), device, 78)
    from msmd_amd import ops
    tile_keep, ops.GEMM_LN_TILE = ops.GEMM_LN_TILE, (15 if lanes > 1 else ops.GEMM_LN_TILE)   # the multi-lane graph's tile choice
    try:
        def run():
            model.sample(af[:Bl], shape[:Bl], style[:Bl], motion_at_T=xT[:Bl], indicator=ind[:Bl], cfg_scale=1.15, noise=noise)
        r = roofline_leg(model, None, "fp16", steps=1, run_step=run, exclude=lambda M, N, K, batch: N == 1024 or batch > 1,
                         traffic_key="sampler")
    finally:
        model.diffusion_sched = sched
        ops.GEMM_LN_TILE = tile_keep
    fam = r.get("both_tiles") or r
    out = dict(bound="mfma", achieved=fam["achieved"], peak=r["peak"], unit="TFLOP/s", frac=fam["frac"], traffic=r["traffic"],
               kernel=("gemm2_kernel<f16,{192|128},128,4,2,2,pipelined> + the persistent form gemm2p_kernel (csrc/gemm.hip): the decoder "
                       "layers' QKV / out-projection / FFN-2 (192 x 128 tile beside another lane) and FFN-1 (128 x 128) GEMMs of one lane"),
               launches_per_step=fam["launches_per_step"] // steps, ms_per_step_in_kernel=round(fam["ms_per_step_in_kernel"] / steps, 4),
               avg_launch_us=round(fam["ms_per_step_in_kernel"] * 1e3 / max(1, fam["launches_per_step"]), 2),
               sequences_per_lane=3 * Bl, lanes=lanes,
               all_gemm_tflops=r["all_gemm_tflops"], all_gemm_launches_per_step=r["all_gemm_launches_per_step"] // steps,
               all_gemm_ms_per_step=round(r["all_gemm_ms_per_step"] / steps, 4), event_pair_overhead_us=r["event_pair_overhead_us"],
               note="HIP-event durations of eager launches of ONE lane alone on the chip (the graph loop runs `lanes` of them side by "
                    "side, so a rocprofv3 trace of the loop shows longer per-launch times at a higher chip throughput); "
                    "rocprofv3 summaries: profiles/r05_sampler_*; traffic = 2 x FETCH_SIZE + WRITE_SIZE of this kernel from the "
                    "sampler's own --pmc passes (profiles/r05_pmc_sampler_fetch_write_per_kernel.json), null when absent")
    return out


def leg_sampler(device, B=64, T=500):
    """configs[4]: sample() B=64, T=500, 3 CFG entries.  Timed in fp16 storage (the config's dtype) AND in the parity-grade
    f16x2 mode, each as hipGraph replays; `mfma_frac` divides the reference-NOMINAL FLOPs (3 x 7.886 GFLOP per sequence and
    step) by the time, `mfma_frac_executed` the FLOPs of the GEMMs the loop actually launches (the diagonal
    cross-attention fast path and the hoisted step-invariant work remove ~30 % of the nominal ones); `fp16_vs_f16x2` is the
    drift of the fp16 sampler against the f16x2 one over all 500 steps under IDENTICAL injected noise (eager loops)."""
    from msmd_amd import ops, synth
    from msmd_amd.config import synthetic_args
    from msmd_amd.model import get_diffusion_model
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    af, style = t(synth.normalish("leg_samp_af", (B, 100, 512))), t(synth.normalish("leg_samp_style", (B, 256)))
    shape, ind = torch.zeros(B, 100, device=device), torch.ones(B, 100, device=device)
    xT = t(synth.normalish("leg_samp_xT", (B, 100, 67)))
    noise = _LazyNoise((B, 100, 67), device, 77)
    out, x0 = {}, {}
    for dtype in ("fp16", "f16x2"):
        model = get_diffusion_model(synthetic_args(compute_dtype=dtype), device).eval()
        model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)      # warm-up: packs, captures the step graph
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, _, _ = model.sample(af, shape, style, indicator=ind, cfg_scale=1.15)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ops.GEMM_FLOPS = [0.0]
        xe, _, _ = model.sample(af, shape, style, motion_at_T=xT, indicator=ind, cfg_scale=1.15, noise=noise)   # eager, injected noise
        torch.cuda.synchronize()
        executed, ops.GEMM_FLOPS = ops.GEMM_FLOPS[0], None
        x0[dtype] = xe.float()
        tf = B * 3 * DENOISER_FLOP * T / dt / 1e12
        roof = None
        if dtype == "fp16":
            roof = sampler_roofline(model, af, shape, style, ind, xT, device)
        out[dtype] = dict(config=f"configs[4]: sample() B={B}, T={T} DDPM steps x 3 CFG entries, {dtype}, one hipGraph replay per step",
                          ms_per_step=round(dt / T * 1e3, 3), frames_per_s=round(B * 100 / dt, 1), tflops_nominal=round(tf, 1),
                          mfma_frac=round(tf / PEAK_MFMA_TFLOPS, 4),
                          gemm_tflop_executed_per_call=round(executed / 1e12, 2),
                          mfma_frac_executed=round(executed / dt / 1e12 / PEAK_MFMA_TFLOPS, 4),
                          finite=bool(torch.isfinite(x).all() and torch.isfinite(xe).all()))
        if roof is not None:
            out[dtype]["roofline"] = roof
        lanes = next(iter(model._step_graphs.values())).lanes if getattr(model, "_step_graphs", None) else 1
        out[dtype]["lanes"] = lanes
        del model
        torch.cuda.empty_cache()
    r = out["fp16"]
    r["f16x2"] = {k: out["f16x2"][k] for k in ("ms_per_step", "frames_per_s", "mfma_frac", "mfma_frac_executed", "finite")}
    r["f16x2"]["note"] = "parity-grade mode (fp32-grade contractions: three f16 MFMAs per product); mfma_frac counts algorithmic FLOPs"
    r["fp16_vs_f16x2"] = dict(max_abs_x0=float((x0["fp16"] - x0["f16x2"]).abs().max()),
                              rms_x0=float((x0["fp16"] - x0["f16x2"]).pow(2).mean().sqrt()),
                              max_abs_ref=float(x0["f16x2"].abs().max()),
                              sample=f"x_0 after all {T} steps, same x_T and the same injected noise at every step, all {B} sequences")
    return r


def leg_lbs(device, frames_list=(6400, 25600)):
    """FLAME blendshapes + LBS (5023 vertices) on N frames: algorithmic bytes 60 936 per frame."""
    from types import SimpleNamespace
    from msmd_amd import synth
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    fl = FLAME(cfg).to(device)
    out = {}
    only = os.environ.get("MSMD_BENCH_LBS")      # profiling runs: "6400" | "25600" | "shape" = one workload per process
    if only in ("6400", "25600"):
        frames_list = (int(only),)
    for n in (() if only in ("shape", "25600_fp16") else frames_list):
        g = torch.Generator(device="cpu").manual_seed(n)
        exp = (0.5 * torch.randn(n, 50, generator=g)).to(device)
        pose = (0.2 * torch.randn(n, 6, generator=g)).to(device)
        shape = torch.zeros(n, 100, device=device)
        for _ in range(3):
            fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 20
        e0.record()
        for _ in range(R):
            fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / R
        gbs = n * LBS_BYTES_PER_FRAME / ms / 1e6
        out[f"lbs_{n}"] = dict(ms=round(ms, 4), frames_per_s=round(n / ms * 1e3), gb_per_s=round(gbs, 1),
                               hbm_frac=round(gbs / PEAK_HBM_GBS, 4), precision=fl.lbs_precision or "bf16x3 (split-bf16 MFMA, fp32 accumulate)",
                               bytes_per_frame=LBS_BYTES_PER_FRAME,
                               inputs="shape = 0 for every frame (SURVEY 8d): the one-subject fold of msmd_flame_prepare")
    if only in (None, "25600_fp16"):
        # opt-in fp16 vertices (msmd_lbs_skin_v2_f16; BASELINE configs[4] names the fp16 LBS pass): rows of 5024 x 3 halves
        n = 25600
        g = torch.Generator(device="cpu").manual_seed(n)
        exp = (0.5 * torch.randn(n, 50, generator=g)).to(device)
        pose = (0.2 * torch.randn(n, 6, generator=g)).to(device)
        shape = torch.zeros(n, 100, device=device)
        fl.vertex_dtype = torch.float16
        try:
            for _ in range(3):
                fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
            e1.record()
            torch.cuda.synchronize()
        finally:
            del fl.vertex_dtype
        ms = e0.elapsed_time(e1) / 20
        bpf = 4 * 165 + 2 * 3 * 5024
        out["lbs_25600_fp16_vertices"] = dict(ms=round(ms, 4), frames_per_s=round(n / ms * 1e3), gb_per_s=round(n * bpf / ms / 1e6, 1),
                                              hbm_frac=round(n * bpf / ms / 1e6 / PEAK_HBM_GBS, 4), bytes_per_frame=bpf,
                                              bound="|v16 - v32| <= 2^-11 |v32| + 2^-10 sum_k |coef_k dirs_k| (fp16 operand planes: one MFMA per K group; "
                                                    "FLAME.vertex_exact = True gives the fp32 kernel's vertex rounded once, slower)",
                                              inputs="as lbs_25600; vertices stored as fp16 in rows of 5024 x 3 (opt-in FLAME.vertex_dtype)")
    if only in ("6400", "25600", "25600_fp16"):
        return out
    # the general path the reference's lbs() computes (utils/lbs.py:185): a different shape vector per frame
    n = frames_list[-1]
    g = torch.Generator(device="cpu").manual_seed(n + 1)
    exp = (0.5 * torch.randn(n, 50, generator=g)).to(device)
    pose = (0.2 * torch.randn(n, 6, generator=g)).to(device)
    shape = (0.3 * torch.randn(n, 100, generator=g)).to(device)
    for _ in range(3):
        fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fl(shape, exp, pose, return_lm2d=False, return_lm3d=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gbs = n * (LBS_BYTES_PER_FRAME + 400) / ms / 1e6
    out[f"lbs_{n}_per_frame_shape"] = dict(ms=round(ms, 4), frames_per_s=round(n / ms * 1e3), gb_per_s=round(gbs, 1),
                                           hbm_frac=round(gbs / PEAK_HBM_GBS, 4), bytes_per_frame=LBS_BYTES_PER_FRAME + 400,
                                           inputs="a different 100-d shape vector per frame: all 66 MFMAs per tile (no fold)")
    return out


def leg_rotations_landmarks(device, N=1 << 22, frames=25600):
    """The small HBM-bound kernels north_star names beside the skinning: rotation conversions (utils/rotation_conversions.py:
    one item per thread, 12-64 B per item) and the 68 landmarks + dynamic-contour LUT row (utils/lbs.py:102-138,
    utils/flame.py:126-172) -- HIP-event time per launch against algorithmic bytes (items x (bytes in + bytes out))."""
    from types import SimpleNamespace
    from msmd_amd import ops, synth
    from msmd_amd.utils import rotation_conversions as rc
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    g = torch.Generator(device="cpu").manual_seed(0)
    aa = torch.randn(N, 3, generator=g).to(device)
    R = rc.axis_angle_to_matrix(aa)
    q = rc.matrix_to_quaternion(R)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    for name, fn, nbytes in (("axis_angle_to_matrix", lambda: rc.axis_angle_to_matrix(aa), 12 + 36),
                             ("matrix_to_quaternion", lambda: rc.matrix_to_quaternion(R), 36 + 16),
                             ("quaternion_to_matrix", lambda: rc.quaternion_to_matrix(q), 16 + 36),
                             ("matrix_to_rotation_6d", lambda: rc.matrix_to_rotation_6d(R), 36 + 24)):
        ms = timed(fn)          # includes the output allocation of the Python wrapper (caching allocator: no device call)
        gbs = N * nbytes / ms / 1e6
        out[name] = dict(items=N, us=round(ms * 1e3, 1), bytes_per_item=nbytes, gb_per_s=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4))
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    fl = FLAME(cfg).to(device)
    exp, pose = (0.5 * torch.randn(frames, 50, generator=g)).to(device), (0.2 * torch.randn(frames, 6, generator=g)).to(device)
    verts, _, _ = fl(torch.zeros(frames, 100, device=device), exp, pose, return_lm2d=False, return_lm3d=False)
    p = fl._pack()
    ms = timed(lambda: ops.landmarks(verts, p["faces"], p["full_idx"], fl.full_lmk_bary_coords))
    nbytes = 68 * (3 * 12 + 12)       # 3 gathered vertices + the written landmark (+ the per-call 68-entry tables, cache-resident)
    gbs = frames * nbytes / ms / 1e6
    out["landmarks_68"] = dict(frames=frames, us=round(ms * 1e3, 1), bytes_per_frame=nbytes, gb_per_s=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4),
                               note="gathers 204 vertices of 5023 per frame: 36-byte pieces of 60 KB rows, sector-granular traffic is ~5x the algorithmic bytes")
    return {"rotations_landmarks": out}


def leg_attention(device):
    """The two attention launches of the path in isolation (csrc/attention.hip attn_whole_kernel: softmax(Q K^T) V per sequence and
    head in one workgroup pass): the encoder's (configs[1]: B = 32, 12 heads, T = 200, bf16) and the denoiser's self-attention as
    the sampler launches it per lane (configs[4]: 96 sequences x 8 heads, 111 tokens, fp16).  Against BOTH roofs: FLOPs =
    4 B H T^2 64 (the two contractions), bytes = the fused Q | K | V rows read + the O rows written."""
    from msmd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    for name, B, H, T, dt in (("encoder_b32_h12_t200_bf16", 32, 12, 200, torch.bfloat16), ("denoiser_n96_h8_t111_fp16", 96, 8, 111, torch.float16),
                              ("denoiser_n192_h8_t111_fp16", 192, 8, 111, torch.float16)):
        d = H * 64
        qkv = torch.randn(B, T, 3 * d, device=device, generator=g).to(dt)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        for _ in range(3):
            ops.attention(q, k, v, H, 0.125)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(50):
                ops.attention(q, k, v, H, 0.125)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        flop, nbytes = 4.0 * B * H * T * T * 64, B * T * 4 * d * 2
        out[name] = dict(us=round(best * 1e3, 2), tflops=round(flop / best / 1e9, 1), mfma_frac=round(flop / best / 1e9 / PEAK_MFMA_TFLOPS, 4),
                         gb_per_s=round(nbytes / best / 1e6, 1), hbm_frac=round(nbytes / best / 1e6 / PEAK_HBM_GBS, 4),
                         bound="hbm (rows read once, 64 FLOP per byte at T = 200: below the machine balance of ~310)")
    return {"attention": out}


def leg_train(device, B=32, steps=5):
    from msmd_amd.config import synthetic_args
    from msmd_amd.model import get_diffusion_model
    from msmd_amd.style_encoder import get_style_encoder
    from msmd_amd.training_script import Trainer, synthetic_batch
    args = synthetic_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
    model = get_diffusion_model(args, device).train()
    se = get_style_encoder(args, "vae2").to(device).train()
    tr = Trainer(args, model, se, use_graph=True)
    batch = synthetic_batch(B, 0, device)
    tr.capture_all(batch)
    for _ in range(4):
        tr.step(batch, it=1)
    torch.cuda.synchronize()
    # four timed stretches of `steps` iterations, the median stretch reported: the host draws truncation patterns and SpecAugment
    # masks per iteration (as the reference does), so a stretch's mix of graph variants and the box's host speed both move a
    # 5-iteration average by 10 % (29.5 / 33.5 ms on two boxes of one afternoon; `--mode train` with its longer run: 28.5 on the second)
    dts = []
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(batch, it=1)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / steps)
    dt = sorted(dts)[1]
    tf = TRAIN_FLOP_PER_SAMPLE * B / dt / 1e12
    return dict(config=f"configs[2] per-GPU shape: local batch {B} x 2 windows, fwd + bwd + fused Adam, bf16, train-mode "
                       f"noise on, whole-iteration hipGraph", ms_per_step=round(dt * 1e3, 2),
                frames_per_s=round(B * 200 / dt), tflops=round(tf, 1), mfma_frac=round(tf / PEAK_MFMA_TFLOPS, 4))


def leg_hubert_large(device, B=32, steps=5):
    from msmd_amd import synth
    from msmd_amd.config import synthetic_args
    from msmd_amd.model import get_diffusion_model
    args = synthetic_args(audio_model="hubert_large", compute_dtype="bf16", n_motions=250)
    model = get_diffusion_model(args, device).eval()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    audio = t(synth.audio_clips(B, 160000, tag="hl_bench"))
    motion, style = t(synth.normalish("hl_motion", (B, 250, 67))), t(synth.normalish("hl_style", (B, 256)))
    eps = t(synth.normalish("hl_eps", (B, 250, 67)))
    ts = [(37 * i + 11) % 500 + 1 for i in range(B)]
    shape, ind = torch.zeros(B, 100, device=device), torch.ones(B, 250, device=device)
    run = lambda: model(motion, audio, shape, style, time_step=ts, indicator=ind, train_with_CFG=False, eps=eps)
    launch = "eager"
    for _ in range(2):
        run()
    try:
        tsd = torch.tensor(ts, device=device, dtype=torch.long)
        cap = model.capture_forward(motion, audio, shape, style, tsd, ind, eps)
        run, launch = (lambda: cap.graph.replay()), "one hipGraph replay per step"
    except Exception as e:
        print(f"[bench] hubert-large leg: hipGraph capture unavailable ({type(e).__name__}: {e}); eager", file=sys.stderr)
        torch.cuda.synchronize()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = B * HUBERT_LARGE_FLOP_PER_CLIP / dt / 1e12
    return dict(config=f"configs[3]: HuBERT-large encoder (24 layers, 1024 wide), B={B} x 10 s clips, bf16, MSMD.forward, {launch}",
                ms_per_step=round(dt * 1e3, 2), frames_per_s=round(B * 250 / dt), encoder_tflops=round(tf, 1),
                mfma_frac=round(tf / PEAK_MFMA_TFLOPS, 4))


def run_legs(device, which="all"):
    """`which`: "all" or a comma-separated subset of sampler,lbs,train,hubert (profiling runs time one leg at a time)."""
    legs = {}
    for name, fn in (("sampler_b64_t500", leg_sampler), ("lbs", leg_lbs), ("rotations_landmarks", leg_rotations_landmarks),
                     ("attention", leg_attention), ("train_step_b32", leg_train), ("hubert_large_10s_b32", leg_hubert_large)):
        if which != "all" and not any(name.startswith(w) for w in which.split(",")):
            continue
        try:
            r = fn(device)
            if name in ("lbs", "rotations_landmarks", "attention"):
                legs.update(r)
            else:
                legs[name] = r
        except Exception as e:   # a leg must never take the headline line down with it
            legs[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return legs


def multi_stream_forward(model, a, rank, device, S):
    """S independent 32-clip steps in flight: one captured hipGraph + input buffers + batch per HIP stream, replays issued
    round-robin.  Every step's output (warm-up and timed) is compared ON THE DEVICE, bit for bit, with that batch's
    one-at-a-time result (three small kernels per step on the step's own stream, inside the timed region).  Round 2 saw
    10-40 % of two-stream passes differ from serial ones; round 3 traced that to packed-fp32 VALU math (first wrong op: the
    conv0 + GroupNorm + GELU kernel, low halves of v_pk_* results in lanes 48-63) and the library is now built without it
    (csrc/Makefile) -- this leg keeps verifying every step.
    -> dict(ms_per_step, frames_per_s, steps, mismatching_streams) or None when capture fails."""
    from msmd_amd import dp
    try:
        b0 = synth_batch(a.batch, rank, device)
        for _ in range(2):
            step(model, b0)
        runs = [graphed_step(model, b0)] + [graphed_step(model, synth_batch(a.batch, 1000 + 16 * rank + i, device)) for i in range(1, S)]
        torch.cuda.synchronize()
        serial = []
        for r in runs:
            o = r()
            torch.cuda.synchronize()
            serial.append(o[1].clone())
        streams = [torch.cuda.Stream() for _ in range(S)]
        flags = [torch.zeros((), dtype=torch.bool, device=device) for _ in range(S)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        state = {"i": 0}

        def run():
            i = state["i"]
            state["i"] = (i + 1) % S
            with torch.cuda.stream(streams[i]):
                o = runs[i]()
                flags[i].logical_or_(torch.ne(o[1], serial[i]).any())
        elapsed = dp.timed_steps(run, a.steps, a.warmup, sync=torch.cuda.synchronize, device=device)
        bad = [i for i in range(S) if bool(flags[i].item())]
        return dict(streams=S, ms_per_step=round(elapsed / a.steps * 1e3, 3), frames_per_s=round(a.batch * 100 * a.steps / elapsed, 1),
                    steps_verified_bit_equal_to_serial=(a.steps + a.warmup) if not bad else 0, mismatching_streams=bad,
                    gemm="the single-stream kernels (no special schedule)",
                    launch=f"{S} steps in flight: hipGraph replays round-robin on {S} HIP streams, one graph + input buffers + batch "
                           f"per stream; every step's output compared on the device with its one-at-a-time result inside the timed region")
    except Exception as e:
        print(f"[bench] multi-stream leg unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        torch.cuda.synchronize()
        return None


# ----------------------------------------------------------------------------------------------- modes
def run_forward(a, rank, world, device):
    from msmd_amd import dp, ops
    for kv in [x for x in a.tune.split(",") if x]:   # developer library only (MSMD_LIB=.../libmsmd_hip_exp.so)
        ops.exp_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    from msmd_amd.config import synthetic_args
    from msmd_amd.model import get_diffusion_model
    args = synthetic_args(compute_dtype=a.dtype)
    model = get_diffusion_model(args, device).eval()
    b = synth_batch(a.batch, rank, device)

    def timed(mode):
        """-> (elapsed s for a.steps steps, launch description, target of rank 0's batch, single-stream ms/step or None).
        A step = one pass over one 32-clip batch.  With --streams S > 1 the steps are issued round-robin onto S HIP
        streams, each with its own captured hipGraph, input buffers and batch: consecutive steps are independent
        requests, so up to S of them are in flight and the under-filled phases of one (300-tile GEMMs, the decoder's
        small grids, attention) run beside another's.  Every timed step still runs every kernel of the path; before
        timing, concurrent replays are checked bit for bit against one-at-a-time replays (else S falls back to 1)."""
        model.set_compute_dtype(mode)
        for _ in range(2):
            out = step(model, b)   # lazy packing / allocator warm-up before any capture
        target = out[1].float().cpu().numpy()
        launch, run, single_ms = "eager", (lambda: step(model, b)), None
        if not a.eager:
            try:
                S = max(1, a.streams)
                runs = [graphed_step(model, b)] + [graphed_step(model, synth_batch(a.batch, 1000 + 16 * rank + i, device))
                                                   for i in range(1, S)]
                run = runs[0]
                launch = "one hipGraph replay per step (the batch sits in the captured input buffers)"
                if S > 1:
                    torch.cuda.synchronize()
                    serial = []
                    for r in runs:
                        o = r()
                        torch.cuda.synchronize()
                        serial.append(o[1].clone())
                    streams = [torch.cuda.Stream() for _ in range(S)]
                    ok = True
                    for _ in range(3):
                        for st in streams:
                            st.wait_stream(torch.cuda.current_stream())
                        outs = []
                        for st, r in zip(streams, runs):
                            with torch.cuda.stream(st):
                                outs.append(r())
                        torch.cuda.synchronize()
                        ok = ok and all(torch.equal(o[1], sref) for o, sref in zip(outs, serial))
                    if os.environ.get("BENCH_DEBUG"):
                        print(f"[bench] mode {mode}: streams {S}, concurrent == serial: {ok}", file=sys.stderr)
                    if ok:
                        single_ms = dp.timed_steps(runs[0], a.steps, a.warmup, sync=torch.cuda.synchronize, device=device) / a.steps * 1e3
                        state = {"i": 0}

                        def run():
                            i = state["i"]
                            state["i"] = (i + 1) % S
                            with torch.cuda.stream(streams[i]):
                                runs[i]()
                        launch = (f"{S} steps in flight: hipGraph replays issued round-robin on {S} HIP streams, one captured graph + "
                                  f"input buffers + batch per stream (each batch written once into its captured buffers; concurrent replays verified "
                                  f"bit-equal to one-at-a-time replays before timing)")
                    else:
                        print("[bench] concurrent replays differ from serial ones; timing one stream", file=sys.stderr)
            except Exception as e:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
                torch.cuda.synchronize()
        elapsed = dp.timed_steps(run, a.steps, a.warmup, sync=torch.cuda.synchronize, device=device)
        if launch != "eager" and max(1, a.streams) == 1:
            # the object that was TIMED (the hipGraph replay), not only the eager step before it, goes to the checker
            o = run()
            torch.cuda.synchronize()
            replay_target[mode] = o[1].float().cpu().numpy()
        return elapsed, launch, target, single_ms
    replay_target = {}
    elapsed, launch, target, single_ms = timed(a.dtype)
    if rank != 0:
        return None
    n = world
    value = a.batch * 100 * a.steps * n / elapsed
    out = {
        "metric": "FLAME frames/sec on 4s@16kHz clips (whole job; MSMD.forward motion-coefficient frames)",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: MSMD.forward, batch=32 x 4 s clips per GPU, wav2vec2-base encoder + "
                               "8-layer motion decoder, eval-mode, synthetic closed-form weights",
                   "batch_per_gpu": a.batch, "clip_seconds": 4, "frames_per_clip": 100,
                   "parallelism": f"dp{n} (independent clips, no collective)", "launch": launch,
                   "inputs": "resident in HBM when the timed region starts (H2D of the 8.2 MB batch excluded); since round 3 the "
                             "batch sits in the captured graph's input buffers, so the per-step D2D refresh of rounds 1-2 "
                             "(about 0.1 ms) is not in the timed replay either: compare ms_per_step across rounds with that in mind"},
        "end_to_end_tflops": round(value * FLOP_PER_FRAME / 1e12 / n, 1),
    }
    if single_ms is not None:
        out["single_stream_ms_per_step"] = round(single_ms, 3)   # one step at a time (latency of a step)
    if n > 1:
        if not a.no_roofline:     # rank 0's dominant kernel (no collective inside: the other ranks wait at the closing barrier)
            out["roofline"] = roofline_leg(model, b, a.dtype)
        out["cpu_baseline"] = None    # timed at N = 1 only
        return out
    ref = None
    if not a.no_cpu_baseline:
        out["cpu_baseline"], ref = cpu_baseline_leg(a.batch)
    err = lambda t: None if ref is None else float(np.abs(t - ref).max())
    out["max_abs_err_vs_oracle"] = err(target)
    if a.dtype in replay_target:
        out["replay_checked_vs_oracle"] = dict(max_abs_err=err(replay_target[a.dtype]),
                                               equals_eager_step_bitwise=bool(np.array_equal(replay_target[a.dtype], target)),
                                               what="output of one more replay of the TIMED hipGraph, after the timed region")
    from msmd_amd.config import PARITY_BOUNDS
    out["tolerance"] = "reference parity bound: 1e-4 max-abs on the motion coefficients (north_star); see parity_mode"
    # the stated bound of the mode that was timed (config.PARITY_BOUNDS; asserted on this very batch by tests/test_model_gpu.py::
    # test_bench_batch_b32_16_bit_modes_within_their_stated_bound): a line whose error exceeds it makes main() exit non-zero
    out["error_bound"] = PARITY_BOUNDS[a.dtype]
    if out["max_abs_err_vs_oracle"] is not None:
        out["within_error_bound"] = bool(out["max_abs_err_vs_oracle"] < PARITY_BOUNDS[a.dtype])
    if not a.no_roofline:
        out["roofline"] = roofline_leg(model, b, a.dtype)
    if not a.no_parity:
        pm = []
        for mode in [m for m in ("f16x2", "fp32", "fp16") if m != a.dtype]:   # fp16: the other 16-bit storage mode (error / time Pareto)
            el, _, tg, sm = timed(mode)
            ent = dict(dtype=mode, ms_per_step=round(el / a.steps * 1e3, 3), single_stream_ms_per_step=None if sm is None else round(sm, 3),
                       frames_per_s=round(a.batch * 100 * a.steps / el, 1),
                       max_abs_err_vs_oracle=err(tg),
                       end_to_end_tflops=round(a.batch * 100 * a.steps / el * FLOP_PER_FRAME / 1e12, 1))
            ent["error_bound"] = PARITY_BOUNDS[mode]
            if ent["max_abs_err_vs_oracle"] is not None:
                ent["meets_1e-4"] = bool(ent["max_abs_err_vs_oracle"] < 1e-4)
                ent["within_error_bound"] = bool(ent["max_abs_err_vs_oracle"] < PARITY_BOUNDS[mode])
            if mode in replay_target:
                ent["replay_max_abs_err_vs_oracle"] = err(replay_target[mode])
            if not a.no_roofline:
                ent["roofline"] = roofline_leg(model, b, mode)
            pm.append(ent)
        out["parity_mode"] = pm
        model.set_compute_dtype(a.dtype)
    # Co-headline: the fastest mode INSIDE north_star's 1e-4 (the reference computes in fp32, training_script.py:548-551) --
    # the f16x2 split-pair mode -- with its own roofline block, at the top level of the line.
    pg = (out if a.dtype == "f16x2" else next((e for e in out.get("parity_mode") or [] if e["dtype"] == "f16x2"), None))
    if pg is not None:
        out["parity_grade"] = dict(dtype="f16x2", ms_per_step=pg["ms_per_step"],
                                   frames_per_s=pg["value"] if pg is out else pg["frames_per_s"],
                                   max_abs_err_vs_oracle=pg.get("max_abs_err_vs_oracle"), error_bound=1e-4,
                                   meets_1e_4=None if pg.get("max_abs_err_vs_oracle") is None else bool(pg["max_abs_err_vs_oracle"] < 1e-4),
                                   roofline=pg.get("roofline"),
                                   note="same workload and timing as the headline; every contraction on MSMD_F16X2 split pairs "
                                        "(three f16 MFMAs per product: roofline.frac is ALGORITHMIC flops / the dense f16 peak, "
                                        "mfma_issue_frac = 3 x that is what the matrix pipe executes)")
    if a.two_streams_leg and not a.eager and a.streams == 1:
        model.set_compute_dtype(a.dtype)
        ms2 = multi_stream_forward(model, a, rank, device, 2)
        if ms2 is not None:
            out["forward_two_streams"] = ms2
        if not a.no_parity and a.dtype != "f16x2":   # the parity-grade mode the same way (its GEMM has one schedule only)
            model.set_compute_dtype("f16x2")
            ms2 = multi_stream_forward(model, a, rank, device, 2)
            if ms2 is not None:
                ms2["gemm"] = "gemm2s_kernel<128,128,4,2,2> + gemm8_kernel<f16, SPLIT>"
                out["forward_two_streams_f16x2"] = ms2
            model.set_compute_dtype(a.dtype)
    del model
    torch.cuda.empty_cache()
    if a.legs != "none":
        out["legs"] = run_legs(device, a.legs)
    return out


def run_train(a, rank, world, device):
    """configs[2]: per-GPU local batch (default 32) x 2 windows; fwd + bwd + gradient all-reduce (RCCL, ~32 MB buckets on
    a side stream) + fused Adam."""
    from msmd_amd import dp
    from msmd_amd.config import synthetic_args
    from msmd_amd.model import get_diffusion_model
    from msmd_amd.style_encoder import get_style_encoder
    from msmd_amd.training_script import Trainer, synthetic_batch
    args = synthetic_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
    model = get_diffusion_model(args, device).train()
    se = get_style_encoder(args, "vae2").to(device).train()
    # MSMD_NATIVE_RCCL=1: the buckets' all-reduce through the C ABI (msmd_allreduce_bucket, csrc/comm.hip) instead of
    # torch.distributed; --bucket-dtype bf16: 16-bit staging of every bucket (half the bytes over xGMI)
    native = os.environ.get("MSMD_NATIVE_RCCL") == "1" and world > 1
    bdt = {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[a.bucket_dtype]
    tr = Trainer(args, model, se, use_graph=not a.eager, comm=dp.RcclComm(device) if native else None, bucket_dtype=bdt if world > 1 else None)
    batch = synthetic_batch(a.batch, rank, device)
    if tr.use_graph:
        tr.capture_all(batch)
    run = lambda: tr.step(batch, it=1)
    warm = max(a.warmup, 2)
    elapsed = dp.timed_steps(run, a.steps, warm, sync=torch.cuda.synchronize, device=device)
    extra = {}
    if world > 1:
        # the exchange alone (all buckets, nothing to overlap with) and the step without any exchange
        ar = dp.timed_steps(tr.reducer.exchange_only, 5, 2, sync=torch.cuda.synchronize, device=device) / 5
        tr.reducer.mute = True
        solo = dp.timed_steps(run, a.steps, 1, sync=torch.cuda.synchronize, device=device) / a.steps
        tr.reducer.mute = False
        exposed = max(elapsed / a.steps - solo, 0.0)
        extra = dict(allreduce_ms=round(ar * 1e3, 3), step_without_exchange_ms=round(solo * 1e3, 3),
                     overlap_frac=round(max(0.0, min(1.0, 1.0 - exposed / ar)), 3) if ar > 0 else None,
                     gradient_bytes=int(tr.reducer.arena.numel() * 4), buckets=len(tr.reducer.buckets))
    roof = None
    if not a.no_roofline:
        # dominant kernel of the iteration (the 128 x 128 forward / data-gradient GEMM: ~27 % of the step's GPU time),
        # per-launch HIP events on an EAGER iteration of the same trainer (the timed one replays hipGraphs)
        was = (tr.use_graph, tr.direct_grad)
        tr.use_graph, tr.direct_grad = False, tr.reducer.world == 1
        try:
            roof = roofline_leg(None, None, "bf16", steps=1, run_step=lambda: tr.step(batch, it=1))
        finally:
            tr.use_graph, tr.direct_grad = was
    if rank != 0:
        return None
    value = a.batch * 200 * a.steps * world / elapsed
    out = {
        "metric": "FLAME frames/sec on 4s@16kHz clips (whole job; training step, 2 windows x 100 frames per sample)",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": warm,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"configs[2]: training_script step, local batch {a.batch} (global {a.batch * world}), 2 windows, "
                               "fwd + bwd + bucketed RCCL gradient all-reduce + fused Adam, train-mode noise on",
                   "batch_per_gpu": a.batch, "parallelism": f"dp{world}",
                   "launch": tr.launch_description()},
        "end_to_end_tflops": round(value / 200 * TRAIN_FLOP_PER_SAMPLE / 1e12 / world, 1),
    }
    out.update(extra)
    if roof is not None:
        roof["note"] = ("dominant kernel of the training iteration = the forward / data-gradient GEMM (msmd_gemm; the weight-gradient "
                        "product is msmd_gemm_tn); launches traced on one eager iteration. " + roof["note"])
        out["roofline"] = roof
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_train_baseline()
    if world == 1 and not a.no_exchange_rehearsal:
        del tr, model, se
        torch.cuda.empty_cache()
        out["exchange_rehearsal_world1"] = exchange_rehearsal(a, device, bdt)
    return out


def exchange_rehearsal(a, device, bucket_dtype):
    """The data-parallel step's exchange with the REAL library on one GPU: a one-rank RCCL communicator through the C ABI
    (msmd_comm_init / msmd_allreduce_bucket), the iteration captured as one hipGraph per gradient bucket, every bucket's
    all-reduce enqueued on the side stream under the rest of backward.  The sum over one rank moves no bytes between GPUs:
    what this leg establishes is that librccl loads and executes inside the step, what the segmented-graph form costs, and
    that the gradients are unchanged; xGMI time is unmeasured until a multi-GPU node runs `--gpus N`."""
    from msmd_amd import dp
    from msmd_amd.config import synthetic_args
    from msmd_amd.model import get_diffusion_model
    from msmd_amd.style_encoder import get_style_encoder
    from msmd_amd.training_script import Trainer, synthetic_batch
    args = synthetic_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
    model = get_diffusion_model(args, device).train()
    se = get_style_encoder(args, "vae2").to(device).train()
    comm = dp.RcclComm(device)
    tr = Trainer(args, model, se, use_graph=not a.eager, comm=comm, exchange_at_world_1=True, bucket_dtype=bucket_dtype)
    batch = synthetic_batch(a.batch, 0, device)
    if tr.use_graph:
        tr.capture_all(batch)
    run = lambda: tr.step(batch, it=1)
    calls = [0]
    real = comm.all_reduce

    def counted(t, stream=None):
        calls[0] += 1
        return real(t, stream)
    comm.all_reduce = counted
    with_x = dp.timed_steps(run, a.steps, 2, sync=torch.cuda.synchronize, device=device) / a.steps
    per_step = calls[0] / (a.steps + 2)
    comm.all_reduce = real
    ar = dp.timed_steps(tr.reducer.exchange_only, 5, 2, sync=torch.cuda.synchronize, device=device) / 5
    tr.reducer.mute = True
    solo = dp.timed_steps(run, a.steps, 1, sync=torch.cuda.synchronize, device=device) / a.steps
    tr.reducer.mute = False
    lib = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
    out = dict(ms_per_step_with_exchange=round(with_x * 1e3, 3), ms_per_step_exchange_muted=round(solo * 1e3, 3),
               allreduce_alone_ms=round(ar * 1e3, 3), allreduce_calls_per_step=round(per_step, 1), buckets=len(tr.reducer.buckets),
               gradient_bytes=int(tr.reducer.arena.numel() * 4), bucket_dtype=a.bucket_dtype, librccl_mapped=lib,
               launch=tr.launch_description(),
               note="one-rank communicator: RCCL executes on every bucket, no bytes cross xGMI; multi-GPU time unmeasured here")
    del tr
    comm.destroy()
    return out


def spawn_command(a, argv, port):
    """The launcher line of `--gpus N` (exactly what the driver would run itself)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn(a, argv):
    """`--gpus N` without a launcher: start N ranks with torch.distributed.run from THIS process, which has not touched
    the GPU (no exec after HIP init anywhere), and relay rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(spawn_command(a, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
        return p.returncode
    return p.returncode or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "f16x2"])
    ap.add_argument("--mode", default="forward", choices=["forward", "train"])
    ap.add_argument("--legs", default="all", help="all | none | comma-separated subset of sampler,lbs,train,hubert")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the f16x2 / fp32 parity-mode timings")
    ap.add_argument("--bucket-dtype", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="train mode: dtype the gradient buckets are summed in over RCCL (16-bit = staged copy, half the bytes)")
    ap.add_argument("--no-exchange-rehearsal", action="store_true",
                    help="train mode, one GPU: skip the leg that runs the step with a one-rank RCCL communicator on every bucket")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of hipGraph replays")
    ap.add_argument("--no-two-streams-leg", dest="two_streams_leg", action="store_false",
                    help="skip the extra leg that times two steps in flight (verified per step)")
    ap.add_argument("--tune", default="", help="developer (experimental library only): comma-separated key=value pairs for msmd_exp_set_tuning")
    ap.add_argument("--streams", type=int, default=1,
                    help="forward mode: steps in flight (one hipGraph + batch per HIP stream; every concurrent replay is checked "
                         "bit-equal to its one-at-a-time replay before timing).  2 gives +20-25 %% throughput; the headline stays one "
                         "step at a time, the two-stream figure is reported as the forward_two_streams leg.  (The round-2 mismatches "
                         "under a second stream were packed-fp32 VALU results, not stream ordering: DESIGN.md 5c; the library is built "
                         "without packed fp32 since)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn(a, sys.argv[1:]))

    # ONE JSON line on stdout, nothing else: libraries write to file descriptor 1 behind Python's back (RCCL prints a
    # five-line version banner when its first communicator comes up), so the descriptor is pointed at stderr for the
    # whole run and the line goes to a duplicate of the original one
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("MSMD_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from msmd_amd import dp
    # MSMD_DIST_BACKEND=gloo + MSMD_ONE_DEVICE=1: rehearsal of the N-rank control flow on ONE GPU (tools/bench_two_ranks_one_gpu.sh);
    # the real launch is RCCL ("nccl"), one device per rank
    dp.init(os.environ.get("MSMD_DIST_BACKEND", "nccl"), device)
    if world > 1:
        import torch.distributed as td
        world = td.get_world_size()    # what RCCL actually sees
    if os.environ.get("MSMD_TUNE"):   # developer A/B knob of the experimental library, e.g. MSMD_TUNE="7=1"
        from msmd_amd import ops as _ops
        for kv in os.environ["MSMD_TUNE"].split(","):
            _ops.exp_set_tuning(*(int(v) for v in kv.split("=")))
    if os.environ.get("MSMD_GEMM_FLAGS") or os.environ.get("MSMD_GEMM_VARIANT"):   # per-call knobs of the product library
        from msmd_amd import ops as _ops
        _ops._GEMM_DEFAULT.update(flags=int(os.environ.get("MSMD_GEMM_FLAGS", "0")) << 16,
                                  variant=int(os.environ.get("MSMD_GEMM_VARIANT", "0")))
    out = run_train(a, rank, world, device) if a.mode == "train" else run_forward(a, rank, world, device)
    if rank == 0 and out is not None:
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
    if rank == 0 and out is not None:
        # a fast line whose results are off is not a result: the JSON above is still printed (it says by how much)
        bad = [out.get("dtype")] if out.get("within_error_bound") is False else []
        bad += [e["dtype"] for e in out.get("parity_mode") or [] if e.get("within_error_bound") is False]
        if bad:
            print(f"[bench] max_abs_err_vs_oracle exceeds the stated error bound in mode(s) {bad}", file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
