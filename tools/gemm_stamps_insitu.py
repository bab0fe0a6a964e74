"""Per-workgroup phases of the encoder GEMMs INSIDE the forward step (developer build):
   MSMD_LIB=ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_exp.so python tools/gemm_stamps_insitu.py
The bench step runs eagerly; the stamps of gemm2_kernel<128,128> are switched on around ONE chosen launch (the 6th
encoder layer's QKV / out-projection / FFN-1 / FFN-2: msmd_gemm_ln calls by shape and occurrence)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from msmd_amd import _lib, ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

lib = _lib.load()
lib.msmd_exp_set_stamps.argtypes = [ctypes.c_void_p]
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
b = bench.synth_batch(32, 0, "cuda")
for _ in range(3):
    bench.step(model, b)
stamps = torch.zeros(4096 * 8, device="cuda", dtype=torch.int64)
real = ops.gemm_ln
targets = {"qkv": (2304, 768), "out": (768, 768), "ffn1": (3072, 768), "ffn2": (768, 3072)}
for name, (N, K) in targets.items():
    seen = [0]

    def wrapped(a, w, *args, **kw):
        hit = w.shape[0] == N and a.shape[-1] == K
        if hit:
            seen[0] += 1
        on = hit and seen[0] == 6
        if on:
            stamps.zero_()
            lib.msmd_exp_set_stamps(stamps.data_ptr())
        r = real(a, w, *args, **kw)
        if on:
            lib.msmd_exp_set_stamps(None)
        return r
    ops.gemm_ln = wrapped
    import msmd_amd.utils.wav2vec2 as w2
    bench.step(model, b)
    torch.cuda.synchronize()
    ops.gemm_ln = real
    s = stamps.view(-1, 8).cpu().numpy()
    s = s[s[:, 0] > 0]
    if len(s) == 0:
        print(name, "no stamps (launch not found)")
        continue
    t0 = s[:, 0].min()
    us = lambda x: x / 100.0
    start, pro, loop, epi, drain, end = us(s[:, 0] - t0), us(s[:, 1] - s[:, 0]), us(s[:, 2] - s[:, 1]), us(s[:, 3] - s[:, 2]), us(s[:, 4] - s[:, 3]), us(s[:, 4] - t0)
    order = np.argsort(start)
    for tag, idx in (("first 512", order[:512]), ("later", order[512:])):
        if len(idx) == 0:
            continue
        q = lambda v: f"{np.median(v[idx]):6.2f} ({np.percentile(v[idx], 10):5.2f}..{np.percentile(v[idx], 90):5.2f})"
        print(f"{name:5s} in situ, {tag:9s} [{len(idx):4d}]: start {q(start)}  prologue {q(pro)}  K loop {q(loop)}  epilogue {q(epi)}  drain {q(drain)}  end {q(end)}   span {end.max():.1f} us")
