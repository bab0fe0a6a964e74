"""conv0 + GroupNorm + GELU (VALU / transcendental heavy, no MFMA) on one HIP stream while ANOTHER stream runs something
else: does its output change?  Found by tools/concurrent_trace.py: under two-stream concurrency the FIRST op of the encoder
pass that differs from its serial result is this kernel -- ~50 scattered (row, 16-element) spots per launch, even
channels only, lanes 48-63 of a wave, O(1) wrong.

  python tools/conv0_corun.py [launches]     env: MSMD_LIB=<other build of the library> (e.g. built with -fno-slp-vectorize)
Partner loads on the second stream: none | gemm (bf16 MFMA GEMM loop) | copy (HBM streaming, no MFMA) | conv0 (itself).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.utils.model_common import pad_audio_plan

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
P = model.audio_encoder.pack_fe(torch.bfloat16)
audio = [bench.synth_batch(32, r, "cuda")["audio"] for r in range(2)]
r_, rep_ = pad_audio_plan(64000)
odt = {"bf16": torch.bfloat16, "fp32": torch.float32}[os.environ.get("ODT", "bf16")]


def conv0(i):
    return ops.conv0_gn_gelu(audio[i], P.w0, P.gn_g, P.gn_b, r_, rep_, odt)


ref = conv0(0).clone()
torch.cuda.synchronize()
A = torch.randn(8192, 3072, device="cuda").to(torch.bfloat16)
W = torch.randn(4096, 3072, device="cuda").to(torch.bfloat16)
big = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
big2 = torch.empty_like(big)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


enc = model.audio_encoder
x0 = conv0(1).clone()                                   # (32, 12815, 512) bf16: the conv stack's input
h768 = torch.randn(32, 200, 768, device="cuda").to(torch.bfloat16)
qkv = torch.randn(32, 200, 2304, device="cuda").to(torch.bfloat16)
PK = enc.pack(torch.bfloat16)


def partner(kind, n):
    for _ in range(n):
        if kind == "gemm":
            ops.gemm(A, W)
        elif kind == "copy":
            big2.copy_(big)
        elif kind == "conv0":
            conv0(1)
        elif kind == "convgemm":     # conv1 of the stack: windowed-A GEMM, M = 32 x 6407, K = 1536
            ops.conv1d_cl(x0, P.conv_w[0], P.conv_b[1], kernel=3, stride=2, act=ops.ACT_GELU)
        elif kind == "attn":
            for _ in range(8):
                ops.attention(qkv[..., :768], qkv[..., 768:1536], qkv[..., 1536:], 12, 0.125)
        elif kind == "ln":
            for _ in range(16):
                ops.layernorm(h768, *PK.enc_ln, residual=h768)
        elif kind == "fe":
            enc.feature_extractor_cl(audio[1], torch.bfloat16, r_, rep_)
        elif kind == "enc":
            enc.encode(audio[1], 25, frame_num=200, dtype=torch.bfloat16, pad=True)


print(f"library: {os.environ.get('MSMD_LIB', 'product')}; output dtype {odt}")
for kind in os.environ.get("PARTNERS", "none,gemm,copy,conv0,convgemm,attn,ln,fe,enc").split(","):
    nbad = torch.zeros((), device="cuda", dtype=torch.int64)
    nlaunch_bad = torch.zeros((), device="cuda", dtype=torch.int64)
    first = None
    for st in (sa, sb):
        st.wait_stream(torch.cuda.current_stream())
    for it in range(N):
        with torch.cuda.stream(sb):
            partner(kind, 2 if kind in ("gemm", "copy") else 1)
        with torch.cuda.stream(sa):
            o = conv0(0)
            ne = o != ref
            c = ne.sum()
            nbad += c
            nlaunch_bad += (c > 0).to(torch.int64)
            if first is None and it % 16 == 0 and int(c.item()):
                first = (o.clone(), it)
    torch.cuda.synchronize()
    print(f"RESULT partner={kind:6s}: {int(nlaunch_bad)} of {N} launches differ from the serial result, {int(nbad)} elements in all", flush=True)
    if first is not None:
        o, it = first
        d = (o.float() - ref.float())
        idx = torch.nonzero(d != 0)
        rows = idx[:, 0] * o.shape[1] + idx[:, 1]
        ur = torch.unique(rows)
        print(f"   launch {it}: {idx.shape[0]} elements in {ur.numel()} rows; columns {sorted(set(idx[:, 2].tolist()))[:40]}")
        b, t, c = idx[0].tolist()
        cs = sorted(set(idx[(idx[:, 0] == b) & (idx[:, 1] == t)][:, 2].tolist()))
        print(f"   first bad row (clip {b}, frame {t}): columns {cs}")
        print(f"      got {[round(float(o[b, t, x]), 4) for x in cs[:8]]}")
        print(f"      ref {[round(float(ref[b, t, x]), 4) for x in cs[:8]]}")
        for dt in (-4, -3, -2, -1, 1, 2, 3, 4):      # is it a neighbouring frame's value?
            if 0 <= t + dt < o.shape[1] and all(float(o[b, t, x]) == float(ref[b, t + dt, x]) for x in cs[:8]):
                print(f"      = the reference values of frame {t + dt}")
