"""Does the encoder's FFN-2 GEMM (6400 x 768 x 3072) slow down when its operands are not already in this XCD's L2?  In the
forward step it takes 53 us against 41 us back-to-back on the same buffers.  Rotates over n weight copies (4.7 MB each) and
n activation copies (39 MB each): n = 1 hot; 8 = beyond L2 (32 MB); 64 weights = beyond the 256 MB Infinity Cache."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from msmd_amd import ops

M, N, K = 6400, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
bias = torch.randn(N, device="cuda", generator=g)
res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for nw, na in ((1, 1), (32, 1), (64, 1), (1, 4), (1, 8)):
    ws = [(torch.randn(N, K, device="cuda", generator=g) / 55).to(torch.bfloat16) for _ in range(nw)]
    As = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(na)]
    for i in range(max(nw, na)):
        ops.gemm(As[i % na], ws[i % nw], bias, res, out=out)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()       # a graph of 64 launches: no host launch gaps in the measurement
    with torch.cuda.graph(gr):
        for i in range(64):
            ops.gemm(As[i % na], ws[i % nw], bias, res, out=out)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{nw:3d} weight copies ({nw * 4.7:6.1f} MB), {na} activation copies ({na * 39.3:6.1f} MB): {e0.elapsed_time(e1) / 320 * 1e3:6.1f} us per launch", flush=True)
    del ws, As

# ---- would a prefetch of the NEXT launch's weights (into the memory-side cache), issued on a second stream while the current
# launch runs, give the hot time back?  64 weight copies (cold), 1 activation copy; the prefetch is a plain read (sum) of w[i + 1]
nw = 64
ws = [(torch.randn(N, K, device="cuda", generator=g) / 55).to(torch.bfloat16) for _ in range(nw)]
A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
sink = torch.zeros(64, device="cuda")
side = torch.cuda.Stream()
for mode in ("no prefetch", "prefetch next weights on a side stream"):
    def body():
        main = torch.cuda.current_stream()
        for i in range(64):
            if mode != "no prefetch":
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    sink[i % 64] = ws[(i + 1) % nw].view(torch.float32).sum()     # a full, coalesced read of the next weights
            ops.gemm(A, ws[i % nw], bias, res, out=out)
            if mode != "no prefetch":
                main.wait_stream(side)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"64 cold weight copies, {mode}: {e0.elapsed_time(e1) / 320 * 1e3:6.1f} us per launch", flush=True)

# ---- conv1 of the feature extractor on HALF the batch (M = 16 x 6407 windows, K = 3 x 512, N = 512): its input, conv0's output,
# is 210 MB -- inside the Infinity Cache if it was JUST written (and if writes allocate there).  Producer stand-in: a copy_ into A.
from msmd_amd.utils.wav2vec2 import CONV_KERNEL, CONV_STRIDE
Bh, T0, C = 16, 12815, 512
x = torch.randn(Bh, T0, C, device="cuda", generator=g).to(torch.bfloat16)
x_src = x.clone()
w = (torch.randn(C, 3 * C, device="cuda", generator=g) / 40).to(torch.bfloat16)
bias1 = torch.randn(C, device="cuda", generator=g)
junk = torch.empty(640 << 20, device="cuda", dtype=torch.uint8)
for mode in ("input just written (producer -> consumer)", "input evicted (640 MB touched in between)"):
    ts = []
    for _ in range(6):
        x.copy_(x_src)                      # the producer: writes all of x
        if mode.startswith("input evicted"):
            junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv1d_cl(x, w, bias1, kernel=3, stride=2, act=ops.ACT_GELU)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"conv1 on 16 clips, {mode}: {sorted(ts)[len(ts) // 2]:6.1f} us", flush=True)
