"""Where does one GEMM launch spend its time, per workgroup?  (developer build: make -C ubisoft-laforge-msmd_amd/csrc EXP=1)
   MSMD_LIB=ubisoft-laforge-msmd_amd/csrc/libmsmd_hip_exp.so python tools/gemm_stamps.py [M N K]
Every workgroup of gemm2_kernel<128,128,4,2,2,pipelined> (variant 17, plain epilogue) stamps s_memrealtime (100 MHz) at
entry, after the first K tile's barrier (= prologue done: arguments, addresses, first operand tile landed), after the K
loop, after its epilogue's last store was ISSUED, and after vmcnt(0) (stores drained), plus the CU it ran on."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from msmd_amd import _lib, ops

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (6400, 2304, 768)
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda", generator=g)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
act = ops.ACT_GELU if os.environ.get("ACT", "none") == "gelu" else ops.ACT_NONE
FLAGS = {"paired": ops.GEMM_PAIRED_STORES, "plain": 0, "wt": ops.GEMM_WRITE_THROUGH | ops.GEMM_PAIRED_STORES}[os.environ.get("FLAGS", "paired")]
BIAS = os.environ.get("BIAS", "1") == "1"
VARIANT = int(os.environ.get("VARIANT", "17"))      # 12 / 9: the 64 x 64 tiles of the decoder-sized GEMMs
for _ in range(5):
    ops.gemm(a, w, b if BIAS else None, None, act, out=out, variant=VARIANT, flags=FLAGS)
stamps = torch.zeros(4096 * 8, device="cuda", dtype=torch.int64)
lib.msmd_exp_set_stamps.argtypes = [__import__("ctypes").c_void_p]
lib.msmd_exp_set_stamps(stamps.data_ptr())
runs = []
for _ in range(5):
    stamps.zero_()
    torch.cuda.synchronize()
    ops.gemm(a, w, b if BIAS else None, None, act, out=out, variant=VARIANT, flags=FLAGS)
    torch.cuda.synchronize()
    s = stamps.view(-1, 8).cpu().numpy()
    runs.append(s[s[:, 0] > 0].copy())
lib.msmd_exp_set_stamps(None)
s = runs[-1]
t0 = s[:, 0].min()
us = lambda x: x / 100.0            # ticks of 10 ns
start, pro, loop, epi, drain = us(s[:, 0] - t0), us(s[:, 1] - s[:, 0]), us(s[:, 2] - s[:, 1]), us(s[:, 3] - s[:, 2]), us(s[:, 4] - s[:, 3])
end = us(s[:, 4] - t0)
hw = s[:, 5]
cu = (s[:, 6] << 16) | (hw & 0xFF00)        # XCC | SE / SH / CU bits of HW_ID
print(f"[act {os.environ.get('ACT', 'none')}, stores {os.environ.get('FLAGS', 'paired')}, bias {BIAS}] {M} x {N} x {K}, {len(s)} workgroups on {len(set(cu.tolist()))} distinct CUs; launch spans {end.max():.2f} us from the first workgroup's entry")
order = np.argsort(start)
first = order[:min(512, len(s))]
late = order[min(512, len(s)):]
for name, idx in (("first 512 to start", first), ("started later (second round)", late)):
    if len(idx) == 0:
        continue
    q = lambda v: f"{np.median(v[idx]):6.2f} (p10 {np.percentile(v[idx], 10):5.2f}, p90 {np.percentile(v[idx], 90):5.2f})"
    print(f"  {name} [{len(idx)}]: start {q(start)}  prologue {q(pro)}  K loop {q(loop)}  epilogue issue {q(epi)}  store drain {q(drain)}  end {q(end)}")
# one CU's timeline
by_cu = {}
for i in order:
    by_cu.setdefault(int(cu[i]), []).append(i)
busiest = sorted(by_cu.values(), key=len)[-1]
print("  one CU's workgroups (start, prologue, K loop, epilogue issue, drain, end):")
for i in busiest:
    print(f"    +{start[i]:6.2f}  {pro[i]:5.2f}  {loop[i]:6.2f}  {epi[i]:5.2f}  {drain[i]:5.2f}  -> {end[i]:6.2f}   tile {int(s[i, 7]) // 1000},{int(s[i, 7]) % 1000}")
hist = np.bincount([len(v) for v in by_cu.values()])
print("  workgroups per CU histogram:", {k: int(v) for k, v in enumerate(hist) if v})
print(f"  sum over workgroups / (CUs x span): prologue {pro.sum() / len(by_cu) / end.max():.2f}  K loop {loop.sum() / len(by_cu) / end.max():.2f}  epilogue+drain {(epi + drain).sum() / len(by_cu) / end.max():.2f}   (2 co-resident workgroups: up to 2.0 in total)")
