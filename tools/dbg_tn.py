import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops, _lib
lib = _lib.load()
def t(fn, n=50):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(6400, 768, 768), (6400, 3072, 768), (6400, 768, 3072), (6400, 2304, 768), (3552, 512, 512), (3552, 2048, 512), (12800, 512, 1536)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
for M, N, K in shapes:
    a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, K, device="cuda").bfloat16()
    c = torch.empty(N, K, device="cuda"); ws = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    fl = 2.0 * M * N * K
    for sp in (1, 2, 4, 0):
        ops.set_tuning(2, sp)
        for dbg in (0, 2, 3):
            ops.set_tuning(3, dbg)
            us = t(lambda: lib.msmd_gemm_tn(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, M, N, K, N, K, K, 1, 0, 0, 0, 0, 0, 0,
                                            ws.data_ptr(), ws.numel(), st))
            print(f"M={M} N={N} K={K} splits={sp} dbg={dbg}: {us:.1f} us  {fl / us / 1e6:.0f} TF", flush=True)
    ops.set_tuning(3, 0); ops.set_tuning(2, 0)
