"""Which host-library (ATen) ops does ONE eager training iteration issue, by op and shape?  (The HIP kernels are called
through ctypes and do not show up here: this lists exactly the glue that is NOT ours.)  python tools/aten_ops_in_train_step.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.style_encoder import get_style_encoder
from msmd_amd.training_script import Trainer, synthetic_batch

B = int(os.environ.get("B", "32"))
args = synthetic_args(compute_dtype="bf16", lr=2e-5, warm_iter=5000)
model = get_diffusion_model(args, "cuda").train()
se = get_style_encoder(args, "vae2").to("cuda").train()
tr = Trainer(args, model, se, use_graph=False)
batch = synthetic_batch(B, 0, "cuda")
tr.step(batch, it=1)
torch.cuda.synchronize()
counts = collections.Counter()
bytes_ = collections.Counter()
by_line = collections.Counter()
REDUCE = ("sum", "mean", "var", "var_mean", "std", "norm", "linalg_vector_norm", "amax", "amin", "max", "min", "all", "any", "prod",
          "logsumexp", "_softmax", "count_nonzero", "cumsum", "argmax")
reductions = collections.Counter()   # (op, input shape, outputs, elements reduced per output, calling line)
engine = collections.Counter()
slow = collections.Counter()      # ops with a non-contiguous (s) or broadcast operand: the generic strided elementwise kernels


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if name.split(".")[0] in ("view", "_unsafe_view", "reshape", "detach", "alias", "t", "transpose", "permute", "expand", "slice",
                                  "select", "unsqueeze", "squeeze", "as_strided", "split", "unbind", "chunk", "_reshape_alias", "empty",
                                  "empty_like", "empty_strided", "new_empty", "is_same_size", "sym_size", "stride", "storage_offset", "numel"):
            return out
        shp = next((tuple(a.shape) for a in args if torch.is_tensor(a)), ())
        t = out if torch.is_tensor(out) else None
        if name.split(".")[0] in REDUCE and t is not None and shp:
            n_in = 1
            for d_ in shp:
                n_in *= d_
            n_out = max(1, t.numel())
            wr = "autograd engine"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "msmd_amd" in fr.filename:
                    wr = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            reductions[(name, shp, n_out, n_in // n_out, wr)] += 1
        counts[(name, shp)] += 1
        where = "autograd engine (backward of a torch op)"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "msmd_amd" in fr.filename:
                where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                break
        by_line[(where, name)] += 1
        ts = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        if where.startswith("autograd engine"):
            engine[(where, name, " ".join(str(tuple(a.shape)) + ("" if a.is_contiguous() else "s") for a in ts)[:60])] += 1
        if any(not a.is_contiguous() for a in ts) or len({tuple(a.shape) for a in ts if a.dim() > 0}) > 1:
            slow[(where, name, " ".join(str(tuple(a.shape)) + ("" if a.is_contiguous() else "s") for a in ts))] += 1
        if t is not None:
            bytes_[(name, shp)] += t.numel() * t.element_size()
        return out


with Count():
    tr.step(batch, it=1)
torch.cuda.synchronize()
tot = sum(counts.values())
print(f"{tot} data-moving ATen ops in one eager iteration (B = {B})")
by_name = collections.Counter()
for (n, s), c in counts.items():
    by_name[n] += c
print("by op:", by_name.most_common(25))
print("by calling line of this package (top 60):")
for (w, n), c in by_line.most_common(60):
    print(f"  {w:44s} {n:28s} x{c}")
print("ops issued by the autograd engine (backward of torch ops in the glue code), by op and first-operand shape:")
eng = collections.Counter()
for (w, n, shp), c in engine.items():
    eng[(n, shp)] += c
for (n, shp), c in eng.most_common(60):
    print(f"  {n:28s} {shp:60s} x{c}")
print(f"ops with a strided (s) or broadcast operand: {sum(slow.values())}")
for (w, n, shp), c in slow.most_common(70):
    print(f"  {w:44s} {n:24s} x{c:3d}  {shp[:90]}")
# The host library runs a reduction over MANY elements into FEW outputs as a multi-block kernel (partial sums in a staging
# buffer + a semaphore zeroed by a memset node): inside the step's hipGraphs two of those returned wrong sums.
# Everything with >= 256 elements per output and fewer than ~2048 outputs is a candidate and must not sit inside a capture.
print("reductions by the host library (op, input shape, outputs, elements per output, calling line):")
for (n, shp, n_out, per, wr), c in sorted(reductions.items(), key=lambda kv: -kv[0][3]):
    print(f"  {'CANDIDATE ' if per >= 256 and n_out < 2048 else '          '}{n:22s} {str(shp):26s} -> {n_out:8d} outputs, {per:8d} per output  x{c:3d}  {wr}")
print("largest by bytes written:")
for (n, s), b in bytes_.most_common(40):
    print(f"  {n:28s} {str(s):28s} x{counts[(n, s)]:4d}  {b / 1e6:9.1f} MB")
