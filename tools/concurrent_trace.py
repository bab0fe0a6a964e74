"""Two encoder passes on two HIP streams with EVERY intermediate tensor kept alive and compared, op by op, with the same
pass run alone: where does a concurrent pass first differ from the serial one, and how (which rows / columns, by how much)?

KEEP=0 runs the same loop without the recorder (intermediates are freed and their blocks recycled inside a pass, as in
production) as the control for "does keeping the buffers alive hide the effect".
  python tools/concurrent_trace.py [reps]     env: KEEP (1), VARIANT / FLAGS (per-call GEMM knobs), STAGE (enc | feat)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
KEEP = os.environ.get("KEEP", "1") == "1"
STAGE = os.environ.get("STAGE", "enc")
ops._GEMM_DEFAULT.update(variant=int(os.environ.get("VARIANT", "0")), flags=int(os.environ.get("FLAGS", "0")) << 16)
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
enc = model.audio_encoder
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]

REC = {"cur": None}


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if REC["cur"] is not None:
            REC["cur"].append((name, out if torch.is_tensor(out) else getattr(out, "t", out)))
        return out
    setattr(ops, name, f)


for n in ("gemm", "layernorm", "attention", "group_pad", "conv0_gn_gelu", "interp_linear"):
    wrap(n)


def fn(i):
    if STAGE == "feat":
        return model.extract_audio_feature(bs[i]["audio"])
    return enc.encode(bs[i]["audio"], 25, frame_num=200, dtype=torch.bfloat16, pad=True).float()


ref_out, ref_rec = [], []
for i in range(2):
    fn(i)
    torch.cuda.synchronize()
    REC["cur"] = []
    o = fn(i)
    torch.cuda.synchronize()
    ref_out.append(o.clone())
    ref_rec.append([(n, t.clone()) for n, t in REC["cur"]])
    REC["cur"] = None
print(f"{len(ref_rec[0])} recorded ops per pass; KEEP={int(KEEP)} stage={STAGE}")
s = [torch.cuda.Stream(), torch.cuda.Stream()]
nbad = shown = 0
for rep in range(reps):
    for st in s:
        st.wait_stream(torch.cuda.current_stream())
    recs = [[], []]
    for k in range(2):
        outs = []
        for i in range(2):
            with torch.cuda.stream(s[i]):
                if KEEP and k == 1:
                    REC["cur"] = recs[i]
                outs.append(fn(i))
                REC["cur"] = None
    torch.cuda.synchronize()
    for i in range(2):
        if torch.equal(outs[i], ref_out[i]):
            continue
        nbad += 1
        if not KEEP or shown >= 6:
            continue
        shown += 1
        for j, ((n, t), (_, r)) in enumerate(zip(recs[i], ref_rec[i])):
            if torch.equal(t, r):
                continue
            d = (t.float() - r.float()).abs()
            d2 = d.reshape(-1, d.shape[-1]) if d.dim() > 1 else d.reshape(1, -1)
            rows = torch.nonzero(d2.amax(dim=1) > 0).flatten()
            cols = torch.nonzero(d2.amax(dim=0) > 0).flatten()
            rl = rows.tolist()
            runs, a = [], None
            for x in rl:            # contiguous row runs
                if a is None:
                    a = b_ = x
                elif x == b_ + 1:
                    b_ = x
                else:
                    runs.append((a, b_)); a = b_ = x
            if a is not None:
                runs.append((a, b_))
            print(f"rep {rep} stream {i}: FIRST differing op #{j} {n} shape {tuple(t.shape)} {t.dtype}: {int((d > 0).sum())} elements, "
                  f"max |d| {float(d.max()):.4g} (max |ref| {float(r.float().abs().max()):.3g}); rows {len(rl)} in {len(runs)} runs "
                  f"{runs[:6]}{'...' if len(runs) > 6 else ''}; cols {int(cols.min())}..{int(cols.max())} ({cols.numel()} distinct)", flush=True)
            if t.dim() == 3:        # detail of the first bad row: which columns, what values, a neighbouring frame's?
                idx = torch.nonzero(d > 0)
                b0, t0 = int(idx[0, 0]), int(idx[0, 1])
                cs = sorted(set(idx[(idx[:, 0] == b0) & (idx[:, 1] == t0)][:, 2].tolist()))
                print(f"    first bad row (clip {b0}, frame {t0}): columns {cs}")
                print(f"      got {[round(float(t[b0, t0, x]), 4) for x in cs[:16]]}")
                print(f"      ref {[round(float(r[b0, t0, x]), 4) for x in cs[:16]]}")
                for dt in range(-8, 9):
                    if dt and 0 <= t0 + dt < t.shape[1] and all(float(t[b0, t0, x]) == float(r[b0, t0 + dt, x]) for x in cs):
                        print(f"      = the reference values of frame {t0 + dt}")
                per_row = torch.bincount((idx[:, 0] * t.shape[1] + idx[:, 1]) - int((idx[:, 0] * t.shape[1] + idx[:, 1]).min()))
                per_row = per_row[per_row > 0]
                print(f"      bad elements per bad row: min {int(per_row.min())} max {int(per_row.max())}; frames mod 64 of bad rows: {sorted(set((idx[:, 1] % 64).tolist()))[:20]}; mod 4: {sorted(set((idx[:, 1] % 4).tolist()))}")
            # how many later ops differ
            later = sum(not torch.equal(t2, r2) for (_, t2), (_, r2) in zip(recs[i][j + 1:], ref_rec[i][j + 1:]))
            print(f"    previous op #{j - 1} {recs[i][j - 1][0] if j else '-'} equal; {later} of {len(recs[i]) - j - 1} later ops differ")
            break
    del recs, outs
print(f"RESULT trace stage={STAGE} KEEP={int(KEEP)} variant={ops._GEMM_DEFAULT['variant']}: mismatching results {nbad} of {2 * reps}")
