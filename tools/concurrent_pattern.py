"""Dev experiment, part 4: where do concurrent extract_audio_feature results differ from serial ones?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
DT = os.environ.get("DT", "bf16")
TD = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "f16x2": torch.float32}[DT]
model = get_diffusion_model(synthetic_args(compute_dtype=DT), "cuda").eval()
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]
from msmd_amd import ops as _ops
# per-call GEMM knobs of the product library: VARIANT (0 = heuristic: 17), FLAGS (1 = write-through, 2 = paired stores)
_ops._GEMM_DEFAULT.update(variant=int(os.environ.get("VARIANT", "0")), flags=int(os.environ.get("FLAGS", "0")) << 16)
which = sys.argv[1] if len(sys.argv) > 1 else "feat"
enc = model.audio_encoder
from msmd_amd.utils.model_common import pad_audio_plan
r_, rep_ = pad_audio_plan(64000)
def fn(i):
    if which == "feat": return model.extract_audio_feature(bs[i]["audio"])
    if which == "enc": return enc.encode(bs[i]["audio"], 25, frame_num=200, dtype=TD, pad=True).float()
    if which == "conv0":      # conv0 moments (2 launches) -> stats -> conv0 + GroupNorm + GELU: a 4-kernel chain
        from msmd_amd import ops
        return ops.conv0_gn_gelu(bs[i]["audio"], FE["w0"], FE["g"], FE["b"], r_, rep_, torch.bfloat16).float()
    if which == "stats":
        from msmd_amd import ops
        B, L = bs[i]["audio"].shape
        st = torch.empty(B, 512, 2, device="cuda"); ws = torch.empty(B, ops.CONV0_SPLITS, 66, device="cuda")
        from msmd_amd import _lib
        _lib.check(_lib.load().msmd_conv0_stats(bs[i]["audio"].data_ptr(), FE["w0"].data_ptr(), st.data_ptr(), ws.data_ptr(), B, L, r_, rep_, 512, 1e-5,
                                                torch.cuda.current_stream().cuda_stream), "stats")
        return st
    if which == "fe_fp":
        x = enc.feature_extractor_cl(bs[i]["audio"], torch.bfloat16, r_, rep_)
        from msmd_amd import ops
        P = enc.pack(torch.bfloat16)
        return ops.gemm(ops.layernorm(x, *P.fp_ln, eps=enc.config.layer_norm_eps), P.fp_w, P.fp_b).float()
FE = None
if which in ("conv0", "stats"):
    _P = enc.pack_fe(torch.bfloat16)
    FE = dict(w0=_P.w0, g=_P.gn_g, b=_P.gn_b)
refs = []
for i in range(2):
    o = fn(i); torch.cuda.synchronize(); refs.append(o.clone())
s = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)] if os.environ.get("PRIO") else [torch.cuda.Stream(), torch.cuda.Stream()]
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
nbad = 0
for rep in range(int(os.environ.get("REPS", "20"))):
    for st in s: st.wait_stream(torch.cuda.current_stream())
    for k in range(2):
        outs = []
        for i in range(2):
            with torch.cuda.stream(s[i]): outs.append(fn(i))
    torch.cuda.synchronize()
    for i in range(2):
        if not torch.equal(outs[i], refs[i]):
            nbad += 1
            d = (outs[i] - refs[i]).abs()
            idx = torch.nonzero(d > 0)
            if nbad <= 6:
                print(f"rep {rep} stream {i}: {idx.shape[0]} of {d.numel()} differ, max {float(d.max()):.3g}; clips {sorted(set(idx[:, 0].tolist()))[:12]}; frames {int(idx[:, 1].min())}..{int(idx[:, 1].max())}")
print(f"RESULT {which} dtype={DT} variant={os.environ.get('VARIANT', '0')} flags={os.environ.get('FLAGS', '0')}: mismatching results", nbad, "of", 2 * int(os.environ.get("REPS", "20")))
