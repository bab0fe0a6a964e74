"""Dev experiment, part 2: the encoder split at its stages, two streams, eager."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmd_amd import ops
from msmd_amd.config import synthetic_args
from msmd_amd.model import get_diffusion_model
from msmd_amd.utils.model_common import pad_audio_plan
model = get_diffusion_model(synthetic_args(compute_dtype="bf16"), "cuda").eval()
enc = model.audio_encoder
bf = torch.bfloat16
bs = [bench.synth_batch(32, r, "cuda") for r in range(2)]
r_, rep_ = pad_audio_plan(64000)
xs = [enc.feature_extractor_cl(b["audio"], bf, r_, rep_) for b in bs]
T50 = xs[0].shape[1]
crop = min(round(200 * 50 / 50), T50)
print("T50", T50, "crop", crop)
xi = [ops.interp_linear(x, 200, crop) if not (crop == T50) else x for x in xs]
P = enc.pack(bf); c = enc.config
def st_interp(i): return ops.interp_linear(xs[i], 200, crop).float() if crop != T50 else xs[i].float()
def st_fp(i):
    h = ops.layernorm(xi[i], *P.fp_ln, eps=c.layer_norm_eps); return ops.gemm(h, P.fp_w, P.fp_b).float()
hs = [ops.gemm(ops.layernorm(xi[i], *P.fp_ln, eps=c.layer_norm_eps), P.fp_w, P.fp_b) for i in range(2)]
def st_pos(i):
    h = hs[i]; B, T, d = h.shape; G, kpos = 16, 128; cg = d // G
    xp = ops.group_pad(h, G, kpos // 2, cg_out=P.pos_cg, split=False); Tp = T + kpos
    y = torch.empty_like(h)
    ops.gemm(xp, P.pos_w, P.pos_b, h, ops.ACT_GELU, out=y, M=B * T, N=cg, K=kpos * P.pos_cg, lda=P.pos_cg, rows_per_batch=T,
             a_batch_stride=G * Tp * P.pos_cg, ldw=kpos * P.pos_cg, ldc=d, batch=G, strideA=Tp * P.pos_cg, strideW=cg * kpos * P.pos_cg,
             strideC=cg, strideBias=cg, strideR=cg)
    return y.float()
def st_features(i): return enc.encode_features(xi[i], bf).float()
f768 = [enc.encode_features(xi[i], bf) for i in range(2)]
w, b_ = model._afm_packed(bf)
def st_afm(i): return ops.gemm(f768[i], w, b_).float()
def st_layer0(i):
    L = P.layers[0]; h = hs[i]; B, T, d = h.shape
    qkv = ops.gemm(h, L.wqkv, L.bqkv)
    a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], 12, 0.125)
    o = ops.gemm(a, L.wo, L.bo)
    h1 = ops.layernorm(o, *L.ln1, residual=h)
    f = ops.gemm(ops.gemm(h1, L.w1, L.b1, None, ops.ACT_GELU), L.w2, L.b2)
    return ops.layernorm(f, *L.ln2, residual=h1).float()
stages = {"interp": st_interp, "feature projection (LN + GEMM K=512)": st_fp, "positional conv (group_pad + 16 batched GEMMs)": st_pos,
          "one encoder layer": st_layer0, "encode_features (pos conv + 12 layers)": st_features, "audio feature map GEMM": st_afm}
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for name, fn in stages.items():
    refs = []
    for i in range(2):
        o = fn(i); torch.cuda.synchronize(); refs.append(o.clone())
        o = fn(i); torch.cuda.synchronize(); assert torch.equal(o, refs[-1]), name
    bad = 0; worst = 0.0
    for rep in range(15):
        for st in s: st.wait_stream(torch.cuda.current_stream())
        for k in range(2):
            outs = []
            for i in range(2):
                with torch.cuda.stream(s[i]): outs.append(fn(i))
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(outs[i], refs[i]):
                bad += 1; worst = max(worst, float((outs[i].float() - refs[i].float()).abs().max()))
    print(f"{name:50s}: {bad} of 30 concurrent results differ (worst {worst:.3g})", flush=True)
