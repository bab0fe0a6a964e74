"""Dev experiment: which op gives different bits when two streams run it concurrently (vs one launch at a time)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops
torch.manual_seed(0)
dev = "cuda"
bf = torch.bfloat16
def mk(*shape, scale=1.0, dtype=bf): return (torch.randn(*shape, device=dev) * scale).to(dtype)
cases = {}
# encoder GEMMs (M = 6400)
for name, (M, N, K) in dict(qkv=(6400, 2304, 768), out=(6400, 768, 768), ffn1=(6400, 3072, 768), ffn2=(6400, 768, 3072), dec=(3520, 512, 512), dec_ff=(3520, 2048, 512)).items():
    a = [mk(M, K) for _ in range(2)]; w = mk(N, K, scale=K ** -0.5); b = torch.randn(N, device=dev)
    cases["gemm_" + name] = [(lambda a=a[i], w=w, b=b: ops.gemm(a, w, b, None, ops.ACT_GELU)) for i in range(2)]
# conv1 as windowed GEMM
x = [mk(32, 12799, 512) for _ in range(2)]; wc = mk(512, 1536, scale=1536 ** -0.5); bc = torch.randn(512, device=dev)
cases["conv1"] = [(lambda x=x[i]: ops.conv1d_cl(x, wc, bc, kernel=3, stride=2, act=ops.ACT_GELU)) for i in range(2)]
# attention T = 200, 12 heads
qkv = [mk(32, 200, 2304) for _ in range(2)]
cases["attn"] = [(lambda t=qkv[i]: ops.attention(t[..., :768], t[..., 768:1536], t[..., 1536:], 12, 0.125)) for i in range(2)]
# layernorm with residual
h = [mk(6400, 768) for _ in range(2)]; r = mk(6400, 768); g = torch.randn(768, device=dev); be = torch.randn(768, device=dev)
cases["layernorm"] = [(lambda t=h[i]: ops.layernorm(t, g, be, residual=r)) for i in range(2)]
# conv0
au = [torch.randn(32, 64000, device=dev) for _ in range(2)]; w0 = torch.randn(512, 10, device=dev) * 0.3; gg = torch.randn(512, device=dev); gb = torch.randn(512, device=dev)
cases["conv0"] = [(lambda t=au[i]: ops.conv0_gn_gelu(t, w0, gg, gb, 0, 0, bf)) for i in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def as_t(o): return o if torch.is_tensor(o) else o[0]
for name, fns in cases.items():
    refs = []
    for f in fns:
        o = as_t(f()); torch.cuda.synchronize(); refs.append(o.clone())
        o2 = as_t(f()); torch.cuda.synchronize()
        assert torch.equal(o2, refs[-1]), name + " not deterministic serially"
    bad = 0; worst = 0.0
    for rep in range(R):
        for st in s: st.wait_stream(torch.cuda.current_stream())
        outs = []
        for k in range(3):           # several back-to-back launches per stream so they really overlap
            outs = []
            for i in range(2):
                with torch.cuda.stream(s[i]): outs.append(as_t(fns[i]()))
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(outs[i], refs[i]):
                bad += 1; worst = max(worst, float((outs[i].float() - refs[i].float()).abs().max()))
    print(f"{name:12s}: {bad} mismatching outputs of {2 * R} concurrent launches (worst |diff| {worst:.3g})", flush=True)
# mixed pairs: op A on stream 0 beside op B on stream 1
import itertools
names = list(cases)
for na, nb in itertools.combinations(names, 2):
    fa, fb = cases[na][0], cases[nb][1]
    ra = as_t(fa()).clone(); rb = as_t(fb()).clone(); torch.cuda.synchronize()
    bad = 0
    for rep in range(10):
        for st in s: st.wait_stream(torch.cuda.current_stream())
        for k in range(3):
            with torch.cuda.stream(s[0]): oa = as_t(fa())
            with torch.cuda.stream(s[1]): ob = as_t(fb())
        torch.cuda.synchronize()
        bad += (not torch.equal(oa, ra)) + (not torch.equal(ob, rb))
    if bad: print(f"pair {na} || {nb}: {bad} mismatches of 20", flush=True)
print("done")
