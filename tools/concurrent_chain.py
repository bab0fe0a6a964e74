"""Dev experiment: minimal reproducer hunt -- chains of dependent ops on two streams (PIPE GEMM variant 17 by default)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msmd_amd import ops
for kv in os.environ.get("TUNE", "").split(","):
    if kv: ops.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
torch.manual_seed(0)
dev, bf = "cuda", torch.bfloat16
M = 6400
W = [(torch.randn(768, 768, device=dev) * 768 ** -0.5).to(bf) for _ in range(4)]
W1 = (torch.randn(3072, 768, device=dev) * 768 ** -0.5).to(bf); W2 = (torch.randn(768, 3072, device=dev) * 3072 ** -0.5).to(bf)
Wq = (torch.randn(2304, 768, device=dev) * 768 ** -0.5).to(bf)
b768 = torch.zeros(768, device=dev); b3072 = torch.zeros(3072, device=dev); b2304 = torch.zeros(2304, device=dev)
g = torch.ones(768, device=dev); be = torch.zeros(768, device=dev)
xs = [torch.randn(M, 768, device=dev).to(bf) for _ in range(2)]
def chain_gemm(x, L=12):
    h = x
    for l in range(L): h = ops.gemm(h, W[l % 4], b768)
    return h.float()
def chain_ffn(x, L=12):
    h = x
    for l in range(L): h = ops.gemm(ops.gemm(h, W1, b3072, None, ops.ACT_GELU), W2, b768)
    return h.float()
def chain_ffn_ln(x, L=12):
    h = x
    for l in range(L): h = ops.layernorm(ops.gemm(ops.gemm(h, W1, b3072, None, ops.ACT_GELU), W2, b768), g, be, residual=h)
    return h.float()
def chain_attn(x, L=12):
    h = x
    for l in range(L):
        qkv = ops.gemm(h, Wq, b2304).view(32, 200, 2304)
        a = ops.attention(qkv[..., :768], qkv[..., 768:1536], qkv[..., 1536:], 12, 0.125).view(M, 768)
        h = ops.layernorm(ops.gemm(a, W[0], b768), g, be, residual=h)
    return h.float()
def chain_layer(x, L=12):
    h = x
    for l in range(L):
        qkv = ops.gemm(h, Wq, b2304).view(32, 200, 2304)
        a = ops.attention(qkv[..., :768], qkv[..., 768:1536], qkv[..., 1536:], 12, 0.125).view(M, 768)
        h = ops.layernorm(ops.gemm(a, W[0], b768), g, be, residual=h)
        h = ops.layernorm(ops.gemm(ops.gemm(h, W1, b3072, None, ops.ACT_GELU), W2, b768), g, be, residual=h)
    return h.float()
chains = dict(gemm=chain_gemm, ffn=chain_ffn, ffn_ln=chain_ffn_ln, attn=chain_attn, layer=chain_layer)
R = int(os.environ.get("REPS", "60"))
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for name in (sys.argv[1:] or list(chains)):
    fn = chains[name]
    refs = []
    for x in xs:
        o = fn(x); torch.cuda.synchronize(); refs.append(o.clone())
    bad = 0; rows = []
    for rep in range(R):
        for st in s: st.wait_stream(torch.cuda.current_stream())
        for k in range(2):
            outs = []
            for i in range(2):
                with torch.cuda.stream(s[i]): outs.append(fn(xs[i]))
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(outs[i], refs[i]):
                bad += 1
                idx = torch.nonzero((outs[i] != refs[i]).any(1)).flatten()
                if len(rows) < 3: rows.append((i, int(idx.min()), int(idx.max()), int(idx.numel())))
    print(f"{name:8s}: {bad} of {2 * R} concurrent results differ; (stream, first row, last row, rows) {rows}", flush=True)
