"""Gradient checks of the HIP autograd building blocks against torch autograd on the same GPU (torch ops are the
checker here, fp32): LinearFn (+GELU/ELU, residual, odd K), LayerNormFn, AttentionFn (self / masked cross)."""
import numpy as np
import pytest
import torch

from msmd_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("M,N,K,act,res", [(96, 64, 128, 0, False), (333, 512, 72, 1, True), (50, 71, 256, 2, False),
                                           (3552, 512, 512, 1, True)])
def test_linear_fn_grads(M, N, K, act, res):
    from msmd_amd import autograd as ag
    ag.CACHE.clear()
    x = dev(synth.normalish(f"ag/x{M}{K}", (M, K))).requires_grad_(True)
    w = dev(synth.uniform(f"ag/w{N}{K}", (N, K)) / K ** 0.5).requires_grad_(True)
    b = dev(0.1 * synth.uniform(f"ag/b{N}", (N,))).requires_grad_(True)
    r = dev(synth.normalish(f"ag/r{M}{N}", (M, N))).requires_grad_(True) if res else None
    gy = dev(synth.normalish(f"ag/gy{M}{N}", (M, N)))
    y = ag.linear(x, w, b, act, r)
    y.backward(gy)
    got = [t.grad.clone() for t in (x, w, b)] + ([r.grad.clone()] if res else [])
    for t in (x, w, b) + ((r,) if res else ()):
        t.grad = None
    z = torch.nn.functional.linear(x, w, b)
    z = torch.nn.functional.gelu(z) if act == 1 else (torch.nn.functional.elu(z) if act == 2 else z)
    yr = z + r if res else z
    assert rel(y, yr) < 1e-5
    yr.backward(gy)
    ref = [t.grad for t in (x, w, b)] + ([r.grad] if res else [])
    for g, e in zip(got, ref):
        assert g.shape == e.shape and rel(g, e) < 2e-5, (g.shape, rel(g, e))


def test_layernorm_fn_grads():
    from msmd_amd import autograd as ag
    for rows, cols in ((37, 512), (200, 768)):
        x = dev(synth.normalish(f"agln/x{rows}", (rows, cols)) * 1.7 + 0.2).requires_grad_(True)
        g = dev(1 + 0.1 * synth.uniform(f"agln/g{cols}", (cols,))).requires_grad_(True)
        b = dev(0.1 * synth.uniform(f"agln/b{cols}", (cols,))).requires_grad_(True)
        gy = dev(synth.normalish(f"agln/gy{rows}", (rows, cols)))
        y = ag.layer_norm(x, g, b)
        y.backward(gy)
        got = [t.grad.clone() for t in (x, g, b)]
        for t in (x, g, b):
            t.grad = None
        yr = torch.nn.functional.layer_norm(x, (cols,), g, b)
        yr.backward(gy)
        assert rel(y, yr) < 1e-5
        for a, e in zip(got, (x.grad, g.grad, b.grad)):
            assert rel(a, e) < 2e-5


@pytest.mark.parametrize("B,H,Tq,Tk,masked", [(2, 12, 200, 200, False), (2, 8, 111, 110, True), (1, 8, 111, 111, False)])
def test_attention_fn_grads(B, H, Tq, Tk, masked):
    from msmd_amd import autograd as ag
    from oracle import diffusion as od
    d = H * 64
    q = dev(synth.normalish(f"aga/q{Tq}", (B, Tq, d))).requires_grad_(True)
    kv = dev(synth.normalish(f"aga/kv{Tk}", (B, Tk, 2 * d))).requires_grad_(True)
    gy = dev(synth.normalish(f"aga/gy{Tq}", (B, Tq, d)))
    mask = dev(od.alignment_mask(10, 100, 1)) if masked else None
    o = ag.attention(q, kv[..., :d], kv[..., d:], H, 0.125, mask)
    o.backward(gy)
    got = [q.grad.clone(), kv.grad.clone()]
    q.grad = kv.grad = None
    qh = q.view(B, Tq, H, 64).transpose(1, 2)
    kh = kv[..., :d].reshape(B, Tk, H, 64).transpose(1, 2)
    vh = kv[..., d:].reshape(B, Tk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s.masked_fill(mask[None, None], float("-inf"))
    orf = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Tq, d)
    assert rel(o, orf) < 1e-5
    orf.backward(gy)
    assert rel(got[0], q.grad) < 2e-5 and rel(got[1], kv.grad) < 2e-5


@pytest.mark.parametrize("B,H,Tq,Tk,masked", [(2, 3, 200, 200, False), (2, 8, 111, 110, True), (1, 2, 100, 100, False),
                                              (2, 2, 70, 250, False), (1, 1, 5, 3, False), (1, 4, 130, 17, True)])
def test_fused_attention_backward_matches_torch_autograd(B, H, Tq, Tk, masked):
    """msmd_attention_bwd (P recomputed in-kernel) against torch autograd of the same attention in fp32 on the
    same bf16-rounded inputs; bf16 P / dS quantisation bounds the error."""
    from msmd_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Tq + Tk)
    d = H * 64
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    q, kv, do = mk(B, Tq, d), mk(B, Tk, 2 * d), mk(B, Tq, d)
    mask = None
    if masked:
        mask = (torch.rand(Tq, Tk, generator=g) < 0.3)
        mask[:, 0] = False
        mask = mask.to(DEV)
    scale = 0.125
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, kv[..., :d], kv[..., d:]))
    heads = lambda t: t.reshape(B, -1, H, 64).transpose(1, 2)
    s = heads(qf) @ heads(kf).transpose(-1, -2) * scale
    if mask is not None:
        s = s.masked_fill(mask, float("-inf"))
    o = (torch.softmax(s, -1) @ heads(vf)).transpose(1, 2).reshape(B, Tq, d)
    o.backward(do.float())
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    ops.attention_bwd(q, kv[..., :d], kv[..., d:], do, dq, dkv[..., :d], dkv[..., d:], H, scale,
                      mask.to(torch.uint8).contiguous() if mask is not None else None)
    torch.cuda.synchronize()
    for got, ref, name in ((dq, qf.grad, "dq"), (dkv[..., :d], kf.grad, "dk"), (dkv[..., d:], vf.grad, "dv")):
        err = float((got.float() - ref).abs().max())
        assert err < 2e-2 * float(ref.abs().max()) + 1e-3, (name, err, float(ref.abs().max()))


def test_fused_attention_functions_match_unfused_path():
    """self_attention / cross_attention (fused forward + backward) against the explicit-P AttentionFn path."""
    from msmd_amd import autograd as ag
    g = torch.Generator(device="cpu").manual_seed(5)
    B, T, H = 2, 111, 8
    d = H * 64
    qkv0 = (torch.randn(B, T, 3 * d, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    res = []
    for fused in (True, False):
        ag.FUSED_ATTENTION = fused
        qkv = qkv0.clone().requires_grad_(True)
        o = ag.self_attention(qkv, H, 0.125)
        (o.float() * w.float()).sum().backward()
        res.append((o.detach().float(), qkv.grad.float()))
    ag.FUSED_ATTENTION = True
    assert float((res[0][0] - res[1][0]).abs().max()) < 2e-2
    assert float((res[0][1] - res[1][1]).abs().max()) < 3e-2 * float(res[1][1].abs().max())


def test_dropout_kernel_statistics_determinism_and_backward():
    """msmd_dropout: keep rate and 1/(1-p) scaling, pure function of (seed, step, site), residual fusion, and the
    backward (same call on dy) uses the very same mask."""
    from msmd_amd import ops
    n, p = 1 << 20, 0.1
    state = torch.tensor([1234, 0], dtype=torch.int64, device=DEV)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.ones(n + 3, device=DEV, dtype=dt)
        y = ops.dropout(x, p, state, 7)
        keep = (y != 0).float().mean().item()
        assert abs(keep - (1 - p)) < 2e-3, keep
        nz = y[y != 0].float()
        assert float((nz - 1 / (1 - p)).abs().max()) < 1e-2
        assert torch.equal(y, ops.dropout(x, p, state, 7))                  # deterministic
        assert not torch.equal(y, ops.dropout(x, p, state, 8))              # other site
        state[1] += 1
        y2 = ops.dropout(x, p, state, 7)                                    # other step
        assert not torch.equal(y, y2) and abs((y2 != 0).float().mean().item() - (1 - p)) < 2e-3
        state[1] -= 1
        r = torch.full_like(x, 2.0)
        assert torch.equal(ops.dropout(x, p, state, 7, residual=r), (y.float() + 2.0).to(dt))
        dy = torch.randn(n + 3, device=DEV).to(dt)
        dx = ops.dropout(dy, p, state, 7)
        assert torch.equal(dx != 0, (y != 0) & (dy != 0))
    # mask bits are uncorrelated between neighbouring elements
    y = ops.dropout(torch.ones(n, device=DEV), 0.5, state, 3)
    k = (y != 0).float()
    assert abs(float((k[1:] * k[:-1]).mean()) - 0.25) < 3e-3


@pytest.mark.parametrize("Tq,Tk,masked", [(64, 64, False), (111, 60, True), (200, 48, False)])
def test_fused_attention_dropout_forward_and_backward(Tq, Tk, masked):
    """Attention-probability dropout inside the fused kernels.  The keep mask is read back by running the forward
    with V = identity (O then IS the dropped probability matrix); forward and backward are then checked against
    torch autograd of softmax(S) o mask / (1 - p) @ V with that mask."""
    from msmd_amd import ops
    B, H, pd, scale = 2, 2, 0.25, 0.125
    d = H * 64
    g = torch.Generator(device="cpu").manual_seed(Tq * 7 + Tk)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    q, k, v, do = mk(B, Tq, d), mk(B, Tk, d), mk(B, Tk, d), mk(B, Tq, d)
    mask = None
    if masked:
        mask = torch.rand(Tq, Tk, generator=g) < 0.3
        mask[:, 0] = False
        mask = mask.to(DEV)
    m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
    state = torch.tensor([99, 5], dtype=torch.int64, device=DEV)
    site = 11
    eye = torch.zeros(B, Tk, d, device=DEV, dtype=torch.bfloat16)
    for h in range(H):
        eye[:, :, h * 64:h * 64 + Tk] = torch.eye(Tk, device=DEV, dtype=torch.bfloat16)
    pdrop = ops.attention(q, k, eye, H, scale, m8, p_drop=pd, rng_state=state, site=site).float()
    pdrop = pdrop.reshape(B, Tq, H, 64)[..., :Tk].permute(0, 2, 1, 3)            # (B, H, Tq, Tk)
    keep = pdrop != 0
    heads = lambda t: t.reshape(B, -1, H, 64).transpose(1, 2)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    s = heads(qf) @ heads(kf).transpose(-1, -2) * scale
    if mask is not None:
        s = s.masked_fill(mask, float("-inf"))
    P = torch.softmax(s, -1)
    valid = P > 1e-6                                   # tiny probabilities may round to 0 in bf16: not "dropped"
    rate = keep[valid].float().mean().item()
    assert abs(rate - (1 - pd)) < 0.03, rate
    # the tiled kernel (fp32 operands take it; 16-bit ones take the whole-sequence kernel at these lengths) draws the SAME mask
    p32 = ops.attention(q.float(), k.float(), eye.float(), H, scale, m8, p_drop=pd, rng_state=state, site=site)
    keep32 = p32.reshape(B, Tq, H, 64)[..., :Tk].permute(0, 2, 1, 3) != 0
    assert torch.equal(keep32[valid], keep[valid])
    assert float((pdrop - P.detach() * keep / (1 - pd)).abs().max()) < 1e-2
    o_ref = ((P * keep / (1 - pd)) @ heads(vf)).transpose(1, 2).reshape(B, Tq, d)
    o_ref.backward(do.float())
    o = ops.attention(q, k, v, H, scale, m8, p_drop=pd, rng_state=state, site=site)
    assert float((o.float() - o_ref.detach()).abs().max()) < 3e-2
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, do, dq, dk, dv, H, scale, m8, pd, state, site)
    torch.cuda.synchronize()
    for got, ref, name in ((dq, qf.grad, "dq"), (dk, kf.grad, "dk"), (dv, vf.grad, "dv")):
        err = float((got.float() - ref).abs().max())
        assert err < 2.5e-2 * float(ref.abs().max()) + 1e-3, (name, err, float(ref.abs().max()))


@pytest.mark.parametrize("act_name,with_res", [("gelu", True), ("none", True), ("gelu", False)])
def test_fused_linear_activation_dropout_epilogue(act_name, with_res):
    """msmd_gemm_ex: y = dropout(act(x W^T + b)) + residual in one launch (z written for the backward), and the
    one-pass backward msmd_act_bwd_dropout: same Philox mask as msmd_dropout on the (M, N) output, forward and all four
    gradients against torch autograd in fp32 with that mask."""
    from msmd_amd import autograd as ag, ops
    M, K, N, p, site = 333, 256, 512, 0.2, 41
    g = torch.Generator(device="cpu").manual_seed(11)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    x, res, gy = mk(M, K), mk(M, N), mk(M, N)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV).requires_grad_(True)
    act = ops.ACT_GELU if act_name == "gelu" else ops.ACT_NONE
    state = torch.tensor([77, 3], dtype=torch.int64, device=DEV)
    old = (ag.TrainNoise.active, ag.TrainNoise.state)
    ag.TrainNoise.active, ag.TrainNoise.state = True, state
    try:
        xq = x.clone().requires_grad_(True)
        rq = res.clone().requires_grad_(True)
        y = ag.LinearFn.apply(xq, w, b, rq if with_res else None, act, p, site)
        y.backward(gy)
        keep = ops.dropout(torch.ones(M, N, device=DEV), p, state, site) != 0
        xf = x.float().requires_grad_(True)
        wf = w.detach().to(torch.bfloat16).float().requires_grad_(True)
        bf = b.detach().clone().requires_grad_(True)
        rf = res.float().requires_grad_(True)
        z = xf @ wf.t() + bf
        a = torch.nn.functional.gelu(z) if act_name == "gelu" else z
        yr = a * keep / (1 - p) + (rf if with_res else 0)
        yr.backward(gy.float())
        torch.cuda.synchronize()
        assert float((y.float() - yr.detach()).abs().max()) < 3e-2
        rel = lambda got, ref: float((got.float() - ref).abs().max() / (ref.abs().max() + 1e-9))
        assert rel(xq.grad, xf.grad) < 2e-2 and rel(w.grad, wf.grad) < 2e-2 and rel(b.grad, bf.grad) < 2e-2
        if with_res:
            assert rel(rq.grad, rf.grad) < 1e-2
    finally:
        ag.TrainNoise.active, ag.TrainNoise.state = old


@pytest.mark.gpu
def test_weight_arena_refresh_matches_per_weight_casts():
    """msmd_cast_transpose_multi: one launch == the per-weight bf16 cast + transpose, bit for bit, and LinearFn
    picks the arena views up through CACHE.persistent."""
    from msmd_amd import autograd as ag, dp
    torch.manual_seed(3)
    shapes = [(64, 32), (8, 8), (3072, 768), (40, 104), (512, 356), (7, 16), (768, 3072)]
    params = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes] + \
             [torch.nn.Parameter(torch.randn(5, 16, 3, device="cuda"))]
    flat = dp.flatten_parameters(params)
    arena = ag.WeightArena(flat, params)
    assert arena.n == 5                                     # (512, 356), (7, 16) and the 3-D tensor are skipped
    for p in params[:5]:
        if p.shape in [(512, 356), (7, 16)]:
            continue
        wc, wct = ag.CACHE.get(p, torch.bfloat16)
        assert wc.data_ptr() >= arena.cast.data_ptr() and wc.data_ptr() < arena.cast.data_ptr() + arena.cast.numel() * 2
        assert torch.equal(wc, p.detach().bfloat16())
        assert torch.equal(wct, p.detach().bfloat16().t().contiguous())
    with torch.no_grad():
        flat.mul_(0.5)
    arena.refresh()
    wc, wct = ag.CACHE.get(params[2], torch.bfloat16)
    assert torch.equal(wc, params[2].detach().bfloat16()) and torch.equal(wct, params[2].detach().bfloat16().t())
    # an in-place write behind the arena's back (e.g. load_state_dict) is noticed: fresh cast until the next refresh
    with torch.no_grad():
        params[0].add_(1.0)
    wc0, _ = ag.CACHE.get(params[0], torch.bfloat16)
    assert torch.equal(wc0, params[0].detach().bfloat16()) and wc0.data_ptr() != arena.cast.data_ptr()
    arena.refresh()
    wc0, _ = ag.CACHE.get(params[0], torch.bfloat16)
    assert torch.equal(wc0, params[0].detach().bfloat16()) and wc0.data_ptr() == arena.cast.data_ptr()
    # a dead parameter's entry is dropped, not served to a new tensor that recycles its id()
    key = (id(params[0]), torch.bfloat16)
    assert key in ag.CACHE.persistent
    ag.CACHE.persistent.clear()


def test_ffn_node_matches_two_linear_nodes_and_gemm_actbwd_matches_the_unfused_pair():
    """autograd.ffn (FFNFn: one node, msmd_gemm_actbwd in the middle of its backward) against the two LinearFn nodes it
    replaces -- same dropout sites, so same masks: outputs bit-equal, gradients within the 16-bit rounding of one
    intermediate (the fused epilogue evaluates GELU' with the fast erf / exp)."""
    import math
    from msmd_amd import autograd as ag, ops
    g = torch.Generator(device="cpu").manual_seed(3)
    M, d, dff = 333, 256, 1024
    x0 = torch.randn(7, M // 7 + 1, d, generator=g)[:, :47].contiguous().to(DEV, torch.bfloat16)
    res0 = torch.randn_like(x0.float()).to(DEV, torch.bfloat16)
    w1 = (torch.randn(dff, d, generator=g) / math.sqrt(d)).to(DEV).requires_grad_(True)
    b1 = (torch.randn(dff, generator=g) * 0.1).to(DEV).requires_grad_(True)
    w2 = (torch.randn(d, dff, generator=g) / math.sqrt(dff)).to(DEV).requires_grad_(True)
    b2 = (torch.randn(d, generator=g) * 0.1).to(DEV).requires_grad_(True)
    dy = torch.randn_like(x0.float()).to(torch.bfloat16)
    ag.TrainNoise.state = torch.tensor([1234, 5], dtype=torch.int64, device=DEV)
    outs = {}
    for fused in (False, True):
        for active in (False, True):
            ag.TrainNoise.active, ag.TrainNoise.site = active, 0
            x = x0.clone().requires_grad_(True)
            res = res0.clone().requires_grad_(True)
            for t in (w1, b1, w2, b2):
                t.grad = None
            if fused:
                y = ag.ffn(x, w1, b1, w2, b2, 0.1, 0.2, residual=res)
            else:
                y = ag.linear_dropout(ag.linear_dropout(x, w1, b1, 0.1, act=ops.ACT_GELU), w2, b2, 0.2, residual=res)
            y.backward(dy)
            torch.cuda.synchronize()
            outs[fused, active] = [y.detach().float()] + [t.grad.float().clone() for t in (x, res, w1, b1, w2, b2)]
    ag.TrainNoise.active = False
    for active in (False, True):
        a, b = outs[False, active], outs[True, active]
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])          # forward output, residual gradient
        for name, u, v in zip(("dx", "dw1", "db1", "dw2", "db2"), [a[1]] + a[3:], [b[1]] + b[3:]):
            scale = float(u.abs().max())
            assert float((u - v).abs().max()) <= 1.5e-2 * scale + 1e-6, (name, active, float((u - v).abs().max()), scale)
        assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])          # dW2, db2 do not pass through the fused launch
    # dropout really on in the active pass
    assert not torch.equal(outs[True, False][0], outs[True, True][0])
    # x + FFN(x) (residual is the input itself): one gradient for x, equal to the sum autograd forms from the two-node form
    ag.TrainNoise.active, ag.TrainNoise.site = True, 0
    xa = x0.clone().requires_grad_(True)
    ag.ffn(xa, w1, b1, w2, b2, 0.1, 0.2, residual=xa).backward(dy)
    ag.TrainNoise.site = 0
    xb = x0.clone().requires_grad_(True)
    ag.linear_dropout(ag.linear_dropout(xb, w1, b1, 0.1, act=ops.ACT_GELU), w2, b2, 0.2, residual=xb).backward(dy)
    ag.TrainNoise.active = False
    torch.cuda.synchronize()
    d = (xa.grad.float() - xb.grad.float()).abs().max()
    assert float(d) <= 1.5e-2 * float(xb.grad.float().abs().max())


@pytest.mark.gpu
def test_layernorm_backward_hands_the_dropped_gradient_to_the_linear_in_front_of_it(monkeypatch):
    """Post-LN blocks h = LN(res + dropout(Linear(a))) and LN(x + FFN(x)): msmd_layernorm_bwd_dropout writes the dropped copy
    of dx the Linear's backward needs (autograd._tag_dropped / _dropped_by_producer).  Bit-equal gradients with the side
    channel on and off; the fused launch really is taken (no msmd_dropout call in the backward); and a LayerNorm whose
    input has a SECOND consumer must not have its dropped copy used (the engine hands the Linear a sum)."""
    import math
    from msmd_amd import autograd as ag, ops
    g = torch.Generator(device="cpu").manual_seed(11)
    B, T, d, dff = 3, 37, 256, 512
    mk = lambda *s: torch.randn(*s, generator=g)
    a0, r0 = mk(B, T, d).to(DEV, torch.bfloat16), mk(B, T, d).to(DEV, torch.bfloat16)
    P = [(mk(d, d) / math.sqrt(d)), mk(d) * 0.1, 1 + 0.1 * mk(d), 0.1 * mk(d), mk(dff, d) / math.sqrt(d), mk(dff) * 0.1,
         mk(d, dff) / math.sqrt(dff), mk(d) * 0.1, 1 + 0.1 * mk(d), 0.1 * mk(d)]
    P = [t.to(DEV).requires_grad_(True) for t in P]
    wo, bo, g1, be1, w1, b1, w2, b2, g2, be2 = P
    dy = mk(B, T, d).to(DEV, torch.bfloat16)
    ag.TrainNoise.state = torch.tensor([77, 3], dtype=torch.int64, device=DEV)
    calls = []
    real_dropout = ops.dropout
    monkeypatch.setattr(ops, "dropout", lambda *a_, **k_: (calls.append(1), real_dropout(*a_, **k_))[1])

    def run(fuse, second_consumer):
        monkeypatch.setattr(ag, "FUSE_LN_DROPOUT_BWD", fuse)
        ag.TrainNoise.active, ag.TrainNoise.site = True, 0
        a, r = a0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        for t in P:
            t.grad = None
        u = ag.linear_dropout(a, wo, bo, 0.1, residual=r)
        h = ag.layer_norm(u, g1, be1)
        y = ag.layer_norm(ag.ffn(h, w1, b1, w2, b2, 0.1, 0.2, residual=h), g2, be2)
        if second_consumer:
            y = y + u * 0.5
        torch.cuda.synchronize()
        del calls[:]
        y.backward(dy)
        torch.cuda.synchronize()
        n = len(calls)
        ag.TrainNoise.active = False
        return [t.grad.float().clone() for t in (a, r, *P)], n

    ref, n_ref = run(False, False)
    got, n_got = run(True, False)
    assert n_ref == 2 and n_got == 0, (n_ref, n_got)
    for i, (u, v) in enumerate(zip(ref, got)):
        assert torch.equal(u, v), i
    ref2, _ = run(False, True)
    got2, n2 = run(True, True)
    assert n2 == 1            # the out-projection's gradient is a sum now: its own dropout launch; the FFN still takes the copy
    for i, (u, v) in enumerate(zip(ref2, got2)):
        assert torch.equal(u, v), i


@pytest.mark.gpu
def test_conv3_node_matches_the_shifted_view_linear_it_replaces(monkeypatch):
    """train_graph.Conv3Fn (padded copy + windowed GEMM; windowed data-gradient GEMM; TN GEMM over the windows) against the
    pad + cat-of-shifted-views + Linear path, and both against torch's conv1d autograd in fp32."""
    import math
    from msmd_amd import ops, train_graph as tg
    g = torch.Generator(device="cpu").manual_seed(5)
    B, T, C, Co = 3, 100, 512, 256
    x0 = torch.randn(B, T, C, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(Co, C, 3, generator=g) / math.sqrt(3 * C)).to(DEV).requires_grad_(True)
    b = (torch.randn(Co, generator=g) * 0.1).to(DEV).requires_grad_(True)
    dy = torch.randn(B, T, Co, generator=g).to(DEV, torch.bfloat16)
    res = {}
    for fn in (True, False):
        monkeypatch.setattr(tg, "USE_CONV3_FN", fn)
        x = x0.clone().requires_grad_(True)
        w.grad = b.grad = None
        y = tg._conv3(x, w, b, ops.ACT_NONE)
        y.backward(dy)
        torch.cuda.synchronize()
        res[fn] = (y.detach().float(), x.grad.float(), w.grad.clone(), b.grad.clone())
    xr = x0.float().requires_grad_(True)
    wr, br = w.detach().to(torch.bfloat16).float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, br, padding=1).transpose(1, 2)
    yr.backward(dy.float())
    ref = (yr.detach(), xr.grad, wr.grad, br.grad)
    assert torch.equal(res[True][0], res[False][0])                    # same K order, same kernel: bit-equal forward
    for name, a, c, r in zip(("y", "dx", "dw", "db"), res[True], res[False], ref):
        scale = float(r.abs().max())
        assert float((a - r).abs().max()) <= 1e-2 * scale, (name, float((a - r).abs().max()), scale)
        assert float((a - c).abs().max()) <= 1e-2 * scale, (name, "vs shifted views")


@pytest.mark.gpu
def test_junction_hands_the_residual_gradient_to_the_projection_gemm(monkeypatch):
    """x -> QKV projection -> attention -> out-projection(+ x): with autograd.Junction the gradient of x comes out of the QKV
    node's data-gradient GEMM (the residual gradient is its epilogue operand) instead of an accumulation kernel: same
    gradients as the plain graph up to one 16-bit rounding, parameters' gradients bit-equal."""
    import math
    from msmd_amd import autograd as ag
    g = torch.Generator(device="cpu").manual_seed(21)
    B, T, d, H = 4, 60, 256, 4
    x0 = torch.randn(B, T, d, generator=g).to(DEV, torch.bfloat16)
    wqkv = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).to(DEV).requires_grad_(True)
    bqkv = (torch.randn(3 * d, generator=g) * 0.1).to(DEV).requires_grad_(True)
    wo = (torch.randn(d, d, generator=g) / math.sqrt(d)).to(DEV).requires_grad_(True)
    bo = (torch.randn(d, generator=g) * 0.1).to(DEV).requires_grad_(True)
    dy = torch.randn(B, T, d, generator=g).to(DEV, torch.bfloat16)
    ag.TrainNoise.state = torch.tensor([5, 9], dtype=torch.int64, device=DEV)
    out = {}
    for use in (False, True):
        monkeypatch.setattr(ag, "USE_JUNCTIONS", use)
        ag.TrainNoise.active, ag.TrainNoise.site = True, 0
        x = x0.clone().requires_grad_(True)
        for t in (wqkv, bqkv, wo, bo):
            t.grad = None
        J = ag.Junction()
        a = ag.self_attention(ag.linear(x, wqkv, bqkv, junction_in=J), H, 0.125, p_drop=0.1)
        y = ag.linear_dropout(a, wo, bo, 0.1, residual=x, junction_out=J)
        y.backward(dy)
        torch.cuda.synchronize()
        ag.TrainNoise.active = False
        assert J.paired == use and J.pending is None
        out[use] = [y.detach().float(), x.grad.float()] + [t.grad.clone() for t in (wqkv, bqkv, wo, bo)]
    assert torch.equal(out[False][0], out[True][0])
    for u, v in zip(out[False][2:], out[True][2:]):
        assert torch.equal(u, v)
    scale = float(out[False][1].abs().max())
    assert float((out[False][1] - out[True][1]).abs().max()) <= 1e-2 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("Tq,Tk,masked", [(200, 200, False), (250, 250, False), (111, 110, True), (300, 140, False)])
def test_attention_dropout_mask_is_the_same_in_forward_and_backward_at_long_sequences(Tq, Tk, masked):
    """The keep mask of attention-probability dropout at the lengths the training step runs (whole-sequence forward kernel with
    13 / 17 key fragments, one or two workgroups per head; backward with 7 / 13 / 16): with V = one-hot rows the forward output IS
    P_drop[q, key_c], with dO = ones on ONE query row the backward's dV IS P_drop[q*, :] -- the two must agree, zeros included."""
    from msmd_amd import ops
    B, H, pd, scale = 2, 3, 0.3, 0.125
    d = H * 64
    g = torch.Generator(device="cpu").manual_seed(Tq + Tk)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    q, k = mk(B, Tq, d), mk(B, Tk, d)
    m8 = None
    if masked:
        m = torch.rand(Tq, Tk, generator=g) < 0.3
        m[:, 0] = False
        m8 = m.to(torch.uint8).contiguous().to(DEV)
    keys = torch.arange(64) * (Tk - 1) // 63                     # 64 distinct keys spread over the row (Tk >= 64)
    v = torch.zeros(B, Tk, d, device=DEV, dtype=torch.bfloat16)
    for h in range(H):
        v[:, keys, h * 64 + torch.arange(64)] = 1.0
    state = torch.tensor([31, 2], dtype=torch.int64, device=DEV)
    o = ops.attention(q, k, v, H, scale, m8, p_drop=pd, rng_state=state, site=5).float()       # o[b, q, h*64+c] = P_drop[q, keys[c]]
    o0 = ops.attention(q, k, v, H, scale, m8).float()
    for qs in (0, Tq // 2 + 3, Tq - 1):
        do = torch.zeros(B, Tq, d, device=DEV, dtype=torch.bfloat16)
        do[:, qs] = 1.0
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(k)
        ops.attention_bwd(q, k, v, do, dq, dk, dv, H, scale, m8, pd, state, 5)
        torch.cuda.synchronize()
        for h in range(H):
            fwd = o[:, qs, h * 64:(h + 1) * 64]                                  # (B, 64): P_drop[qs, keys[c]]
            bwd = dv[:, keys.to(DEV), h * 64].float()                           # (B, 64): P_drop[qs, keys[c]] (any column)
            p0 = o0[:, qs, h * 64:(h + 1) * 64]
            sure = p0 > 2e-3                                                     # clear of bf16 underflow
            assert torch.equal((fwd > 0)[sure], (bwd > 0)[sure]), (qs, h)
            assert float((fwd - bwd).abs().max()) <= 0.03 * float(fwd.abs().max()) + 1e-4
            kept = (fwd > 0)[sure].float().mean()
            assert float((fwd[sure & (fwd > 0)] / p0[sure & (fwd > 0)] - 1 / (1 - pd)).abs().max()) < 0.05
    rate = ((o > 0) & (o0 > 2e-3)).float().sum() / (o0 > 2e-3).float().sum()
    assert abs(float(rate) - (1 - pd)) < 0.02, float(rate)


@pytest.mark.gpu
def test_layerdrop_select_node_matches_torch_where():
    from msmd_amd import train_graph as tg
    g = torch.Generator(device="cpu").manual_seed(2)
    a0 = torch.randn(4, 9, 64, generator=g).to(DEV, torch.bfloat16)
    b0 = torch.randn(4, 9, 64, generator=g).to(DEV, torch.bfloat16)
    dy = torch.randn(4, 9, 64, generator=g).to(DEV, torch.bfloat16)
    for flag in (torch.tensor(True, device=DEV), torch.tensor(False, device=DEV)):
        res = []
        for fn in (lambda f, a, b: tg.LayerDropSelectFn.apply(f, a, b), torch.where):
            a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            y = fn(flag, a * 2, b * 3)
            y.backward(dy)
            res.append((y.detach(), a.grad, b.grad))
        for u, v in zip(*res):
            assert torch.equal(u, v)


@pytest.mark.gpu
def test_fused_alias_operand_gets_gradients_without_a_grad_input_and_sees_in_place_writes_to_its_owners():
    """autograd.FusedAlias (Q | K | V of an encoder layer as ONE arena view): (1) its node writes the owners' gradients even when
    the activation fed to it does not require grad (the view itself carries no autograd history); (2) an in-place write to an
    OWNER parameter -- load_state_dict, p.copy_(), a finite-difference probe -- moves the owner's version counter, not the
    view's: the cached bf16 casts of the view must be dropped for it all the same; (3) with USE_GEMM_TN off the alias branch
    takes the generic weight-gradient products and adds them into the same arena region."""
    import math
    from msmd_amd import autograd as ag
    g = torch.Generator(device="cpu").manual_seed(3)
    d, M = 128, 96
    flat = torch.zeros(3 * d * d + 3 * d, device=DEV)
    garena = torch.zeros_like(flat)
    owners = []
    for i in range(3):      # q, k, v weights next to each other, then the three biases: parameters are views of the flat arena
        w = torch.nn.Parameter(torch.empty(0, device=DEV))
        w.data = flat[i * d * d:(i + 1) * d * d].view(d, d)
        w.data.copy_((torch.randn(d, d, generator=g) / math.sqrt(d)).to(DEV))
        owners.append(w)
    for i in range(3):
        b = torch.nn.Parameter(torch.empty(0, device=DEV))
        b.data = flat[3 * d * d + i * d:3 * d * d + (i + 1) * d]
        b.data.copy_((torch.randn(d, generator=g) * 0.1).to(DEV))
        owners.append(b)
    fa = ag.FusedAlias(flat[:3 * d * d].view(3 * d, d), flat[3 * d * d:], garena[:3 * d * d].view(3 * d, d), garena[3 * d * d:], owners)
    arena = ag.WeightArena(flat, [], aliases=[fa])
    x = torch.randn(M, d, generator=g).to(DEV, torch.bfloat16)          # requires_grad False: e.g. a frozen feature map in front
    dy = torch.randn(M, 3 * d, generator=g).to(DEV, torch.bfloat16)

    def reference():
        wf = flat[:3 * d * d].view(3 * d, d).to(torch.bfloat16).float()
        return x.float() @ wf.t() + flat[3 * d * d:], dy.float().t() @ x.float(), dy.float().sum(0)

    try:
        y = ag.linear_alias(x, fa)
        assert y.requires_grad, "the alias node must be part of the graph although no tensor input requires grad"
        y.backward(dy)
        y0, dw0, db0 = reference()
        assert float((y.float() - y0).abs().max()) <= 0.06
        assert float((garena[:3 * d * d].view(3 * d, d) - dw0).abs().max()) <= 2e-2 * float(dw0.abs().max())
        assert float((garena[3 * d * d:] - db0).abs().max()) <= 2e-2 * float(db0.abs().max())
        # (2) perturb K's weight in place through the OWNER (exactly what load_state_dict does): the next forward must see it
        with torch.no_grad():
            owners[1].mul_(-3.0)
        y2 = ag.linear_alias(x, fa)
        y2r, _, _ = reference()
        assert float((y2.float() - y2r).abs().max()) <= 0.2, "stale bf16 cast of an aliased weight after an in-place owner write"
        assert float((y2.float() - y.float()).abs().max()) > 1.0
        arena.refresh()      # the optimizer-step path: casts rewritten, versions recorded
        y3 = ag.linear_alias(x, fa)
        assert torch.equal(y3, y2)
        # (3) the generic weight-gradient path adds into the same region
        garena.zero_()
        old = ag.USE_GEMM_TN
        ag.USE_GEMM_TN = False
        try:
            ag.linear_alias(x, fa).backward(dy)
        finally:
            ag.USE_GEMM_TN = old
        _, dw1, db1 = reference()
        assert float((garena[:3 * d * d].view(3 * d, d) - dw1).abs().max()) <= 2e-2 * float(dw1.abs().max())
        assert float((garena[3 * d * d:] - db1).abs().max()) <= 2e-2 * float(db1.abs().max())
    finally:
        arena.release()
