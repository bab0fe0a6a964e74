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
