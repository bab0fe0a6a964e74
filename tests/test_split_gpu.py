"""GPU parity of the parity-grade speed mode (compute_dtype "f16x2": contractions on MSMD_F16X2 split pairs, three f16
MFMAs per k-step; include/msmd_hip.h) against the goldens of the imported reference and the numpy oracle.

Tolerance: the north_star's 1e-4 max-abs on the motion coefficients -- the SAME bound the exact-fp32 mode is held to.
Kernel-level tests compare with float64 products and state their bound next to the assertion."""
import math
from unittest import mock

import numpy as np
import pytest
import torch

from msmd_amd import synth
from msmd_amd.config import default_args

from conftest import load_golden
from helpers import denoiser_inputs, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from msmd_amd import ops as _ops
    return _ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def test_split_roundtrip_and_layout():
    """x -> [hi | lo] blocks -> x: |rel err| <= 2^-22 from the pair (+ 2^-24 fp32 recombination), padding columns zero,
    and the physical layout is 32-element blocks hi then lo with hi = RN_f16(x), lo = RN_f16((x - hi) * 2048)."""
    o = ops()
    x = (synth.normalish("split/x", (37, 96)) * np.exp(synth.uniform("split/e", (37, 96), -6, 6))).astype(np.float32)
    s = o.to_split(dev(x))
    assert s.shape == (37, 96) and s.t.shape == (37, 192) and s.t.dtype == torch.float16
    back = s.float().cpu().numpy()
    assert np.all(np.abs(back - x) <= np.abs(x) * 1.25 * 2.0 ** -22 + 2.0 ** -35)   # lo is subnormal below |x| ~ 6e-5
    phys = s.t.cpu().numpy().reshape(37, 3, 2, 32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * 2048.0).astype(np.float16)
    assert np.array_equal(phys[:, :, 0, :].reshape(37, 96), hi)
    assert np.array_equal(phys[:, :, 1, :].reshape(37, 96), lo)
    # zero padding 67 -> 96 columns
    y = synth.normalish("split/y", (5, 67))
    sp = o.to_split(dev(y), 96)
    b2 = sp.float().cpu().numpy()
    assert np.all(b2[:, 67:] == 0) and np.allclose(b2[:, :67], y, rtol=3e-7, atol=0)
    # last-dim slices on block boundaries address the right columns
    assert np.allclose(s[:, 32:64].float().cpu().numpy(), x[:, 32:64], rtol=3e-7)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 200, 96), (37, 71, 256), (1000, 512, 1536), (64, 48, 6144),
                                   (5, 512, 384), (6400, 768, 768)])
def test_split_gemm_matches_float64(M, N, K):
    """C = act(A W^T + b) + R on split operands vs a float64 product of the SAME fp32 inputs: max |err| <= 4e-6 * the
    row scale sqrt(K) * |a||w| -- what the exact-fp32 MFMA kernel achieves (asserted side by side)."""
    o = ops()
    a = synth.normalish(f"sgemm/a/{M}x{K}", (M, K))
    w = (synth.uniform(f"sgemm/w/{N}x{K}", (N, K), -1, 1) / math.sqrt(K)).astype(np.float32)
    b = synth.uniform(f"sgemm/b/{N}", (N,), -0.5, 0.5)
    r = synth.normalish(f"sgemm/r/{M}x{N}", (M, N))
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    ws = o.to_split(dev(w))
    for variant in (0, 1, 5, 14):   # heuristic, 128 x 128, 64 x 64, 256 x 64: per-call hint (MSMD_GEMM_VARIANT), no library state
        c = o.gemm(o.to_split(dev(a)), ws, dev(b), variant=variant).cpu().numpy()
        assert maxabs(c, ref) < 4e-6, (variant, maxabs(c, ref))
    c32 = o.gemm(dev(a), dev(w), dev(b)).cpu().numpy()
    c = o.gemm(dev(a), ws, dev(b)).cpu().numpy()          # fp32 activations are converted on the way in
    assert maxabs(c, ref) < 4e-6 and maxabs(c, ref) < 4 * maxabs(c32, ref) + 1e-7
    # GELU + residual epilogue, fp32 output
    from oracle import nn as onn
    want = onn.gelu(ref) + r
    got = o.gemm(dev(a), ws, dev(b), dev(r), o.ACT_GELU).cpu().numpy()
    assert maxabs(got, want) < 6e-6
    # split output (and split residual) for N % 32 == 0
    if N % 32 == 0:
        cs = o.gemm(o.to_split(dev(a)), ws, dev(b), o.to_split(dev(r)), o.ACT_GELU, out_dtype=o.SPLIT)
        assert isinstance(cs, o.Split) and cs.shape == (M, N)
        assert maxabs(cs.float().cpu().numpy(), want) < 6e-6


@pytest.mark.parametrize("below_32", [False, True])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1000, 512, 1536), (6400, 768, 768), (5, 512, 384), (777, 256, 96),
                                   (20000, 1024, 64), (6400, 2304, 768)])
def test_split_gemm_256_tile_8_phase_kernel_matches_float64(M, N, K, below_32):
    """Variant 80 of the split GEMM (gemm8_kernel, SPLIT: 256 x 256 tiles, the 8-phase schedule, cross terms folded once per K
    tile) against a float64 product of the same fp32 inputs, at the bound the 128 x 128 split kernel is held to: plain, GELU +
    residual, fp32 and split output; ragged M, one tile, an odd K-tile count (96 / 32), the shortest K loop (two K tiles) and a
    grid of 316 tiles (workgroups that walk a second tile).  The hint is honoured only for calls the kernel takes (N % 256 == 0,
    K % 32 == 0, K >= 64): that it really ran is checked by bits -- it sums the cross terms in another order than variant 1.
    below_32: the weight went through ops.split_weight, gemm() passes MSMD_GEMM_W_BELOW_32 and the kernel keeps one sum in units
    of 2^-11 (W's hi plane x 2^11 in registers) instead of folding; one weight sits at 31.5 to pin the stated range."""
    o = ops()
    a = synth.normalish(f"sgemm8/a/{M}x{K}", (M, K))
    w = (synth.uniform(f"sgemm8/w/{N}x{K}", (N, K), -1, 1) / math.sqrt(K)).astype(np.float32)
    b = synth.uniform(f"sgemm8/b/{N}", (N,), -0.5, 0.5)
    r = synth.normalish(f"sgemm8/r/{M}x{N}", (M, N))
    if below_32:
        w[N // 2, K // 3] = 31.5
        a[:, K // 3] *= 2.0 ** -8         # keeps the row scale the bound below is stated for
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    As, Ws = o.to_split(dev(a)), (o.split_weight(dev(w)) if below_32 else o.to_split(dev(w)))
    assert Ws.below_32 == below_32
    c1 = o.gemm(As, Ws, dev(b), variant=1).cpu().numpy()
    c8 = o.gemm(As, Ws, dev(b), variant=80).cpu().numpy()
    assert maxabs(c8, ref) < 4e-6, maxabs(c8, ref)
    assert maxabs(c8, ref) < 2 * maxabs(c1, ref) + 2e-7          # the per-K-tile fold costs no accuracy worth naming
    if below_32:
        # the fold-free forms of both kernels multiply in ONE order per output element: a launch returns the same bits whichever
        # of them its row count routes it to (a clip alone / in a batch of 32)
        assert np.array_equal(c8, c1)
        for v in (5, 14):
            assert np.array_equal(o.gemm(As, Ws, dev(b), variant=v).cpu().numpy(), c1), v
    elif M * N >= 65536 and K >= 96:
        assert not np.array_equal(c8, c1), "variant 80 returned variant 1's bits: the hint was not followed"
    from oracle import nn as onn
    want = onn.gelu(ref) + r
    got = o.gemm(As, Ws, dev(b), dev(r), o.ACT_GELU, variant=80).cpu().numpy()
    assert maxabs(got, want) < 6e-6
    Rs = o.to_split(dev(r))
    cs = o.gemm(As, Ws, dev(b), Rs, o.ACT_GELU, out_dtype=o.SPLIT, variant=80)
    assert isinstance(cs, o.Split) and cs.shape == (M, N)
    assert maxabs(cs.float().cpu().numpy(), want) < 6e-6
    # the split rows it writes are the pairs split_f16x2 makes of its fp32 results: hi = RN_f16(x), lo = RN_f16((x - hi) 2^11)
    c8s = o.gemm(As, Ws, dev(b), out_dtype=o.SPLIT, variant=80)
    assert torch.equal(c8s.t, o.to_split(dev(c8)).t)
    # the library's own choice (variant 0) is one of the two kernels
    c0 = o.gemm(As, Ws, dev(b)).cpu().numpy()
    assert np.array_equal(c0, c8) or np.array_equal(c0, c1)


def onn_gelu(x):
    from oracle import nn as onn
    return onn.gelu(x)


def test_split_gemm_256_tile_kernel_takes_the_windowed_conv_operand():
    """Conv1d over channels-last split rows (overlapping windows as the A operand) on variant 80: 3 taps x 64 channels, stride 2."""
    o = ops()
    B, T, C, k, stride, Cout = 5, 301, 64, 3, 2, 256
    x = synth.normalish("sconv8/x", (B, T, C))
    w = (synth.uniform("sconv8/w", (Cout, C, k), -1, 1) / math.sqrt(C * k)).astype(np.float32)
    Tout = (T - k) // stride + 1
    win = np.stack([x[:, t * stride:t * stride + k, :] for t in range(Tout)], 1).astype(np.float64)     # (B, Tout, k, C)
    ref = np.einsum("btkc,ock->bto", win, w.astype(np.float64))
    wp = o.to_split(dev(np.ascontiguousarray(w.transpose(0, 2, 1).reshape(Cout, k * C))))
    with o.gemm_defaults(split_variant=80):
        y = o.conv1d_cl(o.to_split(dev(x)), wp, None, kernel=k, stride=stride, out_dtype=torch.float32)
        ys = o.conv1d_cl(o.to_split(dev(x)), wp, None, kernel=k, stride=stride, out_dtype=o.SPLIT)
    with o.gemm_defaults(split_variant=1):
        y1 = o.conv1d_cl(o.to_split(dev(x)), wp, None, kernel=k, stride=stride, out_dtype=torch.float32)
    assert maxabs(y.cpu().numpy(), ref) < 4e-6 and maxabs(ys.float().cpu().numpy(), ref) < 4e-6
    assert not torch.equal(y, y1)
    # ... and with a weight that went through split_weight (MSMD_GEMM_W_BELOW_32) both kernels return the same bits
    wb = o.split_weight(dev(np.ascontiguousarray(w.transpose(0, 2, 1).reshape(Cout, k * C))))
    yb = []
    for v in (80, 1):
        with o.gemm_defaults(split_variant=v):
            yb.append(o.conv1d_cl(o.to_split(dev(x)), wb, None, kernel=k, stride=stride, act=o.ACT_GELU, out_dtype=o.SPLIT))
    assert torch.equal(yb[0].t, yb[1].t) and maxabs(yb[0].float().cpu().numpy(), onn_gelu(ref)) < 6e-6


def test_split_gemm_is_strided_conv1d_and_batched():
    """Windowed A operand (Conv1d over channels-last rows) and the batched / grouped form in split storage."""
    o = ops()
    B, T, C, k, stride, Cout = 3, 41, 64, 3, 2, 96
    x = synth.normalish("sconv/x", (B, T, C))
    w = (synth.uniform("sconv/w", (Cout, C, k), -1, 1) / math.sqrt(C * k)).astype(np.float32)
    Tout = (T - k) // stride + 1
    ref = np.zeros((B, Tout, Cout))
    for t in range(Tout):
        win = x[:, t * stride:t * stride + k, :].astype(np.float64)             # (B, k, C)
        ref[:, t] = np.einsum("bkc,ock->bo", win, w.astype(np.float64))
    wp = o.to_split(dev(np.ascontiguousarray(w.transpose(0, 2, 1).reshape(Cout, k * C))))
    y = o.conv1d_cl(o.to_split(dev(x)), wp, None, kernel=k, stride=stride, out_dtype=torch.float32)
    assert maxabs(y.cpu().numpy(), ref) < 4e-6
    ys = o.conv1d_cl(o.to_split(dev(x)), wp, None, kernel=k, stride=stride, out_dtype=o.SPLIT)
    assert maxabs(ys.float().cpu().numpy(), ref) < 4e-6
    # grouped positional conv shape: G groups, 48 -> 64 padded channels, k = 8 taps
    G, cg, kk, T2 = 4, 48, 8, 20
    h = synth.normalish("sgrp/h", (2, T2, G * cg))
    wg = (synth.uniform("sgrp/w", (G, cg, kk, cg), -1, 1) / math.sqrt(cg * kk)).astype(np.float32)  # [g][co][tap][ci]
    xp = o.group_pad(dev(h), G, kk // 2, cg_out=64, split=True)
    assert isinstance(xp, o.Split) and xp.shape == (2, G, T2 + kk, 64)
    xp32 = o.group_pad(dev(h), G, kk // 2, cg_out=64).cpu().numpy()
    assert np.allclose(xp.float().cpu().numpy(), xp32, rtol=3e-7, atol=0)
    wpad = np.zeros((G, cg, kk, 64), np.float32)
    wpad[..., :cg] = wg
    y = torch.empty(2, T2, G * cg, device=DEV)
    Tp = T2 + kk
    o.gemm(xp, o.to_split(dev(wpad.reshape(G, cg, kk * 64))), None, None, o.ACT_NONE, out=y, M=2 * T2, N=cg, K=kk * 64,
           lda=64, rows_per_batch=T2, a_batch_stride=G * Tp * 64, ldw=kk * 64, ldc=G * cg, batch=G, strideA=Tp * 64,
           strideW=cg * kk * 64, strideC=cg)
    ref = np.zeros((2, T2, G * cg))
    hp = np.zeros((2, Tp, G * cg))
    hp[:, kk // 2:kk // 2 + T2] = h
    for g in range(G):
        for t in range(T2):
            win = hp[:, t:t + kk, g * cg:(g + 1) * cg]
            ref[:, t, g * cg:(g + 1) * cg] = np.einsum("bkc,okc->bo", win, wg[g].astype(np.float64))
    assert maxabs(y.cpu().numpy(), ref) < 4e-6


def test_split_layernorm_and_conv0_outputs():
    o = ops()
    from oracle import nn as onn
    x = synth.normalish("sln/x", (70, 768)) * 3
    r = synth.normalish("sln/r", (70, 768))
    g = synth.uniform("sln/g", (768,), 0.5, 1.5)
    b = synth.uniform("sln/b", (768,), -0.5, 0.5)
    want = onn.layer_norm((x + r).astype(np.float32), g, b)
    y32, ys = o.layernorm(dev(x), dev(g), dev(b), residual=dev(r), split="both")
    plain = o.layernorm(dev(x), dev(g), dev(b), residual=dev(r))
    assert torch.equal(y32, plain)
    assert maxabs(ys.float().cpu().numpy(), want) < 3e-6
    only = o.layernorm(dev(x), dev(g), dev(b), residual=dev(r), split="only")
    assert torch.equal(only.t, ys.t)
    # conv0 + GroupNorm + GELU written in split storage == the fp32 kernel's output up to the split rounding (2^-22
    # relative) and the GELU's erf (split / 16-bit outputs use the A&S 7.1.26 form, |abs err| <= 1.5e-7 on erf, i.e.
    # <= 0.5 |x| 1.5e-7 on the output; the fp32 kernel calls libm's erff)
    audio = dev(synth.audio_clips(2, 6400, tag="sconv0"))
    w0 = dev(synth.uniform("sconv0/w", (512, 10), -0.5, 0.5))
    gg, bb = dev(synth.uniform("sconv0/g", (512,), 0.5, 1.5)), dev(synth.uniform("sconv0/b", (512,), -0.2, 0.2))
    f = o.conv0_gn_gelu(audio, w0, gg, bb, 20, 0, torch.float32)
    s = o.conv0_gn_gelu(audio, w0, gg, bb, 20, 0, o.SPLIT)
    assert isinstance(s, o.Split) and s.shape == f.shape
    assert torch.all((s.float() - f).abs() <= f.abs() * 2.0 ** -21 + 1.0e-6)   # 0.5 |x| 1.5e-7 with |x| up to ~6 here


@pytest.mark.parametrize("B,H,Tq,Tk,masked", [(2, 12, 200, 200, False), (3, 8, 111, 111, False), (3, 8, 111, 110, True),
                                              (2, 8, 1, 110, False), (1, 8, 50, 300, False)])
def test_split_attention_matches_float64(B, H, Tq, Tk, masked):
    """softmax(scale Q K^T (masked)) V on split Q / K / V (views of one packed split QKV tensor when Tq == Tk) vs a
    float64 reference: 3e-6 on O(1) outputs, fp32 and split outputs."""
    o = ops()
    d = H * 64
    q = synth.normalish(f"sattn/q/{B}{H}{Tq}", (B, Tq, d))
    k = synth.normalish(f"sattn/k/{B}{H}{Tk}", (B, Tk, d))
    v = synth.normalish(f"sattn/v/{B}{H}{Tk}", (B, Tk, d))
    mask = None
    if masked:
        mask = np.ones((Tq, Tk), bool)
        mask[0] = False
        for t in range(1, Tq):
            mask[t, max(0, t - 2):min(Tk, t + 1)] = False
    scale = 64 ** -0.5
    qh = q.reshape(B, Tq, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    kh = k.reshape(B, Tk, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    vh = v.reshape(B, Tk, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    sc = np.einsum("bhqd,bhkd->bhqk", qh, kh) * scale
    if mask is not None:
        sc = np.where(mask[None, None], -np.inf, sc)
    sc = sc - sc.max(-1, keepdims=True)
    pr = np.exp(sc)
    pr /= pr.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bhkd->bhqd", pr, vh).transpose(0, 2, 1, 3).reshape(B, Tq, d)
    m = dev(mask.astype(np.uint8)) if mask is not None else None
    if Tq == Tk:
        qkv = o.to_split(dev(np.concatenate([q, k, v], axis=-1)))
        qs, ks, vs = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    else:
        qs, ks, vs = o.to_split(dev(q)), o.to_split(dev(k)), o.to_split(dev(v))
    out = o.attention(qs, ks, vs, H, scale, mask=m)
    assert isinstance(out, o.Split)
    assert maxabs(out.float().cpu().numpy(), ref) < 3e-6
    out32 = o.attention(qs, ks, vs, H, scale, mask=m, out_dtype=torch.float32)
    assert maxabs(out32.cpu().numpy(), ref) < 3e-6


_MODELS = {}


def get_model(audio_model="wav2vec2", dtype="f16x2", **kw):
    from msmd_amd.model import get_diffusion_model
    key = (audio_model, dtype, tuple(sorted(kw.items())))
    if key not in _MODELS:
        _MODELS.clear()
        args = default_args(audio_model=audio_model, compute_dtype=dtype, **kw)
        _MODELS[key] = (get_diffusion_model(args, DEV).eval(), args)
    return _MODELS[key]


@pytest.mark.parametrize("am", ["wav2vec2", "hubert"])
def test_extract_audio_feature_split(am):
    """Full-depth encoder in the split mode vs the reference's golden: the fp32 mode's own tolerance (1e-4)."""
    g = load_golden(f"g3_audio_{am}")
    model, args = get_model(am)
    assert model.split_mode and model.compute_dtype == torch.float32
    audio = dev(synth.audio_clips(2, 64000))
    feat = model.extract_audio_feature(audio)
    f768 = model.extract_audio_768_feature(audio)
    torch.cuda.synchronize()
    e768, ef = maxabs(f768.cpu().numpy()[:, ::2, ::3], g["feat768"]), maxabs(feat.cpu().numpy(), g["feat"])
    print(f"split mode {am}: feat768 err {e768:.2e}, feat err {ef:.2e}")
    assert e768 < 1e-4 and ef < 1e-4
    from msmd_amd.utils.model_common import pad_audio
    a2 = dev(synth.audio_clips(1, 32000, tag="audio30"))
    y = model.audio_encoder(pad_audio(a2), 30, frame_num=60).last_hidden_state
    assert maxabs(y.cpu().numpy(), g["hidden_fps30_60"]) < 1e-4


def test_denoiser_and_forward_split():
    g = load_golden("g3_denoiser")
    for width in (1, 2):
        model, args = get_model("wav2vec2", align_mask_width=width)
        x = denoiser_inputs(2, args)
        person = torch.cat([dev(x["shape"])[:, None], dev(x["style"])[:, None]], dim=-1)
        y = model.denoising_net(dev(x["motion"]), dev(x["audio_feat"]), person, dev(x["style"])[:, None],
                                dev(x["prev_motion"]), dev(x["prev_audio"]), dev(g["step"]), dev(x["indicator"]))
        err = maxabs(y.cpu().numpy(), g[f"target_w{width}"])
        print(f"split denoiser width {width}: err {err:.2e}")
        assert y.dtype == torch.float32 and err < 1e-4, width
    # MSMD.forward from raw audio (12 + 8 layers) and the feature-input / previous-window / CFG-masked call
    g = load_golden("g3_forward")
    model, args = get_model("wav2vec2")
    x = denoiser_inputs(2, args, tag="fw")
    audio = dev(synth.audio_clips(2, 64000, tag="fw_audio"))
    eps, target, m_det, afeat = model(dev(x["motion"]), audio, dev(x["shape"]), dev(x["style"]), time_step=[3, 499],
                                      indicator=dev(x["indicator"]), train_with_CFG=False, eps=dev(g["a_eps"]))
    e1, e2 = maxabs(afeat.cpu().numpy()[:, ::2, ::3], g["a_audio_feat"]), maxabs(target.cpu().numpy(), g["a_target"])
    print(f"split MSMD.forward (raw audio, 12+8 layers): audio_feat err {e1:.2e}, target err {e2:.2e}")
    assert e1 < 1e-4 and e2 < 1e-4
    flag = dev(g["b_flag"])
    with mock.patch("torch.rand", return_value=flag):
        _, target, _, _ = model(dev(x["motion"]), dev(x["audio_feat"]), dev(x["shape"]), dev(x["style"]),
                                dev(x["prev_motion"]), dev(x["prev_audio"]), time_step=[250, 1],
                                indicator=dev(x["indicator"]), train_with_CFG=True, eps=dev(g["b_eps"]))
    assert maxabs(target.cpu().numpy(), g["b_target"]) < 1e-4


def test_style_encoder_split():
    from msmd_amd.style_encoder import get_style_encoder
    g = load_golden("g3_style")
    enc = get_style_encoder(default_args(compute_dtype="f16x2"), "vae2").to(DEV).eval()
    assert enc.split_mode
    for B, T in ((2, 100), (1, 60)):
        m = dev(synth.motion_clips(B, T, tag="style_in"))
        mu, logvar = enc.mu_logvar(m)
        assert maxabs(mu.cpu().numpy(), g[f"mu_{B}_{T}"]) < 5e-5 and maxabs(logvar.cpu().numpy(), g[f"logvar_{B}_{T}"]) < 5e-5


def test_sampler_split():
    """sample(): hoisted K / V, the diagonal cross-attention fast path, CFG + DDPM, eager loop with injected noise and
    the hipGraph loop -- in the split mode against the reference's goldens (1e-4, as the fp32 mode)."""
    from msmd_amd.model import DiffusionSchedule
    g = load_golden("g3_sample")
    model, args = get_model("wav2vec2")
    x = denoiser_inputs(2, args, tag="sm")
    T = 3
    old = model.diffusion_sched
    model.diffusion_sched = DiffusionSchedule(T, "cosine").to(DEV)
    xT = dev(synth.normalish("sm/xT", (2, 100, 67)))
    cases = {
        "inc": dict(cfg_mode="incremental", cfg_scale=1.15),
        "ind": dict(cfg_mode="independent", cfg_scale=[1.3, 0.9]),
        "dt": dict(cfg_mode="incremental", cfg_scale=1.4, dynamic_threshold=(0.9, 0.5, 2.0)),
    }
    try:
        for name, kw in cases.items():
            z = g[f"{name}_z"]
            noise = {T - i: dev(z[i]) for i in range(T - 1)}
            y, _, _ = model.sample(dev(x["audio_feat"]), dev(x["shape"]), dev(x["style"]), dev(x["prev_motion"]),
                                   dev(x["prev_audio"]), motion_at_T=xT, indicator=dev(x["indicator"]), noise=noise, **kw)
            err = maxabs(y.cpu().numpy(), g[f"{name}_x0"])
            print(f"split sample {name}: err {err:.2e}")
            assert err < 1e-4, (name, err)
        # hipGraph loop runs and is finite (its noise is drawn on the device)
        out, _, _ = model.sample(dev(x["audio_feat"]), dev(x["shape"]), dev(x["style"]), indicator=dev(x["indicator"]))
        assert out.shape == (2, 100, 67) and bool(torch.isfinite(out).all())
    finally:
        model.diffusion_sched = old


def test_bench_batch_b32_matches_cpu_restatement_in_parity_modes():
    """The bench workload itself (configs[1]: B = 32 x 4 s clips, 12 + 8 layers), not only 2-clip goldens: MSMD.forward
    in the f16x2 and fp32 modes against the torch-CPU restatement of the reference (oracle/torch_cpu.py, pinned to the
    reference goldens) on 3 sampled clips of the batch -- every op is row-independent, so the oracle runs those 3 rows
    only -- plus the hipGraph replay of the full batch equal to the eager forward."""
    import bench
    from oracle import diffusion as od, torch_cpu as tc
    from msmd_amd import shapes
    model, args = get_model("wav2vec2")
    b = bench.synth_batch(32, 0, DEV)
    rows = [0, 13, 31]
    sd = tc.to_torch(synth.fill_state_dict(shapes.msmd_shapes(args)))
    sched = od.diffusion_schedule(500, "cosine")
    cpu = lambda t: t[rows].float().cpu()
    _, ref, _ = tc.msmd_forward(sd, sched, cpu(b["motion"]), cpu(b["audio"]), cpu(b["shape"]), cpu(b["style"]),
                                [b["time_step"][r] for r in rows], cpu(b["eps"]), cpu(b["indicator"]))
    ref = ref.numpy()
    for mode in ("f16x2", "fp32"):
        model.set_compute_dtype(mode)
        _, target, _, _ = bench.step(model, b)
        err = maxabs(target[rows].cpu().numpy(), ref)
        print(f"B=32 bench batch, {mode}: max-abs-err vs CPU restatement on rows {rows} = {err:.2e}")
        assert err < 1e-4, (mode, err)
        if mode == "f16x2":
            run = bench.graphed_step(model, b)     # capture_forward verifies replay == eager on perturbed inputs
            out = run()
            torch.cuda.synchronize()
            assert torch.equal(out[1], target)
    model.set_compute_dtype("f16x2")
