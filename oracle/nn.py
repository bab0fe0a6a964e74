"""Numpy fp32 primitives with torch / HF semantics (oracle; test infrastructure).

Semantics cheat-sheet: SURVEY.md Appendix A.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32


def f32(x):
    return np.asarray(x, dtype=np.float32)


def linear(x, w, b=None):
    """torch.nn.functional.linear: y = x @ w.T + b, w is (out, in)."""
    x = f32(x)
    w = f32(w)
    # 2-D GEMM only: numpy's N-D @ 2-D broadcast path bypasses BLAS (100x slower)
    y = (np.ascontiguousarray(x).reshape(-1, x.shape[-1]) @ np.ascontiguousarray(w.T)).reshape(
        x.shape[:-1] + (w.shape[0],))
    if b is not None:
        y = y + f32(b)
    return y.astype(F32)


def gelu(x):
    """Exact erf GELU (HF ACT2FN['gelu'], torch F.gelu approximate='none')."""
    x = f32(x)
    return (F32(0.5) * x * (F32(1.0) + _erf(x * F32(1.0 / math.sqrt(2.0))).astype(F32))).astype(F32)


def elu(x):
    """nn.ELU(alpha=1) (reference style_encoder.py:140,148,168)."""
    x = f32(x)
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(F32)


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim, biased variance."""
    x = f32(x)
    mean = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=np.float32)
    y = xc / np.sqrt(var + F32(eps))
    return (y * f32(w) + f32(b)).astype(F32)


def softmax(x, axis=-1):
    x = f32(x)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True, dtype=np.float32)).astype(F32)


def conv1d_cl(x, w, b=None, stride=1, padding=0):
    """nn.Conv1d on a channels-last activation.

    x: (B, T, Cin); w: torch layout (Cout, Cin, k); returns (B, T_out, Cout).
    Implemented as a window view + one matmul (K index = kk * Cin + cin).
    """
    x = f32(x)
    if padding:
        x = np.pad(x, ((0, 0), (padding, padding), (0, 0)))
    B, T, C = x.shape
    Cout, Cin, k = w.shape
    assert Cin == C
    T_out = (T - k) // stride + 1
    x = np.ascontiguousarray(x)
    sB, sT, sC = x.strides
    win = np.lib.stride_tricks.as_strided(x, shape=(B, T_out, k, C), strides=(sB, sT * stride, sT, sC),
                                          writeable=False)
    wm = np.ascontiguousarray(f32(w).transpose(2, 1, 0).reshape(k * Cin, Cout))
    a = np.empty((B, T_out, k, C), dtype=np.float32)  # explicit im2col copy: BLAS needs a dense 2-D operand
    a[...] = win
    y = (a.reshape(B * T_out, k * C) @ wm).reshape(B, T_out, Cout)
    if b is not None:
        y = y + f32(b)
    return y.astype(F32)


def interp_linear_table(in_len: int, out_len: int):
    """Index/weight table of F.interpolate(mode='linear', align_corners=False).

    Returns (i0, i1, w1) with out[j] = (1-w1[j]) * x[i0[j]] + w1[j] * x[i1[j]].
    ATen's area_pixel_compute_source_index: src = scale*(dst+0.5)-0.5 clamped
    at 0, computed in fp32 (reference call sites utils/wav2vec2.py:57-63 and
    model.py:260).
    """
    scale = np.float32(in_len) / np.float32(out_len)
    dst = np.arange(out_len, dtype=np.float32)
    # ATen contracts scale*(dst+0.5)-0.5 into ONE fused multiply-add (single rounding) on the
    # CPU build that generated the goldens and under nvcc's default -fmad; emulate fmaf through
    # fp64 (the fp32 x fp32 product is exact there).
    src = (scale.astype(np.float64) * (dst + np.float32(0.5)).astype(np.float64) - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0.0)).astype(np.float32)
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, in_len - 1)
    i1 = np.minimum(i0 + 1, in_len - 1)
    w1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, w1


def interp_linear_cl(x, out_len: int):
    """Linear resample along time of a channels-last (B, T, C) tensor."""
    x = f32(x)
    i0, i1, w1 = interp_linear_table(x.shape[1], out_len)
    w1 = w1[None, :, None]
    return ((F32(1.0) - w1) * x[:, i0] + w1 * x[:, i1]).astype(F32)


def mha(query, key, value, in_w, in_b, out_w, out_b, n_heads, mask=None):
    """torch.nn.MultiheadAttention (batch_first, eval) forward.

    in_w is the packed (3d, d) in_proj_weight with rows ordered q, k, v;
    mask is bool (Tq, Tk) with True = masked out (-inf).
    """
    d = query.shape[-1]
    hd = d // n_heads
    q = linear(query, in_w[:d], in_b[:d])
    k = linear(key, in_w[d:2 * d], in_b[d:2 * d])
    v = linear(value, in_w[2 * d:], in_b[2 * d:])
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    q = q.reshape(B, Tq, n_heads, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, Tk, n_heads, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, Tk, n_heads, hd).transpose(0, 2, 1, 3)
    s = np.matmul(q, k.transpose(0, 1, 3, 2)) * F32(1.0 / math.sqrt(hd))
    if mask is not None:
        s = np.where(mask[None, None], F32(-np.inf), s)
    p = softmax(s, axis=-1)
    o = np.matmul(p, v).transpose(0, 2, 1, 3).reshape(B, Tq, d)
    return linear(o, out_w, out_b)


def sinusoid_table(max_len: int, d_model: int):
    """PositionalEncoding buffer (reference utils/model_common.py:90-97)."""
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    position = np.arange(max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * F32(-math.log(10000.0) / d_model)).astype(F32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe[None]
