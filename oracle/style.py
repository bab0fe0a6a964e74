"""Oracle: StyleEncoder_VAE2 (numpy fp32; test infrastructure).

Follows reference style_encoder.py:119-213 in eval mode (dropouts off) and
utils/model_common.py:99-101 (PositionalEncoding.forward adds the SINGLE row
pe[0, seq_len] to every position).  nn.TransformerEncoderLayer (post-norm,
GELU, d=512, 8 heads, ff=512) is third-party arithmetic (torch==2.0.0) restated
in encoder_layer().
"""
from __future__ import annotations

import numpy as np

from . import nn


def encoder_layer(sd, p, x, n_heads=8):
    sa = nn.mha(x, x, x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], n_heads)
    x = nn.layer_norm(x + sa, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    ff = nn.linear(nn.gelu(nn.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                   sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return nn.layer_norm(x + ff, sd[p + "norm2.weight"], sd[p + "norm2.bias"])


def style_encoder_mu_logvar(sd, motion_coef, prefix=""):
    """(B, T, 67) -> (mu (B, d_style), logvar (B, d_style)); reference style_encoder.py:178-199."""
    P = prefix
    x = nn.f32(motion_coef)
    T = x.shape[1]
    x = nn.conv1d_cl(x, sd[P + "input_layers.1.weight"], sd[P + "input_layers.1.bias"], padding=1)
    x = nn.layer_norm(nn.elu(x), sd[P + "input_layers.5.weight"], sd[P + "input_layers.5.bias"])
    x = nn.conv1d_cl(x, sd[P + "input_layers.7.weight"], sd[P + "input_layers.7.bias"], padding=1)
    x = nn.layer_norm(nn.elu(x), sd[P + "input_layers.11.weight"], sd[P + "input_layers.11.bias"])
    pe = nn.sinusoid_table(600, x.shape[-1])
    x = x + pe[:, T, :]  # the row-index quirk of utils/model_common.py:100
    x = encoder_layer(sd, P + "encoder.", x)
    x = nn.conv1d_cl(x, sd[P + "output_layers.1.weight"], sd[P + "output_layers.1.bias"], padding=1)
    x = nn.layer_norm(nn.elu(x), sd[P + "output_layers.5.weight"], sd[P + "output_layers.5.bias"])
    x = nn.conv1d_cl(x, sd[P + "output_layers.7.weight"], sd[P + "output_layers.7.bias"], padding=1)
    out = x.mean(axis=1, dtype=np.float32)
    h = out.shape[1] // 2
    return out[:, :h], out[:, h:]


def reparam(mu, logvar, eps):
    """mu + eps * exp(0.5 * logvar) (reference style_encoder.py:201-207)."""
    return (nn.f32(mu) + nn.f32(eps) * np.exp(np.float32(0.5) * nn.f32(logvar))).astype(np.float32)
