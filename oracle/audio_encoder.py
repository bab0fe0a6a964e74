"""Oracle: audio padding + wav2vec2 / HuBERT-base encoder (numpy fp32; test infrastructure).

Follows reference utils/model_common.py:110-123 (pad_audio), utils/wav2vec2.py:57-119
and utils/hubert.py:13-51 (wrapper forward), model.py:250-280 (extract_audio_feature).
The HF modules the wrappers call are third-party (transformers==4.44.2, absent
from /root/reference): Wav2Vec2FeatureEncoder, Wav2Vec2FeatureProjection,
Wav2Vec2PositionalConvEmbedding, Wav2Vec2Encoder(Layer) (post-LN variant used by
the *-base checkpoints).  Their published algorithm is restated here; parity is
anchored on goldens generated from the imported reference (tests/golden).

All activations are channels-last (B, T, C).
"""
from __future__ import annotations

import math

import numpy as np

from . import nn

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)


# --------------------------------------------------------------------------- index maths (bit-exact)
def pad_audio_plan(audio_len: int, audio_unit: int = 320, pad_threshold: int = 80):
    """(reflect_len, replicate_len) of reference utils/model_common.py:110-123.
    reflect_len is applied TWICE per side, replicate_len (0/1) once."""
    n_units = audio_len // audio_unit
    side_len = math.ceil((audio_unit * n_units + pad_threshold - audio_len) / 2)
    if side_len >= 0:
        return side_len // 2, side_len % 2
    return 0, 0


def pad_audio_gather_index(audio_len: int, audio_unit: int = 320, pad_threshold: int = 80) -> np.ndarray:
    """Source index of every output sample of pad_audio (a pure gather)."""
    r, rep = pad_audio_plan(audio_len, audio_unit, pad_threshold)
    idx = np.arange(audio_len, dtype=np.int64)
    for _ in range(2):
        if r > 0:
            # F.pad(mode='reflect'): left = x[r..1], right = x[n-2 .. n-1-r]
            n = idx.shape[0]
            idx = np.concatenate([idx[1:r + 1][::-1], idx, idx[n - 1 - r:n - 1][::-1]])
    if rep > 0:
        idx = np.concatenate([idx[:1], idx, idx[-1:]])
    return idx


def pad_audio(audio: np.ndarray) -> np.ndarray:
    """reference utils/model_common.py:110-123 on (B, L) fp32."""
    audio = nn.f32(audio)
    return np.ascontiguousarray(audio[:, pad_audio_gather_index(audio.shape[1])])


def conv_out_lengths(n_samples: int):
    """T chain of the 7-layer feature extractor: T' = (T - k)//s + 1."""
    out = []
    t = n_samples
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        t = (t - k) // s + 1
        out.append(t)
    return out


def crop_len(frame_num: int, output_fps) -> int:
    """round(frame_num * 50 / fps) with Python banker's rounding
    (reference utils/wav2vec2.py:82, utils/hubert.py:25)."""
    return round(frame_num * 50 / output_fps)


# --------------------------------------------------------------------------- modules
def feature_extractor(sd, prefix, audio_padded):
    """HF Wav2Vec2FeatureEncoder / HubertFeatureEncoder (called at reference utils/wav2vec2.py:79 /
    utils/hubert.py:22).  (B, S) -> (B, T, 512).  feat_extract_norm='group' (base checkpoints: GroupNorm on layer 0
    only, no conv bias) or 'layer' (large checkpoints: conv bias + LayerNorm over channels after EVERY conv) is
    read off the state dict."""
    fe = f"{prefix}feature_extractor.conv_layers."
    layer_mode = (fe + "1.layer_norm.weight") in sd
    x = nn.f32(audio_padded)[:, :, None]  # (B, S, 1)
    for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE)):
        w = sd[f"{fe}{i}.conv.weight"]
        x = nn.conv1d_cl(x, w, sd.get(f"{fe}{i}.conv.bias"), stride=s)
        if layer_mode:
            x = nn.layer_norm(x, sd[f"{fe}{i}.layer_norm.weight"], sd[f"{fe}{i}.layer_norm.bias"])
        elif i == 0:
            # GroupNorm(num_groups=512, num_channels=512): per (sample, channel) over time
            gw = sd[f"{fe}0.layer_norm.weight"]
            gb = sd[f"{fe}0.layer_norm.bias"]
            mean = x.mean(axis=1, keepdims=True, dtype=np.float64)
            var = x.var(axis=1, keepdims=True, dtype=np.float64)
            x = ((x - mean) / np.sqrt(var + 1e-5)).astype(np.float32) * nn.f32(gw) + nn.f32(gb)
        x = nn.gelu(x)
    return x


def folded_pos_conv_weight(sd, prefix):
    """weight_norm(dim=2): w = g * v / ||v||_{dims 0,1}; accepts both key spellings
    (SURVEY.md Appendix B)."""
    base = f"{prefix}encoder.pos_conv_embed.conv."
    if base + "weight_g" in sd:
        g, v = sd[base + "weight_g"], sd[base + "weight_v"]
    else:
        g, v = sd[base + "parametrizations.weight.original0"], sd[base + "parametrizations.weight.original1"]
    g = nn.f32(g)
    v = nn.f32(v)
    norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    return (v * (g / norm)).astype(np.float32)


def pos_conv_embed(sd, prefix, h):
    """HF Wav2Vec2PositionalConvEmbedding: grouped Conv1d(768,768,k=128,pad=64,groups=16),
    drop the last frame (even kernel), GELU.  h: (B, T, 768)."""
    w = folded_pos_conv_weight(sd, prefix)  # (768, 48, 128)
    b = nn.f32(sd[f"{prefix}encoder.pos_conv_embed.conv.bias"])
    B, T, C = h.shape
    G = 16
    cg = C // G   # 48 (base) / 64 (large)
    out = np.empty((B, T, C), dtype=np.float32)
    for g in range(G):
        y = nn.conv1d_cl(h[:, :, g * cg:(g + 1) * cg], w[g * cg:(g + 1) * cg], b[g * cg:(g + 1) * cg],
                         stride=1, padding=64)
        out[:, :, g * cg:(g + 1) * cg] = y[:, :T]
    return nn.gelu(out)


def _self_attention(sd, p, h, n_heads):
    d = h.shape[-1]
    hd = d // n_heads
    B, T, _ = h.shape
    q = nn.linear(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]) * np.float32(hd ** -0.5)
    k = nn.linear(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"])
    v = nn.linear(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"])
    q = q.reshape(B, T, n_heads, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, T, n_heads, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, T, n_heads, hd).transpose(0, 2, 1, 3)
    pr = nn.softmax(np.matmul(q, k.transpose(0, 1, 3, 2)), axis=-1)
    a = np.matmul(pr, v).transpose(0, 2, 1, 3).reshape(B, T, d)
    return nn.linear(a, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])


def _feed_forward(sd, p, h):
    f = nn.gelu(nn.linear(h, sd[p + "feed_forward.intermediate_dense.weight"],
                          sd[p + "feed_forward.intermediate_dense.bias"]))
    return nn.linear(f, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])


def encoder_layer(sd, p, h, n_heads=12):
    """HF Wav2Vec2EncoderLayer / HubertEncoderLayer (post-LN)."""
    h = nn.layer_norm(h + _self_attention(sd, p, h, n_heads), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
    return nn.layer_norm(h + _feed_forward(sd, p, h), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"])


def encoder_layer_stable(sd, p, h, n_heads=16):
    """HF HubertEncoderLayerStableLayerNorm (pre-LN; do_stable_layer_norm=True, large checkpoints)."""
    h = h + _self_attention(sd, p, nn.layer_norm(h, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]), n_heads)
    return h + _feed_forward(sd, p, nn.layer_norm(h, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"]))


def num_encoder_layers(sd, prefix):
    n = 0
    while f"{prefix}encoder.layers.{n}.attention.q_proj.weight" in sd:
        n += 1
    return n


def audio_encoder(sd, prefix, audio_padded, output_fps=25, frame_num=None, n_heads=12, return_stages=False,
                  stable_layer_norm=False):
    """Eval-mode forward of the wrapper (reference utils/wav2vec2.py:71-119,
    utils/hubert.py:13-51) -> last_hidden_state (B, frame_num, 768)."""
    stages = {}
    x = feature_extractor(sd, prefix, audio_padded)  # (B, T50, 512)
    stages["conv"] = x
    if frame_num is not None:
        x = x[:, :crop_len(frame_num, output_fps)]
        out_len = frame_num
    else:
        out_len = int(x.shape[1] / 50.0 * output_fps)
    x = nn.interp_linear_cl(x, out_len)
    stages["interp"] = x
    x = nn.layer_norm(x, sd[f"{prefix}feature_projection.layer_norm.weight"],
                      sd[f"{prefix}feature_projection.layer_norm.bias"])
    x = nn.linear(x, sd[f"{prefix}feature_projection.projection.weight"],
                  sd[f"{prefix}feature_projection.projection.bias"])
    stages["proj"] = x
    x = x + pos_conv_embed(sd, prefix, x)
    if stable_layer_norm:   # HubertEncoderStableLayerNorm: pre-LN layers, one LayerNorm at the very end
        stages["posconv"] = x
        for i in range(num_encoder_layers(sd, prefix)):
            x = encoder_layer_stable(sd, f"{prefix}encoder.layers.{i}.", x, n_heads)
            stages[f"layer{i}"] = x
        x = nn.layer_norm(x, sd[f"{prefix}encoder.layer_norm.weight"], sd[f"{prefix}encoder.layer_norm.bias"])
        return (x, stages) if return_stages else x
    x = nn.layer_norm(x, sd[f"{prefix}encoder.layer_norm.weight"], sd[f"{prefix}encoder.layer_norm.bias"])
    stages["posconv_ln"] = x
    for i in range(num_encoder_layers(sd, prefix)):
        x = encoder_layer(sd, f"{prefix}encoder.layers.{i}.", x, n_heads)
        stages[f"layer{i}"] = x
    return (x, stages) if return_stages else x


def extract_audio_768_feature(sd, audio, fps=25, frame_num=100, prefix="audio_encoder."):
    """reference model.py:266-280."""
    h = audio_encoder(sd, prefix, pad_audio(audio), fps, frame_num=frame_num * 2)
    return nn.interp_linear_cl(h, frame_num)


def extract_audio_feature(sd, audio, fps=25, frame_num=100, prefix="audio_encoder."):
    """reference model.py:250-264: encoder at 2L -> linear resample to L -> Linear 768->feature_dim."""
    h = extract_audio_768_feature(sd, audio, fps, frame_num, prefix)
    return nn.linear(h, sd["audio_feature_map.weight"], sd["audio_feature_map.bias"])
