"""Shared test helpers: synthetic state dicts and inputs (same recipes as tests/golden/make_goldens.py)."""
import functools

import numpy as np

from msmd_amd import shapes, synth
from msmd_amd.config import default_args


@functools.lru_cache(maxsize=4)
def msmd_state_dict(audio_model="wav2vec2", n_encoder_layers=12, **kw):
    args = default_args(audio_model=audio_model, **kw)
    return synth.fill_state_dict(shapes.msmd_shapes(args, n_encoder_layers)), args


@functools.lru_cache(maxsize=2)
def style_state_dict():
    args = default_args()
    return synth.fill_state_dict(shapes.style_encoder_shapes(args)), args


def denoiser_inputs(B, args, tag="dn"):
    d = args.feature_dim
    return dict(
        motion=synth.normalish(f"{tag}/motion", (B, 100, 67)),
        audio_feat=synth.normalish(f"{tag}/audio_feat", (B, 100, d)),
        shape=(0.3 * synth.normalish(f"{tag}/shape", (B, 100))).astype(np.float32),
        style=synth.normalish(f"{tag}/style", (B, args.d_style)),
        prev_motion=synth.normalish(f"{tag}/prev_motion", (B, 10, 67)),
        prev_audio=synth.normalish(f"{tag}/prev_audio", (B, 10, d)),
        indicator=np.concatenate([np.ones((B, 80), np.float32), np.zeros((B, 20), np.float32)], axis=1),
    )


def flame_inputs(B, tag="flame"):
    return dict(shape=(0.5 * synth.normalish(f"{tag}/shape", (B, 100))).astype(np.float32),
                exp=(0.5 * synth.normalish(f"{tag}/exp", (B, 50))).astype(np.float32),
                pose=(0.4 * synth.normalish(f"{tag}/pose", (B, 6))).astype(np.float32))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@functools.lru_cache(maxsize=1)
def hubert_large_state_dict(n_layers=2):
    """Encoder-only synthetic state dict of the HuBERT-large ARCHITECTURE (keys prefixed audio_encoder.)."""
    shp = shapes.audio_encoder_shapes(n_layers, 1024, 4096, feat_extract_norm="layer", conv_bias=True)
    return synth.fill_state_dict({"audio_encoder." + k: v for k, v in shp.items()})
