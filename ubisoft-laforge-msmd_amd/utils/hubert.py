"""HuBERT-base audio encoder (drop-in surface of reference utils/hubert.py:9-51).

hubert-base-ls960 shares wav2vec2-base's architecture (group-norm conv stack, post-LN encoder,
same state_dict key names), so it reuses the same HIP forward."""
from __future__ import annotations

from .wav2vec2 import Wav2Vec2Model, linear_interpolation  # noqa: F401


class HubertModel(Wav2Vec2Model):
    model_type = "hubert"
