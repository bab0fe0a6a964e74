"""HuBERT-base audio encoder (drop-in surface of reference utils/hubert.py:9-51).

hubert-base-ls960 shares wav2vec2-base's architecture (group-norm conv stack, post-LN encoder,
same state_dict key names), so it reuses the same HIP forward."""
from __future__ import annotations

from .wav2vec2 import Wav2Vec2Model, linear_interpolation  # noqa: F401


# facebook/hubert-large-ls960-ft config.json (BASELINE.json configs[3]: "HuBERT-large encoder swap")
LARGE_CONFIG = dict(num_hidden_layers=24, hidden_size=1024, intermediate_size=4096, num_attention_heads=16,
                    feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)


class HubertModel(Wav2Vec2Model):
    model_type = "hubert"

    @classmethod
    def large(cls, **overrides):
        """HuBERT-large architecture (LayerNorm conv stack with biases, stable-layer-norm encoder, 1024 wide)."""
        return cls(dict(LARGE_CONFIG, **overrides))
