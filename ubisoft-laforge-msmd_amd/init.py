"""Parameter initialisation of freshly constructed modules -- the reference's, under the caller's torch RNG.

The reference never writes an init routine: its parameters get whatever ``torch.nn`` gives the modules it instantiates
(reference model.py:116,120,135,138 ``torch.randn`` start / null tokens; model.py:856-908 ``nn.Linear`` /
``nn.TransformerDecoderLayer`` / ``nn.TransformerDecoder``; style_encoder.py:133-176 ``nn.Conv1d`` / ``nn.LayerNorm`` /
``nn.TransformerEncoderLayer``).  The product's modules hold flat parameter trees (utils/model_common.ParamTree) with the
reference's ``state_dict`` keys and no ``torch.nn`` layers, so this file instantiates the SAME ``torch.nn`` layers the
reference does, in the SAME order, as throw-away "donors" on the CPU and copies their freshly drawn values in.  That makes
the distributions the reference's by construction (``nn.Linear``: Kaiming-uniform(a=sqrt 5) = U(+-1/sqrt fan_in) weights and
biases; ``nn.MultiheadAttention``: Xavier-uniform in-projection, zero biases, Linear-default out-projection weight;
``nn.LayerNorm`` 1 / 0; ``nn.Conv1d`` defaults), it consumes the CPU generator exactly as the reference's constructor does --
so ``torch.manual_seed(s)`` before ``get_diffusion_model`` / ``get_style_encoder`` gives, tensor for tensor, the values the
reference's constructor draws from the same generator state (tests/golden/g10_init.npz, recorded from the reference) -- and it
reproduces ``nn.TransformerDecoder``'s deep-copied layers (all ``n_layers`` start from ONE draw: torch ``_get_clones``).

The closed-form synthetic fill (msmd_amd.synth) is used ONLY when asked for by name: ``args.audio_encoder_weights ==
"synthetic"``, or ``MSMD_SYNTHETIC_WEIGHTS=1`` with no checkpoint named (tests, bench.py, smoke()).  A pretrained audio encoder
is never touched after ``from_pretrained`` returned it.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn


def synthetic_requested(args) -> bool:
    """True when the caller asked for the closed-form synthetic weights by name (never implied)."""
    src = getattr(args, "audio_encoder_weights", None)
    if src == "synthetic":
        return True
    return src is None and os.environ.get("MSMD_SYNTHETIC_WEIGHTS", "0") not in ("", "0")


@torch.no_grad()
def _adopt(module: nn.Module, prefix: str, donor: nn.Module) -> None:
    """Copy every parameter of ``donor`` onto ``module``'s parameter called ``prefix + <donor key>``."""
    own = dict(module.named_parameters())
    for key, val in donor.named_parameters():
        dst = own[prefix + key]
        if dst.shape != val.shape:
            raise ValueError(f"{prefix + key}: {tuple(dst.shape)} vs torch.nn donor {tuple(val.shape)}")
        dst.copy_(val)


@torch.no_grad()
def _randn(param: nn.Parameter) -> None:
    param.copy_(torch.randn(param.shape))   # CPU generator, as `nn.Parameter(torch.randn(...))` in the reference


def msmd_front_(model, args) -> None:
    """reference model.py:115-120, the draws BEFORE the denoiser is constructed: audio_feature_map, start tokens."""
    _adopt(model, "audio_feature_map.", nn.Linear(model.audio_encoder.config.hidden_size, args.feature_dim))
    _randn(model.start_audio_feat)
    _randn(model.start_motion_feat)


def msmd_back_(model, args) -> None:
    """reference model.py:131-138, the draws AFTER the denoiser: the classifier-free-guidance null tokens."""
    if "style" in model.guiding_conditions:
        _randn(model.null_style_feat)
    if "audio" in model.guiding_conditions:
        _randn(model.null_audio_feat)


def denoiser_(net, args) -> None:
    """reference model.py:856-908 in construction order."""
    d = net.feature_dim
    _adopt(net, "diff_step_map.", nn.Sequential(nn.Linear(d, d), nn.GELU(), nn.Linear(d, d)))
    if net.use_learnable_pe:
        _randn(net.PE)
    _adopt(net, "person_proj.", nn.Linear(net.person_feat_dim, d))
    _adopt(net, "feature_proj.", nn.Linear(net.motion_feat_dim + (1 if net.use_indicator else 0), d))
    layer = nn.TransformerDecoderLayer(d_model=d, nhead=net.n_heads, dim_feedforward=net.mlp_ratio * d,
                                       activation="gelu", batch_first=True)
    for n in range(net.n_layers):               # nn.TransformerDecoder deep-copies the ONE layer it is given
        _adopt(net, f"transformer.layers.{n}.", layer)
    for b in range(net.num_of_basis):
        _adopt(net, f"static_feature_mapping.{b}.",
               nn.Sequential(nn.Linear(args.d_style, d), nn.GELU(), nn.Linear(d, net.motion_feat_dim)))
    _adopt(net, "motion_dec.", nn.Sequential(nn.Linear(d, d // 2), nn.GELU(),
                                             nn.Linear(d // 2, net.motion_feat_dim + net.num_of_basis)))


def style_encoder_(enc, args) -> None:
    """reference style_encoder.py:133-176 in construction order; the keys are the Sequential positions of the reference
    (permutes / dropouts / ELUs hold no parameters)."""
    c, cin, out = enc.conv_feature_dim, enc.motion_coef_dim, enc.output_size
    _adopt(enc, "input_layers.1.", nn.Conv1d(cin, c, kernel_size=3, padding=1))
    _adopt(enc, "input_layers.5.", nn.LayerNorm(c))
    _adopt(enc, "input_layers.7.", nn.Conv1d(c, c, kernel_size=3, padding=1))
    _adopt(enc, "input_layers.11.", nn.LayerNorm(c))
    _adopt(enc, "encoder.", nn.TransformerEncoderLayer(d_model=c, nhead=8, dim_feedforward=c, activation="gelu",
                                                       batch_first=True))
    _adopt(enc, "output_layers.1.", nn.Conv1d(c, out, kernel_size=3, padding=1))
    _adopt(enc, "output_layers.5.", nn.LayerNorm(out))
    _adopt(enc, "output_layers.7.", nn.Conv1d(out, out, kernel_size=3, padding=1))
