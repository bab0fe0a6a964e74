"""state_dict key names and shapes of the reference's modules.

The product modules create their parameters from these tables so that reference
checkpoints ({'args','model','style_enc','iter'}, reference training_script.py:227-233)
load without renaming; tests use the same tables to build synthetic weights.
"""
from __future__ import annotations

from collections import OrderedDict

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)


def audio_encoder_shapes(n_layers: int = 12, hidden: int = 768, ffn: int = 3072, conv_dim: int = 512,
                         pos_k: int = 128, pos_groups: int = 16, feat_extract_norm: str = "group",
                         conv_bias: bool = False) -> "OrderedDict[str, tuple]":
    """HF Wav2Vec2Model / HubertModel.  Base checkpoints: feat_extract_norm='group' (GroupNorm on conv layer 0 only,
    no conv bias), post-LN encoder.  Large checkpoints (hubert-large-ls960-ft, wav2vec2-large): 'layer' (LayerNorm
    after EVERY conv layer, conv biases) and the stable-layer-norm (pre-LN) encoder, which uses the same key names."""
    s = OrderedDict()
    s["masked_spec_embed"] = (hidden,)
    for i, k in enumerate(CONV_KERNEL):
        cin = 1 if i == 0 else conv_dim
        s[f"feature_extractor.conv_layers.{i}.conv.weight"] = (conv_dim, cin, k)
        if conv_bias:
            s[f"feature_extractor.conv_layers.{i}.conv.bias"] = (conv_dim,)
        if i == 0 or feat_extract_norm == "layer":
            s[f"feature_extractor.conv_layers.{i}.layer_norm.weight"] = (conv_dim,)
            s[f"feature_extractor.conv_layers.{i}.layer_norm.bias"] = (conv_dim,)
    s["feature_projection.layer_norm.weight"] = (conv_dim,)
    s["feature_projection.layer_norm.bias"] = (conv_dim,)
    s["feature_projection.projection.weight"] = (hidden, conv_dim)
    s["feature_projection.projection.bias"] = (hidden,)
    s["encoder.pos_conv_embed.conv.bias"] = (hidden,)
    s["encoder.pos_conv_embed.conv.weight_g"] = (1, 1, pos_k)
    s["encoder.pos_conv_embed.conv.weight_v"] = (hidden, hidden // pos_groups, pos_k)
    s["encoder.layer_norm.weight"] = (hidden,)
    s["encoder.layer_norm.bias"] = (hidden,)
    for n in range(n_layers):
        p = f"encoder.layers.{n}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"attention.{nm}.weight"] = (hidden, hidden)
            s[p + f"attention.{nm}.bias"] = (hidden,)
        s[p + "layer_norm.weight"] = (hidden,)
        s[p + "layer_norm.bias"] = (hidden,)
        s[p + "feed_forward.intermediate_dense.weight"] = (ffn, hidden)
        s[p + "feed_forward.intermediate_dense.bias"] = (ffn,)
        s[p + "feed_forward.output_dense.weight"] = (hidden, ffn)
        s[p + "feed_forward.output_dense.bias"] = (hidden,)
        s[p + "final_layer_norm.weight"] = (hidden,)
        s[p + "final_layer_norm.bias"] = (hidden,)
    return s


def denoiser_shapes(args, motion_dim: int = 67) -> "OrderedDict[str, tuple]":
    """DenoisingNetwork_MSMD (reference model.py:820-908)."""
    d = args.feature_dim
    nb = int(args.num_of_basis)
    L = 1 + args.n_prev_motions + args.n_motions
    person_dim = 100 + args.d_style
    s = OrderedDict()
    if not getattr(args, "no_use_learnable_pe", False):
        s["PE"] = (1, L, d)   # learnable PE; the sinusoidal variant holds a computed buffer instead (model.py:862-866)
    for i in (0, 2):
        s[f"diff_step_map.{i}.weight"] = (d, d)
        s[f"diff_step_map.{i}.bias"] = (d,)
    s["person_proj.weight"] = (d, person_dim)
    s["person_proj.bias"] = (d,)
    s["feature_proj.weight"] = (d, motion_dim + (1 if args.use_indicator else 0))
    s["feature_proj.bias"] = (d,)
    ff = args.mlp_ratio * d
    for n in range(args.n_layers):
        p = f"transformer.layers.{n}."
        for att in ("self_attn", "multihead_attn"):
            s[p + f"{att}.in_proj_weight"] = (3 * d, d)
            s[p + f"{att}.in_proj_bias"] = (3 * d,)
            s[p + f"{att}.out_proj.weight"] = (d, d)
            s[p + f"{att}.out_proj.bias"] = (d,)
        s[p + "linear1.weight"] = (ff, d)
        s[p + "linear1.bias"] = (ff,)
        s[p + "linear2.weight"] = (d, ff)
        s[p + "linear2.bias"] = (d,)
        for k in (1, 2, 3):
            s[p + f"norm{k}.weight"] = (d,)
            s[p + f"norm{k}.bias"] = (d,)
    for b in range(nb):
        s[f"static_feature_mapping.{b}.0.weight"] = (d, args.d_style)
        s[f"static_feature_mapping.{b}.0.bias"] = (d,)
        s[f"static_feature_mapping.{b}.2.weight"] = (motion_dim, d)
        s[f"static_feature_mapping.{b}.2.bias"] = (motion_dim,)
    s["motion_dec.0.weight"] = (d // 2, d)
    s["motion_dec.0.bias"] = (d // 2,)
    s["motion_dec.2.weight"] = (motion_dim + nb, d // 2)
    s["motion_dec.2.bias"] = (motion_dim + nb,)
    return s


def msmd_shapes(args, n_encoder_layers: int = 12, motion_dim: int = 67) -> "OrderedDict[str, tuple]":
    """MSMD learnable parameters (reference model.py:73-140), registration order."""
    s = OrderedDict()
    for k, v in audio_encoder_shapes(n_encoder_layers).items():
        s["audio_encoder." + k] = v
    s["audio_feature_map.weight"] = (args.feature_dim, 768)
    s["audio_feature_map.bias"] = (args.feature_dim,)
    s["start_audio_feat"] = (1, args.n_prev_motions, args.feature_dim)
    s["start_motion_feat"] = (1, args.n_prev_motions, motion_dim)
    for k, v in denoiser_shapes(args, motion_dim).items():
        s["denoising_net." + k] = v
    conds = [c for c in (args.guiding_conditions.split(",") if args.guiding_conditions else [])
             if c in ("style", "audio")]
    if "style" in conds:
        s["null_style_feat"] = (1, 1, args.d_style)
    if "audio" in conds:
        s["null_audio_feat"] = (1, 1, args.feature_dim)
    return s


def style_encoder_shapes(args, input_dim: int = 67, conv_dim: int = 512) -> "OrderedDict[str, tuple]":
    """StyleEncoder_VAE2 (reference style_encoder.py:119-176)."""
    out = args.d_style * 2
    s = OrderedDict()
    s["input_layers.1.weight"] = (conv_dim, input_dim, 3)
    s["input_layers.1.bias"] = (conv_dim,)
    s["input_layers.5.weight"] = (conv_dim,)
    s["input_layers.5.bias"] = (conv_dim,)
    s["input_layers.7.weight"] = (conv_dim, conv_dim, 3)
    s["input_layers.7.bias"] = (conv_dim,)
    s["input_layers.11.weight"] = (conv_dim,)
    s["input_layers.11.bias"] = (conv_dim,)
    s["encoder.self_attn.in_proj_weight"] = (3 * conv_dim, conv_dim)
    s["encoder.self_attn.in_proj_bias"] = (3 * conv_dim,)
    s["encoder.self_attn.out_proj.weight"] = (conv_dim, conv_dim)
    s["encoder.self_attn.out_proj.bias"] = (conv_dim,)
    s["encoder.linear1.weight"] = (conv_dim, conv_dim)
    s["encoder.linear1.bias"] = (conv_dim,)
    s["encoder.linear2.weight"] = (conv_dim, conv_dim)
    s["encoder.linear2.bias"] = (conv_dim,)
    s["encoder.norm1.weight"] = (conv_dim,)
    s["encoder.norm1.bias"] = (conv_dim,)
    s["encoder.norm2.weight"] = (conv_dim,)
    s["encoder.norm2.bias"] = (conv_dim,)
    s["output_layers.1.weight"] = (out, conv_dim, 3)
    s["output_layers.1.bias"] = (out,)
    s["output_layers.5.weight"] = (out,)
    s["output_layers.5.bias"] = (out,)
    s["output_layers.7.weight"] = (out, out, 3)
    s["output_layers.7.bias"] = (out,)
    return s
