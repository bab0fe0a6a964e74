"""Explicit model configuration.

The reference's ``training_script.py`` parser (reference training_script.py:448-515)
never defines a dozen attributes that ``MSMD`` / ``DenoisingNetwork_MSMD`` /
the losses read (reference model.py:78-129, 843-849; utils/common.py:220,245).
SURVEY.md §5.6 records the upstream DiffPoseTalk defaults; this module ships
them as one explicit namespace so that the reference's CLI flag names keep
working and nothing is left undefined.
"""
from __future__ import annotations

import argparse

DEFAULTS = dict(
    # model
    target="sample", architecture="decoder", style_enc_ckpt=None, d_style=256, fps=25,
    n_motions=100, n_prev_motions=10, audio_model="wav2vec2", feature_dim=512,
    n_diff_steps=500, diff_schedule="cosine", cfg_mode="incremental",
    guiding_conditions="audio,style", num_of_basis=4, style_enc_model_style="vae2",
    dataset_type="ravdess+celebv-text-medium", rot_repr="aa", no_head_pose=False,
    use_indicator=True, n_heads=8, n_layers=8, mlp_ratio=4, align_mask_width=1,
    no_use_learnable_pe=False, regularize_alpha="None",
    # losses (reference training_script.py:476-486, utils/common.py)
    criterion="l2", no_constrain_prev=False, l_vert=1.0, l_vel=0.5, l_smooth=10.0,
    l_head_angle=1.0, l_head_vel=0.5, l_head_smooth=0.5, l_head_trans=0.5,
    l_kl_div=1e-7, use_vertex_space=False,
    # training (reference training_script.py:488-513, training_specs.sh)
    lr=2e-5, warm_iter=5000, scheduler="Warmup", cos_max_iter=1_000_000, min_lr_ratio=0.1, batch_size=16, max_iter=2_000_000,
    gradient_accumulation_steps=1, log_iter=100, save_iter=10000, val_iter=10000, log_smooth_win=50, trunc_prob1=0.5, trunc_prob2=0.4,
    prob_cross_style=0.3, use_cross_style=True,
    # engine (new in this build)
    compute_dtype="bf16",       # "bf16" speed mode | "fp32" parity mode
    encoder_layers=None,        # override HF num_hidden_layers (tests use 1-2)
    # pretrained audio encoder (reference model.py:95 / :100 hard-code the hub id and /code/models/Huggingface/hub2):
    audio_encoder_weights=None,  # None: the hub id's local checkpoint, else RAISE | a checkpoint directory | "synthetic"
    hf_cache_dir=None,           # hub-cache root searched first (then HF_HUB_CACHE, HF_HOME, ~/.cache, the reference's)
)


# Stated max-abs-error bounds of each compute mode against the reference's fp32 CPU arithmetic (oracle/torch_cpu.py, pinned to
# the reference goldens), at FULL size: the B = 32 forward of configs[1] (motion coefficients, |x| <= ~5), the 24-layer
# HuBERT-large hidden states (LayerNorm-ed, |x| <= ~6) and the B = 64 sampler.  fp32 / f16x2 meet the north_star's 1e-4.
# The 16-bit storage modes are throughput modes: their error is the operands' own rounding (tools/bf16_error_budget.py: with
# bf16 WEIGHTS ALONE a 12-layer post-LN encoder is already 0.024 off; an fp32 residual stream buys 0.037 -> 0.032), measured
# 0.05-0.088 (bf16) and 0.006-0.011 (fp16) over boxes, batches and seeds (rounds 2-4); the bounds are the worst measurement
# plus a margin of about 1/8 (a last-bit change upstream moves the maximum over 214 k outputs by a few per cent), so a regression
# of the 16-bit arithmetic shows up here instead of hiding under a power of two.
# tests/test_model_gpu.py asserts them on the bench workloads and bench.py exits non-zero when its own batch exceeds them.
PARITY_BOUNDS = {"fp32": 1e-4, "f16x2": 1e-4, "fp16": 2.0 ** -6, "bf16": 0.1}
# The 24-layer HuBERT-large hidden states in bf16 (1 M values per clip, |h| up to 3.7, i.e. a bf16 step of 2^-6 at the top):
# the MAXIMUM error over them is a chaotic statistic -- it read 0.088 in round 5 and 0.125 in round 6, when nothing but the
# association of the LayerNorm row statistics changed (a last-bit change at every layer, made so that a clip's result no longer
# depends on its batch).  Bound: ten bf16 steps at the top of the range.  The motion coefficients' bound above is unchanged.
HUBERT_LARGE_BF16_HIDDEN_BOUND = 10 * 2.0 ** -6


def default_args(**overrides) -> argparse.Namespace:
    cfg = dict(DEFAULTS)
    cfg.update(overrides)
    return argparse.Namespace(**cfg)


def synthetic_args(**overrides) -> argparse.Namespace:
    """default_args with the audio encoder on the closed-form synthetic weights of msmd_amd.synth, said out loud:
    benchmarks, smoke() and the developer tools have no pretrained asset (no network)."""
    overrides.setdefault("audio_encoder_weights", "synthetic")
    return default_args(**overrides)
