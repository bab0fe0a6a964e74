"""Oracle: window plan of inference.infer_coeffs (integer maths; test infrastructure).

Follows reference inference.py:34-75.
"""
from __future__ import annotations

import math

import numpy as np

from . import diffusion
from .audio_encoder import extract_audio_feature


def window_plan(n_samples: int, fps=25, n_motions=100, audio_unit=640.0):
    """reference inference.py:38-43."""
    clip_len = int(n_samples / 16000 * fps)
    n_audio_samples = round(audio_unit * n_motions)
    n_subdivision = 1 if clip_len <= n_motions else math.ceil(clip_len / n_motions)
    n_padding_audio_samples = n_audio_samples * n_subdivision - n_samples
    n_padding_frames = math.ceil(n_padding_audio_samples / audio_unit)
    return dict(clip_len=clip_len, n_audio_samples=n_audio_samples, n_subdivision=n_subdivision,
                n_padding_audio_samples=n_padding_audio_samples, n_padding_frames=n_padding_frames)


def infer_coeffs(sd, sched, audio, shape_coef, style_feat, motion_at_T, z_lists, *, fps=25, n_motions=100,
                 n_prev=10, audio_unit=640.0, cfg_mode=None, cfg_cond=("audio", "style"), cfg_scale=1.15,
                 dynamic_threshold=None, use_indicator=True, **net_kw):
    """reference inference.py:34-75 with noise injected.  audio: (S,) fp32; n_repetitions == motion_at_T.shape[0];
    z_lists[i] is the per-step noise dict of window i.  Window i>0 re-uses window 0's x_T (inference.py:64)."""
    plan = window_plan(audio.shape[0], fps, n_motions, audio_unit)
    if plan["n_padding_audio_samples"] > 0:
        audio = np.concatenate([audio, np.zeros(plan["n_padding_audio_samples"], np.float32)])
    n_sub = plan["n_subdivision"]
    n_rep = motion_at_T.shape[0]
    audio_feat = extract_audio_feature(sd, audio[None], fps, n_motions * n_sub)
    coef_list = []
    prev_m = prev_a = None
    for i in range(n_sub):
        indicator = np.ones((n_rep, n_motions), np.float32) if use_indicator else None
        if indicator is not None and i == n_sub - 1 and plan["n_padding_frames"] > 0:
            indicator[:, -plan["n_padding_frames"]:] = 0
        a_in = np.broadcast_to(audio_feat[:, i * n_motions:(i + 1) * n_motions], (n_rep, n_motions, audio_feat.shape[-1]))
        out = diffusion.sample(sd, sched, a_in, shape_coef, style_feat, motion_at_T, z_lists[i], prev_m, prev_a,
                               indicator=indicator, cfg_mode=cfg_mode or "incremental", cfg_cond=cfg_cond,
                               cfg_scale=cfg_scale, dynamic_threshold=dynamic_threshold, n_motions=n_motions,
                               n_prev=n_prev, **net_kw)
        prev_m = out[:, -n_prev:].copy()
        prev_a = np.ascontiguousarray(a_in[:, -n_prev:])
        if i == n_sub - 1 and plan["n_padding_frames"] > 0:
            out = out[:, :-plan["n_padding_frames"]]
        coef_list.append(out)
    return np.concatenate(coef_list, axis=1)
