"""Windowed inference driver (drop-in surface of reference inference.py:34-75, 85-103).

``infer_coeffs`` keeps the reference's signature and window maths (bit-exact integer arithmetic) and
its behaviours (one encoder pass over the whole zero-padded clip; window i>0 re-uses window 0's x_T,
inference.py:64; last 10 motion/audio frames handed to the next window).  File IO / CLI of the
reference's script (librosa, cv2, pickle outputs) is outside the hot path.
"""
from __future__ import annotations

import math
from pathlib import Path

import torch
import torch.nn.functional as F

from .model import get_diffusion_model
from .style_encoder import get_style_encoder
from .utils.model_common import load_args


def window_plan(n_samples: int, fps, n_motions: int, audio_unit: float):
    """reference inference.py:38-43."""
    clip_len = int(n_samples / 16000 * fps)
    n_audio_samples = round(audio_unit * n_motions)
    n_subdivision = 1 if clip_len <= n_motions else math.ceil(clip_len / n_motions)
    n_padding_audio_samples = n_audio_samples * n_subdivision - n_samples
    n_padding_frames = math.ceil(n_padding_audio_samples / audio_unit)
    return clip_len, n_audio_samples, n_subdivision, n_padding_audio_samples, n_padding_frames


@torch.no_grad()
def infer_coeffs(model, args, audio, shape_coef, audio_unit, style_feats=None, n_repetitions: int = 1, cfg_mode=None,
                 cfg_cond=None, cfg_scale: float = 1.15, include_shape: bool = False, dynamic_threshold=(0, 1, 4),
                 noise=None):
    """reference inference.py:34-75.  ``noise`` (optional): {'xT': tensor, 'z': [dict per window]} for replay."""
    _, _, n_subdivision, n_pad_samples, n_padding_frames = window_plan(len(audio), args.fps, args.n_motions, audio_unit)
    stride = args.n_motions
    if n_pad_samples > 0:
        audio = F.pad(audio, (0, n_pad_samples), value=0)
    audio_feat = model.extract_audio_feature(audio.unsqueeze(0), args.n_motions * n_subdivision)
    coef_list = []
    prev_motion_feat = prev_audio_feat = noise_T = None
    for i in range(n_subdivision):
        start_idx = i * stride
        end_idx = start_idx + args.n_motions
        indicator = torch.ones((n_repetitions, args.n_motions)).to(model.device) if args.use_indicator else None
        if indicator is not None and i == n_subdivision - 1 and n_padding_frames > 0:
            indicator[:, -n_padding_frames:] = 0
        audio_in = audio_feat[:, start_idx:end_idx].expand(n_repetitions, -1, -1)
        style_feat = style_feats[i] if isinstance(style_feats, list) else style_feats
        zi = noise["z"][i] if noise is not None else None
        if i == 0:
            motion_feat, noise_T, prev_audio_feat = model.sample(
                audio_in, shape_coef, style_feat, motion_at_T=noise["xT"] if noise is not None else None,
                indicator=indicator, cfg_mode=cfg_mode, cfg_cond=cfg_cond, cfg_scale=cfg_scale,
                dynamic_threshold=dynamic_threshold, noise=zi)
        else:
            motion_feat, noise_T, prev_audio_feat = model.sample(
                audio_in, shape_coef, style_feat, prev_motion_feat, prev_audio_feat, noise_T, indicator=indicator,
                cfg_mode=cfg_mode, cfg_cond=cfg_cond, cfg_scale=cfg_scale, dynamic_threshold=dynamic_threshold,
                noise=zi)
        prev_motion_feat = motion_feat[:, -args.n_prev_motions:].clone()
        prev_audio_feat = prev_audio_feat[:, -args.n_prev_motions:]
        motion_coef = motion_feat
        if i == n_subdivision - 1 and n_padding_frames > 0:
            motion_coef = motion_coef[:, :-n_padding_frames]
        coef_list.append(motion_coef)
    return torch.cat(coef_list, dim=1)


def load_model(model_root: str, model_name: str, iter_num: str, device: torch.device):
    """reference inference.py:85-103 (same directory layout and checkpoint keys)."""
    import os
    model_args = load_args(Path(os.path.join(model_root, "DPT", model_name)))
    model = get_diffusion_model(model_args, device)
    ckpt = Path(model_root) / "DPT" / model_name / "checkpoints" / f"iter_{iter_num}.pt"
    data = torch.load(ckpt, map_location=device)
    style_enc = get_style_encoder(model_args, model_args.style_enc_model_style)
    style_enc.load_state_dict(data["style_enc"])
    style_enc.to(device).eval()
    model.load_state_dict(data["model"])
    model.eval()
    return model, style_enc, model_args
