"""Windowed inference driver (drop-in surface of reference inference.py:34-75, 85-103).

``infer_coeffs`` keeps the reference's signature and window maths (bit-exact integer arithmetic) and
its behaviours (one encoder pass over the whole zero-padded clip; window i>0 re-uses window 0's x_T,
inference.py:64; last 10 motion/audio frames handed to the next window).

SURVEY.md section 8(f) n3, the callers either side of the sampler:
  * ``normalize_motion_coeff`` / ``query_for_motion_coeff`` -- style-clip ingestion (inference.py:109-183): coefficient
    statistics, optional 30 -> 25 fps linear resampling on a normalised time axis (scipy.interpolate.interp1d's
    linear rule), concatenation to (1, T, 53) + the dummy (1, 100) shape row;
  * ``denormalize_coeffs`` -- output de-normalisation (inference.py:274-275);
  * ``infer_coeffs_batch`` -- many clips at once: window i of every clip that still has one goes through ONE
    ``model.sample`` call (the sampler is launch-bound at batch 1: 0.71 ms/step at B = 1 vs 3.5 ms/step at B = 64), one
    encoder pass per distinct padded length; per-clip results equal ``infer_coeffs`` on that clip.
The reference script's media IO (librosa, cv2, wav / video writing) stays outside the hot path.
"""
from __future__ import annotations

import math
import pickle as pkl
from pathlib import Path

import numpy as np

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
    """Coefficients for one clip of any length (reference inference.py:34-75; same signature, window arithmetic and hand-off
    rules).  The clip is zero-padded to a whole number of n_motions-frame windows and encoded ONCE; each window is one
    `model.sample` call conditioned on the previous window's last n_prev_motions motion / audio-feature frames, every
    window after the first starts from window 0's x_T, and the frames that only cover the padding are cut from the result.
    ``noise`` (optional, replay): {'xT': tensor, 'z': [per-window dict of per-step draws]}."""
    L, keep = args.n_motions, args.n_prev_motions
    _, _, n_windows, pad_samples, pad_frames = window_plan(len(audio), args.fps, L, audio_unit)
    tail = max(pad_frames, 0)                       # frames of the last window that lie entirely in the zero padding
    wave = F.pad(audio, (0, pad_samples), value=0) if pad_samples > 0 else audio
    per_window = model.extract_audio_feature(wave.unsqueeze(0), L * n_windows).split(L, dim=1)
    guidance = dict(cfg_mode=cfg_mode, cfg_cond=cfg_cond, cfg_scale=cfg_scale, dynamic_threshold=dynamic_threshold)
    history = (None, None, None if noise is None else noise["xT"])      # (prev motion, prev audio features, x_T)
    pieces = []
    for w, feat in enumerate(per_window):
        cut = tail if w == n_windows - 1 else 0
        indicator = None
        if args.use_indicator:
            indicator = torch.ones((n_repetitions, L), device=model.device)
            if cut:
                indicator[:, L - cut:] = 0
        style = style_feats[w] if isinstance(style_feats, list) else style_feats
        x0, x_T, feat_used = model.sample(feat.expand(n_repetitions, -1, -1), shape_coef, style, *history, indicator=indicator,
                                          noise=None if noise is None else noise["z"][w], **guidance)
        history = (x0[:, -keep:].clone(), feat_used[:, -keep:], x_T)
        pieces.append(x0[:, :L - cut] if cut else x0)
    return torch.cat(pieces, dim=1)


def load_model(model_root: str, model_name: str, iter_num: str, device: torch.device):
    """reference inference.py:85-103 (same directory layout and checkpoint keys)."""
    import os
    model_args = load_args(Path(os.path.join(model_root, "DPT", model_name)))
    # the checkpoint holds every audio-encoder tensor (strict load below): no Hugging Face copy is needed to rebuild the model
    model_args.audio_encoder_weights = "checkpoint"
    model = get_diffusion_model(model_args, device)
    ckpt = Path(model_root) / "DPT" / model_name / "checkpoints" / f"iter_{iter_num}.pt"
    data = torch.load(ckpt, map_location=device, weights_only=False)  # reference checkpoints pickle their args Namespace
    style_enc = get_style_encoder(model_args, model_args.style_enc_model_style)
    style_enc.load_state_dict(data["style_enc"])
    style_enc.to(device).eval()
    model.load_state_dict(data["model"])
    model.eval()
    return model, style_enc, model_args


# ----------------------------------------------------------------------------- style-clip ingestion / output scaling
def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def resample_linear(x, n_out):
    """scipy.interpolate.interp1d(linspace(0, 1, n), x, axis=0)(linspace(0, 1, n_out)) (inference.py:160-170): linear
    interpolation on a normalised time axis, endpoints included."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if n_out == n:
        return x
    xs, xn = np.linspace(0, 1, num=n), np.linspace(0, 1, num=n_out)
    hi = np.clip(np.searchsorted(xs, xn, side="left"), 1, n - 1)   # interp1d: x_new in (xs[hi-1], xs[hi]]
    lo = hi - 1
    slope = (x[hi] - x[lo]) / (xs[hi] - xs[lo])[:, None]
    return slope * (xn - xs[lo])[:, None] + x[lo]


def normalize_motion_coeff(expression_coef, head_rot, coef_stats, device="cuda", original_fps=30, target_fps=25):
    """Body of reference query_for_motion_coeff (inference.py:139-181) on in-memory arrays / tensors."""
    e = (_np(expression_coef) - _np(coef_stats["exp_mean"])) / (_np(coef_stats["exp_std"]) + 1e-9)
    h = (_np(head_rot) - _np(coef_stats["pose_mean"])) / (_np(coef_stats["pose_std"]) + 1e-9)
    if original_fps is not None and original_fps != target_fps:
        n_new = int(round(e.shape[0] / original_fps * target_fps))
        e, h = resample_linear(e, n_new), resample_linear(h, n_new)
    et = torch.from_numpy(np.ascontiguousarray(e)).to(device).unsqueeze(0).float()
    ht = torch.from_numpy(np.ascontiguousarray(h)).to(device).unsqueeze(0).float()
    return torch.cat([et, ht], dim=2).float(), torch.zeros((1, 100), device=device).float()


def query_for_motion_coeff(args, expression_code_full_path, head_rot_full_path, device="cuda", original_fps=30,
                           target_fps=25):
    """reference inference.py:109-183 (same signature; pickle inputs)."""
    with open(args.coef_dict_path, "rb") as f:
        coef_stats = pkl.load(f)
    with open(expression_code_full_path, "rb") as f:
        expression_coef = pkl.load(f)
    with open(head_rot_full_path, "rb") as f:
        head_rot = pkl.load(f)
    return normalize_motion_coeff(expression_coef, head_rot, coef_stats, device, original_fps, target_fps)


def denormalize_coeffs(overall_coef, coef_stats):
    """reference inference.py:274-275: (n_rep, T, 53) normalised -> (expression code (T, 50), head rotation (T, 3))
    of repetition 0 in the data's units."""
    st = {k: (v.to(overall_coef.device) if torch.is_tensor(v) else torch.as_tensor(v, device=overall_coef.device))
          for k, v in coef_stats.items()}
    return (overall_coef[0, :, :-3] * st["exp_std"] + st["exp_mean"],
            overall_coef[0, :, -3:] * st["pose_std"] + st["pose_mean"])


# ----------------------------------------------------------------------------- many clips per denoise step
@torch.no_grad()
def infer_coeffs_batch(model, args, audios, shape_coefs, audio_unit, style_feats, cfg_mode=None, cfg_cond=None,
                       cfg_scale: float = 1.15, dynamic_threshold=(0, 1, 4), noise=None):
    """`infer_coeffs` for a list of clips (1-D audio tensors of any lengths), one repetition each, with window i of
    all clips batched into one `model.sample` call.  shape_coefs: (n_clips, 100); style_feats: (n_clips, d_style).
    `noise` (optional): list of per-clip {'xT', 'z'} dicts as `infer_coeffs` takes.  Returns a list of (1, clip_len, C)
    tensors, clip c equal to infer_coeffs(model, args, audios[c], shape_coefs[c:c+1], audio_unit, style_feats[c:c+1])."""
    n = len(audios)
    L, Lp = args.n_motions, args.n_prev_motions
    plans = [window_plan(len(a), args.fps, L, audio_unit) for a in audios]
    # one encoder pass per distinct padded length
    feats = [None] * n
    by_sub = {}
    for c, pl in enumerate(plans):
        by_sub.setdefault(pl[2], []).append(c)
    for n_sub, clips in by_sub.items():
        batch = torch.stack([F.pad(audios[c], (0, plans[c][3]), value=0) if plans[c][3] > 0 else audios[c]
                             for c in clips])
        f = model.extract_audio_feature(batch, L * n_sub)
        for j, c in enumerate(clips):
            feats[c] = f[j:j + 1]
    outs = [[] for _ in range(n)]
    prev_m = [None] * n
    prev_a = [None] * n
    x_T = [None] * n
    for i in range(max(pl[2] for pl in plans)):
        act = [c for c in range(n) if i < plans[c][2]]
        ind = torch.ones((len(act), L), device=model.device) if args.use_indicator else None
        for j, c in enumerate(act):
            if ind is not None and i == plans[c][2] - 1 and plans[c][4] > 0:
                ind[j, -plans[c][4]:] = 0
        audio_in = torch.cat([feats[c][:, i * L:(i + 1) * L] for c in act], dim=0)
        kw = dict(indicator=ind, cfg_mode=cfg_mode, cfg_cond=cfg_cond, cfg_scale=cfg_scale,
                  dynamic_threshold=dynamic_threshold)
        if noise is not None:
            kw["noise"] = {t: torch.cat([noise[c]["z"][i][t] for c in act], dim=0) for t in noise[act[0]]["z"][i]}
        idx = torch.as_tensor(act, device=model.device)
        shape_in, style_in = shape_coefs[idx], style_feats[idx]
        if i == 0:
            xT = torch.cat([noise[c]["xT"] for c in act], dim=0) if noise is not None else None
            motion, nT, pa = model.sample(audio_in, shape_in, style_in, motion_at_T=xT, **kw)
        else:
            # windows i > 0 re-use their clip's x_T and take the previous window's last frames (inference.py:60-69);
            # model.sample substitutes the learned start tokens only when BOTH prev tensors are None, which cannot
            # happen here because every active clip ran window i - 1
            motion, nT, pa = model.sample(audio_in, shape_in, style_in, torch.cat([prev_m[c] for c in act], dim=0),
                                          torch.cat([prev_a[c] for c in act], dim=0),
                                          torch.cat([x_T[c] for c in act], dim=0), **kw)
        for j, c in enumerate(act):
            prev_m[c] = motion[j:j + 1, -Lp:].clone()
            prev_a[c] = pa[j:j + 1, -Lp:]
            x_T[c] = nT[j:j + 1]
            m = motion[j:j + 1]
            if i == plans[c][2] - 1 and plans[c][4] > 0:
                m = m[:, :-plans[c][4]]
            outs[c].append(m)
    return [torch.cat(o, dim=1) for o in outs]


# ----------------------------------------------------------------------------- command line (reference flag names)
def build_parser():
    """The reference's inference flags (inference.py:190-201), same names and defaults."""
    import argparse
    ap = argparse.ArgumentParser(description="Single style + audio inference for MSMD on MI355X.")
    for name in ("model_root", "model_name", "model_iter", "style_clip_exp_code_path", "style_clip_head_rot_path",
                 "audio_clip"):
        ap.add_argument("--" + name, type=str, required=True)
    ap.add_argument("--coef_dict_path", type=str, default="PATH-TO-COEF-STATS")
    ap.add_argument("--cfg_level", type=float, default=1.4)
    ap.add_argument("--output_dir", type=str, default="/experiments/refactor")
    ap.add_argument("--versions_of_render", type=int, default=1)
    return ap


def load_audio_16k(path):
    """16 kHz mono samples from a decoded file (.npy / pickle of a float array).  Decoding compressed media (the
    reference calls librosa.load, inference.py:226) is outside the hot path; hand this function the decoded samples."""
    path = str(path)
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    with open(path, "rb") as f:
        return np.asarray(pkl.load(f), dtype=np.float32)


def main(argv=None):
    """The non-media part of reference inference.py:189-279: load model + style encoder, ingest the style clip, z-norm
    the audio, sample the style code, run infer_coeffs per repetition seed, de-normalise and write the two pickles
    (`overall_exp_code_*`, `overall_head_rot_*`) under <output_dir>/<model>_iter_<iter>/temp/.  Mesh decoding and
    video rendering of the reference script are not part of this path."""
    import os
    args = build_parser().parse_args(argv)
    device = torch.device("cuda")
    model, style_enc, model_args = load_model(args.model_root, args.model_name, args.model_iter, device)
    motion_coeff, shape_coef = query_for_motion_coeff(args, args.style_clip_exp_code_path, args.style_clip_head_rot_path,
                                                      device=device)
    shape_coef = shape_coef.unsqueeze(1)
    audio = load_audio_16k(args.audio_clip)
    audio = (audio - audio.mean()) / (audio.std() + 1e-5)
    audio_tensor = torch.from_numpy(audio).float().to(device)
    style_clip = motion_coeff[:, :100, :]
    style_coeff = style_enc.sample(style_clip) if model_args.style_enc_model_style.startswith("vae") else style_enc(style_clip)
    with open(args.coef_dict_path, "rb") as f:
        coef_stats = {k: v.to(device) for k, v in pkl.load(f).items()}
    style_name = os.path.splitext(os.path.basename(args.style_clip_exp_code_path))[0]
    audio_name = os.path.splitext(os.path.basename(args.audio_clip))[0]
    clip = f"style=_{style_name}_audio={audio_name}"
    temp = os.path.join(args.output_dir, f"{args.model_name}_iter_{args.model_iter}", "temp")
    os.makedirs(temp, exist_ok=True)
    written = []
    for seed in range(args.versions_of_render):
        np.random.seed(seed)
        torch.manual_seed(seed)
        coef = infer_coeffs(model, model_args, audio_tensor, shape_coef, 640.0, style_coeff, cfg_scale=args.cfg_level,
                            dynamic_threshold=None)
        exp_code, head_rot = denormalize_coeffs(coef, coef_stats)
        for tag, val in (("exp_code", exp_code), ("head_rot", head_rot)):
            out = os.path.join(temp, f"overall_{tag}_{clip}_seed_{seed}.pkl")
            with open(out, "wb") as f:
                pkl.dump(val.cpu().numpy(), f)
            written.append(out)
    return written


if __name__ == "__main__":
    main()
