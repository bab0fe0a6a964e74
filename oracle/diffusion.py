"""Oracle: diffusion schedule, denoising network, MSMD.forward and the CFG/DDPM sampler
(numpy fp32; test infrastructure).

Follows reference model.py:20-71 (DiffusionSchedule), 146-248 (MSMD.forward),
283-440 (MSMD.sample), 820-996 (DenoisingNetwork_MSMD) and
utils/model_common.py:86-107 (PositionalEncoding, enc_dec_mask).
torch.nn.TransformerDecoderLayer (post-norm, GELU, batch_first) is third-party
arithmetic (torch==2.0.0); its published algorithm is restated in decoder_layer().
"""
from __future__ import annotations

import math

import numpy as np

from . import nn
from .audio_encoder import extract_audio_feature

F32 = np.float32


# --------------------------------------------------------------------------- schedule
def diffusion_schedule(num_steps: int, mode: str = "cosine", beta_1=1e-4, beta_T=0.02, s=0.008):
    """reference model.py:20-60.  Returns dict of five (T+1,) fp32 buffers.
    All arithmetic in fp32 in the reference's operation order."""
    if mode == "linear":
        betas = _torch_linspace(beta_1, beta_T, num_steps)
    elif mode == "quadratic":
        betas = _torch_linspace(beta_1 ** 0.5, beta_T ** 0.5, num_steps) ** F32(2)
    elif mode == "sigmoid":
        x = _torch_linspace(-5, 5, num_steps)
        betas = (F32(1) / (F32(1) + np.exp(-x))) * F32(beta_T - beta_1) + F32(beta_1)
    elif mode == "cosine":
        x = _torch_linspace(0, num_steps, num_steps + 1)
        ab = np.cos(((x / F32(num_steps)) + F32(s)) / F32(1 + s) * F32(math.pi) * F32(0.5)).astype(F32) ** F32(2)
        ab = (ab / ab[0]).astype(F32)
        betas = (F32(1) - (ab[1:] / ab[:-1])).astype(F32)
        betas = np.clip(betas, F32(0.0001), F32(0.999))
    else:
        raise ValueError(f"Unknown diffusion schedule {mode}!")
    betas = np.concatenate([np.zeros(1, F32), betas.astype(F32)])
    alphas = (F32(1) - betas).astype(F32)
    log_alphas = np.log(alphas).astype(F32)
    for i in range(1, log_alphas.shape[0]):
        log_alphas[i] = log_alphas[i] + log_alphas[i - 1]
    alpha_bars = np.exp(log_alphas).astype(F32)
    sigmas_flex = np.sqrt(betas).astype(F32)
    sigmas_inflex = np.zeros_like(sigmas_flex)
    for i in range(1, sigmas_flex.shape[0]):
        sigmas_inflex[i] = ((F32(1) - alpha_bars[i - 1]) / (F32(1) - alpha_bars[i])) * betas[i]
    sigmas_inflex = np.sqrt(sigmas_inflex).astype(F32)
    return dict(betas=betas, alphas=alphas, alpha_bars=alpha_bars, sigmas_flex=sigmas_flex,
                sigmas_inflex=sigmas_inflex)


def _torch_linspace(start, end, steps):
    """torch.linspace fp32 semantics: step computed in fp32, symmetric fill
    (values in the upper half are computed as end - step*(steps-1-i))."""
    start = F32(start)
    end = F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    half = steps // 2
    lo = (start + step * i.astype(F32)).astype(F32)
    hi = (end - step * (steps - 1 - i).astype(F32)).astype(F32)
    return np.where(i < half, lo, hi).astype(F32)


# --------------------------------------------------------------------------- masks
def enc_dec_mask(T, S, frame_width=2, expansion=0):
    """reference utils/model_common.py:103-107 (True = masked)."""
    mask = np.ones((T, S), dtype=bool)
    for i in range(T):
        mask[i, max(0, (i - expansion) * frame_width):(i + expansion + 1) * frame_width] = False
    return mask


def alignment_mask(n_prev=10, n_motions=100, align_mask_width=1):
    """reference model.py:879-883: (1+L, L) bool, first row all-False."""
    L = n_prev + n_motions
    m = enc_dec_mask(L, L, 1, align_mask_width - 1)
    return np.concatenate([np.zeros((1, L), dtype=bool), m], axis=0)


# --------------------------------------------------------------------------- denoiser
def decoder_layer(sd, p, x, mem, mem_mask, n_heads):
    """nn.TransformerDecoderLayer(norm_first=False, activation='gelu'), eval mode."""
    sa = nn.mha(x, x, x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], n_heads)
    x = nn.layer_norm(x + sa, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    ca = nn.mha(x, mem, mem, sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"],
                sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"], n_heads,
                mask=mem_mask)
    x = nn.layer_norm(x + ca, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    ff = nn.linear(nn.gelu(nn.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                   sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return nn.layer_norm(x + ff, sd[p + "norm3.weight"], sd[p + "norm3.bias"])


def _mlp2(sd, p, x):
    """nn.Sequential(Linear, GELU, Linear) with indices 0 and 2."""
    return nn.linear(nn.gelu(nn.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])), sd[p + "2.weight"],
                     sd[p + "2.bias"])


def denoising_net(sd, motion_feat, audio_feat, person_feat, static_style_feat, prev_motion_feat,
                  prev_audio_feat, step, indicator=None, *, prefix="denoising_net.", n_heads=8,
                  n_prev=10, num_of_basis=4, n_diff_steps=500, align_mask_width=1,
                  use_head_alpha=False, keep_separate=False, regularize_alpha="None"):
    """reference model.py:914-996 (architecture='decoder', use_indicator).  Learnable PE when the state dict holds
    `PE`; otherwise the sinusoidal module's eval forward, which adds the single table row pe[seq_len] to every position
    (utils/model_common.py:99-101).  regularize_alpha='sigmoid': model.py:973-974."""
    P = prefix
    d = sd[P + "person_proj.weight"].shape[0]
    te = nn.sinusoid_table(n_diff_steps + 1, d)  # TE.pe
    step = np.asarray(step, dtype=np.int64)
    diff_emb = _mlp2(sd, P + "diff_step_map.", te[0, step])[:, None]  # (N,1,d)
    person = nn.linear(person_feat, sd[P + "person_proj.weight"], sd[P + "person_proj.bias"]) + diff_emb
    feats = np.concatenate([prev_motion_feat, motion_feat], axis=1)
    if indicator is not None:
        N = indicator.shape[0]
        ind = np.concatenate([np.zeros((N, n_prev), F32), nn.f32(indicator)], axis=1)[..., None]
        feats = np.concatenate([feats, ind], axis=-1)
    feats = nn.linear(feats, sd[P + "feature_proj.weight"], sd[P + "feature_proj.bias"])
    feats = np.concatenate([person, feats], axis=1)
    if P + "PE" in sd:
        feats = feats + nn.f32(sd[P + "PE"])
    else:
        feats = feats + nn.sinusoid_table(600, d)[0, feats.shape[1]]
    mem = np.concatenate([prev_audio_feat, audio_feat], axis=1)
    L = mem.shape[1]
    mask = alignment_mask(n_prev, L - n_prev, align_mask_width) if align_mask_width > 0 else None
    x = feats
    i = 0
    while f"{P}transformer.layers.{i}.linear1.weight" in sd:
        x = decoder_layer(sd, f"{P}transformer.layers.{i}.", x, mem, mask, n_heads)
        i += 1
    target = _mlp2(sd, P + "motion_dec.", x[:, 1:])  # (N, L, d_motion + nb)
    Lm = target.shape[1]
    static = []
    for b in range(num_of_basis):
        sb = _mlp2(sd, f"{P}static_feature_mapping.{b}.", static_style_feat)  # (N,1,dm)
        static.append(np.tile(sb, (1, Lm, 1))[:, :, None])
    static = np.concatenate(static, axis=2)  # (N, L, nb, dm)
    alphas = target[:, :, -num_of_basis:]
    if regularize_alpha == "sigmoid":
        alphas = (F32(1) / (F32(1) + np.exp(-alphas))).astype(F32)
    dynamic = target[:, :, :-num_of_basis]
    if use_head_alpha:
        summed = (static * alphas[..., None]).sum(axis=2)
    else:
        if static.shape[0] != alphas.shape[0]:
            static = np.tile(static, (alphas.shape[0], 1, 1, 1))
        face = (static[..., :-3] * alphas[..., None]).sum(axis=2)
        pose = static[..., -3:].sum(axis=2)
        summed = np.concatenate([face, pose], axis=-1)
    if keep_separate:
        return dynamic, static, alphas
    return (dynamic + summed).astype(F32)


# --------------------------------------------------------------------------- MSMD.forward (deterministic form)
def msmd_forward(sd, sched, motion_feat, audio_or_feat, shape_feat, style_feat, time_step, eps,
                 prev_motion_feat=None, prev_audio_feat=None, indicator=None, *, fps=25, n_motions=100,
                 null_style_mask=None, null_audio_mask=None, **net_kw):
    """reference model.py:146-248 with the stochastic draws injected:
    ``time_step`` (N,), ``eps`` (N,L,67) and optional boolean CFG masks.
    Returns (eps, target (N,110,67), audio_feat (N,100,512))."""
    N = motion_feat.shape[0]
    if audio_or_feat.ndim == 2:
        assert audio_or_feat.shape[1] == 16000 * n_motions / fps
        audio_feat_saved = extract_audio_feature(sd, audio_or_feat, fps, n_motions)
    else:
        assert audio_or_feat.shape[1] == n_motions
        audio_feat_saved = nn.f32(audio_or_feat)
    audio_feat = audio_feat_saved.copy()
    if shape_feat.ndim == 2:
        shape_feat = shape_feat[:, None]
    if style_feat.ndim == 2:
        style_feat = style_feat[:, None]
    if prev_motion_feat is None:
        prev_motion_feat = np.broadcast_to(sd["start_motion_feat"], (N,) + sd["start_motion_feat"].shape[1:])
    if prev_audio_feat is None:
        prev_audio_feat = np.broadcast_to(sd["start_audio_feat"], (N,) + sd["start_audio_feat"].shape[1:])
    if null_style_mask is not None:
        style_feat = np.where(null_style_mask[:, None, None], sd["null_style_feat"], style_feat)
    if null_audio_mask is not None:
        audio_feat = np.where(null_audio_mask[:, None, None], sd["null_audio_feat"], audio_feat)
    person = np.concatenate([nn.f32(shape_feat), nn.f32(style_feat)], axis=-1)
    ab = sched["alpha_bars"][np.asarray(time_step)]
    c0 = np.sqrt(ab).astype(F32)[:, None, None]
    c1 = np.sqrt(F32(1) - ab).astype(F32)[:, None, None]
    noisy = (c0 * nn.f32(motion_feat) + c1 * nn.f32(eps)).astype(F32)
    target = denoising_net(sd, noisy, audio_feat, person, nn.f32(style_feat), nn.f32(prev_motion_feat),
                           nn.f32(prev_audio_feat), time_step, indicator, **net_kw)
    return eps, target, audio_feat_saved


# --------------------------------------------------------------------------- sampler
def cfg_entries(cfg_cond, cfg_mode):
    """Which (audio, style) each CFG entry sees: list of (use_audio, use_style) flags
    in batch order (reference model.py:340-366).  Entry 0 is the null entry."""
    entries = [("audio" not in cfg_cond, "style" not in cfg_cond)]
    for cond in cfg_cond:
        if cond == "audio":
            entries.append((True, "style" not in cfg_cond))
        elif cond == "style":
            if cfg_mode == "independent":
                entries.append(("audio" not in cfg_cond, True))
            elif cfg_mode == "incremental":
                entries.append((True, True))
            else:
                raise NotImplementedError(f"Unknown cfg_mode {cfg_mode}")
    return entries


def sample(sd, sched, audio_feat, shape_feat, style_feat, motion_at_T, z_list, prev_motion_feat=None,
           prev_audio_feat=None, indicator=None, cfg_mode="incremental", cfg_cond=("audio", "style"),
           cfg_scale=1.15, flexibility=0, dynamic_threshold=None, target="sample", n_motions=100, guidance=None,
           separate=False, **net_kw):
    """reference model.py:283-440 with noise injected (guidance: model.py:762-767; separate: model.py:442-651): ``z_list[t]`` is the draw used at
    step t (t = T..2; step 1 uses zeros).  audio_feat is (N, L, 512) features."""
    N = audio_feat.shape[0]
    T = sched["betas"].shape[0] - 1
    cfg_cond = [c for c in cfg_cond if c in ("audio", "style")]
    if not isinstance(cfg_scale, (list, tuple)):
        cfg_scale = [cfg_scale] * len(cfg_cond)
    if cfg_cond:
        pairs = sorted(zip(cfg_cond, cfg_scale), key=lambda x: ["audio", "style"].index(x[0]))
        cfg_cond, cfg_scale = [p[0] for p in pairs], [p[1] for p in pairs]
    if shape_feat.ndim == 2:
        shape_feat = shape_feat[:, None]
    if style_feat.ndim == 2:
        style_feat = style_feat[:, None]
    if prev_motion_feat is None:
        prev_motion_feat = np.broadcast_to(sd["start_motion_feat"], (N,) + sd["start_motion_feat"].shape[1:])
    if prev_audio_feat is None:
        prev_audio_feat = np.broadcast_to(sd["start_audio_feat"], (N,) + sd["start_audio_feat"].shape[1:])
    null_audio = np.broadcast_to(sd.get("null_audio_feat", np.zeros((1, 1, audio_feat.shape[-1]), F32)),
                                 audio_feat.shape)
    null_style = np.broadcast_to(sd.get("null_style_feat", np.zeros_like(style_feat[:1])), style_feat.shape)
    audio_in, person_in = [], []
    for use_a, use_s in cfg_entries(cfg_cond, cfg_mode):
        audio_in.append(audio_feat if use_a else null_audio)
        person_in.append(np.concatenate([shape_feat, style_feat if use_s else null_style], axis=-1))
    n_entries = len(audio_in)
    audio_in = nn.f32(np.concatenate(audio_in, axis=0))
    person_in = nn.f32(np.concatenate(person_in, axis=0))
    prev_m = nn.f32(np.concatenate([prev_motion_feat] * n_entries, axis=0))
    prev_a = nn.f32(np.concatenate([prev_audio_feat] * n_entries, axis=0))
    ind_in = np.concatenate([indicator] * n_entries, axis=0) if indicator is not None else None
    style_in = nn.f32(np.concatenate([style_feat] * n_entries, axis=0))  # static branch: real style everywhere
    x = nn.f32(motion_at_T)
    cum_static = np.zeros_like(x)
    for t in range(T, 0, -1):
        z = nn.f32(z_list[t]) if t > 1 else np.zeros_like(x)
        alpha = sched["alphas"][t]
        alpha_bar = sched["alpha_bars"][t]
        alpha_bar_prev = sched["alpha_bars"][t - 1]
        sigma = F32(sched["sigmas_flex"][t] * F32(flexibility) + sched["sigmas_inflex"][t] * F32(1 - flexibility))
        motion_in = np.concatenate([x] * n_entries, axis=0)
        if guidance is not None:
            motion_in[:, guidance[0], :] = guidance[1]
        step_in = np.full((N * n_entries,), t, dtype=np.int64)
        if separate:
            dyn, static4, alpha_t = denoising_net(sd, motion_in, audio_in, person_in, style_in, prev_m, prev_a, step_in,
                                                  ind_in, keep_separate=True, **net_kw)
            static = np.concatenate([(static4[..., :-3] * alpha_t[..., None]).sum(axis=2), static4[..., -3:].sum(axis=2)],
                                    axis=-1).astype(F32)
            res = (dyn + static).astype(F32)
        else:
            res = denoising_net(sd, motion_in, audio_in, person_in, style_in, prev_m, prev_a, step_in, ind_in, **net_kw)
        if dynamic_threshold:
            dt_ratio, dt_min, dt_max = dynamic_threshold
            absr = np.abs(res[:, -n_motions:].reshape(N * n_entries, -1))
            s = np.quantile(absr.astype(np.float64), dt_ratio, axis=1).astype(F32)
            s = np.clip(s, dt_min, dt_max)[:, None, None]
            res = np.clip(res, -s, s)
        streams = [res] + ([static, dyn, alpha_t] if separate else [])
        streams = [[r.copy() for r in np.split(v, n_entries, axis=0)] for v in streams]
        heads = [st[0][:, -n_motions:] for st in streams]  # VIEWS of entry 0: in-place accumulation (model.py:407-415)
        for i in range(n_entries - 1):
            for st, hd in zip(streams, heads):
                if cfg_mode == "independent":
                    hd += F32(cfg_scale[i]) * (st[i + 1][:, -n_motions:] - st[0][:, -n_motions:])
                elif cfg_mode == "incremental":
                    hd += F32(cfg_scale[i]) * (st[i + 1][:, -n_motions:] - st[i][:, -n_motions:])
                else:
                    raise NotImplementedError(f"Unknown cfg_mode {cfg_mode}")
        theta = heads[0]
        if target == "noise":
            c0 = F32(1) / np.sqrt(alpha)
            c1 = (F32(1) - alpha) / np.sqrt(F32(1) - alpha_bar)
            x = (c0 * (x - c1 * theta) + sigma * z).astype(F32)
        elif target == "sample":
            c0 = (F32(1) - alpha_bar_prev) * np.sqrt(alpha) / (F32(1) - alpha_bar)
            c1 = (F32(1) - alpha) * np.sqrt(alpha_bar_prev) / (F32(1) - alpha_bar)
            x = (c0 * x + c1 * theta + sigma * z).astype(F32)
        else:
            raise ValueError(f"Unknown target type: {target}")
        if separate:
            cum_static = (cum_static + c1 * heads[1]).astype(F32)
    if separate:
        return x, heads[2], cum_static, heads[3]
    return x
