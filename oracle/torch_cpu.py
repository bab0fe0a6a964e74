"""Oracle, torch-CPU form: the reference's CPU path restated on torch.nn.functional (test / measurement infrastructure;
never imported by the product).

The reference runs on CPU as fp32 torch modules (HF wav2vec2 + nn.TransformerDecoder + FLAME lbs, no autocast:
/root/reference/training_script.py:548-551); on the GPU box `/root/reference` does not exist, so bench.py's
`cpu_baseline` leg times THIS restatement on the host cores (torch.set_num_threads(all cores), fp32, eval arithmetic),
as SURVEY.md section 8(d) prescribes.  It mirrors the numpy oracle function for function (oracle/audio_encoder.py,
oracle/diffusion.py, oracle/flame.py -- each of which cites the reference lines it follows) and is pinned to it by
tests/test_oracle_vs_golden.py::test_torch_cpu_restatement_matches_reference_goldens.

  msmd_forward   reference model.py:146-248 (raw audio -> utils/wav2vec2.py:71-119 -> model.py:250-264 -> 820-996)
  denoise_step   one denoiser call on n_entries x B sequences = one step of model.py:283-440's loop body
  flame_lbs      reference utils/lbs.py:141-223 through utils/flame.py:180-244 (vertices only)
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .audio_encoder import CONV_KERNEL, CONV_STRIDE, crop_len, pad_audio_gather_index
from .diffusion import alignment_mask
from .nn import interp_linear_table, sinusoid_table


def to_torch(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd.items()}


def _interp(x, out_len):
    i0, i1, w1 = interp_linear_table(x.shape[1], out_len)
    if x.shape[1] == out_len:
        return x
    w1 = torch.from_numpy(w1)[None, :, None]
    return (1.0 - w1) * x[:, torch.from_numpy(i0)] + w1 * x[:, torch.from_numpy(i1)]


def audio_encoder(sd, audio_padded, output_fps=25, frame_num=None, n_heads=12, prefix="audio_encoder.",
                  stable_layer_norm=False):
    """oracle.audio_encoder.audio_encoder on torch ops; activations (B, C, T) inside the conv stack as the reference's HF
    modules keep them, (B, T, C) afterwards.  Base checkpoints: GroupNorm conv stack + post-LN layers; large ones
    (read off the state dict / `stable_layer_norm`): conv bias + LayerNorm over channels after every conv, pre-LN layers
    and ONE LayerNorm after the last (HubertEncoderStableLayerNorm; reference utils/hubert.py:13-51)."""
    fe = f"{prefix}feature_extractor.conv_layers."
    layer_mode = (fe + "1.layer_norm.weight") in sd
    x = audio_padded[:, None, :]
    for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE)):
        x = F.conv1d(x, sd[f"{fe}{i}.conv.weight"], sd.get(f"{fe}{i}.conv.bias"), stride=s)
        if layer_mode:
            x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[f"{fe}{i}.layer_norm.weight"],
                             sd[f"{fe}{i}.layer_norm.bias"], 1e-5).transpose(1, 2)
        elif i == 0:
            x = F.group_norm(x, x.shape[1], sd[f"{fe}0.layer_norm.weight"], sd[f"{fe}0.layer_norm.bias"], 1e-5)
        x = F.gelu(x)
    x = x.transpose(1, 2)
    if frame_num is not None:
        x = x[:, :crop_len(frame_num, output_fps)]
        out_len = frame_num
    else:
        out_len = int(x.shape[1] / 50.0 * output_fps)
    x = _interp(x, out_len)
    x = F.layer_norm(x, x.shape[-1:], sd[f"{prefix}feature_projection.layer_norm.weight"],
                     sd[f"{prefix}feature_projection.layer_norm.bias"], 1e-5)
    x = F.linear(x, sd[f"{prefix}feature_projection.projection.weight"], sd[f"{prefix}feature_projection.projection.bias"])
    base = f"{prefix}encoder.pos_conv_embed.conv."
    if base + "weight_g" in sd:
        g, v = sd[base + "weight_g"], sd[base + "weight_v"]
    else:
        g, v = sd[base + "parametrizations.weight.original0"], sd[base + "parametrizations.weight.original1"]
    w = v * (g / v.double().pow(2).sum(dim=(0, 1), keepdim=True).sqrt().float())
    pos = F.conv1d(x.transpose(1, 2), w, sd[base + "bias"], padding=64, groups=16)[:, :, :x.shape[1]].transpose(1, 2)
    x = x + F.gelu(pos)
    d = x.shape[-1]
    B, T, _ = x.shape
    hd = d // n_heads

    def attn(p, h):
        q = F.linear(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
        k = F.linear(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, d)
        return F.linear(a, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])

    def ffn(p, h):
        return F.linear(F.gelu(F.linear(h, sd[p + "feed_forward.intermediate_dense.weight"],
                                        sd[p + "feed_forward.intermediate_dense.bias"])),
                        sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
    ln = lambda h, key: F.layer_norm(h, (d,), sd[key + ".weight"], sd[key + ".bias"], 1e-5)
    if not stable_layer_norm:
        x = ln(x, f"{prefix}encoder.layer_norm")
    i = 0
    while f"{prefix}encoder.layers.{i}.attention.q_proj.weight" in sd:
        p = f"{prefix}encoder.layers.{i}."
        if stable_layer_norm:      # pre-LN (HubertEncoderLayerStableLayerNorm)
            x = x + attn(p, ln(x, p + "layer_norm"))
            x = x + ffn(p, ln(x, p + "final_layer_norm"))
        else:                      # post-LN
            x = ln(x + attn(p, x), p + "layer_norm")
            x = ln(x + ffn(p, x), p + "final_layer_norm")
        i += 1
    return ln(x, f"{prefix}encoder.layer_norm") if stable_layer_norm else x


def extract_audio_feature(sd, audio, fps=25, frame_num=100):
    idx = torch.from_numpy(pad_audio_gather_index(audio.shape[1]))
    h = audio_encoder(sd, audio[:, idx], fps, frame_num=frame_num * 2)
    return F.linear(_interp(h, frame_num), sd["audio_feature_map.weight"], sd["audio_feature_map.bias"])


def _mha(sd, p, q_in, kv_in, n_heads, mask=None):
    d = q_in.shape[-1]
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    B, Tq, _ = q_in.shape
    Tk = kv_in.shape[1]
    hd = d // n_heads
    q = F.linear(q_in, w[:d], b[:d]).view(B, Tq, n_heads, hd).transpose(1, 2)
    k = F.linear(kv_in, w[d:2 * d], b[d:2 * d]).view(B, Tk, n_heads, hd).transpose(1, 2)
    v = F.linear(kv_in, w[2 * d:], b[2 * d:]).view(B, Tk, n_heads, hd).transpose(1, 2)
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=None if mask is None else ~mask)
    return F.linear(a.transpose(1, 2).reshape(B, Tq, d), sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _mlp2(sd, p, x):
    return F.linear(F.gelu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])), sd[p + "2.weight"], sd[p + "2.bias"])


def denoising_net(sd, motion_feat, audio_feat, person_feat, static_style_feat, prev_motion_feat, prev_audio_feat, step,
                  indicator, n_heads=8, n_prev=10, num_of_basis=4, n_diff_steps=500, prefix="denoising_net."):
    """oracle.diffusion.denoising_net (learnable PE, use_indicator, align_mask_width 1, use_head_alpha False)."""
    P = prefix
    d = sd[P + "person_proj.weight"].shape[0]
    te = torch.from_numpy(sinusoid_table(n_diff_steps + 1, d))
    diff_emb = _mlp2(sd, P + "diff_step_map.", te[0, step])[:, None]
    person = F.linear(person_feat, sd[P + "person_proj.weight"], sd[P + "person_proj.bias"]) + diff_emb
    N = indicator.shape[0]
    ind = torch.cat([torch.zeros(N, n_prev), indicator], dim=1)[..., None]
    feats = torch.cat([torch.cat([prev_motion_feat, motion_feat], dim=1), ind], dim=-1)
    feats = F.linear(feats, sd[P + "feature_proj.weight"], sd[P + "feature_proj.bias"])
    x = torch.cat([person, feats], dim=1) + sd[P + "PE"]
    mem = torch.cat([prev_audio_feat, audio_feat], dim=1)
    mask = torch.from_numpy(alignment_mask(n_prev, mem.shape[1] - n_prev, 1))
    i = 0
    while f"{P}transformer.layers.{i}.linear1.weight" in sd:
        p = f"{P}transformer.layers.{i}."
        x = F.layer_norm(x + _mha(sd, p + "self_attn.", x, x, n_heads), (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        x = F.layer_norm(x + _mha(sd, p + "multihead_attn.", x, mem, n_heads, mask), (d,), sd[p + "norm2.weight"],
                         sd[p + "norm2.bias"], 1e-5)
        ff = F.linear(F.gelu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"],
                      sd[p + "linear2.bias"])
        x = F.layer_norm(x + ff, (d,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
        i += 1
    target = _mlp2(sd, P + "motion_dec.", x[:, 1:])
    Lm = target.shape[1]
    static = torch.stack([_mlp2(sd, f"{P}static_feature_mapping.{b}.", static_style_feat).expand(-1, Lm, -1)
                          for b in range(num_of_basis)], dim=2)                      # (Ns, L, nb, dm)
    alphas, dynamic = target[..., -num_of_basis:], target[..., :-num_of_basis]
    if static.shape[0] != alphas.shape[0]:
        static = static.repeat(alphas.shape[0] // static.shape[0], 1, 1, 1)
    face = (static[..., :-3] * alphas[..., None]).sum(dim=2)
    pose = static[..., -3:].sum(dim=2)
    return dynamic + torch.cat([face, pose], dim=-1)


@torch.no_grad()
def msmd_forward(sd, sched, motion_feat, audio, shape_feat, style_feat, time_step, eps, indicator, fps=25, n_motions=100):
    """oracle.diffusion.msmd_forward for raw audio, no CFG masking, start tokens as the previous window."""
    N = motion_feat.shape[0]
    audio_feat = extract_audio_feature(sd, audio, fps, n_motions)
    shape_feat, style_feat = shape_feat[:, None], style_feat[:, None]
    prev_m = sd["start_motion_feat"].expand(N, -1, -1)
    prev_a = sd["start_audio_feat"].expand(N, -1, -1)
    person = torch.cat([shape_feat, style_feat], dim=-1)
    ab = torch.from_numpy(np.asarray(sched["alpha_bars"]))[torch.as_tensor(time_step)]
    noisy = ab.sqrt()[:, None, None] * motion_feat + (1 - ab).sqrt()[:, None, None] * eps
    target = denoising_net(sd, noisy, audio_feat, person, style_feat, prev_m, prev_a, torch.as_tensor(time_step), indicator)
    return eps, target, audio_feat


@torch.no_grad()
def denoise_step(sd, x, audio_feat, shape_feat, style_feat, t, indicator, n_entries=3):
    """One loop body of the sampler (reference model.py:376-419): the denoiser on n_entries x B sequences
    (null / audio / audio + style entries of incremental CFG) -- the part that is > 99 % of inference time."""
    B = x.shape[0]
    null_a = sd["null_audio_feat"].expand(B, audio_feat.shape[1], -1)
    null_s = sd["null_style_feat"].expand(B, -1, -1)
    style_feat, shape_feat = style_feat[:, None], shape_feat[:, None]
    audio_in = torch.cat([null_a, audio_feat, audio_feat][:n_entries], 0)
    person_in = torch.cat([torch.cat([shape_feat, s], -1) for s in (null_s, null_s, style_feat)][:n_entries], 0)
    rep = lambda v: torch.cat([v] * n_entries, 0)
    return denoising_net(sd, rep(x), audio_in, person_in, rep(style_feat), rep(sd["start_motion_feat"].expand(B, -1, -1)),
                         rep(sd["start_audio_feat"].expand(B, -1, -1)), torch.full((B * n_entries,), int(t)), rep(indicator))


@torch.no_grad()
def sample(sd, sched, audio_feat, shape_feat, style_feat, motion_at_T, z_list, indicator, cfg_scale=1.15, n_motions=100):
    """oracle.diffusion.sample for the reference's default inference configuration (model.py:283-440: incremental CFG
    over [audio, style] -> entries [null, audio, audio + style], target = 'sample', start tokens as the previous window,
    flexibility 0) on torch ops; `z_list[t]` is the draw used at step t (t = T .. 2; step 1 uses zeros).  The head of
    entry 0 is accumulated in place as in the reference (model.py:407-415); in incremental mode entry 0 is only read
    before its first update, so this equals the textbook sum."""
    T = sched["betas"].shape[0] - 1
    x = motion_at_T.clone()
    f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
    scales = [cfg_scale, cfg_scale] if not isinstance(cfg_scale, (list, tuple)) else list(cfg_scale)
    for t in range(T, 0, -1):
        z = z_list[t].float() if t > 1 else torch.zeros_like(x)
        alpha, ab, abp = f32(sched["alphas"][t]), f32(sched["alpha_bars"][t]), f32(sched["alpha_bars"][t - 1])
        sigma = f32(sched["sigmas_inflex"][t])
        res = denoise_step(sd, x, audio_feat, shape_feat, style_feat, t, indicator, n_entries=3)
        e = [r.clone() for r in res.chunk(3, dim=0)]
        theta = e[0][:, -n_motions:]                      # a VIEW of entry 0, accumulated in place
        theta += f32(scales[0]) * (e[1][:, -n_motions:] - e[0][:, -n_motions:])
        theta += f32(scales[1]) * (e[2][:, -n_motions:] - e[1][:, -n_motions:])
        c0 = (1 - abp) * torch.sqrt(alpha) / (1 - ab)
        c1 = (1 - alpha) * torch.sqrt(abp) / (1 - ab)
        x = c0 * x + c1 * theta + sigma * z
    return x


class FlameTorch:
    """oracle.flame.FlameOracle buffers as torch tensors; forward = vertices of utils/flame.py:180-244 (pose2rot)."""

    def __init__(self, fo):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        self.v_template, self.shapedirs, self.posedirs = t(fo.v_template), t(fo.shapedirs), t(fo.posedirs)
        self.J_regressor, self.lbs_weights = t(fo.J_regressor), t(fo.lbs_weights)
        self.parents = [int(p) for p in fo.parents]

    @staticmethod
    def rodrigues(r):
        angle = (r + 1e-8).norm(dim=1, keepdim=True)
        d = r / angle
        cos, sin = angle.cos()[:, None], angle.sin()[:, None]
        rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
        z = torch.zeros_like(rx)
        K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(-1, 3, 3)
        return torch.eye(3)[None] + sin * K + (1 - cos) * torch.bmm(K, K)

    @torch.no_grad()
    def forward(self, shape, exp, pose):
        return self.forward_grad(shape, exp, pose)

    def forward_grad(self, shape, exp, pose):
        """Same arithmetic with autograd left on: the oracle of the differentiable FLAME pass (the reference trains
        through utils/flame.py / utils/lbs.py with torch autograd, training_script.py:167-176)."""
        B = shape.shape[0]
        V = self.v_template.shape[0]
        betas = torch.cat([shape, exp], dim=1)
        full_pose = torch.cat([pose[:, :3], torch.zeros(B, 3), pose[:, 3:], torch.zeros(B, 6)], dim=1)
        v_shaped = self.v_template[None] + torch.einsum("bl,mkl->bmk", betas, self.shapedirs)
        J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)
        rot = self.rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
        pose_feature = (rot[:, 1:] - torch.eye(3)).reshape(B, -1)
        v_posed = torch.matmul(pose_feature, self.posedirs).view(B, V, 3) + v_shaped
        rel = J.clone()
        rel[:, 1:] -= J[:, self.parents[1:]]
        tm = torch.zeros(B, J.shape[1], 4, 4)
        tm[..., :3, :3], tm[..., :3, 3], tm[..., 3, 3] = rot, rel, 1.0
        chain = [tm[:, 0]]
        for i in range(1, J.shape[1]):
            chain.append(torch.matmul(chain[self.parents[i]], tm[:, i]))
        tr = torch.stack(chain, dim=1)
        jh = torch.cat([J, torch.zeros(B, J.shape[1], 1)], dim=2)[..., None]
        A = tr.clone()
        A[..., 3:4] -= torch.matmul(tr, jh)
        T = torch.matmul(self.lbs_weights[None], A.view(B, -1, 16)).view(B, V, 4, 4)
        vh = torch.cat([v_posed, torch.ones(B, V, 1)], dim=2)[..., None]
        return torch.matmul(T, vh)[:, :, :3, 0]


def vertex_space_loss(args, is_starting_sample, shape_coef, motion_coef_gt, target, prev_motion_coef, coef_stats, flame,
                      end_idx=None):
    """Oracle restatement of reference utils/common.py:456-620 (target='sample', rot_repr='aa', legacy 54-d motion) with
    torch autograd: dict of the eight terms (each already / 2 as the reference returns them; head_trans None for the
    first window).  `flame` is a FlameTorch; coef_stats a dict of torch tensors (utils/common.py:140-173 de-normalises
    exp / pose / shape with them and zeroes the global rotation)."""
    l2 = args.criterion.lower() == "l2"
    crit = (lambda a, b: (a - b) ** 2) if l2 else (lambda a, b: (a - b).abs())
    P = args.n_prev_motions
    if is_starting_sample:
        target = target[:, P:]
    else:
        motion_coef_gt = torch.cat([prev_motion_coef, motion_coef_gt], dim=1)
        if args.no_constrain_prev:
            target = torch.cat([prev_motion_coef, target[:, P:]], dim=1)
    N, T = target.shape[:2]

    def coef_dict(m):
        exp = m[..., :50] * coef_stats["exp_std"] + coef_stats["exp_mean"]
        pose = torch.cat([torch.zeros_like(m[..., :3]), m[..., -1:], torch.zeros_like(m[..., :2])], dim=-1)
        pose = pose * coef_stats["pose_std"] + coef_stats["pose_mean"]
        pose = torch.cat([torch.zeros_like(pose[..., :3]), pose[..., 3:]], dim=-1)
        shp = shape_coef[:, None].expand(-1, m.shape[1], -1) * coef_stats["shape_std"] + coef_stats["shape_mean"]
        return shp.reshape(-1, 100), exp.reshape(-1, 50), pose.reshape(-1, 6)
    vg = flame.forward_grad(*coef_dict(motion_coef_gt)).view(N, T, 5023, 3)
    vp = flame.forward_grad(*coef_dict(target)).view(N, T, 5023, 3)
    if end_idx is None:
        mask = torch.ones(N, args.n_motions, dtype=torch.bool)
    else:
        mask = torch.arange(args.n_motions).expand(N, -1) < end_idx.unsqueeze(1)
    if not is_starting_sample:
        lead = torch.zeros_like if args.no_constrain_prev else torch.ones_like
        mask = torch.cat([lead(mask[:, :P]), mask], dim=1)
    d1 = lambda x: x[:, 1:] - x[:, :-1]
    out = {"noise": crit(motion_coef_gt, target)[mask].mean() / 2,
           "vert": crit(vg, vp)[mask].mean() / 2,
           "vel": crit(d1(vg), d1(vp))[mask[:, 1:]].mean() / 2,
           "smooth": crit(d1(vp)[:, 1:], d1(vp)[:, :-1])[mask[:, 2:]].mean() / 2}
    hg, hp = motion_coef_gt[:, :, 50:53], target[:, :, 50:53]
    out["head_angle"] = crit(hg, hp)[mask].mean() / 2
    out["head_vel"] = crit(d1(hg), d1(hp))[mask[:, 1:]].mean() / 2
    out["head_smooth"] = crit(d1(hp)[:, 1:], d1(hp)[:, :-1])[mask[:, 2:]].mean() / 2
    out["head_trans"] = None
    if not is_starting_sample and args.l_head_trans > 0:
        seq = torch.cat([hg[:, P - 3:P], hp[:, P:P + 3]], dim=1)
        v = d1(seq)
        a = d1(v)
        out["head_trans"] = crit(v[:, 2:4], v[:, 1:3])[mask[:, P:P + 2]].mean() + crit(a[:, 1:], a[:, :-1])[mask[:, P:P + 3]].mean()
    return out
