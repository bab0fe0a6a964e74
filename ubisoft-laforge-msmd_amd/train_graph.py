"""Differentiable (training) forward of the audio -> motion path, assembled from the HIP autograd blocks.

Mirrors the inference graph of model.py / utils/wav2vec2.py / style_encoder.py layer for layer, but every dense
op is an `autograd.Function` whose forward and backward are HIP kernels (autograd.py).  What is NOT a HIP kernel
here, on purpose, for round 1 of the training row: tensor concatenation / slicing / dtype casts / residual adds
and the final static-basis mix on (N, 110, 67)-sized tensors run as PyTorch autograd ops (plumbing-sized data).

Semantics: with `autograd.TrainNoise.active` False this is eval-mode arithmetic, and gradient parity is asserted
against the reference's autograd in eval mode (tests/golden g6_train).  With it True (Trainer: model.training, the
reference trains under model.train(), training_script.py:55) the stochastic regularisers are applied where the
reference's modules apply them: nn.Dropout after every projection / activation / residual branch and on the
attention probabilities (Philox masks regenerated in the backward, never stored), HF LayerDrop, and SpecAugment
time masks drawn by the reference's own index routine (bit-identical for a given numpy seed).  The frozen conv
feature extractor runs the no-grad inference kernels (reference model.py:97: `_freeze_parameters`).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import autograd as ag
from . import ops
from .utils.model_common import pad_audio_plan
from .utils.wav2vec2 import compute_mask_indices, compute_mask_indices_hf
from .utils.wav2vec2 import CONV_KERNEL, CONV_STRIDE  # noqa: F401


USE_CONV3_FN = os.environ.get("MSMD_CONV3_FN", "1") != "0"
USE_LAYERDROP_FN = os.environ.get("MSMD_LAYERDROP_FN", "1") != "0"


def _p(tree, name):
    return tree.get(name) if hasattr(tree, "get") else getattr(tree, name)


# ----------------------------------------------------------------------------- positional grouped conv
class PosConvFn(torch.autograd.Function):
    """z = grouped Conv1d(k=128, pad=64, groups=16)(h)[:, :T] + bias on a channels-last (B, T, 768) tensor.
    Forward: G windowed GEMMs over the zero-padded group-major copy.  Backward: data gradient = the same windowed
    GEMM on the zero-padded upstream gradient with kernel-flipped, (ci <-> co)-swapped weights; weight gradient =
    dZ_g^T . unfold_g (transposed unfold + batched GEMM); bias gradient = column sum."""

    @staticmethod
    def forward(ctx, h, w, bias):
        B, T, d = h.shape
        G = 16
        cg = d // G
        kpos = w.shape[2]
        dt = h.dtype
        wp = w.detach().reshape(G, cg, cg, kpos).permute(0, 1, 3, 2).reshape(G, cg, kpos * cg).to(dt).contiguous()
        xp = ops.group_pad(h.contiguous(), G, kpos // 2)
        Tp = T + kpos
        z = torch.empty_like(h)
        ops.gemm(xp, wp, bias.detach().float().contiguous(), None, ops.ACT_NONE, out=z, M=B * T, N=cg, K=kpos * cg,
                 lda=cg, rows_per_batch=T, a_batch_stride=G * Tp * cg, ldw=kpos * cg, ldc=d, batch=G, strideA=Tp * cg,
                 strideW=cg * kpos * cg, strideC=cg, strideBias=cg)
        ctx.save_for_backward(xp, w)
        ctx.dims = (B, T, d, G, cg, kpos)
        return z

    @staticmethod
    def backward(ctx, dz):
        xp, w = ctx.saved_tensors
        B, T, d, G, cg, kpos = ctx.dims
        dt = dz.dtype
        dz = dz.contiguous()
        dh = dw = db = None
        if ctx.needs_input_grad[0]:
            # dx[s] = sum_{kk', co} dzp[s + 65 + kk'][co] * w[co][ci][127 - kk'],  dzp = dz padded by 128 each side
            w2 = w.detach().reshape(G, cg, cg, kpos).flip(3).permute(0, 2, 3, 1).reshape(G, cg, kpos * cg).to(dt).contiguous()
            dzp = ops.group_pad(dz, G, kpos)  # (B, G, T + 2*kpos, cg)
            Tp2 = T + 2 * kpos
            dh = torch.empty(B, T, d, device=dz.device, dtype=dt)
            a_view = dzp.reshape(-1)[(kpos // 2 + 1) * cg:]
            ops.gemm(a_view, w2, None, None, ops.ACT_NONE, out=dh, M=B * T, N=cg, K=kpos * cg, lda=cg, rows_per_batch=T,
                     a_batch_stride=G * Tp2 * cg, ldw=kpos * cg, ldc=d, batch=G, strideA=Tp2 * cg,
                     strideW=cg * kpos * cg, strideC=cg)
        if ctx.needs_input_grad[1]:
            Tp = T + kpos
            if dt == torch.bfloat16:
                # dW_g = dZ_g^T . windows_g(xp): the overlapping conv windows are read in place by the TN GEMM
                dwp = ops.gemm_tn(dz, xp, M=B * T, N=cg, K=kpos * cg, lda=d, ldb=cg, batch=G, strideA=cg,
                                  strideB=Tp * cg, b_rows_per_window=T, b_window_stride=G * Tp * cg)
            else:
                unf = ops.unfold_t(xp, T, kpos)                     # (G, kpos*cg, Mp)
                Mp = unf.shape[-1]
                dzT = torch.zeros(G, cg, Mp, device=dz.device, dtype=dt)
                # dz (B*T, G*cg) -> per group (cg, B*T)
                ops.transpose(dz, dzT, B * T, cg, d, Mp, G, cg, cg * Mp)
                dwp = torch.empty(G, cg, kpos * cg, device=dz.device, dtype=torch.float32)
                ops.gemm(dzT, unf, None, None, ops.ACT_NONE, out=dwp, M=cg, N=kpos * cg, K=Mp, lda=Mp, ldw=Mp,
                         ldc=kpos * cg, batch=G, strideA=cg * Mp, strideW=kpos * cg * Mp, strideC=cg * kpos * cg)
            dw = dwp.reshape(G, cg, kpos, cg).permute(0, 1, 3, 2).reshape(w.shape)  # (co, ci, kk)
        if ctx.needs_input_grad[2]:
            db = ops.colsum(dz.reshape(B * T, d))
        return dh, dw, db


class FoldWeightNormFn(torch.autograd.Function):
    """weight_norm(dim=2) of the positional conv: w[o, i, k] = g[k] v[o, i, k] / n[k],  n[k] = ||v[:, :, k]|| over (o, i).
    Forward and backward use elementwise tensor ops and msmd_colsum for the two reductions over the 36 864 (o, i)
    pairs.  (It used to be `v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())` on the host library's reduction:
    inside a SEGMENTED hipGraph -- the data-parallel mode -- that multi-block reduction returned NaN for window 0 on every
    second replay (params finite, window 1 fine; tools/dp_sidestream_check.py now fails on non-finite losses), so the
    training graph keeps no multi-block host-library reduction.)
      d g[k] = sum_{o,i} dw v / n,      d v = (g / n) dw - v (g / n^3) sum_{o,i} dw v"""

    @staticmethod
    def forward(ctx, g, v):
        O, I, K = v.shape
        v2 = v.detach().reshape(O * I, K).float()
        inv = torch.rsqrt(ops.colsum((v2 * v2).contiguous()))          # 1 / n  (K)
        s = g.detach().reshape(K).float() * inv                         # g / n
        ctx.save_for_backward(v, s, inv)
        ctx.g_shape = g.shape
        return (v.detach().float() * s.view(1, 1, K)).to(v.dtype)

    @staticmethod
    def backward(ctx, dw):
        v, s, inv = ctx.saved_tensors
        O, I, K = v.shape
        vf, dwf = v.detach().float(), dw.float()
        dwv = ops.colsum((dwf * vf).reshape(O * I, K).contiguous())     # sum_{o,i} dw v  (K)
        dg = (dwv * inv).reshape(ctx.g_shape)
        dv = dwf * s.view(1, 1, K) - vf * (s * dwv * inv * inv).view(1, 1, K)
        return dg.to(v.dtype), dv.to(v.dtype)


def _fold_weight_norm(g, v):
    """weight_norm(dim=2): w = g * v / ||v||_{dims 0,1} on the (768, 48, 128) parameter (CPU tensors: plain autograd ops)."""
    if not v.is_cuda:
        return v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
    return FoldWeightNormFn.apply(g, v)


# ----------------------------------------------------------------------------- audio encoder
_ZERO = {}


class LayerDropSelectFn(torch.autograd.Function):
    """Graph-safe LayerDrop: out = flag ? h_in : h for a 0-dim device flag.  torch.where's own backward builds a scalar-zero
    tensor per operand (a fill launch each) next to its two selects; here the zero is made once per (device, dtype)."""

    @staticmethod
    def forward(ctx, flag, h_in, h):
        ctx.save_for_backward(flag)
        return torch.where(flag, h_in, h)

    @staticmethod
    def backward(ctx, g):
        (flag,) = ctx.saved_tensors
        key = (g.device, g.dtype)
        z = _ZERO.get(key)
        if z is None:
            z = _ZERO[key] = torch.zeros((), device=g.device, dtype=g.dtype)
        return None, torch.where(flag, g, z), torch.where(flag, z, g)


class NullTokenSelectFn(torch.autograd.Function):
    """Classifier-free-guidance masking of reference model.py:205-218: out[b] = mask[b] ? token : x[b] for a (1, 1, C) learned null
    token and x (B, T, C).  Written as torch.where on the expanded token, autograd sums the token's gradient over (B, T) with
    the host library's reduction: 800 x 512 for null_audio_feat, which that library runs as a MULTI-BLOCK reduction (partial
    sums in a staging buffer, a semaphore word zeroed by a memset node in front of the kernel).  Inside the hipGraphs of the
    training step that launch returned a wrong sum on the first replay after another graph of the shared pool had run (round
    4: tools/dp_rccl_debug.py -- the eager backward and every further replay agreed; only this gradient was off, by O(1)),
    the same family as round 3's NaN weight-norm reduction.  The token's gradient is a column sum of the masked rows by
    msmd_colsum here (deterministic, one launch, its workspace written before it is read)."""

    @staticmethod
    def forward(ctx, mask, token, x):
        """mask: bool (B,) [per sequence: the CFG tokens] or (B, T) [per frame: SpecAugment's masked_spec_embed,
        utils/wav2vec2.py:99-105]; token: (C,) or (1, 1, C), any float dtype; x: (B, T, C)."""
        m = mask.reshape(mask.shape + (1,) * (x.ndim - mask.ndim))
        ctx.save_for_backward(m)
        ctx.token_shape, ctx.token_dtype = token.shape, token.dtype
        return torch.where(m, token.reshape(-1).to(x.dtype), x)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        mf = m.to(g.dtype)
        gm = (g * mf).reshape(-1, g.shape[-1]).contiguous()
        if gm.is_cuda and gm.dtype in (torch.float32, torch.bfloat16):
            d_token = ops.colsum(gm)
        else:
            d_token = gm.float().sum(0)
        return None, d_token.reshape(ctx.token_shape).to(ctx.token_dtype), g * (1 - mf)


def audio_encoder_train(enc, audio, output_fps, frame_num, dtype, groups=1):
    """Differentiable counterpart of Wav2Vec2Model.encode (utils/wav2vec2.py): (B, L) audio -> (B, frame_num, 768).
    groups > 1: the batch is `groups` equal blocks of rows that the reference would have encoded in separate calls (the two
    windows of a training iteration): SpecAugment consumes one injected mask per block and LayerDrop draws one coin per
    block and layer, as separate calls would."""
    c = enc.config
    with torch.no_grad():
        r, rep = pad_audio_plan(audio.shape[1])
        x = enc.feature_extractor_cl(audio, dtype, r, rep)
        T50 = x.shape[1]
        crop = min(round(frame_num * 50 / output_fps), T50)
        if not (crop == T50 and frame_num == T50):
            x = ops.interp_linear(x, frame_num, crop)
    g = lambda n: enc.get_parameter(n)
    noise = ag.TrainNoise
    h = ag.layer_norm(x, g("feature_projection.layer_norm.weight"), g("feature_projection.layer_norm.bias"))
    h = ag.linear_dropout(h, g("feature_projection.projection.weight"), g("feature_projection.projection.bias"),
                          c.feat_proj_dropout)
    if noise.active and c.apply_spec_augment and c.mask_time_prob > 0:
        # SpecAugment (utils/wav2vec2.py:99-105 / HF _mask_hidden_states): masked frames <- masked_spec_embed
        m = noise.next_spec_mask()
        if m is not None and groups > 1 and m.shape[0] * groups == h.shape[0]:
            m = torch.cat([m] + [noise.next_spec_mask() for _ in range(groups - 1)], 0)
        if m is None:
            fn = compute_mask_indices_hf if enc.model_type == "hubert" else compute_mask_indices
            m = torch.from_numpy(fn((h.shape[0], h.shape[1]), c.mask_time_prob, c.mask_time_length,
                                    c.mask_time_min_masks, noise.host_rng)).to(h.device)
        h = NullTokenSelectFn.apply(m, g("masked_spec_embed"), h)
    w = _fold_weight_norm(g("encoder.pos_conv_embed.conv.weight_g"), g("encoder.pos_conv_embed.conv.weight_v"))
    z = PosConvFn.apply(h, w, g("encoder.pos_conv_embed.conv.bias"))
    h = h + ag_act(z, ops.ACT_GELU)
    stable = bool(getattr(c, "do_stable_layer_norm", False))   # pre-LN blocks + one final LayerNorm (large checkpoints)
    if not stable:
        h = ag.layer_norm(h, g("encoder.layer_norm.weight"), g("encoder.layer_norm.bias"))
    h = ag.dropout(h, c.hidden_dropout)
    d, H = c.hidden_size, c.num_attention_heads
    for n in range(c.num_hidden_layers):
        # LayerDrop (HF Wav2Vec2Encoder): skip the layer with probability layerdrop.  Eager mode skips for real (host
        # draw, as HF's torch.rand([])); graph-safe mode computes the layer and selects on a device-side draw.
        skip_flag = None
        if noise.active and c.layerdrop > 0:
            if noise.graph_safe:
                skip_flag = torch.rand((), device=h.device) < c.layerdrop if groups == 1 else \
                    (torch.rand((groups, 1, 1), device=h.device) < c.layerdrop).repeat_interleave(h.shape[0] // groups, 0)
            else:
                rng = noise.host_rng if noise.host_rng is not None else np.random
                coins = [rng.rand() < c.layerdrop for _ in range(groups)]
                if all(coins):
                    continue
                if any(coins):      # some blocks skip this layer: compute it and select per block
                    skip_flag = torch.tensor(coins, device=h.device).view(groups, 1, 1).repeat_interleave(h.shape[0] // groups, 0)
        h_in = h
        p = f"encoder.layers.{n}."
        wq, wk, wv = (g(p + f"attention.{k}_proj.weight") for k in "qkv")
        bq, bk, bv = (g(p + f"attention.{k}_proj.bias") for k in "qkv")
        ln1 = (g(p + "layer_norm.weight"), g(p + "layer_norm.bias"))
        ln2 = (g(p + "final_layer_norm.weight"), g(p + "final_layer_norm.bias"))
        fa = ag.FUSED.get(("enc_qkv", id(enc), n)) if ag.DIRECT_GRAD else None
        J = ag.Junction()       # h feeds the QKV projection and is the out-projection's residual (post-LN branch)
        if fa is not None:      # Q | K | V sit next to each other in the arenas: the fused operand is a view
            qkv_proj = lambda t, j=None: ag.linear_alias(t, fa, junction_in=j)
        else:
            wqkv, bqkv = torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0)
            qkv_proj = lambda t, j=None: ag.linear(t, wqkv, bqkv, junction_in=j)
        # the attention launch also pulls the layer's remaining weight casts (and the next layer's QKV) through the
        # memory-side cache: the GEMMs behind it would otherwise read them from HBM inside their K loops
        nfa = ag.FUSED.get(("enc_qkv", id(enc), n + 1)) if ag.DIRECT_GRAD else None
        pf = tuple(t for t in (ag.cast_of(g(p + "attention.out_proj.weight")),
                               ag.cast_of(g(p + "feed_forward.intermediate_dense.weight")),
                               ag.cast_of(g(p + "feed_forward.output_dense.weight")),
                               ag.cast_of(nfa.w) if nfa is not None else None) if t is not None) or None
        if stable:   # HubertEncoderLayerStableLayerNorm
            a = ag.self_attention(qkv_proj(ag.layer_norm(h, *ln1, sole_consumer=False)), H, (d // H) ** -0.5,
                                  p_drop=c.attention_dropout, prefetch=pf)
            h = ag.linear_dropout(a, g(p + "attention.out_proj.weight"), g(p + "attention.out_proj.bias"),
                                  c.hidden_dropout, residual=h)
            h = ag.ffn(ag.layer_norm(h, *ln2, sole_consumer=False), g(p + "feed_forward.intermediate_dense.weight"),
                       g(p + "feed_forward.intermediate_dense.bias"), g(p + "feed_forward.output_dense.weight"),
                       g(p + "feed_forward.output_dense.bias"), c.activation_dropout, c.hidden_dropout, residual=h)
        else:
            a = ag.self_attention(qkv_proj(h, J), H, (d // H) ** -0.5, p_drop=c.attention_dropout, prefetch=pf)
            h = ag.layer_norm(ag.linear_dropout(a, g(p + "attention.out_proj.weight"),
                                                g(p + "attention.out_proj.bias"), c.hidden_dropout, residual=h,
                                                junction_out=J), *ln1)
            # feed-forward block as one autograd node: its backward's middle (linear2's data gradient + dropout + GELU
            # backward) is one launch
            h = ag.layer_norm(ag.ffn(h, g(p + "feed_forward.intermediate_dense.weight"),
                                     g(p + "feed_forward.intermediate_dense.bias"),
                                     g(p + "feed_forward.output_dense.weight"), g(p + "feed_forward.output_dense.bias"),
                                     c.activation_dropout, c.hidden_dropout, residual=h), *ln2)
        if skip_flag is not None:
            h = LayerDropSelectFn.apply(skip_flag, h_in, h) if USE_LAYERDROP_FN else torch.where(skip_flag, h_in, h)
    if stable:
        h = ag.layer_norm(h, g("encoder.layer_norm.weight"), g("encoder.layer_norm.bias"))
    return h


# ----------------------------------------------------------------------------- fused operands as arena views
def adjacent_parameter_groups(model):
    """dp.ADJACENT for this model: per encoder layer [q.weight, k.weight, v.weight] and [q.bias, k.bias, v.bias]."""
    enc = model.audio_encoder
    groups = []
    for n in range(enc.config.num_hidden_layers):
        p = f"encoder.layers.{n}.attention."
        for kind in ("weight", "bias"):
            try:
                groups.append([enc.get_parameter(p + f"{x}_proj.{kind}") for x in "qkv"])
            except AttributeError:
                pass
    return groups


def build_fused_aliases(model, flat_param, grad_arena):
    """autograd.FUSED for this model (the Trainer calls this once both arenas exist): encoder layers' fused QKV operand;
    the decoder layers' cross-attention Q and KV rows of in_proj_weight / in_proj_bias.  Returns (aliases, parameters
    that are only used through an alias)."""
    base = flat_param.data_ptr()

    def views(p, r0=None, r1=None, rows=None, cols=None):
        """(arena view, gradient-arena view) of rows [r0, r1) of parameter p, or of `rows` x `cols` starting at p."""
        off = (p.data_ptr() - base) // 4
        if rows is None:
            inner = p[0].numel() if p.dim() > 1 else 1
            off += r0 * inner
            shape = (r1 - r0,) + tuple(p.shape[1:])
        else:
            shape = (rows, cols) if cols else (rows,)
        n = 1
        for s_ in shape:
            n *= s_
        return flat_param[off:off + n].view(shape), grad_arena[off:off + n].view(shape)

    aliases, only = [], []
    enc = model.audio_encoder
    d = enc.config.hidden_size
    for n in range(enc.config.num_hidden_layers):
        p = f"encoder.layers.{n}.attention."
        try:
            ws = [enc.get_parameter(p + f"{x}_proj.weight") for x in "qkv"]
            bs = [enc.get_parameter(p + f"{x}_proj.bias") for x in "qkv"]
        except AttributeError:
            continue
        if not all(t.requires_grad for t in ws + bs):
            continue
        adjacent = all(ws[i + 1].data_ptr() == ws[i].data_ptr() + 4 * ws[i].numel() for i in range(2)) and \
            all(bs[i + 1].data_ptr() == bs[i].data_ptr() + 4 * bs[i].numel() for i in range(2))
        if not adjacent:
            continue
        w, wg = views(ws[0], rows=3 * d, cols=d)
        b, bg = views(bs[0], rows=3 * d)
        fa = ag.FusedAlias(w, b, wg, bg, ws + bs)
        ag.FUSED[("enc_qkv", id(enc), n)] = fa
        aliases.append(fa)
        only += ws
    net = model.denoising_net
    dd = net.feature_dim
    for n in range(net.n_layers):
        p = f"transformer.layers.{n}.multihead_attn."
        wp, bp = net.get_parameter(p + "in_proj_weight"), net.get_parameter(p + "in_proj_bias")
        if not (wp.requires_grad and bp.requires_grad):
            continue
        for kind, r0, r1 in (("dec_q", 0, dd), ("dec_kv", dd, 3 * dd)):
            w, wg = views(wp, r0, r1)
            b, bg = views(bp, r0, r1)
            fa = ag.FusedAlias(w, b, wg, bg, [wp, bp])
            ag.FUSED[(kind, id(net), n)] = fa
            aliases.append(fa)
        only.append(wp)
    return aliases, only


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, act):
        z = z.contiguous()
        ctx.save_for_backward(z)
        ctx.act = act
        return ops.act_fwd(z, act)

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), z, ctx.act), None


def ag_act(z, act):
    return ActFn.apply(z, act)


def audio_feat_train(model, audio, frame_num, dtype, groups=1):
    """reference model.py:250-264 (differentiable): encoder at 2L -> pairwise mean -> audio_feature_map."""
    h = audio_encoder_train(model.audio_encoder, audio, model.fps, frame_num * 2, dtype, groups)
    h = 0.5 * (h[:, 0::2] + h[:, 1::2])  # exact 2:1 linear resample
    return ag.linear(h.contiguous(), model.audio_feature_map.weight, model.audio_feature_map.bias)


# ----------------------------------------------------------------------------- denoiser
def denoiser_train(net, motion_noisy, audio_feat, person_feat, static_style_feat, prev_motion_feat, prev_audio_feat,
                   step, indicator, dtype):
    """reference model.py:914-996 (differentiable).  Returns (N, Lp + L, dm) fp32."""
    g = lambda n: net.get_parameter(n)
    d, H, nb, dm = net.feature_dim, net.n_heads, net.num_of_basis, net.motion_feat_dim
    N = person_feat.shape[0]
    step = torch.as_tensor(step, device=net.device, dtype=torch.long)
    te = net.TE.pe[0].float()[step].to(dtype)   # the sinusoidal step table is a buffer: no weight pack needed here
    emb = ag.linear(ag.linear(te, g("diff_step_map.0.weight"), g("diff_step_map.0.bias"), act=ops.ACT_GELU),
                    g("diff_step_map.2.weight"), g("diff_step_map.2.bias"))
    person = ag.linear(person_feat.reshape(N, -1).to(dtype), g("person_proj.weight"), g("person_proj.bias")) + emb
    feats = torch.cat([prev_motion_feat, motion_noisy], dim=1)
    if net.use_indicator:
        ind = torch.cat([torch.zeros(N, net.n_prev_motions, device=feats.device), indicator.float()], dim=1)
        feats = torch.cat([feats, ind.unsqueeze(-1)], dim=-1)
    x = ag.linear(feats.to(dtype), g("feature_proj.weight"), g("feature_proj.bias"))
    x = torch.cat([person.unsqueeze(1), x], dim=1)
    if net.use_learnable_pe:
        x = x + g("PE").to(dtype)
    else:   # sinusoidal: the single row pe[seq_len] on every position, then dropout 0.1 (utils/model_common.py:99-101)
        x = ag.dropout(x + net.PE.pe[0, x.shape[1]].to(dtype), net.PE.p_drop)
    mem = torch.cat([prev_audio_feat.to(dtype), audio_feat.to(dtype)], dim=1)
    scale = (d // H) ** -0.5
    mask = net.alignment_mask
    pd = 0.1  # nn.TransformerDecoderLayer default dropout (model.py:874-877 passes none): dropout1-3, FFN, attention
    for n in range(net.n_layers):
        p = f"transformer.layers.{n}."
        J1, J2 = ag.Junction(), ag.Junction()   # x -> (QKV | Q projection, the block's residual): one gradient, no add
        qkv = ag.linear(x, g(p + "self_attn.in_proj_weight"), g(p + "self_attn.in_proj_bias"), junction_in=J1)
        a = ag.self_attention(qkv, H, scale, p_drop=pd)
        x = ag.layer_norm(ag.linear_dropout(a, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"), pd,
                                            residual=x, junction_out=J1), g(p + "norm1.weight"), g(p + "norm1.bias"))
        fq = ag.FUSED.get(("dec_q", id(net), n)) if ag.DIRECT_GRAD else None
        if fq is not None:      # rows of in_proj_weight as arena views: no slice nodes, gradients straight into the arena
            q = ag.linear_alias(x, fq, junction_in=J2)
            kv = ag.linear_alias(mem, ag.FUSED[("dec_kv", id(net), n)])
        else:
            w, b = g(p + "multihead_attn.in_proj_weight"), g(p + "multihead_attn.in_proj_bias")
            q = ag.linear(x, w[:d], b[:d], junction_in=J2)
            kv = ag.linear(mem, w[d:], b[d:])
        cattn = ag.cross_attention(q, kv, H, scale, mask, p_drop=pd)
        x = ag.layer_norm(ag.linear_dropout(cattn, g(p + "multihead_attn.out_proj.weight"),
                                            g(p + "multihead_attn.out_proj.bias"), pd, residual=x, junction_out=J2),
                          g(p + "norm2.weight"), g(p + "norm2.bias"))
        x = ag.layer_norm(ag.ffn(x, g(p + "linear1.weight"), g(p + "linear1.bias"), g(p + "linear2.weight"),
                                 g(p + "linear2.bias"), pd, pd, residual=x),
                          g(p + "norm3.weight"), g(p + "norm3.bias"))
    dec = ag.linear(ag.linear(x[:, 1:].contiguous(), g("motion_dec.0.weight"), g("motion_dec.0.bias"),
                              act=ops.ACT_GELU), g("motion_dec.2.weight"), g("motion_dec.2.bias")).float()
    s = static_style_feat.reshape(N, -1).to(dtype)
    stat = torch.stack([ag.linear(ag.linear(s, g(f"static_feature_mapping.{b}.0.weight"),
                                            g(f"static_feature_mapping.{b}.0.bias"), act=ops.ACT_GELU),
                                  g(f"static_feature_mapping.{b}.2.weight"), g(f"static_feature_mapping.{b}.2.bias"))
                        for b in range(nb)], dim=1).float()                       # (N, nb, dm)
    dyn, alpha = dec[..., :dm], dec[..., dm:]
    if net.regularize_alpha == "sigmoid":        # model.py:973-974
        alpha = torch.sigmoid(alpha)
    if net.use_head_alpha:
        static = (stat[:, None] * alpha.unsqueeze(-1)).sum(dim=2)
    else:
        static = torch.cat([(stat[:, None, :, :-3] * alpha.unsqueeze(-1)).sum(dim=2),
                            stat[:, None, :, -3:].sum(dim=2).expand(-1, dec.shape[1], -1)], dim=-1)
    return dyn + static


# ----------------------------------------------------------------------------- style encoder
class Conv3Fn(torch.autograd.Function):
    """y = Conv1d(k=3, padding=1)(x) + b on a channels-last (B, T, C) tensor with the PACKED weight wp (Cout, 3 C), K index =
    tap * C + c (style_encoder.py:137-140 convs).  Forward: one zero-padded copy + ONE windowed GEMM (msmd_gemm reads the
    overlapping 3-frame windows in place: no (B, T, 3 C) im2col tensor).  Backward: data gradient = the same windowed GEMM on
    the padded upstream gradient with tap-flipped, (ci <-> co)-swapped weights; weight + bias gradient = one TN GEMM that reads
    the windows of the padded input in place.  Replaces pad + cat of three shifted views + Linear, whose backward alone was
    3 slice-backwards (a fill and a strided copy each), 2 accumulations and the pad's backward per conv."""

    @staticmethod
    def forward(ctx, x, wp, b):
        B, T, C = x.shape
        xp = ops.group_pad(x.contiguous(), 1, 1).reshape(B, T + 2, C)
        y = ops.conv1d_cl(xp, wp.detach().to(x.dtype).contiguous(), b.detach().float().contiguous(), kernel=3, stride=1)
        ctx.save_for_backward(xp, wp)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        B, Tp, C = xp.shape
        T, Co, dt = Tp - 2, wp.shape[0], dy.dtype
        dy = dy.contiguous()
        dx = dwp = db = None
        if ctx.needs_input_grad[0]:
            # dx[s] = sum_{k', co} dyp[s + k'][co] * wp[co][2 - k'][ci],  dyp = dy padded by one frame on each side
            w2 = wp.detach().reshape(Co, 3, C).flip(1).permute(2, 1, 0).reshape(C, 3 * Co).to(dt).contiguous()
            dyp = ops.group_pad(dy, 1, 1).reshape(B, T + 2, Co)
            dx = ops.conv1d_cl(dyp, w2, None, kernel=3, stride=1)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dwp, db = ops.gemm_tn(dy.reshape(B * T, Co), xp, want_colsum=True, M=B * T, N=Co, K=3 * C, lda=Co, ldb=C,
                                  b_rows_per_window=T, b_window_stride=(T + 2) * C)
        return dx, dwp, db


def _conv3(x, w, b, act):
    """Conv1d(k=3, padding=1) on channels-last x: Conv3Fn (windowed GEMMs) for the 16-bit 64-aligned widths, else a linear over
    the 3-frame window (torch.cat of shifted views)."""
    if (act == ops.ACT_NONE and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 64 == 0 and w.shape[0] % 8 == 0
            and b is not None and USE_CONV3_FN):
        return Conv3Fn.apply(x, w.permute(0, 2, 1).reshape(w.shape[0], -1), b)
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1))
    win = torch.cat([xp[:, :-2], xp[:, 1:-1], xp[:, 2:]], dim=-1)
    return ag.linear(win.contiguous(), w.permute(0, 2, 1).reshape(w.shape[0], -1), b, act=act)


def style_encoder_train(se, motion_coef, dtype):
    """reference style_encoder.py:178-199 (differentiable) -> (mu, logvar) fp32."""
    g = lambda n: se.get_parameter(n)
    B, T, _ = motion_coef.shape
    x = motion_coef.to(dtype)
    on = ag.TrainNoise.active

    def conv_drop_elu(x, name, p):
        """Conv1d -> Dropout(p) -> ELU (style_encoder.py:137-140): the activation stays fused in eval mode."""
        if not on:
            return _conv3(x, g(name + ".weight"), g(name + ".bias"), ops.ACT_ELU)
        return ag_act(ag.dropout(_conv3(x, g(name + ".weight"), g(name + ".bias"), ops.ACT_NONE), p), ops.ACT_ELU)

    x = ag.layer_norm(conv_drop_elu(x, "input_layers.1", 0.2), g("input_layers.5.weight"), g("input_layers.5.bias"))
    x = ag.layer_norm(conv_drop_elu(x, "input_layers.7", 0.2), g("input_layers.11.weight"), g("input_layers.11.bias"),
                      post_add=se.PE.pe[0, T].float().contiguous())
    x = ag.dropout(x, 0.1)                                   # PositionalEncoding dropout (utils/model_common.py:101)
    d = se.conv_feature_dim
    pd = 0.1                                                 # nn.TransformerEncoderLayer default dropout
    qkv = ag.linear(x, g("encoder.self_attn.in_proj_weight"), g("encoder.self_attn.in_proj_bias"))
    a = ag.self_attention(qkv, 8, 64 ** -0.5, p_drop=pd)
    x = ag.layer_norm(ag.linear_dropout(a, g("encoder.self_attn.out_proj.weight"), g("encoder.self_attn.out_proj.bias"),
                                        pd, residual=x), g("encoder.norm1.weight"), g("encoder.norm1.bias"))
    f = ag.linear_dropout(x, g("encoder.linear1.weight"), g("encoder.linear1.bias"), pd, act=ops.ACT_GELU)
    x = ag.layer_norm(ag.linear_dropout(f, g("encoder.linear2.weight"), g("encoder.linear2.bias"), pd, residual=x),
                      g("encoder.norm2.weight"), g("encoder.norm2.bias"))
    x = ag.layer_norm(conv_drop_elu(x, "output_layers.1", 0.1), g("output_layers.5.weight"), g("output_layers.5.bias"))
    x = _conv3(x, g("output_layers.7.weight"), g("output_layers.7.bias"), ops.ACT_NONE)
    out = x.float().mean(dim=1)
    h = se.output_size // 2
    return out[:, :h], out[:, h:]


# ----------------------------------------------------------------------------- MSMD.forward (training)
def msmd_forward_train(model, motion_feat, audio_or_feat, shape_feat, style_feat, prev_motion_feat, prev_audio_feat,
                       time_step, indicator, eps, null_style_mask=None, null_audio_mask=None):
    """reference model.py:146-248 with gradients; stochastic draws are passed in (time_step, eps, CFG masks)."""
    dtype = model.compute_dtype
    B = motion_feat.shape[0]
    if audio_or_feat.ndim == 2:
        audio_feat_saved = audio_feat_train(model, audio_or_feat, model.n_motions, dtype).float()
    else:
        audio_feat_saved = audio_or_feat
    audio_feat = audio_feat_saved
    if shape_feat.ndim == 2:
        shape_feat = shape_feat.unsqueeze(1)
    if style_feat.ndim == 2:
        style_feat = style_feat.unsqueeze(1)
    if prev_motion_feat is None:
        prev_motion_feat = model.start_motion_feat.expand(B, -1, -1)
    if prev_audio_feat is None:
        prev_audio_feat = model.start_audio_feat.expand(B, -1, -1)
    if null_style_mask is not None:
        style_feat = NullTokenSelectFn.apply(null_style_mask, model.null_style_feat, style_feat)
    if null_audio_mask is not None:
        audio_feat = NullTokenSelectFn.apply(null_audio_mask, model.null_audio_feat, audio_feat)
    person_feat = torch.cat([shape_feat, style_feat], dim=-1)
    ts = torch.as_tensor(time_step, device=model.device, dtype=torch.long)
    ab = model.diffusion_sched.alpha_bars[ts]
    noisy = torch.sqrt(ab).view(-1, 1, 1) * motion_feat + torch.sqrt(1 - ab).view(-1, 1, 1) * eps
    target = denoiser_train(model.denoising_net, noisy, audio_feat, person_feat, style_feat, prev_motion_feat,
                            prev_audio_feat, ts, indicator, dtype)
    return eps, target, motion_feat.detach(), audio_feat_saved.detach()


# ----------------------------------------------------------------------------- differentiable losses
def _masked_mean(v, mask):
    """mean over selected (n, t) rows of a (N, T[, C]) tensor (reference: loss[mask].mean()), written as a masked
    sum / count so that nothing synchronises with the host (no boolean gather; hipGraph-capturable).  An empty
    selection gives 0 with zero gradient (the reference would produce NaN there)."""
    w = mask.to(v.dtype)
    per_row = v.shape[-1] if v.ndim == 3 else 1
    if v.ndim == 3:
        w = w.unsqueeze(-1)
    return (v * w).sum() / (mask.sum().to(v.dtype) * per_row).clamp(min=1.0)


def loss_no_vert_train(args, is_starting_sample, motion_coef_gt, target, prev_motion_coef, end_idx=None, halve=True):
    """Differentiable restatement of reference utils/common.py:198-442 (target='sample', l2/l1) on (N, 110, 67)
    tensors with PyTorch autograd ops (plumbing-sized data; the forward-only HIP versions live in utils/common.py).
    Returns the reference's 7-tuple; halve=False returns the first six terms WITHOUT their final / 2 (the caller applies the
    factor where it weights the terms: Trainer._combine_losses), the seventh is never halved."""
    crit = (lambda a, b: (a - b) ** 2) if args.criterion.lower() == "l2" else (lambda a, b: (a - b).abs())
    n_prev = args.n_prev_motions
    if is_starting_sample:
        target = target[:, n_prev:]
    else:
        motion_coef_gt = torch.cat([prev_motion_coef, motion_coef_gt], dim=1)
        if getattr(args, "no_constrain_prev", False):     # reference utils/common.py:245-246
            target = torch.cat([prev_motion_coef, target[:, n_prev:]], dim=1)
    N = target.shape[0]
    if end_idx is None:
        mask = torch.ones((N, args.n_motions), dtype=torch.bool, device=target.device)
    else:
        mask = torch.arange(args.n_motions, device=target.device).expand(N, -1) < end_idx.unsqueeze(1)
    if not is_starting_sample:
        lead = torch.zeros_like if getattr(args, "no_constrain_prev", False) else torch.ones_like   # common.py:382-385
        mask = torch.cat([lead(mask[:, :n_prev]), mask], dim=1)
    d1 = lambda x: x[:, 1:] - x[:, :-1]
    gt, pr = motion_coef_gt, target
    C = pr.shape[-1]
    if pr.is_cuda:
        # every term is a masked mean of crit(D^k gt, D^k pred) over a column range: six msmd_masked_seq_loss launches
        # forward, six backward launches adding into ONE gradient buffer (instead of ~400 elementwise launches fwd + bwd)
        prefix = 0 if is_starting_sample else (-n_prev if getattr(args, "no_constrain_prev", False) else n_prev)
        specs = ((0, 0, 0, C), (1, 0, 0, C - 3), (1, 0, C - 3, C), (2, 1, 0, C - 3), (2, 1, C - 3, C), (0, 0, C - 3, C))
        l_noise, vel_a, vel_b, sm_a, sm_b, l_head = SeqLossTermsFn.apply(
            pr, gt, end_idx, prefix, 0 if args.criterion.lower() == "l2" else 1, specs)
        loss_noise, loss_vel, loss_smooth = l_noise, vel_a + vel_b, sm_a + sm_b
        loss_head_angle, loss_head_vel, loss_head_smooth = l_head, vel_b, sm_b
        hg, hp = gt[..., -3:], pr[..., -3:]
    else:
        loss_noise = _masked_mean(crit(gt, pr), mask)
        vg, vp = d1(gt), d1(pr)
        loss_vel = _masked_mean(crit(vg[..., :-3], vp[..., :-3]).mean(-1) + crit(vg[..., -3:], vp[..., -3:]).mean(-1),
                                mask[:, 1:])
        sp = d1(vp)
        loss_smooth = _masked_mean(crit(sp[..., :-3], 0 * sp[..., :-3]).mean(-1) + crit(sp[..., -3:], 0 * sp[..., -3:]).mean(-1),
                                   mask[:, 2:])
        hg, hp = gt[..., -3:], pr[..., -3:]
        loss_head_angle = _masked_mean(crit(hg, hp), mask)
        loss_head_vel = _masked_mean(crit(d1(hg), d1(hp)).mean(-1), mask[:, 1:])
        hs = d1(d1(hp))
        loss_head_smooth = _masked_mean(crit(hs, 0 * hs).mean(-1), mask[:, 2:])
    loss_head_trans = None
    if not is_starting_sample and args.l_head_trans > 0:
        seq = torch.cat([hg[:, n_prev - 3:n_prev], hp[:, n_prev:n_prev + 3]], dim=1)
        v = d1(seq)
        a = d1(v)
        loss_head_trans = (crit(v[:, 2:4], v[:, 1:3]).mean(-1).mean(-1) + crit(a[:, 1:], a[:, :-1]).mean(-1).mean(-1)).mean()
    if not halve:
        return (loss_noise, loss_vel, loss_smooth, loss_head_angle, loss_head_vel, loss_head_smooth, loss_head_trans)
    return (loss_noise / 2, loss_vel / 2, loss_smooth / 2, loss_head_angle / 2, loss_head_vel / 2, loss_head_smooth / 2,
            loss_head_trans)


class SeqLossTermsFn(torch.autograd.Function):
    """A tuple of masked sequence-loss terms on one (N, T, C) prediction: specs = ((order, mode, c_lo, c_hi), ...), each
    one msmd_masked_seq_loss launch (mean over valid frames of the mean over columns [c_lo, c_hi) of crit on the
    order-th temporal difference; mode 1 compares the prediction's difference with 0).  The backward ADDS every term's
    gradient into one (N, T, C) buffer (msmd_masked_seq_loss_bwd).  Reference: utils/common.py:198-442."""

    @staticmethod
    def forward(ctx, pred, gt, end_idx, prefix, crit, specs):
        pred, gt = pred.float().contiguous(), gt.float().contiguous()
        e32 = end_idx.to(torch.int32).contiguous() if end_idx is not None else None
        outs, wss = [], []
        for order, mode, c_lo, c_hi in specs:
            v, ws = ops.masked_seq_loss(gt, pred, e32, c_lo, c_hi, order, prefix, crit, mode, return_ws=True)
            outs.append(v)
            wss.append(ws)
        ctx.save_for_backward(pred, gt)
        ctx.misc = (e32, prefix, crit, specs, wss)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        pred, gt = ctx.saved_tensors
        e32, prefix, crit, specs, wss = ctx.misc
        grad = torch.zeros_like(pred)
        for (order, mode, c_lo, c_hi), ws, g in zip(specs, wss, gs):
            if g is not None:
                ops.masked_seq_loss_bwd_(grad, gt, pred, e32, ws, g, c_lo, c_hi, order, prefix, crit, mode)
        return grad, None, None, None, None, None


class VertexSeqLossFn(torch.autograd.Function):
    """The three vertex-space terms of reference utils/common.py:486-513, 566-574 on (N, T, 15069) vertex sequences:
    masked means of crit(gt - pred), of crit on first differences, and of crit(second difference of pred) -- each
    one msmd_masked_seq_loss launch; the backward ADDS the three gradients into ONE (N, T, 15069) buffer
    (msmd_masked_seq_loss_bwd), instead of autograd's chain of (N, T, 5023, 3) temporaries per term."""

    @staticmethod
    def forward(ctx, pred, gt, end_idx, prefix, crit, want):
        pred, gt = pred.float().contiguous(), gt.float().contiguous()
        C = pred.shape[-1]
        e32 = end_idx.to(torch.int32).contiguous() if end_idx is not None else None
        outs, wss = [], []
        for k, (order, mode) in enumerate(((0, 0), (1, 0), (2, 1))):
            if want[k]:
                v, ws = ops.masked_seq_loss(gt, pred, e32, 0, C, order, prefix, crit, mode, return_ws=True)
            else:
                v, ws = pred.new_zeros(()), None
            outs.append(v)
            wss.append(ws)
        ctx.save_for_backward(pred, gt)
        ctx.misc = (e32, prefix, crit, want, wss)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        pred, gt = ctx.saved_tensors
        e32, prefix, crit, want, wss = ctx.misc
        C = pred.shape[-1]
        grad = torch.zeros_like(pred)
        for k, (order, mode, g) in enumerate(((0, 0, g0), (1, 0, g1), (2, 1, g2))):
            if want[k] and g is not None:
                ops.masked_seq_loss_bwd_(grad, gt, pred, e32, wss[k], g, 0, C, order, prefix, crit, mode)
        return grad, None, None, None, None, None


def loss_vert_train(args, is_starting_sample, shape_coef, motion_coef_gt, target, prev_motion_coef, coef_stats, flame,
                    end_idx=None):
    """Differentiable restatement of the reference's vertex-space loss (utils/common.py:456-620, target='sample',
    legacy 54-d motion = 50 expression + 4 pose coefficients): parameter-space noise / head terms with autograd ops on
    (N, 110, 54) tensors, the vertex / velocity / smoothness terms through FLAME (differentiable pass of
    utils.flame.FLAME.forward) and VertexSeqLossFn.  Returns the reference's dict."""
    from .utils.common import get_coef_dict
    l2 = args.criterion.lower() == "l2"
    crit = (lambda a, b: (a - b) ** 2) if l2 else (lambda a, b: (a - b).abs())
    n_prev = args.n_prev_motions
    if is_starting_sample:
        target = target[:, n_prev:]
        prefix = 0
    else:
        motion_coef_gt = torch.cat([prev_motion_coef, motion_coef_gt], dim=1)
        if getattr(args, "no_constrain_prev", False):
            target = torch.cat([prev_motion_coef, target[:, n_prev:]], dim=1)
        prefix = n_prev
    N, T = target.shape[0], target.shape[1]
    if end_idx is None:
        mask = torch.ones((N, args.n_motions), dtype=torch.bool, device=target.device)
    else:
        mask = torch.arange(args.n_motions, device=target.device).expand(N, -1) < end_idx.unsqueeze(1)
    if not is_starting_sample:
        lead = torch.zeros_like if getattr(args, "no_constrain_prev", False) else torch.ones_like
        mask = torch.cat([lead(mask[:, :n_prev]), mask], dim=1)
        if getattr(args, "no_constrain_prev", False):
            prefix = -prefix
    d1 = lambda x: x[:, 1:] - x[:, :-1]
    gt, pr = motion_coef_gt.float(), target.float()
    out = {"noise": _masked_mean(crit(gt, pr), mask) / 2, "vert": 0, "vel": 0, "smooth": 0, "head_angle": 0, "head_vel": 0,
           "head_smooth": 0, "head_trans": None}
    if args.l_vert > 0 or args.l_vel > 0:
        if args.rot_repr != "aa":
            raise ValueError(f"Unknown rotation representation {args.rot_repr}!")
        cg = get_coef_dict(gt.detach(), shape_coef, coef_stats, with_global_pose=False, rot_repr=args.rot_repr)
        cp = get_coef_dict(pr, shape_coef, coef_stats, with_global_pose=False, rot_repr=args.rot_repr)
        with torch.no_grad():
            vg = flame(cg["shape"].reshape(-1, 100), cg["exp"].reshape(-1, 50), cg["pose"].reshape(-1, 6),
                       return_lm2d=False, return_lm3d=False)[0].view(N, T, -1)
        vp = flame(cp["shape"].reshape(-1, 100), cp["exp"].reshape(-1, 50), cp["pose"].reshape(-1, 6),
                   return_lm2d=False, return_lm3d=False)[0].view(N, T, -1)
        lv, lvel, lsm = VertexSeqLossFn.apply(vp, vg, end_idx, prefix, 0 if l2 else 1,
                                              (args.l_vert > 0, args.l_vel > 0, args.l_smooth > 0))
        out["vert"], out["vel"], out["smooth"] = lv / 2, lvel / 2, lsm / 2
    if not args.no_head_pose:
        hg, hp = gt[:, :, 50:53], pr[:, :, 50:53]
        if args.l_head_angle > 0:
            out["head_angle"] = _masked_mean(crit(hg, hp), mask) / 2
        if args.l_head_vel > 0:
            out["head_vel"] = _masked_mean(crit(d1(hg), d1(hp)), mask[:, 1:]) / 2
        if args.l_head_smooth > 0:
            hv = d1(hp)
            out["head_smooth"] = _masked_mean(crit(hv[:, 1:], hv[:, :-1]), mask[:, 2:]) / 2
        if not is_starting_sample and args.l_head_trans > 0:
            seq = torch.cat([hg[:, n_prev - 3:n_prev], hp[:, n_prev:n_prev + 3]], dim=1)
            v = d1(seq)
            a = d1(v)
            out["head_trans"] = (_masked_mean(crit(v[:, 2:4], v[:, 1:3]), mask[:, n_prev:n_prev + 2])
                                 + _masked_mean(crit(a[:, 1:], a[:, :-1]), mask[:, n_prev:n_prev + 3]))
    return out


def kl_train(mu, logvar):
    """reference utils/common.py:443-454."""
    # summed per row, then over the rows: ONE sum over all 8 192 elements is a multi-block reduction of the host library (a
    # semaphore-and-staging-buffer kernel), which does not belong inside the step's hipGraphs (NullTokenSelectFn above)
    return -0.5 * (1 + logvar - mu.pow(2) - logvar.exp()).sum(dim=1).sum()
