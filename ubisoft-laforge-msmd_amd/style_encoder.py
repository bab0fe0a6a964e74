"""VAE style encoder on the MI355X (drop-in surface of reference style_encoder.py:7-12, 119-213)."""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import init, ops, shapes, synth
from .model import _cd, _is_split
from .utils.model_common import ParamTree, sinusoid_table


def get_style_encoder(args, style_encoder_model_style="diffposetalk"):
    """reference style_encoder.py:7-12: returns None for anything but 'vae2' -- its own default included -- as the reference
    does (every caller passes args.style_enc_model_style = "vae2": training_script.py:527, inference.py:97)."""
    if style_encoder_model_style == "vae2":
        return StyleEncoder_VAE2(args)
    return None


class _PE(nn.Module):
    def __init__(self, d_model, max_len=600):
        super().__init__()
        self.register_buffer("pe", sinusoid_table(d_model, max_len))


class StyleEncoder_VAE2(nn.Module):
    """2x(Conv1d k3 + ELU + LN) -> + PE row -> TransformerEncoderLayer(d512, 8h, ff512) -> Conv1d/ELU/LN ->
    Conv1d -> mean over time -> (mu, logvar) -> reparameterised sample.  All convs are windowed MFMA GEMMs
    over the channels-last motion clip (input channels zero-padded 67 -> 72 for 16-byte rows)."""

    def __init__(self, args) -> None:
        super().__init__()
        self.input_dim = 67
        if args.dataset_type[:9] == "HDTF_TFHP" or args.dataset_type == "flame_mead_ravdess":
            self.input_dim = 54
        self.motion_coef_dim = self.input_dim
        self.conv_feature_dim = 512
        self.output_size = args.d_style * 2
        self.compute_dtype = _cd(args)
        self.split_mode = _is_split(args)
        tree = ParamTree(shapes.style_encoder_shapes(args, self.input_dim, self.conv_feature_dim))
        for name, p in tree._parameters.items():
            self.register_parameter(name, p)
        for name, m in tree._modules.items():
            self.add_module(name, m)
        self.PE = _PE(self.conv_feature_dim)
        if init.synthetic_requested(args):   # closed-form weights, asked for by name (tests / bench / smoke)
            synth.load_synthetic(self)
        else:                                # the reference's torch.nn defaults under the caller's RNG (style_encoder.py:133-176)
            init.style_encoder_(self, args)
        self._packed = None
        self._packed_dtype = None

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def pack(self, dtype):
        split = bool(self.split_mode) and dtype == torch.float32
        if self._packed is not None and self._packed_dtype == (dtype, split):
            return self._packed
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        f32 = lambda t: t.float().contiguous()
        cd = ops.split_weight if split else (lambda t: t.to(dtype).contiguous())
        P = SimpleNamespace()
        mult = 32 if split else 8
        self.cin_pad = (self.input_dim + mult - 1) // mult * mult

        def conv_w(w, cpad=None):  # (Cout, Cin, 3) -> (Cout, 3*Cpad), K = kk*Cpad + c
            Cout, Cin, k = w.shape
            cpad = cpad or Cin
            o = torch.zeros(Cout, k, cpad, device=w.device, dtype=torch.float32)
            o[:, :, :Cin] = w.float().permute(0, 2, 1)
            return cd(o.reshape(Cout, k * cpad))
        P.c1 = (conv_w(sd["input_layers.1.weight"], self.cin_pad), f32(sd["input_layers.1.bias"]))
        P.n1 = (f32(sd["input_layers.5.weight"]), f32(sd["input_layers.5.bias"]))
        P.c2 = (conv_w(sd["input_layers.7.weight"]), f32(sd["input_layers.7.bias"]))
        P.n2 = (f32(sd["input_layers.11.weight"]), f32(sd["input_layers.11.bias"]))
        P.pe = f32(sd["PE.pe"][0])
        P.qkv = (cd(sd["encoder.self_attn.in_proj_weight"]), f32(sd["encoder.self_attn.in_proj_bias"]))
        P.ow = (cd(sd["encoder.self_attn.out_proj.weight"]), f32(sd["encoder.self_attn.out_proj.bias"]))
        P.l1 = (cd(sd["encoder.linear1.weight"]), f32(sd["encoder.linear1.bias"]))
        P.l2 = (cd(sd["encoder.linear2.weight"]), f32(sd["encoder.linear2.bias"]))
        P.en1 = (f32(sd["encoder.norm1.weight"]), f32(sd["encoder.norm1.bias"]))
        P.en2 = (f32(sd["encoder.norm2.weight"]), f32(sd["encoder.norm2.bias"]))
        P.c3 = (conv_w(sd["output_layers.1.weight"]), f32(sd["output_layers.1.bias"]))
        P.n3 = (f32(sd["output_layers.5.weight"]), f32(sd["output_layers.5.bias"]))
        P.c4 = (conv_w(sd["output_layers.7.weight"]), f32(sd["output_layers.7.bias"]))
        self._packed, self._packed_dtype = P, (dtype, split)
        return P

    @staticmethod
    def _conv3(x, wb, act):
        """Conv1d(k=3, padding=1) on channels-last x: zero-pad one frame each side, windowed GEMM."""
        xp = ops.group_pad(x, 1, 1)[:, 0]  # (B, T+2, C)
        return ops.conv1d_cl(xp, wb[0], wb[1], kernel=3, stride=1, act=act)

    @torch.no_grad()
    def mu_logvar(self, motion_coef, dtype=None):
        """Deterministic part of reference style_encoder.py:178-199 -> (mu, logvar) fp32 (B, d_style)."""
        dtype = dtype or self.compute_dtype
        P = self.pack(dtype)
        B, T, _ = motion_coef.shape
        x = ops.pad_cols(motion_coef.float().contiguous(), self.cin_pad, dtype)
        x = ops.layernorm(self._conv3(x, P.c1, ops.ACT_ELU), *P.n1)
        # second LN also adds the single PE row pe[0, T] (row-index quirk, utils/model_common.py:99-101)
        x = ops.layernorm(self._conv3(x, P.c2, ops.ACT_ELU), *P.n2, post_add=P.pe[T].contiguous())
        d = self.conv_feature_dim
        qkv = ops.gemm(x, *P.qkv)
        a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], 8, 64 ** -0.5)
        x = ops.layernorm(ops.gemm(a, *P.ow, residual=x), *P.en1)
        f = ops.gemm(x, *P.l1, act=ops.ACT_GELU)
        x = ops.layernorm(ops.gemm(f, *P.l2, residual=x), *P.en2)
        x = ops.layernorm(self._conv3(x, P.c3, ops.ACT_ELU), *P.n3)
        x = self._conv3(x, P.c4, ops.ACT_NONE)
        out = ops.mean_time(x)
        h = self.output_size // 2
        return out[:, :h].contiguous(), out[:, h:].contiguous()

    @torch.no_grad()
    def forward(self, motion_coef, do_sample=False):
        """reference style_encoder.py:178-207."""
        mu, logvar = self.mu_logvar(motion_coef)
        std = torch.exp(0.5 * logvar)
        eps = torch.randn_like(std)
        if do_sample:
            return mu + eps * std
        return mu + eps * std, mu, logvar

    @torch.no_grad()
    def sample(self, motion_coef):
        """reference style_encoder.py:209-213 (a second, independent reparameterisation draw)."""
        out, mu, logvar = self.forward(motion_coef)
        std = torch.exp(0.5 * logvar)
        eps = torch.randn_like(std)
        return mu + eps * std
