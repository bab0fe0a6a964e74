"""Host-side helpers with the reference's names (reference utils/model_common.py:9-123).

Index maths (pad plan, masks) and the checkpoint / args.json wire format stay on the host;
the data movement runs in HIP kernels (csrc/audio.hip).
"""
from __future__ import annotations

import argparse
import json
import math
from pathlib import Path

import torch
import torch.nn as nn

from .. import ops


def pad_audio_plan(audio_len: int, audio_unit: int = 320, pad_threshold: int = 80):
    """(reflect_len applied twice per side, replicate_len) -- reference utils/model_common.py:110-123."""
    n_units = audio_len // audio_unit
    side_len = math.ceil((audio_unit * n_units + pad_threshold - audio_len) / 2)
    if side_len >= 0:
        return side_len // 2, side_len % 2
    return 0, 0


def pad_audio(audio, audio_unit=320, pad_threshold=80):
    """reference utils/model_common.py:110-123 as one gather kernel."""
    r, rep = pad_audio_plan(audio.shape[1], audio_unit, pad_threshold)
    if r == 0 and rep == 0:
        return audio
    return ops.pad_audio(audio.float().contiguous(), r, rep)


def enc_dec_mask(T, S, frame_width=2, expansion=0, device="cuda"):
    """reference utils/model_common.py:103-107 (True = masked)."""
    mask = torch.ones(T, S)
    for i in range(T):
        mask[i, max(0, (i - expansion) * frame_width):(i + expansion + 1) * frame_width] = 0
    return (mask == 1).to(device=device)


def sinusoid_table(d_model: int, max_len: int) -> torch.Tensor:
    """The `pe` buffer of the reference's PositionalEncoding, built with the same torch ops
    (utils/model_common.py:90-97) so the table is bit-identical to the reference's."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


class PositionalEncoding(nn.Module):
    """Buffer holder with the reference's name; forward keeps the row-index quirk
    (adds the single row pe[:, seq_len], utils/model_common.py:99-101)."""

    def __init__(self, d_model, dropout=0.1, max_len=600):
        super().__init__()
        self.p_drop = dropout
        self.register_buffer("pe", sinusoid_table(d_model, max_len))

    def forward(self, x):
        x = x + self.pe[:, x.shape[1], :]
        return torch.nn.functional.dropout(x, self.p_drop, self.training)


# ----------------------------------------------------------------------------- args / checkpoint IO
def save_args(args, save_dir):
    """reference utils/model_common.py:9-27 (drops None-valued keys, stringifies paths)."""
    save_dir = Path(save_dir)
    d = {}
    for k, v in vars(args).items():
        if isinstance(v, Path):
            v = str(v)
        if v is None or v == "None":
            continue
        d[k] = v
    with open(save_dir / "args.json", "w") as f:
        json.dump(d, f)


def load_args(save_dir):
    with open(Path(save_dir) / "args.json", "r") as f:
        return argparse.Namespace(**json.load(f))


def load_args_with_defaults(save_dir, parser):
    with open(Path(save_dir) / "args.json", "r") as f:
        saved = json.load(f)
    d = vars(parser.parse_args([]))
    d.update(saved)
    return argparse.Namespace(**d)


def load_pretrained_model(args, model, style_encoder, device="cuda", parser=None):
    """reference utils/model_common.py:57-81: latest checkpoints/iter_*.pt under args.continue_from."""
    exp_dir = Path(args.continue_from)
    try:
        saved_args = load_args(exp_dir) if parser is None else load_args_with_defaults(exp_dir, parser)
        saved_args.continue_from = str(exp_dir)
        saved_args.max_iter = args.max_iter
    except Exception:
        raise ValueError("Could not load the args from the experiment directory")
    files = sorted((exp_dir / "checkpoints").glob("iter_*.pt"))
    if len(files) == 0:
        raise ValueError(f"No checkpoints found in {exp_dir / 'checkpoints'}")
    ckpt = torch.load(files[-1], map_location=device, weights_only=False)  # holds the args Namespace
    style_encoder.load_state_dict(ckpt["style_enc"])
    model.load_state_dict(ckpt["model"])
    return args, model, style_encoder, ckpt.get("iter", 0)


class ParamTree(nn.Module):
    """Nested parameter container whose state_dict keys equal the given dotted names, so that
    reference checkpoints (HF + torch.nn key names) load unchanged."""

    def __init__(self, shapes=None):
        super().__init__()
        for name, shape in (shapes or {}).items():
            self._add(name.split("."), tuple(shape))

    def _add(self, parts, shape):
        if len(parts) == 1:
            self.register_parameter(parts[0], nn.Parameter(torch.zeros(shape)))
            return
        child = self._modules.get(parts[0])
        if child is None:
            child = ParamTree()
            self.add_module(parts[0], child)
        child._add(parts[1:], shape)

    def get(self, name):
        obj = self
        for p in name.split("."):
            obj = obj._modules[p] if p in obj._modules else obj._parameters[p]
        return obj

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        # torch >= 2.1 stores weight_norm as parametrizations.weight.original{0,1}; the reference's
        # torch 2.0 checkpoints hold weight_g / weight_v.  Accept both.
        for old, new in (("parametrizations.weight.original0", "weight_g"),
                         ("parametrizations.weight.original1", "weight_v")):
            ko = prefix + old
            if ko in state_dict:
                state_dict[prefix + new] = state_dict.pop(ko)
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)
