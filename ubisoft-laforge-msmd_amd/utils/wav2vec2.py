"""wav2vec 2.0 / HuBERT-base audio encoder on the MI355X.

Drop-in surface of reference utils/wav2vec2.py:66-119 (``Wav2Vec2Model.forward(input_values,
output_fps, frame_num=...)`` -> object with ``.last_hidden_state``) with HF's state_dict key names,
but no dependency on ``transformers``: the conv feature extractor, feature projection, positional
grouped conv and the post-LN transformer layers are hand-written HIP kernels (csrc/*.hip) driven
from here.  Activations are channels-last (B, T, C) end to end; every Conv1d is a windowed MFMA
GEMM over that layout, so nothing is transposed or im2col'ed.

Differences from the reference that are deliberate:
  * attention maps are never materialised (the reference forces output_attentions=True and keeps
    737 MB of them alive per batch-32 forward for nothing);
  * SpecAugment / LayerDrop / dropout are training-time noise: the inference path here is eval-mode, the
    training graph (train_graph.audio_encoder_train) applies them when the model is in train() mode.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import ops, shapes, synth
from .model_common import ParamTree, pad_audio_plan

CONV_KERNEL = shapes.CONV_KERNEL
CONV_STRIDE = shapes.CONV_STRIDE


def compute_mask_indices(shape, mask_prob, mask_length, min_masks=0, rng=None):
    """SpecAugment time-mask positions of the reference's wav2vec2 wrapper (utils/wav2vec2.py:17-53, no
    attention_mask): same draws in the same order from `rng` (a numpy RandomState; default the global np.random,
    which is what the reference consumes), so a seeded run reproduces the reference's masks bit for bit."""
    import numpy as np
    rng = np.random if rng is None else rng
    bsz, T = shape
    n_mask = max(min_masks, int(mask_prob * T / float(mask_length) + rng.rand()))
    spans = []
    for _ in range(bsz):
        lengths = np.full(n_mask, mask_length)
        if lengths.sum() == 0:
            lengths[0] = min(mask_length, T - 1)
        min_len = lengths.min()
        if T - min_len <= n_mask:
            min_len = T - n_mask - 1
        starts = rng.choice(T - min_len, n_mask, replace=False)
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1) if n_mask else np.zeros(0, np.int64)
        spans.append(np.unique(idx[idx < T]))
    keep = min(len(m) for m in spans)
    mask = np.zeros((bsz, T), dtype=bool)
    for i, idx in enumerate(spans):
        if len(idx) > keep:
            idx = rng.choice(idx, keep, replace=False)
        mask[i, idx] = True
    return mask


def _compute_mask_indices(shape, mask_prob, mask_length, attention_mask=None, min_masks=0):
    """The reference's name and argument order (utils/wav2vec2.py:17-18).  attention_mask is never passed on the
    reference's path (its model calls the encoder without one, model.py:257), so only that case is built."""
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is never passed on the reference's path (model.py:257)")
    return compute_mask_indices(shape, mask_prob, mask_length, min_masks)


def compute_mask_indices_hf(shape, mask_prob, mask_length, min_masks=0, rng=None):
    """transformers' `_compute_mask_indices` (modeling_wav2vec2.py, used by HubertModel._mask_hidden_states, which the
    reference's HuBERT wrapper calls at utils/hubert.py:35), no attention_mask: one epsilon draw, then one
    `choice` of span starts per row."""
    import numpy as np
    rng = np.random if rng is None else rng
    bsz, T = shape
    if mask_length < 1 or mask_length > T:
        raise ValueError("mask_length must be in [1, sequence_length]")
    eps = rng.rand(1).item()

    def n_spans(length):
        n = max(int(mask_prob * length / mask_length + eps), min_masks)
        if n * mask_length > T:
            n = T // mask_length
        if length - (mask_length - 1) < n:
            n = max(length - (mask_length - 1), 0)
        return n

    n_max = n_spans(T)
    mask = np.zeros((bsz, T), dtype=bool)
    if n_max == 0:
        return mask
    for i in range(bsz):
        starts = rng.choice(np.arange(T - (mask_length - 1)), n_spans(T), replace=False)
        dummy = T - 1 if len(starts) == 0 else starts[0]
        starts = np.concatenate([starts, np.full(n_max - len(starts), dummy, dtype=np.int32)])
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)
        idx[idx > T - 1] = T - 1
        mask[i, idx] = True
    return mask


def linear_interpolation(features, input_fps, output_fps, output_len=None):
    """reference utils/wav2vec2.py:57-63 on a channels-last (N, L, C) tensor."""
    seq_len = features.shape[1] / float(input_fps)
    if output_len is None:
        output_len = int(seq_len * output_fps)
    return ops.interp_linear(features, output_len)


def _hub_roots(cache_dir=None):
    """Directories searched for a hub-cache layout, in order (the last one is the reference's own, model.py:95)."""
    import os
    roots = [cache_dir, os.environ.get("HF_HUB_CACHE"),
             os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
             os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub"), "/code/models/Huggingface/hub2"]
    return [str(r) for r in roots if r]


def _is_checkpoint_dir(d):
    import os
    return os.path.isfile(os.path.join(d, "config.json")) and any(
        os.path.isfile(os.path.join(d, f)) for f in ("model.safetensors", "pytorch_model.bin"))


def find_pretrained(name, cache_dir=None):
    """Local directory of checkpoint `name` (a path, or a hub id resolved in the hub-cache layout), or None."""
    import glob
    import os
    name = str(name)
    if os.path.isdir(name) and _is_checkpoint_dir(name):
        return name
    for root in _hub_roots(cache_dir):
        direct = os.path.join(root, name)
        if os.path.isdir(direct) and _is_checkpoint_dir(direct):
            return direct
        repo = os.path.join(root, "models--" + name.replace("/", "--"))
        ref = os.path.join(repo, "refs", "main")
        if os.path.isfile(ref):
            snap = os.path.join(repo, "snapshots", open(ref).read().strip())
            if _is_checkpoint_dir(snap):
                return snap
        for snap in sorted(glob.glob(os.path.join(repo, "snapshots", "*"))):
            if _is_checkpoint_dir(snap):
                return snap
    return None


def load_hf_state_dict(directory):
    """name -> tensor of a Hugging Face checkpoint directory (safetensors preferred, as transformers does)."""
    import os
    st = os.path.join(directory, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        return load_file(st, device="cpu")
    sd = torch.load(os.path.join(directory, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


class Wav2Vec2Model(nn.Module):
    """HF-key-compatible parameter tree + HIP forward.  ``config`` carries num_hidden_layers etc."""

    model_type = "wav2vec2"

    def __init__(self, config=None):
        super().__init__()
        cfg = dict(num_hidden_layers=12, hidden_size=768, intermediate_size=3072, num_attention_heads=12,
                   conv_dim=512, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                   layer_norm_eps=1e-5, output_attentions=False,
                   feat_extract_norm="group", conv_bias=False, do_stable_layer_norm=False,
                   # training-time noise (facebook/wav2vec2-base-960h / hubert-base-ls960 config.json values)
                   hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, feat_proj_dropout=0.1,
                   layerdrop=0.1, apply_spec_augment=True, mask_time_prob=0.05, mask_time_length=10,
                   mask_time_min_masks=2)
        if config is not None:
            cfg.update({k: v for k, v in (config if isinstance(config, dict) else vars(config)).items() if v is not None})
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if isinstance(c.conv_dim, (tuple, list)):   # HF configs carry one entry per conv layer (all 512)
            c.conv_dim = int(c.conv_dim[0])
        if c.feat_extract_norm not in ("group", "layer"):
            raise ValueError(f"Unknown feat_extract_norm {c.feat_extract_norm!r}")
        tree = ParamTree(shapes.audio_encoder_shapes(c.num_hidden_layers, c.hidden_size, c.intermediate_size,
                                                     c.conv_dim, c.num_conv_pos_embeddings,
                                                     c.num_conv_pos_embedding_groups, c.feat_extract_norm,
                                                     c.conv_bias))
        # adopt the tree's children so state_dict keys carry no extra prefix
        for name, p in tree._parameters.items():
            self.register_parameter(name, p)
        for name, m in tree._modules.items():
            self.add_module(name, m)
        self._packed = None
        self._packed_dtype = None
        self._packed_fe = None
        self._packed_fe_dtype = None
        self.split_mode = False   # set by MSMD for compute_dtype "f16x2" (contractions on MSMD_F16X2 split pairs)

    @classmethod
    def from_pretrained(cls, name=None, cache_dir=None, *, config=None, synthetic=None, **kw):
        """reference model.py:95 / :100 -- ``Wav2Vec2Model.from_pretrained('facebook/wav2vec2-base-960h', cache_dir=...)``.

        Loads a LOCAL Hugging Face checkpoint (there is no network here and no `transformers` on the path):
        ``name`` may be a directory holding ``config.json`` + ``model.safetensors`` / ``pytorch_model.bin``, or a hub id
        that is looked up in the hub cache layout (``<cache_dir>/models--org--name/snapshots/<rev>/``) under
        ``cache_dir``, ``$HF_HUB_CACHE``, ``$HF_HOME/hub``, ``~/.cache/huggingface/hub`` and the reference's hard-coded
        ``/code/models/Huggingface/hub2``.  ``config`` entries override those of ``config.json`` (tests cut
        ``num_hidden_layers``).  Task-head checkpoints (``wav2vec2.*`` / ``hubert.*`` prefixes, ``lm_head``) and both
        weight-norm spellings load; a missing encoder tensor raises.

        When no checkpoint is found this RAISES, as the reference would, unless synthetic weights were asked for:
        ``synthetic=True`` or ``MSMD_SYNTHETIC_WEIGHTS=1`` in the environment (tests, bench.py and smoke() set it: they
        run on the closed-form weights of msmd_amd.synth).  A run never trains on a noise-initialised encoder silently.
        ``synthetic="checkpoint"`` builds the bare architecture for callers that load a full state_dict right after."""
        import os
        if synthetic == "checkpoint":      # the caller loads a complete state_dict next (inference.load_model, Trainer.load_checkpoint)
            m = cls(config)
            m.weights_source = "checkpoint"
            return m
        found = find_pretrained(name, cache_dir) if name is not None else None
        if found is None:
            if synthetic is None:
                synthetic = os.environ.get("MSMD_SYNTHETIC_WEIGHTS", "0") not in ("", "0")
            if not synthetic:
                raise FileNotFoundError(
                    f"{cls.__name__}.from_pretrained({name!r}): no local Hugging Face checkpoint (config.json + "
                    f"model.safetensors | pytorch_model.bin) under {_hub_roots(cache_dir)}; pass a checkpoint directory, "
                    f"set cache_dir / HF_HOME, or ask for synthetic weights explicitly (synthetic=True, "
                    f"args.audio_encoder_weights='synthetic' or MSMD_SYNTHETIC_WEIGHTS=1)")
            m = cls(config)
            synth.load_synthetic(m, prefix="audio_encoder.")
            m.weights_source = "synthetic"
            return m
        import json
        with open(os.path.join(found, "config.json")) as f:
            hf = json.load(f)
        cfg = {k: hf[k] for k in ("num_hidden_layers", "hidden_size", "intermediate_size", "num_attention_heads", "conv_dim",
                                  "num_conv_pos_embeddings", "num_conv_pos_embedding_groups", "layer_norm_eps",
                                  "feat_extract_norm", "conv_bias", "do_stable_layer_norm", "hidden_dropout",
                                  "attention_dropout", "activation_dropout", "feat_proj_dropout", "layerdrop",
                                  "apply_spec_augment", "mask_time_prob", "mask_time_length", "mask_time_min_masks") if k in hf}
        if tuple(hf.get("conv_kernel", CONV_KERNEL)) != CONV_KERNEL or tuple(hf.get("conv_stride", CONV_STRIDE)) != CONV_STRIDE:
            raise ValueError(f"{found}: conv_kernel / conv_stride {hf.get('conv_kernel')} / {hf.get('conv_stride')} are not the "
                             f"wav2vec2 / HuBERT feature extractor this path implements ({CONV_KERNEL} / {CONV_STRIDE})")
        if len(set(hf.get("conv_dim", [512]))) != 1:
            raise ValueError(f"{found}: conv_dim {hf['conv_dim']} varies per layer; the path's feature extractor is uniform")
        if config is not None:     # e.g. fewer layers than the checkpoint holds (tests): the rest of the checkpoint is dropped
            over = dict(config if isinstance(config, dict) else vars(config))
            cfg.update({k: v for k, v in over.items() if v is not None})
        m = cls(cfg)
        sd = load_hf_state_dict(found)
        want = m.state_dict()
        clean = {}
        for k, v in sd.items():
            for head in ("wav2vec2.", "hubert."):      # task-head checkpoints (Wav2Vec2ForCTC: the 960h model) nest the encoder
                if k.startswith(head):
                    k = k[len(head):]
            k = synth.canonical_name(k)
            if k in want:
                clean[k] = v
        missing = [k for k in want if k not in clean and k != "masked_spec_embed"]
        if missing:
            raise KeyError(f"{found}: checkpoint lacks {len(missing)} encoder tensors, e.g. {missing[:4]}")
        m.load_state_dict(clean, strict=False)
        m.weights_source = found
        return m

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        self._packed = self._packed_fe = None
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = self._packed_fe = None
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ weight packing (load time)
    def pack_fe(self, dtype):
        """The convolutional feature extractor's weights alone (frozen in training, model.py freeze_feature_encoder:
        the Trainer keeps this pack across optimizer steps instead of re-casting it with the rest)."""
        split = bool(self.split_mode) and dtype == torch.float32
        if self._packed_fe is not None and self._packed_fe_dtype == (dtype, split):
            return self._packed_fe
        c = self.config
        fe = "feature_extractor.conv_layers."
        sd = {k: v.detach() for k, v in self.state_dict().items() if k.startswith(fe)}
        P = SimpleNamespace()
        P.split = split
        f32 = lambda t: t.float().contiguous()
        cd = ops.split_weight if split else (lambda t: t.to(dtype).contiguous())
        P.w0 = f32(sd[fe + "0.conv.weight"].reshape(c.conv_dim, CONV_KERNEL[0]))
        P.gn_g, P.gn_b = f32(sd[fe + "0.layer_norm.weight"]), f32(sd[fe + "0.layer_norm.bias"])
        nconv = len(CONV_KERNEL)
        P.conv_b = [f32(sd[fe + f"{i}.conv.bias"]) if c.conv_bias else None for i in range(nconv)]
        P.conv_ln = [(f32(sd[fe + f"{i}.layer_norm.weight"]), f32(sd[fe + f"{i}.layer_norm.bias"]))
                     if c.feat_extract_norm == "layer" else None for i in range(nconv)]
        # conv i>=1: (Cout, Cin, k) -> (Cout, k*Cin), K index = kk*Cin + cin (channels-last window)
        P.conv_w = [cd(sd[fe + f"{i}.conv.weight"].permute(0, 2, 1).reshape(c.conv_dim, -1))
                    for i in range(1, len(CONV_KERNEL))]
        self._packed_fe, self._packed_fe_dtype = P, (dtype, split)
        return P

    def pack(self, dtype):
        split = bool(self.split_mode) and dtype == torch.float32
        if self._packed is not None and self._packed_dtype == (dtype, split):
            return self._packed
        c = self.config
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        P = SimpleNamespace(**vars(self.pack_fe(dtype)))
        f32 = lambda t: t.float().contiguous()
        cd = ops.split_weight if split else (lambda t: t.to(dtype).contiguous())
        P.fp_ln = (f32(sd["feature_projection.layer_norm.weight"]), f32(sd["feature_projection.layer_norm.bias"]))
        P.fp_w, P.fp_b = cd(sd["feature_projection.projection.weight"]), f32(sd["feature_projection.projection.bias"])
        # positional conv: fold weight norm (dim=2): w = g * v / ||v||_{dims 0,1}
        g, v = sd["encoder.pos_conv_embed.conv.weight_g"].float(), sd["encoder.pos_conv_embed.conv.weight_v"].float()
        w = ops.fold_weight_norm(g, v) if v.is_cuda else v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        G = c.num_conv_pos_embedding_groups
        cg = c.hidden_size // G
        kpos = c.num_conv_pos_embeddings
        # split storage keeps 32-element blocks whole, so the per-group channel count is zero-padded 48 -> 64 there
        P.pos_cg = (cg + 31) // 32 * 32 if split else cg
        wp = torch.zeros(G, cg, kpos, P.pos_cg, device=w.device, dtype=torch.float32)
        wp[..., :cg] = w.reshape(G, cg, cg, kpos).permute(0, 1, 3, 2)
        P.pos_w = cd(wp.reshape(G, cg, kpos * P.pos_cg))  # K = kk*cg_pad + ci
        P.pos_b = f32(sd["encoder.pos_conv_embed.conv.bias"])
        P.enc_ln = (f32(sd["encoder.layer_norm.weight"]), f32(sd["encoder.layer_norm.bias"]))
        P.layers = []
        for n in range(c.num_hidden_layers):
            p = f"encoder.layers.{n}."
            L = SimpleNamespace()
            L.wqkv = cd(torch.cat([sd[p + f"attention.{x}_proj.weight"] for x in ("q", "k", "v")], 0))
            L.bqkv = f32(torch.cat([sd[p + f"attention.{x}_proj.bias"] for x in ("q", "k", "v")], 0))
            L.wo, L.bo = cd(sd[p + "attention.out_proj.weight"]), f32(sd[p + "attention.out_proj.bias"])
            L.ln1 = (f32(sd[p + "layer_norm.weight"]), f32(sd[p + "layer_norm.bias"]))
            L.w1, L.b1 = cd(sd[p + "feed_forward.intermediate_dense.weight"]), f32(sd[p + "feed_forward.intermediate_dense.bias"])
            L.w2, L.b2 = cd(sd[p + "feed_forward.output_dense.weight"]), f32(sd[p + "feed_forward.output_dense.bias"])
            L.ln2 = (f32(sd[p + "final_layer_norm.weight"]), f32(sd[p + "final_layer_norm.bias"]))
            P.layers.append(L)
        # LayerNorm folded into the GEMMs around it (ops.gemm_ln): the Linear that consumes LN(u) carries gamma in its
        # weights, beta in its bias and the row sums of the folded weights for the mean correction
        P.fold = (not split and dtype in (torch.bfloat16, torch.float16) and c.hidden_size % 128 == 0
                  and c.intermediate_size % 128 == 0 and sd["encoder.layer_norm.weight"].is_cuda)
        if P.fold:
            for n, L in enumerate(P.layers):
                p = f"encoder.layers.{n}."
                wqkv = torch.cat([sd[p + f"attention.{x}_proj.weight"] for x in ("q", "k", "v")], 0)
                w1 = sd[p + "feed_forward.intermediate_dense.weight"]
                if c.do_stable_layer_norm:      # pre-LN block: LN1 feeds QKV, LN2 feeds the FFN
                    L.f_qkv = ops.fold_layernorm(wqkv, L.bqkv, *L.ln1, dtype)
                    L.f_w1 = ops.fold_layernorm(w1, L.b1, *L.ln2, dtype)
                else:                           # post-LN block: the PREVIOUS layer's LN2 feeds QKV, this layer's LN1 the FFN
                    L.f_qkv = ops.fold_layernorm(wqkv, L.bqkv, *P.layers[n - 1].ln2, dtype) if n else None
                    L.f_w1 = ops.fold_layernorm(w1, L.b1, *L.ln1, dtype)
        self._packed, self._packed_dtype = P, (dtype, split)
        return P

    # ------------------------------------------------------------------ forward pieces
    def feature_extractor_cl(self, audio, dtype, reflect_len=None, replicate_len=None):
        """HF Wav2Vec2FeatureEncoder on UNPADDED audio (B, L) with pad_audio fused into conv0's loads.
        Returns (B, T50, 512) channels-last."""
        P = self.pack_fe(dtype)
        if reflect_len is None:
            reflect_len, replicate_len = 0, 0
        eps = self.config.layer_norm_eps
        if self.config.feat_extract_norm == "layer":
            # HubertLayerNormConvLayer x 7: conv(+bias) -> LayerNorm over channels -> GELU
            x = ops.conv0_ln_gelu(audio.float().contiguous(), P.w0, P.conv_b[0], *P.conv_ln[0], reflect_len,
                                  replicate_len, dtype, eps)
            for i, w in enumerate(P.conv_w):
                z = ops.conv1d_cl(x, w, P.conv_b[i + 1], kernel=CONV_KERNEL[i + 1], stride=CONV_STRIDE[i + 1])
                x = ops.layernorm(z, *P.conv_ln[i + 1], post_act=ops.ACT_GELU, eps=eps)
            return x
        if P.split:
            # parity-grade speed mode: the stack's activations stay in split storage between the convs (conv0 and
            # every GEMM epilogue write [hi | lo] rows), the last conv hands fp32 to the crop / resample / LayerNorm
            x = ops.conv0_gn_gelu(audio.float().contiguous(), P.w0, P.gn_g, P.gn_b, reflect_len, replicate_len,
                                  ops.SPLIT, eps)
            for i, w in enumerate(P.conv_w):
                last = i == len(P.conv_w) - 1
                x = ops.conv1d_cl(x, w, P.conv_b[i + 1], kernel=CONV_KERNEL[i + 1], stride=CONV_STRIDE[i + 1],
                                  act=ops.ACT_GELU, out_dtype=torch.float32 if last else ops.SPLIT)
            return x
        x = ops.conv0_gn_gelu(audio.float().contiguous(), P.w0, P.gn_g, P.gn_b, reflect_len, replicate_len, dtype, eps)
        for i, w in enumerate(P.conv_w):
            x = ops.conv1d_cl(x, w, P.conv_b[i + 1], kernel=CONV_KERNEL[i + 1], stride=CONV_STRIDE[i + 1],
                              act=ops.ACT_GELU)
        return x

    def encode_features(self, x, dtype):
        """feature projection + positional conv + transformer layers on (B, T, 512) -> (B, T, 768)."""
        P = self.pack(dtype)
        c = self.config
        B, T, _ = x.shape
        H = c.num_attention_heads
        d = c.hidden_size
        if P.split and not c.do_stable_layer_norm:
            return self._encode_features_split(x, P)
        h = ops.layernorm(x, *P.fp_ln, eps=c.layer_norm_eps)
        h = ops.gemm(h, P.fp_w, P.fp_b)
        # positional grouped conv (k=128, pad=64, drop last frame) as G windowed GEMMs + GELU + residual
        G, cg, kpos = c.num_conv_pos_embedding_groups, d // c.num_conv_pos_embedding_groups, c.num_conv_pos_embeddings
        cgp = P.pos_cg
        xp = ops.group_pad(h, G, kpos // 2, cg_out=cgp, split=P.split)  # (B, G, T + kpos, cgp)
        Tp = T + kpos
        y = torch.empty_like(h)
        ops.gemm(xp, P.pos_w, P.pos_b, h, ops.ACT_GELU, out=y, M=B * T, N=cg, K=kpos * cgp, lda=cgp, rows_per_batch=T,
                 a_batch_stride=G * Tp * cgp, ldw=kpos * cgp, ldc=d, batch=G, strideA=Tp * cgp, strideW=cg * kpos * cgp,
                 strideC=cg, strideBias=cg, strideR=cg)
        scale, eps = (d // H) ** -0.5, c.layer_norm_eps
        if P.fold and ops.FOLD_LN and c.do_stable_layer_norm:
            # pre-LN blocks without LayerNorm launches: every residual GEMM also writes the row statistics of what it
            # stored, the GEMM that consumes LN(h) applies them in its epilogue (include/msmd_hip.h msmd_gemm_ln)
            h, st = y, None                     # y comes from the grouped positional conv: no statistics yet
            for li, L in enumerate(P.layers):
                if st is None:
                    qkv = ops.gemm(ops.layernorm(h, *L.ln1, eps=eps), L.wqkv, L.bqkv)
                else:
                    qkv = ops.gemm_ln(h, L.f_qkv[0], L.f_qkv[2], a_stats=st, w_colsum=L.f_qkv[1], eps=eps)
                nxt = P.layers[li + 1].f_qkv[0] if li + 1 < len(P.layers) else None
                a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale,
                                  prefetch=(L.wo, L.f_w1[0], L.w2, nxt))
                h, st = ops.gemm_ln(a, L.wo, L.bo, h, stats_out=True)
                f = ops.gemm_ln(h, L.f_w1[0], L.f_w1[2], act=ops.ACT_GELU, a_stats=st, w_colsum=L.f_w1[1], eps=eps)
                h, st = ops.gemm_ln(f, L.w2, L.b2, h, stats_out=True)
            return ops.layernorm(h, *P.enc_ln, eps=eps)
        if c.do_stable_layer_norm:
            # HubertEncoderStableLayerNorm: pre-LN blocks, ONE LayerNorm after the last layer
            h = y
            for L in P.layers:
                qkv = ops.gemm(ops.layernorm(h, *L.ln1, eps=c.layer_norm_eps), L.wqkv, L.bqkv)
                a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, (d // H) ** -0.5)
                h = ops.gemm(a, L.wo, L.bo, residual=h)
                f = ops.gemm(ops.layernorm(h, *L.ln2, eps=c.layer_norm_eps), L.w1, L.b1, act=ops.ACT_GELU)
                h = ops.gemm(f, L.w2, L.b2, residual=h)
            return ops.layernorm(h, *P.enc_ln, eps=c.layer_norm_eps)
        h = ops.layernorm(y, *P.enc_ln, eps=c.layer_norm_eps)
        if P.fold and ops.FOLD_LN:
            # post-LN blocks without LayerNorm launches: u = the un-normalised rows a residual GEMM stored (+ their row
            # statistics); LN(u) is applied on the fly where it is the next GEMM's operand (folded weights) and where it
            # is the next residual (r_stats).  Only the last layer's LN2 is a kernel of its own.
            u = st = ln = None
            for n, L in enumerate(P.layers):
                if n == 0:
                    qkv = ops.gemm(h, L.wqkv, L.bqkv)
                else:
                    qkv = ops.gemm_ln(u, L.f_qkv[0], L.f_qkv[2], a_stats=st, w_colsum=L.f_qkv[1], eps=eps)
                nxt = P.layers[n + 1].f_qkv[0] if n + 1 < len(P.layers) else None
                a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale,
                                  prefetch=(L.wo, L.f_w1[0], L.w2, nxt))
                if n == 0:
                    u1, st1 = ops.gemm_ln(a, L.wo, L.bo, h, stats_out=True)
                else:
                    u1, st1 = ops.gemm_ln(a, L.wo, L.bo, u, r_stats=st, r_gamma=ln[0], r_beta=ln[1], stats_out=True, eps=eps)
                f = ops.gemm_ln(u1, L.f_w1[0], L.f_w1[2], act=ops.ACT_GELU, a_stats=st1, w_colsum=L.f_w1[1], eps=eps)
                u, st = ops.gemm_ln(f, L.w2, L.b2, u1, r_stats=st1, r_gamma=L.ln1[0], r_beta=L.ln1[1], stats_out=True, eps=eps)
                ln = L.ln2
            return ops.layernorm(u, *ln, eps=eps)
        for li, L in enumerate(P.layers):
            qkv = ops.gemm(h, L.wqkv, L.bqkv)
            nxt = P.layers[li + 1].wqkv if li + 1 < len(P.layers) else None
            a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, (d // H) ** -0.5,
                              prefetch=None if P.split else (L.wo, L.w1, L.w2, nxt))
            h = ops.layernorm(ops.gemm(a, L.wo, L.bo, residual=h), *L.ln1, eps=c.layer_norm_eps)
            f = ops.gemm(h, L.w1, L.b1, act=ops.ACT_GELU)
            h = ops.layernorm(ops.gemm(f, L.w2, L.b2, residual=h), *L.ln2, eps=c.layer_norm_eps)
        return h

    def _encode_features_split(self, x, P):
        """encode_features (post-LN encoder) in the parity-grade speed mode.  Data flow per layer: LayerNorm writes the
        row twice in one pass (fp32 = the next residual, split = the next GEMM's A operand); the QKV GEMM writes split
        Q / K / V, the split attention writes split O, the FFN's GELU output goes GEMM -> GEMM in split storage; every
        residual add and LayerNorm input stays fp32.  No standalone conversion pass."""
        c = self.config
        B, T, _ = x.shape
        H, d, eps = c.num_attention_heads, c.hidden_size, c.layer_norm_eps
        h = ops.gemm(ops.layernorm(x, *P.fp_ln, eps=eps, split="only"), P.fp_w, P.fp_b)          # fp32 (B, T, d)
        G, cg, kpos = c.num_conv_pos_embedding_groups, d // c.num_conv_pos_embedding_groups, c.num_conv_pos_embeddings
        cgp = P.pos_cg
        xp = ops.group_pad(h, G, kpos // 2, cg_out=cgp, split=True)
        Tp = T + kpos
        y = torch.empty_like(h)
        ops.gemm(xp, P.pos_w, P.pos_b, h, ops.ACT_GELU, out=y, M=B * T, N=cg, K=kpos * cgp, lda=cgp, rows_per_batch=T,
                 a_batch_stride=G * Tp * cgp, ldw=kpos * cgp, ldc=d, batch=G, strideA=Tp * cgp, strideW=cg * kpos * cgp,
                 strideC=cg, strideBias=cg, strideR=cg)
        h, hs = ops.layernorm(y, *P.enc_ln, eps=eps, split="both")
        scale = (d // H) ** -0.5
        for li, L in enumerate(P.layers):
            qkv = ops.gemm(hs, L.wqkv, L.bqkv, out_dtype=ops.SPLIT)
            # (no weight prefetch here: measured 8.90 -> 8.99 ms -- the three-MFMA GEMMs hide the cold weights themselves)
            a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale)           # split out
            h, hs = ops.layernorm(ops.gemm(a, L.wo, L.bo, residual=h), *L.ln1, eps=eps, split="both")
            f = ops.gemm(hs, L.w1, L.b1, act=ops.ACT_GELU, out_dtype=ops.SPLIT)
            t = ops.gemm(f, L.w2, L.b2, residual=h)
            if li + 1 < len(P.layers):
                h, hs = ops.layernorm(t, *L.ln2, eps=eps, split="both")
            else:
                h = ops.layernorm(t, *L.ln2, eps=eps)
        return h

    def encode(self, audio, output_fps=25, frame_num=None, dtype=torch.bfloat16, pad=True):
        """Whole encoder from raw (B, L) audio; pad=True applies the reference's pad_audio plan
        (model.py:257 passes pad_audio(audio)) inside conv0.  Returns (B, frame_num, 768) in `dtype`."""
        r, rep = pad_audio_plan(audio.shape[1]) if pad else (0, 0)
        x = self.feature_extractor_cl(audio, dtype, r, rep)
        T50 = x.shape[1]
        if frame_num is not None:
            crop = min(round(frame_num * 50 / output_fps), T50)   # Python banker's round, utils/wav2vec2.py:82
            out_len = frame_num
        else:
            crop = T50
            out_len = int(T50 / 50.0 * output_fps)
        if not (crop == T50 and out_len == T50):                   # 50 fps -> 2*25 fps is the identity
            x = ops.interp_linear(x, out_len, crop)
        return self.encode_features(x, dtype)

    def forward(self, input_values, output_fps=25, attention_mask=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, frame_num=None, dtype=None):
        """reference utils/wav2vec2.py:71-119 / utils/hubert.py:13-51 (input is ALREADY padded audio)."""
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed on the reference's path (model.py:257)")
        dtype = dtype or torch.float32
        h = self.encode(input_values, output_fps, frame_num, dtype, pad=False)
        return SimpleNamespace(last_hidden_state=h, hidden_states=None, attentions=None)
