"""MSMD diffusion model on the MI355X: drop-in call surface of reference model.py.

Same class names, constructor arguments, method signatures, attributes, exceptions and state_dict
keys as the reference (model.py:7-17, 20-71, 73-440, 820-996); the arithmetic runs in hand-written
gfx950 HIP kernels reached through the C ABI in include/msmd_hip.h.  PyTorch supplies device
memory, streams and host-side RNG only.

Compute dtype: ``args.compute_dtype`` = "bf16" (speed mode: bf16 storage, fp32 accumulation, fp32
LayerNorm/softmax/GELU maths) or "fp32" (parity mode, exact-fp32 MFMA).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import init, ops, shapes, synth
from .utils.model_common import ParamTree, PositionalEncoding, enc_dec_mask, sinusoid_table  # noqa: F401  (enc_dec_mask re-exported)


def _cd(args) -> torch.dtype:
    name = getattr(args, "compute_dtype", "bf16")
    if name in ("bf16", torch.bfloat16):
        return torch.bfloat16
    if name in ("fp32", "float32", torch.float32):
        return torch.float32
    if name in ("f16x2", "split"):   # parity-grade speed mode: fp32 activations in HBM, contractions on split pairs
        return torch.float32
    if name in ("fp16", "float16", "half", torch.float16):   # inference only: the training kernels are bf16 / fp32
        return torch.float16
    raise ValueError(f"Unknown compute dtype {name}!")


def _is_split(args) -> bool:
    """compute_dtype "f16x2": fp32-grade results (reference arithmetic is fp32, training_script.py:548-551) at 1/3 of
    the f16 MFMA rate -- every contraction runs on MSMD_F16X2 split pairs (three f16 MFMAs per k-step), everything
    else (LayerNorm, softmax, GELU, residuals) in fp32 exactly as in the "fp32" parity mode."""
    return getattr(args, "compute_dtype", "bf16") in ("f16x2", "split")


def get_diffusion_model(args, device="cuda"):
    """reference model.py:7-17."""
    if not hasattr(args, "style_enc_ckpt"):
        args.style_enc_ckpt = None
    regularizer = getattr(args, "regularize_alpha", "None")
    return MSMD(args, device, True, use_head_alpha=False, regularize_alpha=regularizer)


class DiffusionSchedule(nn.Module):
    """reference model.py:20-71.  Init-time host maths with the reference's torch op order, so the
    five (T+1,) buffers are bit-identical to the reference's on the same torch build."""

    def __init__(self, num_steps, mode="linear", beta_1=1e-4, beta_T=0.02, s=0.008):
        super().__init__()
        if mode == "linear":
            betas = torch.linspace(beta_1, beta_T, num_steps)
        elif mode == "quadratic":
            betas = torch.linspace(beta_1 ** 0.5, beta_T ** 0.5, num_steps) ** 2
        elif mode == "sigmoid":
            betas = torch.sigmoid(torch.linspace(-5, 5, num_steps)) * (beta_T - beta_1) + beta_1
        elif mode == "cosine":
            x = torch.linspace(0, num_steps, num_steps + 1)
            alpha_bars = torch.cos(((x / num_steps) + s) / (1 + s) * torch.pi * 0.5) ** 2
            alpha_bars = alpha_bars / alpha_bars[0]
            betas = torch.clip(1 - (alpha_bars[1:] / alpha_bars[:-1]), 0.0001, 0.999)
        else:
            raise ValueError(f"Unknown diffusion schedule {mode}!")
        betas = torch.cat([torch.zeros(1), betas], dim=0)
        alphas = 1 - betas
        log_alphas = torch.log(alphas)
        for i in range(1, log_alphas.shape[0]):
            log_alphas[i] += log_alphas[i - 1]
        alpha_bars = log_alphas.exp()
        sigmas_flex = torch.sqrt(betas)
        sigmas_inflex = torch.zeros_like(sigmas_flex)
        for i in range(1, sigmas_flex.shape[0]):
            sigmas_inflex[i] = ((1 - alpha_bars[i - 1]) / (1 - alpha_bars[i])) * betas[i]
        sigmas_inflex = torch.sqrt(sigmas_inflex)
        self.num_steps = num_steps
        self.register_buffer("betas", betas)
        self.register_buffer("alphas", alphas)
        self.register_buffer("alpha_bars", alpha_bars)
        self.register_buffer("sigmas_flex", sigmas_flex)
        self.register_buffer("sigmas_inflex", sigmas_inflex)
        self._host = None
        self._qs = None

    def qsample_tables(self):
        """(sqrt(alpha_bar), sqrt(1 - alpha_bar)) as fp32 tables: the two torch ops the reference applies to the gathered
        alpha_bar_t on every call (model.py:231-236), applied to the whole schedule once; rebuilt when the buffer changes."""
        ab = self.alpha_bars
        key = (ab.data_ptr(), ab._version, ab.device)
        if self._qs is None or self._qs[0] != key:
            self._qs = (key, torch.sqrt(ab).float().contiguous(), torch.sqrt(1 - ab).float().contiguous())
        return self._qs[1], self._qs[2]

    def host_tables(self):
        """CPU copies for the sampler's per-step scalar coefficients (no device sync inside the loop)."""
        if self._host is None:
            self._host = {k: getattr(self, k).detach().float().cpu() for k in
                          ("betas", "alphas", "alpha_bars", "sigmas_flex", "sigmas_inflex")}
        return self._host

    def _apply(self, fn, *a, **k):
        self._host = None
        self._qs = None
        return super()._apply(fn, *a, **k)

    def uniform_sample_t(self, batch_size):
        ts = torch.randint(1, self.num_steps + 1, (batch_size,))
        return ts.tolist()

    def uniform_sample_t_device(self, batch_size):
        """Same distribution drawn on the device (no host round trip; usable under hipGraph capture)."""
        return torch.randint(1, self.num_steps + 1, (batch_size,), device=self.betas.device)

    def get_sigmas(self, t, flexibility=0):
        assert 0 <= flexibility <= 1
        return self.sigmas_flex[t] * flexibility + self.sigmas_inflex[t] * (1 - flexibility)


class _TE(nn.Module):
    """Holds the `TE.pe` buffer under the reference's key (denoising_net.TE.pe)."""

    def __init__(self, d_model, max_len):
        super().__init__()
        self.register_buffer("pe", sinusoid_table(d_model, max_len))


class DenoisingNetwork_MSMD(nn.Module):
    """reference model.py:820-996 (architecture='decoder')."""

    def __init__(self, args, device="cuda", motion_feat_dim=50, use_head_alpha=True, regularize_alpha="None",
                 init_parameters=True):
        super().__init__()
        self.regularize_alpha = regularize_alpha
        self.use_head_alpha = use_head_alpha
        self.num_of_basis = int(args.num_of_basis)
        self.use_style = args.style_enc_ckpt is not None or not args.style_enc_model_style == "diffposetalk"
        self.motion_feat_dim = motion_feat_dim
        if (args.dataset_type[:9] == "HDTF_TFHP" or args.dataset_type == "flame_mead_ravdess") and motion_feat_dim == 50:
            if args.rot_repr == "aa":
                self.motion_feat_dim += 1 if args.no_head_pose else 4
            else:
                raise ValueError(f"Unknown rotation representation {args.rot_repr}!")
        self.shape_feat_dim = 100
        if self.use_style:
            self.style_feat_dim = args.d_style
            self.person_feat_dim = self.shape_feat_dim + self.style_feat_dim
        else:
            self.person_feat_dim = self.shape_feat_dim
        self.use_indicator = args.use_indicator
        self.architecture = args.architecture
        self.feature_dim = args.feature_dim
        self.n_heads = args.n_heads
        self.n_layers = args.n_layers
        self.mlp_ratio = args.mlp_ratio
        self.align_mask_width = args.align_mask_width
        self.use_learnable_pe = not args.no_use_learnable_pe
        self.n_prev_motions = args.n_prev_motions
        self.n_motions = args.n_motions
        self.n_diff_steps = args.n_diff_steps
        self.compute_dtype = _cd(args)
        self.split_mode = _is_split(args)
        if self.architecture != "decoder":
            raise ValueError(f"Unknown architecture: {self.architecture}")
        if self.feature_dim // self.n_heads != 64:
            raise NotImplementedError("attention kernels are specialised for head_dim 64")

        self.TE = _TE(self.feature_dim, args.n_diff_steps + 1)
        tree = ParamTree(shapes.denoiser_shapes(args, self.motion_feat_dim))
        for name, p in tree._parameters.items():
            self.register_parameter(name, p)
        for name, m in tree._modules.items():
            self.add_module(name, m)
        if not self.use_learnable_pe:
            # reference model.py:866: sinusoidal table under the key PE.pe; its forward adds the ONE row pe[seq_len]
            # to every position (utils/model_common.py:99-101) and applies dropout 0.1 in train mode
            self.PE = PositionalEncoding(self.feature_dim)
        if self.align_mask_width > 0:
            motion_len = self.n_prev_motions + self.n_motions
            mask = enc_dec_mask(motion_len, motion_len, 1, self.align_mask_width - 1, device="cpu")
            mask = torch.nn.functional.pad(mask, (0, 0, 1, 0), value=False)
            self.register_buffer("alignment_mask", mask)
            # purely diagonal mask (width 1)?  decided here on the host, once: pack() may run under hipGraph capture
            Tq, Tk = mask.shape
            want = torch.ones(Tq, Tk, dtype=torch.bool)
            want[0] = False
            if Tq == Tk + 1:
                want[torch.arange(1, Tq), torch.arange(0, Tk)] = False
            self._diag_mask = bool(Tq == Tk + 1 and torch.equal(mask.bool(), want))
        else:
            self.alignment_mask = None
            self._diag_mask = False
        self._packed = None
        self._packed_dtype = None
        if init_parameters:     # the torch.nn defaults of the layers reference model.py:856-908 instantiates, caller's RNG
            init.denoiser_(self, args)
        self.to(device)

    @property
    def device(self):
        return next(self.parameters()).device

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def pack(self, dtype):
        split = bool(getattr(self, "split_mode", False)) and dtype == torch.float32
        if self._packed is not None and self._packed_dtype == (dtype, split):
            return self._packed
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        d, nb, dm = self.feature_dim, self.num_of_basis, self.motion_feat_dim
        f32 = lambda t: t.float().contiguous()
        cd = ops.split_weight if split else (lambda t: t.to(dtype).contiguous())

        def padk(w, mult=32 if split else 8):
            K = w.shape[1]
            Kp = (K + mult - 1) // mult * mult
            out = torch.zeros(w.shape[0], Kp, device=w.device, dtype=torch.float32)
            out[:, :K] = w.float()
            return cd(out)
        P = SimpleNamespace()
        P.split = split
        P.te = f32(sd["TE.pe"][0])
        P.te_cd = P.te if split else P.te.to(dtype)
        if self.use_learnable_pe:
            P.pe = f32(sd["PE"][0])
        else:
            T = 1 + self.n_prev_motions + self.n_motions
            P.pe = f32(sd["PE.pe"][0, T]).expand(T, -1).contiguous()
        P.ds0 = (cd(sd["diff_step_map.0.weight"]), f32(sd["diff_step_map.0.bias"]))
        P.ds2 = (cd(sd["diff_step_map.2.weight"]), f32(sd["diff_step_map.2.bias"]))
        P.pp = (padk(sd["person_proj.weight"]), f32(sd["person_proj.bias"]))
        # feature_proj's K = 68 is padded to a whole 64-element K tile in the 16-bit modes (zeros: the packed input rows are
        # zero-filled to the same width), so it runs on the LDS-DMA GEMM like every other projection (K = 72 took the
        # register-staged kernel: 29.8 us per sampler step for 1.6 GFLOP)
        P.fp = (padk(sd["feature_proj.weight"], 32 if split else (64 if dtype != torch.float32 else 8)), f32(sd["feature_proj.bias"]))
        P.kp_person, P.kp_feat = P.pp[0].shape[1], P.fp[0].shape[1]
        P.mask = self.alignment_mask.to(torch.uint8).contiguous() if self.alignment_mask is not None else None
        # align_mask_width == 1 (the default): motion token t >= 1 sees exactly audio frame t - 1 and the person token
        # sees everything.  A softmax over ONE key is exactly 1, so cross-attention returns V[t - 1] for those rows
        # whatever Q is: only row 0 needs a query, scores and a softmax (see trunk / memory_cross).
        P.diag = self._diag_mask
        P.layers = []
        for n in range(self.n_layers):
            p = f"transformer.layers.{n}."
            L = SimpleNamespace()
            L.sa_w, L.sa_b = cd(sd[p + "self_attn.in_proj_weight"]), f32(sd[p + "self_attn.in_proj_bias"])
            L.sa_ow, L.sa_ob = cd(sd[p + "self_attn.out_proj.weight"]), f32(sd[p + "self_attn.out_proj.bias"])
            w, b = sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"]
            L.ca_qw, L.ca_qb = cd(w[:d]), f32(b[:d])
            L.ca_qw_valu = f32(w[:d]) if split else L.ca_qw   # person_query_attention multiplies on the vector ALU
            L.ca_kvw, L.ca_kvb = cd(w[d:]), f32(b[d:])
            L.ca_ow, L.ca_ob = cd(sd[p + "multihead_attn.out_proj.weight"]), f32(sd[p + "multihead_attn.out_proj.bias"])
            L.l1 = (cd(sd[p + "linear1.weight"]), f32(sd[p + "linear1.bias"]))
            L.l2 = (cd(sd[p + "linear2.weight"]), f32(sd[p + "linear2.bias"]))
            L.n1 = (f32(sd[p + "norm1.weight"]), f32(sd[p + "norm1.bias"]))
            L.n2 = (f32(sd[p + "norm2.weight"]), f32(sd[p + "norm2.bias"]))
            L.n3 = (f32(sd[p + "norm3.weight"]), f32(sd[p + "norm3.bias"]))
            P.layers.append(L)
        # LayerNorm folded into the GEMMs around it (ops.gemm_ln): a Linear that consumes LN(u) carries gamma in its
        # weights, beta in its bias and the row sums of the folded weights
        P.fold = not split and dtype in (torch.bfloat16, torch.float16) and d % 64 == 0 and sd["PE" if self.use_learnable_pe else "PE.pe"].is_cuda
        if P.fold:
            for n, L in enumerate(P.layers):
                p = f"transformer.layers.{n}."
                L.f_sa = (ops.fold_layernorm(sd[p + "self_attn.in_proj_weight"], L.sa_b, *P.layers[n - 1].n3, dtype)
                          if n else None)                                   # the previous layer's norm3 feeds QKV
                L.f_caq = ops.fold_layernorm(sd[p + "multihead_attn.in_proj_weight"][:d], L.ca_qb, *L.n1, dtype)
                L.f_l1 = ops.fold_layernorm(sd[p + "linear1.weight"], L.l1[1], *L.n2, dtype)
        if not split:
            P.ca_kvw_all = torch.cat([L.ca_kvw for L in P.layers], 0).contiguous()
            P.ca_kvb_all = torch.cat([L.ca_kvb for L in P.layers], 0).contiguous()
        P.md0 = (cd(sd["motion_dec.0.weight"]), f32(sd["motion_dec.0.bias"]))
        P.md2 = (cd(sd["motion_dec.2.weight"]), f32(sd["motion_dec.2.bias"]))
        # static bases: first linears stacked (nb*d, d_style); second linears batched (nb, dm, d)
        P.st0 = (cd(torch.cat([sd[f"static_feature_mapping.{b}.0.weight"] for b in range(nb)], 0)),
                 f32(torch.cat([sd[f"static_feature_mapping.{b}.0.bias"] for b in range(nb)], 0)))
        P.st2 = (cd(torch.stack([sd[f"static_feature_mapping.{b}.2.weight"] for b in range(nb)], 0)),
                 f32(torch.stack([sd[f"static_feature_mapping.{b}.2.bias"] for b in range(nb)], 0)))
        self._packed, self._packed_dtype = P, (dtype, split)
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1   # monotonic: captured graphs are keyed on it (ids get reused)
        return P

    # ------------------------------------------------------------------ pieces (shared with the sampler)
    def static_bases(self, static_style_feat, dtype, out_dtype=None):
        """(Ns, 1, d_style) -> (Ns, nb, dm): the 4 style->static-pose MLPs (model.py:964-971); step-invariant."""
        P = self.pack(dtype)
        nb, dm, d = self.num_of_basis, self.motion_feat_dim, self.feature_dim
        s = ops.cast(static_style_feat.reshape(-1, static_style_feat.shape[-1]).contiguous(), dtype)
        Ns = s.shape[0]
        h = ops.gemm(s, *P.st0, act=ops.ACT_GELU)  # (Ns, nb*d)
        stat = torch.empty(Ns, nb, dm, device=s.device, dtype=out_dtype or dtype)
        ops.gemm(h, P.st2[0], P.st2[1], None, out=stat, M=Ns, N=dm, K=d, lda=nb * d, ldw=d, ldc=nb * dm, batch=nb,
                 strideA=d, strideW=dm * d, strideC=dm, strideBias=dm)
        return stat

    def person_token(self, person_feat, step, dtype):
        """person_proj(person_feat) + diff_step_map(TE.pe[0, step]) -> (N, d) (model.py:931-933)."""
        P = self.pack(dtype)
        te = P.te_cd[step]
        emb = ops.gemm(ops.gemm(te, *P.ds0, act=ops.ACT_GELU), *P.ds2)
        pf = ops.pad_cols(person_feat.reshape(person_feat.shape[0], -1).float().contiguous(), P.kp_person, dtype)
        return ops.gemm(pf, *P.pp, residual=emb)

    def memory_kv(self, mem, dtype, stacked=False):
        """Cross-attention K/V projections of the audio memory for every layer: step-invariant in the
        sampler (hoisted out of the T x n_entries loop)."""
        P = self.pack(dtype)
        if stacked and not P.split:
            # one GEMM for all layers (N = n_layers x 2d) instead of n_layers small ones; per-layer column views
            d2 = 2 * self.feature_dim
            kv = ops.gemm(mem, P.ca_kvw_all, P.ca_kvb_all)
            return [kv[..., li * d2:(li + 1) * d2] for li in range(len(P.layers))]
        return [ops.gemm(mem, L.ca_kvw, L.ca_kvb) for L in P.layers]

    def memory_cross(self, kv_list, dtype):
        """Diagonal-mask fast path: the cross-attention branch output (attention + out-projection + bias) of rows
        t >= 1 is V[t - 1] W_o^T + b_o -- independent of the queries, hence of the denoising step: computed here once
        per layer as R (N, 1 + L, d); row 0 is left for trunk() to fill per call."""
        P = self.pack(dtype)
        d = self.feature_dim
        out = []
        for L, kv in zip(P.layers, kv_list):
            N, Tk, _ = kv.shape
            R = torch.zeros(N, Tk + 1, d, device=kv.device, dtype=dtype)     # row 0: finite until trunk() fills it (the last layer leaves it)
            v = kv[..., d:]
            if P.split:
                v = ops.to_split(kv)[..., d:]   # same (N, Tk, 2d) layout in split storage, then the V half
            # one batched launch (batch = sequence) writes straight into rows 1.. of every sequence
            ops.gemm(v, L.ca_ow, L.ca_ob, None, ops.ACT_NONE, out=R[:, 1:, :], M=Tk, N=d, K=d, lda=2 * d, ldc=d, batch=N,
                     strideA=Tk * 2 * d, strideW=0, strideC=(Tk + 1) * d)
            out.append(R)
        return out

    def trunk(self, feats, tok0, mem, dtype, kv_list=None, row0_add=None, cross_list=None):
        """feature_proj + PE + 8 post-LN decoder layers + motion_dec head.  feats: packed (N, 111, Kpad);
        tok0 (N, d); mem (N, 110, d).  Returns dec (N, 110, dm+nb) fp32."""
        P = self.pack(dtype)
        d, H = self.feature_dim, self.n_heads
        N, Tn, _ = feats.shape
        x = ops.gemm(feats, *P.fp)
        ops.add_pe_token(x, P.pe, tok0, row0_add)
        scale = (d // H) ** -0.5
        if P.split:
            return self._trunk_split(P, x, mem, kv_list, cross_list, scale)
        # the fast path pays when R is hoisted over many calls (the sampler passes cross_list); for a single forward
        # the general masked kernels are as fast (measured 5.37 vs 5.46 ms on the bench step), so it stays opt-in there
        diag = P.diag and getattr(self, "diag_fast_path", True) and (cross_list is not None or
                                                                     getattr(self, "diag_single_pass", False))
        if diag and cross_list is None:
            if kv_list is None:
                kv_list = self.memory_kv(mem, dtype)
            cross_list = self.memory_cross(kv_list, dtype)
        fold = P.fold and ops.FOLD_LN
        if kv_list is None and not diag:
            kv_list = self.memory_kv(mem, dtype, stacked=True)
        if fold and not diag:
            # post-LN decoder layers without LayerNorm launches: u* = the un-normalised rows a residual GEMM stored, st* their
            # row statistics; the three LayerNorms are applied where their output is consumed -- as the next GEMM's operand
            # (folded weights) and as the next residual (r_stats).  Only the last layer's norm3 is a kernel of its own.
            u = st = ln = None
            for li, L in enumerate(P.layers):
                if li == 0:
                    qkv = ops.gemm(x, L.sa_w, L.sa_b)
                else:
                    qkv = ops.gemm_ln(u, L.f_sa[0], L.f_sa[2], a_stats=st, w_colsum=L.f_sa[1])
                a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale,
                                  prefetch=(L.sa_ow, L.f_caq[0], L.ca_ow))      # the weights the next launches read
                if li == 0:
                    u1, st1 = ops.gemm_ln(a, L.sa_ow, L.sa_ob, x, stats_out=True)
                else:
                    u1, st1 = ops.gemm_ln(a, L.sa_ow, L.sa_ob, u, r_stats=st, r_gamma=ln[0], r_beta=ln[1], stats_out=True)
                kv = kv_list[li] if kv_list is not None else ops.gemm(mem, L.ca_kvw, L.ca_kvb)
                q = ops.gemm_ln(u1, L.f_caq[0], L.f_caq[2], a_stats=st1, w_colsum=L.f_caq[1])
                nxt = P.layers[li + 1].f_sa[0] if li + 1 < len(P.layers) else P.md0[0]
                c = ops.attention(q, kv[..., :d], kv[..., d:], H, scale, mask=P.mask, prefetch=(L.f_l1[0], L.l2[0], nxt))
                u2, st2 = ops.gemm_ln(c, L.ca_ow, L.ca_ob, u1, r_stats=st1, r_gamma=L.n1[0], r_beta=L.n1[1], stats_out=True)
                f = ops.gemm_ln(u2, L.f_l1[0], L.f_l1[2], act=ops.ACT_GELU, a_stats=st2, w_colsum=L.f_l1[1])
                u, st = ops.gemm_ln(f, L.l2[0], L.l2[1], u2, r_stats=st2, r_gamma=L.n2[0], r_beta=L.n2[1], stats_out=True)
                ln = L.n3
            x = ops.layernorm(u, *ln)
        u = st = ln = None        # diagonal path with fold: the previous layer's un-normalised norm3 input
        # Diagonal path: after a layer's self-attention only ROW 0 of a sequence (the person token) depends on that layer's real
        # cross-attention, and everything from there to the next layer's self-attention is row-wise.  The output is rows 1..
        # (motion_dec below), so the LAST layer's person chain (query + Tq = 1 attention + 192-row out-projection, ~50 us of
        # latency-bound launches) feeds nothing: skipped, same bits in every returned row (sampler B = 64: -1.0 .. -1.4 %).
        # (Round 6 also ran the chain of the other layers -- with norm1 / norm2 / FFN on those rows alone -- as a compact (N, d)
        # problem on a forked stream beside the main stream's norm + FFN, rejoining rows and statistics by a scatter kernel:
        # correct, +0.8 % on one lane (five more small launches and a K = 2048 GEMM of 192 rows are as long as the main
        # stream's FFN), and two lanes already hide the chain: removed)
        n_layers = len(P.layers)
        for li, L in enumerate(P.layers if not (fold and not diag) else ()):
            if u is None:
                qkv = ops.gemm(x, L.sa_w, L.sa_b)
            else:
                qkv = ops.gemm_ln(u, L.f_sa[0], L.f_sa[2], a_stats=st, w_colsum=L.f_sa[1])
            nxt = None
            if li + 1 < len(P.layers):
                nxt = P.layers[li + 1].f_sa[0] if (fold and diag) else P.layers[li + 1].sa_w
            a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale,
                              prefetch=None if P.split else (L.sa_ow, L.l1[0], L.l2[0], nxt))   # the layer's next weights
            # (one launch for any N: choosing by the sequence count made a clip's result depend on the batch around it;
            # against the two-launch form it is -1 % at N = 192 sequences and +1.5 % at N = 3)
            fused_pq = getattr(self, "fused_person_query", True)
            # norm1 without a launch of its own (diagonal path, fused person query): its two consumers apply it -- the
            # person-token query projection through folded weights, the norm2 launch as its first stage (layernorm_pre)
            fold_n1 = fold and diag and fused_pq
            u1 = ops.gemm(a, L.sa_ow, L.sa_ob, residual=x) if u is None else \
                ops.gemm_ln(a, L.sa_ow, L.sa_ob, u, r_stats=st, r_gamma=ln[0], r_beta=ln[1])
            if not fold_n1:
                x = ops.layernorm(u1, *L.n1)
            kv = kv_list[li] if kv_list is not None else ops.gemm(mem, L.ca_kvw, L.ca_kvb)
            last = li + 1 == n_layers
            if diag and last and getattr(self, "skip_dead_person_chain", True):
                # rows 1.. of the output do not see this layer's person token: R[:, 0] keeps whatever it holds
                R = cross_list[li]
                x = ops.layernorm_pre(u1, *L.n1, R, *L.n2) if fold_n1 else ops.layernorm(x, *L.n2, residual=R)
            elif diag:
                # only the person token (row 0) has a real softmax; rows t >= 1 come from the precomputed R
                R = cross_list[li]
                # same-box A/B in the sampler graph: fused -1 % at N = 192 sequences, +1.5 % at N = 3 (a longer serial
                # chain per wave than the two more parallel launches), so it is used from 64 sequences up
                if fold_n1:
                    a0 = ops.person_query_attention(u1, L.f_caq[0], L.f_caq[2], kv, H, scale, wq_colsum=L.f_caq[1])
                elif fused_pq:
                    a0 = ops.person_query_attention(x, L.ca_qw_valu, L.ca_qb, kv, H, scale)      # (N, d), one launch
                else:
                    q0 = ops.gemm(x, L.ca_qw, L.ca_qb, M=N, K=d, lda=Tn * d)                    # (N, d) from x[:, 0]
                    a0 = ops.attention(q0.view(N, 1, d), kv[..., :d], kv[..., d:], H, scale)     # (N, 1, d)
                ops.gemm(a0, L.ca_ow, L.ca_ob, None, ops.ACT_NONE, out=R, M=N, K=d, ldc=Tn * d)  # -> R[:, 0]
                x = ops.layernorm_pre(u1, *L.n1, R, *L.n2) if fold_n1 else ops.layernorm(x, *L.n2, residual=R)
            else:
                q = ops.gemm(x, L.ca_qw, L.ca_qb)
                c = ops.attention(q, kv[..., :d], kv[..., d:], H, scale, mask=P.mask)
                x = ops.layernorm(ops.gemm(c, L.ca_ow, L.ca_ob, residual=x), *L.n2)
            f = ops.gemm(x, *L.l1, act=ops.ACT_GELU)
            if fold and diag and li + 1 < len(P.layers):
                # norm3 is consumed by the next layer's QKV GEMM and out-projection residual only: folded into them
                u, st = ops.gemm_ln(f, L.l2[0], L.l2[1], x, stats_out=True)
                ln = L.n3
            else:
                x = ops.layernorm(ops.gemm(f, *L.l2, residual=x), *L.n3)
        # motion_dec on rows 1.. (windowed view of x, no copy)
        Lm = Tn - 1
        h = torch.empty(N, Lm, d // 2, device=x.device, dtype=dtype)
        xs = ops.to_split(x) if P.split else x
        ops.gemm(xs[:, 1:], *P.md0, None, ops.ACT_GELU, out=h, M=N * Lm, K=d, lda=d, rows_per_batch=Lm,
                 a_batch_stride=Tn * d)
        return ops.gemm(h, *P.md2, out_dtype=torch.float32)

    def _trunk_split(self, P, x, mem, kv_list, cross_list, scale):
        """The decoder layers + motion_dec head of `trunk` in the parity-grade speed mode (x: fp32 (N, Tn, d) after the
        PE / token add).  LayerNorms write fp32 (residual) and split (next GEMM operand) rows in one pass; Q / K / V,
        attention outputs and the FFN hidden layer move between kernels in split storage.  With the diagonal mask and a
        hoisted cross_list (sampler) only the person row runs a real cross-attention, as in the other modes."""
        d, H = self.feature_dim, self.n_heads
        N, Tn, _ = x.shape
        diag = P.diag and getattr(self, "diag_fast_path", True) and cross_list is not None
        xs = ops.to_split(x)
        mem_s = None
        for li, L in enumerate(P.layers):
            qkv = ops.gemm(xs, L.sa_w, L.sa_b, out_dtype=ops.SPLIT)
            a = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, scale)
            x, xs = ops.layernorm(ops.gemm(a, L.sa_ow, L.sa_ob, residual=x), *L.n1, split="both")
            if diag:
                R, kv = cross_list[li], kv_list[li]                                      # fp32 R (N, Tn, d); kv fp32
                if not (li + 1 == len(P.layers) and getattr(self, "skip_dead_person_chain", True)):   # (see trunk)
                    a0 = ops.person_query_attention(x, L.ca_qw_valu, L.ca_qb, kv, H, scale)   # (N, d) fp32, vector ALU
                    ops.gemm(a0, L.ca_ow, L.ca_ob, None, ops.ACT_NONE, out=R, M=N, K=d, ldc=Tn * d)
                x, xs = ops.layernorm(x, *L.n2, residual=R, split="both")
            else:
                if kv_list is not None:
                    kv = ops.to_split(kv_list[li])
                else:
                    mem_s = ops.to_split(mem) if mem_s is None else mem_s
                    kv = ops.gemm(mem_s, L.ca_kvw, L.ca_kvb, out_dtype=ops.SPLIT)
                q = ops.gemm(xs, L.ca_qw, L.ca_qb, out_dtype=ops.SPLIT)
                c = ops.attention(q, kv[..., :d], kv[..., d:], H, scale, mask=P.mask)
                x, xs = ops.layernorm(ops.gemm(c, L.ca_ow, L.ca_ob, residual=x), *L.n2, split="both")
            f = ops.gemm(xs, *L.l1, act=ops.ACT_GELU, out_dtype=ops.SPLIT)
            x, xs = ops.layernorm(ops.gemm(f, *L.l2, residual=x), *L.n3, split="both")
        Lm = Tn - 1
        h = ops.empty((N, Lm, d // 2), x.device, ops.SPLIT)
        ops.gemm(xs[:, 1:], *P.md0, None, ops.ACT_GELU, out=h, M=N * Lm, K=d, lda=d, rows_per_batch=Lm,
                 a_batch_stride=Tn * d)
        return ops.gemm(h, *P.md2, out_dtype=torch.float32)

    def forward(self, motion_feat, audio_feat, person_feat, static_style_feat, prev_motion_feat, prev_audio_feat, step,
                indicator=None, keep_separate=False, dtype=None, _qsample=None, _audio_cd=None):
        """reference model.py:914-996.  Returns (N, L_p + L, d_motion) fp32."""
        dtype = dtype or getattr(self, "compute_dtype", torch.float32)
        if self.use_indicator and indicator is None:
            raise TypeError("expected Tensor as element 1 in argument 0, but got NoneType")  # reference model.py:944
        P = self.pack(dtype)
        N = person_feat.shape[0]
        L, Lp, dm, nb = motion_feat.shape[1], prev_motion_feat.shape[1], self.motion_feat_dim, self.num_of_basis
        step = torch.as_tensor(step, device=self.device, dtype=torch.long)
        tok0 = self.person_token(person_feat, step, dtype)
        feats = torch.empty(N, 1 + Lp + L, P.kp_feat, device=self.device, dtype=dtype)
        eps, c0, c1 = _qsample if _qsample is not None else (None, None, None)
        ops.denoiser_pack_input(motion_feat.float().contiguous(), prev_motion_feat.float().contiguous(),
                                indicator.float().contiguous() if self.use_indicator else None, feats, eps, c0, c1)
        # audio memory [previous window | this window] in the compute dtype: two casting copies into one buffer
        La, Lpa = audio_feat.shape[1], prev_audio_feat.shape[1]
        mem = torch.empty(N, Lpa + La, audio_feat.shape[2], device=self.device, dtype=dtype)
        mem[:, :Lpa].copy_(prev_audio_feat)
        mem[:, Lpa:].copy_(_audio_cd if _audio_cd is not None else audio_feat)
        dec = self.trunk(feats, tok0, mem, dtype)
        stat = self.static_bases(static_style_feat, dtype, out_dtype=torch.float32)   # the head mixes in fp32
        if keep_separate:
            dynamic = dec[:, :, :dm]
            alphas = dec[:, :, dm:]
            if self.regularize_alpha == "sigmoid":
                alphas = torch.sigmoid(alphas.float()).to(alphas.dtype)
            static = stat.float()[:, None].expand(-1, Lp + L, -1, -1)
            if static.shape[0] != N:
                static = static.repeat(N // static.shape[0], 1, 1, 1)
            return dynamic, static, alphas
        return ops.heads_static_mix(dec, stat.float().contiguous(), Lp + L, dm, nb, self.use_head_alpha,
                                    self.regularize_alpha == "sigmoid")


class MSMD(nn.Module):
    """reference model.py:73-818."""

    def __init__(self, args, device="cuda", vae_style=False, conditioned=True, denoisingnset_version=1,
                 use_head_alpha=True, regularize_alpha="None"):
        super().__init__()
        self.target = args.target
        self.regularize_alpha = regularize_alpha
        self.architecture = args.architecture
        self.use_style = (args.style_enc_ckpt is not None) or vae_style
        self.conditioned = conditioned
        self.motion_feat_dim = 67
        self.denoisingnset_version = denoisingnset_version
        self.use_head_alpha = use_head_alpha
        self.fps = args.fps
        self.n_motions = args.n_motions
        self.n_prev_motions = args.n_prev_motions
        self.compute_dtype = _cd(args)
        self.split_mode = _is_split(args)
        if self.use_style:
            self.style_feat_dim = args.d_style
        self.audio_model = args.audio_model
        enc_cfg = dict(num_hidden_layers=getattr(args, "encoder_layers", None))
        # reference model.py:95 / :100: the pretrained encoder, from a LOCAL Hugging Face checkpoint.  args.audio_encoder_weights:
        # None = the reference's hub id (raises when no local copy exists), a directory = that checkpoint, "synthetic" = the
        # closed-form weights (benchmarks / tests), "checkpoint" = bare architecture, a full state_dict follows (inference.load_model);
        # args.hf_cache_dir = hub-cache root (the reference hard-codes its own)
        src = getattr(args, "audio_encoder_weights", None)
        pre = dict(cache_dir=getattr(args, "hf_cache_dir", None),
                   synthetic=True if src == "synthetic" else ("checkpoint" if src == "checkpoint" else None))
        hub = lambda default: default if src in (None, "synthetic", "checkpoint") else src
        if self.audio_model == "wav2vec2":
            from .utils.wav2vec2 import Wav2Vec2Model
            self.audio_encoder = Wav2Vec2Model.from_pretrained(hub("facebook/wav2vec2-base-960h"), config=enc_cfg, **pre)
            frozen = ("feature_extractor",)
        elif self.audio_model == "hubert":
            from .utils.hubert import HubertModel
            self.audio_encoder = HubertModel.from_pretrained(hub("facebook/hubert-base-ls960"), config=enc_cfg, **pre)
            frozen = ("feature_extractor", "feature_projection", "encoder.layers.0.", "encoder.layers.1.")
        elif self.audio_model == "hubert_large":
            # BASELINE.json configs[3] "HuBERT-large encoder swap": not reachable in the reference (model.py:100
            # hard-codes hubert-base and a 768-wide audio_feature_map); same wrapper (utils/hubert.py) and freezing
            # rule, architecture of facebook/hubert-large-ls960-ft.
            from .utils.hubert import HubertModel, LARGE_CONFIG
            cfg = dict(LARGE_CONFIG)
            if getattr(args, "encoder_layers", None):
                cfg["num_hidden_layers"] = args.encoder_layers
            self.audio_encoder = HubertModel.from_pretrained(hub("facebook/hubert-large-ls960-ft"), config=cfg, **pre)
            frozen = ("feature_extractor", "feature_projection", "encoder.layers.0.", "encoder.layers.1.")
        else:
            raise ValueError(f"Unknown audio model {self.audio_model}!")
        for name, p in self.audio_encoder.named_parameters():  # model.py:97,101-110
            if name.startswith(frozen):
                p.requires_grad = False
        # The encoder is what from_pretrained returned and is never written again here.  Everything else starts from the
        # reference's initialisation (msmd_amd.init: the torch.nn defaults, drawn from the caller's torch RNG in the
        # reference's construction order) -- unless the closed-form synthetic weights were asked for by name, which then fill
        # the WHOLE model (tests / bench / smoke: the goldens are recorded on them).
        synthetic = getattr(self.audio_encoder, "weights_source", None) == "synthetic"
        if args.architecture == "decoder":
            self.audio_feature_map = ParamTree({"weight": (args.feature_dim, self.audio_encoder.config.hidden_size),
                                                "bias": (args.feature_dim,)})
            self.start_audio_feat = nn.Parameter(torch.zeros(1, self.n_prev_motions, args.feature_dim))
        else:
            raise ValueError(f"Unknown architecture {args.architecture}!")
        self.start_motion_feat = nn.Parameter(torch.zeros(1, self.n_prev_motions, self.motion_feat_dim))
        if not synthetic:
            init.msmd_front_(self, args)
        self.denoising_net = DenoisingNetwork_MSMD(args, "cpu", motion_feat_dim=self.motion_feat_dim,
                                                   use_head_alpha=self.use_head_alpha,
                                                   regularize_alpha=self.regularize_alpha, init_parameters=not synthetic)
        self.diffusion_sched = DiffusionSchedule(args.n_diff_steps, args.diff_schedule)
        self.cfg_mode = args.cfg_mode
        guiding_conditions = args.guiding_conditions.split(",") if args.guiding_conditions else []
        self.guiding_conditions = [cond for cond in guiding_conditions if cond in ["style", "audio"]]
        if "style" in self.guiding_conditions:
            if not self.use_style:
                raise ValueError("Cannot use style guiding without enabling it!")
            self.null_style_feat = nn.Parameter(torch.zeros(1, 1, self.style_feat_dim))
        if "audio" in self.guiding_conditions:
            self.null_audio_feat = nn.Parameter(torch.zeros(1, 1, args.feature_dim))
        if synthetic:
            synth.load_synthetic(self)   # closed-form fill of every parameter, the (already synthetic) encoder included
        else:
            init.msmd_back_(self, args)
        self.audio_encoder.split_mode = self.split_mode
        self._afm = None
        self.to(device)

    @property
    def device(self):
        return next(self.parameters()).device

    def _apply(self, fn, *a, **k):
        self._afm = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        self._afm = None
        return super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def set_compute_dtype(self, dtype):
        ns = SimpleNamespace(compute_dtype=dtype)
        self.compute_dtype = _cd(ns)
        self.split_mode = _is_split(ns)
        self.denoising_net.compute_dtype = self.compute_dtype
        self.denoising_net.split_mode = self.audio_encoder.split_mode = self.split_mode
        return self

    def _afm_packed(self, dtype):
        split = self.split_mode and dtype == torch.float32
        if self._afm is None or self._afm[0] != (dtype, split):
            w = self.audio_feature_map.weight.detach()
            self._afm = ((dtype, split), ops.split_weight(w) if split else w.to(dtype).contiguous(),
                         self.audio_feature_map.bias.detach().float().contiguous())
        return self._afm[1], self._afm[2]

    # ------------------------------------------------------------------ audio features
    def _audio_768(self, audio, frame_num, dtype):
        h = self.audio_encoder.encode(audio, self.fps, frame_num=frame_num * 2, dtype=dtype, pad=True)  # (N, 2L, 768)
        return ops.interp_linear(h, frame_num)  # 2:1 linear resample == exact pairwise mean (model.py:260)

    def _audio_feat(self, audio, frame_num, dtype):
        w, b = self._afm_packed(dtype)
        return ops.gemm(self._audio_768(audio, frame_num, dtype), w, b)

    @torch.no_grad()
    def extract_audio_feature(self, audio, frame_num=None):
        """reference model.py:250-264 -> (N, L, feature_dim) fp32."""
        frame_num = frame_num or self.n_motions
        return self._audio_feat(audio, frame_num, self.compute_dtype).float()

    @torch.no_grad()
    def extract_audio_768_feature(self, audio, frame_num=None):
        """reference model.py:266-280."""
        frame_num = frame_num or self.n_motions
        return self._audio_768(audio, frame_num, self.compute_dtype).float()

    # ------------------------------------------------------------------ training-forward semantics
    @torch.no_grad()
    def forward(self, motion_feat, audio_or_feat, shape_feat, style_feat=None, prev_motion_feat=None,
                prev_audio_feat=None, time_step=None, indicator=None, train_with_CFG=True, keep_separate=False,
                eps=None):
        """reference model.py:146-248 (inference-mode arithmetic; the autograd path is the next build row).
        ``eps`` may be injected for deterministic replay (the reference draws torch.randn_like)."""
        dtype = self.compute_dtype
        if self.use_style:
            assert style_feat is not None, "Missing style features!"
        batch_size = motion_feat.shape[0]
        if audio_or_feat.ndim == 2:
            assert audio_or_feat.shape[1] == 16000 * self.n_motions / self.fps, \
                f"Incorrect audio length {audio_or_feat.shape[1]}"
            audio_cd = self._audio_feat(audio_or_feat, self.n_motions, dtype)      # compute dtype: what the denoiser reads
            audio_feat_saved = audio_cd.float()
        elif audio_or_feat.ndim == 3:
            assert audio_or_feat.shape[1] == self.n_motions, f"Incorrect audio feature length {audio_or_feat.shape[1]}"
            audio_feat_saved = audio_or_feat
        else:
            raise ValueError(f"Incorrect audio input shape {audio_or_feat.shape}")
        audio_feat = audio_feat_saved
        if audio_or_feat.ndim != 2:
            audio_cd = None
        if shape_feat.ndim == 2:
            shape_feat = shape_feat.unsqueeze(1)
        if style_feat is not None and style_feat.ndim == 2:
            style_feat = style_feat.unsqueeze(1)
        if prev_motion_feat is None:
            prev_motion_feat = self.start_motion_feat.expand(batch_size, -1, -1)
        if prev_audio_feat is None:
            prev_audio_feat = self.start_audio_feat.expand(batch_size, -1, -1)
        # classifier-free guidance masking (model.py:190-218)
        if len(self.guiding_conditions) > 0 and train_with_CFG:
            assert len(self.guiding_conditions) <= 2, "Only support 1 or 2 CFG conditions!"
            if len(self.guiding_conditions) == 1 or self.cfg_mode == "independent":
                null_cond_prob = 0.5 if len(self.guiding_conditions) >= 2 else 0.1
                if "style" in self.guiding_conditions:
                    mask_style = torch.rand(batch_size, device=self.device) < null_cond_prob
                    style_feat = torch.where(mask_style.view(-1, 1, 1),
                                             self.null_style_feat.expand(batch_size, -1, -1), style_feat)
                if "audio" in self.guiding_conditions:
                    mask_audio = torch.rand(batch_size, device=self.device) < null_cond_prob
                    audio_feat = torch.where(mask_audio.view(-1, 1, 1),
                                             self.null_audio_feat.expand(batch_size, self.n_motions, -1), audio_feat)
            else:
                mask_flag = torch.rand(batch_size, device=self.device)
                if "style" in self.guiding_conditions:
                    style_feat = torch.where((mask_flag > 0.55).view(-1, 1, 1),
                                             self.null_style_feat.expand(batch_size, -1, -1), style_feat)
                if "audio" in self.guiding_conditions:
                    audio_feat = torch.where((mask_flag > 0.9).view(-1, 1, 1),
                                             self.null_audio_feat.expand(batch_size, self.n_motions, -1), audio_feat)
        person_feat = shape_feat if style_feat is None else torch.cat([shape_feat, style_feat], dim=-1)
        if time_step is None:
            time_step = self.diffusion_sched.uniform_sample_t(batch_size)
        ts = torch.as_tensor(time_step, device=self.device, dtype=torch.long)
        t0, t1 = self.diffusion_sched.qsample_tables()
        c0, c1 = t0[ts], t1[ts]                                     # sqrt(alpha_bar_t), sqrt(1 - alpha_bar_t)
        if eps is None:
            eps = torch.randn_like(motion_feat)
        eps = eps.float().contiguous()
        # q-sample is fused into the denoiser's input packing kernel (model.py:231-236)
        out = self.denoising_net(motion_feat, audio_feat, person_feat, style_feat, prev_motion_feat, prev_audio_feat,
                                 ts, indicator, keep_separate=keep_separate, dtype=dtype, _qsample=(eps, c0, c1),
                                 _audio_cd=audio_cd if audio_feat is audio_feat_saved else None)
        if keep_separate:
            dyn, stat, alpha_t = out
            if self.use_head_alpha:
                target = dyn + (alpha_t.unsqueeze(-1) * stat).sum(dim=2)
            else:
                target = dyn + torch.cat([(alpha_t.unsqueeze(-1) * stat[..., :-3]).sum(2), stat[..., -3:].sum(2)], -1)
            return eps, target, motion_feat.detach(), audio_feat_saved.detach(), dyn, stat, alpha_t
        return eps, out, motion_feat.detach(), audio_feat_saved.detach()

    # ------------------------------------------------------------------ static-shape replay
    @torch.no_grad()
    def capture_forward(self, motion_feat, audio, shape_feat, style_feat, time_step, indicator, eps,
                        train_with_CFG=False, verify=True, lanes=1):
        """Capture `forward` for these (static) shapes as ONE hipGraph and return `run(**new_inputs) -> outputs`.
        The ~300 launches of a forward are then re-issued by the GPU's command processor: host-side launch jitter
        disappears (it matters when several ranks share a host).  `run` copies any tensors it is given into the
        captured input buffers (device-to-device) and replays; the returned tensors are the graph's output buffers
        (valid until the next replay).  time_step must be a device LongTensor.  With `verify` the first replay is
        checked bit for bit against the eager forward on perturbed inputs (a replay can never serve stale results).

        ``lanes`` > 1: the batch is cut into that many contiguous groups of clips and every group runs the WHOLE forward on a
        HIP stream of its own, forked and joined inside the one captured graph.  Clips are independent in every operator of
        the path, so every lane's result IS `forward` of its group of clips, bit for bit (checked by `verify`
        against the eager per-group forwards); against the one-lane forward of the whole batch the 16-bit modes differ in
        last bits only (row-statistics slabs and tile shapes follow the row count of a launch), which `run.lane_drift`
        reports.  What changes is the schedule: the under-filled phases of one lane's launches (300-tile GEMM grids, the
        decoder's small grids, attention, every launch's tail and epilogue burst) run beside another lane's K loops
        instead of beside nothing.  lanes must divide the batch."""
        lanes = max(1, int(lanes))
        if motion_feat.shape[0] % lanes:
            raise ValueError(f"capture_forward: lanes={lanes} does not divide the batch of {motion_feat.shape[0]}")
        static = dict(motion_feat=motion_feat.clone(), audio=audio.clone(), shape_feat=shape_feat.clone(),
                      style_feat=style_feat.clone(), time_step=torch.as_tensor(time_step, device=self.device).long().clone(),
                      indicator=indicator.clone(), eps=eps.clone())

        def one(sl=slice(None)):
            return self.forward(static["motion_feat"][sl], static["audio"][sl], static["shape_feat"][sl], static["style_feat"][sl],
                                time_step=static["time_step"][sl], indicator=static["indicator"][sl],
                                train_with_CFG=train_with_CFG, eps=static["eps"][sl])
        lane_streams = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else []

        def call():
            if lanes == 1:
                return one()
            if train_with_CFG:
                raise ValueError("capture_forward: lanes > 1 draws nothing inside the graph (train_with_CFG=False, eps given)")
            cur = torch.cuda.current_stream()
            per = static["motion_feat"].shape[0] // lanes
            parts = []
            for i, st in enumerate(lane_streams):        # fork: every lane starts behind the caller's stream ...
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    parts.append(one(slice(i * per, (i + 1) * per)))
            for st in lane_streams:                      # ... and the caller's stream continues behind all of them
                cur.wait_stream(st)
            return tuple(torch.cat([p[k] for p in parts], dim=0) for k in range(len(parts[0])))
        one()                                        # lazy packing before any capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with ops.capture_guard():
            with torch.cuda.graph(graph):
                out = call()
        if verify:
            keep = {k: v.clone() for k, v in static.items()}
            static["audio"].copy_(keep["audio"].flip(0))
            static["motion_feat"].copy_(keep["motion_feat"].flip(0) * 0.5)
            graph.replay()
            torch.cuda.synchronize()
            got = out[1].clone()
            if not torch.equal(call()[1], got):      # the eager forward(s) of the same group(s) of clips
                raise RuntimeError("hipGraph replay of MSMD.forward differs from the eager forward")
            drift = float((one()[1].float() - got.float()).abs().max()) if lanes > 1 else 0.0
            for k, v in keep.items():
                static[k].copy_(v)

        def run(**new_inputs):
            for k, v in new_inputs.items():
                static[k].copy_(v, non_blocking=True)
            graph.replay()
            return out
        run.graph, run.static = graph, static
        run.lanes, run.lane_drift = lanes, (drift if verify else None)   # max |lanes - one-lane| on the verification inputs
        return run

    # ------------------------------------------------------------------ sampler
    @torch.no_grad()
    def sample(self, audio_or_feat, shape_feat, style_feat=None, prev_motion_feat=None, prev_audio_feat=None,
               motion_at_T=None, indicator=None, cfg_mode=None, cfg_cond=None, cfg_scale=1.15, flexibility=0,
               dynamic_threshold=None, ret_traj=False, noise=None):
        """reference model.py:283-440: DDPM ancestral sampling with 1-3-way classifier-free guidance.
        ``noise``: optional dict {t: z_t} of injected draws (deterministic replay); default torch.randn_like."""
        from .sampler import sample as _sample
        return _sample(self, audio_or_feat, shape_feat, style_feat, prev_motion_feat, prev_audio_feat, motion_at_T,
                       indicator, cfg_mode, cfg_cond, cfg_scale, flexibility, dynamic_threshold, ret_traj, noise)

    @torch.no_grad()
    def sample_separate(self, audio_or_feat, shape_feat, style_feat=None, prev_motion_feat=None, prev_audio_feat=None,
                        motion_at_T=None, indicator=None, cfg_mode=None, cfg_cond=None, cfg_scale=1.15, flexibility=0,
                        dynamic_threshold=None, ret_traj=False, alpah_t_modification=None, return_all_alpha=False,
                        noise=None):
        """reference model.py:442-651: same loop, additionally returns the CFG-combined dynamic part of the last
        step, the accumulated static pose and the blend weights (argument spelling kept from the reference)."""
        from .sampler import sample as _sample
        return _sample(self, audio_or_feat, shape_feat, style_feat, prev_motion_feat, prev_audio_feat, motion_at_T,
                       indicator, cfg_mode, cfg_cond, cfg_scale, flexibility, dynamic_threshold, ret_traj, noise,
                       separate=dict(alpha_mod=alpah_t_modification, return_all_alpha=return_all_alpha))

    @torch.no_grad()
    def sample_with_guide(self, audio_or_feat, shape_feat, style_feat=None, prev_motion_feat=None,
                          prev_audio_feat=None, motion_at_T=None, indicator=None, cfg_mode=None, cfg_cond=None,
                          cfg_scale=1.15, flexibility=0, dynamic_threshold=None, ret_traj=False, guidance_indice=None,
                          guidance_values=None, noise=None):
        """reference model.py:653-818 (naive in-painting: guided frames overwrite the denoiser INPUT each step).
        The reference's call at model.py:770 omits `static_style_feat` and raises TypeError; here the static
        branch receives the real style exactly as in `sample` (documented fix)."""
        from .sampler import sample as _sample
        return _sample(self, audio_or_feat, shape_feat, style_feat, prev_motion_feat, prev_audio_feat, motion_at_T,
                       indicator, cfg_mode, cfg_cond, cfg_scale, flexibility, dynamic_threshold, ret_traj, noise,
                       guidance=(guidance_indice, guidance_values))
