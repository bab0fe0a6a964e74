"""Tensor-level shims over the C ABI (include/msmd_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator, so no
hipMalloc in steady state) and the current HIP stream; every op below hands raw
device pointers + sizes to a hand-written gfx950 kernel.  All ops launch on
``torch.cuda.current_stream()`` and never synchronise, so sequences of them can
be captured in a hipGraph (torch.cuda.CUDAGraph).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib

F32, BF16, F16, F16X2 = 0, 1, 2, 3
SPLIT = "f16x2"   # dtype token of the parity-grade speed mode (MSMD_F16X2 split storage, include/msmd_hip.h)


class Split:
    """A tensor in MSMD_F16X2 split storage: logical shape (..., C) fp32-grade values, physically `.t` = (..., 2 C)
    fp16 in 32-element blocks [hi x 32 | lo x 32] (x ~= hi + lo / 2048).  Only what the call sites need: logical shape /
    strides, slicing (last-dim slices on multiples of 32), the device pointer.  `.float()` converts back."""
    __slots__ = ("t", "below_32")
    dtype = SPLIT
    is_cuda = True

    def __init__(self, t, below_32=False):
        if t.dtype != torch.float16 or t.shape[-1] % 64 or t.stride(-1) != 1:
            raise TypeError("Split wraps an fp16 tensor whose last dim is 2 x (a multiple of 32)")
        self.t = t
        self.below_32 = below_32    # every |value| < 32, checked once by split_weight(): lets gemm() pass MSMD_GEMM_W_BELOW_32

    @property
    def shape(self):
        return torch.Size((*self.t.shape[:-1], self.t.shape[-1] // 2))

    @property
    def device(self):
        return self.t.device

    def dim(self):
        return self.t.dim()

    def numel(self):
        return self.t.numel() // 2

    def stride(self, i):
        i = i % self.t.dim()
        return 1 if i == self.t.dim() - 1 else self.t.stride(i) // 2

    def data_ptr(self):
        return self.t.data_ptr()

    def is_contiguous(self):
        return self.t.is_contiguous()

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if Ellipsis in idx:
            k = idx.index(Ellipsis)
            idx = idx[:k] + (slice(None),) * (self.t.dim() - (len(idx) - 1)) + idx[k + 1:]
        if len(idx) == self.t.dim():
            last = idx[-1]
            if not isinstance(last, slice) or last.step not in (None, 1):
                raise IndexError("Split: only contiguous slices along the last dim")
            C = self.shape[-1]
            a, b, _ = last.indices(C)
            if a % 32 or (b % 32 and b != C):
                raise IndexError("Split: last-dim slices must fall on multiples of 32")
            idx = idx[:-1] + (slice(2 * a, 2 * b),)
        return Split(self.t[idx], self.below_32)

    def view(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
        return Split(self.t.view(*shape[:-1], 2 * shape[-1]), self.below_32)

    reshape = view

    def float(self):
        return unsplit(self)


def empty(shape, device, dtype):
    """torch.empty that understands the SPLIT token (last dim must then be a multiple of 32)."""
    if dtype == SPLIT:
        if shape[-1] % 32:
            raise ValueError("split storage needs a last dim that is a multiple of 32")
        return Split(torch.empty(*shape[:-1], 2 * shape[-1], device=device, dtype=torch.float16))
    return torch.empty(*shape, device=device, dtype=dtype)


def to_split(x, cols_out=None):
    """fp32 (..., C) -> Split (..., cols_out) (zero-padded to a multiple of 32); rows may be strided (last dim
    contiguous, uniform row stride)."""
    if isinstance(x, Split):
        return x
    _need_cuda(x)
    if x.dtype != torch.float32:
        x = x.float()
    C = x.shape[-1]
    cols_out = (C + 31) // 32 * 32 if cols_out is None else cols_out
    if not x.is_contiguous():
        x = x.contiguous()
    rows = x.numel() // C
    out = torch.empty(*x.shape[:-1], 2 * cols_out, device=x.device, dtype=torch.float16)
    _lib.check(_lib.load().msmd_split_f16x2(_p(x), _p(out), rows, C, C, cols_out, _stream()), "msmd_split_f16x2")
    return Split(out)


def split_weight(w):
    """to_split for a WEIGHT (packed once at load): also records whether every |w| < 32, which gemm() hands to the library as
    MSMD_GEMM_W_BELOW_32 (include/msmd_hip.h).  One host sync per weight, at pack time only."""
    s = to_split(w.float().contiguous())
    s.below_32 = bool(w.numel() == 0 or float(w.detach().abs().max()) < 31.99)   # RN_f16(31.99) = 31.984375 = 65504 / 2^11
    return s


def unsplit(s, cols=None):
    C = s.shape[-1]
    cols = C if cols is None else cols
    t = s.t if s.t.is_contiguous() else s.t.contiguous()
    rows = t.numel() // (2 * C)
    out = torch.empty(*s.shape[:-1], cols, device=t.device, dtype=torch.float32)
    _lib.check(_lib.load().msmd_unsplit_f16x2(_p(t), _p(out), rows, cols, C, cols, _stream()), "msmd_unsplit_f16x2")
    return out
ACT_NONE, ACT_GELU, ACT_ELU = 0, 1, 2
CONV0_SPLITS = 16

# Optional per-launch timing of the dominant kernel (bench.py roofline leg): when a list is installed
# here, every msmd_gemm launch is bracketed by HIP events recorded on the launch stream.
GEMM_TRACE = None
# Optional FLOP counter (bench.py sampler leg): a one-element list that every msmd_gemm call adds 2 M N K batch to.
GEMM_FLOPS = None


def _dt(t: torch.Tensor) -> int:
    if t.dtype == SPLIT:
        return F16X2
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if isinstance(t, Split):
            t = t.t
        if t is not None and not t.is_cuda:
            raise RuntimeError("msmd_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")


# Host-side GEMM autotune: the first call of a (shape, epilogue) key times the LDS-DMA kernel variants on the real
# operands (all variants are bit-identical in their results) and remembers the winner; later calls pass it as a hint
# in bits 8-15 of `act`.  Never runs during hipGraph capture (it synchronises).  Opt-in (MSMD_GEMM_AUTOTUNE=1): on the
# bench workload the library's shape heuristic is within run-to-run noise of the tuned choice (5.20 vs 5.22 ms/step).
GEMM_AUTOTUNE = os.environ.get("MSMD_GEMM_AUTOTUNE", "0") == "1"
_TUNE_MIN_FLOP = 1.0e9
_TUNE_CANDIDATES = (17, 13, 9, 12)
_TUNED = {}
# Per-call GEMM knobs (include/msmd_hip.h: MSMD_GEMM_VARIANT / _WRITE_THROUGH / _PAIRED_STORES).  The C library has no
# global state; these module-level defaults are what `gemm()` passes when the caller gives none (`gemm_defaults`
# scopes a change, e.g. while a hipGraph is captured -- the choice is then baked into that graph).
GEMM_WRITE_THROUGH, GEMM_PAIRED_STORES, GEMM_STAGGER = 1 << 16, 1 << 17, 1 << 18
GEMM_ONE_TILE_PER_WORKGROUP = 1 << 19      # opt out of the persistent form of multi-round launches (A/B; same bits)
GEMM_NO_256_TILE = 1 << 20                 # opt out of the 256 x 256 8-phase kernel where the library would pick it (A/B; plain outputs same bits)
GEMM_W_BELOW_32 = 1 << 21                  # split operands: |W| < 32 everywhere (set by gemm() from Split.below_32)
GEMM_LN_FLAGS = 0              # extra `act` bits msmd_gemm_ln calls carry (A/B hook: GEMM_ONE_TILE_PER_WORKGROUP)
GEMM_LN_ROUTER = None          # optional (M, N, K) -> 15 | 17 | None: tile hint for msmd_gemm_ln's big-tile family
GEMM_LN_ALL_IN_ONE = False     # A/B hook (tools/ab_forward.py): msmd_gemm_ln on the one-kernel-with-every-epilogue form (variant 66)


class capture_guard:
    """Around every hipGraph stream capture of this package: collect garbage first and keep Python's cyclic collector off for
    the duration.  A collection that starts INSIDE a capture runs the destructors of whatever earlier code left in reference
    cycles -- captured graphs, events, communicators: device-runtime calls that are not permitted while a stream is capturing --
    and the process aborts (seen as a rare `Fatal Python error: Aborted ... Garbage-collecting` under the segmented training
    capture, round 4).  torch.cuda.graph() collects at entry too, but does not hold the collector off."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False
# attention launches also pull the layer's remaining weights through the memory-side cache (msmd_attention_prefetch)
PREFETCH_WEIGHTS = os.environ.get("MSMD_PREFETCH", "1") != "0"
# transformer blocks: LayerNorm folded into the neighbouring GEMMs (gemm_ln); False (MSMD_FOLD_LN=0) = LayerNorm kernels
FOLD_LN = os.environ.get("MSMD_FOLD_LN", "1") != "0"
GEMM_ROUTER = None   # developer hook (tools/ab_forward.py): callable (M, N, K, batch) -> variant or None, consulted per call
_GEMM_DEFAULT = {"variant": 0, "flags": GEMM_PAIRED_STORES, "split_variant": 0}   # paired 16-byte stores: -1 % on the forward step
if os.environ.get("MSMD_GEMM_ONE_TILE", "0") == "1":      # developers' A/B switch (tools/ab_env.sh): every launch in the one-tile-per-workgroup form
    _GEMM_DEFAULT["flags"] |= GEMM_ONE_TILE_PER_WORKGROUP
    GEMM_LN_FLAGS = GEMM_ONE_TILE_PER_WORKGROUP


if os.environ.get("MSMD_GEMM_NO_256", "0") == "1":        # developers' A/B switch: no launch on the 256 x 256 kernel
    _GEMM_DEFAULT["flags"] |= GEMM_NO_256_TILE
    GEMM_LN_FLAGS |= GEMM_NO_256_TILE


class gemm_defaults:
    """with ops.gemm_defaults(variant=13, flags=ops.GEMM_WRITE_THROUGH): ...   (None = leave as is)"""

    def __init__(self, variant=None, flags=None, split_variant=None):
        self.new = {k: v for k, v in (("variant", variant), ("flags", flags), ("split_variant", split_variant))
                    if v is not None}

    def __enter__(self):
        self.old = dict(_GEMM_DEFAULT)
        _GEMM_DEFAULT.update(self.new)
        return self

    def __exit__(self, *exc):
        _GEMM_DEFAULT.clear()
        _GEMM_DEFAULT.update(self.old)
        return False


def _autotune_gemm(lib, args, key, out, residual):
    if torch.cuda.is_current_stream_capturing() or (residual is not None and residual.data_ptr() == out.data_ptr()):
        return 0
    act = args[16]
    best, best_t = 0, float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for v in _TUNE_CANDIDATES:
        args[16] = act | (v << 8)
        if lib.msmd_gemm(*args) != 0:
            continue
        e0.record()
        for _ in range(3):
            lib.msmd_gemm(*args)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if t < best_t:
            best, best_t = v, t
    args[16] = act
    _TUNED[key] = best
    return best


def gemm(a, w, bias=None, residual=None, act=ACT_NONE, out=None, out_dtype=None, *, M=None, K=None, lda=None,
         rows_per_batch=0, a_batch_stride=0, batch=1, strideA=0, strideW=0, strideC=0, strideBias=0, strideR=0,
         N=None, ldw=None, ldc=None, z_out=None, p_drop=0.0, rng_state=None, site=0, variant=None, flags=None):
    """C = dropout_p(act(A @ W^T + bias)) + residual (p_drop = 0: no dropout); z_out (like C) receives the
    pre-activation A @ W^T + bias when given (training epilogue, msmd_gemm_ex).  a: (..., K) contiguous unless M/K/lda describe a windowed view;
    w: (N, K) (or (batch, N, K) with strideW).  Returns C with a's leading dims + (N,)."""
    _need_cuda(a, w, bias, residual)
    lib = _lib.load()
    if isinstance(w, Split) and not isinstance(a, Split):
        # parity-grade speed mode with an fp32 producer: convert here (hot producers hand over Split tensors directly).
        # Only whole contiguous tensors: the caller's lda / strides describe the ORIGINAL layout.
        if not a.is_contiguous() or a.shape[-1] % 32:
            raise ValueError("split GEMM: pass a contiguous fp32 tensor with a last dim % 32 == 0, or a Split")
        a = to_split(a)
    if isinstance(a, Split) and not isinstance(w, Split):
        raise TypeError("split activations need split weights")
    if isinstance(a, Split) and out_dtype is None and out is None:
        out_dtype = torch.float32
    if K is None:
        K = a.shape[-1]
    if M is None:
        M = a.numel() // K
    if N is None:
        N = w.shape[-2]
    if lda is None:
        lda = K
    if ldw is None:
        ldw = w.shape[-1]
    out_dtype = out_dtype or a.dtype
    if out is None:
        lead = a.shape[:-1] if a.numel() // a.shape[-1] == M and batch == 1 else (M,)
        out = empty((*lead, N), a.device, out_dtype)
    if ldc is None:
        ldc = N
    ldr = residual.stride(-2) if residual is not None and residual.dim() >= 2 else N
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be fp32")
    if GEMM_FLOPS is not None:
        GEMM_FLOPS[0] += 2.0 * M * N * K * batch
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if variant is None:
        variant = _GEMM_DEFAULT["split_variant"] if isinstance(a, Split) else _GEMM_DEFAULT["variant"]
        if GEMM_ROUTER is not None and not isinstance(a, Split):
            variant = GEMM_ROUTER(M, N, K, batch) or variant
    flags = _GEMM_DEFAULT["flags"] if flags is None else flags
    if isinstance(w, Split) and w.below_32:
        flags |= GEMM_W_BELOW_32
    act = act | (variant << 8) | flags
    args = [_p(a), _p(w), _p(bias), _p(residual), _p(out), M, N, K, _dt(a), _dt(out), lda, rows_per_batch,
            a_batch_stride, ldw, ldc, ldr, act, batch, strideA, strideW, strideC, strideBias, strideR, _stream()]
    if (GEMM_AUTOTUNE and variant == 0 and a.dtype == torch.bfloat16 and K % 64 == 0
            and 2.0 * M * N * K * batch >= _TUNE_MIN_FLOP):
        key = (M, N, K, batch, out.dtype, act, bias is not None, residual is not None, rows_per_batch > 0, lda, ldc)
        v = _TUNED.get(key)
        if v is None:
            v = _autotune_gemm(lib, args, key, out, residual)
        args[16] = act | (v << 8)
    if z_out is not None or p_drop > 0.0:
        _lib.check(lib.msmd_gemm_ex(*args[:-1], _p(z_out), float(p_drop), _p(rng_state), int(site), args[-1]),
                   "msmd_gemm_ex")
    else:
        _lib.check(lib.msmd_gemm(*args), "msmd_gemm")
    if GEMM_TRACE is not None:
        e1.record()
        # last field: did the library run this launch on its 256 x 256-tile kernel (what the kernel takes + the shape rule)
        if isinstance(a, Split):
            t256 = (variant in (0, 80) and not (flags & GEMM_NO_256_TILE) and batch == 1 and N % 256 == 0 and K % 32 == 0 and K >= 64
                    and (variant == 80 or bool(lib.msmd_gemm_256_tile_rule_f16x2(M, N, K, int(bool(flags & GEMM_W_BELOW_32))))))
        else:
            t256 = (variant in (0, 80) and not (flags & GEMM_NO_256_TILE) and batch == 1 and z_out is None and not p_drop > 0.0
                    and a.dtype in (torch.bfloat16, torch.float16) and out.dtype == a.dtype
                    and (variant == 80 or bool(lib.msmd_gemm_256_tile_rule(M, N, K))) and N % 256 == 0 and K % 64 == 0 and K >= 128)
        GEMM_TRACE.append((M, N, K, batch, _dt(a), e0, e1, t256))
    return out


GEMM_LN_TILE = None            # 15 | 17 | None: tile of msmd_gemm_ln's big-tile family for the calls made while it is set (the
                               # sampler sets 15 -- 192 x 128 -- around the capture of a multi-lane step graph, see sampler.py)


def gemm_ln(a, w, bias=None, residual=None, act=ACT_NONE, out=None, out_dtype=None, *, a_stats=None, w_colsum=None,
            r_stats=None, r_gamma=None, r_beta=None, stats_out=False, eps=1e-5):
    """C = act(LN_A(a) @ w^T + bias) + LN_R(residual) with the LayerNorms folded into the GEMM epilogue (msmd_gemm_ln):
    a_stats / r_stats are (cols / slab, M, 2) per-row partial (sum, sum of squares) written by a producer's stats_out;
    w must hold gamma-folded weights with w_colsum = their row sums and bias the beta-folded bias (see fold_layernorm).
    stats_out=True: returns (C, stats) with the partial statistics of the stored rows of C (the slab -- 64 or 32 columns --
    follows the tile the grid is routed to and is read back from the statistics' shape by the consumer)."""
    _need_cuda(a, w, bias, residual)
    lib = _lib.load()
    K = a.shape[-1]
    M = a.numel() // K
    N = w.shape[0]
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = empty((*a.shape[:-1], N), a.device, out_dtype)
    st, slab_out, slab_in = None, 0, 0
    if stats_out:
        # by the problem's N alone, never its row count: a clip's rows must get the same statistics (bit for bit) alone and
        # inside a large batch, so the producer is always the 128 x 128-tile family where that tile divides N
        slab_out = 64 if N % 128 == 0 else 32
        st = empty((N // slab_out, M, 2), a.device, torch.float32)
    sin = a_stats if a_stats is not None else r_stats
    if sin is not None:
        cols = K if a_stats is not None else N
        if sin.dim() != 3 or sin.shape[1] != M or sin.shape[2] != 2 or cols % sin.shape[0]:
            raise ValueError("gemm_ln: statistics must be (cols / slab, M, 2)")
        slab_in = cols // sin.shape[0]
    for t in (bias, w_colsum, r_gamma, r_beta, a_stats, r_stats):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError("gemm_ln: bias / colsum / gamma / beta / stats must be contiguous fp32")
    if GEMM_FLOPS is not None:
        GEMM_FLOPS[0] += 2.0 * M * N * K
    ldr = residual.stride(-2) if residual is not None and residual.dim() >= 2 else N
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.msmd_gemm_ln(_p(a), _p(w), _p(bias), _p(residual), _p(out), M, N, K, _dt(a), _dt(out), a.stride(-2) if a.dim() >= 2 else K,
                                w.stride(0), N, ldr, act | GEMM_LN_FLAGS | (((66 if GEMM_LN_ALL_IN_ONE else (GEMM_LN_ROUTER(M, N, K) or 0) if GEMM_LN_ROUTER is not None else (GEMM_LN_TILE or 0))) << 8), _p(a_stats), _p(w_colsum), _p(r_stats), _p(r_gamma),
                                _p(r_beta), _p(st), slab_in, slab_out, float(eps), _stream()), "msmd_gemm_ln")
    if GEMM_TRACE is not None:
        e1.record()
        hint = 66 if GEMM_LN_ALL_IN_ONE else ((GEMM_LN_ROUTER(M, N, K) or 0) if GEMM_LN_ROUTER is not None else (GEMM_LN_TILE or 0))
        t256 = ((hint == 80 or (hint == 0 and not (GEMM_LN_FLAGS & GEMM_NO_256_TILE) and bool(lib.msmd_gemm_256_tile_rule(M, N, K))))
                and N % 256 == 0 and K >= 128 and (st is None or slab_out == 64) and (sin is None or M % 2 == 0))
        GEMM_TRACE.append((M, N, K, 1, _dt(a), e0, e1, t256))
    return (out, st) if st is not None else out


def fold_layernorm(w, b, gamma, beta, dtype):
    """Weights of `Linear(LayerNorm(u))` for gemm_ln: (W' = dtype(W * gamma), colsum(W') fp32, b + W @ beta)."""
    w32, g32, be32 = w.float(), gamma.float(), beta.float()
    wf = (w32 * g32[None, :]).to(dtype).contiguous()
    cs = wf.float().sum(dim=1).contiguous()
    bf = (w32 @ be32 + (b.float() if b is not None else 0.0)).contiguous()
    return wf, cs, bf


def gemm_act_bwd(dy, wt, z, act, p_drop=0.0, rng_state=None, site=0):
    """dz = keep_mask / (1 - p) * act'(z) * (dy @ wt^T): msmd_gemm_actbwd (the data gradient of the Linear AFTER an
    activation + dropout, with their backward in its epilogue).  dy (..., K) 16-bit, wt (N, K) = the transposed weight
    cast, z (..., N) the forward's pre-activation."""
    _need_cuda(dy, wt, z)
    K = dy.shape[-1]
    M = dy.numel() // K
    N = wt.shape[0]
    if z.numel() != M * N or not z.is_contiguous() or not dy.is_contiguous():
        raise ValueError("gemm_act_bwd: z must be the contiguous (M, N) pre-activation")
    out = torch.empty_like(z)
    if GEMM_FLOPS is not None:
        GEMM_FLOPS[0] += 2.0 * M * N * K
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().msmd_gemm_actbwd(_p(dy), _p(wt), _p(z), _p(out), M, N, K, _dt(dy), _dt(out), K, wt.stride(0),
                                            act | (_GEMM_DEFAULT["flags"] & GEMM_PAIRED_STORES), float(p_drop), _p(rng_state),
                                            int(site), _stream()), "msmd_gemm_actbwd")
    if GEMM_TRACE is not None:
        e1.record()
        GEMM_TRACE.append((M, N, K, 1, _dt(dy), e0, e1, False))
    return out


def exp_set_tuning(key, value):
    """Developer builds only (make -C csrc EXP=1; MSMD_LIB=.../libmsmd_hip_exp.so): msmd_exp_set_tuning(key, value).
    The product library has no such switch -- use the per-call `variant` / `flags` / `splits` arguments."""
    lib = _lib.load()
    fn = getattr(lib, "msmd_exp_set_tuning", None)
    if fn is None:
        raise _lib.MsmdLibraryError("msmd_exp_set_tuning needs the experimental library (make -C csrc EXP=1, MSMD_LIB=...)")
    _lib.check(fn(int(key), int(value)), "msmd_exp_set_tuning")


g_tn_split = True


def gemm_tn(a, b, want_colsum=False, *, M=None, N=None, K=None, lda=None, ldb=None, batch=1, strideA=0, strideB=0,
            b_rows_per_window=0, b_window_stride=0, out=None, colsum_out=None, accumulate=False, splits=0):
    """C (N, K) fp32 = a^T @ b for bf16 a (M, N), b (M, K) (contraction over rows: the weight-gradient product, no
    transposes).  Returns C, or (C, colsum) with colsum[n] = sum_m a[m, n] (the bias gradient) when asked."""
    _need_cuda(a, b)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise TypeError("gemm_tn takes bf16 operands")
    lib = _lib.load()
    M = a.shape[-2] if M is None else M
    N = a.shape[-1] if N is None else N
    K = b.shape[-1] if K is None else K
    lda = a.stride(-2) if lda is None else lda
    ldb = b.stride(-2) if ldb is None else ldb
    if out is None:
        out = torch.empty((batch, N, K) if batch > 1 else (N, K), device=a.device, dtype=torch.float32)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != batch * N * K:
        raise ValueError("gemm_tn out must be a contiguous fp32 (N, K) tensor")
    cs = colsum_out if colsum_out is not None else (
        torch.empty(N, device=a.device, dtype=torch.float32) if want_colsum else None)
    nws = lib.msmd_gemm_tn_workspace(M, N, K, batch) if g_tn_split else 0
    ws = torch.empty(nws, device=a.device, dtype=torch.uint8) if nws > 0 else None
    _lib.check(lib.msmd_gemm_tn(_p(a), _p(b), _p(out), _p(cs), M, N, K, lda, ldb, K, batch, strideA, strideB, N * K,
                                b_rows_per_window, b_window_stride, int(bool(accumulate)) | (int(splits) << 8), _p(ws), nws,
                                _stream()), "msmd_gemm_tn")
    return (out, cs) if (want_colsum or colsum_out is not None) else out


def conv1d_cl(x, w_packed, bias=None, *, kernel, stride, act=ACT_NONE, out_dtype=None):
    """Strided Conv1d over a channels-last (B, T, C) signal as ONE windowed GEMM (no im2col).
    w_packed: (Cout, kernel*C) with K index = kk*C + c."""
    B, T, C = x.shape
    T_out = (T - kernel) // stride + 1
    out = empty((B, T_out, w_packed.shape[0]), x.device, out_dtype or x.dtype)
    gemm(x, w_packed, bias, None, act, out=out, M=B * T_out, K=kernel * C, lda=stride * C, rows_per_batch=T_out,
         a_batch_stride=T * C)
    return out


def layernorm(x, gamma, beta, residual=None, post_add=None, act=ACT_NONE, eps=1e-5, out_dtype=None, post_act=ACT_NONE,
              split=None):
    """y = post_act(LayerNorm(act(x + residual)) * gamma + beta) + post_add.
    split="both": returns (y fp32, y Split) from ONE pass; split="only": the Split alone (fp32 input, cols % 32 == 0)."""
    _need_cuda(x, gamma, beta)
    lib = _lib.load()
    cols = x.shape[-1]
    rows = x.numel() // cols
    if split:
        if x.dtype != torch.float32 or (residual is not None and residual.dtype != torch.float32):
            raise TypeError("layernorm(split=...) takes fp32 input")
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32) if split == "both" else None
        ys = empty(tuple(x.shape), x.device, SPLIT)
        _lib.check(lib.msmd_layernorm_f16x2(_p(x), _p(residual), _p(gamma), _p(beta), _p(post_add), _p(y), _p(ys), rows,
                                            cols, eps, act | (post_act << 8), _stream()), "msmd_layernorm_f16x2")
        return (y, ys) if split == "both" else ys
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    _lib.check(lib.msmd_layernorm(_p(x), _p(residual), _p(gamma), _p(beta), _p(post_add), _p(y), rows, cols, eps,
                                  act | (post_act << 8),
                                  _dt(x), _dt(y), _stream()), "msmd_layernorm")
    return y


def layernorm_pre(x, pre_gamma, pre_beta, residual, gamma, beta, eps=1e-5):
    """LN_{gamma, beta}(LN_{pre}(x) + residual) in one launch (16-bit rows): msmd_layernorm_pre."""
    _need_cuda(x, gamma, beta, pre_gamma, pre_beta)
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty_like(x)
    _lib.check(_lib.load().msmd_layernorm_pre(_p(x), _p(pre_gamma), _p(pre_beta), _p(residual), _p(gamma), _p(beta), _p(y),
                                              rows, cols, eps, _dt(x), _stream()), "msmd_layernorm_pre")
    return y


def dropout(x, p, rng_state, site, residual=None, out=None):
    """y = x * keep / (1 - p) (+ residual); keep = Philox(rng_state[seed, step], site, element / 4).  Calling it on
    the upstream gradient with the same (rng_state, site) is the backward."""
    _need_cuda(x, rng_state)
    x = x.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().msmd_dropout(_p(x), _p(residual), _p(out), x.numel(), float(p), _p(rng_state), int(site),
                                        _dt(x), _stream()), "msmd_dropout")
    return out


def _prefetch_ranges(tensors):
    """(pointer array, byte-count array, count) of the first four live, contiguous device tensors (Split: its storage)."""
    ts = []
    for t in tensors:
        t = t.t if isinstance(t, Split) else t
        if t is not None and t.is_cuda and t.is_contiguous():
            ts.append(t)
    ts = ts[:4]
    ptrs = (ctypes.c_void_p * 4)(*([t.data_ptr() for t in ts] + [None] * (4 - len(ts))))
    nbytes = (ctypes.c_long * 4)(*([t.numel() * t.element_size() for t in ts] + [0] * (4 - len(ts))))
    return ptrs, nbytes, len(ts)


def attention(q, k, v, n_heads, scale, mask=None, out=None, p_drop=0.0, rng_state=None, site=0, out_dtype=None,
              prefetch=None):
    """q: (B, Tq, H*64) view, k/v: (B, Tk, H*64) views (last dim contiguous; may be slices of a packed QKV).
    Split q / k / v (parity-grade speed mode): msmd_attention_f16x2; out_dtype SPLIT (default) or torch.float32."""
    _need_cuda(q, k, v)
    lib = _lib.load()
    B, Tq, d = q.shape
    Tk = k.shape[1]
    assert d == n_heads * 64 and q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
    if isinstance(q, Split):
        if not (isinstance(k, Split) and isinstance(v, Split)) or p_drop > 0.0:
            raise TypeError("split attention: q, k, v must all be Split (inference only)")
        if out is None:
            out = empty((B, Tq, d), q.device, out_dtype or SPLIT)
        m = None
        if mask is not None:
            assert mask.dtype in (torch.bool, torch.uint8) and mask.shape == (Tq, Tk) and mask.is_contiguous()
            m = mask
        _lib.check(lib.msmd_attention_f16x2(_p(q), _p(k), _p(v), _p(out), B, n_heads, Tq, Tk, q.stride(0), q.stride(1),
                                            k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0),
                                            out.stride(1), float(scale), _p(m), _dt(out), _stream()),
                   "msmd_attention_f16x2")
        return out
    if out is None:
        out = torch.empty(B, Tq, d, device=q.device, dtype=q.dtype)
    m = None
    if mask is not None:
        assert mask.dtype in (torch.bool, torch.uint8) and mask.shape == (Tq, Tk) and mask.is_contiguous()
        m = mask
    if p_drop > 0.0 and prefetch and PREFETCH_WEIGHTS:
        ptrs, nbytes, n = _prefetch_ranges(prefetch)
        _lib.check(lib.msmd_attention_dropout_prefetch(_p(q), _p(k), _p(v), _p(out), B, n_heads, Tq, Tk, q.stride(0),
                                                       q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                                       out.stride(0), out.stride(1), float(scale), _p(m), float(p_drop),
                                                       _p(rng_state), int(site), _dt(q), ptrs, nbytes, n, _stream()),
                   "msmd_attention_dropout_prefetch")
        return out
    if p_drop > 0.0:
        _lib.check(lib.msmd_attention_dropout(_p(q), _p(k), _p(v), _p(out), B, n_heads, Tq, Tk, q.stride(0),
                                              q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                              out.stride(0), out.stride(1), float(scale), _p(m), float(p_drop),
                                              _p(rng_state), int(site), _dt(q), _stream()), "msmd_attention_dropout")
        return out
    if prefetch and PREFETCH_WEIGHTS:
        # up to four tensors (the weights of the GEMMs that follow) pulled through the memory-side cache by this launch
        ptrs, nbytes, n = _prefetch_ranges(prefetch)
        _lib.check(lib.msmd_attention_prefetch(_p(q), _p(k), _p(v), _p(out), B, n_heads, Tq, Tk, q.stride(0), q.stride(1),
                                               k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0),
                                               out.stride(1), float(scale), _p(m), _dt(q), ptrs, nbytes, n,
                                               _stream()), "msmd_attention_prefetch")
        return out
    _lib.check(lib.msmd_attention(_p(q), _p(k), _p(v), _p(out), B, n_heads, Tq, Tk, q.stride(0), q.stride(1),
                                  k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                  float(scale), _p(m), _dt(q), _stream()), "msmd_attention")
    return out


def attention_bwd(q, k, v, do, dq, dk, dv, n_heads, scale, mask=None, p_drop=0.0, rng_state=None, site=0):
    """Fused backward of `attention` (bf16, Tk <= 256): fills dq / dk / dv (views with last dim contiguous)."""
    _need_cuda(q, k, v, do, dq, dk, dv)
    lib = _lib.load()
    B, Tq, d = q.shape
    Tk = k.shape[1]
    for t in (q, k, v, do, dq, dk, dv):
        if t.dtype != torch.bfloat16 or t.stride(-1) != 1:
            raise TypeError("attention_bwd takes bf16 tensors with a contiguous last dim")
    m = None
    if mask is not None:
        assert mask.dtype in (torch.bool, torch.uint8) and mask.shape == (Tq, Tk) and mask.is_contiguous()
        m = mask
    _lib.check(lib.msmd_attention_bwd(_p(q), _p(k), _p(v), _p(do), _p(dq), _p(dk), _p(dv), B, n_heads, Tq, Tk,
                                      q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                      do.stride(0), do.stride(1), dq.stride(0), dq.stride(1), dk.stride(0),
                                      dk.stride(1), dv.stride(0), dv.stride(1), float(scale), _p(m), float(p_drop),
                                      _p(rng_state), int(site), _stream()),
               "msmd_attention_bwd")


def dynamic_threshold_(res, L, ratio, dt_min, dt_max):
    """In-place dynamic thresholding of a (N, T_all, C) fp32 denoiser output on its last L frames' quantile."""
    _need_cuda(res)
    if res.dtype != torch.float32 or not res.is_contiguous():
        raise TypeError("dynamic_threshold_ takes a contiguous fp32 tensor")
    N, T_all, C = res.shape
    _lib.check(_lib.load().msmd_dynamic_threshold(_p(res), N, T_all, L, C, float(ratio), float(dt_min), float(dt_max),
                                                  _stream()), "msmd_dynamic_threshold")
    return res


def pad_audio(audio, reflect_len, replicate_len):
    lib = _lib.load()
    B, L = audio.shape
    out = torch.empty(B, L + 4 * reflect_len + 2 * replicate_len, device=audio.device, dtype=torch.float32)
    _lib.check(lib.msmd_pad_audio(_p(audio), _p(out), B, L, reflect_len, replicate_len, _stream()), "msmd_pad_audio")
    return out


def conv0_gn_gelu(audio, w0, gamma, beta, reflect_len, replicate_len, out_dtype, eps=1e-5):
    """GELU(GroupNorm(conv0(pad_audio(audio)))) -> (B, T0, C) channels-last."""
    lib = _lib.load()
    B, L = audio.shape
    C = w0.shape[0]
    Lp = L + 4 * reflect_len + 2 * replicate_len
    T0 = (Lp - 10) // 5 + 1
    stats = torch.empty(B, C, 2, device=audio.device, dtype=torch.float32)
    ws = torch.empty(B, CONV0_SPLITS, 66, device=audio.device, dtype=torch.float32)
    _lib.check(lib.msmd_conv0_stats(_p(audio), _p(w0), _p(stats), _p(ws), B, L, reflect_len, replicate_len, C, eps,
                                    _stream()), "msmd_conv0_stats")
    out = empty((B, T0, C), audio.device, out_dtype)
    _lib.check(lib.msmd_conv0_gn_gelu(_p(audio), _p(w0), _p(stats), _p(gamma), _p(beta), _p(out), B, L, reflect_len,
                                      replicate_len, C, _dt(out), _stream()), "msmd_conv0_gn_gelu")
    return out


def conv0_ln_gelu(audio, w0, bias, gamma, beta, reflect_len, replicate_len, out_dtype, eps=1e-5):
    """GELU(LayerNorm_channels(conv0(pad_audio(audio)) + bias)) -> (B, T0, 512) (feat_extract_norm="layer" stacks)."""
    lib = _lib.load()
    B, L = audio.shape
    C = w0.shape[0]
    Lp = L + 4 * reflect_len + 2 * replicate_len
    T0 = (Lp - 10) // 5 + 1
    out = torch.empty(B, T0, C, device=audio.device, dtype=out_dtype)
    _lib.check(lib.msmd_conv0_ln_gelu(_p(audio), _p(w0), _p(bias), _p(gamma), _p(beta), _p(out), B, L, reflect_len,
                                      replicate_len, C, eps, _dt(out), _stream()), "msmd_conv0_ln_gelu")
    return out


def interp_linear(x, t_out, t_crop=None):
    lib = _lib.load()
    B, T, C = x.shape
    t_crop = T if t_crop is None else min(int(t_crop), T)  # slicing past the end clamps (utils/wav2vec2.py:83)
    y = torch.empty(B, t_out, C, device=x.device, dtype=x.dtype)
    _lib.check(lib.msmd_interp_linear(_p(x), _p(y), B, T, t_crop, t_out, C, _dt(x), _stream()), "msmd_interp_linear")
    return y


def group_pad(x, groups, pad, cg_out=None, split=False):
    """(B, T, G*Cg) -> zero-padded group-major (B, G, T + 2 pad, cg_out >= Cg); split=True writes MSMD_F16X2 rows
    (fp32 input, cg_out % 32 == 0)."""
    lib = _lib.load()
    B, T, C = x.shape
    cg = C // groups
    cg_out = cg if cg_out is None else cg_out
    y = empty((B, groups, T + 2 * pad, cg_out), x.device, SPLIT if split else x.dtype)
    _lib.check(lib.msmd_group_pad(_p(x), _p(y), B, T, groups, cg, cg_out, pad, _dt(x), _dt(y), _stream()),
               "msmd_group_pad")
    return y


def denoiser_pack_input(motion, prev_motion, indicator, feats, eps=None, c0=None, c1=None):
    lib = _lib.load()
    N, Tn, Kpad = feats.shape
    L, dm = motion.shape[1], motion.shape[2]
    Lp = prev_motion.shape[1]
    assert Tn == 1 + Lp + L
    _lib.check(lib.msmd_denoiser_pack_input(_p(motion), _p(eps), _p(c0), _p(c1), _p(prev_motion), _p(indicator),
                                            _p(feats), N, L, Lp, dm, Kpad, motion.shape[0], _dt(feats), _stream()),
               "msmd_denoiser_pack_input")
    return feats


def add_pe_token(x, pe, tok0, row0_add=None):
    lib = _lib.load()
    N, T, d = x.shape
    _lib.check(lib.msmd_add_pe_token(_p(x), _p(pe), _p(tok0), _p(row0_add), N, T, d, _dt(x), _stream()),
               "msmd_add_pe_token")
    return x


def heads_static_mix(dec, stat, L, dm, nb, use_head_alpha=False, sigmoid_alpha=False):
    lib = _lib.load()
    N = dec.shape[0]
    out = torch.empty(N, L, dm, device=dec.device, dtype=torch.float32)
    _lib.check(lib.msmd_heads_static_mix(_p(dec), dec.stride(1), _p(stat), _p(out), N, L, dm, nb, stat.shape[0],
                                         int(bool(use_head_alpha)) | (2 if sigmoid_alpha else 0), _dt(dec), _stream()),
               "msmd_heads_static_mix")
    return out


def cfg_ddpm_step(x, res, z, scales, n_entries, Lp, mode, target, c0, c1, sigma):
    lib = _lib.load()
    B, L, dm = x.shape
    _lib.check(lib.msmd_cfg_ddpm_step(_p(x), _p(res), _p(z), _p(scales), n_entries, B, L, Lp, dm, mode, target,
                                      float(c0), float(c1), float(sigma), _stream()), "msmd_cfg_ddpm_step")
    return x


def sampler_step_select(emb_all, coef_table, t_dev, emb_row, coefs):
    lib = _lib.load()
    _lib.check(lib.msmd_sampler_step_select(_p(emb_all), _p(coef_table), _p(t_dev), _p(emb_row), _p(coefs),
                                            emb_all.shape[-1], _dt(emb_all), _stream()), "msmd_sampler_step_select")


def cfg_ddpm_step_dev(x, res, z, scales, coefs, n_entries, Lp, mode, target):
    lib = _lib.load()
    B, L, dm = x.shape
    _lib.check(lib.msmd_cfg_ddpm_step_dev(_p(x), _p(res), _p(z), _p(scales), _p(coefs), n_entries, B, L, Lp, dm, mode,
                                          target, _stream()), "msmd_cfg_ddpm_step_dev")
    return x


def pad_cols(x, cols_out, out_dtype=None):
    lib = _lib.load()
    cols_in = x.shape[-1]
    rows = x.numel() // cols_in
    y = torch.empty(*x.shape[:-1], cols_out, device=x.device, dtype=out_dtype or x.dtype)
    _lib.check(lib.msmd_pad_cols(_p(x), _p(y), rows, cols_in, cols_out, _dt(x), _dt(y), _stream()), "msmd_pad_cols")
    return y


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    return pad_cols(x.contiguous(), x.shape[-1], dtype)


def mean_time(x):
    lib = _lib.load()
    B, T, C = x.shape
    y = torch.empty(B, C, device=x.device, dtype=torch.float32)
    _lib.check(lib.msmd_mean_time(_p(x), _p(y), B, T, C, _dt(x), _stream()), "msmd_mean_time")
    return y


# ----------------------------------------------------------------------------- FLAME
SKIN_TILE_BYTES = 18432   # msmd_lbs_skin_v2's input record per 16 frames (include/msmd_hip.h)


def lbs_prepare(betas, pose, JS, parents, Kp=192, pose_is_matrix=False, want_joints=True, want_split=False,
                want_blend_tiles=False):
    """-> coef, coef_hl, A, joints[, skin_tiles]: want_blend_tiles adds msmd_lbs_skin_v2's 18 KB-per-16-frames records
    (coefficients as bf16 hi / lo + fp16 blend rows); want_split the row-major coef_hl of msmd_lbs_skin_bf16x3."""
    lib = _lib.load()
    B, NB = betas.shape
    J = parents.shape[0]
    coef = torch.empty(B, Kp, device=betas.device, dtype=torch.float32)
    coef_hl = torch.empty(B, 2, Kp, device=betas.device, dtype=torch.bfloat16) if want_split else None
    A = torch.empty(B, J, 12, device=betas.device, dtype=torch.float32)
    joints = torch.empty(B, J, 3, device=betas.device, dtype=torch.float32) if want_joints else None
    at = torch.empty((B + 15) // 16, SKIN_TILE_BYTES // 2, device=betas.device, dtype=torch.float16) if want_blend_tiles else None
    _lib.check(lib.msmd_lbs_prepare(_p(betas), _p(pose), _p(JS), _p(parents), _p(coef), _p(coef_hl), _p(A), _p(joints),
                                    _p(at), B, NB, J, Kp, int(pose_is_matrix), _stream()), "msmd_lbs_prepare")
    return (coef, coef_hl, A, joints, at) if want_blend_tiles else (coef, coef_hl, A, joints)


def flame_prepare(shape, expr, pose6, eye, JS, parents, ignore_global_rot=False, dirs=None, v_template_planes=None):
    """FLAME.forward's (B, NS) shape, (B, NE) expression, (B, 6) [global | jaw] pose (+ optional (B, 6) eye pose) ->
    msmd_lbs_skin_v2's tile records, without the concatenated betas / full_pose tensors.  With `dirs` (3, 192, Vp) and the
    template planes also -> (shape_varies flag, folded template) for the one-subject fast path of lbs_skin_v2."""
    lib = _lib.load()
    _need_cuda(shape, expr, pose6)
    B = shape.shape[0]
    tiles = torch.empty((B + 15) // 16, SKIN_TILE_BYTES // 2, device=shape.device, dtype=torch.float16)
    flag = folded = None
    Vp = 0
    if dirs is not None and v_template_planes is not None:
        Vp = dirs.shape[-1]
        flag = torch.empty(1, device=shape.device, dtype=torch.int32)
        folded = torch.empty(3, Vp, device=shape.device, dtype=torch.float32)
    _lib.check(lib.msmd_flame_prepare(_p(shape), _p(expr), _p(pose6), _p(eye), _p(JS), _p(parents), None, None, None,
                                      _p(tiles), B, shape.shape[1], expr.shape[1], int(ignore_global_rot), _p(flag),
                                      _p(folded), _p(dirs), _p(v_template_planes), Vp, _stream()), "msmd_flame_prepare")
    return tiles, flag, folded


def lbs_skin(coef, A, v_template_planes, dirs, weight_planes, V):
    lib = _lib.load()
    B, Kp = coef.shape
    J = A.shape[1]
    Vp = dirs.shape[-1]
    verts = torch.empty(B, V, 3, device=coef.device, dtype=torch.float32)
    _lib.check(lib.msmd_lbs_skin(_p(coef), _p(A), _p(v_template_planes), _p(dirs), _p(weight_planes), _p(verts), B, J,
                                 V, Vp, Kp, _stream()), "msmd_lbs_skin")
    return verts


def lbs_skin_bf16x3(coef_hl, A, v_template_planes, dirs_hl, weight_planes, V):
    lib = _lib.load()
    B, _, Kp = coef_hl.shape
    J = A.shape[1]
    Vp = dirs_hl.shape[-2]
    verts = torch.empty(B, V, 3, device=A.device, dtype=torch.float32)
    _lib.check(lib.msmd_lbs_skin_bf16x3(_p(coef_hl), _p(A), _p(v_template_planes), _p(dirs_hl), _p(weight_planes),
                                        _p(verts), B, J, V, Vp, Kp, _stream()), "msmd_lbs_skin_bf16x3")
    return verts


def lbs_skin_v2(skin_tiles, B, v_template_planes, dirs_hl, weight_planes, V, Kp=192, shape_varies=None, folded=None,
                out_dtype=torch.float32, dirs_f16=None):
    """out_dtype=torch.float16 (opt-in, msmd_lbs_skin_v2_f16): the vertices as fp16 -- a (B, V, 3) VIEW of rows padded to an
    even vertex count (row stride (V + V % 2) * 3); half the store stream.  With dirs_f16 (ONE fp16 plane of the blendshape
    directions, LbsConstants.dirs_f16) the product runs on fp16 operand planes (1 MFMA per K group instead of 3; the tile
    records are converted by msmd_lbs_tiles_f16 first): |error| <= 2^-11 |v| + 2^-10 sum_k |coef_k| |dirs_k|; without it the fp32 kernel's arithmetic,
    rounded once at the store."""
    lib = _lib.load()
    assert skin_tiles.numel() * skin_tiles.element_size() == ((B + 15) // 16) * SKIN_TILE_BYTES
    J = weight_planes.shape[0]
    Vp = dirs_hl.shape[-2]
    if out_dtype == torch.float16:
        V_ld = V + (V & 1)
        v16 = torch.empty(B, V_ld, 3, device=skin_tiles.device, dtype=torch.float16)
        tiles, dirs, single = skin_tiles, dirs_hl, 0
        if dirs_f16 is not None:
            tiles = torch.empty((B + 15) // 16, 12288 // 2, device=skin_tiles.device, dtype=torch.float16)
            _lib.check(lib.msmd_lbs_tiles_f16(_p(skin_tiles), _p(tiles), B, _stream()), "msmd_lbs_tiles_f16")
            dirs, single = dirs_f16, 1
        _lib.check(lib.msmd_lbs_skin_v2_f16(_p(tiles), _p(v_template_planes), _p(dirs), _p(weight_planes), _p(v16), B, J, V,
                                            V_ld, Vp, Kp, _p(shape_varies), _p(folded), single, _stream()), "msmd_lbs_skin_v2_f16")
        return v16[:, :V]
    if out_dtype != torch.float32:
        raise TypeError("lbs_skin_v2: fp32 or fp16 vertices")
    verts = torch.empty(B, V, 3, device=skin_tiles.device, dtype=torch.float32)
    _lib.check(lib.msmd_lbs_skin_v2(_p(skin_tiles), _p(v_template_planes), _p(dirs_hl), _p(weight_planes),
                                    _p(verts), B, J, V, Vp, Kp, _p(shape_varies), _p(folded), _stream()),
               "msmd_lbs_skin_v2")
    return verts


def lbs_pack(coef, A, want_split=False):
    """(coef (B, Kp), A (B, 5, 12)) fp32 -> skin_tiles (, coef_hl): the skinning kernels' operand formats."""
    lib = _lib.load()
    B, Kp = coef.shape
    coef_hl = torch.empty(B, 2, Kp, device=coef.device, dtype=torch.bfloat16) if want_split else None
    at = torch.empty((B + 15) // 16, SKIN_TILE_BYTES // 2, device=coef.device, dtype=torch.float16)
    _lib.check(lib.msmd_lbs_pack(_p(coef), _p(A), _p(coef_hl), _p(at), B, Kp, _stream()), "msmd_lbs_pack")
    return (at, coef_hl) if want_split else at


def lbs_skin_v2_train(skin_tiles, B, v_template_planes, dirs_hl, weight_planes, V, Kp=192):
    """-> (verts, v_posed): msmd_lbs_skin_v2 that also stores the un-skinned vertices (needed by lbs_skin_bwd)."""
    lib = _lib.load()
    assert skin_tiles.numel() * skin_tiles.element_size() == ((B + 15) // 16) * SKIN_TILE_BYTES
    J = weight_planes.shape[0]
    Vp = dirs_hl.shape[-2]
    verts = torch.empty(B, V, 3, device=skin_tiles.device, dtype=torch.float32)
    vposed = torch.empty(B, V, 3, device=skin_tiles.device, dtype=torch.float32)
    _lib.check(lib.msmd_lbs_skin_v2_train(_p(skin_tiles), _p(v_template_planes), _p(dirs_hl),
                                          _p(weight_planes), _p(verts), _p(vposed), B, J, V, Vp, Kp, _stream()),
               "msmd_lbs_skin_v2_train")
    return verts, vposed


def lbs_skin_bwd(grad_verts, vposed, A, weight_planes):
    """-> (dp planes (B, 3, Vp), dA (B, 5, 12)) of the skinning's backward."""
    lib = _lib.load()
    B, V, _ = grad_verts.shape
    J, Vp = weight_planes.shape
    dp = torch.empty(B, 3, Vp, device=grad_verts.device, dtype=torch.float32)
    dA = torch.empty(B, J, 12, device=grad_verts.device, dtype=torch.float32)
    _lib.check(lib.msmd_lbs_skin_bwd(_p(grad_verts), _p(vposed), _p(A), _p(weight_planes), _p(dp), _p(dA), B, J, V, Vp,
                                     _stream()), "msmd_lbs_skin_bwd")
    return dp, dA


def landmarks(verts, faces_i32, lmk_faces_idx_i32, bary):
    """lmk_faces_idx: (L,) / (1, L) shared or (B, L) per frame; bary likewise (.., L, 3)."""
    lib = _lib.load()
    B, V, _ = verts.shape
    idx = lmk_faces_idx_i32.reshape(-1, lmk_faces_idx_i32.shape[-1])
    bc = bary.reshape(-1, bary.shape[-2], 3)
    L = idx.shape[1]
    out = torch.empty(B, L, 3, device=verts.device, dtype=torch.float32)
    _lib.check(lib.msmd_landmarks(_p(verts), _p(faces_i32), _p(idx), L if idx.shape[0] > 1 else 0, _p(bc),
                                  L * 3 if bc.shape[0] > 1 else 0, _p(out), B, V, L, _stream()), "msmd_landmarks")
    return out


def dynamic_lmk_row(full_pose, neck_chain_i32, pose_is_matrix=False):
    lib = _lib.load()
    B = full_pose.shape[0]
    J = full_pose.shape[1] // (9 if pose_is_matrix else 3)
    row = torch.empty(B, device=full_pose.device, dtype=torch.int32)
    _lib.check(lib.msmd_dynamic_lmk_row(_p(full_pose), _p(neck_chain_i32), neck_chain_i32.shape[0], _p(row), B, J,
                                        int(pose_is_matrix), _stream()), "msmd_dynamic_lmk_row")
    return row


def batch_rodrigues(rot_vecs):
    lib = _lib.load()
    N = rot_vecs.shape[0]
    R = torch.empty(N, 3, 3, device=rot_vecs.device, dtype=torch.float32)
    _lib.check(lib.msmd_batch_rodrigues(_p(rot_vecs), _p(R), N, _stream()), "msmd_batch_rodrigues")
    return R


def rotation_convert(op, x, in_width, out_shape_tail, x2=None, conv=0):
    """Generic elementwise rotation conversion: x (..., in_width[, in_width2]) -> (..., *out_shape_tail)."""
    lib = _lib.load()
    x = x.contiguous().float()
    n = x.numel() // in_width
    lead = x.shape[:-1] if in_width in (3, 4, 6) else x.shape[:-2]
    out = torch.empty(*lead, *out_shape_tail, device=x.device, dtype=torch.float32)
    if x2 is not None:
        x2 = x2.contiguous().float()
    _lib.check(lib.msmd_rotation_convert(op, _p(x), _p(x2), _p(out), n, conv, _stream()), "msmd_rotation_convert")
    return out


# ----------------------------------------------------------------------------- losses / training pieces
def masked_seq_loss(gt, pred, end_idx, c_lo, c_hi, order, prefix, criterion=0, mode=0, scale=1.0, return_ws=False):
    """0-dim fp32 tensor = scale * masked mean (see msmd_masked_seq_loss in include/msmd_hip.h); return_ws also hands
    back the workspace (valid-row count) that masked_seq_loss_bwd_ needs."""
    lib = _lib.load()
    _need_cuda(gt, pred)
    N, T, C = pred.shape
    out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = torch.empty(2, device=pred.device, dtype=torch.float64)
    _lib.check(lib.msmd_masked_seq_loss(_p(gt), _p(pred), _p(end_idx), _p(out), _p(ws), N, T, C, c_lo, c_hi, order,
                                        prefix, criterion, mode, float(scale), _stream()), "msmd_masked_seq_loss")
    return (out[0], ws) if return_ws else out[0]


def masked_seq_loss_bwd_(grad_pred, gt, pred, end_idx, ws, upstream, c_lo, c_hi, order, prefix, criterion=0, mode=0,
                         scale=1.0):
    """grad_pred += d(masked_seq_loss) / d(pred) * upstream (0-dim device tensor)."""
    lib = _lib.load()
    N, T, C = pred.shape
    up = upstream.reshape(1).float().contiguous()
    _lib.check(lib.msmd_masked_seq_loss_bwd(_p(gt), _p(pred), _p(end_idx), _p(ws), _p(up), _p(grad_pred), N, T, C, c_lo,
                                            c_hi, order, prefix, criterion, mode, float(scale), _stream()),
               "msmd_masked_seq_loss_bwd")
    return grad_pred


def kl_loss(mu, logvar):
    lib = _lib.load()
    out = torch.empty(1, device=mu.device, dtype=torch.float32)
    ws = torch.empty(2, device=mu.device, dtype=torch.float64)
    _lib.check(lib.msmd_kl_loss(_p(mu), _p(logvar), _p(out), _p(ws), mu.numel(), _stream()), "msmd_kl_loss")
    return out[0]


def truncate_rows_(x, end_idx_i32, unit=1, replicate=False):
    """In place on x (N, L[, inner])."""
    lib = _lib.load()
    N, L = x.shape[0], x.shape[1]
    inner = x.numel() // (N * L)
    _lib.check(lib.msmd_truncate_rows(_p(x), _p(end_idx_i32), N, L, inner, unit, int(replicate), _stream()),
               "msmd_truncate_rows")
    return x


def adam_step_(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    lib = _lib.load()
    _lib.check(lib.msmd_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr),
                                  float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream()),
               "msmd_adam_step")
    return param


# ----------------------------------------------------------------------------- backward building blocks
def gemm_batched2(a, w, out, M, N, K, lda, ldw, ldc, batch_outer, sA_o, sW_o, sC_o, batch_inner, sA_i, sW_i, sC_i):
    lib = _lib.load()
    _lib.check(lib.msmd_gemm_batched2(_p(a), _p(w), _p(out), M, N, K, _dt(a), _dt(out), lda, ldw, ldc, batch_outer,
                                      sA_o, sW_o, sC_o, batch_inner, sA_i, sW_i, sC_i, _stream()), "msmd_gemm_batched2")
    return out


def transpose(x, out, rows, cols, ldx, ldy, batch=1, sx=0, sy=0, batch_inner=1, sx_i=0, sy_i=0):
    lib = _lib.load()
    _lib.check(lib.msmd_transpose(_p(x), _p(out), rows, cols, ldx, ldy, batch, sx, sy, batch_inner, sx_i, sy_i, _dt(x),
                                  _stream()), "msmd_transpose")
    return out


def transpose2d(x, pad_to=8):
    """(rows, cols) -> (cols, rows_padded) with zero padding of the new inner dim to a multiple of `pad_to`."""
    rows, cols = x.shape
    rp = (rows + pad_to - 1) // pad_to * pad_to
    out = torch.zeros(cols, rp, device=x.device, dtype=x.dtype) if rp != rows else \
        torch.empty(cols, rp, device=x.device, dtype=x.dtype)
    return transpose(x, out, rows, cols, x.stride(0), rp)


def colsum(x2d, out=None, accumulate=False):
    """out[c] (+)= sum_r x2d[r, c] in fp32, deterministic (per-row-block partial sums added in block order)."""
    lib = _lib.load()
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty(cols, device=x2d.device, dtype=torch.float32)
    nws = lib.msmd_colsum_workspace(rows, cols)
    ws = torch.empty(nws, device=x2d.device, dtype=torch.uint8)
    _lib.check(lib.msmd_colsum(_p(x2d), _p(out), rows, cols, x2d.stride(0), int(accumulate), _dt(x2d), _p(ws), nws, _stream()),
               "msmd_colsum")
    return out


def fold_weight_norm(g, v):
    """weight_norm(dim=2) of the positional conv, no autograd: w[o, i, k] = g[k] v[o, i, k] / ||v[:, :, k]||.  The norm
    is a column sum by msmd_colsum: the host library's multi-block reduction returned NaN from the second replay on when it
    sat inside a segmented hipGraph, so no pack / training graph uses it."""
    O, I, K = v.shape
    v2 = v.detach().reshape(O * I, K).float()
    s = g.detach().reshape(K).float() * torch.rsqrt(colsum((v2 * v2).contiguous()))
    return v.detach().float() * s.view(1, 1, K)


def act_fwd(z, act):
    lib = _lib.load()
    y = torch.empty_like(z)
    _lib.check(lib.msmd_act_fwd(_p(z), _p(y), z.numel(), act, _dt(z), _stream()), "msmd_act_fwd")
    return y


def person_query_attention(x, wq, bq, kv, n_heads, scale, wq_colsum=None, eps=1e-5):
    """Row 0 of every sequence of x (N, T, d): softmax(scale (x0 Wq^T + bq)_h K_h^T) V_h against kv (N, Tk, 2d) =
    [K | V]; returns (N, d).  One launch for the projection and the Tq = 1 attention (msmd_person_query_attention)."""
    _need_cuda(x, wq, kv)
    N, _, d = x.shape
    Tk = kv.shape[1]
    if kv.shape[2] != 2 * d or wq.shape[0] != d or wq.shape[1] != d or x.stride(2) != 1 or kv.stride(2) != 1:
        raise ValueError("person_query_attention: shapes")
    if not (x.dtype == wq.dtype == kv.dtype):
        raise TypeError("person_query_attention: x, wq, kv must share a dtype")
    out = torch.empty(N, d, device=x.device, dtype=x.dtype)
    if wq_colsum is not None:      # the LayerNorm in front of the projection folded in (wq / bq folded, x un-normalised)
        _lib.check(_lib.load().msmd_person_query_attention_ln(_p(x), x.stride(0), _p(wq), _p(bq), _p(wq_colsum), float(eps),
                                                              _p(kv), _p(kv[..., d:]), kv.stride(0), kv.stride(1), _p(out),
                                                              N, n_heads, Tk, d, float(scale), _dt(x), _stream()),
                   "msmd_person_query_attention_ln")
        return out
    _lib.check(_lib.load().msmd_person_query_attention(_p(x), x.stride(0), _p(wq), _p(bq), _p(kv), _p(kv[..., d:]),
                                                       kv.stride(0), kv.stride(1), _p(out), N, n_heads, Tk, d,
                                                       float(scale), _dt(x), _stream()), "msmd_person_query_attention")
    return out


def cast_transpose_multi(flat, meta, n, tiles, cast_arena, t_arena):
    """One launch: bf16 cast + bf16 transpose of n (N, K) matrices of the flat fp32 arena (meta: see msmd_hip.h)."""
    _need_cuda(flat, meta, cast_arena, t_arena)
    _lib.check(_lib.load().msmd_cast_transpose_multi(_p(flat), _p(meta), int(n), int(tiles), _p(cast_arena),
                                                     _p(t_arena), _stream()), "msmd_cast_transpose_multi")


def act_bwd_dropout(dy, z, act, p, rng_state, site):
    """dz = dropout_mask(dy) * act'(z) (one pass; mask of the forward msmd_gemm_ex / msmd_dropout)."""
    dz = torch.empty_like(z)
    _lib.check(_lib.load().msmd_act_bwd_dropout(_p(dy), _p(z), _p(dz), z.numel(), act, float(p), _p(rng_state),
                                                int(site), _dt(z), _stream()), "msmd_act_bwd_dropout")
    return dz


def act_bwd(dy, z, act):
    lib = _lib.load()
    dz = torch.empty_like(z)
    _lib.check(lib.msmd_act_bwd(_p(dy), _p(z), _p(dz), z.numel(), act, _dt(z), _stream()), "msmd_act_bwd")
    return dz


def layernorm_bwd(dy, x, gamma, eps=1e-5, dg_out=None, db_out=None, drop=None):
    """dx, dgamma, dbeta.  With dg_out / db_out (fp32, cols) the parameter gradients are ACCUMULATED into those
    buffers (e.g. the parameters' .grad views) instead of fresh tensors.  drop = (p, rng_state, site): the launch also
    writes dropout_mask(dx) / (1 - p) (what the Linear in front of the LayerNorm wants); returned as a 4th value."""
    lib = _lib.load()
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty_like(x)
    if dg_out is None:
        dgb = torch.zeros(2, cols, device=x.device, dtype=torch.float32)
        dg, db = dgb[0], dgb[1]
    else:
        dg, db = dg_out, db_out
    nws = lib.msmd_layernorm_bwd_workspace(rows, cols)
    ws = torch.empty(nws, device=x.device, dtype=torch.uint8)
    if drop is not None:
        p_drop, rng_state, site = drop
        dxd = torch.empty_like(x)
        _lib.check(lib.msmd_layernorm_bwd_dropout(_p(dy), _p(x), _p(gamma), _p(dx), _p(dxd), _p(dg), _p(db), rows, cols, eps,
                                                  float(p_drop), _p(rng_state), int(site), _dt(x), _p(ws), nws, _stream()),
                   "msmd_layernorm_bwd_dropout")
        return dx, dg, db, dxd
    _lib.check(lib.msmd_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(dx), _p(dg), _p(db), rows, cols, eps, _dt(x),
                                      _p(ws), nws, _stream()), "msmd_layernorm_bwd")
    return dx, dg, db


def softmax_rows_(s, cols, ld, Tq, scale, mask=None):
    lib = _lib.load()
    rows = s.numel() // ld
    _lib.check(lib.msmd_softmax_rows(_p(s), _p(mask), rows, cols, ld, Tq, float(scale), _dt(s), _stream()),
               "msmd_softmax_rows")
    return s


def softmax_bwd_rows_(P, dP, cols, ld, scale):
    lib = _lib.load()
    rows = P.numel() // ld
    _lib.check(lib.msmd_softmax_bwd_rows(_p(P), _p(dP), rows, cols, ld, float(scale), _dt(P), _stream()),
               "msmd_softmax_bwd_rows")
    return dP


def unfold_t(xp, T, Kk):
    """xp (B, G, Tp, Cg) -> (G, Kk*Cg, Mp) with Mp = B*T rounded up to 8 (zero padded)."""
    lib = _lib.load()
    B, G, Tp, Cg = xp.shape
    Mp = (B * T + 7) // 8 * 8
    out = torch.empty(G, Kk * Cg, Mp, device=xp.device, dtype=xp.dtype)
    _lib.check(lib.msmd_unfold_t(_p(xp), _p(out), B, T, Tp, G, Cg, Kk, Mp, _dt(xp), _stream()), "msmd_unfold_t")
    return out
