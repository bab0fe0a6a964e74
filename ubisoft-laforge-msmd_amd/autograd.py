"""Differentiable building blocks for the training path: torch.autograd.Function shims whose forward AND
backward run on the hand-written HIP kernels (csrc/gemm.hip, backward.hip, norm.hip).

Mixed precision as the inference path: activations in the compute dtype (bf16 or fp32), master weights and all
gradients of parameters in fp32.  dgrad / wgrad are the same MFMA GEMM on transposed operands:
    dX = dZ . W        -> gemm(dZ, W^T as the (K, N) operand)
    dW = dZ^T . X      -> gemm(dZ^T (N, M), X^T (K, M)) with fp32 output
Training attention materialises P (needed by the backward) with batched GEMMs + a row-softmax kernel.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import ops

ACT_NONE, ACT_GELU, ACT_ELU = ops.ACT_NONE, ops.ACT_GELU, ops.ACT_ELU


class WeightCache:
    """Per-optimizer-step cache of compute-dtype weight copies and their transposes (like AMP's cast cache).
    `persistent` entries are views into arenas a Trainer refreshes with ONE launch per step (WeightArena)."""

    def __init__(self):
        self.store = {}
        self.persistent = {}

    def clear(self):
        self.store.clear()

    def get(self, w: torch.Tensor, dtype):
        key = (id(w), dtype)
        ver = w._version
        e = self.persistent.get(key)
        if e is not None:
            if e[0]() is w and e[1] == (w.data_ptr(), tuple(w.shape)):
                ver = e[5]()        # an alias view answers with its owner parameters' counters (FusedAlias.version)
                if e[4][0] == ver:
                    return e[2], e[3]
                # written in place behind the arena's back (load_state_dict, manual edits): cast afresh below until
                # the owner's next refresh() picks the new values up
            else:
                del self.persistent[key]        # the parameter died or moved: drop the stale views
        e = self.store.get(key)
        # id() values are recycled once a tensor dies: an entry is valid only for the SAME live tensor object (weak
        # reference), at the same version and storage address
        if e is None or e[0]() is not w or e[1] != (ver, w.data_ptr(), tuple(w.shape)):
            w2 = w.detach().reshape(w.shape[0], -1)
            K = w2.shape[1]
            Kp = (K + 7) // 8 * 8
            wc = ops.pad_cols(w2.float().contiguous(), Kp, dtype)     # (N, Kp)
            wct = ops.transpose2d(wc[:, :K] if Kp != K else wc, 8)     # (K, Np)
            e = self.store[key] = (weakref.ref(w), (ver, w.data_ptr(), tuple(w.shape)), wc, wct)
        return e[2], e[3]


CACHE = WeightCache()


class FusedAlias:
    """A weight operand that is a VIEW of the flat parameter arena spanning several parameters (Q | K | V of an encoder
    layer, laid out next to each other: dp.ADJACENT) or part of one (the Q / KV rows of nn.MultiheadAttention's
    in_proj_weight): `w`, `b` plain tensors for the forward, `wgrad`, `bgrad` the same regions of the gradient arena for
    the backward's wgrad GEMM to add into, `owners` the leaf parameters to report as written.  No torch.cat / slice nodes
    in the autograd graph, no per-iteration cast / transpose of a temporary, no accumulate kernels."""

    def __init__(self, w, b, wgrad, bgrad, owners):
        self.w, self.b, self.wgrad, self.bgrad, self.owners = w, b, wgrad, bgrad, list(owners)
        # `w` / `b` carry no autograd history: without an input that requires grad the node's output would not either and its
        # backward -- the only place the owners' gradients are written -- would silently never run.  The token is a leaf that
        # always requires grad and rides along as an extra input of the node (its own gradient is None).
        self.token = torch.zeros((), device=w.device, requires_grad=True)

    def version(self):
        """In-place writes to the OWNER parameters (load_state_dict, p.copy_(), a finite-difference probe) move THEIR version
        counters, not the arena view's (parameters got their own counter through `p.data = view`): the cached casts of the
        alias are valid for this number only."""
        return self.w._version + sum(o._version for o in self.owners)


FUSED = {}   # (kind, id(module), layer index) -> FusedAlias; filled by the Trainer (direct-gradient mode only)


class WeightArena:
    """bf16 casts and transposes of all 8-aligned 2-D trainable weights living in one flat fp32 parameter arena, kept
    in two bf16 arenas and rewritten by ONE kernel launch (msmd_cast_transpose_multi) after every optimizer step;
    LinearFn finds them through CACHE.persistent, so an iteration issues no per-weight cast / transpose launches."""

    def __init__(self, flat_param, params, aliases=(), skip=()):
        """aliases: FusedAlias objects (or plain 2-D views of the arena) whose `w` gets its own cast / transpose; skip:
        parameters that are only ever used through an alias."""
        self.flat = flat_param
        rows, self.views = [], []
        off = tiles = 0
        base = flat_param.data_ptr()
        skip = {id(p) for p in skip}
        vfn = {}     # id(tensor) -> its version getter (an alias answers with its owners' counters)
        alias_w = []
        for a in aliases:
            t = a.w if isinstance(a, FusedAlias) else a
            alias_w.append(t)
            vfn[id(t)] = a.version if isinstance(a, FusedAlias) else (lambda t=t: t._version)
        aliases = alias_w
        alias_ids = {id(t) for t in aliases}
        for p in list(params) + list(aliases):
            if id(p) in skip:
                continue
            if p.ndim != 2 or not (p.requires_grad or id(p) in alias_ids) or p.shape[0] % 8 or p.shape[1] % 8 or not p.is_contiguous():
                continue
            src = (p.data_ptr() - base) // 4
            if src < 0 or src + p.numel() > flat_param.numel():
                continue
            N, K = p.shape
            rows.append((src, N, K, off, off, tiles))
            self.views.append((p, off, N, K))
            off += (N * K + 63) // 64 * 64
            tiles += ((N + 31) // 32) * ((K + 31) // 32)
        self.n, self.tiles = len(rows), tiles
        if not rows:
            return
        dev = flat_param.device
        self.meta = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.cast = torch.empty(off, device=dev, dtype=torch.bfloat16)
        self.tr = torch.empty(off, device=dev, dtype=torch.bfloat16)
        self.versions = []
        for p, o, N, K in self.views:
            get = vfn.get(id(p)) or (lambda r=weakref.ref(p): r()._version)
            box = [get()]
            self.versions.append((box, get))
            CACHE.persistent[(id(p), torch.bfloat16)] = (weakref.ref(p), (p.data_ptr(), tuple(p.shape)),
                                                         self.cast[o:o + N * K].view(N, K), self.tr[o:o + N * K].view(K, N),
                                                         box, get)
        self.refresh()

    def refresh(self):
        if self.n:
            ops.cast_transpose_multi(self.flat, self.meta, self.n, self.tiles, self.cast, self.tr)
            for box, get in self.versions:
                box[0] = get()

    def release(self):
        """Drop this arena's views from the cache (the bf16 arenas are freed with them)."""
        for p, _, _, _ in self.views:
            e = CACHE.persistent.get((id(p), torch.bfloat16))
            if e is not None and e[0]() is p:
                del CACHE.persistent[(id(p), torch.bfloat16)]
        self.views, self.n = [], 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


USE_GEMM_TN = True
# Set by Trainer when bucket launches do not depend on per-parameter hooks (hipGraph mode / single process):
# LinearFn.backward then adds weight / bias gradients directly into the pre-allocated .grad arena views.
DIRECT_GRAD = False
# Called with every leaf parameter whose gradient a backward kernel has just added STRAIGHT into its .grad arena view
# (no autograd accumulation, hence no post-accumulate hook): dp.GradBucketReducer.on_write, which is how the reducer
# knows in hipGraph mode when a gradient bucket is final.
GRAD_WRITTEN = None


_MASKS = {}


def _mask_u8(mask):
    """The uint8 form of a (static) boolean attention mask, converted once per mask tensor and version."""
    if mask is None:
        return None
    if mask.dtype == torch.uint8 and mask.is_contiguous():
        return mask
    key = id(mask)
    e = _MASKS.get(key)
    if e is None or e[0]() is not mask or e[1] != (mask._version, mask.data_ptr()):
        if len(_MASKS) > 64:
            # drop entries of masks that died; a LIVE mask's uint8 copy may be baked into a captured hipGraph as a raw
            # pointer (the denoiser's alignment mask in graph-mode training) and must never be freed behind it
            for k in [k for k, v in _MASKS.items() if v[0]() is None]:
                del _MASKS[k]
        e = _MASKS[key] = (weakref.ref(mask), (mask._version, mask.data_ptr()), mask.to(torch.uint8).contiguous())
    return e[2]


def _param_grads(dz2, xin, w, b_ref, has_b, K, need_w, need_b, alias):
    """dW, db of y = x W^T + b from dz2 (M, N) and the saved input xin (..., Kp): returned, or -- direct-gradient mode /
    FusedAlias operands -- added straight into the gradient arena (then None is returned for them)."""
    dtype = xin.dtype
    N = w.shape[0]
    Kp = xin.shape[-1]
    M = xin.numel() // Kp
    dw = db = None
    want_b = has_b and need_b
    if alias is not None:
        fa = alias
        if dtype == torch.bfloat16 and USE_GEMM_TN and N % 8 == 0 and Kp == K:
            ops.gemm_tn(dz2, xin.reshape(M, Kp), out=fa.wgrad.view(N, Kp), colsum_out=fa.bgrad, accumulate=True)
        else:       # the forms the TN kernel does not take (fp32 operands, USE_GEMM_TN off): the generic products, then added
            dwa, dba = _param_grads(dz2, xin, w, None, has_b, K, True, True, None)
            fa.wgrad.add_(dwa.reshape(fa.wgrad.shape))
            if has_b:
                fa.bgrad.add_(dba.reshape(fa.bgrad.shape))
        if GRAD_WRITTEN is not None:
            for o in fa.owners:
                GRAD_WRITTEN(o)
        return None, None
    direct = (DIRECT_GRAD and need_w and dtype == torch.bfloat16 and N % 8 == 0 and Kp == K
              and w.is_leaf and w.grad is not None and w.grad.is_contiguous() and USE_GEMM_TN)
    if direct:
        # accumulate dW (and db) straight into the parameters' .grad views of the flat gradient arena: no fp32
        # temporary, no autograd add kernel per parameter and window
        bgrad = None
        if want_b:
            b_leaf = b_ref
            if b_leaf is not None and b_leaf.grad is not None and b_leaf.grad.is_contiguous():
                bgrad = b_leaf.grad
        ops.gemm_tn(dz2, xin.reshape(M, Kp), out=w.grad.view(N, Kp), colsum_out=bgrad, accumulate=True)
        if GRAD_WRITTEN is not None:
            GRAD_WRITTEN(w)
            if bgrad is not None:
                GRAD_WRITTEN(b_ref)
        if want_b and bgrad is None:
            db = ops.colsum(dz2)
    elif need_w and dtype == torch.bfloat16 and N % 8 == 0 and USE_GEMM_TN:
        # dW = dZ^T X straight from the row-major operands (transposing LDS reads), bias gradient fused
        x2 = xin.reshape(M, Kp)
        if want_b:
            dwe, db = ops.gemm_tn(dz2, x2, want_colsum=True)
        else:
            dwe = ops.gemm_tn(dz2, x2)
        dw = dwe[:, :K].reshape(w.shape)
    elif need_w and dtype == torch.bfloat16 and USE_GEMM_TN and PAD_N_FOR_TN:
        # N not a multiple of 8 (the 67 / 71-wide motion heads): zero-pad dZ's columns and take the same TN GEMM (3 launches
        # instead of the 6 of the transposed-operand form below)
        Np = (N + 7) // 8 * 8
        dzp = ops.pad_cols(dz2.contiguous(), Np, dtype)
        x2 = xin.reshape(M, Kp)
        if want_b:
            dwe, dbp = ops.gemm_tn(dzp, x2, want_colsum=True)
            db = dbp[:N]
        else:
            dwe = ops.gemm_tn(dzp, x2)
        dw = dwe[:N, :K].reshape(w.shape)
    elif need_w:
        dzT = ops.transpose2d(dz2, 8)                      # (N, Mp)
        # x^T with 8 extra rows: row Kp is all ones, so column Kp of the product is the bias gradient
        # (dz^T . 1) for free inside the wgrad GEMM instead of a separate column-sum pass.
        Mp = dzT.shape[1]
        xT = torch.zeros(Kp + 8, Mp, device=xin.device, dtype=dtype)
        ops.transpose(xin.reshape(M, Kp), xT, M, Kp, Kp, Mp)
        xT[Kp, :M] = 1
        dwe = ops.gemm(dzT, xT, out_dtype=torch.float32)   # (N, Kp + 8) fp32
        dw = dwe[:, :K].reshape(w.shape)
        if want_b:
            db = dwe[:, Kp].contiguous()
    elif want_b:
        db = ops.colsum(dz2)
    return dw, db


# Grad mode of the CALLER of the node about to run (inside Function.forward grad mode is always off): the wrappers below
# set it right before .apply, and the forwards skip tensors only a backward would read (pre-activations) when it is off.
_GRAD_ON = [True]


def _apply(fn, *args):
    _GRAD_ON[0] = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _GRAD_ON[0] = True


class Junction:
    """x feeds a Linear (the `jin` node: a block's QKV / Q projection) AND is the residual of a later Linear (the `jout`
    node: the block's output projection).  Autograd would add the two gradients of x with an elementwise kernel per block;
    with a Junction the jout node hands its residual gradient over instead of returning it, and the jin node passes it to
    its data-gradient GEMM as the epilogue's residual operand.  Legal only when the jout node's input depends on the jin
    node's output (then its backward runs first); the jin node raises if that was not so.  One Junction per block and pass."""
    __slots__ = ("armed", "paired", "pending", "shape", "ident")

    def __init__(self):
        self.armed = self.paired = False
        self.pending = self.shape = self.ident = None


class LinearFn(torch.autograd.Function):
    """y = act(x @ w.T + b) + residual   (x: (..., K) compute dtype; w (N, K), b (N) fp32 master parameters)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, act, p_drop=0.0, site=0, alias=None, jin=None, jout=None, token=None):
        """y = dropout_p(act(x W^T + b)) + residual in ONE GEMM launch: the epilogue also writes the pre-activation z
        (needed by the backward) when there is an activation, and applies the Philox keep mask.  alias (FusedAlias):
        w / b are arena views without autograd history; the backward adds dW / db into alias.wgrad / alias.bgrad.
        jin / jout (Junction): see Junction -- the residual gradient of the jout node rides into the jin node's
        data-gradient GEMM instead of an autograd accumulation kernel."""
        ctx.alias = alias
        dtype = x.dtype
        wc, wct = CACHE.get(w, dtype)
        K = w.shape[1]
        ctx.jin = ctx.jout = None
        if jin is not None and ctx.needs_input_grad[0] and wc.shape[1] == K and x.dtype in (torch.bfloat16, torch.float16):
            ctx.jin, jin.armed, jin.shape = jin, True, tuple(x.shape)
            jin.ident = (x.data_ptr(), x._version, x.dtype)
        # the residual must BE the tensor the jin node consumed (same storage, same version), not merely look like it: a
        # same-shaped neighbour (h_in vs h after a refactor) would get its gradient routed into the wrong dx without an error
        if (jout is not None and jout.armed and residual is not None and ctx.needs_input_grad[3]
                and tuple(residual.shape) == jout.shape and (residual.data_ptr(), residual._version, residual.dtype) == jout.ident
                and residual.dtype == x.dtype):
            ctx.jout, jout.paired = jout, True
        xin = x if wc.shape[1] == K else ops.pad_cols(x.contiguous(), wc.shape[1], dtype)
        xin = xin.contiguous()
        bf = b.detach().float().contiguous() if b is not None else None
        res = residual.contiguous() if residual is not None else None
        z = None
        if act != ACT_NONE and _GRAD_ON[0]:   # no-grad passes (the reference's extra encoder pass) keep no z
            z = torch.empty(*xin.shape[:-1], w.shape[0], device=x.device, dtype=dtype)
        y = ops.gemm(xin, wc, bf, res, act, z_out=z, p_drop=p_drop, rng_state=TrainNoise.state if p_drop > 0 else None,
                     site=site)
        ctx.save_for_backward(xin, z, w)
        ctx.act, ctx.has_b, ctx.has_r, ctx.K = act, b is not None, residual is not None, K
        ctx.b_ref = b if (b is not None and b.is_leaf) else None
        ctx.drop = (p_drop, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        xin, z, w = ctx.saved_tensors
        dtype = xin.dtype
        p_drop, site = ctx.drop
        dropped = _dropped_by_producer(dy, p_drop, site) if ctx.act == ACT_NONE else None
        dy = dy.contiguous()
        if dropped is not None:
            dz = dropped
        elif p_drop > 0.0:
            dz = (ops.dropout(dy, p_drop, TrainNoise.state, site) if ctx.act == ACT_NONE
                  else ops.act_bwd_dropout(dy, z, ctx.act, p_drop, TrainNoise.state, site))
        else:
            dz = dy if ctx.act == ACT_NONE else ops.act_bwd(dy, z, ctx.act)
        N = w.shape[0]
        Kp = xin.shape[-1]
        M = xin.numel() // Kp
        dz2 = dz.reshape(M, N)
        dx = dw = db = None
        wc, wct = CACHE.get(w, dtype)
        if ctx.needs_input_grad[0]:
            dzp = dz2 if wct.shape[1] == N else ops.pad_cols(dz2, wct.shape[1], dtype)
            jres = None
            if ctx.jin is not None and ctx.jin.paired:
                pend, ctx.jin.pending = ctx.jin.pending, None
                if pend is None:   # the provider is downstream of this node in every legal use: it has run
                    raise RuntimeError("Junction: the residual node's backward has not handed its gradient over")
                jres = pend.reshape(M, -1)
            dx = ops.gemm(dzp, wct, residual=jres).reshape(*xin.shape[:-1], wct.shape[0])
        dw, db = _param_grads(dz2, xin, w, ctx.b_ref, ctx.has_b, ctx.K, ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                              ctx.alias)
        dres = dy if ctx.has_r and ctx.needs_input_grad[3] else None
        if ctx.jout is not None and dres is not None:
            ctx.jout.pending, dres = dres, None
        return dx, dw, db, dres, None, None, None, None, None, None, None


class FFNFn(torch.autograd.Function):
    """y = dropout_p2(W2 dropout_p1(act(W1 x + b1)) + b2) + residual as ONE autograd node: two GEMM launches forward (as two
    LinearFn nodes would), and a backward whose middle -- linear2's data gradient, the first dropout's and the
    activation's backward -- is ONE launch (msmd_gemm_actbwd) instead of a GEMM + an elementwise pass over the
    (rows, d_ff) tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, act, p1, site1, p2, site2, residual_is_x=False):
        """residual_is_x: the block is x + FFN(x) (post-LN layers): x itself is the residual operand, and the backward
        returns ONE gradient for x -- the upstream gradient rides into linear1's data-gradient GEMM as its residual instead
        of an autograd accumulation kernel."""
        dtype = x.dtype
        w1c, _ = CACHE.get(w1, dtype)
        w2c, _ = CACHE.get(w2, dtype)
        xin = x.contiguous()
        z1 = torch.empty(*xin.shape[:-1], w1.shape[0], device=x.device, dtype=dtype) if _GRAD_ON[0] else None
        f = ops.gemm(xin, w1c, b1.detach().float().contiguous(), None, act, z_out=z1, p_drop=p1,
                     rng_state=TrainNoise.state if p1 > 0 else None, site=site1)
        res = xin if residual_is_x else (residual.contiguous() if residual is not None else None)
        y = ops.gemm(f, w2c, b2.detach().float().contiguous(), res, ACT_NONE, p_drop=p2,
                     rng_state=TrainNoise.state if p2 > 0 else None, site=site2)
        ctx.save_for_backward(xin, z1, f, w1, w2)
        ctx.cfg = (act, p1, site1, p2, site2, residual is not None, residual_is_x)
        ctx.refs = (b1 if b1.is_leaf else None, b2 if b2.is_leaf else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xin, z1, f, w1, w2 = ctx.saved_tensors
        act, p1, site1, p2, site2, has_r, res_is_x = ctx.cfg
        dtype = xin.dtype
        dropped = _dropped_by_producer(dy, p2, site2)
        dy = dy.contiguous()
        dz2 = dropped if dropped is not None else (ops.dropout(dy, p2, TrainNoise.state, site2) if p2 > 0.0 else dy)
        M = xin.numel() // xin.shape[-1]
        dz2 = dz2.reshape(M, w2.shape[0])
        _, w1ct = CACHE.get(w1, dtype)
        _, w2ct = CACHE.get(w2, dtype)
        dw2, db2 = _param_grads(dz2, f, w2, ctx.refs[1], True, w2.shape[1], ctx.needs_input_grad[3], ctx.needs_input_grad[4], None)
        # linear2's data gradient with dropout1's and the activation's backward in its epilogue
        dz1 = ops.gemm_act_bwd(dz2, w2ct, z1.reshape(M, -1), act, p1, TrainNoise.state if p1 > 0 else None, site1)
        dw1, db1 = _param_grads(dz1, xin, w1, ctx.refs[0], True, w1.shape[1], ctx.needs_input_grad[1], ctx.needs_input_grad[2], None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dz1, w1ct, residual=dy.reshape(M, -1) if res_is_x else None).reshape(xin.shape)
        dres = dy if has_r and ctx.needs_input_grad[5] else None
        return dx, dw1, db1, dw2, db2, dres, None, None, None, None, None, None


def ffn(x, w1, b1, w2, b2, p_act, p_out, residual=None, act=ops.ACT_GELU):
    """The transformer feed-forward block dropout(W2 dropout(act(W1 x))) + residual.  One fused node (FFNFn) when the shapes
    allow the fused epilogues (16-bit, 8-aligned widths); else the two LinearFn nodes."""
    ok = (x.dtype in (torch.bfloat16, torch.float16) and w1.shape[0] % 8 == 0 and w1.shape[1] % 8 == 0
          and w2.shape[0] % 64 == 0 and b1 is not None and b2 is not None and USE_GEMM_TN)
    if not ok:
        return linear_dropout(linear_dropout(x, w1, b1, p_act, act=act), w2, b2, p_out, residual=residual)
    if not TrainNoise.active:
        p_act = p_out = 0.0
    s1 = TrainNoise.next_site() if p_act > 0 else 0
    s2 = TrainNoise.next_site() if p_out > 0 else 0
    if residual is x:       # x + FFN(x): one gradient for x out of the node (the sum is made in a GEMM epilogue)
        return _tag_dropout(_apply(FFNFn, x, w1, b1, w2, b2, None, act, float(p_act), s1, float(p_out), s2, True), float(p_out), s2)
    return _tag_dropout(_apply(FFNFn, x, w1, b1, w2, b2, residual, act, float(p_act), s1, float(p_out), s2), float(p_out), s2)


# Post-LN blocks: h = LayerNorm(residual + dropout_p(Linear(..))).  The Linear's backward needs dropout_mask(dy) with dy = the
# LayerNorm's dx; the LayerNorm backward launch can write that tensor next to dx (msmd_layernorm_bwd_dropout) instead of a
# separate msmd_dropout launch over the same rows.  The two autograd nodes stay separate and talk through attributes:
#   forward : the Linear / FFN node's OUTPUT tensor carries _msmd_drop = (p, site); layer_norm() reads it off its input;
#   backward: the LayerNorm node attaches _msmd_dropped = (p, site, version, tensor) to the dx it RETURNS; the Linear node
#             uses it only if it finds it on the very tensor it is handed, at the same version -- if the LayerNorm was not
#             the only consumer, the engine hands over a sum (another tensor, or this one after an in-place add_ that
#             bumps its version) and the node falls back to its own msmd_dropout launch.
FUSE_LN_DROPOUT_BWD = os.environ.get("MSMD_FUSE_LN_DROPOUT_BWD", "1") != "0"
USE_JUNCTIONS = os.environ.get("MSMD_JUNCTIONS", "1") != "0"
PAD_N_FOR_TN = os.environ.get("MSMD_PAD_N_FOR_TN", "1") != "0"


def _tag_dropout(y, p, site):
    if FUSE_LN_DROPOUT_BWD and p > 0.0:
        y._msmd_drop = (float(p), int(site))
    return y


def _dropped_by_producer(dy, p, site):
    st = getattr(dy, "_msmd_dropped", None)
    if st is None or p <= 0.0:
        return None
    sp, ssite, ver, t = st
    if sp == p and ssite == site and ver == dy._version and t.shape == dy.shape and t.dtype == dy.dtype:
        return t
    return None


def _tag_dropped(res, drop):
    """res = ops.layernorm_bwd(...): (dx, dg, db[, dx_dropped]) -> (dx, dg, db) with the dropped copy attached to dx."""
    if drop is None:
        return res
    dx, dg, db, dxd = res
    dx._msmd_dropped = (drop[0], drop[1], dx._version, dxd)
    return dx, dg, db


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x) * gamma + beta (+ post_add constant row).  drop = (p, site) of the dropout-Linear that produced x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, post_add, drop=None):
        ctx.drop = drop if (drop is not None and x.dtype == torch.bfloat16 and x.shape[-1] % 4 == 0) else None
        x = x.contiguous()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = ops.layernorm(x, g, b, post_add=post_add)
        ctx.save_for_backward(x, g)
        ctx.refs = (gamma, beta) if (gamma.is_leaf and beta.is_leaf) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        refs = ctx.refs
        drop = (ctx.drop[0], TrainNoise.state, ctx.drop[1]) if ctx.drop is not None else None
        if (DIRECT_GRAD and refs is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
                and all(r.grad is not None and r.grad.is_contiguous() and r.grad.dtype == torch.float32 for r in refs)):
            # the kernel accumulates dgamma / dbeta: aim it at the parameters' .grad views of the gradient arena
            dx = _tag_dropped(ops.layernorm_bwd(dy.contiguous(), x, g, dg_out=refs[0].grad.view(-1),
                                                db_out=refs[1].grad.view(-1), drop=drop), ctx.drop)[0]
            if GRAD_WRITTEN is not None:
                GRAD_WRITTEN(refs[0])
                GRAD_WRITTEN(refs[1])
            return dx, None, None, None, None
        dx, dg, db = _tag_dropped(ops.layernorm_bwd(dy.contiguous(), x, g, drop=drop), ctx.drop)
        return dx, dg, db, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(scale * Q K^T, masked) V per head (head_dim 64) with P materialised for the backward.
    q: (B, Tq, H*64), k / v: (B, Tk, H*64); last-dim-contiguous views of packed projections are accepted."""

    @staticmethod
    def forward(ctx, q, k, v, n_heads, scale, mask, p_drop=0.0):
        B, Tq, d = q.shape
        Tk = k.shape[1]
        H = n_heads
        dt = q.dtype
        Tkp = (Tk + 7) // 8 * 8
        P = torch.empty(B, H, Tq, Tkp, device=q.device, dtype=dt)
        ops.gemm_batched2(q, k, P, Tq, Tk, 64, q.stride(1), k.stride(1), Tkp, B, q.stride(0), k.stride(0), H * Tq * Tkp,
                          H, 64, 64, Tq * Tkp)
        m8 = _mask_u8(mask)
        ops.softmax_rows_(P, Tk, Tkp, Tq, scale, m8)
        site = TrainNoise.next_site() if p_drop > 0.0 else 0
        Pv = ops.dropout(P, p_drop, TrainNoise.state, site) if p_drop > 0.0 else P   # dropped probabilities feed P.V
        VT = (torch.zeros if Tkp != Tk else torch.empty)(B, H, 64, Tkp, device=q.device, dtype=dt)
        ops.transpose(v, VT, Tk, 64, v.stride(1), Tkp, B, v.stride(0), H * 64 * Tkp, H, 64, 64 * Tkp)
        O = torch.empty(B, Tq, d, device=q.device, dtype=dt)
        ops.gemm_batched2(Pv, VT, O, Tq, 64, Tkp, Tkp, Tkp, d, B, H * Tq * Tkp, H * 64 * Tkp, Tq * d, H, Tq * Tkp,
                          64 * Tkp, 64)
        ctx.save_for_backward(q, k, v, P)
        ctx.dims = (B, H, Tq, Tk, Tkp, d, scale)
        ctx.drop = (p_drop, site)
        return O

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P = ctx.saved_tensors
        B, H, Tq, Tk, Tkp, d, scale = ctx.dims
        dt = q.dtype
        dev = q.device
        dO = dO.contiguous()
        Tqp = (Tq + 7) // 8 * 8
        # dV_h = P_h^T . dO_h
        zq = torch.zeros if Tqp != Tq else torch.empty
        zk = torch.zeros if Tkp != Tk else torch.empty
        p_drop, site = ctx.drop
        Pv = ops.dropout(P, p_drop, TrainNoise.state, site) if p_drop > 0.0 else P
        PT = zq(B, H, Tk, Tqp, device=dev, dtype=dt)
        ops.transpose(Pv, PT, Tq, Tk, Tkp, Tqp, B * H, Tq * Tkp, Tk * Tqp)
        dOT = zq(B, H, 64, Tqp, device=dev, dtype=dt)
        ops.transpose(dO, dOT, Tq, 64, d, Tqp, B, Tq * d, H * 64 * Tqp, H, 64, 64 * Tqp)
        dV = torch.empty(B, Tk, d, device=dev, dtype=dt)
        ops.gemm_batched2(PT, dOT, dV, Tk, 64, Tqp, Tqp, Tqp, d, B, H * Tk * Tqp, H * 64 * Tqp, Tk * d, H, Tk * Tqp,
                          64 * Tqp, 64)
        # dP_h = dO_h . V_h^T ; dS = scale * P o (dP - rowsum(dP o P))
        dP = torch.empty(B, H, Tq, Tkp, device=dev, dtype=dt)
        ops.gemm_batched2(dO, v, dP, Tq, Tk, 64, d, v.stride(1), Tkp, B, Tq * d, v.stride(0), H * Tq * Tkp, H, 64, 64,
                          Tq * Tkp)
        if p_drop > 0.0:
            dP = ops.dropout(dP, p_drop, TrainNoise.state, site)
        ops.softmax_bwd_rows_(P, dP, Tk, Tkp, scale)
        dS = dP
        # dQ_h = dS_h . K_h
        KT = zk(B, H, 64, Tkp, device=dev, dtype=dt)
        ops.transpose(k, KT, Tk, 64, k.stride(1), Tkp, B, k.stride(0), H * 64 * Tkp, H, 64, 64 * Tkp)
        dQ = torch.empty(B, Tq, d, device=dev, dtype=dt)
        ops.gemm_batched2(dS, KT, dQ, Tq, 64, Tkp, Tkp, Tkp, d, B, H * Tq * Tkp, H * 64 * Tkp, Tq * d, H, Tq * Tkp,
                          64 * Tkp, 64)
        # dK_h = dS_h^T . Q_h
        dST = zq(B, H, Tk, Tqp, device=dev, dtype=dt)
        ops.transpose(dS, dST, Tq, Tk, Tkp, Tqp, B * H, Tq * Tkp, Tk * Tqp)
        QT = zq(B, H, 64, Tqp, device=dev, dtype=dt)
        ops.transpose(q, QT, Tq, 64, q.stride(1), Tqp, B, q.stride(0), H * 64 * Tqp, H, 64, 64 * Tqp)
        dK = torch.empty(B, Tk, d, device=dev, dtype=dt)
        ops.gemm_batched2(dST, QT, dK, Tk, 64, Tqp, Tqp, Tqp, d, B, H * Tk * Tqp, H * 64 * Tqp, Tk * d, H, Tk * Tqp,
                          64 * Tqp, 64)
        return dQ, dK, dV, None, None, None, None


class TrainNoise:
    """Training-mode noise state shared by the autograd functions (reference: model.train() at
    training_script.py:55 turns on every nn.Dropout, HF LayerDrop and SpecAugment).
      active      -- regularisers on (Trainer sets it from model.training)
      state       -- device int64 [seed, step]; kernels read it, so a captured hipGraph draws new masks per replay
      site        -- per-iteration call-site counter (reset by begin()): with (seed, step) it keys each mask
      graph_safe  -- no host-side decisions (LayerDrop becomes a device-side select)
    """
    active = False
    state = None
    site = 0
    graph_safe = False
    host_rng = None
    spec_masks = None   # list of (B, T) bool device tensors consumed in order by audio_encoder_train, or None

    @classmethod
    def begin(cls):
        cls.site = 0
        cls._spec_i = 0

    @classmethod
    def next_site(cls):
        cls.site += 1
        return cls.site

    @classmethod
    def next_spec_mask(cls):
        if cls.spec_masks is None:
            return None
        m = cls.spec_masks[cls._spec_i]
        cls._spec_i += 1
        return m


class DropoutFn(torch.autograd.Function):
    """y = dropout_p(x) (+ residual); the mask is regenerated in the backward from (state, site)."""

    @staticmethod
    def forward(ctx, x, residual, p, site):
        ctx.p, ctx.site, ctx.has_r = p, site, residual is not None
        return ops.dropout(x, p, TrainNoise.state, site, residual)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = ops.dropout(dy, ctx.p, TrainNoise.state, ctx.site)
        return dx, (dy if ctx.has_r else None), None, None


def dropout(x, p, residual=None):
    """nn.Dropout(p) in the current mode, optionally fused with the residual add that follows it."""
    if not TrainNoise.active or p <= 0.0:
        return x if residual is None else x + residual
    return DropoutFn.apply(x, residual, float(p), TrainNoise.next_site())


def linear_dropout(x, w, b, p, residual=None, act=ACT_NONE, junction_out=None):
    """dropout_p(act(x W^T + b)) + residual: eval mode keeps the residual fused in the GEMM epilogue."""
    if not TrainNoise.active or p <= 0.0:
        return linear(x, w, b, act=act, residual=residual, junction_out=junction_out)
    if w.shape[0] % 4 == 0:   # mask index needs N % 4 == 0 (every Linear on the path); else compose
        site = TrainNoise.next_site()
        y = _apply(LinearFn, x, w, b, residual, act, float(p), site, None, None, junction_out if USE_JUNCTIONS else None)
        return _tag_dropout(y, float(p), site) if act == ACT_NONE else y
    return dropout(linear(x, w, b, act=act), p, residual)


class FusedSelfAttnFn(torch.autograd.Function):
    """Self-attention on a packed (B, T, 3 d) projection: fused forward (msmd_attention) and fused backward
    (msmd_attention_bwd, P recomputed); the gradient comes back as ONE packed tensor (no slice / cat glue)."""

    @staticmethod
    def forward(ctx, qkv, n_heads, scale, mask, p_drop, site, prefetch=None):
        d = qkv.shape[-1] // 3
        qkv = qkv.contiguous()
        m8 = _mask_u8(mask)
        o = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], n_heads, scale, m8, p_drop=p_drop,
                          rng_state=TrainNoise.state, site=site, prefetch=prefetch)
        ctx.save_for_backward(qkv, m8)
        ctx.cfg = (n_heads, scale, d, p_drop, site)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, m8 = ctx.saved_tensors
        H, scale, d, p_drop, site = ctx.cfg
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], do.contiguous(), dqkv[..., :d],
                          dqkv[..., d:2 * d], dqkv[..., 2 * d:], H, scale, m8, p_drop, TrainNoise.state, site)
        return dqkv, None, None, None, None, None, None


class FusedCrossAttnFn(torch.autograd.Function):
    """Cross-attention: q (B, Tq, d), packed kv (B, Tk, 2 d)."""

    @staticmethod
    def forward(ctx, q, kv, n_heads, scale, mask, p_drop, site):
        d = q.shape[-1]
        q, kv = q.contiguous(), kv.contiguous()
        m8 = _mask_u8(mask)
        o = ops.attention(q, kv[..., :d], kv[..., d:], n_heads, scale, m8, p_drop=p_drop, rng_state=TrainNoise.state,
                          site=site)
        ctx.save_for_backward(q, kv, m8)
        ctx.cfg = (n_heads, scale, d, p_drop, site)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, m8 = ctx.saved_tensors
        H, scale, d, p_drop, site = ctx.cfg
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attention_bwd(q, kv[..., :d], kv[..., d:], do.contiguous(), dq, dkv[..., :d], dkv[..., d:], H, scale, m8,
                          p_drop, TrainNoise.state, site)
        return dq, dkv, None, None, None, None, None


FUSED_ATTENTION = True


def _fusable(x, Tk):
    return FUSED_ATTENTION and x.dtype == torch.bfloat16 and Tk <= 256


def _pdrop(p):
    return float(p) if (TrainNoise.active and p > 0.0) else 0.0


def cast_of(w, dtype=torch.bfloat16):
    """The compute-dtype cast of a weight as LinearFn will read it (a view of the weight arena when there is one, else
    None: nothing worth prefetching)."""
    e = CACHE.persistent.get((id(w), dtype))
    return e[2] if e is not None and e[0]() is w else None


def self_attention(qkv, n_heads, scale, mask=None, p_drop=0.0, prefetch=None):
    """softmax(scale Q K^T) V on a packed (B, T, 3 d) projection; p_drop = attention-probability dropout, applied
    in train mode only.  prefetch: tensors the launch also pulls through the memory-side cache (ops.attention)."""
    d = qkv.shape[-1] // 3
    p_drop = _pdrop(p_drop)
    if _fusable(qkv, qkv.shape[1]):
        return FusedSelfAttnFn.apply(qkv, n_heads, scale, mask, p_drop, TrainNoise.next_site() if p_drop else 0, prefetch)
    return attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], n_heads, scale, mask, p_drop)


def cross_attention(q, kv, n_heads, scale, mask=None, p_drop=0.0):
    d = q.shape[-1]
    p_drop = _pdrop(p_drop)
    if _fusable(q, kv.shape[1]):
        return FusedCrossAttnFn.apply(q, kv, n_heads, scale, mask, p_drop, TrainNoise.next_site() if p_drop else 0)
    return attention(q, kv[..., :d], kv[..., d:], n_heads, scale, mask, p_drop)


def linear(x, w, b=None, act=ACT_NONE, residual=None, junction_in=None, junction_out=None):
    return _apply(LinearFn, x, w, b, residual, act, 0.0, 0, None, junction_in if USE_JUNCTIONS else None,
                  junction_out if USE_JUNCTIONS else None)


def linear_alias(x, fa, act=ACT_NONE, residual=None, junction_in=None):
    """linear() on a FusedAlias operand (arena views; gradients go straight into the gradient arena)."""
    return _apply(LinearFn, x, fa.w, fa.b, residual, act, 0.0, 0, fa, junction_in if USE_JUNCTIONS else None, None, fa.token)


def layer_norm(x, gamma, beta, post_add=None, sole_consumer=True):
    """sole_consumer=False: x also feeds something else (pre-LN blocks: the residual stream), so the LayerNorm's dx is not
    the whole gradient of the Linear that produced x and the dropped copy would go unused."""
    drop = getattr(x, "_msmd_drop", None) if (sole_consumer and FUSE_LN_DROPOUT_BWD and TrainNoise.active) else None
    return LayerNormFn.apply(x, gamma, beta, post_add, drop)


def attention(q, k, v, n_heads, scale, mask=None, p_drop=0.0):
    return AttentionFn.apply(q, k, v, n_heads, scale, mask, _pdrop(p_drop))
