"""CFG + DDPM ancestral sampler (reference model.py:283-440) driven from the host, arithmetic in HIP.

Per step the device runs: pack x_t into the denoiser input -> decoder trunk -> heads/static mix ->
one fused CFG-combine + DDPM-posterior kernel that updates x_t in place.  Step-invariant work is
hoisted out of the T x n_entries loop: cross-attention K/V projections of
the audio memory for all layers, the static-style bases, the person projection and ALL T step
embeddings (one GEMM pair).  Nothing is copied to the host inside the loop (the reference moves
traj[t] to the CPU every step, model.py:433); per-step scalars come from host copies of the schedule.
"""
from __future__ import annotations

import os

import torch

from . import ops


def _entries(cfg_cond, cfg_mode):
    """(use_audio, use_style) per CFG entry in batch order; entry 0 is the null entry (model.py:340-366)."""
    ent = [("audio" not in cfg_cond, "style" not in cfg_cond)]
    for cond in cfg_cond:
        if cond == "audio":
            ent.append((True, "style" not in cfg_cond))
        elif cond == "style":
            if cfg_mode == "independent":
                ent.append(("audio" not in cfg_cond, True))
            elif cfg_mode == "incremental":
                ent.append((True, True))
            else:
                raise NotImplementedError(f"Unknown cfg_mode {cfg_mode}")
    return ent


def sample(model, audio_or_feat, shape_feat, style_feat=None, prev_motion_feat=None, prev_audio_feat=None,
           motion_at_T=None, indicator=None, cfg_mode=None, cfg_cond=None, cfg_scale=1.15, flexibility=0,
           dynamic_threshold=None, ret_traj=False, noise=None, guidance=None, separate=None):
    """guidance = (indices, values): naive in-painting of the denoiser INPUT (reference model.py:762-767).
    separate = dict(alpha_mod=callable|None, return_all_alpha=bool): also track the dynamic / static / alpha
    streams (reference sample_separate, model.py:442-651)."""
    net = model.denoising_net
    dtype = model.compute_dtype
    dev = model.device
    batch_size = audio_or_feat.shape[0]
    if cfg_mode is None:
        cfg_mode = model.cfg_mode
    if cfg_cond is None:
        cfg_cond = model.guiding_conditions
    cfg_cond = [c for c in cfg_cond if c in ["audio", "style"]]
    if not isinstance(cfg_scale, list):
        cfg_scale = [cfg_scale] * len(cfg_cond)
    if len(cfg_cond) > 0:
        cfg_cond, cfg_scale = zip(*sorted(zip(cfg_cond, cfg_scale), key=lambda x: ["audio", "style"].index(x[0])))
    else:
        cfg_cond, cfg_scale = [], []
    if cfg_mode not in ("incremental", "independent") and len(cfg_cond) > 0:
        raise NotImplementedError(f"Unknown cfg_mode {cfg_mode}")
    if model.target not in ("sample", "noise"):
        raise ValueError("Unknown target type: {}".format(model.target))
    if "style" in cfg_cond:
        assert model.use_style and style_feat is not None
    if model.use_style:
        if style_feat is None:
            style_feat = model.null_style_feat.expand(batch_size, -1, -1)
    else:
        assert style_feat is None, "This model does not support style feature input!"

    if audio_or_feat.ndim == 2:
        assert audio_or_feat.shape[1] == 16000 * model.n_motions / model.fps, \
            f"Incorrect audio length {audio_or_feat.shape[1]}"
        audio_feat = model._audio_feat(audio_or_feat, model.n_motions, dtype).float()
    elif audio_or_feat.ndim == 3:
        assert audio_or_feat.shape[1] == model.n_motions, f"Incorrect audio feature length {audio_or_feat.shape[1]}"
        audio_feat = audio_or_feat
    else:
        raise ValueError(f"Incorrect audio input shape {audio_or_feat.shape}")
    if shape_feat.ndim == 2:
        shape_feat = shape_feat.unsqueeze(1)
    if style_feat is not None and style_feat.ndim == 2:
        style_feat = style_feat.unsqueeze(1)
    if shape_feat.shape[0] != batch_size:
        shape_feat = shape_feat.expand(batch_size, -1, -1)
    if prev_motion_feat is None:
        prev_motion_feat = model.start_motion_feat.expand(batch_size, -1, -1)
    if prev_audio_feat is None:
        prev_audio_feat = model.start_audio_feat.expand(batch_size, -1, -1)
    if motion_at_T is None:
        motion_at_T = torch.randn((batch_size, model.n_motions, model.motion_feat_dim)).to(dev)

    L, Lp, dm, nb = model.n_motions, model.n_prev_motions, net.motion_feat_dim, net.num_of_basis
    null_audio = model.null_audio_feat.expand(batch_size, L, -1) if "audio" in cfg_cond else audio_feat
    audio_in, person_in = [], []
    for use_a, use_s in _entries(cfg_cond, cfg_mode):
        audio_in.append(audio_feat if use_a else null_audio)
        st = style_feat if (use_s or "style" not in cfg_cond) else model.null_style_feat.expand(batch_size, -1, -1)
        person_in.append(torch.cat([shape_feat, st], dim=-1) if model.use_style else shape_feat)
    n_entries = len(audio_in)
    N = n_entries * batch_size
    audio_in = torch.cat(audio_in, dim=0)
    person_in = torch.cat(person_in, dim=0)
    prev_m = torch.cat([prev_motion_feat] * n_entries, dim=0).float().contiguous()
    prev_a = torch.cat([prev_audio_feat] * n_entries, dim=0)
    ind_in = torch.cat([indicator] * n_entries, dim=0).float().contiguous() if indicator is not None else None
    if net.use_indicator and ind_in is None:
        raise TypeError("expected Tensor as element 1 in argument 0, but got NoneType")  # reference model.py:944

    # ---- step-invariant work, hoisted
    P = net.pack(dtype)
    T = model.diffusion_sched.num_steps
    tab = model.diffusion_sched.host_tables()
    mem = torch.cat([ops.cast(prev_a.contiguous(), dtype), ops.cast(audio_in.contiguous(), dtype)], dim=1)
    kv_list = net.memory_kv(mem, dtype)
    cross_list = (net.memory_cross(kv_list, dtype)
                  if (net.pack(dtype).diag and getattr(net, "diag_fast_path", True)) else None)
    stat = net.static_bases(style_feat, dtype).float().contiguous()  # real style for every entry (model.py:374)
    pf = ops.pad_cols(person_in.reshape(N, -1).float().contiguous(), P.kp_person, dtype)
    tok_person = ops.gemm(pf, *P.pp)                                   # (N, d) without the step embedding
    te_all = ops.cast(P.te[: T + 1].contiguous(), dtype)
    emb_all = ops.gemm(ops.gemm(te_all, *P.ds0, act=ops.ACT_GELU), *P.ds2)  # (T+1, d)
    scales = torch.tensor(list(cfg_scale), device=dev, dtype=torch.float32) if n_entries > 1 else None
    mode = 1 if cfg_mode == "independent" else 0
    target = 0 if model.target == "sample" else 1

    def coefficients(t):
        alpha = tab["alphas"][t]
        alpha_bar = tab["alpha_bars"][t]
        alpha_bar_prev = tab["alpha_bars"][t - 1]
        sigma = tab["sigmas_flex"][t] * flexibility + tab["sigmas_inflex"][t] * (1 - flexibility)
        if target == 1:
            c0 = 1 / torch.sqrt(alpha)
            c1 = (1 - alpha) / torch.sqrt(1 - alpha_bar)
        else:
            c0 = (1 - alpha_bar_prev) * torch.sqrt(alpha) / (1 - alpha_bar)
            c1 = (1 - alpha) * torch.sqrt(alpha_bar_prev) / (1 - alpha_bar)
        return float(c0), float(c1), float(sigma)

    use_graph = (noise is None and not ret_traj and getattr(model, "use_hip_graph", True)
                 and T > 1 and guidance is None and separate is None)
    if use_graph:
        x = _graph_loop(model, net, dtype, dev, T, N, n_entries, Lp, L, dm, nb, mode, target, P, motion_at_T, prev_m,
                        ind_in, mem, kv_list, stat, tok_person, emb_all, scales, coefficients,
                        tuple(dynamic_threshold) if dynamic_threshold else None, cross_list)
        return x, motion_at_T, audio_feat

    x = motion_at_T.float().clone().contiguous()
    traj = {T: motion_at_T} if ret_traj else None
    feats = torch.empty(N, 1 + Lp + L, P.kp_feat, device=dev, dtype=dtype)
    if separate is not None:
        cum_static = torch.zeros_like(x)
        alpha_traj = []
        stat_full = stat[torch.arange(N, device=dev) % stat.shape[0]]  # (N, nb, dm): real style for every entry
    for t in range(T, 0, -1):
        if t > 1:
            z = noise[t].float().contiguous() if noise is not None else torch.randn_like(x)
        else:
            z = None
        c0, c1, sigma = coefficients(t)
        x_in = x
        if guidance is not None and guidance[0] is not None:
            x_in = x.clone()
            x_in[:, guidance[0], :] = guidance[1].to(x_in.dtype)
        ops.denoiser_pack_input(x_in, prev_m, ind_in, feats)
        dec = net.trunk(feats, tok_person, mem, dtype, kv_list=kv_list, row0_add=emb_all[t], cross_list=cross_list)
        if separate is None:
            res = ops.heads_static_mix(dec, stat, Lp + L, dm, nb, net.use_head_alpha, net.regularize_alpha == "sigmoid")
        else:
            # diagnostic variant: stream bookkeeping in host-library tensor algebra on (N, 110, 4, 67)-sized data
            dyn, alpha_t = dec[..., :dm], dec[..., dm:]
            if separate.get("alpha_mod") is not None:
                alpha_t = separate["alpha_mod"](alpha_t)
            sf = stat_full[:, None]  # (N, 1, nb, dm)
            if net.use_head_alpha:
                static = (sf * alpha_t.unsqueeze(-1)).sum(dim=2)
            else:
                static = torch.cat([(sf[..., :-3] * alpha_t.unsqueeze(-1)).sum(dim=2),
                                    sf[..., -3:].sum(dim=2).expand(-1, Lp + L, -1)], dim=-1)
            res = dyn + static
        if dynamic_threshold:
            # K15 (off in the reference's inference driver, inference.py:272): quantile + clamp in one launch
            dt_ratio, dt_min, dt_max = dynamic_threshold
            res = ops.dynamic_threshold_(res.float().contiguous(), L, dt_ratio, dt_min, dt_max)
        if separate is None:
            ops.cfg_ddpm_step(x, res, z, scales, n_entries, Lp, mode, target, c0, c1, sigma)
        else:
            streams = [list(v.contiguous().clone().chunk(n_entries)) for v in (res, static, dyn, alpha_t)]
            heads = [st[0][:, -L:] for st in streams]  # views: in-place accumulation as the reference (model.py:590-618)
            for i in range(n_entries - 1):
                for st, hd in zip(streams, heads):
                    base = st[0] if cfg_mode == "independent" else st[i]
                    hd += cfg_scale[i] * (st[i + 1][:, -L:] - base[:, -L:])
            theta, theta_static, theta_dyn, theta_alpha = heads
            zz = z if z is not None else torch.zeros_like(x)
            if target == 1:
                x = c0 * (x - c1 * theta) + sigma * zz
            else:
                x = c0 * x + c1 * theta + sigma * zz
            cum_static = cum_static + c1 * theta_static
            alpha_traj.append(theta_alpha)
        if ret_traj:
            traj[t - 1] = x.clone()
    if ret_traj:
        return traj, motion_at_T, audio_feat
    if separate is not None:
        last_alpha = torch.cat(alpha_traj, dim=0) if separate.get("return_all_alpha") else theta_alpha
        return x, motion_at_T, audio_feat, theta_dyn, cum_static, last_alpha
    return x, motion_at_T, audio_feat


# clamped to [1, 50]: k bodies in one graph keep k steps of intermediates alive in the graph's private pool (about 0.2 GB per
# step at B = 64, fp16: 2 GB at the default 10 of the 288 GB)
STEPS_PER_GRAPH = min(50, max(1, int(os.environ.get("MSMD_SAMPLER_STEPS_PER_GRAPH", "10"))))


# Lanes: the batch of a hipGraph loop is cut into LANES contiguous groups of clips; every group runs its own chain of denoising
# steps on a HIP stream of its own, forked and joined inside each captured graph.  Sequences are independent, so
# every lane computes exactly what it would compute alone; the lanes drift against each other, and one lane's launch tails,
# epilogue store bursts and small grids (person-token attention, heads, CFG / DDPM update) run under another lane's K loops.
# Used from MIN_LANE_SEQS sequences per lane up (a lane must still fill the chip on its own); MSMD_SAMPLER_LANES=1 turns it off.
LANES = min(4, max(1, int(os.environ.get("MSMD_SAMPLER_LANES", "2"))))
MIN_LANE_SEQS = 48


def _step_noise(B, L, dm, dev):
    """z ~ N(0, I) of one denoising step for the whole batch (reference model.py:383: torch.randn_like(x)); tests patch this."""
    return torch.randn(B, L, dm, device=dev, dtype=torch.float32)


class _Lane:
    """Static operand buffers + the step body of one lane (Bl clips x n_entries CFG entries, entry-major rows)."""

    def __init__(self, net, dtype, dev, T, Bl, n_entries, Lp, L, dm, nb, mode, target, P, like, shared, rows, clips, dyn):
        Nl = Bl * n_entries
        take = lambda t: torch.zeros((len(rows),) + tuple(t.shape[1:]), device=dev, dtype=t.dtype)
        self.rows, self.clips = rows, clips          # index tensors into the N-row / B-row operands of the whole batch
        self.x = torch.zeros(Bl, L, dm, device=dev, dtype=torch.float32)
        self.prev_m = take(like["prev_m"])
        self.ind = take(like["ind_in"]) if like["ind_in"] is not None else None
        self.mem = take(like["mem"])
        self.kv = [take(k) for k in like["kv_list"]]
        self.cross = [take(r) for r in like["cross_list"]] if like.get("cross_list") is not None else None
        self.stat_per_clip = bool(like["stat_per_clip"])     # stated by sample(): one row of static bases per clip (not one shared row)
        self.stat = (torch.zeros((Bl,) + tuple(like["stat"].shape[1:]), device=dev, dtype=like["stat"].dtype)
                     if self.stat_per_clip else shared["stat"])
        self.tok = take(like["tok_person"])
        self.t_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.emb_row = torch.zeros(shared["emb_all"].shape[-1], device=dev, dtype=shared["emb_all"].dtype)
        self.coefs = torch.zeros(3, device=dev, dtype=torch.float32)
        self.feats = torch.zeros(Nl, 1 + Lp + L, P.kp_feat, device=dev, dtype=dtype)

        def body(z):
            ops.sampler_step_select(shared["emb_all"], shared["coef_table"], self.t_dev, self.emb_row, self.coefs)
            ops.denoiser_pack_input(self.x, self.prev_m, self.ind, self.feats)
            dec = net.trunk(self.feats, self.tok, self.mem, dtype, kv_list=self.kv, row0_add=self.emb_row,
                            cross_list=self.cross)
            res = ops.heads_static_mix(dec, self.stat, Lp + L, dm, nb, net.use_head_alpha,
                                       net.regularize_alpha == "sigmoid")
            if dyn:
                res = ops.dynamic_threshold_(res.float().contiguous(), L, *dyn)
            # z: this lane's clips of the step's noise, drawn for the WHOLE batch on the forking stream (_StepGraph.bodies): what a
            # clip receives under a given seed does not depend on the lane count; sigma_1 = 0 reproduces z = 0 at t = 1
            ops.cfg_ddpm_step_dev(self.x, res, z, shared["scales"], self.coefs, n_entries, Lp, mode, target)
        self.body = body

    def load(self, motion_at_T, ops_in):
        self.x.copy_(motion_at_T[self.clips])
        for name in ("prev_m", "mem"):
            getattr(self, name).copy_(ops_in[name][self.rows])
        if self.stat_per_clip:
            self.stat.copy_(ops_in["stat"][self.clips])
        self.tok.copy_(ops_in["tok_person"][self.rows])
        if self.ind is not None:
            self.ind.copy_(ops_in["ind_in"][self.rows])
        for dst, src in zip(self.kv, ops_in["kv_list"]):
            dst.copy_(src[self.rows])
        if self.cross is not None:
            for dst, src in zip(self.cross, ops_in["cross_list"]):
                dst.copy_(src[self.rows])


class _StepGraph:
    """k captured denoise steps per lane (one hipGraph): device-side step counters, static operand buffers."""

    def __init__(self, model, net, dtype, dev, T, N, n_entries, Lp, L, dm, nb, mode, target, P, like, dyn=None, lanes=1):
        B = N // n_entries
        self.lanes = lanes
        self.shared = dict(lanes=lanes, emb_all=torch.zeros_like(like["emb_all"]), stat=torch.zeros_like(like["stat"]),
                           scales=torch.zeros_like(like["scales"]) if like["scales"] is not None else None,
                           coef_table=torch.zeros(T + 1, 3, device=dev, dtype=torch.float32))
        Bl = B // lanes
        self.lane = []
        for l in range(lanes):
            clips = torch.arange(l * Bl, (l + 1) * Bl, device=dev)
            rows = torch.cat([clips + e * B for e in range(n_entries)])
            self.lane.append(_Lane(net, dtype, dev, T, Bl, n_entries, Lp, L, dm, nb, mode, target, P, like, self.shared,
                                   rows, clips, dyn))
        self.T = T
        self.streams = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else [None]

        def bodies(k):
            # one (B, L, dm) draw per step from the graph-safe philox stream, on the forking stream, sliced per lane: the same
            # generator calls whatever the lane count (each lane drawing its own made seeded output depend on MSMD_SAMPLER_LANES)
            zs = [_step_noise(B, L, dm, dev) for _ in range(k)]
            if lanes == 1:
                for s_ in range(k):
                    self.lane[0].body(zs[s_])
                return
            cur = torch.cuda.current_stream()
            for li, (ln, st) in enumerate(zip(self.lane, self.streams)):     # fork ... every lane records its k steps back to back ...
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    for s_ in range(k):
                        ln.body(zs[s_][li * Bl:(li + 1) * Bl])
            for st in self.streams:                         # ... join
                cur.wait_stream(st)
        # With more than one lane the LayerNorm-epilogue GEMMs (QKV, out-projection, FFN-2 of a lane: M = 10 656 rows at B = 64) take
        # the 192 x 128 tile: alone on the chip its grids are short of a round (672 / 224 tiles on 512 slots) and the 128 x 128
        # tile wins, but beside another lane's launches the holes are filled and the larger tile's better loop shows (same box,
        # alternating: 2.286 -> 2.237, 2.208 -> 2.177 ms per step).  Same products in the same order: results are bit-identical.
        lane_tile = int(os.environ.get("MSMD_SAMPLER_LANE_TILE", "15")) or None      # developers' A/B: 0 = the library's own choice
        tile_keep, ops.GEMM_LN_TILE = ops.GEMM_LN_TILE, (lane_tile if lanes > 1 else ops.GEMM_LN_TILE)
        # warm-up on a side stream (allocator + lazy kernel loading), then capture
        for ln in self.lane:
            ln.t_dev.fill_(1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bodies(1)
        torch.cuda.current_stream().wait_stream(side)
        # STEPS_PER_GRAPH consecutive denoising steps per captured graph (the step counter lives on the device, so the body is
        # simply recorded that many times): T / k replays instead of T.  k = the largest divisor of T up to the cap, so one
        # graph serves the whole loop.
        self.k = max(d for d in range(1, min(STEPS_PER_GRAPH, T) + 1) if T % d == 0)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with ops.capture_guard():
                with torch.cuda.graph(self.graph):
                    bodies(self.k)
        finally:
            ops.GEMM_LN_TILE = tile_keep

    def run(self, T, motion_at_T, ops_in, coefficients):
        for ln in self.lane:
            ln.load(motion_at_T, ops_in)
        self.shared["emb_all"].copy_(ops_in["emb_all"])
        self.shared["stat"].copy_(ops_in["stat"])
        if self.shared["scales"] is not None:
            self.shared["scales"].copy_(ops_in["scales"])
        tab = torch.zeros(T + 1, 3)
        for t in range(1, T + 1):
            c0, c1, sg = coefficients(t)
            tab[t, 0], tab[t, 1], tab[t, 2] = c0, c1, (sg if t > 1 else 0.0)
        self.shared["coef_table"].copy_(tab)
        for ln in self.lane:
            ln.t_dev.fill_(T)
        for _ in range(T // self.k):
            self.graph.replay()
        return torch.cat([ln.x for ln in self.lane], dim=0) if self.lanes > 1 else self.lane[0].x.clone()


def _graph_loop(model, net, dtype, dev, T, N, n_entries, Lp, L, dm, nb, mode, target, P, motion_at_T, prev_m, ind_in,
                mem, kv_list, stat, tok_person, emb_all, scales, coefficients, dyn=None, cross_list=None):
    B = N // n_entries
    like = dict(prev_m=prev_m, ind_in=ind_in, mem=mem, kv_list=kv_list, cross_list=cross_list, stat=stat, tok_person=tok_person,
                emb_all=emb_all, scales=scales, stat_per_clip=(stat.shape[0] == B and B > 1))
    lanes = getattr(model, "sampler_lanes", LANES)
    while lanes > 1 and (B % lanes or N // lanes < MIN_LANE_SEQS):
        lanes -= 1
    key = (T, N, n_entries, Lp, L, mode, target, dtype, ind_in is not None, getattr(net, "_pack_gen", 0), dyn, cross_list is not None,
           STEPS_PER_GRAPH, lanes)
    cache = model.__dict__.setdefault("_step_graphs", {})
    g = cache.get(key)
    if g is None:
        cache.clear()  # one resident graph (its private memory pool holds all step intermediates)
        g = cache[key] = _StepGraph(model, net, dtype, dev, T, N, n_entries, Lp, L, dm, nb, mode, target, P, like, dyn, lanes)
    return g.run(T, motion_at_T.float(), like, coefficients)
