"""Training step on the MI355X (drop-in counterpart of reference training_script.py:49-241, 406-438, 548-579).

`Trainer.step` reproduces one iteration of the reference's `train()` loop -- two consecutive 100-frame windows of
the same clips, VAE style codes, optional cross-style swap and truncation, window 0 handing its last 10 motion /
audio-feature frames to window 1, parameter-space losses + KL, one backward, Adam with linear warm-up -- on the
HIP autograd path (train_graph.py), data-parallel over one process per GPU:

  * gradients live in one flat fp32 arena; bucketed all-reduce (RCCL over xGMI, backend "nccl") is launched from
    post-accumulate-grad hooks on a side stream while backward continues (dp.GradBucketReducer);
  * parameters live in one flat fp32 arena updated by ONE fused Adam launch (msmd_adam_step), the 1/world_size
    gradient scaling folded in;
  * the KL term is a batch SUM in the reference (utils/common.py:454): its weight is multiplied by world_size so
    that DP over N ranks equals a single-process run on the global batch;
  * `it % gradient_accumulation_steps == 0` stepping as the reference (training_script.py:199).

  * model.train() / model.eval() select the reference's training-time noise (dropout everywhere the reference's
    modules have it, HF LayerDrop, SpecAugment; see train_graph.py) exactly as `model.train()` does at
    training_script.py:55; masks come from a Philox stream keyed by a device-side (seed, step) pair, so the
    hipGraph replay of iteration k draws iteration k's masks.

Not reproduced (documented): the reference's 3x empty_cache()+gc per iteration and its per-step .item() syncs
(losses are returned as device tensors; log asynchronously).
The dataset / loader / tensorboard / CLI plumbing of the reference script is out of scope (SURVEY.md section 2
rows 17-18); `synthetic_batch` produces the loader's tensor contract.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch

from . import autograd as ag
from . import dp, ops, synth
from . import train_graph as tg


def load_loss_weights(args, device=None):
    """reference training_script.py:406-438 (python floats; same keys and rescaling)."""
    w = {"noise": 1.0, "vert": float(args.l_vert), "vel": float(args.l_vel), "smooth": float(args.l_smooth),
         "head_angle": float(args.l_head_angle), "head_vel": float(args.l_head_vel),
         "head_smooth": float(args.l_head_smooth), "head_trans": float(args.l_head_trans)}
    if not args.use_vertex_space:
        w["vel"] *= 4.5e-8
        w["smooth"] *= 4e-7
    legacy = args.dataset_type[:9] == "HDTF_TFHP" or args.dataset_type == "flame_mead_ravdess"
    if not legacy and args.use_vertex_space:
        w["vert"] *= 1e-7
        w["vel"] *= 1e-7
        w["smooth"] *= 2e-8
    w["kl_div"] = float(args.l_kl_div)
    return w


def synthetic_batch(B, rank=0, device="cuda", it=0):
    """([audio_0, audio_1], [motion_0, motion_1], shape) with the loader's shapes / normalisation."""
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    tag = f"train_r{rank}_it{it}"
    return ([t(synth.audio_clips(B, 64000, tag=f"{tag}_a{i}")) for i in range(2)],
            [t(synth.motion_clips(B, tag=f"{tag}_m{i}")) for i in range(2)],
            torch.zeros(B, 100, device=device))


@torch.no_grad()
def test(args, loss_weights, model, style_enc, test_loader, current_iter=0, n_rounds=10, mode="val", writer=None,
         flame=None, out_abc_dir=None, do_save=False, do_save_path=None, do_ignore_style=False, coef_stats=None):
    """Validation / test pass (reference training_script.py:243-403, same signature): eval-mode forward of both
    windows with the inference kernels, the other window's style when use_cross_style (l.311-312), no truncation,
    the HIP loss kernels, KL, per-batch weighted sums; returns the reference's loss_log (lists, or
    {mean, std, n_samples} per key when do_save, also written to do_save_path as JSON).  `test_loader` is any iterable
    of the loader tuple (audio_pair, coef_pair, audio_stats), e.g. datasets.ResidentDataset.batch outputs."""
    import json
    from collections import defaultdict
    from .utils.common import compute_KL_loss, compute_loss, compute_loss_no_vert
    was_training = model.training
    model.eval()
    device = model.device
    if coef_stats is None:
        ds = getattr(test_loader, "dataset", None)
        coef_stats = getattr(ds, "coef_stats", None)
    if coef_stats is not None:
        coef_stats = {k: v.to(device) for k, v in coef_stats.items()}
    legacy = args.dataset_type[:9] == "HDTF_TFHP" or args.dataset_type == "flame_mead_ravdess"
    loss_log = defaultdict(list)
    for _ in range(n_rounds):
        for audio_pair, coef_pair, _stats in test_loader:
            audio_pair = [a.to(device) for a in audio_pair]
            motion_pair = [coef_pair[i]["motion"].to(device) for i in range(2)]
            shape = coef_pair[0]["shape"].to(device)
            shape_coef = shape.clone() if shape.ndim == 2 else shape[:, 0].clone()
            src = [torch.zeros_like(m) for m in motion_pair] if do_ignore_style else motion_pair
            enc = [style_enc(src[i]) for i in range(2)]
            style_pair, mu_pair, logvar_pair = ([e[j] for e in enc] for j in range(3))
            losses = {k: torch.zeros((), device=device) for k in loss_weights}
            prev_motion = prev_audio = None
            for i in range(2):
                style = style_pair[1 - i] if args.use_cross_style else style_pair[i]
                B = audio_pair[i].shape[0]
                indicator = torch.ones(B, args.n_motions, device=device) if args.use_indicator else None
                shape_in = shape_coef if not getattr(args, "do_ignore_shape", False) else torch.zeros_like(shape_coef)
                cfg = not getattr(args, "do_ignore_cfg", False)
                if i == 0:
                    noise, target, pm, pa = model(motion_pair[i], audio_pair[i], shape_in, style, indicator=indicator,
                                                  train_with_CFG=cfg)
                    prev_motion, prev_audio = pm[:, -args.n_prev_motions:], pa[:, -args.n_prev_motions:]
                else:
                    noise, target, _, _ = model(motion_pair[i], audio_pair[i], shape_in, style, prev_motion, prev_audio,
                                                indicator=indicator, train_with_CFG=cfg)
                fn = compute_loss if (args.use_vertex_space and legacy) else compute_loss_no_vert
                ld = fn(args, i == 0, shape_coef, motion_pair[i], noise, target, prev_motion, coef_stats, flame, None,
                        return_dict=True)
                ld["kl_div"] = compute_KL_loss(mu_pair[i], logvar_pair[i])
                for k, v in ld.items():
                    if loss_weights.get(k, 0) > 0 and v is not None:
                        losses[k] = losses[k] + v
            total = 0.0
            for k in losses:
                if loss_weights[k] > 0:
                    loss_log[k].append(float(losses[k]))
                    total = total + losses[k] * loss_weights[k]
            loss_log["loss"].append(float(total))
    if writer is not None:
        writer.add_scalar(f"{mode}/loss", float(np.mean(loss_log["loss"])), current_iter)
        for k in loss_log:
            if k != "loss" and loss_weights[k] > 0:
                writer.add_scalar(f"{mode}/{k}", float(np.mean(loss_log[k])), current_iter)
    if do_save:
        for k in list(loss_log):
            vals = loss_log[k]
            loss_log[k] = {"mean": float(np.mean(vals)), "std": float(np.std(vals)), "n_samples": len(vals)}
        with open(do_save_path, "w") as f:
            json.dump(loss_log, f)
    if was_training:
        model.train()
    return loss_log


def batch_from_loader(item):
    """(audio_pair, coef_pair, audio_stats) as the reference's loader / datasets.ResidentDataset.batch yields it ->
    the step's (audio_pair, motion_pair, shape) (training_script.py:77-93: motion = coef['motion'], shape = the first
    frame's shape row)."""
    audio_pair, coef_pair, _ = item
    shape = coef_pair[0]["shape"]
    shape = shape.clone() if shape.ndim == 2 else shape[:, 0].clone()
    return list(audio_pair), [coef_pair[0]["motion"], coef_pair[1]["motion"]], shape


class Trainer:
    def __init__(self, args, model, style_enc, process_group=None, bucket_mb=32.0, use_graph=False, flame=None,
                 coef_stats=None, comm=None, bucket_dtype=None, exchange_at_world_1=False, batch_windows=None):
        self.args, self.model, self.style_enc = args, model, style_enc
        # both windows of an iteration through each network as one batch of 2 B rows (_two_windows_batched); False or
        # MSMD_TRAIN_BATCH_WINDOWS=0: window 0, then window 1 (the reference's order)
        self.batch_windows = (os.environ.get("MSMD_TRAIN_BATCH_WINDOWS", "1") != "0") if batch_windows is None else bool(batch_windows)
        self.device = model.device
        # vertex-space training branch (reference training_script.py:167-170: use_vertex_space on the legacy FLAME
        # dataset types): the loss runs through FLAME, differentiably (train_graph.loss_vert_train)
        legacy = args.dataset_type[:9] == "HDTF_TFHP" or args.dataset_type == "flame_mead_ravdess"
        self.vertex_space = bool(getattr(args, "use_vertex_space", False)) and legacy
        self.flame = flame
        self.coef_stats = {k: torch.as_tensor(v).float().to(self.device) for k, v in coef_stats.items()} if coef_stats else None
        if self.vertex_space and flame is None:
            raise ValueError("use_vertex_space on a FLAME dataset type needs the FLAME module (Trainer(..., flame=...))")
        # optimizer param groups of the reference: style encoder first, then the model (same lr)
        params = [p for p in style_enc.parameters() if p.requires_grad] + \
                 [p for p in model.parameters() if p.requires_grad]
        # the encoder layers' Q / K / V projections next to each other in the arenas: the fused QKV operand is a view
        dp.ADJACENT = tg.adjacent_parameter_groups(model)
        try:
            self.flat_param = dp.flatten_parameters(params)
            self.reducer = dp.GradBucketReducer(params, bucket_mb=bucket_mb, process_group=process_group, comm=comm,
                                                bucket_dtype=bucket_dtype, exchange_at_world_1=exchange_at_world_1)
        finally:
            dp.ADJACENT = []
        if self.reducer.world > 1:
            # replicas start from rank 0's parameters whatever each rank's generator drew at construction (a fresh run draws
            # the non-encoder parameters from torch's RNG, msmd_amd.init); frozen tensors come from the checkpoint on every rank
            self.reducer.td.broadcast(self.flat_param, src=self.reducer.td.get_global_rank(process_group, 0)
                                      if process_group is not None else 0, group=process_group)
        self.weight_arena = None
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.opt_step = 0
        self.loss_weights = load_loss_weights(args)
        self.rng = np.random.RandomState(1234 + (dp.env_rank()[0]))
        self.lr = float(args.lr)
        self.warm_iter = int(getattr(args, "warm_iter", 0) or 0)
        self.sched_step = 0
        self._build_lr_schedule()
        self.use_graph = bool(use_graph)
        # truncation geometry as the reference's train() takes it from the dataset / args (training_script.py:67,125)
        self.audio_unit = float(getattr(args, "audio_unit", None) or 16000.0 / float(args.fps))
        pad_mode = getattr(args, "pad_mode", "zero")
        if pad_mode not in ("zero", "replicate"):
            raise ValueError(f"Unknown pad mode {pad_mode}!")   # reference utils/common.py:777
        self.pad_replicate = pad_mode == "replicate"
        self._graphs, self._graph_pool, self._flag_table = {}, None, None
        # hipGraph mode with more than one rank: capture the iteration as bucket-aligned SEGMENTS so every bucket's
        # all-reduce starts as soon as its gradients are final (MSMD_SEGMENT_GRAPHS=1 forces it on one rank: tests)
        self.segment_graphs = self.use_graph and (self.reducer.world > 1 or os.environ.get("MSMD_SEGMENT_GRAPHS") == "1"
                                                  or self.reducer.exchange_at_world_1)
        self._stepping = True
        # weight gradients are added straight into the arena by the wgrad GEMM when bucket launches do not hang
        # on per-parameter autograd hooks (graph mode launches them from finish(); world 1 launches nothing)
        self.direct_grad = self.use_graph or self.reducer.world == 1
        for key in [k for k in ag.FUSED if k[1] in (id(model.audio_encoder), id(model.denoising_net))]:
            del ag.FUSED[key]       # a recycled module id must never meet an older trainer's views
        aliases, only_aliased = (tg.build_fused_aliases(model, self.flat_param, self.reducer.arena)
                                 if self.direct_grad and model.compute_dtype == torch.bfloat16 else ([], []))
        self._aliases = aliases     # keeps the alias tensors (and with them their cache keys) alive
        if model.compute_dtype == torch.bfloat16:
            self.weight_arena = ag.WeightArena(self.flat_param, params, aliases=aliases, skip=only_aliased)
        # Philox (seed, step) for dropout masks: device memory, advanced once per iteration
        self.noise_state = torch.tensor([0x5EED0000 + 7919 * dp.env_rank()[0], 0], dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------ learning-rate schedule
    def _build_lr_schedule(self):
        from .utils.scheduler import LrSchedule
        self._lr = LrSchedule(self.args)

    def current_lr(self):
        # tests poke `self.lr` when no scheduler is configured
        return self._lr.lr if self._lr.sched is not None else self.lr

    def _scheduler_step(self, it):
        self._lr.step(it)

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path, it):
        """The reference's checkpoint dict (training_script.py:227-233: 'args', 'model', 'style_enc', 'iter'; loadable
        by the reference's load_pretrained_model) plus resumable-training extensions the reference lacks (it restarts
        Adam and the warm-up from scratch): 'optimizer' (flat Adam moments + step), 'scheduler', 'rng'."""
        torch.save({
            "args": self.args,
            "model": self.model.state_dict(),
            "style_enc": self.style_enc.state_dict(),
            "iter": int(it),
            "optimizer": {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "step": self.opt_step,
                          "layout": [(n, int(p.numel())) for n, p in self._named_trainable()],
                          "arena_order": self._arena_names()},
            "scheduler": {"sched_step": self.sched_step, "n_calls": self._lr.calls, "lr": self.current_lr()},
            "rng": {"numpy": self.rng.get_state(), "noise_state": self.noise_state.cpu(), "rank": dp.env_rank()[0]},
        }, path)

    def load_checkpoint(self, path_or_dict):
        """Restores weights (reference key names; reference-written checkpoints load as they are) and, when present,
        the optimizer / scheduler / RNG extensions.  Returns the stored iteration."""
        ck = torch.load(path_or_dict, map_location="cpu", weights_only=False) if not isinstance(path_or_dict, dict) \
            else path_or_dict
        self.model.load_state_dict(ck["model"])
        self.style_enc.load_state_dict(ck["style_enc"])
        self._invalidate_caches()
        opt = ck.get("optimizer")
        if opt is not None:
            if [(n, k) for n, k in opt["layout"]] != [(n, int(p.numel())) for n, p in self._named_trainable()]:
                raise ValueError("optimizer state layout does not match this model's trainable parameters")
            # the flat moments are stored in the WRITER's arena order (older checkpoints: plain reverse registration
            # order; since round 3 the encoder's Q / K / V parameters are pulled together): copy parameter by parameter
            names = dict((n, p) for n, p in self._named_trainable())
            saved_order = opt.get("arena_order") or [n for n, _ in self._named_trainable()][::-1]
            base = self.flat_param.data_ptr()
            need = sum((names[n].numel() + dp.ALIGN - 1) // dp.ALIGN * dp.ALIGN for n in saved_order)
            if need != int(opt["exp_avg"].numel()) or need != int(opt["exp_avg_sq"].numel()):
                raise ValueError(f"optimizer moments hold {int(opt['exp_avg'].numel())} elements, the checkpoint's arena order "
                                 f"accounts for {need}")
            off = 0
            for n in saved_order:
                p = names[n]
                k = p.numel()
                dst = (p.data_ptr() - base) // 4
                self.exp_avg[dst:dst + k].copy_(opt["exp_avg"][off:off + k])
                self.exp_avg_sq[dst:dst + k].copy_(opt["exp_avg_sq"][off:off + k])
                off += (k + dp.ALIGN - 1) // dp.ALIGN * dp.ALIGN
            self.opt_step = int(opt["step"])
        sch = ck.get("scheduler")
        if sch is not None:   # the schedulers are pure functions of their call count: rebuild and replay
            self.sched_step = int(sch["sched_step"])
            self._build_lr_schedule()
            if self._lr.sched is not None:
                self._lr.replay(sch["n_calls"])
        rng = ck.get("rng")
        if rng is not None:
            rank = dp.env_rank()[0]
            if rank == int(rng.get("rank", 0)):
                self.rng.set_state(rng["numpy"])
                self.noise_state.copy_(rng["noise_state"])
            else:
                # the checkpoint holds ONE rank's streams: every other rank re-derives its own from (rank, iteration)
                # instead of cloning them (identical dropout / SpecAugment / CFG draws on all ranks otherwise)
                it = int(ck.get("iter", 0))
                self.rng = np.random.RandomState((1234 + rank + 7919 * (it + 1)) % (2 ** 32))
                self.noise_state.copy_(torch.tensor([0x5EED0000 + 7919 * rank, int(rng["noise_state"][1])],
                                                    dtype=torch.int64))
        return int(ck.get("iter", 0))

    def _arena_names(self):
        """Trainable parameter names in arena order (the order of the flat Adam moments)."""
        base = self.flat_param.data_ptr()
        return [n for n, p in sorted(self._named_trainable(), key=lambda np_: np_[1].data_ptr() - base)]

    def _named_trainable(self):
        out = [("style_enc." + n, p) for n, p in self.style_enc.named_parameters() if p.requires_grad]
        return out + [("model." + n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    def _invalidate_caches(self):
        ag.CACHE.clear()
        if self.weight_arena is not None:
            self.weight_arena.refresh()   # ONE launch: bf16 casts + transposes of every trainable Linear weight
        self.model.denoising_net._packed = None
        self.model._afm = None
        self.style_enc._packed = None
        enc = self.model.audio_encoder
        enc._packed = None  # lazily re-packed by the inference paths (the training graph reads parameters directly)
        if any(p.requires_grad for n, p in enc.named_parameters() if n.startswith("feature_extractor.")):
            enc._packed_fe = None   # the conv stack's pack survives optimizer steps while it is frozen

    # ------------------------------------------------------------------ forward + backward of one iteration
    def _fwd_bwd(self, batch, draws, trunc, cross):
        """Two windows, losses, one backward into the flat gradient arena.  `trunc` = [bool, bool] (host decision:
        is window i truncated); `cross` = [bool | 0-d bool device tensor] * 2 (use the other window's style).
        Everything stochastic that is not in `draws` is drawn ON THE DEVICE, and nothing reads back to the host,
        so the whole function can be captured in a hipGraph."""
        args, model, se = self.args, self.model, self.style_enc
        dtype = model.compute_dtype
        audio_pair, motion_pair, shape = batch
        B = audio_pair[0].shape[0]
        n_prev = args.n_prev_motions
        lw = dict(self.loss_weights)
        lw["kl_div"] *= self.reducer.kl_weight_scale
        noise = ag.TrainNoise
        noise.active = bool(model.training)
        noise.state, noise.host_rng = self.noise_state, self.rng
        noise.begin()
        with torch.enable_grad():
            if self.batch_windows:
                terms = self._two_windows_batched(batch, draws, trunc, cross, lw, dtype)
            else:
                terms = self._two_windows_in_turn(batch, draws, trunc, cross, lw, dtype)
            losses, loss = self._combine_losses(terms, lw)
            loss.backward()
        noise.active = False    # module-level switch: never leak train-mode noise into other callers of the graph
        out = {k: v.detach() for k, v in losses.items()}
        out["loss"] = loss.detach()
        return out

    def _two_windows_in_turn(self, batch, draws, trunc, cross, lw, dtype):
        """The reference's order (training_script.py:99-195): window 0 through the whole model, then window 1.  Kept as the
        comparison form of _two_windows_batched (MSMD_TRAIN_BATCH_WINDOWS=0)."""
        args, model, se = self.args, self.model, self.style_enc
        audio_pair, motion_pair, shape = batch
        B = audio_pair[0].shape[0]
        n_prev = args.n_prev_motions
        noise = ag.TrainNoise
        styles, mus, logvars = [], [], []
        for i in range(2):
            mu, logvar = tg.style_encoder_train(se, motion_pair[i], dtype)
            e = draws["style_eps"][i] if "style_eps" in draws else torch.randn_like(mu)
            styles.append(mu + e * torch.exp(0.5 * logvar))
            mus.append(mu)
            logvars.append(logvar)
        terms = {k: [] for k in lw}     # every loss term of both windows, summed per key and weighted in ONE pass below
        prev_motion = prev_audio = None
        for i in range(2):
            audio, motion = audio_pair[i], motion_pair[i]
            if torch.is_tensor(cross[i]):
                style = torch.where(cross[i], styles[1 - i], styles[i])
            else:
                style = styles[1 - i] if cross[i] else styles[i]
            if "end_idx" in draws:
                end_idx = draws["end_idx"][i]
            else:
                end_idx = torch.randint(1, args.n_motions, (B,), device=self.device) if trunc[i] else None
            if end_idx is not None:
                # reference utils/common.py:816-832: audio is cut at (end_idx * audio_unit).long() with the DATASET's
                # audio_unit = 16000 / fps (a float: 533.33 at 30 fps) and both tensors padded per args.pad_mode
                e32 = end_idx.to(torch.int32).contiguous()
                a32 = (end_idx.to(torch.float32) * self.audio_unit).long().to(torch.int32).contiguous()
                audio_in = ops.truncate_rows_(audio.float().clone().contiguous(), a32, 1, self.pad_replicate)
                motion_in = ops.truncate_rows_(motion.float().clone().contiguous(), e32, 1, self.pad_replicate)
                indicator = (torch.arange(args.n_motions, device=self.device).expand(B, -1) < end_idx.unsqueeze(1)).float()
            else:
                audio_in, motion_in = audio, motion
                indicator = torch.ones(B, args.n_motions, device=self.device)
            ts = draws["t"][i] if "t" in draws else model.diffusion_sched.uniform_sample_t_device(B)
            eps = draws["eps"][i] if "eps" in draws else torch.randn_like(motion_in)
            ns, na = self._cfg_masks(draws, i, B)
            shape_in = torch.zeros_like(shape) if getattr(args, "do_ignore_shape", False) else shape
            _, target, _, audio_feat = tg.msmd_forward_train(model, motion_in, audio_in, shape_in, style, prev_motion,
                                                             prev_audio, ts, indicator, eps, ns, na)
            if i == 0:
                if end_idx is not None:  # truncated: hand over the COMPLETE clip's features (training_script.py:152-155)
                    prev_motion = motion[:, -n_prev:]
                    with torch.no_grad():
                        if noise.active:   # the reference's extra pass also runs under model.train()
                            prev_audio = tg.audio_feat_train(model, audio, model.n_motions, dtype).float()[:, -n_prev:]
                        else:
                            prev_audio = model.extract_audio_feature(audio)[:, -n_prev:]
                else:
                    prev_motion = motion_in[:, -n_prev:].detach()
                    prev_audio = audio_feat[:, -n_prev:]
            if self.vertex_space:
                ld = tg.loss_vert_train(args, i == 0, shape, motion_in, target, prev_motion if i == 1 else None,
                                        self.coef_stats, self.flame, end_idx)
                pairs = ld.items()
            else:
                tup = tg.loss_no_vert_train(args, i == 0, motion_in, target, prev_motion if i == 1 else None, end_idx,
                                            halve=False)     # the / 2 of the first six terms rides in _combine_losses
                pairs = zip(("noise", "vel", "smooth", "head_angle", "head_vel", "head_smooth", "head_trans"),
                            [(v, 0.5) for v in tup[:6]] + [(tup[6], 1.0)])
            for key, val in pairs:
                val, sc = val if isinstance(val, tuple) else (val, 1.0)
                if val is not None and torch.is_tensor(val) and lw.get(key, 0) > 0:
                    terms[key].append((val, sc))
            terms["kl_div"].append((tg.kl_train(mus[i], logvars[i]), 1.0))
        return terms

    def _window_inputs(self, i, batch, draws, trunc, cross, styles):
        """What window i feeds the model, drawn / derived exactly as the in-turn form does (reference training_script.py
        :120-150, utils/common.py:816-832 for the truncation)."""
        args, model = self.args, self.model
        audio_pair, motion_pair, shape = batch
        B = audio_pair[0].shape[0]
        audio, motion = audio_pair[i], motion_pair[i]
        if torch.is_tensor(cross[i]):
            style = torch.where(cross[i], styles[1 - i], styles[i])
        else:
            style = styles[1 - i] if cross[i] else styles[i]
        if "end_idx" in draws:
            end_idx = draws["end_idx"][i]
        else:
            end_idx = torch.randint(1, args.n_motions, (B,), device=self.device) if trunc[i] else None
        if end_idx is not None:
            e32 = end_idx.to(torch.int32).contiguous()
            a32 = (end_idx.to(torch.float32) * self.audio_unit).long().to(torch.int32).contiguous()
            audio_in = ops.truncate_rows_(audio.float().clone().contiguous(), a32, 1, self.pad_replicate)
            motion_in = ops.truncate_rows_(motion.float().clone().contiguous(), e32, 1, self.pad_replicate)
            indicator = (torch.arange(args.n_motions, device=self.device).expand(B, -1) < end_idx.unsqueeze(1)).float()
        else:
            audio_in, motion_in = audio, motion
            indicator = torch.ones(B, args.n_motions, device=self.device)
        ts = draws["t"][i] if "t" in draws else model.diffusion_sched.uniform_sample_t_device(B)
        eps = draws["eps"][i] if "eps" in draws else torch.randn_like(motion_in)
        ns, na = self._cfg_masks(draws, i, B)
        shape_in = torch.zeros_like(shape) if getattr(args, "do_ignore_shape", False) else shape
        return dict(style=style, end_idx=end_idx, audio_in=audio_in, motion_in=motion_in, indicator=indicator,
                    ts=torch.as_tensor(ts, device=self.device, dtype=torch.long), eps=eps, ns=ns, na=na, shape_in=shape_in)

    def _two_windows_batched(self, batch, draws, trunc, cross, lw, dtype):
        """Both windows through every network as ONE batch of 2 B rows.  Window 1 takes from window 0 only the last
        n_prev frames of its ground-truth motion and of its (detached) audio features (reference training_script.py:152-160):
        nothing a window's denoiser pass computes reaches the other window, and the audio features of both windows depend on the
        audio alone -- so the style encoder, the audio encoder and the denoiser each run once on [window 0 rows | window 1
        rows], window 1's hand-off read from the encoder's output in between.  Per row the arithmetic is the in-turn form's
        (GEMM rows are independent); weight gradients sum 2 B rows in one product instead of two accumulated ones.  Train-mode
        noise keeps the reference's granularity: one SpecAugment mask and one LayerDrop coin per window and layer."""
        args, model, se = self.args, self.model, self.style_enc
        audio_pair, motion_pair, shape = batch
        B = audio_pair[0].shape[0]
        n_prev = args.n_prev_motions
        noise = ag.TrainNoise
        mu2, logvar2 = tg.style_encoder_train(se, torch.cat([motion_pair[0], motion_pair[1]], 0), dtype)
        styles, mus, logvars = [], [], []
        for i in range(2):
            mu, logvar = mu2[i * B:(i + 1) * B], logvar2[i * B:(i + 1) * B]
            e = draws["style_eps"][i] if "style_eps" in draws else torch.randn_like(mu)
            styles.append(mu + e * torch.exp(0.5 * logvar))
            mus.append(mu)
            logvars.append(logvar)
        terms = {k: [] for k in lw}
        W = [self._window_inputs(i, batch, draws, trunc, cross, styles) for i in range(2)]
        feat2 = tg.audio_feat_train(model, torch.cat([W[0]["audio_in"].float(), W[1]["audio_in"].float()], 0), model.n_motions, dtype,
                                    groups=2).float()
        if W[0]["end_idx"] is not None:  # truncated: hand over the COMPLETE clip's features (training_script.py:152-155)
            prev_motion = motion_pair[0][:, -n_prev:]
            with torch.no_grad():
                if noise.active:   # the reference's extra pass also runs under model.train()
                    prev_audio = tg.audio_feat_train(model, audio_pair[0], model.n_motions, dtype).float()[:, -n_prev:]
                else:
                    prev_audio = model.extract_audio_feature(audio_pair[0])[:, -n_prev:]
        else:
            prev_motion = W[0]["motion_in"][:, -n_prev:].detach()
            prev_audio = feat2[:B, -n_prev:].detach()
        cat = lambda k: torch.cat([W[0][k], W[1][k]], 0)
        masks = []
        for k in ("ns", "na"):
            m0, m1 = W[0][k], W[1][k]
            if m0 is None and m1 is None:
                masks.append(None)
            else:
                z = torch.zeros(B, dtype=torch.bool, device=self.device)
                masks.append(torch.cat([m0 if m0 is not None else z, m1 if m1 is not None else z], 0))
        prev_m2 = torch.cat([model.start_motion_feat.expand(B, -1, -1), prev_motion.to(model.start_motion_feat.dtype)], 0)
        prev_a2 = torch.cat([model.start_audio_feat.expand(B, -1, -1), prev_audio.to(model.start_audio_feat.dtype)], 0)
        _, target2, _, _ = tg.msmd_forward_train(model, cat("motion_in"), feat2, cat("shape_in"), cat("style"), prev_m2, prev_a2,
                                                 cat("ts"), cat("indicator"), cat("eps"), masks[0], masks[1])
        for i in range(2):
            motion_in, end_idx, target = W[i]["motion_in"], W[i]["end_idx"], target2[i * B:(i + 1) * B]
            if self.vertex_space:
                ld = tg.loss_vert_train(args, i == 0, shape, motion_in, target, prev_motion if i == 1 else None,
                                        self.coef_stats, self.flame, end_idx)
                pairs = ld.items()
            else:
                tup = tg.loss_no_vert_train(args, i == 0, motion_in, target, prev_motion if i == 1 else None, end_idx,
                                            halve=False)     # the / 2 of the first six terms rides in _combine_losses
                pairs = zip(("noise", "vel", "smooth", "head_angle", "head_vel", "head_smooth", "head_trans"),
                            [(v, 0.5) for v in tup[:6]] + [(tup[6], 1.0)])
            for key, val in pairs:
                val, sc = val if isinstance(val, tuple) else (val, 1.0)
                if val is not None and torch.is_tensor(val) and lw.get(key, 0) > 0:
                    terms[key].append((val, sc))
            terms["kl_div"].append((tg.kl_train(mus[i], logvars[i]), 1.0))
        return terms

    def _combine_losses(self, terms, lw):
        """per-key sums (for the log) and the weighted total of all loss terms: stack + two masked reductions instead of a zero
        tensor, an add per term and a multiply-add per key (about forty 0-dim launches per iteration, most of them again in
        the backward).  Same arithmetic order per key is NOT kept (fp32 sums of <= 4 terms); reference: training_script.py
        :163-195 (loss_dict accumulation and the weighted sum)."""
        keys = list(lw)
        if os.environ.get("MSMD_STACK_LOSSES", "1") == "0":     # the term-by-term form, kept for comparison
            losses = {k: sum((t * sc if sc != 1.0 else t for t, sc in terms[k]), torch.zeros((), device=self.device)) for k in keys}
            return losses, sum(losses[k] * lw[k] for k in keys if lw[k] > 0)
        flat, owner, scales = [], [], []
        for j, k in enumerate(keys):
            for t, sc in terms[k]:      # sc: a constant factor of the term (the reference's / 2), applied in the selection matrix
                flat.append(t.float().reshape(()))
                owner.append(j)
                scales.append(float(sc))
        sig = (tuple(owner), tuple(scales), tuple(float(lw[k]) for k in keys))
        cached = self._loss_consts.get(sig) if hasattr(self, "_loss_consts") else None
        if cached is None:
            if not hasattr(self, "_loss_consts"):
                self._loss_consts = {}
            sel = torch.zeros(len(keys), max(len(flat), 1))
            for n, j in enumerate(owner):
                sel[j, n] = scales[n]
            w = torch.tensor([float(lw[k]) if lw[k] > 0 else 0.0 for k in keys])
            cached = self._loss_consts[sig] = (sel.to(self.device), w.to(self.device))
        sel, w = cached
        per_key = (sel * torch.stack(flat)).sum(dim=1)
        return {k: per_key[j] for j, k in enumerate(keys)}, (per_key * w).sum()

    def _cfg_masks(self, draws, i, B):
        """Null-style / null-audio masks of the training forward (reference model.py:190-218, switched off by
        --do_ignore_cfg, training_script.py:146): incremental mode draws ONE uniform per sample (style nulled above
        0.55, audio above 0.9); a single condition or cfg_mode 'independent' draws one uniform PER condition against
        0.1 / 0.5.  Injected draws: cfg_flag[i] = tensor (incremental) or (style_u, audio_u) (independent)."""
        model, args = self.model, self.args
        conds = model.guiding_conditions
        if getattr(args, "do_ignore_cfg", False) or len(conds) == 0:
            return None, None
        if len(conds) > 2:
            raise AssertionError("Only support 1 or 2 CFG conditions!")
        inj = draws["cfg_flag"][i] if "cfg_flag" in draws else None
        if "cfg_flag" in draws and inj is None:
            return None, None
        if len(conds) == 1 or model.cfg_mode == "independent":
            p = 0.5 if len(conds) >= 2 else 0.1
            us = ua = None
            if inj is not None:
                us, ua = inj if isinstance(inj, (tuple, list)) else (inj, inj)
            ns = ((us if us is not None else torch.rand(B, device=self.device)) < p) if "style" in conds else None
            na = ((ua if ua is not None else torch.rand(B, device=self.device)) < p) if "audio" in conds else None
            return ns, na
        if model.cfg_mode != "incremental":
            raise NotImplementedError(f"Unknown cfg_mode {model.cfg_mode}")
        flag = inj if inj is not None else torch.rand(B, device=self.device)
        return ((flag > 0.55) if "style" in conds else None), ((flag > 0.9) if "audio" in conds else None)

    def launch_description(self):
        if not self.use_graph:
            return "eager launches; bucket all-reduces from post-accumulate-grad hooks on a side stream"
        if self.segment_graphs:
            return ("hipGraph segments cut where a gradient bucket becomes final; each bucket's all-reduce is launched "
                    "on a side stream between segment replays (overlaps the rest of the backward)")
        return "whole-iteration hipGraph (one rank: nothing to exchange)"

    def _host_choices(self, draws):
        """The reference's host-side coin flips (training_script.py:99-141): cross-style per window, truncation."""
        args = self.args
        cross, trunc = [], []
        for i in range(2):
            cross.append(bool(draws["cross"][i]) if "cross" in draws else
                         bool(args.use_cross_style and self.rng.rand() < args.prob_cross_style))
            if "end_idx" in draws:
                trunc.append(draws["end_idx"][i] is not None)
            else:
                trunc.append(bool(self.rng.rand() < (args.trunc_prob1 if i == 0 else args.trunc_prob2)))
        return cross, trunc

    def _optimizer_step(self):
        flat_grad, scale = self.reducer.finish()
        self.opt_step += 1
        ops.adam_step_(self.flat_param, flat_grad, self.exp_avg, self.exp_avg_sq, self.current_lr(), self.opt_step,
                       grad_scale=scale)
        self.reducer.zero_grad()
        self._invalidate_caches()

    def step(self, batch, it=1, draws=None):
        """One iteration.  batch = ([audio_0, audio_1], [motion_0, motion_1], shape).  `draws` may inject the
        stochastic choices: dict(cross=[bool, bool], end_idx=[tensor|None]*2, t=[list]*2, eps=[tensor]*2,
        style_eps=[tensor]*2, cfg_flag=[tensor|None]*2).  Returns dict of detached loss tensors (+ 'loss').
        With `use_graph` the forward+backward runs as ONE hipGraph replay (see `_graph_fwd_bwd`)."""
        draws = draws or {}
        ag.DIRECT_GRAD = self.direct_grad
        stepping = (it % max(1, self.args.gradient_accumulation_steps) == 0)
        cross, trunc = self._host_choices(draws)
        self.noise_state[1] += 1
        ag.TrainNoise.graph_safe = self.use_graph
        ag.TrainNoise.spec_masks = None
        self.reducer.begin_backward()         # re-arm per backward, not per optimizer step (gradient accumulation)
        self._stepping = stepping
        if self.use_graph:
            self.reducer.enabled = False      # python hooks do not run on replay: buckets are launched by finish()
            out = self._graph_fwd_bwd(batch, draws, trunc, cross)
        else:
            self.reducer.enabled = stepping
            out = self._fwd_bwd(batch, draws, trunc, cross)
        if stepping:
            self._optimizer_step()
        self.sched_step += 1
        self._scheduler_step(it)
        return out

    # ------------------------------------------------------------------ hipGraph mode
    def _graph_fwd_bwd(self, batch, draws, trunc, cross):
        """Replay (capturing on first use) the hipGraph of `_fwd_bwd` for this (batch size, truncation pattern,
        injected-draw signature).  The iteration is ~10^4 kernel launches and host-issue bound in eager mode; the
        replay is one launch.  Inputs are copied into static buffers; the cross-style choice is a device flag, the
        truncation pattern selects the graph variant (it changes the set of kernels: window 0 truncated adds the
        no-grad encoder pass of training_script.py:152-155); t / eps / CFG flags / end_idx are drawn inside the
        graph by torch's graph-safe Philox generator unless injected."""
        audio_pair, motion_pair, shape = batch
        B = audio_pair[0].shape[0]
        key = (B, bool(trunc[0]), bool(trunc[1]), tuple(sorted(k for k in draws if k != "cross")))
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._capture(key, batch, draws, trunc)
        sb, sd, flags, g, out, spec = ent
        for dst, src in zip(sb[0] + sb[1] + [sb[2]], list(audio_pair) + list(motion_pair) + [shape]):
            dst.copy_(src, non_blocking=True)
        for dst in spec:   # fresh SpecAugment masks, drawn on the host exactly as the reference does
            dst.copy_(self._draw_spec_mask(dst.shape), non_blocking=True)
        for k, lst in sd.items():
            for dst, src in zip(lst, draws[k]):
                if dst is not None:
                    dst.copy_(torch.as_tensor(src, device=self.device), non_blocking=True)
        flags.copy_(self._flag_table[int(cross[0]) * 2 + int(cross[1])], non_blocking=True)
        stepping = self._stepping
        for gseg, buckets in g:
            if gseg is not None:     # a stretch in which nothing was recorded (e.g. after the last gradient write) has no graph
                gseg.replay()
            if stepping:
                for b in buckets:        # final for this backward: reduce it under the segments that follow
                    self.reducer._launch(b)
        return {k: v.clone() for k, v in out.items()}

    def _draw_spec_mask(self, shape):
        from .utils.wav2vec2 import compute_mask_indices, compute_mask_indices_hf
        c = self.model.audio_encoder.config
        fn = compute_mask_indices_hf if self.model.audio_encoder.model_type == "hubert" else compute_mask_indices
        m = fn(tuple(shape), c.mask_time_prob, c.mask_time_length, c.mask_time_min_masks, self.rng)
        return torch.from_numpy(m).pin_memory() if self.device.type == "cuda" else torch.from_numpy(m)

    def capture_all(self, batch):
        """Capture the four truncation variants for this batch shape up front (otherwise each is captured on its
        first occurrence, inside whatever is being timed)."""
        B = batch[0][0].shape[0]
        ag.DIRECT_GRAD = self.direct_grad
        for t0 in (False, True):
            for t1 in (False, True):
                key = (B, t0, t1, ())
                if key not in self._graphs:
                    self._capture(key, batch, {}, [t0, t1])

    def _capture(self, key, batch, draws, trunc):
        from . import ops
        with ops.capture_guard():      # no cyclic garbage collection while a stream is capturing
            return self._capture_guarded(key, batch, draws, trunc)

    def _capture_guarded(self, key, batch, draws, trunc):
        audio_pair, motion_pair, shape = batch
        dev = self.device
        sb = ([a.clone() for a in audio_pair], [m.clone() for m in motion_pair], shape.clone())
        sd = {}
        for k in key[3]:
            sd[k] = [None if v is None else torch.as_tensor(v, device=dev).clone() for v in draws[k]]
        flags = torch.zeros(2, dtype=torch.bool, device=dev)
        if self._flag_table is None:
            self._flag_table = torch.tensor([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=torch.bool, device=dev)
        cross = [flags[0], flags[1]]
        # SpecAugment masks are inputs of the graph: one static (B, 2 L) buffer per encoder pass of this variant
        spec = []
        enc_cfg = self.model.audio_encoder.config
        if self.model.training and enc_cfg.apply_spec_augment and enc_cfg.mask_time_prob > 0:
            B = audio_pair[0].shape[0]
            n_pass = 2 + (1 if trunc[0] else 0)
            spec = [self._draw_spec_mask((B, 2 * self.model.n_motions)).to(dev) for _ in range(n_pass)]
        ag.TrainNoise.graph_safe = True
        saved = self.reducer.arena.clone()
        segmented = self.segment_graphs
        red = self.reducer
        # no exchange from inside a warm-up or a capture, whoever calls (step() disables the hooks' launches itself, but
        # capture_all() is also called directly): buckets are launched by the replay loop / finish() only
        was_enabled, red.enabled = red.enabled, False
        # warm-up on a side stream (allocator / lazy init), then capture; gradients written by both are discarded.
        # Segmented mode: the warm-up also RECORDS the backward's gradient-write sequence (autograd accumulations and
        # direct arena writes), from which the capture knows after which write each bucket is final.
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        mt = torch.autograd.set_multithreading_enabled(False) if segmented else contextlib.nullcontext()
        with mt:   # single-threaded backward: hooks run on THIS thread, so a capture can be ended / begun inside them
            with torch.cuda.stream(s):
                self._invalidate_caches()
                ag.TrainNoise.spec_masks = spec or None
                if segmented:
                    red.trace_begin()
                    ag.GRAD_WRITTEN = red.on_write
                self._fwd_bwd(sb, sd, trunc, cross)
                final_pos = red.trace_end() if segmented else None
            torch.cuda.current_stream().wait_stream(s)
            self._invalidate_caches()
            ag.TrainNoise.spec_masks = spec or None
            if not segmented:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._graph_pool):
                    out = self._fwd_bwd(sb, sd, trunc, cross)
                if self._graph_pool is None:
                    self._graph_pool = g.pool()
                segs = [(g, [])]
            else:
                # One hipGraph per stretch of the iteration that ends where a gradient bucket becomes final: the replay
                # loop launches that bucket's all-reduce on the side stream while the next segment (the rest of the
                # backward) runs -- the overlap the eager mode gets from autograd hooks, without ~10^4 host launches.
                segs, cur = [], [torch.cuda.CUDAGraph(), []]
                torch.cuda.synchronize()
                cs = torch.cuda.Stream(device=dev)
                cs.wait_stream(torch.cuda.current_stream())

                def begin(gr):
                    if self._graph_pool is None:
                        gr.capture_begin()
                    else:
                        gr.capture_begin(pool=self._graph_pool)

                def end(gr):
                    """capture_end; returns None instead of the graph when NOTHING was recorded in the stretch (the host
                    library says so with a warning): such a segment keeps its bucket list and is never replayed."""
                    import warnings
                    with warnings.catch_warnings(record=True) as caught:
                        warnings.simplefilter("always")
                        gr.capture_end()
                    for w in caught:
                        if "Graph is empty" not in str(w.message):
                            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
                    if self._graph_pool is None:
                        self._graph_pool = gr.pool()
                    return None if any("Graph is empty" in str(w.message) for w in caught) else gr

                def cut(bucket):
                    if getattr(red, "_last_cut_count", None) == red.write_count:
                        segs[-1][1].append(bucket)   # several buckets final at the same write: one cut, and they all
                        return                       # belong to the segment that just closed
                    cur[1].append(bucket)
                    red._last_cut_count = red.write_count
                    segs.append((end(cur[0]), cur[1]))
                    cur[0], cur[1] = torch.cuda.CUDAGraph(), []
                    begin(cur[0])
                red.final_pos, red.write_count, red._last_cut_count = final_pos, 0, None
                red.on_bucket_final = cut
                try:
                    with torch.cuda.stream(cs):
                        begin(cur[0])
                        out = self._fwd_bwd(sb, sd, trunc, cross)
                        tail = end(cur[0])
                        if tail is not None or cur[1]:      # nothing after the last cut and no bucket left: no segment
                            segs.append((tail, cur[1]))
                finally:
                    red.on_bucket_final, red.final_pos = None, None
                    ag.GRAD_WRITTEN = None
                torch.cuda.current_stream().wait_stream(cs)
        ag.TrainNoise.spec_masks = None
        red.enabled = was_enabled
        self._invalidate_caches()             # cached casts now live in the graph's pool: eager code must re-make them
        self.reducer.arena.copy_(saved)
        self.reducer.begin_backward()
        # (segments with a graph, segments that recorded nothing): asserted by tests/test_train_gpu.py
        self.graph_segments = getattr(self, "graph_segments", {})
        self.graph_segments[key] = (sum(1 for g_, _ in segs if g_ is not None), sum(1 for g_, _ in segs if g_ is None))
        ent = (sb, sd, flags, segs, out, spec)
        self._graphs[key] = ent
        return ent


# ----------------------------------------------------------------------------- command line (reference flag names)
# (flag, type | "flag", default): the reference's argparse surface (training_script.py:449-513) as data; args.json files
# written by either implementation load into either (utils.model_common.load_args_with_defaults takes this parser).
_FLAGS = [
    ("mode", str, "train"), ("exp_name", str, None), ("data_root", str, None), ("max_iter", int, 2000000),
    ("batch_size", int, 16), ("num_workers", int, 2), ("generator_model_style", str, "MSMD"),
    ("style_enc_model_style", str, "vae2"), ("training_loss_style", str, "MSMD"),
    ("dataset_type", str, "ravdess+celebv-text-medium"), ("audio_model", str, "hubert"), ("d_style", int, 256),
    ("use_indicator", "flag", False), ("use_cross_style", "flag", False), ("use_vertex_space", "flag", False),
    ("num_of_basis", int, 4), ("prob_cross_style", float, 0.5), ("l_vert", float, 1.0), ("l_vel", float, 0.5),
    ("l_smooth", float, 10.0), ("l_kl_div", float, 1e-7), ("l_head_angle", float, 1.0), ("l_head_vel", float, 0.5),
    ("l_head_smooth", float, 0.5), ("l_head_trans", float, 0.5), ("scheduler", str, "Warmup"), ("lr", float, 2e-5),
    ("warm_iter", int, 5000), ("cos_max_iter", int, 1000000), ("min_lr_ratio", float, 0.1),
    ("gradient_accumulation_steps", int, 1), ("n_motions", int, 750), ("n_prev_motions", int, 100), ("fps", int, 30),
    ("trunc_prob1", float, 0.5), ("trunc_prob2", float, 0.5), ("pad_mode", str, "zero"), ("rot_repr", str, "euler"),
    ("no_head_pose", "flag", False), ("do_ignore_shape", "flag", False), ("do_ignore_cfg", "flag", False),
    ("log_iter", int, 100), ("save_iter", int, 10000), ("val_iter", int, 10000), ("log_smooth_win", int, 50),
    ("continue_from", str, None),
]


def infinite_data_loader(data_loader):
    """reference training_script.py:28-31."""
    while True:
        for data in data_loader:
            yield data


def count_parameters(model):
    """reference training_script.py:441-443."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def train(args, model, style_enc, train_loader, val_loader, optimizer, save_dir, scheduler=None, writer=None, flame=None,
          out_abc_dir=None, start_iter=0, trainer=None, use_graph=False):
    """Drop-in for the reference's train() (training_script.py:49-243): same positional arguments, same loop --
    infinite loader, one two-window iteration per step, smoothed loss log, `writer.add_scalar` hooks, checkpoints
    `iter_%07d.pt` with {args, model, style_enc, iter} every save_iter, validation every val_iter.

    The iteration itself is `Trainer.step` (HIP forward / backward, fused Adam on the flat parameter arena, bucketed
    RCCL all-reduce).  `optimizer` and `scheduler` are accepted for call-site compatibility: the optimizer's lr (when
    given) seeds the base learning rate; Adam's moments and the args.scheduler rule (training_script.py:590-621) live
    in the Trainer, so the objects themselves are not stepped.  Loader items are the reference's
    (audio_pair, coef_pair, extra) tuples (datasets.py:230-323) or already-unpacked batches."""
    import time
    from pathlib import Path
    rank = dp.env_rank()[0]
    if trainer is None:
        if optimizer is not None:
            args.lr = float(optimizer.param_groups[0]["lr"]) if scheduler is None else float(args.lr)
        _ds = getattr(train_loader, "dataset", None)
        if len(args.dataset_type.split("+")) > 1 and hasattr(_ds, "datasets"):
            _ds = _ds.datasets[0]
        trainer = Trainer(args, model, style_enc, use_graph=use_graph, flame=flame,
                          coef_stats=getattr(_ds, "coef_stats", None))
    save_dir = Path(save_dir)
    if rank == 0:
        save_dir.mkdir(parents=True, exist_ok=True)
    model.train()
    lw = load_loss_weights(args)
    dataset = getattr(train_loader, "dataset", None)
    if len(args.dataset_type.split("+")) > 1 and hasattr(dataset, "datasets"):   # ConcatDataset (reference l.60-63)
        dataset = dataset.datasets[0]
    coef_stats = getattr(dataset, "coef_stats", None)
    data = infinite_data_loader(train_loader)
    log, t0 = [], time.time()
    for it in range(start_iter, args.max_iter + 1):
        item = next(data)
        batch = batch_from_loader(item) if isinstance(item[0], (list, tuple)) and isinstance(item[1][0], dict) else item
        out = trainer.step(batch, it=it)
        log.append(out["loss"])
        if it % args.log_iter == 0 and it != start_iter:
            # EVERY rank looks at its own smoothed loss (the only host sync, once per log interval) and the ranks agree on the
            # verdict with a one-element MAX all-reduce, so a non-finite loss stops all of them together -- a lone rank 0
            # raising would leave the others blocked in the next bucket all-reduce.  Both reduction incidents of rounds 3 / 4
            # (the host library's multi-block sums inside hipGraphs) would have surfaced at this check.
            val_t = torch.stack(log[-args.log_smooth_win:]).mean()
            bad = (~torch.isfinite(val_t)).float()
            if trainer.reducer.world > 1:
                trainer.reducer.td.all_reduce(bad, op=trainer.reducer.td.ReduceOp.MAX, group=trainer.reducer.group)
            val = val_t.item()
            if float(bad.item()) != 0.0:
                raise FloatingPointError(f"non-finite training loss at iteration {it} on some rank (this rank {rank}: {val}; "
                                         f"hipGraph mode: {trainer.use_graph})")
            log = log[-args.log_smooth_win:]
        if rank == 0 and it % args.log_iter == 0 and it != start_iter:
            if writer is not None:
                writer.add_scalar("train/loss", val, it)
                writer.add_scalar("opt/lr", trainer.current_lr(), it)
            print(f"iter {it}: loss {val:.5f}  lr {trainer.current_lr():.3e}  "
                  f"{(time.time() - t0) / max(1, it - start_iter) * 1e3:.1f} ms/it")
        if rank == 0 and ((it % args.save_iter == 0 and it not in (0, start_iter)) or it == args.max_iter):
            trainer.save_checkpoint(save_dir / f"iter_{it:07}.pt", it)
        if val_loader is not None and ((it % args.val_iter == 0 and it not in (0, start_iter)) or it == args.max_iter):
            res = test(args, lw, model, style_enc, val_loader, it, 1, "val", writer, coef_stats=coef_stats)
            if rank == 0:
                print(f"iter {it}: val loss {np.mean(res['loss']):.5f}")
    return trainer


class _ResidentLoader:
    """Loader facade over datasets.ResidentDataset: `.dataset` as the reference's DataLoader exposes it, batches of
    args.batch_size random items per iteration (index draws from a per-rank numpy stream)."""

    def __init__(self, dataset, batch_size, seed):
        self.dataset, self.batch_size, self.rng = dataset, batch_size, np.random.RandomState(seed)

    def __iter__(self):
        while True:
            yield self.dataset.batch(self.rng.randint(0, len(self.dataset), size=self.batch_size))


def build_parser():
    """argparse parser with the reference's flag names, types and defaults, plus this build's knobs."""
    import argparse
    ap = argparse.ArgumentParser(description="MSMD training on MI355X (reference-compatible flags)")
    for name, kind, default in _FLAGS:
        if kind == "flag":
            ap.add_argument("--" + name, action="store_true")
        elif name == "mode":
            ap.add_argument("--mode", type=str, default="train", choices=["train", "test"])
        elif name == "scheduler":
            ap.add_argument("--scheduler", type=str, default=default, choices=["Warmup", "WarmupThenDecay"])
        else:
            ap.add_argument("--" + name, type=kind, default=default, required=name in ("exp_name", "data_root"))
    ap.add_argument("--compute_dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--hip_graph", action="store_true", help="replay forward+backward as one hipGraph per truncation pattern")
    ap.add_argument("--exp_root", type=str, default="experiments")
    ap.add_argument("--audio_encoder_weights", type=str, default=None,
                    help="Hugging Face checkpoint directory of the pretrained audio encoder (default: the reference's hub id, looked up "
                         "locally; the run stops if none is found), or 'synthetic' for closed-form test weights")
    ap.add_argument("--hf_cache_dir", type=str, default=None, help="hub-cache root searched first (reference: /code/models/Huggingface/hub2)")
    return ap


def main(argv=None):
    """Training / test driver over a DECODED corpus: --data_root is a (chunked) pickle of
    {clip: {audio, expression_code, head_orientation}} (datasets.load_dict_in_chunks); the reference's per-dataset
    directory layouts, torchaudio decoding, tensorboard and renderer hooks are outside this path."""
    import time
    from pathlib import Path
    from .config import default_args
    from .datasets import ResidentDataset, load_dict_in_chunks
    from .model import get_diffusion_model
    from .style_encoder import get_style_encoder
    from .utils.model_common import save_args
    cli = build_parser().parse_args(argv)
    args = default_args(**{k: v for k, v in vars(cli).items() if v is not None})
    rank, local_rank, world = dp.env_rank()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dp.init("nccl", device)
    raw = {}
    for chunk in load_dict_in_chunks(cli.data_root):
        raw.update(chunk)
    names = sorted(raw)
    n_val = max(1, len(names) // 20)
    train_set = ResidentDataset(raw, names[n_val:], coef_fps=args.fps, original_fps=getattr(args, "original_fps", 30),
                                n_motions=args.n_motions, clip_len=args.n_motions, device=device, seed=1234 + rank)
    val_set = ResidentDataset(raw, names[:n_val], coef_stats=train_set.coef_stats, coef_fps=args.fps,
                              original_fps=getattr(args, "original_fps", 30), n_motions=args.n_motions,
                              clip_len=args.n_motions, device=device, random_crop=False)
    if cli.continue_from and args.audio_encoder_weights is None:
        args.audio_encoder_weights = "checkpoint"      # every encoder tensor comes from the checkpoint loaded below
    model = get_diffusion_model(args, device)
    style_enc = get_style_encoder(args, args.style_enc_model_style).to(device)
    trainer = Trainer(args, model, style_enc, use_graph=cli.hip_graph)
    exp_dir = Path(cli.exp_root) / cli.exp_name
    ckpt_dir = exp_dir / "checkpoints"
    start_iter = 0
    if cli.continue_from:
        found = sorted((Path(cli.continue_from) / "checkpoints").glob("iter_*.pt"))
        if not found:
            raise ValueError(f"No checkpoints found in {cli.continue_from}/checkpoints")
        start_iter = trainer.load_checkpoint(found[-1])
    lw = load_loss_weights(args)
    val_loader = [val_set.batch(list(range(i, min(i + 2, len(val_set))))) for i in range(0, len(val_set), 2)]
    if cli.mode == "test":
        res = test(args, lw, model, style_enc, val_loader, args.max_iter, n_rounds=5, mode="test", coef_stats=train_set.coef_stats)
        if rank == 0:
            for k, v in res.items():
                print(f"{k}: {np.mean(v):.4f}")
        return res
    if rank == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        save_args(args, exp_dir)
    return train(args, model, style_enc, _ResidentLoader(train_set, args.batch_size, 99 + rank), val_loader, None,
                 ckpt_dir, start_iter=start_iter, trainer=trainer)


if __name__ == "__main__":
    main()
