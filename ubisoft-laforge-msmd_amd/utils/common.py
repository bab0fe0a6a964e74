"""Coefficient glue, losses and truncation augmentation on the MI355X
(drop-in surface of reference utils/common.py:118-196, 198-620, 769-832).

Losses are evaluated by one generic masked-reduction HIP kernel (csrc/losses.hip); they return 0-dim fp32
tensors (or None where the reference returns None).  These are the FORWARD values (validation `test()` path,
reference training_script.py:245-403); the autograd path is the next build row.
"""
from __future__ import annotations

from functools import reduce

import torch

from .. import ops


# ----------------------------------------------------------------------------- script plumbing (host only)
class NullableArgs:
    """Namespace view for checkpoints saved by older configs (reference utils/common.py:9-27): absent attributes
    read as None, except three renamed switches that are derived from their predecessors."""

    _DERIVED = {
        "align_mask_width": lambda d: (1 if d["use_alignment_mask"] else 0) if "use_alignment_mask" in d else 0,
        "no_head_pose": lambda d: not d.get("predict_head_pose"),
        "no_use_learnable_pe": lambda d: not d.get("use_learnable_pe"),
    }

    def __init__(self, namespace):
        self.__dict__.update(vars(namespace))

    def __getattr__(self, key):            # reached only when normal lookup fails
        rule = NullableArgs._DERIVED.get(key)
        return rule(self.__dict__) if rule is not None else None


def count_parameters(model):
    """Trainable parameter count (reference utils/common.py:94-95)."""
    return sum(int(p.numel()) for p in model.parameters() if p.requires_grad)


def get_option_text(args, parser):
    """One line per option, sorted, with the parser default noted where the value differs
    (reference utils/common.py:98-106; same column layout)."""
    lines = []
    for key in sorted(vars(args)):
        value, default = getattr(args, key), parser.get_default(key)
        note = f"\t[default: {default}]" if value != default else ""
        lines.append(f"{key:>30}: {str(value):<30}{note}\n")
    return "".join(lines)


def get_model_path(exp_name, iteration, model_type="DPT"):
    """(checkpoint path, experiment dir relative to the experiments root) for experiments/<model_type>/<exp_name>*
    (reference utils/common.py:109-115): an exact directory match wins, else the first one with that prefix."""
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent / "experiments" / model_type
    exp_dir = root / exp_name
    if not exp_dir.exists():
        exp_dir = next(root.glob(f"{exp_name}*"))
    return exp_dir / "checkpoints" / f"iter_{iteration:07}.pt", exp_dir.relative_to(root)


# ----------------------------------------------------------------------------- coefficient glue (host views)
def get_pose_input(coef_dict, rot_repr, with_global_pose):
    """reference utils/common.py:118-125."""
    if rot_repr == "aa":
        pose_input = coef_dict["pose"] if with_global_pose else coef_dict["pose"][..., -3:]
        return pose_input[..., :-2]
    raise ValueError(f"Unknown rotation representation: {rot_repr}")


def get_motion_coef(coef_dict, rot_repr, with_global_pose=False, norm_stats=None):
    """reference utils/common.py:128-137."""
    if norm_stats is not None:
        if rot_repr != "aa":
            raise ValueError(f"Unknown rotation representation {rot_repr}!")
        coef_dict = {k: (coef_dict[k] - norm_stats[f"{k}_mean"]) / norm_stats[f"{k}_std"] for k in ("exp", "pose")}
    return torch.cat([coef_dict["exp"], get_pose_input(coef_dict, rot_repr, with_global_pose)], dim=-1)


def get_coef_dict(motion_coef, shape_coef=None, denorm_stats=None, with_global_pose=False, rot_repr="aa"):
    """reference utils/common.py:140-173."""
    coef_dict = {"exp": motion_coef[..., :50]}
    if rot_repr == "aa":
        if with_global_pose:
            coef_dict["pose"] = motion_coef[..., 50:]
        else:
            placeholder = torch.zeros_like(motion_coef[..., :3])
            coef_dict["pose"] = torch.cat([placeholder, motion_coef[..., -1:]], dim=-1)
        coef_dict["pose"] = torch.cat([coef_dict["pose"], torch.zeros_like(motion_coef[..., :2])], dim=-1)
    else:
        raise ValueError(f"Unknown rotation representation {rot_repr}!")
    if shape_coef is not None:
        if motion_coef.ndim == 3:
            if shape_coef.ndim == 2:
                shape_coef = shape_coef.unsqueeze(1)
            if shape_coef.shape[1] == 1:
                shape_coef = shape_coef.expand(-1, motion_coef.shape[1], -1)
        coef_dict["shape"] = shape_coef
    if denorm_stats is not None:
        coef_dict = {k: coef_dict[k] * denorm_stats[f"{k}_std"] + denorm_stats[f"{k}_mean"] for k in coef_dict}
    if not with_global_pose:
        if rot_repr == "aa":
            coef_dict["pose"][..., :3] = 0
        else:
            raise ValueError(f"Unknown rotation representation {rot_repr}!")
    return coef_dict


def coef_dict_to_vertices(coef_dict, flame, rot_repr="aa", ignore_global_rot=False, flame_batch_size=512):
    """reference utils/common.py:176-196.  The fused LBS kernel has no materialised (B, V, 4, 4) intermediates, so
    chunking is kept only for signature parity (one launch per chunk)."""
    shape = coef_dict["exp"].shape[:-1]
    coef_dict = {k: v.reshape(-1, v.shape[-1]) for k, v in coef_dict.items()}
    n_samples = reduce(lambda x, y: x * y, shape, 1)
    vert_list = []
    for i in range(0, n_samples, flame_batch_size):
        b = {k: v[i:i + flame_batch_size] for k, v in coef_dict.items()}
        if rot_repr != "aa":
            raise ValueError(f"Unknown rot_repr: {rot_repr}")
        vert, _, _ = flame(b["shape"], b["exp"], b["pose"], pose2rot=True, ignore_global_rot=ignore_global_rot,
                           return_lm2d=False, return_lm3d=False)
        vert_list.append(vert)
    return torch.cat(vert_list, dim=0).view(*shape, -1, 3)


# ----------------------------------------------------------------------------- losses
def _crit(args):
    c = args.criterion.lower()
    if c == "l2":
        return 0
    if c == "l1":
        return 1
    raise NotImplementedError(f"Criterion {args.criterion} not implemented.")


def _none_if_nan_free(x):
    return x


def compute_KL_loss(mu, logvar):
    """reference utils/common.py:443-454 (a SUM over the batch, not a mean)."""
    return ops.kl_loss(mu.float().contiguous(), logvar.float().contiguous())


def _prep(args, is_starting_sample, motion_coef_gt, target, prev_motion_coef):
    """Window bookkeeping shared by both loss functions (reference utils/common.py:240-252, 476-484)."""
    if is_starting_sample:
        target = target[:, args.n_prev_motions:]
        prefix = 0
    else:
        motion_coef_gt = torch.cat([prev_motion_coef, motion_coef_gt], dim=1)
        if args.no_constrain_prev:
            target = torch.cat([prev_motion_coef, target[:, args.n_prev_motions:]], dim=1)
        prefix = args.n_prev_motions
    return motion_coef_gt.float().contiguous(), target.float().contiguous(), prefix


def compute_loss_no_vert(args, is_starting_sample, shape_coef, motion_coef_gt, noise, target, prev_motion_coef,
                         coef_stats, flame, end_idx=None, return_dict=False):
    """reference utils/common.py:198-442: parameter-space noise / velocity / smoothness / head-pose losses."""
    crit = _crit(args)
    loss_vel = loss_smooth = loss_head_angle = loss_head_vel = loss_head_smooth = loss_head_trans = None
    e32 = end_idx.to(torch.int32).contiguous() if end_idx is not None else None
    if args.target == "noise":
        gt = noise.float().contiguous()
        pr = target[:, args.n_prev_motions:].float().contiguous()
        loss_noise = ops.masked_seq_loss(gt, pr, e32, 0, gt.shape[-1], 0, 0, crit)
        if not args.no_head_pose:
            raise UnboundLocalError("loss_head_angle")  # reference utils/common.py:401 fails the same way
    elif args.target == "sample":
        gt, pr, prefix = _prep(args, is_starting_sample, motion_coef_gt, target, prev_motion_coef)
        C = gt.shape[-1]
        if args.no_constrain_prev and not is_starting_sample:
            prefix = -prefix   # previous-window frames are masked OUT (reference utils/common.py:382-385)
        ms = lambda c_lo, c_hi, order, mode=0: ops.masked_seq_loss(gt, pr, e32, c_lo, c_hi, order, prefix, crit, mode)
        loss_noise = ms(0, C, 0)
        if args.l_vel > 0:
            loss_vel = ms(0, C - 3, 1) + ms(C - 3, C, 1)
        if args.l_smooth > 0:
            loss_smooth = ms(0, C - 3, 2, 1) + ms(C - 3, C, 2, 1)
        if not args.no_head_pose:
            if args.rot_repr != "aa":
                raise ValueError(f"Unknown rotation representation {args.rot_repr}!")
            loss_head_angle = ms(C - 3, C, 0)
            if args.l_head_vel > 0:
                loss_head_vel = ms(C - 3, C, 1)
            if args.l_head_smooth > 0:
                loss_head_smooth = ms(C - 3, C, 2, 1)
            if not is_starting_sample and args.l_head_trans > 0:
                loss_head_trans = _head_trans(args, gt[:, :, -3:], pr[:, :, -3:], crit)
    else:
        raise ValueError(f"Unknown diffusion target: {args.target}")
    half = lambda v: None if v is None else v / 2
    if not return_dict:
        if loss_vel is None or loss_smooth is None or loss_head_angle is None or loss_head_vel is None \
                or loss_head_smooth is None:
            raise TypeError("unsupported operand type(s) for /: 'NoneType' and 'int'")  # reference l.419
        return (loss_noise / 2, loss_vel / 2, loss_smooth / 2, loss_head_angle / 2, loss_head_vel / 2,
                loss_head_smooth / 2, loss_head_trans)
    z = lambda v: 0 if v is None else v
    return {"noise": z(loss_noise) / 2, "vel": z(loss_vel) / 2, "smooth": z(loss_smooth) / 2,
            "head_angle": z(loss_head_angle) / 2, "head_vel": z(loss_head_vel) / 2,
            "head_smooth": z(loss_head_smooth) / 2, "head_trans": loss_head_trans}


def _head_trans(args, head_gt, head_pred, crit, end_idx=None):
    """Transition term (reference utils/common.py:344-371, 535-545): 3 gt frames before the window + 3 predicted
    frames; constrain consecutive velocities and accelerations across the seam.  (N, 6, 3) tensors: a dozen
    scalars per sample, evaluated with the generic kernel on the spliced sequence.  The vertex-space variant
    masks the 2 / 3 seam frames with the truncation mask (end_idx); the parameter-space one does not."""
    n = args.n_prev_motions
    seq = torch.cat([head_gt[:, n - 3:n], head_pred[:, n:n + 3]], dim=1).contiguous()  # (N, 6, 3)
    vel = (seq[:, 1:] - seq[:, :-1]).contiguous()     # (N, 5, 3)
    acc = (vel[:, 1:] - vel[:, :-1]).contiguous()     # (N, 4, 3)
    lv = ops.masked_seq_loss(vel[:, 2:4].contiguous(), vel[:, 1:3].contiguous(), end_idx, 0, 3, 0, 0, crit)
    la = ops.masked_seq_loss(acc[:, 1:].contiguous(), acc[:, :-1].contiguous(), end_idx, 0, 3, 0, 0, crit)
    return lv + la


def compute_loss(args, is_starting_sample, shape_coef, motion_coef_gt, noise, target, prev_motion_coef, coef_stats,
                 flame, end_idx=None, return_dict=False):
    """reference utils/common.py:456-620 (vertex-space variant through FLAME; legacy 54-d motion)."""
    crit = _crit(args)
    e32 = end_idx.to(torch.int32).contiguous() if end_idx is not None else None
    loss_vert = loss_vel = loss_smooth = loss_head_angle = loss_head_vel = loss_head_smooth = loss_head_trans = None
    if args.target == "noise":
        gt = noise.float().contiguous()
        pr = target[:, args.n_prev_motions:].float().contiguous()
        loss_noise = ops.masked_seq_loss(gt, pr, e32, 0, gt.shape[-1], 0, 0, crit)
    elif args.target == "sample":
        gt, pr, prefix = _prep(args, is_starting_sample, motion_coef_gt, target, prev_motion_coef)
        if args.no_constrain_prev and not is_starting_sample:
            prefix = -prefix   # reference utils/common.py:561-563
        C = gt.shape[-1]
        loss_noise = ops.masked_seq_loss(gt, pr, e32, 0, C, 0, prefix, crit)
        if args.l_vert > 0 or args.l_vel > 0:
            seq_len = pr.shape[1]
            cg = get_coef_dict(gt, shape_coef, coef_stats, with_global_pose=False, rot_repr=args.rot_repr)
            cp = get_coef_dict(pr, shape_coef, coef_stats, with_global_pose=False, rot_repr=args.rot_repr)
            f = lambda c: flame(c["shape"].reshape(-1, 100), c["exp"].reshape(-1, 50), c["pose"].reshape(-1, 6),
                                return_lm2d=False, return_lm3d=False)[0].view(-1, seq_len, 5023 * 3)
            vg, vp = f(cg), f(cp)
            mv = lambda order, mode=0: ops.masked_seq_loss(vg, vp, e32, 0, 5023 * 3, order, prefix, crit, mode)
            if args.l_vert > 0:
                loss_vert = mv(0)
            if args.l_vel > 0:
                loss_vel = mv(1)
            if args.l_smooth > 0:
                loss_smooth = mv(2, 1)
        if not args.no_head_pose:
            if args.rot_repr != "aa":
                raise ValueError(f"Unknown rotation representation {args.rot_repr}!")
            mh = lambda order, mode=0: ops.masked_seq_loss(gt, pr, e32, 50, 53, order, prefix, crit, mode)
            if args.l_head_angle > 0:
                loss_head_angle = mh(0)
            if args.l_head_vel > 0:
                loss_head_vel = mh(1)
            if args.l_head_smooth > 0:
                loss_head_smooth = mh(2, 1)
            if not is_starting_sample and args.l_head_trans > 0:
                loss_head_trans = _head_trans(args, gt[:, :, 50:53], pr[:, :, 50:53], crit, e32)
    else:
        raise ValueError(f"Unknown diffusion target: {args.target}")
    z = lambda v: 0 if v is None else v
    vals = [z(v) / 2 for v in (loss_noise, loss_vert, loss_vel, loss_smooth, loss_head_angle, loss_head_vel,
                               loss_head_smooth)]
    if return_dict:
        return dict(noise=vals[0], vert=vals[1], vel=vals[2], smooth=vals[3], head_angle=vals[4], head_vel=vals[5],
                    head_smooth=vals[6], head_trans=loss_head_trans)
    return (*vals, loss_head_trans)


# ----------------------------------------------------------------------------- truncation augmentation
def _truncate_audio(audio, end_idx, pad_mode="zero"):
    """reference utils/common.py:769-782 (end_idx in samples)."""
    if pad_mode not in ("zero", "replicate"):
        raise ValueError(f"Unknown pad mode {pad_mode}!")
    out = audio.float().clone().contiguous()
    return ops.truncate_rows_(out, end_idx.to(torch.int32).contiguous(), 1, pad_mode == "replicate")


def _truncate_coef_dict(coef_dict, end_idx, pad_mode="zero"):
    """reference utils/common.py:785-798."""
    if pad_mode not in ("zero", "replicate"):
        raise ValueError(f"Unknown pad mode: {pad_mode}!")
    e = end_idx.to(torch.int32).contiguous()
    return {k: ops.truncate_rows_(v.float().clone().contiguous(), e, 1, pad_mode == "replicate")
            for k, v in coef_dict.items()}


def truncate_motion_coef_and_audio(audio, motion_coef, n_motions, audio_unit=640, pad_mode="zero",
                                   expression_code_size=50):
    """reference utils/common.py:816-832: one random end index per sample (host RNG as the reference)."""
    batch_size = audio.shape[0]
    end_idx = torch.randint(1, n_motions, (batch_size,), device=audio.device)
    audio_trunc = _truncate_audio(audio, (end_idx * audio_unit).long(), pad_mode=pad_mode)
    motion_trunc = _truncate_coef_dict({"m": motion_coef}, end_idx, pad_mode=pad_mode)["m"]
    return audio_trunc, motion_trunc, end_idx


def truncate_coef_dict_and_audio(audio, coef_dict, n_motions, audio_unit=640, pad_mode="zero"):
    """reference utils/common.py:801-813."""
    batch_size = audio.shape[0]
    end_idx = torch.randint(1, n_motions, (batch_size,), device=audio.device)
    audio_trunc = _truncate_audio(audio, (end_idx * audio_unit).long(), pad_mode=pad_mode)
    return audio_trunc, _truncate_coef_dict(coef_dict, end_idx, pad_mode=pad_mode), end_idx


# ----------------------------------------------------------------------------- vertex-space evaluation
@torch.no_grad()
def vertex_space_metrics(motion_pred, motion_gt, shape_coef, flame, coef_stats=None, rot_repr="aa", end_idx=None,
                         flame_batch_size=512):
    """Evaluation of predicted against ground-truth motion in FLAME vertex space, on the device the sampler left its
    output on (no host round trip): coefficients -> `get_coef_dict` -> one FLAME pass per chunk (the HIP LBS + landmark
    kernels) for both sequences -> per-frame Euclidean errors.  motion_*: (N, L, 54) legacy layout (50 expression + 4
    pose, as `compute_loss` consumes, utils/common.py:486-489); shape_coef (N, 100); end_idx (N,) optional valid length.

    Returns 0-dim fp32 tensors (metres, the FLAME asset's unit):
      mve        mean over valid frames and vertices of |v_pred - v_gt|
      lmk3d      the same over the 68 3-D landmarks
      mouth_lmk  over landmarks 48..67 (outer + inner lip contours)
      mouth_max  mean over valid frames of the per-frame maximum mouth-landmark error (a lip-sync worst case per frame)
    The reference stops at `coef_dict_to_vertices` (utils/common.py:176-196); these four reductions are this build's
    extension of that export step."""
    N, L = motion_pred.shape[:2]
    verts, lmks = [], []
    for m in (motion_pred, motion_gt):
        cd = get_coef_dict(m.float(), shape_coef, coef_stats, with_global_pose=False, rot_repr=rot_repr)
        flat = {k: v.reshape(-1, v.shape[-1]).contiguous() for k, v in cd.items()}
        vs, ls = [], []
        for i in range(0, N * L, flame_batch_size):
            v, _, l3 = flame(flat["shape"][i:i + flame_batch_size], flat["exp"][i:i + flame_batch_size],
                             flat["pose"][i:i + flame_batch_size], return_lm2d=False, return_lm3d=True)
            vs.append(v)
            ls.append(l3)
        verts.append(torch.cat(vs).view(N, L, -1, 3))
        lmks.append(torch.cat(ls).view(N, L, -1, 3))
    valid = torch.ones(N, L, dtype=torch.bool, device=motion_pred.device) if end_idx is None else \
        torch.arange(L, device=motion_pred.device).expand(N, -1) < end_idx.to(motion_pred.device).unsqueeze(1)
    w = valid.float()
    cnt = w.sum().clamp(min=1.0)
    dv = (verts[0] - verts[1]).norm(dim=-1)          # (N, L, V)
    dl = (lmks[0] - lmks[1]).norm(dim=-1)            # (N, L, 68)
    mouth = dl[..., 48:68]
    per_frame = lambda x: (x * w).sum() / cnt
    return {"mve": per_frame(dv.mean(-1)), "lmk3d": per_frame(dl.mean(-1)), "mouth_lmk": per_frame(mouth.mean(-1)),
            "mouth_max": per_frame(mouth.max(-1).values)}
