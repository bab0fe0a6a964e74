"""GPU parity of losses / coefficient glue / truncation / Adam against goldens from the reference
(utils/common.py, utils/scheduler.py, torch.optim.Adam)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from msmd_amd import synth
from msmd_amd.config import default_args

from conftest import load_golden
from helpers import maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def vals(r):
    return np.array([np.nan if v is None else float(v) for v in r], np.float64)


def test_compute_loss_no_vert_and_kl():
    from msmd_amd.utils import common as C
    g = load_golden("g5_losses")
    args = default_args()
    N = 3
    gt = dev(synth.normalish("loss/gt", (N, 100, 67)))
    prev = dev(synth.normalish("loss/prev", (N, 10, 67)))
    target = dev(synth.normalish("loss/target", (N, 110, 67)))
    end_idx = torch.tensor([100, 37, 1], device=DEV)
    for start in (True, False):
        for use_end in (False, True):
            r = C.compute_loss_no_vert(args, start, None, gt, None, target, prev, None, None,
                                       end_idx=end_idx if use_end else None)
            ref = g[f"nv_{int(start)}_{int(use_end)}"]
            got = vals(r)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), (start, use_end)
            # losses are O(1) means of squared differences of O(1) values: fp32 reduction noise
            assert np.nanmax(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) < 2e-6, (start, use_end, got, ref)
    r = C.compute_loss_no_vert(default_args(criterion="l1"), False, None, gt, None, target, prev, None, None,
                               end_idx=end_idx)
    assert np.nanmax(np.abs(vals(r) - g["nv_l1"])) < 5e-6
    d = C.compute_loss_no_vert(args, False, None, gt, None, target, prev, None, None, end_idx=end_idx, return_dict=True)
    assert set(d) == {"noise", "vel", "smooth", "head_angle", "head_vel", "head_smooth", "head_trans"}
    mu = dev(synth.normalish("loss/mu", (N, 256)))
    logvar = dev((0.3 * synth.normalish("loss/logvar", (N, 256))).astype(np.float32))
    assert abs(float(C.compute_KL_loss(mu, logvar)) - float(g["kl"])) < 1e-3 * abs(float(g["kl"])) * 1e-2 + 1e-3
    with pytest.raises(NotImplementedError):
        C.compute_loss_no_vert(default_args(criterion="huber"), True, None, gt, None, target, prev, None, None)
    with pytest.raises(ValueError):
        C.compute_loss_no_vert(default_args(target="bogus"), True, None, gt, None, target, prev, None, None)


def _flame():
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    return FLAME(cfg).to(DEV)


def test_vertex_space_loss_and_coef_glue():
    from msmd_amd.utils import common as C
    g = load_golden("g5_losses")
    fl = _flame()
    L = 12
    argsv = default_args(n_motions=L, n_prev_motions=4)
    gt54 = dev((0.5 * synth.normalish("loss/gt54", (2, L, 54))).astype(np.float32))
    prev54 = dev((0.5 * synth.normalish("loss/prev54", (2, 4, 54))).astype(np.float32))
    tgt54 = dev((0.5 * synth.normalish("loss/tgt54", (2, L + 4, 54))).astype(np.float32))
    shape = dev((0.5 * synth.normalish("loss/shape", (2, 100))).astype(np.float32))
    stats = {"exp_mean": dev(0.1 * synth.normalish("st/em", (50,))),
             "exp_std": dev(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": dev(0.05 * synth.normalish("st/pm", (6,))),
             "pose_std": dev(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": dev(np.zeros(100, np.float32)), "shape_std": dev(np.ones(100, np.float32))}
    for start in (True, False):
        r = C.compute_loss(argsv, start, shape, gt54, None, tgt54, prev54, stats, fl,
                           end_idx=torch.tensor([L, 5], device=DEV))
        ref = g[f"vert_{int(start)}"]
        got = vals(r)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), start
        # vertex terms are ~1e-4 .. 1e-3 in magnitude (FLAME-scale vertices): relative tolerance
        assert np.nanmax(np.abs(got - ref) / np.maximum(1e-6, np.abs(ref))) < 2e-3, (start, got, ref)
    cd = C.get_coef_dict(gt54, shape, stats, with_global_pose=False)
    assert maxabs(cd["exp"].cpu().numpy(), g["coef_exp"]) < 1e-6 and maxabs(cd["pose"].cpu().numpy(), g["coef_pose"]) < 1e-6
    assert maxabs(cd["shape"].cpu().numpy(), g["coef_shape"]) < 1e-6
    verts = C.coef_dict_to_vertices(cd, fl, flame_batch_size=7)
    assert verts.shape == (2, L, 5023, 3)
    assert maxabs(verts.cpu().numpy()[:, :, ::79], g["coef_verts_sub"]) < 5e-6
    mc = C.get_motion_coef({"exp": cd["exp"], "pose": cd["pose"]}, "aa", with_global_pose=False)
    assert maxabs(mc.cpu().numpy(), g["motion_coef"]) < 1e-6


def test_truncation_scheduler_adam():
    from msmd_amd.utils import common as C
    from msmd_amd.utils.scheduler import GradualWarmupScheduler
    from msmd_amd import ops
    g = load_golden("g5_losses")
    a = dev(synth.audio_clips(2, 64000, tag="trunc"))
    m = dev(synth.motion_clips(2, tag="trunc_m"))
    e = torch.tensor([3, 77], device=DEV)
    assert np.array_equal(C._truncate_audio(a, (e * 640).long(), "zero").cpu().numpy()[:, ::97], g["trunc_audio_zero"])
    assert np.array_equal(C._truncate_audio(a, (e * 640).long(), "replicate").cpu().numpy()[:, ::97], g["trunc_audio_rep"])
    cdt = C._truncate_coef_dict({"exp": m[..., :50], "pose_any": m[..., 50:]}, e, "replicate")
    assert np.array_equal(torch.cat([cdt["exp"], cdt["pose_any"]], -1).cpu().numpy(), g["trunc_motion_rep"])
    at, mt, ei = C.truncate_motion_coef_and_audio(a, m, 100)
    assert at.shape == a.shape and mt.shape == m.shape and ei.shape == (2,) and int(ei.min()) >= 1 and int(ei.max()) < 100
    k = int(ei[0])
    assert float(at[0, k * 640:].abs().max()) == 0.0 and float(mt[0, k:].abs().max()) == 0.0
    with pytest.raises(ValueError):
        C._truncate_audio(a, e, "bogus")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([{"params": [p], "lr": 2e-5}])
    sch = GradualWarmupScheduler(opt, 1, 10)
    lrs = []
    for _ in range(15):
        opt.step(); sch.step(); lrs.append(opt.param_groups[0]["lr"])
    assert np.allclose(lrs, g["warmup_lrs"], rtol=1e-12, atol=0)
    # fused Adam kernel vs torch.optim.Adam trace
    w = dev(synth.normalish("adam/w", (1000,))).clone()
    m1, v1 = torch.zeros_like(w), torch.zeros_like(w)
    for i in range(3):
        ops.adam_step_(w, dev(synth.normalish(f"adam/g{i}", (1000,))), m1, v1, 2e-3, i + 1)
    assert maxabs(w.cpu().numpy(), g["adam_w3"]) < 2e-6


def test_no_constrain_prev_losses_match_reference():
    """--no_constrain_prev (deprecated in the reference but live, utils/common.py:245, 382, 481, 561): the previous
    window's part of the prediction is replaced by ground truth and its frames are masked OUT of every term; both loss
    functions against goldens from the reference, and the differentiable training restatement against the forward one."""
    from msmd_amd.utils import common as C
    from msmd_amd import train_graph as tg
    g = load_golden("g5_losses_no_constrain_prev")
    args = default_args(no_constrain_prev=True)
    N = 3
    gt = dev(synth.normalish("loss/gt", (N, 100, 67)))
    prev = dev(synth.normalish("loss/prev", (N, 10, 67)))
    target = dev(synth.normalish("loss/target", (N, 110, 67)))
    end_idx = torch.tensor([100, 37, 1], device=DEV)
    for start in (True, False):
        for use_end in (False, True):
            e = end_idx if use_end else None
            r = C.compute_loss_no_vert(args, start, None, gt, None, target, prev, None, None, end_idx=e)
            ref, got = g[f"nv_{int(start)}_{int(use_end)}"], vals(r)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), (start, use_end)
            assert np.nanmax(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) < 2e-6, (start, use_end, got, ref)
            rt = vals(tg.loss_no_vert_train(args, start, gt, target, prev, end_idx=e))
            assert np.nanmax(np.abs(rt - ref) / np.maximum(1.0, np.abs(ref))) < 2e-6, (start, use_end, rt, ref)
    # differs from the constrained mode (the masked frames carried loss there)
    r0 = vals(C.compute_loss_no_vert(default_args(), False, None, gt, None, target, prev, None, None, end_idx=end_idx))
    assert abs(r0[0] - g["nv_0_1"][0]) > 1e-3
    fl = _flame()
    L = 12
    argsv = default_args(n_motions=L, n_prev_motions=4, no_constrain_prev=True)
    gt54 = dev((0.5 * synth.normalish("loss/gt54", (2, L, 54))).astype(np.float32))
    prev54 = dev((0.5 * synth.normalish("loss/prev54", (2, 4, 54))).astype(np.float32))
    tgt54 = dev((0.5 * synth.normalish("loss/tgt54", (2, L + 4, 54))).astype(np.float32))
    shape = dev((0.5 * synth.normalish("loss/shape", (2, 100))).astype(np.float32))
    stats = {"exp_mean": dev(0.1 * synth.normalish("st/em", (50,))),
             "exp_std": dev(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": dev(0.05 * synth.normalish("st/pm", (6,))),
             "pose_std": dev(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": dev(np.zeros(100, np.float32)), "shape_std": dev(np.ones(100, np.float32))}
    for start in (True, False):
        r = C.compute_loss(argsv, start, shape, gt54, None, tgt54, prev54, stats, fl,
                           end_idx=torch.tensor([L, 5], device=DEV))
        ref, got = g[f"vert_{int(start)}"], vals(r)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), start
        assert np.nanmax(np.abs(got - ref) / np.maximum(1e-6, np.abs(ref))) < 2e-3, (start, got, ref)


def test_vertex_space_metrics_against_numpy_flame():
    """vertex_space_metrics (HIP FLAME pass + on-device reductions) against the numpy oracle's LBS / landmarks on the
    same coefficients: fp32, relative tolerance 1e-4 on metre-scale errors; truncated sequences via end_idx."""
    from msmd_amd.utils import common as C
    from oracle import flame as ofl
    fl = _flame()
    N, L = 2, 6
    pred = (0.5 * synth.normalish("vm/pred", (N, L, 54))).astype(np.float32)
    gt = (0.5 * synth.normalish("vm/gt", (N, L, 54))).astype(np.float32)
    shape = (0.5 * synth.normalish("vm/shape", (N, 100))).astype(np.float32)
    end_idx = torch.tensor([L, 4], device=DEV)
    got = C.vertex_space_metrics(dev(pred), dev(gt), dev(shape), fl, end_idx=end_idx, flame_batch_size=5)
    orc = ofl.FlameOracle(synth.flame_asset())

    def numpy_side(m):
        cd = C.get_coef_dict(torch.from_numpy(m), torch.from_numpy(shape), None, with_global_pose=False)
        flat = {k: v.reshape(-1, v.shape[-1]).numpy() for k, v in cd.items()}
        v, _, lm = orc.forward(flat["shape"], flat["exp"], flat["pose"], return_lm2d=False, return_lm3d=True)
        return v.reshape(N, L, -1, 3), lm.reshape(N, L, -1, 3)
    (vp, lp), (vg, lg) = numpy_side(pred), numpy_side(gt)
    valid = np.arange(L)[None] < np.array([L, 4])[:, None]
    dv, dl = np.linalg.norm(vp - vg, axis=-1), np.linalg.norm(lp - lg, axis=-1)
    ref = {"mve": dv.mean(-1)[valid].mean(), "lmk3d": dl.mean(-1)[valid].mean(),
           "mouth_lmk": dl[..., 48:68].mean(-1)[valid].mean(), "mouth_max": dl[..., 48:68].max(-1)[valid].mean()}
    for k, r in ref.items():
        assert abs(float(got[k]) - r) <= 1e-4 * abs(r) + 1e-7, (k, float(got[k]), r)
    assert float(got["mouth_max"]) >= float(got["mouth_lmk"]) > 0


def _vgrad_case():
    t = dev
    L, P, N = 12, 4, 3
    gt = (0.5 * synth.normalish("vgrad/gt", (N, L, 54))).astype(np.float32)
    prev = (0.5 * synth.normalish("vgrad/prev", (N, P, 54))).astype(np.float32)
    tgt = (0.5 * synth.normalish("vgrad/tgt", (N, L + P, 54))).astype(np.float32)
    shape = (0.5 * synth.normalish("vgrad/shape", (N, 100))).astype(np.float32)
    stats = {"exp_mean": t(0.1 * synth.normalish("st/em", (50,))), "exp_std": t(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": t(0.05 * synth.normalish("st/pm", (6,))), "pose_std": t(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": t(np.zeros(100, np.float32)), "shape_std": t(np.ones(100, np.float32))}
    return L, P, N, gt, prev, tgt, shape, stats


def _flame():
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    return FLAME(cfg).to(DEV)


def test_flame_differentiable_pass_matches_reference_autograd():
    """d <verts, probe> / d (exp, pose) through the differentiable FLAME pass -- per-frame kinematics by autograd on
    (B, 5, 3, 3)-sized tensors, per-vertex skinning forward (msmd_lbs_skin_v2_train) and backward (msmd_lbs_skin_bwd +
    one GEMM) in HIP -- against gradients recorded from the reference's autograd through utils/flame.py / utils/lbs.py
    (g8_vertex_grad): 2e-4 of the largest entry; the forward equals the inference kernel's vertices."""
    from helpers import flame_inputs
    g = load_golden("g8_vertex_grad")
    fl = _flame()
    x = flame_inputs(6, tag="vgrad_flame")
    ex, po = dev(x["exp"]).requires_grad_(True), dev(x["pose"]).requires_grad_(True)
    with torch.enable_grad():
        v, _, _ = fl(dev(x["shape"]), ex, po, return_lm2d=False, return_lm3d=False)
        (v * dev(synth.normalish("vgrad/probe", (6, 5023, 3)))).sum().backward()
    v0, _, _ = fl(dev(x["shape"]), dev(x["exp"]), dev(x["pose"]), return_lm2d=False, return_lm3d=False)
    assert maxabs(v.detach().cpu().numpy(), v0.cpu().numpy()) <= 2e-6
    for got, key in ((ex.grad, "flame_dexp"), (po.grad, "flame_dpose")):
        err, scale = maxabs(got.cpu().numpy(), g[key]), np.abs(g[key]).max()
        assert err <= 2e-4 * scale, (key, err, scale)


def test_vertex_space_training_loss_gradients_match_reference_autograd():
    """train_graph.loss_vert_train (the use_vertex_space training branch, reference training_script.py:167-176 ->
    utils/common.py:456-620): all eight terms and d(weighted total) / d(target) against the reference's autograd
    (g8_vertex_grad), both windows, with and without truncation."""
    from msmd_amd import train_graph as tg
    g = load_golden("g8_vertex_grad")
    L, P, N, gt, prev, tgt, shape, stats = _vgrad_case()
    args = default_args(n_motions=L, n_prev_motions=P, use_vertex_space=True, dataset_type="flame_mead_ravdess")
    fl = _flame()
    keys = ("noise", "vert", "vel", "smooth", "head_angle", "head_vel", "head_smooth", "head_trans")
    w = dict(zip(keys, g["weights"]))
    end_idx = torch.tensor([L, 5, 9], device=DEV)
    for start in (True, False):
        for use_end in (False, True):
            tgd = dev(tgt).clone().requires_grad_(True)
            with torch.enable_grad():
                ld = tg.loss_vert_train(args, start, dev(shape), dev(gt), tgd, dev(prev), stats, fl,
                                        end_idx if use_end else None)
                total = sum(w[k] * v for k, v in ld.items() if v is not None and not isinstance(v, int))
                total.backward()
            key = f"{int(start)}_{int(use_end)}"
            want = g["loss_" + key]
            got = np.array([np.nan if ld[k] is None else float(ld[k]) for k in keys])
            ok = ~np.isnan(want)
            assert np.allclose(got[ok], want[ok], rtol=2e-4, atol=1e-8), (key, got, want)
            err, scale = maxabs(tgd.grad.cpu().numpy(), g["grad_" + key]), np.abs(g["grad_" + key]).max()
            assert err <= 2e-4 * scale, (key, err, scale)


def test_training_losses_when_every_sample_is_truncated_to_one_frame():
    """A starting window whose samples are ALL truncated at end_idx == 1 has no valid row for the velocity / smoothness
    differences: the reference maps the empty selections to None -> 0 (utils/common.py:389-397), so the accumulated
    training loss stays finite.  The HIP loss kernels must give 0 for those terms (not NaN), the plain term its usual
    masked mean, and the backward a finite gradient -- compared with the same function's torch-op branch on CPU."""
    from msmd_amd import train_graph as tg
    args = default_args()
    N = 4
    gt = synth.normalish("loss1/gt", (N, 100, 67))
    target = synth.normalish("loss1/target", (N, 110, 67))
    end_idx = np.ones(N, np.int64)
    res = {}
    for where in ("cpu", DEV):
        t = torch.from_numpy(target).to(where).requires_grad_(True)
        tup = tg.loss_no_vert_train(args, True, torch.from_numpy(gt).to(where), t, None, torch.from_numpy(end_idx).to(where))
        total = sum(v for v in tup if v is not None and torch.is_tensor(v))
        total.backward()
        res[where] = ([None if v is None else float(v) for v in tup], t.grad.detach().cpu().numpy())
    (lc, gc), (lg, gg) = res["cpu"], res[DEV]
    assert all(v is None or np.isfinite(v) for v in lg), lg
    assert lg[1] == 0.0 and lg[2] == 0.0, lg                   # vel, smooth: empty -> 0
    for a, b in zip(lc, lg):
        assert (a is None) == (b is None) and (a is None or abs(a - b) < 2e-6 * max(1.0, abs(a))), (lc, lg)
    assert np.isfinite(gg).all() and maxabs(gg, gc) < 1e-6
