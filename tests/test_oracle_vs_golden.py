"""Pin the numpy oracle against vectors produced by the imported reference
(tests/golden/make_goldens.py).  Index maths bit-exact; fp32 tolerances stated per test."""
import math

import numpy as np
import pytest

from oracle import audio_encoder as oa, diffusion as od, flame as ofl, infer as oi, nn as onn, rotations as orot
from oracle import style as ost
from msmd_amd import synth
from msmd_amd.config import default_args

from conftest import load_golden
from helpers import denoiser_inputs, flame_inputs, maxabs, msmd_state_dict, style_state_dict


# ----------------------------------------------------------------------------- G1 index maths (bit-exact)
def test_pad_audio_index_bit_exact():
    g = load_golden("g1_index")
    for L in (31999, 32000, 32001, 32081, 32320, 64000, 64001, 64079, 64080, 64081, 160000):
        idx = oa.pad_audio_gather_index(L)
        assert idx.shape[0] == int(g[f"pad_len_{L}"]), L
        assert np.array_equal(idx[:48], g[f"pad_head_{L}"]), L
        assert np.array_equal(idx[-48:], g[f"pad_tail_{L}"]), L


def test_conv_length_chain():
    g = load_golden("g1_index")
    for S, T in ((32080, 100), (64080, 200), (160080, 500)):
        assert oa.conv_out_lengths(S)[-1] == int(g[f"conv_T_{S}"]) == T
    assert oa.conv_out_lengths(64080) == [12815, 6407, 3203, 1601, 800, 400, 200]


def test_crop_and_interp_tables_bit_exact():
    g = load_golden("g1_index")
    for fps, frame_num, T50 in ((25, 200, 200), (30, 200, 400), (25, 500, 500), (25, 100, 100), (30, 120, 250)):
        crop = oa.crop_len(frame_num, fps)
        assert crop == int(g[f"crop_{fps}_{frame_num}"])
        ramp = np.arange(T50, dtype=np.float32)[None, :crop, None]
        y = onn.interp_linear_cl(ramp, frame_num)[0, :, 0]
        assert np.array_equal(y, g[f"interp_{fps}_{frame_num}_{T50}"]), (fps, frame_num)
    y = onn.interp_linear_cl(np.arange(200, dtype=np.float32)[None, :, None], 100)[0, :, 0]
    assert np.array_equal(y, g["interp_200_to_100"])
    # 200 -> 100 is exactly the pairwise mean
    i0, i1, w1 = onn.interp_linear_table(200, 100)
    assert np.array_equal(i0, np.arange(100) * 2) and np.array_equal(i1, i0 + 1) and np.all(w1 == 0.5)


def test_infer_window_plan():
    g = load_golden("g1_index")
    for S in (32000, 64000, 100000, 200001, 64001, 63999):
        p = oi.window_plan(S)
        assert [p["clip_len"], p["n_subdivision"], p["n_padding_audio_samples"], p["n_padding_frames"]] == \
            g[f"plan_{S}"].tolist()


# ----------------------------------------------------------------------------- G2 schedule
def test_diffusion_schedule():
    g = load_golden("g2_schedule")
    for T in (5, 500):
        for mode in ("linear", "quadratic", "sigmoid", "cosine"):
            s = od.diffusion_schedule(T, mode)
            for k, v in s.items():
                ref = g[f"{mode}_{T}_{k}"]
                assert v.shape == ref.shape
                # fp32 libm (cos/exp/log) ulp differences between torch and numpy, amplified by the
                # 1 - ab[i]/ab[i-1] cancellation: <= 1e-5 absolute on values in [0, 1]
                assert maxabs(v, ref) <= 1e-5, (mode, T, k, maxabs(v, ref))


def test_masks_and_pe():
    g = load_golden("g2_schedule")
    assert np.array_equal(od.enc_dec_mask(110, 110, 1, 0), g["enc_dec_mask_110_1_0"])
    assert np.array_equal(od.enc_dec_mask(110, 110, 1, 1), g["enc_dec_mask_110_1_1"])
    m = od.alignment_mask(10, 100, 1)
    assert m.shape == (111, 110) and not m[0].any()
    assert all(np.flatnonzero(~m[i]).tolist() == [i - 1] for i in range(1, 111))
    pe = onn.sinusoid_table(501, 512)
    assert maxabs(pe[0, [0, 1, 7, 250, 500]], g["pe_512_501_rows"]) <= 1e-4  # 1-ulp exp() differences x position 500


# ----------------------------------------------------------------------------- G3 blocks
@pytest.mark.parametrize("am", ["wav2vec2", "hubert"])
def test_audio_encoder(am):
    g = load_golden(f"g3_audio_{am}")
    sd, args = msmd_state_dict(am)
    audio = synth.audio_clips(2, 64000)
    h, st = oa.audio_encoder(sd, "audio_encoder.", oa.pad_audio(audio), 25, frame_num=200, return_stages=True)
    assert maxabs(st["conv"][:, ::3, ::5], g["conv"]) <= 2e-5
    assert maxabs(st["proj"][:, ::3, ::5], g["proj"]) <= 5e-5
    assert maxabs(st["layer0"][:, ::3, ::5], g["layer0"]) <= 1e-4
    assert maxabs(st["layer11"][:, ::3, ::5], g["layer11"]) <= 1e-4
    feat768 = onn.interp_linear_cl(h, 100)
    assert maxabs(feat768[:, ::2, ::3], g["feat768"]) <= 1e-4
    feat = onn.linear(feat768, sd["audio_feature_map.weight"], sd["audio_feature_map.bias"])
    assert maxabs(feat, g["feat"]) <= 1e-4
    # 30 fps crop/interp path, 2 s clip
    a2 = synth.audio_clips(1, 32000, tag="audio30")
    y = oa.audio_encoder(sd, "audio_encoder.", oa.pad_audio(a2), 30, frame_num=60)
    assert maxabs(y, g["hidden_fps30_60"]) <= 1e-4


def test_conv0_groupnorm_slice():
    """conv0 + GroupNorm(512,512) + GELU slice (the hook captured the whole first conv layer)."""
    g = load_golden("g3_audio_wav2vec2")
    sd, _ = msmd_state_dict("wav2vec2")
    audio = oa.pad_audio(synth.audio_clips(2, 64000))
    p = "audio_encoder.feature_extractor.conv_layers.0."
    x = onn.conv1d_cl(audio[:, :, None], sd[p + "conv.weight"], None, 5)
    assert x.shape == (2, 12815, 512)
    mean = x.mean(axis=1, keepdims=True, dtype=np.float64)
    var = x.var(axis=1, keepdims=True, dtype=np.float64)
    y = onn.gelu(((x - mean) / np.sqrt(var + 1e-5)).astype(np.float32) * sd[p + "layer_norm.weight"]
                 + sd[p + "layer_norm.bias"])
    assert maxabs(y[:, ::61, ::7], g["conv0"]) <= 1e-5


def test_state_dict_keys_match_reference():
    g = load_golden("g0_keys")
    sd, _ = msmd_state_dict("wav2vec2")
    ref_keys = {synth.canonical_name(str(k)) for k in g["keys"]}
    learnable = {k for k in ref_keys if not synth.is_computed_buffer(k)}
    assert learnable == set(sd.keys())
    assert int(g["n_params"]) == 130_017_905
    ssd, _ = style_state_dict()
    assert {str(k) for k in g["style_keys"] if "PE.pe" not in str(k)} == set(ssd.keys())


def test_denoiser():
    g = load_golden("g3_denoiser")
    sd, args = msmd_state_dict("wav2vec2")
    x = denoiser_inputs(2, args)
    person = np.concatenate([x["shape"][:, None], x["style"][:, None]], axis=-1)
    for width in (1, 2):
        y = od.denoising_net(sd, x["motion"], x["audio_feat"], person, x["style"][:, None], x["prev_motion"],
                             x["prev_audio"], g["step"], x["indicator"], align_mask_width=width)
        assert y.shape == (2, 110, 67)
        assert maxabs(y, g[f"target_w{width}"]) <= 5e-5, width
    dyn, stat, al = od.denoising_net(sd, x["motion"], x["audio_feat"], person, x["style"][:, None], x["prev_motion"],
                                     x["prev_audio"], g["step"], x["indicator"], keep_separate=True)
    assert maxabs(dyn, g["dynamic"]) <= 5e-5 and maxabs(al, g["alphas"]) <= 5e-5
    assert maxabs(stat[:, :2], g["static"]) <= 5e-5


def test_denoiser_options():
    """regularize_alpha='sigmoid' and the sinusoidal PE module (no_use_learnable_pe): both still read by the reference
    (model.py:13-17, 862-866, 950-953, 973-974); goldens from the reference built with each switch."""
    g = load_golden("g3_denoiser_options")
    for name, kw, okw in (("sigmoid", dict(regularize_alpha="sigmoid"), dict(regularize_alpha="sigmoid")),
                          ("sinpe", dict(no_use_learnable_pe=True), {})):
        sd, args = msmd_state_dict("wav2vec2", **kw)
        assert ("denoising_net.PE" in sd) == (name != "sinpe")
        x = denoiser_inputs(2, args)
        person = np.concatenate([x["shape"][:, None], x["style"][:, None]], axis=-1)
        a = (sd, x["motion"], x["audio_feat"], person, x["style"][:, None], x["prev_motion"], x["prev_audio"], g["step"],
             x["indicator"])
        assert maxabs(od.denoising_net(*a, **okw), g[f"target_{name}"]) <= 5e-5, name
        _, _, al = od.denoising_net(*a, keep_separate=True, **okw)
        assert maxabs(al, g[f"alphas_{name}"]) <= 5e-5, name
        if name == "sigmoid":
            assert al.min() > 0 and al.max() < 1


def test_msmd_forward():
    g = load_golden("g3_forward")
    sd, args = msmd_state_dict("wav2vec2")
    sched = od.diffusion_schedule(500, "cosine")
    x = denoiser_inputs(2, args, tag="fw")
    audio = synth.audio_clips(2, 64000, tag="fw_audio")
    _, target, afeat = od.msmd_forward(sd, sched, x["motion"], audio, x["shape"], x["style"], [3, 499], g["a_eps"],
                                       indicator=x["indicator"])
    assert maxabs(afeat[:, ::2, ::3], g["a_audio_feat"]) <= 1e-4
    assert maxabs(target, g["a_target"]) <= 1e-4
    flag = g["b_flag"]
    _, target, _ = od.msmd_forward(sd, sched, x["motion"], x["audio_feat"], x["shape"], x["style"], [250, 1],
                                   g["b_eps"], x["prev_motion"], x["prev_audio"], x["indicator"],
                                   null_style_mask=flag > 0.55, null_audio_mask=flag > 0.9)
    assert maxabs(target, g["b_target"]) <= 1e-4


def test_style_encoder():
    g = load_golden("g3_style")
    sd, _ = style_state_dict()
    for B, T in ((2, 100), (1, 60)):
        m = synth.motion_clips(B, T, tag="style_in")
        mu, logvar = ost.style_encoder_mu_logvar(sd, m)
        assert maxabs(mu, g[f"mu_{B}_{T}"]) <= 2e-5 and maxabs(logvar, g[f"logvar_{B}_{T}"]) <= 2e-5
        assert maxabs(ost.reparam(mu, logvar, g[f"eps_{B}_{T}"]), g[f"z_{B}_{T}"]) <= 5e-5


def test_sampler():
    g = load_golden("g3_sample")
    sd, args = msmd_state_dict("wav2vec2")
    x = denoiser_inputs(2, args, tag="sm")
    T = 3
    sched = od.diffusion_schedule(T, "cosine")
    xT = synth.normalish("sm/xT", (2, 100, 67))
    cases = {
        "inc": dict(cfg_mode="incremental", cfg_scale=1.15),
        "ind": dict(cfg_mode="independent", cfg_scale=[1.3, 0.9]),
        "audio_only": dict(cfg_cond=["audio"], cfg_scale=2.0),
        "nocfg": dict(cfg_cond=[]),
        "dt": dict(cfg_mode="incremental", cfg_scale=1.4, dynamic_threshold=(0.9, 0.5, 2.0)),
        "flex": dict(cfg_mode="incremental", cfg_scale=1.15, flexibility=0.5),
    }
    for name, kw in cases.items():
        z = g[f"{name}_z"]
        z_list = {T - i: z[i] for i in range(T - 1)}
        y = od.sample(sd, sched, x["audio_feat"], x["shape"], x["style"], xT, z_list, x["prev_motion"],
                      x["prev_audio"], x["indicator"], **kw)
        assert maxabs(y, g[f"{name}_x0"]) <= 1e-4, (name, maxabs(y, g[f"{name}_x0"]))
    z = g["sep_z"]
    y, dyn, cum, al = od.sample(sd, sched, x["audio_feat"][:1], x["shape"][:1], x["style"][:1], xT[:1],
                                {T - i: z[i] for i in range(T - 1)}, x["prev_motion"][:1], x["prev_audio"][:1],
                                x["indicator"][:1], cfg_scale=1.3, separate=True)
    assert maxabs(y, g["sep_x0"]) <= 1e-4 and maxabs(dyn, g["sep_dyn"]) <= 1e-4
    assert maxabs(cum, g["sep_static"]) <= 1e-4 and maxabs(al, g["sep_alpha"]) <= 1e-4
    z = g["noise_z"]
    y = od.sample(sd, od.diffusion_schedule(T, "linear"), x["audio_feat"], x["shape"], x["style"], xT,
                  {T - i: z[i] for i in range(T - 1)}, indicator=x["indicator"], target="noise")
    assert maxabs(y, g["noise_x0"]) <= 2e-4


def test_infer_coeffs():
    g = load_golden("g3_infer")
    sd, args = msmd_state_dict("wav2vec2")
    T = 2
    sched = od.diffusion_schedule(T, "cosine")
    for S in (100000, 32000):
        audio = synth.audio_clips(1, S, tag="infer")[0]
        style = synth.normalish("infer/style", (1, args.d_style))
        shape = np.zeros((1, 1, 100), np.float32)
        draws = g[f"draws_{S}"]
        n_sub = oi.window_plan(S)["n_subdivision"]
        z_lists = [{2: draws[1 + i]} for i in range(n_sub)]
        y = oi.infer_coeffs(sd, sched, audio, shape, style, draws[0], z_lists, cfg_scale=1.4)
        assert y.shape == g[f"coef_{S}"].shape
        assert maxabs(y, g[f"coef_{S}"]) <= 1e-4, S


# ----------------------------------------------------------------------------- G4 FLAME / rotations
def test_flame_lbs_and_landmarks():
    g = load_golden("g4_flame")
    fl = ofl.FlameOracle(synth.flame_asset())
    x = flame_inputs(8)
    pose = g["pose"]
    v, lm2d, lm3d = fl.forward(x["shape"], x["exp"], pose)
    assert v.shape == (8, 5023, 3)
    assert maxabs(v[:, ::79], g["verts_sub"]) <= 2e-6
    assert maxabs(v.sum(axis=1, dtype=np.float64), g["verts_sum"]) <= 2e-3  # sum of 5023 fp32 values
    assert maxabs(lm2d, g["lm2d"]) <= 2e-6 and maxabs(lm3d, g["lm3d"]) <= 2e-6
    v2, _, _ = fl.forward(x["shape"], x["exp"], pose, ignore_global_rot=True, return_lm2d=False, return_lm3d=False)
    assert maxabs(v2[:, ::79], g["verts_nog_sub"]) <= 2e-6
    assert maxabs(ofl.batch_rodrigues(g["rodrigues_in"]), g["rodrigues_out"]) <= 1e-6


def test_rotation_conversions():
    g = load_golden("g4_rotations")
    aa, q, q2, pts, d6, eul = (g[k] for k in ("aa", "q", "q2", "pts", "d6", "eul"))
    R = orot.axis_angle_to_matrix(aa)
    checks = dict(
        axis_angle_to_matrix=R, axis_angle_to_quaternion=orot.axis_angle_to_quaternion(aa),
        quaternion_to_matrix=orot.quaternion_to_matrix(q), matrix_to_quaternion=orot.matrix_to_quaternion(g["axis_angle_to_matrix"]),
        quaternion_to_axis_angle=orot.quaternion_to_axis_angle(q),
        matrix_to_axis_angle=orot.matrix_to_axis_angle(g["axis_angle_to_matrix"]),
        rotation_6d_to_matrix=orot.rotation_6d_to_matrix(d6), matrix_to_rotation_6d=orot.matrix_to_rotation_6d(g["axis_angle_to_matrix"]),
        axis_angle_to_rotation_6d=orot.axis_angle_to_rotation_6d(aa),
        quaternion_raw_multiply=orot.quaternion_raw_multiply(q, q2), quaternion_multiply=orot.quaternion_multiply(q, q2),
        quaternion_invert=orot.quaternion_invert(q), quaternion_apply=orot.quaternion_apply(q, pts),
        standardize_quaternion=orot.standardize_quaternion(q))
    for k, v in checks.items():
        # outputs up to ~6 in magnitude (angle * axis of un-normalised quaternions): a few fp32 ulps
        tol = 2e-5 if k in ("matrix_to_axis_angle", "quaternion_apply", "matrix_to_quaternion",
                            "quaternion_to_axis_angle") else 2e-6
        assert maxabs(v, g[k]) <= tol, (k, maxabs(v, g[k]))
    for conv in ("XYZ", "ZYX", "YXZ", "XYX", "ZXZ"):
        Re = orot.euler_angles_to_matrix(eul, conv)
        assert maxabs(Re, g[f"euler_angles_to_matrix_{conv}"]) <= 2e-6
        assert maxabs(orot.matrix_to_euler_angles(g[f"euler_angles_to_matrix_{conv}"], conv),
                      g[f"matrix_to_euler_angles_{conv}"]) <= 2e-5


def test_specaugment_mask_indices_match_reference_and_hf():
    """Host-side SpecAugment index routines (product code, pure numpy) against masks drawn by the reference's
    wav2vec2 wrapper and by transformers' routine (HuBERT path) from the same numpy seeds."""
    from msmd_amd.utils.wav2vec2 import compute_mask_indices, compute_mask_indices_hf
    g = load_golden("g1_specaug")
    for i, (b, T, prob, length, mn) in enumerate(g["cases"]):
        for seed in (0, 1234):
            a = compute_mask_indices((int(b), int(T)), float(prob), int(length), int(mn), np.random.RandomState(seed))
            assert np.array_equal(a, g[f"ref_{i}_{seed}"]), ("ref", i, seed)
            h = compute_mask_indices_hf((int(b), int(T)), float(prob), int(length), int(mn), np.random.RandomState(seed))
            assert np.array_equal(h, g[f"hf_{i}_{seed}"]), ("hf", i, seed)


def test_audio_encoder_hubert_large_architecture():
    """feat_extract_norm='layer' conv stack + stable-layer-norm encoder (HuBERT-large architecture, 2 layers) against
    the reference's HubertModel wrapper; 10 s clip (BASELINE.json configs[3]) and the 30 fps crop path."""
    from helpers import hubert_large_state_dict
    g = load_golden("g3_audio_hubert_large")
    sd = hubert_large_state_dict()
    a10 = synth.audio_clips(1, 160000, tag="audio10s")
    h, st = oa.audio_encoder(sd, "audio_encoder.", oa.pad_audio(a10), 25, frame_num=500, n_heads=16,
                             return_stages=True, stable_layer_norm=True)
    assert h.shape == (1, 500, 1024)
    assert maxabs(st["conv"][:, ::5, ::5], g["conv_10s"]) <= 2e-5
    assert maxabs(st["proj"][:, ::5, ::5], g["proj_10s"]) <= 5e-5
    assert maxabs(st["layer0"][:, ::5, ::5], g["layer0_10s"]) <= 2e-4
    assert maxabs(st["layer1"][:, ::5, ::5], g["layer1_10s"]) <= 2e-4
    assert maxabs(h[:, ::2, ::3], g["hidden_10s"]) <= 1e-4
    a2 = synth.audio_clips(1, 32000, tag="audio30")
    y = oa.audio_encoder(sd, "audio_encoder.", oa.pad_audio(a2), 30, frame_num=60, n_heads=16, stable_layer_norm=True)
    assert maxabs(y, g["hidden_fps30_60"]) <= 1e-4


def test_torch_cpu_restatement_matches_reference_goldens():
    """oracle/torch_cpu.py (the restatement bench.py's cpu_baseline leg times on the GPU box's host cores) against the
    SAME reference goldens the numpy oracle is pinned to: MSMD.forward from raw audio at full depth (g3_forward), one
    sampler-step denoiser call vs the numpy oracle, FLAME vertices (g4_flame)."""
    import torch
    from oracle import torch_cpu as tc
    g = load_golden("g3_forward")
    sd, args = msmd_state_dict("wav2vec2")
    tsd = tc.to_torch(sd)
    sched = od.diffusion_schedule(500, "cosine")
    x = denoiser_inputs(2, args, tag="fw")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    audio = synth.audio_clips(2, 64000, tag="fw_audio")
    _, target, afeat = tc.msmd_forward(tsd, sched, t(x["motion"]), t(audio), t(x["shape"]), t(x["style"]), [3, 499],
                                       t(g["a_eps"]), t(x["indicator"]))
    assert maxabs(afeat.numpy()[:, ::2, ::3], g["a_audio_feat"]) <= 1e-4
    assert maxabs(target.numpy(), g["a_target"]) <= 1e-4
    # one loop body of the sampler (3 CFG entries) vs the numpy denoiser on the same stacked inputs
    B = 2
    xt = synth.normalish("tc/xt", (B, 100, 67))
    got = tc.denoise_step(tsd, t(xt), t(x["audio_feat"]), t(x["shape"]), t(x["style"]), 250, t(x["indicator"])).numpy()
    null_a = np.broadcast_to(sd["null_audio_feat"], x["audio_feat"].shape)
    null_s = np.broadcast_to(sd["null_style_feat"], (B, 1, 256))
    st, sh = x["style"][:, None], x["shape"][:, None]
    rep = lambda v: np.concatenate([v] * 3, 0)
    want = od.denoising_net(sd, rep(xt), np.concatenate([null_a, x["audio_feat"], x["audio_feat"]], 0),
                            np.concatenate([np.concatenate([sh, s], -1) for s in (null_s, null_s, st)], 0), rep(st),
                            rep(np.broadcast_to(sd["start_motion_feat"], (B, 10, 67))),
                            rep(np.broadcast_to(sd["start_audio_feat"], (B, 10, 512))), np.full(3 * B, 250),
                            rep(x["indicator"]))
    assert maxabs(got, want) <= 1e-4
    # the whole sampler loop (default inference configuration) vs the numpy oracle's, which test_sampler pins to g3_sample
    T = 4
    sch = od.diffusion_schedule(T, "cosine")
    xT = synth.normalish("tc/xT", (B, 100, 67))
    zs = {tt: synth.normalish(f"tc/z{tt}", (B, 100, 67)) for tt in range(2, T + 1)}
    want = od.sample(sd, sch, x["audio_feat"], x["shape"], x["style"], xT, zs, indicator=x["indicator"], cfg_scale=1.15)
    got = tc.sample(tsd, sch, t(x["audio_feat"]), t(x["shape"]), t(x["style"]), t(xT), {k: t(v) for k, v in zs.items()},
                    t(x["indicator"]), cfg_scale=1.15).numpy()
    assert maxabs(got, want) <= 1e-4
    # HuBERT-large architecture (LayerNorm conv stack, pre-LN layers) through the torch restatement vs the reference golden
    from helpers import hubert_large_state_dict
    gl = load_golden("g3_audio_hubert_large")
    hsd = hubert_large_state_dict()
    a10 = synth.audio_clips(1, 160000, tag="audio10s")
    h = tc.audio_encoder(tc.to_torch(hsd), t(oa.pad_audio(a10)), 25, frame_num=500, n_heads=16, stable_layer_norm=True).numpy()
    assert h.shape == (1, 500, 1024) and maxabs(h[:, ::2, ::3], gl["hidden_10s"]) <= 1e-4
    # FLAME vertices
    gf = load_golden("g4_flame")
    fo = ofl.FlameOracle(synth.flame_asset())
    fx = flame_inputs(8)
    v = tc.FlameTorch(fo).forward(t(fx["shape"]), t(fx["exp"]), t(gf["pose"])).numpy()
    assert maxabs(v[:, ::79], gf["verts_sub"]) <= 2e-6


def test_vertex_space_loss_gradient_oracle_matches_reference_autograd():
    """oracle.torch_cpu.vertex_space_loss (+ FlameTorch.forward_grad) against gradients recorded from the reference's
    own autograd through utils/common.py:456-620 and FLAME (g8_vertex_grad): losses to 1e-6 relative, d loss / d
    target to 2e-5 of the largest gradient entry, and the FLAME pass alone (d <verts, probe> / d exp, pose)."""
    import torch
    from oracle import torch_cpu as tc
    g = load_golden("g8_vertex_grad")
    fo = ofl.FlameOracle(synth.flame_asset())
    ft = tc.FlameTorch(fo)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    L, P, N = 12, 4, 3
    args = default_args(n_motions=L, n_prev_motions=P, use_vertex_space=True)
    gt = (0.5 * synth.normalish("vgrad/gt", (N, L, 54))).astype(np.float32)
    prev = (0.5 * synth.normalish("vgrad/prev", (N, P, 54))).astype(np.float32)
    tgt = (0.5 * synth.normalish("vgrad/tgt", (N, L + P, 54))).astype(np.float32)
    shape = (0.5 * synth.normalish("vgrad/shape", (N, 100))).astype(np.float32)
    stats = {"exp_mean": t(0.1 * synth.normalish("st/em", (50,))), "exp_std": t(1 + 0.1 * np.abs(synth.normalish("st/es", (50,)))),
             "pose_mean": t(0.05 * synth.normalish("st/pm", (6,))), "pose_std": t(1 + 0.1 * np.abs(synth.normalish("st/ps", (6,)))),
             "shape_mean": t(np.zeros(100, np.float32)), "shape_std": t(np.ones(100, np.float32))}
    end_idx = torch.tensor([L, 5, 9])
    keys = ("noise", "vert", "vel", "smooth", "head_angle", "head_vel", "head_smooth", "head_trans")
    w = dict(zip(keys, g["weights"]))
    for start in (True, False):
        for use_end in (False, True):
            tg = t(tgt).clone().requires_grad_(True)
            ld = tc.vertex_space_loss(args, start, t(shape), t(gt), tg, t(prev), stats, ft, end_idx if use_end else None)
            total = sum(w[k] * v for k, v in ld.items() if v is not None)
            total.backward()
            key = f"{int(start)}_{int(use_end)}"
            want = g["loss_" + key]
            got = np.array([np.nan if ld[k] is None else float(ld[k]) for k in keys])
            assert np.allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=2e-5, atol=1e-9), (key, got, want)
            assert maxabs(tg.grad.numpy(), g["grad_" + key]) <= 2e-5 * np.abs(g["grad_" + key]).max(), key
    x = flame_inputs(6, tag="vgrad_flame")
    ex, po = t(x["exp"]).clone().requires_grad_(True), t(x["pose"]).clone().requires_grad_(True)
    v = ft.forward_grad(t(x["shape"]), ex, po)
    (v * t(synth.normalish("vgrad/probe", (6, 5023, 3)))).sum().backward()
    assert maxabs(ex.grad.numpy(), g["flame_dexp"]) <= 2e-5 * np.abs(g["flame_dexp"]).max()
    assert maxabs(po.grad.numpy(), g["flame_dpose"]) <= 2e-5 * np.abs(g["flame_dpose"]).max()
