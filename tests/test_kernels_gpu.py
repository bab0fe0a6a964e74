"""GPU parity tests of the individual HIP kernels through the C ABI (ctypes) against the numpy oracle.

fp32 parity mode: tolerances are absolute fp32 noise for O(1) values (stated per test).
bf16 speed mode: inputs are rounded to bf16 first so that only accumulation order / output rounding differ;
tolerance 2^-7 relative to the output scale (one bf16 ulp of the largest value)."""
import math

import numpy as np
import pytest
import torch

from oracle import audio_encoder as oa, diffusion as od, flame as ofl, nn as onn, rotations as orot
from msmd_amd import synth

from conftest import load_golden
from helpers import flame_inputs, maxabs

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from msmd_amd import ops as _ops
    return _ops


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).float().numpy()


def test_library_shares_torch_hip_runtime():
    """A pointer allocated by torch must be usable by our kernels on torch's current stream."""
    o = ops()
    x = torch.arange(4096, device=DEV, dtype=torch.float32).reshape(4, 1024)
    y = o.pad_cols(x, 1024, torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(x, y)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 200, 96), (37, 71, 256), (1000, 512, 1536), (64, 48, 6144),
                                   (5, 512, 360)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gemm_matches_oracle(M, N, K, dtype):
    o = ops()
    a = synth.normalish(f"gemm/a/{M}x{K}", (M, K))
    w = synth.uniform(f"gemm/w/{N}x{K}", (N, K), -1, 1) / math.sqrt(K)
    b = synth.uniform(f"gemm/b/{N}", (N,), -0.5, 0.5)
    r = synth.normalish(f"gemm/r/{M}x{N}", (M, N))
    if dtype == "bf16":
        a, w, r = bf16_round(a), bf16_round(w), bf16_round(r)
        td = torch.bfloat16
    else:
        td = torch.float32
    for act, use_r in ((0, False), (1, True), (2, False)):
        ref = onn.linear(a, w, b)
        ref = onn.gelu(ref) if act == 1 else (onn.elu(ref) if act == 2 else ref)
        if use_r:
            ref = ref + r
        out = o.gemm(dev(a, td), dev(w, td), dev(b), dev(r, td) if use_r else None, act)
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        tol = 2e-5 if dtype == "fp32" else 2 ** -7 * max(1.0, float(np.abs(ref).max()))
        assert maxabs(got, ref) <= tol, (M, N, K, dtype, act, maxabs(got, ref))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gemm_is_strided_conv1d(dtype):
    """Windowed-A addressing == nn.Conv1d(k=3, s=2) / (k=2, s=2) over a channels-last signal."""
    o = ops()
    B, T, C, Co = 3, 101, 64, 80
    x = synth.normalish("conv/x", (B, T, C))
    td = torch.float32
    if dtype == "bf16":
        x, td = bf16_round(x), torch.bfloat16
    for k, s in ((3, 2), (2, 2), (3, 1)):
        w = synth.uniform(f"conv/w{k}", (Co, C, k)) / math.sqrt(C * k)
        if dtype == "bf16":
            w = bf16_round(w)
        ref = onn.gelu(onn.conv1d_cl(x, w, None, stride=s))
        wp = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(Co, k * C))
        out = o.conv1d_cl(dev(x, td), dev(wp, td), None, kernel=k, stride=s, act=1)
        torch.cuda.synchronize()
        assert out.shape == ref.shape
        tol = 2e-5 if dtype == "fp32" else 2 ** -7
        assert maxabs(out.float().cpu().numpy(), ref) <= tol, (k, s, dtype)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_layernorm(dtype):
    o = ops()
    for rows, cols in ((7, 512), (33, 768), (5, 256), (3, 1024)):
        x = synth.normalish(f"ln/x{rows}x{cols}", (rows, cols)) * 2 + 0.3
        r = synth.normalish(f"ln/r{rows}x{cols}", (rows, cols))
        g = 1 + 0.1 * synth.uniform(f"ln/g{cols}", (cols,))
        b = 0.1 * synth.uniform(f"ln/b{cols}", (cols,))
        post = synth.uniform(f"ln/p{cols}", (cols,))
        td = torch.float32
        if dtype == "bf16":
            x, r, td = bf16_round(x), bf16_round(r), torch.bfloat16
        ref = onn.layer_norm(onn.elu(x + r), g, b) + post
        out = o.layernorm(dev(x, td), dev(g), dev(b), residual=dev(r, td), post_add=dev(post), act=2)
        torch.cuda.synchronize()
        tol = 1e-5 if dtype == "fp32" else 2 ** -6
        assert maxabs(out.float().cpu().numpy(), ref) <= tol, (rows, cols, dtype)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,H,Tq,Tk,masked", [(2, 12, 200, 200, False), (3, 8, 111, 110, True), (1, 8, 111, 111, False),
                                              (1, 2, 500, 500, False), (2, 8, 100, 100, False), (2, 8, 261, 261, False),
                                              (2, 8, 261, 260, True), (1, 4, 300, 209, False)])
def test_attention(dtype, B, H, Tq, Tk, masked):
    o = ops()
    d = H * 64
    q = synth.normalish(f"att/q{B}{H}{Tq}", (B, Tq, d))
    k = synth.normalish(f"att/k{B}{H}{Tk}", (B, Tk, d))
    v = synth.normalish(f"att/v{B}{H}{Tk}", (B, Tk, d))
    td = torch.float32
    if dtype == "bf16":
        q, k, v, td = bf16_round(q), bf16_round(k), bf16_round(v), torch.bfloat16
    mask = od.alignment_mask(10, Tk - 10, 1) if masked else None      # (1 + 10 + L) x (10 + L)
    qh = q.reshape(B, Tq, H, 64).transpose(0, 2, 1, 3)
    kh = k.reshape(B, Tk, H, 64).transpose(0, 2, 1, 3)
    vh = v.reshape(B, Tk, H, 64).transpose(0, 2, 1, 3)
    s = np.matmul(qh, kh.transpose(0, 1, 3, 2)) * np.float32(0.125)
    if mask is not None:
        s = np.where(mask[None, None], -np.inf, s)
    ref = np.matmul(onn.softmax(s), vh).transpose(0, 2, 1, 3).reshape(B, Tq, d)
    # packed QKV layout exercises the stride arguments
    qkv = torch.cat([dev(q, td), dev(k, td), dev(v, td)], dim=-1) if Tq == Tk else None
    if qkv is not None:
        out = o.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, 0.125)
    else:
        out = o.attention(dev(q, td), dev(k, td), dev(v, td), H, 0.125, mask=dev(mask) if mask is not None else None)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == "fp32" else 2 ** -6
    assert maxabs(out.float().cpu().numpy(), ref) <= tol, (dtype, B, H, Tq, Tk, maxabs(out.float().cpu().numpy(), ref))


def test_pad_audio_bit_exact():
    o = ops()
    for L in (31999, 32000, 32001, 32081, 64000, 64079, 160000):
        r, rep = oa.pad_audio_plan(L)
        x = np.arange(2 * L, dtype=np.float32).reshape(2, L)
        got = o.pad_audio(dev(x), r, rep).cpu().numpy()
        assert np.array_equal(got, oa.pad_audio(x)), L


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_conv0_groupnorm_gelu(dtype):
    o = ops()
    g = load_golden("g3_audio_wav2vec2")
    from helpers import msmd_state_dict
    sd, _ = msmd_state_dict("wav2vec2")
    p = "audio_encoder.feature_extractor.conv_layers.0."
    audio = synth.audio_clips(2, 64000)
    r, rep = oa.pad_audio_plan(64000)
    w0 = np.ascontiguousarray(sd[p + "conv.weight"].reshape(512, 10))
    out = o.conv0_gn_gelu(dev(audio), dev(w0), dev(sd[p + "layer_norm.weight"]), dev(sd[p + "layer_norm.bias"]), r, rep,
                          torch.float32 if dtype == "fp32" else torch.bfloat16)
    torch.cuda.synchronize()
    assert out.shape == (2, 12815, 512)
    got = out.float().cpu().numpy()
    tol = 2e-5 if dtype == "fp32" else 2 ** -7 * 4
    assert maxabs(got[:, ::61, ::7], g["conv0"]) <= tol  # golden = reference's conv0+GN+GELU output


def test_interp_linear_bit_exact_index_and_weights():
    o = ops()
    g = load_golden("g1_index")
    for fps, frame_num, T50 in ((25, 200, 200), (30, 200, 400), (25, 500, 500), (30, 120, 250)):
        crop = oa.crop_len(frame_num, fps)
        ramp = np.arange(T50, dtype=np.float32)[None, :, None] * np.ones((1, 1, 4), np.float32)
        got = o.interp_linear(dev(ramp), frame_num, crop).cpu().numpy()[0, :, 0]
        assert np.array_equal(got, g[f"interp_{fps}_{frame_num}_{T50}"]), (fps, frame_num)
    x = synth.normalish("interp/x", (2, 200, 768))
    got = o.interp_linear(dev(x), 100).cpu().numpy()
    assert np.array_equal(got, onn.interp_linear_cl(x, 100))  # exact pairwise mean


def test_group_pad():
    o = ops()
    x = synth.normalish("gp/x", (2, 50, 96))
    y = o.group_pad(dev(x), 2, 8).cpu().numpy()
    assert y.shape == (2, 2, 66, 48)
    assert np.all(y[:, :, :8] == 0) and np.all(y[:, :, 58:] == 0)
    assert np.array_equal(y[:, 1, 8:58], x[:, :, 48:])


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16x3_valu"])
def test_flame_lbs_matches_oracle_and_golden(precision):
    """fp32: exact-fp32 MFMA.  bf16x3: split-bf16 blendshape products with fp32 accumulation and the joint blend as one
    fp16-split MFMA per transform component (msmd_lbs_skin_v2, the default); bf16x3_valu: the blend on the vector ALU
    (the round-1 kernel) -- same 5e-6 tolerance on FLAME-scale (|v| ~ 0.1) vertices."""
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    from types import SimpleNamespace
    g = load_golden("g4_flame")
    asset = synth.flame_asset()
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = asset
    cfg.lbs_precision = precision
    fl = FLAME(cfg).to(DEV)
    x = flame_inputs(8)
    pose = g["pose"]
    v, lm2d, lm3d = fl(dev(x["shape"]), dev(x["exp"]), dev(pose))
    torch.cuda.synchronize()
    v = v.cpu().numpy()
    assert v.shape == (8, 5023, 3)
    # FLAME vertex coordinates are O(0.1); fp32 accumulation-order noise only
    assert maxabs(v[:, ::79], g["verts_sub"]) <= 5e-6
    assert maxabs(lm2d.cpu().numpy(), g["lm2d"]) <= 5e-6 and maxabs(lm3d.cpu().numpy(), g["lm3d"]) <= 5e-6
    orc = ofl.FlameOracle(asset)
    vo, _, _ = orc.forward(x["shape"], x["exp"], pose, return_lm2d=False, return_lm3d=False)
    assert maxabs(v, vo) <= 5e-6
    v2, _, _ = fl(dev(x["shape"]), dev(x["exp"]), dev(pose), ignore_global_rot=True, return_lm2d=False,
                  return_lm3d=False)
    assert maxabs(v2.cpu().numpy()[:, ::79], g["verts_nog_sub"]) <= 5e-6
    # dynamic-contour LUT row: bit-exact integers
    row = ops().dynamic_lmk_row(dev(orc.full_pose(pose)), dev(orc.neck_kin_chain.astype(np.int32)))
    assert np.array_equal(row.cpu().numpy(), ofl.dynamic_lmk_index(orc.full_pose(pose), orc.neck_kin_chain))
    # ragged frame counts (not a multiple of the 16-frame tile) and a large batch
    for B in (1, 17, 100, 1000):
        xi = flame_inputs(B, tag=f"flame{B}")
        vi, _, _ = fl(dev(xi["shape"]), dev(xi["exp"]), dev(xi["pose"]), return_lm2d=False, return_lm3d=False)
        vr, _, _ = orc.forward(xi["shape"], xi["exp"], xi["pose"], return_lm2d=False, return_lm3d=False)
        assert maxabs(vi.cpu().numpy(), vr) <= 5e-6, B
    # one subject, many frames: every frame carries the same shape row -> folded template, shape K groups skipped on the
    # device's own decision (msmd_flame_prepare); one differing frame must send the call back to the general path
    for B, poke in ((100, False), (37, False), (100, True)):
        xi = flame_inputs(B, tag=f"flame_uni{B}")
        shp = np.repeat(xi["shape"][:1], B, 0).copy()
        if poke:
            shp[B // 2, 5] += 0.5
        vi, _, _ = fl(dev(shp), dev(xi["exp"]), dev(xi["pose"]), return_lm2d=False, return_lm3d=False)
        vr, _, _ = orc.forward(shp, xi["exp"], xi["pose"], return_lm2d=False, return_lm3d=False)
        assert maxabs(vi.cpu().numpy(), vr) <= 5e-6, (B, poke)
    # explicit eye poses (utils/flame.py:196-203): the in-place kinematics path (msmd_flame_prepare), the general path
    # (concatenated full pose) and the oracle's lbs() on the same 15-d pose
    xi = flame_inputs(33, tag="flame_eye")
    eye = (0.3 * synth.normalish("flame_eye/eye", (33, 6))).astype(np.float32)
    v_fast = fl(dev(xi["shape"]), dev(xi["exp"]), dev(xi["pose"]), dev(eye), return_lm2d=False, return_lm3d=False)[0]
    v_gen = fl(dev(xi["shape"]), dev(xi["exp"]), dev(xi["pose"]), dev(eye), return_lm2d=True, return_lm3d=False)[0]
    fp = np.concatenate([xi["pose"][:, :3], np.zeros((33, 3), np.float32), xi["pose"][:, 3:], eye], 1)
    vr, _ = ofl.lbs(np.concatenate([xi["shape"], xi["exp"]], 1), fp, orc.v_template, orc.shapedirs, orc.posedirs,
                    orc.J_regressor, orc.parents, orc.lbs_weights)
    assert maxabs(v_fast.cpu().numpy(), vr) <= 5e-6 and maxabs(v_gen.cpu().numpy(), vr) <= 5e-6


def test_flame_skinning_kernels_agree_over_many_launches():
    """Soak: msmd_lbs_skin_v2 (joint blend on the matrix pipe) against msmd_lbs_skin_bf16x3 (blend on the vector ALU) on
    fresh random inputs, 120 launches over ragged and tile-aligned frame counts.  The two kernels share the blendshape
    arithmetic, so they agree to fp32 noise; a schedule-dependent hazard in either (an experimental build of round 2
    mis-computed rows 13 / 15 of the first 16-frame tile in some workgroups only) shows as an O(1e-2) outlier."""
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    from types import SimpleNamespace
    o = ops()
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    fl = FLAME(cfg).to(DEV)
    fl(torch.zeros(2, 100, device=DEV), torch.zeros(2, 50, device=DEV), torch.zeros(2, 6, device=DEV), return_lm2d=False,
       return_lm3d=False)
    c = fl._pack()["lbs"]
    for it in range(120):
        B = (16, 64, 100, 1000, 17, 3200)[it % 6]
        g = torch.Generator(device=DEV).manual_seed(it)
        betas = torch.cat([0.3 * torch.randn(B, 100, device=DEV, generator=g),
                           0.5 * torch.randn(B, 50, device=DEV, generator=g)], 1)
        pose = 0.3 * torch.randn(B, 15, device=DEV, generator=g)
        coef, coef_hl, A, joints, at = o.lbs_prepare(betas, pose, c.JS, c.parents, 192, want_split=True,
                                                     want_blend_tiles=True)
        ref = o.lbs_skin_bf16x3(coef_hl, A, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
        out = o.lbs_skin_v2(at, B, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
        assert float((out - ref).abs().max()) < 2e-6, (it, B)


def test_batch_rodrigues():
    from msmd_amd.utils.lbs import batch_rodrigues
    g = load_golden("g4_flame")
    got = batch_rodrigues(dev(g["rodrigues_in"])).cpu().numpy()
    assert maxabs(got, g["rodrigues_out"]) <= 1e-6


def test_rotation_conversions():
    from msmd_amd.utils import rotation_conversions as RC
    g = load_golden("g4_rotations")
    aa, q, q2, pts, d6, eul = (dev(g[k]) for k in ("aa", "q", "q2", "pts", "d6", "eul"))
    R = dev(g["axis_angle_to_matrix"])
    got = dict(
        axis_angle_to_matrix=RC.axis_angle_to_matrix(aa), axis_angle_to_quaternion=RC.axis_angle_to_quaternion(aa),
        quaternion_to_matrix=RC.quaternion_to_matrix(q), matrix_to_quaternion=RC.matrix_to_quaternion(R),
        quaternion_to_axis_angle=RC.quaternion_to_axis_angle(q), matrix_to_axis_angle=RC.matrix_to_axis_angle(R),
        rotation_6d_to_matrix=RC.rotation_6d_to_matrix(d6), matrix_to_rotation_6d=RC.matrix_to_rotation_6d(R),
        axis_angle_to_rotation_6d=RC.axis_angle_to_rotation_6d(aa),
        quaternion_raw_multiply=RC.quaternion_raw_multiply(q, q2), quaternion_multiply=RC.quaternion_multiply(q, q2),
        quaternion_invert=RC.quaternion_invert(q), quaternion_apply=RC.quaternion_apply(q, pts),
        standardize_quaternion=RC.standardize_quaternion(q))
    for k, v in got.items():
        tol = 2e-5 if k in ("matrix_to_axis_angle", "quaternion_apply", "matrix_to_quaternion",
                            "quaternion_to_axis_angle") else 2e-6
        assert v.shape == g[k].shape, k
        assert maxabs(v.cpu().numpy(), g[k]) <= tol, (k, maxabs(v.cpu().numpy(), g[k]))
    for conv in ("XYZ", "ZYX", "YXZ", "XYX", "ZXZ"):
        assert maxabs(RC.euler_angles_to_matrix(eul, conv).cpu().numpy(), g[f"euler_angles_to_matrix_{conv}"]) <= 2e-6
        assert maxabs(RC.matrix_to_euler_angles(dev(g[f"euler_angles_to_matrix_{conv}"]), conv).cpu().numpy(),
                      g[f"matrix_to_euler_angles_{conv}"]) <= 2e-5
    with pytest.raises(ValueError):
        RC.euler_angles_to_matrix(eul, "XXY")
    with pytest.raises(ValueError):
        RC.matrix_to_quaternion(torch.zeros(2, 3, 4, device=DEV))


def test_cfg_ddpm_step_matches_reference_inplace_semantics():
    o = ops()
    B, L, Lp, dm = 2, 100, 10, 67
    for mode, name in ((0, "incremental"), (1, "independent")):
        res = synth.normalish(f"cfg/res{mode}", (3 * B, Lp + L, dm))
        x = synth.normalish("cfg/x", (B, L, dm))
        z = synth.normalish("cfg/z", (B, L, dm))
        scales = np.array([1.3, 0.9], np.float32)
        r = [c.copy() for c in np.split(res, 3, axis=0)]
        theta = r[0][:, -L:]
        for i in range(2):
            theta += scales[i] * (r[i + 1][:, -L:] - (r[0] if mode == 1 else r[i])[:, -L:])
        c0, c1, sg = np.float32(0.7), np.float32(0.25), np.float32(0.1)
        ref = c0 * x + c1 * theta + sg * z
        xt = dev(x).clone()
        o.cfg_ddpm_step(xt, dev(res), dev(z), dev(scales), 3, Lp, mode, 0, c0, c1, sg)
        torch.cuda.synchronize()
        assert maxabs(xt.cpu().numpy(), ref) <= 2e-6, name


@pytest.mark.parametrize("M,N,K", [(6400, 768, 768), (3552, 512, 72), (200, 64, 200), (111, 2304, 512), (64, 8, 8),
                                   (1000, 136, 3072)])
def test_gemm_tn_weight_gradient_product(M, N, K):
    """msmd_gemm_tn: C = A^T B over row-major operands (+ fused column sums) against fp64 on the same bf16 inputs;
    covers contraction tails (M % 64 != 0), ragged N / K tiles and the split-contraction atomics path."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    b = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ref = a.double().t() @ b.double()
    cs_ref = a.double().sum(0)
    for splits in (0, 1, 3):
        c, cs = o.gemm_tn(a, b, want_colsum=True, splits=splits)
        torch.cuda.synchronize()
        tol = 2e-3 * float(ref.abs().max()) + 1e-3
        assert float((c.double() - ref).abs().max()) < tol, (splits, float((c.double() - ref).abs().max()))
        assert float((cs.double() - cs_ref).abs().max()) < 2e-3 * float(cs_ref.abs().max()) + 1e-3
    # batched, strided views
    a3 = (torch.randn(3, 120, 64, generator=g)).to(torch.bfloat16).to(DEV)
    b3 = (torch.randn(3, 120, 40, generator=g)).to(torch.bfloat16).to(DEV)
    c3 = o.gemm_tn(a3, b3, batch=3, strideA=120 * 64, strideB=120 * 40)
    ref3 = a3.double().transpose(1, 2) @ b3.double()
    assert float((c3.double() - ref3).abs().max()) < 2e-3 * float(ref3.abs().max())


def test_gemm_tn_windowed_operand_is_conv_weight_gradient():
    """B operand as overlapping conv windows of a padded group-major signal (PosConvFn.backward): equals the
    explicit unfold + matmul in fp64."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(7)
    Bn, T, G, cg, kpos = 3, 50, 4, 16, 8
    Tp = T + kpos
    xp = torch.randn(Bn, G, Tp, cg, generator=g).to(torch.bfloat16).to(DEV)
    dz = torch.randn(Bn, T, G * cg, generator=g).to(torch.bfloat16).to(DEV)
    got = o.gemm_tn(dz, xp, M=Bn * T, N=cg, K=kpos * cg, lda=G * cg, ldb=cg, batch=G, strideA=cg, strideB=Tp * cg,
                    b_rows_per_window=T, b_window_stride=G * Tp * cg)
    win = torch.stack([xp[:, :, t:t + kpos].reshape(Bn, G, kpos * cg) for t in range(T)], dim=1).double()  # (B,T,G,K)
    ref = torch.einsum("btgn,btgk->gnk", dz.double().reshape(Bn, T, G, cg), win)
    assert float((got.double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("N,T_all,L,C,q", [(6, 110, 100, 67, 0.9), (3, 110, 100, 67, 0.0), (2, 260, 250, 67, 0.995),
                                            (4, 10, 10, 7, 0.5), (2, 110, 100, 67, 1.0)])
def test_dynamic_threshold_matches_torch_quantile(N, T_all, L, C, q):
    """msmd_dynamic_threshold against the reference's expression (model.py:396-402) evaluated with torch: the
    threshold is an order statistic / ATen lerp of two of them, so the result is bit-exact."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(N * 100 + L)
    res = (torch.randn(N, T_all, C, generator=g) * 1.3).to(DEV)
    res[0, -L:, :3] = 0.25                              # ties around the order statistic
    s = torch.quantile(res[:, -L:].reshape(N, -1).abs(), q, dim=1)
    s = torch.clamp(s, min=0.5, max=2.0)[..., None, None]
    want = torch.clamp(res, min=-s, max=s)
    got = o.dynamic_threshold_(res.clone(), L, q, 0.5, 2.0)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_flame_rotation_matrix_pose_equals_axis_angle_pose():
    """pose2rot=False (reference utils/flame.py:199-205, 144-149): feeding the rotation matrices of the same
    axis-angle pose reproduces the golden vertices, dynamic (2-D) and static landmarks, and the LUT rows exactly."""
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    from types import SimpleNamespace
    g = load_golden("g4_flame")
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    fl = FLAME(cfg).to(DEV)
    x = flame_inputs(8)
    pose = dev(g["pose"])                                             # (8, 6): global head + jaw axis-angle
    R = ops().batch_rodrigues(pose.reshape(-1, 3).contiguous()).reshape(8, 18)
    v, lm2d, lm3d = fl(dev(x["shape"]), dev(x["exp"]), R, pose2rot=False)
    torch.cuda.synchronize()
    assert maxabs(v.cpu().numpy()[:, ::79], g["verts_sub"]) <= 5e-6
    assert maxabs(lm2d.cpu().numpy(), g["lm2d"]) <= 5e-6 and maxabs(lm3d.cpu().numpy(), g["lm3d"]) <= 5e-6
    va, la, _ = fl(dev(x["shape"]), dev(x["exp"]), pose)
    assert maxabs(v.cpu().numpy(), va.cpu().numpy()) <= 2e-6 and maxabs(lm2d.cpu().numpy(), la.cpu().numpy()) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(6400, 768, 768), (300, 136, 512), (64, 72, 40)])
def test_gemm_and_attention_fp16_storage(M, N, K):
    """fp16 operands through the LDS-DMA MFMA kernels and the register-staged fallback (odd K) against fp64 on the
    same fp16-rounded inputs; attention in fp16 against torch in fp32."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).half().to(DEV)
    y = o.gemm(a, w, b, r, o.ACT_GELU)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
    assert y.dtype == torch.float16 and float((y.double() - ref).abs().max()) < 4e-3
    y32 = o.gemm(a, w, b, None, o.ACT_NONE, out_dtype=torch.float32)
    assert float((y32.double() - (a.double() @ w.double().t() + b.double())).abs().max()) < 2e-4
    if K == 768:
        B, T, H = 2, 200, 12
        qkv = (torch.randn(B, T, 3 * H * 64, generator=g) * 0.7).half().to(DEV)
        d = H * 64
        out = o.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, 0.125)
        hd = lambda t: t.float().reshape(B, T, H, 64).transpose(1, 2)
        want = (torch.softmax(hd(qkv[..., :d]) @ hd(qkv[..., d:2 * d]).transpose(-1, -2) * 0.125, -1)
                @ hd(qkv[..., 2 * d:])).transpose(1, 2).reshape(B, T, d)
        assert float((out.float() - want).abs().max()) < 3e-3
        ln = o.layernorm(qkv[..., :d].contiguous(), torch.ones(d, device=DEV), torch.zeros(d, device=DEV))
        assert ln.dtype == torch.float16
        assert float((ln.float() - torch.nn.functional.layer_norm(qkv[..., :d].float(), (d,))).abs().max()) < 4e-3


@pytest.mark.gpu
@pytest.mark.parametrize("N,T,Tk,H", [(5, 111, 110, 8), (128, 111, 110, 8), (3, 1, 1, 8), (2, 7, 512, 4), (9, 4, 65, 12)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
def test_person_query_attention_matches_projection_plus_attention(N, T, Tk, H, dtype, tol):
    """msmd_person_query_attention (one launch) == q-projection GEMM of row 0 + Tq = 1 msmd_attention + torch fp64
    reference; fp32 tolerance 2e-6 (same arithmetic, different summation order), bf16 / fp16: storage rounding."""
    O = ops()
    torch.manual_seed(N * 1000 + Tk)
    d = H * 64
    x = torch.randn(N, T, d, device="cuda")
    wq = torch.randn(d, d, device="cuda") / d ** 0.5
    bq = torch.randn(d, device="cuda") * 0.1
    kv = torch.randn(N, Tk, 2 * d, device="cuda")
    scale = 64 ** -0.5
    xd, wd, kvd = x.to(dtype), wq.to(dtype), kv.to(dtype)
    got = O.person_query_attention(xd, wd, bq, kvd, H, scale).float()
    # fp64 reference on the rounded operands
    q = (xd[:, 0].double() @ wd.double().t() + bq.double()).view(N, H, 1, 64)
    k = kvd[..., :d].double().view(N, Tk, H, 64).permute(0, 2, 1, 3)
    v = kvd[..., d:].double().view(N, Tk, H, 64).permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).permute(0, 2, 1, 3).reshape(N, d).float()
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    # and the two-launch path it replaces
    q0 = O.gemm(xd, wd, bq, M=N, K=d, lda=T * d)
    two = O.attention(q0.view(N, 1, d), kvd[..., :d], kvd[..., d:], H, scale).view(N, d).float()
    assert (got - two).abs().max().item() <= 2 * tol * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_lbs_building_blocks_match_reference():
    """utils/lbs.py's pieces under the reference's names (blend_shapes, vertices2joints, transform_mat,
    batch_rigid_transform, find_dynamic_lmk_idx_and_bcoords) against goldens from the reference on the synthetic FLAME
    asset; fp32 tolerance 5e-6 on coordinates (exact-fp32 MFMA contraction vs einsum), LUT rows exact."""
    from msmd_amd.utils import lbs as L
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    from types import SimpleNamespace
    g = load_golden("g4_lbs_blocks")
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = synth.flame_asset()
    fl = FLAME(cfg).to(DEV)
    B = 4
    x = flame_inputs(B, tag="blocks")
    betas = torch.cat([dev(x["shape"]), dev(x["exp"])], dim=1)
    bs = L.blend_shapes(betas, fl.shapedirs)
    assert bs.shape == (B, fl.v_template.shape[0], 3) and maxabs(bs.cpu().numpy()[:, ::79], g["blend_sub"]) <= 5e-6
    v_shaped = fl.v_template.unsqueeze(0) + bs
    J = L.vertices2joints(fl.J_regressor, v_shaped)
    assert maxabs(J.cpu().numpy(), g["joints"]) <= 5e-6
    full_pose = dev(g["full_pose"])
    rot = L.batch_rodrigues(full_pose.view(-1, 3)).view(B, -1, 3, 3)
    posed, rel = L.batch_rigid_transform(rot, J, fl.parents)
    assert maxabs(posed.cpu().numpy(), g["posed"]) <= 5e-6 and maxabs(rel.cpu().numpy(), g["rel"]) <= 5e-6
    tm = L.transform_mat(rot[:, 1], J[:, 1].unsqueeze(-1))
    assert maxabs(tm.cpu().numpy(), g["tmat"]) <= 5e-6
    fi, bc = L.find_dynamic_lmk_idx_and_bcoords(v_shaped, full_pose, fl.dynamic_lmk_faces_idx,
                                                fl.dynamic_lmk_bary_coords, fl.neck_kin_chain)
    assert np.array_equal(fi.cpu().numpy(), g["dyn_idx"]) and np.array_equal(bc.cpu().numpy(), g["dyn_bary"])
    # FLAME's own method (reference utils/flame.py:126-172, what forward() uses) looks the tables up at PLUS the yaw, the
    # module-level function above at minus the yaw: rows from the oracle's restatement of the method
    fi2, bc2 = fl._find_dynamic_lmk_idx_and_bcoords(full_pose, fl.dynamic_lmk_faces_idx, fl.dynamic_lmk_bary_coords, fl.neck_kin_chain)
    rows = ofl.dynamic_lmk_index(g["full_pose"], fl.neck_kin_chain.cpu().numpy())
    assert np.array_equal(fi2.cpu().numpy(), fl.dynamic_lmk_faces_idx.cpu().numpy()[rows])
    assert np.array_equal(bc2.cpu().numpy(), fl.dynamic_lmk_bary_coords.cpu().numpy()[rows])
    from msmd_amd.utils.wav2vec2 import _compute_mask_indices, compute_mask_indices
    np.random.seed(4)
    a = _compute_mask_indices((3, 199), 0.05, 10, None, 2)
    np.random.seed(4)
    assert np.array_equal(a, compute_mask_indices((3, 199), 0.05, 10, 2))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_store_flags_and_variants_are_bit_identical(dtype):
    """Per-call knobs (include/msmd_hip.h MSMD_GEMM_VARIANT / _WRITE_THROUGH / _PAIRED_STORES) change tile shape and
    store instructions only: same K order inside a tile => every combination must give the SAME bits (interior tiles
    and ragged edges, with bias + GELU + residual)."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 1000, 776, 512            # ragged M, N % 8 == 0 but not a tile multiple
    a = torch.randn(M, K, generator=g).to(DEV, dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV, dtype)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV, dtype)
    ref = o.gemm(a, w, b, r, act=o.ACT_GELU, variant=17, flags=0)
    # 60-64: the v4 kernels (fragment reads pipelined inside the wave, 256 x 128 / 128 x 128 tiles; bf16 instantiations)
    variants = (0, 9, 12, 14, 15, 17) + ((13, 60, 61, 62, 63, 64) if dtype == torch.bfloat16 else ())
    for v in variants:
        for fl in (0, o.GEMM_WRITE_THROUGH, o.GEMM_PAIRED_STORES, o.GEMM_WRITE_THROUGH | o.GEMM_PAIRED_STORES,
                   o.GEMM_PAIRED_STORES | o.GEMM_STAGGER):
            c = o.gemm(a, w, b, r, act=o.ACT_GELU, variant=v, flags=fl)
            assert torch.equal(c, ref), (v, fl, float((c.float() - ref.float()).abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_persistent_multi_round_form_is_bit_identical(dtype):
    """Launches of more than 512 tiles of the 128 x 128 kernel run as 512 persistent workgroups (gemm2p_kernel: the next tile's
    first operand stage is issued by the last K tile of the current one).  Same products in the same order per tile: the
    outputs and the LayerNorm statistics equal the one-tile-per-workgroup form (MSMD_GEMM_ONE_TILE_PER_WORKGROUP) bit for
    bit, on a grid with ragged edges in M (6500 rows = 51 tiles x 18 / 24 / 6 column tiles), for every epilogue family."""
    from msmd_amd import ops as O
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(19)
    M, D, F = 6500, 768, 3072
    a = torch.randn(M, D, generator=g).to(DEV, dtype)
    w_qkv = (torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(DEV, dtype)
    b_qkv = torch.randn(3 * D, generator=g).to(DEV)
    r_qkv = torch.randn(M, 3 * D, generator=g).to(DEV, dtype)
    one = o.GEMM_ONE_TILE_PER_WORKGROUP
    for act in (o.ACT_NONE, o.ACT_GELU):          # plain epilogue, with and without residual
        for r in (None, r_qkv):
            c0 = o.gemm(a, w_qkv, b_qkv, r, act=act, variant=17, flags=o.GEMM_PAIRED_STORES | one)
            c1 = o.gemm(a, w_qkv, b_qkv, r, act=act, variant=17, flags=o.GEMM_PAIRED_STORES)
            assert torch.equal(c0, c1), (act, r is not None)
    # LayerNorm-residual form (statistics out, 64-column slabs) and LayerNorm-operand form, tall enough for two rounds
    M2 = 12900                                    # 101 x 6 = 606 tiles of 128 x 128
    u0 = (torch.randn(M2, D, generator=g) * 2 + 0.3).to(DEV, dtype)
    a2 = torch.randn(M2, D, generator=g).to(DEV, dtype)
    w1 = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV, dtype)
    b1 = torch.randn(D, generator=g).to(DEV)
    g0, be0 = (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    x = u0.double().reshape(M2, -1, 64)
    st0 = torch.stack([x.sum(-1), (x * x).sum(-1)], -1).transpose(0, 1).float().contiguous()
    w2f, cs2, b2f = o.fold_layernorm(torch.randn(F, D, generator=g).to(DEV) / math.sqrt(D), torch.randn(F, generator=g).to(DEV),
                                     (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV), dtype)
    outs = []
    for fl in (one, 0):
        O.GEMM_LN_FLAGS = fl
        try:
            c1, s1 = o.gemm_ln(a2, w1, b1, u0, r_stats=st0, r_gamma=g0, r_beta=be0, stats_out=True)
            c1p, s1p = o.gemm_ln(a2, w1, b1, u0, stats_out=True)
            f = o.gemm_ln(c1, w2f, b2f, act=o.ACT_GELU, a_stats=s1, w_colsum=cs2)
            q = o.gemm_ln(c1, w2f, b2f, a_stats=s1, w_colsum=cs2)
        finally:
            O.GEMM_LN_FLAGS = 0
        outs.append((c1, s1, c1p, s1p, f, q))
    assert s1.shape[0] == D // 64
    for x0, x1 in zip(*outs):
        assert torch.equal(x0, x1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_256_tile_8_phase_kernel_equals_the_128_tile_kernels(dtype):
    """Variant 80 (gemm8_kernel: 256 x 256 tiles, 8-phase schedule, epilogue through LDS) multiplies in the same order per
    output element as the 128 x 128 kernels: plain (bias, GELU, residual) outputs must be the SAME bits, on ragged M (6 500 and 300 rows: the last row tile mostly / the only row tile partly empty), K = 128 (no
    steady-state K tile), 192 (one) and 768, and a windowed (conv) A operand.  Through msmd_gemm_ln the LayerNorm forms sum
    their row statistics in another order: outputs within one 16-bit step (and equal almost everywhere), statistics to 2e-3.
    A call it does not take (N % 256 != 0) falls back to the library's own choice."""
    from msmd_amd import ops as O
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(23)
    # (17000, 1024, 192) = 268 tiles on 256 workgroups: the persistent loop's second tile, odd K-tile count (buffer parity)
    for M, N, K in ((6500, 768, 768), (300, 512, 128), (1100, 256, 192), (17000, 1024, 192)):
        a = torch.randn(M, K, generator=g).to(DEV, dtype)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV, dtype)
        b = torch.randn(N, generator=g).to(DEV)
        r = torch.randn(M, N, generator=g).to(DEV, dtype)
        for act in (o.ACT_NONE, o.ACT_GELU):
            for res in (None, r):
                c0 = o.gemm(a, w, b, res, act=act, variant=17, flags=0)
                c1 = o.gemm(a, w, b, res, act=act, variant=80, flags=0)
                if act == o.ACT_GELU and res is not None:
                    # GELU's last multiply and the residual add may or may not be contracted into one FMA by the compiler
                    # in either kernel: one 16-bit step at rounding ties, nothing else
                    d = (c0.float() - c1.float()).abs()
                    assert float(d.max()) <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * 8 and float((d > 0).float().mean()) < 0.01
                else:
                    assert torch.equal(c0, c1), (M, N, K, act, res is not None, float((c0.float() - c1.float()).abs().max()))
        ref = o.gemm(a, w, None, None, variant=17)
        for _ in range(4):        # the same bits every time (the staging / epilogue hand-offs are ordered by counted waits and barriers)
            assert torch.equal(o.gemm(a, w, None, None, variant=80), ref)
    # not taken: falls back
    a = torch.randn(500, 256, generator=g).to(DEV, dtype)
    w = (torch.randn(384, 256, generator=g) / 16).to(DEV, dtype)
    assert torch.equal(o.gemm(a, w, None, None, variant=80), o.gemm(a, w, None, None))
    # windowed A: Conv1d(k=3, stride=2) rows over a (B, T, C) signal
    B, T, C, Nc = 3, 2001, 512, 512
    x = torch.randn(B, T, C, generator=g).to(DEV, dtype)
    wc = (torch.randn(Nc, 3 * C, generator=g) / math.sqrt(3 * C)).to(DEV, dtype)
    To = (T - 3) // 2 + 1
    kw = dict(M=B * To, N=Nc, K=3 * C, lda=2 * C, rows_per_batch=To, a_batch_stride=T * C, ldw=3 * C, ldc=Nc)
    y0 = torch.empty(B, To, Nc, device=DEV, dtype=dtype)
    y1 = torch.empty_like(y0)
    o.gemm(x, wc, None, None, o.ACT_GELU, out=y0, variant=15, **kw)
    o.gemm(x, wc, None, None, o.ACT_GELU, out=y1, variant=80, **kw)
    assert torch.equal(y0, y1)
    # LayerNorm forms through msmd_gemm_ln (tile hint 80 against the 128 x 128 kernel); 22 400 rows = 264 tiles of the
    # 768-column producer: its second tile per workgroup too (the 3 072-column consumer has 312 / 1 056)
    for M, D, F in ((6500, 768, 3072), (22400, 768, 3072)):
        _ln_forms_256_tile(o, O, g, dtype, M, D, F)


def test_flame_lbs_fp16_vertices_against_the_numpy_oracle():
    """msmd_lbs_skin_v2_f16 (opt-in FLAME.vertex_dtype = torch.float16; BASELINE configs[4] names the fp16 LBS pass), both forms:
    `vertex_exact` = the fp32 kernel's arithmetic with one fp16 rounding at the store (equal to the fp32 kernel's output rounded
    to fp16 bit for bit; 2^-11 relative + the fp32 kernel's 5e-6 against the numpy FLAME oracle), and the default on fp16 operand
    planes (the store's 2^-11 relative + 2^-10 of sum_k |coef_k| |dirs_k|: the operands' own rounding); ragged frame counts (tile of 16, the lane pairs' frame trade), the odd vertex count (5023:
    rows padded to 5024), the one-subject shape fold (flame_inputs share no shape here: general path) and landmarks from the fp16
    vertices."""
    from msmd_amd.utils.flame import FLAME, FLAMEConfig
    from types import SimpleNamespace
    asset = synth.flame_asset()
    cfg = SimpleNamespace(**vars(FLAMEConfig))
    cfg.asset = asset
    fl = FLAME(cfg).to(DEV)
    orc = ofl.FlameOracle(asset)
    for B in (1, 17, 100, 1003):
        xi = flame_inputs(B, tag=f"flame16_{B}")
        args = (dev(xi["shape"]), dev(xi["exp"]), dev(xi["pose"]))
        v32 = fl(*args, return_lm2d=False, return_lm3d=False)[0]
        vr, _, lr = orc.forward(xi["shape"], xi["exp"], xi["pose"], return_lm2d=False, return_lm3d=True)
        for exact in (True, False):
            fl.vertex_dtype, fl.vertex_exact = torch.float16, exact
            try:
                v16, _, lm3d = fl(*args, return_lm2d=False, return_lm3d=True)
            finally:
                del fl.vertex_dtype, fl.vertex_exact
            assert v16.dtype == torch.float16 and v16.shape == (B, 5023, 3) and v16.stride(0) == 5024 * 3
            err = np.abs(v16.float().cpu().numpy() - vr)
            if exact:       # the fp32 kernel's arithmetic, one rounding at the store
                assert torch.equal(v16, v32.to(torch.float16)), B
                assert np.all(err <= np.abs(vr) * 2.0 ** -11 + 5e-6), (B, float(err.max()))
            else:
                # the default fp16 form: fp16 operand planes (one MFMA per K group).  The header's bound: the store's rounding + the
                # operands' own (2^-11 relative each: 2^-10 on every product) on S = sum_k |coef_k| |dirs_k| of the un-skinned
                # offset (a rotation mixes at most the three coordinates' worth of it) + the fp32 kernel's 5e-6
                betas = np.concatenate([xi["shape"], xi["exp"]], 1).astype(np.float64)
                rot = ofl.batch_rodrigues(orc.full_pose(xi["pose"]).reshape(-1, 3)).reshape(B, -1, 3, 3)
                pf = np.abs((rot[:, 1:] - np.eye(3, dtype=np.float32)).reshape(B, -1)).astype(np.float64)
                S = np.einsum("bl,vcl->bvc", np.abs(betas), np.abs(orc.shapedirs).astype(np.float64)) + \
                    (pf @ np.abs(orc.posedirs).astype(np.float64)).reshape(B, -1, 3)
                bound = np.abs(vr) * 2.0 ** -11 + 2.0 ** -10 * S.sum(-1, keepdims=True) + 5e-6
                assert np.all(err <= bound), (B, float(err.max()), float((err / bound).max()))
                print(f"fp16 LBS, fp16 operand planes, B = {B}: max |err| {err.max():.2e} m (bound used up to {float((err / bound).max()):.2f})")
            assert maxabs(lm3d.cpu().numpy(), lr) <= 2.0 ** -11 * float(np.abs(vr).max()) + (5e-6 if exact else 2e-4)


def _ulp_of_row_max(ref, dtype):
    """one unit in the last place of the 16-bit type at each row's largest |reference| value"""
    rm = ref.abs().amax(dim=1, keepdim=True).clamp_min(2.0 ** -14)
    return torch.exp2(torch.floor(torch.log2(rm)) - (7 if dtype == torch.bfloat16 else 10))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("variant", [80, 17, 15, 13, 9, 12])
def test_gemm_every_routed_variant_meets_the_float64_reference(dtype, variant):
    """Every kernel variant the library routes 16-bit launches to (80 = gemm8_kernel 256 x 256 8-phase, 17 / 13 = 128 x 128,
    15 = 192 x 128, 9 / 12 = 64 x 64), FORCED by its per-call hint, against a float64 product of the same (already rounded)
    operands at the shapes the forward / HuBERT-large / sampler steps route to variant 80: bias, GELU, residual, GELU +
    residual, and the conv stack's windowed A operand.  Bound: ONE unit in the last place of the output type at the row's
    largest value (fp32 accumulation error is far below the output rounding)."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(80 + variant)
    gelu64 = lambda t: torch.nn.functional.gelu(t)
    for M, N, K in ((6400, 2304, 768), (15968, 3072, 1024), (15968, 1024, 1024), (15968, 1024, 4096), (21312, 1536, 512)):
        if variant in (9, 12) and M * N > 6400 * 2304:
            continue          # the 64 x 64 tiles are routed to small grids only: one large shape is enough for them
        a = torch.randn(M, K, generator=g).to(DEV, dtype)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV, dtype)
        b = torch.randn(N, generator=g).to(DEV)
        r = torch.randn(M, N, generator=g).to(DEV, dtype)
        lin = a.double() @ w.double().T + b.double()
        for act in (o.ACT_NONE, o.ACT_GELU):
            for res in (None, r):
                ref = gelu64(lin) if act == o.ACT_GELU else lin
                if res is not None:
                    ref = ref + res.double()
                got = o.gemm(a, w, b, res, act=act, variant=variant, flags=0).double()
                bad = ((got - ref).abs() > _ulp_of_row_max(ref, dtype))
                assert not bool(bad.any()), (variant, M, N, K, act, res is not None, int(bad.sum()), float((got - ref).abs().max()))
        del lin
    # windowed A: conv1 of the feature extractor's k = 3 / stride 2 layers on 4 clips of 4 s (12 815 input frames of 512 channels)
    if variant in (80, 15, 17):
        B, T, C, Nc = 4, 12815, 512, 512
        x = torch.randn(B, T, C, generator=g).to(DEV, dtype)
        wc = (torch.randn(Nc, 3 * C, generator=g) / math.sqrt(3 * C)).to(DEV, dtype)
        To = (T - 3) // 2 + 1
        rows = x.unfold(1, 3, 2).permute(0, 1, 3, 2).reshape(B * To, 3 * C)      # (B, To, C, 3) -> rows [tap][channel]
        ref = gelu64(rows.double() @ wc.double().T)
        y = torch.empty(B, To, Nc, device=DEV, dtype=dtype)
        o.gemm(x, wc, None, None, o.ACT_GELU, out=y, variant=variant, M=B * To, N=Nc, K=3 * C, lda=2 * C, rows_per_batch=To,
               a_batch_stride=T * C, ldw=3 * C, ldc=Nc)
        got = y.reshape(B * To, Nc).double()
        assert not bool(((got - ref).abs() > _ulp_of_row_max(ref, dtype)).any())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [80, 17])
def test_gemm_ln_forms_meet_the_float64_reference(dtype, tile):
    """msmd_gemm_ln's two epilogue families (LayerNorm of the residual + statistics of the stored rows; LayerNorm folded into the
    A operand) on the 256 x 256 kernel (tile hint 80) and the 128 x 128 kernel (17), each against float64 arithmetic on the same
    16-bit operands and the same fp32 row statistics: outputs within TWO units in the last place at the row's largest value (the
    folded form subtracts mu * colsum from the product in fp32), statistics of the stored rows to 1e-5 relative of the row's sum
    of squares."""
    from msmd_amd import ops as O
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(8017 + tile)
    M, D, F = 6400, 768, 3072
    u0 = (torch.randn(M, D, generator=g) * 2 + 0.3).to(DEV, dtype)
    a2 = torch.randn(M, D, generator=g).to(DEV, dtype)
    w1 = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV, dtype)
    b1 = torch.randn(D, generator=g).to(DEV)
    g0, be0 = (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    xs = u0.double().reshape(M, -1, 64)
    st0 = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).transpose(0, 1).float().contiguous()      # (D / 64, M, 2)
    w2f, cs2, b2f = o.fold_layernorm(torch.randn(F, D, generator=g).to(DEV) / math.sqrt(D), torch.randn(F, generator=g).to(DEV),
                                     (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV), dtype)

    def moments(st, cols):      # what the kernels derive from the fp32 slab statistics, in float64
        S, Q = st.double().sum(0)[:, 0], st.double().sum(0)[:, 1]
        mu = S / cols
        return mu[:, None], torch.rsqrt((Q / cols - mu * mu).clamp_min(0) + 1e-5)[:, None]

    O.GEMM_LN_TILE = tile
    try:
        c1, s1 = o.gemm_ln(a2, w1, b1, u0, r_stats=st0, r_gamma=g0, r_beta=be0, stats_out=True)
        f = o.gemm_ln(c1, w2f, b2f, act=o.ACT_GELU, a_stats=s1, w_colsum=cs2)
    finally:
        O.GEMM_LN_TILE = None
    mu0, rs0 = moments(st0, D)
    ref1 = a2.double() @ w1.double().T + b1.double() + (u0.double() - mu0) * rs0 * g0.double() + be0.double()
    assert not bool(((c1.double() - ref1).abs() > 2 * _ulp_of_row_max(ref1, dtype)).any())
    cs = c1.double().reshape(M, -1, 64)
    want = torch.stack([cs.sum(-1), (cs * cs).sum(-1)], -1).transpose(0, 1)
    assert float(((s1.double() - want).abs() / want[..., 1:2].clamp_min(1.0)).max()) < 1e-5
    mu1, rs1 = moments(s1, D)
    ref2 = torch.nn.functional.gelu(rs1 * (c1.double() @ w2f.double().T - mu1 * cs2.double()) + b2f.double())
    assert not bool(((f.double() - ref2).abs() > 2 * _ulp_of_row_max(ref2, dtype)).any())


def _ln_forms_256_tile(o, O, g, dtype, M, D, F):
    u0 = (torch.randn(M, D, generator=g) * 2 + 0.3).to(DEV, dtype)
    a2 = torch.randn(M, D, generator=g).to(DEV, dtype)
    w1 = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV, dtype)
    b1 = torch.randn(D, generator=g).to(DEV)
    g0, be0 = (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    xs = u0.double().reshape(M, -1, 64)
    st0 = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).transpose(0, 1).float().contiguous()
    w2f, cs2, b2f = o.fold_layernorm(torch.randn(F, D, generator=g).to(DEV) / math.sqrt(D), torch.randn(F, generator=g).to(DEV),
                                     (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV), dtype)
    outs = []
    for tile in (17, 80):
        O.GEMM_LN_TILE = tile
        try:
            c1, s1 = o.gemm_ln(a2, w1, b1, u0, r_stats=st0, r_gamma=g0, r_beta=be0, stats_out=True)
            c1p, s1p = o.gemm_ln(a2, w1, b1, u0, stats_out=True)
            O.GEMM_LN_TILE = 17
            cin, sin = outs[0][0:2] if outs else (c1, s1)       # both consumers read the SAME producer output
            O.GEMM_LN_TILE = tile
            f = o.gemm_ln(cin, w2f, b2f, act=o.ACT_GELU, a_stats=sin, w_colsum=cs2)
            q = o.gemm_ln(cin, w2f, b2f, a_stats=sin, w_colsum=cs2)
        finally:
            O.GEMM_LN_TILE = None
        outs.append((c1, s1, c1p, s1p, f, q))
    (c1a, s1a, c1pa, s1pa, fa, qa), (c1b, s1b, c1pb, s1pb, fb, qb) = outs
    step = 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * 8      # one 16-bit step of |x| < 8
    # Round 6: both kernels sum a row's statistics in ONE association (as producer and as consumer), so every LayerNorm form
    # is the same bits on either tile -- a row's result does not depend on the kernel its launch was routed to
    assert torch.equal(c1pa, c1pb) and torch.equal(c1a, c1b)
    assert s1a.shape == s1b.shape and torch.equal(s1pa, s1pb) and torch.equal(s1a, s1b)
    xs = c1b.double().reshape(M, -1, 64)            # the statistics are those of the rows this kernel stored
    assert float((s1b - torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).transpose(0, 1).float()).abs().max()) < 2e-3
    # LayerNorm-operand form on the same operand and statistics: the same bits (GELU's last multiply meets no residual here)
    assert torch.equal(fa, fb) and torch.equal(qa, qb)
    # statistics in 32-column slabs (what a 64 x 64-tile producer writes): 24 slabs per row here, i.e. past the 16 the kernel
    # keeps in LDS -- the rest is summed from memory
    xs = c1a.double().reshape(M, -1, 32)
    st32 = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).transpose(0, 1).float().contiguous()
    q32 = []
    for tile in (17, 80):
        O.GEMM_LN_TILE = tile
        try:
            q32.append(o.gemm_ln(c1a, w2f, b2f, a_stats=st32, w_colsum=cs2))
        finally:
            O.GEMM_LN_TILE = None
    assert torch.equal(q32[0], q32[1])
    assert float((q32[1].float() - qa.float()).abs().max()) <= 2 * step      # and the same rows as with 64-column statistics


@pytest.mark.parametrize("M", [1000, 6500, 16100])     # small grid (consumer on 64 x 64 tiles); 128 x 128 and (tall grids) 192 x 128 tiles; 64-column slabs always; ragged M
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_ln_folds_layernorm_into_producer_and_consumer(dtype, M):
    """msmd_gemm_ln: Linear -> +residual -> LayerNorm -> Linear as two launches.  Producer: C1 = A W1^T + b1 + LN_R(r)
    (r un-normalised, its row statistics given) and the row statistics of the stored C1; consumer: LN(C1) W2^T + b2
    through gamma-folded weights.  Reference: the same chain in fp64 from the SAME rounded operands; tolerance = the
    16-bit output rounding of O(1) values plus the statistics being exact sums of the stored values (5e-3 * scale for
    bf16, 1e-3 for fp16: the plain chain layernorm kernel + gemm kernel has the same error against fp64)."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(11)
    D, F = 768, 3072
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    u0 = (torch.randn(M, D, generator=g) * 2 + 0.3).to(dtype)          # un-normalised residual rows
    a = torch.randn(M, D, generator=g).to(dtype)
    w1 = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(dtype)
    b1 = torch.randn(D, generator=g)
    g0, be0 = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    g1, be1 = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    w2 = torch.randn(F, D, generator=g) / math.sqrt(D)
    b2 = torch.randn(F, generator=g)

    def stats(x, slab):      # (cols / slab, M, 2) partial sums of the stored values
        x = x.double().reshape(x.shape[0], -1, slab)
        return torch.stack([x.sum(-1), (x * x).sum(-1)], -1).transpose(0, 1).float().contiguous()

    def ln(x, gm, bt):
        x = x.double()
        mu = x.mean(-1, keepdim=True)
        var = (x * x).mean(-1, keepdim=True) - mu * mu
        return (x - mu) / torch.sqrt(var + 1e-5) * gm.double() + bt.double()

    # producer
    c1, st1 = o.gemm_ln(a.to(DEV), w1.to(DEV), b1.to(DEV), u0.to(DEV), r_stats=stats(u0, 32 if M >= 6500 else 64).to(DEV), r_gamma=g0.to(DEV),
                        r_beta=be0.to(DEV), stats_out=True)
    ref1 = a.double() @ w1.double().T + b1.double() + ln(u0, g0, be0)
    assert c1.dtype == dtype
    assert maxabs(c1.double().cpu().numpy(), ref1.numpy()) < tol * 4      # O(4) values
    # the statistics are those of the STORED rows (what the consumer will multiply), fp32 sums of a slab's values
    slab = D // st1.shape[0]
    assert slab == 64       # by N alone (round 6): a row's statistics do not depend on the row count of the launch that wrote them
    assert maxabs(st1.cpu().numpy(), stats(c1.cpu(), slab).numpy()) < 2e-3
    # consumer: GELU(LN(c1) W2^T + b2)
    wf, cs, bf = o.fold_layernorm(w2.to(DEV), b2.to(DEV), g1.to(DEV), be1.to(DEV), dtype)
    c2 = o.gemm_ln(c1, wf, bf, act=o.ACT_GELU, a_stats=st1, w_colsum=cs)
    z = ln(c1.cpu(), g1, be1) @ w2.double().T + b2.double()
    ref2 = 0.5 * z * (1 + torch.erf(z / math.sqrt(2)))
    assert maxabs(c2.double().cpu().numpy(), ref2.numpy()) < tol * 4
    # and against the unfused chain of the library itself (LayerNorm kernel -> 16-bit rows -> GEMM): same error class
    h = o.layernorm(c1, g1.to(DEV), be1.to(DEV))
    c2u = o.gemm(h, w2.to(DEV, dtype), b2.to(DEV), act=o.ACT_GELU)
    e_f, e_u = maxabs(c2.double().cpu().numpy(), ref2.numpy()), maxabs(c2u.double().cpu().numpy(), ref2.numpy())
    assert e_f < 2.0 * e_u + 1e-3, (e_f, e_u)
    # rejected shapes: N not a multiple of 128, fp32 operands
    with pytest.raises(Exception):
        o.gemm_ln(a.to(DEV), w1[:700].to(DEV).contiguous(), None)
    with pytest.raises(Exception):
        o.gemm_ln(a.to(DEV).float(), w1.to(DEV).float(), None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_prefetch_reads_the_ranges_and_changes_nothing(dtype):
    """msmd_attention_prefetch: the launch also pulls byte ranges (the next GEMMs' weights) through the memory-side cache.
    The attention output must be bit-identical to msmd_attention's, whatever the ranges' sizes (below one wave-load,
    not a multiple of 1 KB, larger than one grid sweep), and the ranges must come back untouched."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(21)
    for B, H, T in ((3, 12, 200), (2, 8, 111)):
        d = 64 * H
        qkv = torch.randn(B, T, 3 * d, generator=g).to(DEV, dtype)
        ref = o.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, 0.125)
        ranges = [torch.randn(n, generator=g).to(DEV) for n in (100, 1000, 70001, 3_000_000)]
        keep = [r.clone() for r in ranges]
        out = o.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, 0.125, prefetch=ranges)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        assert all(torch.equal(a, b) for a, b in zip(ranges, keep))
        # None entries and more than four tensors are tolerated (the first four live ones are used)
        out = o.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, 0.125, prefetch=[None] + ranges + ranges)
        assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_pre_and_folded_person_query(dtype):
    """msmd_layernorm_pre = the two launches LayerNorm -> (+ residual) -> LayerNorm it replaces (the inner result is rounded
    to the storage type as the first launch stored it: equal up to the rare last-bit flip where the compiler contracted
    the affine differently in the two kernels); msmd_person_query_attention_ln = the
    person query on LayerNorm'ed rows through folded weights, against the unfused pair within the 16-bit modes' error."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(31)
    N, T, d, H, Tk = 70, 111, 512, 8, 110
    u = (torch.randn(N, T, d, generator=g) * 1.7 + 0.2).to(DEV, dtype)
    R = torch.randn(N, T, d, generator=g).to(DEV, dtype)
    g1, b1 = (torch.rand(d, generator=g) + 0.5).to(DEV), (torch.randn(d, generator=g) * 0.1).to(DEV)
    g2, b2 = (torch.rand(d, generator=g) + 0.5).to(DEV), (torch.randn(d, generator=g) * 0.1).to(DEV)
    x1 = o.layernorm(u, g1, b1)
    ref = o.layernorm(x1, g2, b2, residual=R)
    got = o.layernorm_pre(u, g1, b1, R, g2, b2)
    diff = (got.float() - ref.float()).abs()
    assert float((diff == 0).float().mean()) > 0.99 and float(diff.max()) <= (6e-2 if dtype == torch.bfloat16 else 8e-3)
    wq = (torch.randn(d, d, generator=g) / math.sqrt(d)).to(DEV)
    bq = torch.randn(d, generator=g).to(DEV)
    kv = torch.randn(N, Tk, 2 * d, generator=g).to(DEV, dtype)
    a_ref = o.person_query_attention(x1, wq.to(dtype), bq, kv, H, 0.125)
    wf, cs, bf = o.fold_layernorm(wq, bq, g1, b1, dtype)
    a_got = o.person_query_attention(u, wf, bf, kv, H, 0.125, wq_colsum=cs)
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert maxabs(a_got.float().cpu().numpy(), a_ref.float().cpu().numpy()) < tol

