"""Linear blend skinning on the MI355X (drop-in surface of reference utils/lbs.py).

``lbs(...)`` keeps the reference signature (utils/lbs.py:141) but runs as two HIP launches
(csrc/flame.hip): a per-frame kinematics kernel and one fused blendshape+skinning kernel; the
reference's (B, V, 4, 4) transforms and homogeneous coordinates are never materialised.
"""
from __future__ import annotations

import torch

from .. import ops

KP = 192  # 150 shape/expression coefficients + 36 pose-corrective features, padded to a multiple of 16


class LbsConstants:
    """Frame-invariant FLAME tensors repacked once for the kernels (plumbing, load time)."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
        dev = v_template.device
        V = v_template.shape[0]
        NB = shapedirs.shape[2]
        J = J_regressor.shape[0]
        assert NB + (J - 1) * 9 <= KP
        Vp = (V + 63) // 64 * 64
        self.V, self.Vp, self.NB, self.J = V, Vp, NB, J
        vt = v_template.float()
        sd = shapedirs.float()
        Jr = J_regressor.float()
        # joint regression table: rows [template ; each blendshape] -> (NB+1, J*3)
        js0 = (Jr @ vt).reshape(1, J * 3)
        jsl = torch.einsum("ji,ikl->ljk", Jr, sd).reshape(NB, J * 3)
        self.JS = torch.cat([js0, jsl], 0).contiguous()
        dirs = torch.zeros(3, KP, Vp, device=dev, dtype=torch.float32)
        dirs[:, :NB, :V] = sd.permute(1, 2, 0)                      # (3, NB, V)
        pd = posedirs.float().reshape(-1, V, 3)                      # (P, V, 3)
        dirs[:, NB:NB + pd.shape[0], :V] = pd.permute(2, 0, 1)
        self.dirs = dirs.contiguous()
        # split-bf16 form for the fast kernel: dirs ~= hi + lo, regrouped in 8-element K octets (2, 3, KP/8, Vp, 8)
        hi = dirs.to(torch.bfloat16)
        lo = (dirs - hi.float()).to(torch.bfloat16)
        oct_ = lambda t: t.reshape(3, KP // 8, 8, Vp).permute(0, 1, 3, 2)
        self.dirs_hl = torch.stack([oct_(hi), oct_(lo)], 0).contiguous()
        self.dirs_f16 = oct_(dirs.to(torch.float16)).contiguous()      # ONE fp16 plane: the operand of the fp16-vertex form (ops.lbs_skin_v2)
        tp = torch.zeros(3, Vp, device=dev, dtype=torch.float32)
        tp[:, :V] = vt.t()
        self.template_planes = tp.contiguous()
        wp = torch.zeros(J, Vp, device=dev, dtype=torch.float32)
        wp[:, :V] = lbs_weights.float().t()
        self.weight_planes = wp.contiguous()
        par = parents.to(torch.int32).clone()
        par[0] = -1
        self.parents = par.contiguous()
        # host copy for the python loops of the differentiable pass: reading the device tensor there would be a D2H
        # sync on every call, and a sync inside a stream capture (Trainer(use_graph=True) + use_vertex_space) kills it
        self.parents_host = [int(x) for x in par.cpu().tolist()]

    def dirs_t(self):
        """(KP, 3 * Vp) fp32: W operand of the backward contraction dcoef = dp . dirs^T (built on first use)."""
        if getattr(self, "_dirs_t", None) is None:
            self._dirs_t = self.dirs.permute(1, 0, 2).reshape(KP, 3 * self.Vp).contiguous()
        return self._dirs_t


_CACHE = {}


def _constants(v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    key = (v_template.data_ptr(), shapedirs.data_ptr(), posedirs.data_ptr(), lbs_weights.data_ptr())
    c = _CACHE.get(key)
    if c is None:
        vt = v_template[0] if v_template.dim() == 3 else v_template
        c = _CACHE[key] = LbsConstants(vt, shapedirs, posedirs, J_regressor, parents, lbs_weights)
    return c


# "bf16x3": split-bf16 MFMA products with fp32 accumulation (max-abs-err ~1e-6 on FLAME-scale vertices, 5x fewer
# matrix cycles), the joint blend as one fp16-split MFMA per transform component (msmd_lbs_skin_v2);
# "bf16x3_valu": the same blendshape product with the blend on the vector ALU (msmd_lbs_skin_bf16x3, the round-1 kernel);
# "fp32": exact-fp32 MFMA.  All far inside the 1e-4 budget of BASELINE.json.
DEFAULT_PRECISION = "bf16x3"


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot=True,
        dtype=torch.float32, constants: LbsConstants = None, precision: str = None):
    """reference utils/lbs.py:141-223.  Returns (verts (B, V, 3), posed joints (B, J, 3))."""
    if dtype != torch.float32:
        raise TypeError("lbs runs in fp32")
    c = constants or _constants(v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights)
    B = max(betas.shape[0], pose.shape[0])
    betas = betas.float().expand(B, -1).contiguous()
    pose = pose.float().reshape(pose.shape[0], -1).expand(B, -1).contiguous()
    precision = precision or DEFAULT_PRECISION
    if precision not in ("bf16x3", "bf16x3_valu", "fp32"):
        raise ValueError(f"Unknown LBS precision {precision}!")
    v2 = precision == "bf16x3" and c.J == 5
    split = precision != "fp32" and not v2
    res = ops.lbs_prepare(betas, pose, c.JS, c.parents, KP, pose_is_matrix=not pose2rot, want_split=split,
                          want_blend_tiles=v2)
    coef, coef_hl, A, joints = res[:4]
    if v2:
        verts = ops.lbs_skin_v2(res[4], B, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
    elif split:
        verts = ops.lbs_skin_bf16x3(coef_hl, A, c.template_planes, c.dirs_hl, c.weight_planes, c.V)
    else:
        verts = ops.lbs_skin(coef, A, c.template_planes, c.dirs, c.weight_planes, c.V)
    return verts, joints


# ----------------------------------------------------------------------------- differentiable pass (training)
def _rodrigues_torch(r):
    """reference utils/lbs.py:285-300 on (N, 3), autograd-differentiable (tiny tensors)."""
    angle = torch.norm(r + 1e-8, dim=1, keepdim=True)
    d = r / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = torch.zeros_like(rx)
    K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(-1, 3, 3)
    return torch.eye(3, device=r.device, dtype=r.dtype)[None] + sin * K + (1 - cos) * torch.bmm(K, K)


def kinematics_torch(c: LbsConstants, betas, pose):
    """What msmd_lbs_prepare computes -- coef (B, KP) = [betas | R[1:] - I | 0] and the relative rigid transforms
    A (B, J, 12) -- as autograd ops on (B, J, 3, 3)-sized tensors (utils/lbs.py:141-223, 317-371): the per-frame part
    of the differentiable FLAME pass.  The per-vertex part is SkinFn (HIP forward and backward)."""
    B, J = betas.shape[0], c.J
    joints = c.JS[0].view(1, J, 3) + (betas @ c.JS[1:]).view(B, J, 3)
    R = _rodrigues_torch(pose.reshape(B * J, 3)).view(B, J, 3, 3)
    eye = torch.eye(3, device=R.device, dtype=R.dtype)
    coef = torch.cat([betas, (R[:, 1:] - eye).reshape(B, (J - 1) * 9),
                      betas.new_zeros(B, KP - betas.shape[1] - (J - 1) * 9)], dim=1)
    par = c.parents_host
    wR, wt = [R[:, 0]], [joints[:, 0]]
    for i in range(1, J):
        pa = par[i]
        wR.append(torch.bmm(wR[pa], R[:, i]))
        wt.append(torch.bmm(wR[pa], (joints[:, i] - joints[:, pa]).unsqueeze(-1)).squeeze(-1) + wt[pa])
    A = []
    for i in range(J):
        t = wt[i] - torch.bmm(wR[i], joints[:, i].unsqueeze(-1)).squeeze(-1)
        A.append(torch.cat([wR[i], t.unsqueeze(-1)], dim=-1).reshape(B, 12))
    return coef, torch.stack(A, dim=1)


class SkinFn(torch.autograd.Function):
    """verts (B, V, 3) = sum_j w_j A_j [template + coef . dirs ; 1] with the HIP forward (msmd_lbs_skin_v2_train) and
    backward (msmd_lbs_skin_bwd + one GEMM for dcoef): the reference differentiates utils/lbs.py:185-221 by autograd
    through its (B, V, 4, 4) intermediates."""

    @staticmethod
    def forward(ctx, coef, A, c):
        coef, A = coef.float().contiguous(), A.float().contiguous()
        tiles = ops.lbs_pack(coef, A)
        verts, vposed = ops.lbs_skin_v2_train(tiles, coef.shape[0], c.template_planes, c.dirs_hl, c.weight_planes, c.V)
        ctx.save_for_backward(vposed, A)
        ctx.c = c
        return verts

    @staticmethod
    def backward(ctx, g):
        vposed, A = ctx.saved_tensors
        c = ctx.c
        dp, dA = ops.lbs_skin_bwd(g.float().contiguous(), vposed, A, c.weight_planes)
        dcoef = ops.gemm(dp.view(dp.shape[0], 3 * c.Vp), c.dirs_t())
        return dcoef, dA, None


def lbs_train(betas, pose, constants: LbsConstants):
    """Differentiable lbs(): (betas (B, NB), axis-angle pose (B, J*3)) -> verts (B, V, 3) with gradients to both."""
    coef, A = kinematics_torch(constants, betas.float(), pose.float())
    return SkinFn.apply(coef, A, constants)


def batch_rodrigues(rot_vecs, epsilon=1e-8, dtype=torch.float32):
    """reference utils/lbs.py:270-301."""
    return ops.batch_rodrigues(rot_vecs.float().contiguous())


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary_coords):
    """reference utils/lbs.py:102-138."""
    return ops.landmarks(vertices.float().contiguous(), faces.to(torch.int32).contiguous(),
                         lmk_faces_idx.to(torch.int32).contiguous(), lmk_bary_coords.float().contiguous())


def rot_mat_to_euler(rot_mats):
    """reference utils/lbs.py:26-32 (host-side helper on tiny tensors; the LUT path uses the fused kernel)."""
    sy = torch.sqrt(rot_mats[:, 0, 0] * rot_mats[:, 0, 0] + rot_mats[:, 1, 0] * rot_mats[:, 1, 0])
    return torch.atan2(-rot_mats[:, 2, 0], sy)


# ----------------------------------------------------------------------------- the reference's building blocks
# `lbs()` above never calls these (its two kernels fuse them); they keep the reference's names and results for code
# that uses the pieces directly.  The two contractions over vertices / coefficients run on the exact-fp32 MFMA GEMM;
# the 4x4 joint chain is a few hundred bytes per frame of host-library glue.
def _pad4(x):
    k = x.shape[-1]
    return x if k % 4 == 0 else ops.pad_cols(x.contiguous(), (k + 3) // 4 * 4)


def blend_shapes(betas, shape_disps):
    """(B, L) coefficients x (V, 3, L) directions -> (B, V, 3) displacements (reference utils/lbs.py:246-267)."""
    V, _, L = shape_disps.shape
    out = ops.gemm(_pad4(betas.float().contiguous()), _pad4(shape_disps.float().reshape(V * 3, L).contiguous()))
    return out.view(betas.shape[0], V, 3)


def vertices2joints(J_regressor, vertices):
    """(J, V) regressor applied to (B, V, 3) vertices -> (B, J, 3) joints (reference utils/lbs.py:226-243)."""
    B, V, _ = vertices.shape
    vt = vertices.float().transpose(1, 2).contiguous().view(B * 3, V)        # rows (b, xyz), V contiguous
    out = ops.gemm(_pad4(vt), _pad4(J_regressor.float().contiguous()))       # (B*3, J)
    return out.view(B, 3, -1).transpose(1, 2).contiguous()


def transform_mat(R, t):
    """(B, 3, 3) rotations + (B, 3, 1) translations -> (B, 4, 4) homogeneous transforms (utils/lbs.py:304-314)."""
    T = torch.zeros(R.shape[0], 4, 4, device=R.device, dtype=R.dtype)
    T[:, :3, :3] = R
    T[:, :3, 3:] = t
    T[:, 3, 3] = 1
    return T


def batch_rigid_transform(rot_mats, joints, parents, dtype=torch.float32):
    """Pose the kinematic tree (reference utils/lbs.py:317-375): returns the posed joints (B, N, 3) and each joint's
    transform relative to its rest position (B, N, 4, 4)."""
    B, N = joints.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    local = transform_mat(rot_mats.reshape(-1, 3, 3), rel.reshape(-1, 3, 1)).view(B, N, 4, 4)
    world = [local[:, 0]]
    for i in range(1, N):
        world.append(torch.matmul(world[int(parents[i])], local[:, i]))
    world = torch.stack(world, dim=1)
    posed = world[:, :, :3, 3]
    # subtract the rest-pose joint carried through the rotation: only the translation column changes
    rest = torch.matmul(world[:, :, :, :3], joints.unsqueeze(-1))            # (B, N, 4, 1); row 3 is zero
    rel_t = world.clone()
    rel_t[:, :, :, 3:] = world[:, :, :, 3:] - rest
    return posed, rel_t


def find_dynamic_lmk_idx_and_bcoords(vertices, pose, dynamic_lmk_faces_idx, dynamic_lmk_b_coords, neck_kin_chain,
                                     dtype=torch.float32):
    """Contour-landmark faces / barycentric weights for the current neck yaw (reference utils/lbs.py:35-99): the
    yaw -> table-row lookup is msmd_dynamic_lmk_row.  This module-level function looks the table up at MINUS the yaw
    (utils/lbs.py:86-87), unlike FLAME's own method (utils/flame.py:158-160, what the model uses): the kernel's row
    for +yaw is mirrored here (0 -> 0, r <= 39 -> r + 39, r > 39 -> r - 39; exact, clamps included)."""
    B = vertices.shape[0]
    chain = torch.as_tensor(neck_kin_chain, device=pose.device).to(torch.int32).contiguous()
    row = ops.dynamic_lmk_row(pose.reshape(B, -1).float().contiguous(), chain).long()
    row = torch.where(row == 0, row, torch.where(row <= 39, row + 39, row - 39))
    return dynamic_lmk_faces_idx[row], dynamic_lmk_b_coords[row]
