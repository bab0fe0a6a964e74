"""Oracle: FLAME blendshapes + linear blend skinning + landmarks (numpy fp32; test infrastructure).

Follows reference utils/lbs.py:26-371 and utils/flame.py:59-244.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def batch_rodrigues(rot_vecs):
    """reference utils/lbs.py:270-301: angle = ||r + 1e-8||, dir = r / angle (un-shifted r),
    R = I + sin*K + (1-cos)*K@K."""
    r = np.asarray(rot_vecs, dtype=F32)
    angle = np.sqrt(((r + F32(1e-8)) ** 2).sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    d = (r / angle).astype(F32)
    cos = np.cos(angle)[:, None].astype(F32)
    sin = np.sin(angle)[:, None].astype(F32)
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = np.zeros_like(rx)
    K = np.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], axis=1).reshape(-1, 3, 3).astype(F32)
    ident = np.eye(3, dtype=F32)[None]
    return (ident + sin * K + (F32(1) - cos) * np.matmul(K, K)).astype(F32)


def transform_mat(R, t):
    """reference utils/lbs.py:304-314."""
    B = R.shape[0]
    T = np.zeros((B, 4, 4), dtype=F32)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    T[:, 3, 3] = 1
    return T


def batch_rigid_transform(rot_mats, joints, parents):
    """reference utils/lbs.py:317-371."""
    B, J = joints.shape[:2]
    rel = joints.copy()
    rel[:, 1:] -= joints[:, parents[1:]]
    tm = transform_mat(rot_mats.reshape(-1, 3, 3), rel.reshape(-1, 3)).reshape(B, J, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, J):
        chain.append(np.matmul(chain[parents[i]], tm[:, i]).astype(F32))
    transforms = np.stack(chain, axis=1)
    posed = transforms[:, :, :3, 3]
    jh = np.concatenate([joints, np.zeros((B, J, 1), F32)], axis=2)[..., None]  # (B,J,4,1)
    tj = np.matmul(transforms, jh)  # (B,J,4,1)
    rel_t = transforms.copy()
    rel_t[:, :, :, 3:4] -= tj
    return posed, rel_t.astype(F32)


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """reference utils/lbs.py:141-223 (pose2rot=True).
    betas (B,150), pose (B,15), shapedirs (V,3,150), posedirs (36, V*3)."""
    betas = np.asarray(betas, F32)
    B = betas.shape[0]
    V = v_template.shape[0]
    v_shaped = v_template[None] + np.einsum("bl,mkl->bmk", betas, shapedirs).astype(F32)
    J = np.einsum("bik,ji->bjk", v_shaped, J_regressor).astype(F32)
    rot = batch_rodrigues(np.asarray(pose, F32).reshape(-1, 3)).reshape(B, -1, 3, 3)
    pose_feature = (rot[:, 1:] - np.eye(3, dtype=F32)).reshape(B, -1)
    v_posed = (np.matmul(pose_feature, posedirs).reshape(B, V, 3) + v_shaped).astype(F32)
    J_t, A = batch_rigid_transform(rot, J, parents)
    T = np.matmul(lbs_weights[None], A.reshape(B, -1, 16)).reshape(B, V, 4, 4).astype(F32)
    vh = np.concatenate([v_posed, np.ones((B, V, 1), F32)], axis=2)[..., None]
    verts = np.matmul(T, vh)[:, :, :3, 0].astype(F32)
    return verts, J_t


def rot_mat_to_euler(R):
    """reference utils/lbs.py:26-32."""
    sy = np.sqrt(R[:, 0, 0] * R[:, 0, 0] + R[:, 1, 0] * R[:, 1, 0])
    return np.arctan2(-R[:, 2, 0], sy).astype(F32)


def dynamic_lmk_index(full_pose, neck_kin_chain):
    """LUT row of reference utils/flame.py:126-172 (pose2rot=True).  Bit-exact integer output."""
    B = full_pose.shape[0]
    aa = np.asarray(full_pose, F32).reshape(B, -1, 3)[:, neck_kin_chain]
    rot = batch_rodrigues(aa.reshape(-1, 3)).reshape(B, -1, 3, 3)
    rel = np.broadcast_to(np.eye(3, dtype=F32), (B, 3, 3)).copy()
    for i in range(len(neck_kin_chain)):
        rel = np.matmul(rot[:, i], rel).astype(F32)
    ang = (rot_mat_to_euler(rel) * F32(180.0) / F32(np.pi)).astype(F32)
    y = np.round(np.minimum(ang, F32(39))).astype(np.int64)  # torch.round = half-to-even = np.round
    neg = (y < 0).astype(np.int64)
    mask = (y < -39).astype(np.int64)
    neg_vals = mask * 78 + (1 - mask) * (39 - y)
    return neg * neg_vals + (1 - neg) * y


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary):
    """reference utils/lbs.py:102-138.  lmk_faces_idx (B,L) int, lmk_bary (B,L,3)."""
    B = vertices.shape[0]
    lf = faces[lmk_faces_idx.reshape(-1)].reshape(B, -1, 3)
    lv = vertices[np.arange(B)[:, None, None], lf]  # (B,L,3,3)
    return np.einsum("blfi,blf->bli", lv, lmk_bary).astype(F32)


class FlameOracle:
    """Buffers of reference utils/flame.py:66-124 built from the synthetic asset dict."""

    def __init__(self, asset, n_shape=100, n_exp=50):
        self.faces = asset["f"].astype(np.int64)
        self.v_template = asset["v_template"].astype(F32)
        sd = asset["shapedirs"].astype(F32)
        self.shapedirs = np.concatenate([sd[:, :, :n_shape], sd[:, :, 300:300 + n_exp]], axis=2)
        npb = asset["posedirs"].shape[-1]
        self.posedirs = np.reshape(asset["posedirs"], [-1, npb]).T.astype(F32).copy()
        self.J_regressor = asset["J_regressor"].astype(F32)
        parents = asset["kintree_table"][0].astype(np.int64).copy()
        parents[0] = -1
        self.parents = parents
        self.lbs_weights = asset["weights"].astype(F32)
        lmk = asset["lmk"]
        self.lmk_faces_idx = lmk["static_lmk_faces_idx"].astype(np.int64)
        self.lmk_bary = lmk["static_lmk_bary_coords"].astype(F32)
        self.dyn_faces_idx = lmk["dynamic_lmk_faces_idx"].astype(np.int64)
        self.dyn_bary = lmk["dynamic_lmk_bary_coords"].astype(F32)
        self.full_faces_idx = lmk["full_lmk_faces_idx"].astype(np.int64)
        self.full_bary = lmk["full_lmk_bary_coords"].astype(F32)
        chain, cur = [], 1
        while cur != -1:
            chain.append(cur)
            cur = int(parents[cur])
        self.neck_kin_chain = np.array(chain, dtype=np.int64)

    def full_pose(self, pose_params, ignore_global_rot=False):
        """reference utils/flame.py:196-203: [global3 | neck3 = 0 | jaw3 | eyes6 = 0]."""
        B = pose_params.shape[0]
        head = np.zeros((B, 3), F32) if ignore_global_rot else pose_params[:, :3]
        return np.concatenate([head, np.zeros((B, 3), F32), pose_params[:, 3:], np.zeros((B, 6), F32)],
                              axis=1).astype(F32)

    def forward(self, shape_params, expression_params, pose_params, ignore_global_rot=False,
                return_lm2d=True, return_lm3d=True):
        """reference utils/flame.py:180-244 (pose2rot=True)."""
        B = shape_params.shape[0]
        betas = np.concatenate([shape_params, expression_params], axis=1).astype(F32)
        fp = self.full_pose(np.asarray(pose_params, F32), ignore_global_rot)
        verts, _ = lbs(betas, fp, self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
                       self.parents, self.lbs_weights)
        lm2d = lm3d = None
        if return_lm2d:
            row = dynamic_lmk_index(fp, self.neck_kin_chain)
            fidx = np.concatenate([self.dyn_faces_idx[row], np.broadcast_to(self.lmk_faces_idx, (B, 51))], axis=1)
            bary = np.concatenate([self.dyn_bary[row], np.broadcast_to(self.lmk_bary, (B, 51, 3))], axis=1)
            lm2d = vertices2landmarks(verts, self.faces, fidx, bary)
        if return_lm3d:
            lm3d = vertices2landmarks(verts, self.faces, np.tile(self.full_faces_idx, (B, 1)),
                                      np.tile(self.full_bary, (B, 1, 1)))
        return verts, lm2d, lm3d
