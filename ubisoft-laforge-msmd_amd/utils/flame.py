"""FLAME head model on the MI355X (drop-in surface of reference utils/flame.py:59-244)."""
from __future__ import annotations

import pickle
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from . import lbs as _lbs

FLAMEConfig = SimpleNamespace(
    flame_model_path="/code/models/flame_data/FLAME2020/generic_model.pkl",
    n_shape=100, n_exp=50, n_tex=50, tex_type="BFM",
    tex_path="/code/models/flame_data/FLAME2020/FLAME_albedo_from_BFM.npz",
    flame_lmk_embedding_path="/code/models/flame_data/landmark_embedding.npy",
)


def _np(a, dtype=np.float32):
    if "scipy.sparse" in str(type(a)):
        a = a.todense()
    return np.array(a, dtype=dtype)


# the reference module's loader helpers, under its names (utils/flame.py:28-42)
to_np = _np


def to_tensor(array, dtype=torch.float32):
    """Array-like -> tensor (reference l.28-30; its lower-case type-name test never matches "torch.Tensor", so
    tensors are copied like any other input -- kept)."""
    return None if "torch.tensor" in str(type(array)) else torch.tensor(array, dtype=dtype)


class Struct(object):
    """Attribute bag over the pickle's dict (reference l.39-42)."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)


class FLAME(nn.Module):
    """Same buffers, constructor argument and forward signature as the reference class.
    ``config`` may also carry ``asset`` (a dict with the pickle's arrays + 'lmk') so tests can
    build the module from the synthetic asset without touching the filesystem."""

    def __init__(self, config):
        super().__init__()
        asset = getattr(config, "asset", None)
        if asset is None:
            with open(config.flame_model_path, "rb") as f:
                asset = dict(pickle.load(f, encoding="latin1"))
            lmk = np.load(config.flame_lmk_embedding_path, allow_pickle=True, encoding="latin1")[()]
        else:
            lmk = asset["lmk"]
        self.dtype = torch.float32
        self.register_buffer("faces_tensor", torch.tensor(_np(asset["f"], np.int64), dtype=torch.long))
        self.register_buffer("v_template", torch.tensor(_np(asset["v_template"])))
        sd = torch.tensor(_np(asset["shapedirs"]))
        sd = torch.cat([sd[:, :, :config.n_shape], sd[:, :, 300:300 + config.n_exp]], 2)
        self.register_buffer("shapedirs", sd)
        npb = asset["posedirs"].shape[-1]
        self.register_buffer("posedirs", torch.tensor(_np(np.reshape(asset["posedirs"], [-1, npb]).T)))
        self.register_buffer("J_regressor", torch.tensor(_np(asset["J_regressor"])))
        parents = torch.tensor(_np(asset["kintree_table"][0], np.int64)).long()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("lbs_weights", torch.tensor(_np(asset["weights"])))
        self.register_parameter("eye_pose", nn.Parameter(torch.zeros(1, 6), requires_grad=False))
        self.register_parameter("eye_pose_mat", nn.Parameter(torch.eye(3).view(1, 9).repeat(1, 2), requires_grad=False))
        self.register_parameter("neck_pose", nn.Parameter(torch.zeros(1, 3), requires_grad=False))
        self.register_parameter("neck_pose_mat", nn.Parameter(torch.eye(3).view(1, 9), requires_grad=False))

        def _t(x, dt):
            return (x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x))).to(dt)
        self.register_buffer("lmk_faces_idx", _t(lmk["static_lmk_faces_idx"], torch.long))
        self.register_buffer("lmk_bary_coords", _t(lmk["static_lmk_bary_coords"], torch.float32))
        self.register_buffer("dynamic_lmk_faces_idx", _t(lmk["dynamic_lmk_faces_idx"], torch.long))
        self.register_buffer("dynamic_lmk_bary_coords", _t(lmk["dynamic_lmk_bary_coords"], torch.float32))
        self.register_buffer("full_lmk_faces_idx", _t(lmk["full_lmk_faces_idx"], torch.long))
        self.register_buffer("full_lmk_bary_coords", _t(lmk["full_lmk_bary_coords"], torch.float32))
        chain, cur = [], 1
        while cur != -1:
            chain.append(cur)
            cur = int(parents[cur])
        self.register_buffer("neck_kin_chain", torch.tensor(chain, dtype=torch.long))
        self._packed = None
        self.lbs_precision = getattr(config, "lbs_precision", None)  # None -> utils.lbs.DEFAULT_PRECISION

    def _pack(self):
        if self._packed is None or self._packed["dev"] != self.v_template.device:
            self._packed = dict(
                dev=self.v_template.device,
                lbs=_lbs.LbsConstants(self.v_template, self.shapedirs, self.posedirs, self.J_regressor, self.parents,
                                      self.lbs_weights),
                faces=self.faces_tensor.to(torch.int32).contiguous(),
                chain=self.neck_kin_chain.to(torch.int32).contiguous(),
                static_idx=self.lmk_faces_idx.to(torch.int32).contiguous(),
                dyn_idx=self.dynamic_lmk_faces_idx.to(torch.int32).contiguous(),
                full_idx=self.full_lmk_faces_idx.to(torch.int32).contiguous(),
            )
        return self._packed

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _find_dynamic_lmk_idx_and_bcoords(self, pose, dynamic_lmk_faces_idx, dynamic_lmk_b_coords, neck_kin_chain,
                                          dtype=torch.float32, pose2rot=True):
        """reference utils/flame.py:126-172: the contour's face ids and barycentric weights for each frame's yaw -- the
        neck chain's relative rotation, its yaw in degrees rounded, clamped to <= 39 and negative angles folded to 39..78,
        is the row of both tables (one kernel: msmd_dynamic_lmk_row; forward() above uses the same launch)."""
        chain = torch.as_tensor(neck_kin_chain, device=pose.device).to(torch.int32).contiguous()
        row = ops.dynamic_lmk_row(pose.float().contiguous(), chain, pose_is_matrix=not pose2rot).long()
        return torch.index_select(dynamic_lmk_faces_idx, 0, row), torch.index_select(dynamic_lmk_b_coords, 0, row)

    def seletec_3d68(self, vertices):
        p = self._pack()
        return ops.landmarks(vertices.float().contiguous(), p["faces"], p["full_idx"], self.full_lmk_bary_coords)

    def forward(self, shape_params=None, expression_params=None, pose_params=None, eye_pose_params=None,
                pose2rot=True, ignore_global_rot=False, return_lm2d=True, return_lm3d=True):
        """reference utils/flame.py:180-244."""
        p = self._pack()
        B = shape_params.shape[0]
        c = p["lbs"]
        needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in
                                                     (shape_params, expression_params, pose_params, eye_pose_params))
        if (pose2rot and not return_lm2d and pose_params is not None and not needs_grad
                and (self.lbs_precision or _lbs.DEFAULT_PRECISION) == "bf16x3" and c.J == 5 and pose_params.shape[1] == 6 and shape_params.shape[1] + expression_params.shape[1] + 36 <= 192
                and shape_params.is_cuda and shape_params.shape[0] == expression_params.shape[0] == pose_params.shape[0]):
            # inference fast path: the kinematics kernel reads shape / expression / pose where they are (no concatenated
            # betas / full_pose copies) and writes the skinning kernel's tile records directly
            f32c = lambda t: None if t is None else t.float().contiguous()
            tiles, varies, folded = ops.flame_prepare(f32c(shape_params), f32c(expression_params), f32c(pose_params),
                                                      f32c(eye_pose_params), c.JS, c.parents, ignore_global_rot, c.dirs,
                                                      c.template_planes)
            # vertex_dtype = torch.float16 (opt-in attribute; BASELINE configs[4]'s fp16 LBS pass): fp16 vertices, half the stores
            vd = getattr(self, "vertex_dtype", torch.float32)
            # fp16 vertices: on fp16 operand planes by default (vertex_exact = True: the fp32 kernel's arithmetic, rounded once)
            vertices = ops.lbs_skin_v2(tiles, B, c.template_planes, c.dirs_hl, c.weight_planes, c.V, shape_varies=varies,
                                       folded=folded, out_dtype=vd,
                                       dirs_f16=c.dirs_f16 if (vd == torch.float16 and not getattr(self, "vertex_exact", False)) else None)
            landmarks3d = None
            if return_lm3d:
                landmarks3d = ops.landmarks(vertices if vd == torch.float32 else vertices.float().contiguous(), p["faces"],
                                            p["full_idx"], self.full_lmk_bary_coords)
            return vertices, None, landmarks3d
        betas = torch.cat([shape_params, expression_params], dim=1).float().contiguous()
        if pose2rot:
            if pose_params is None:
                pose_params = self.eye_pose.expand(B, -1)
            if eye_pose_params is None:
                eye_pose_params = self.eye_pose.expand(B, -1)
            head = pose_params[:, :3] if not ignore_global_rot else torch.zeros_like(pose_params[:, :3])
            full_pose = torch.cat([head, self.neck_pose.expand(B, -1), pose_params[:, 3:], eye_pose_params], dim=1)
        else:
            if pose_params is None:
                pose_params = self.eye_pose_mat.expand(B, -1)
            if eye_pose_params is None:
                eye_pose_params = self.eye_pose_mat.expand(B, -1)
            head = pose_params[:, :9] if not ignore_global_rot else self.eye_pose_mat.expand(B, -1)[:, :9]
            full_pose = torch.cat([head, self.neck_pose_mat.expand(B, -1), pose_params[:, 9:], eye_pose_params], dim=1)
        full_pose = full_pose.float().contiguous()
        if torch.is_grad_enabled() and (betas.requires_grad or full_pose.requires_grad):
            # training through the vertex-space loss (reference training_script.py:167-176): differentiable pass --
            # per-frame kinematics by autograd on tiny tensors, per-vertex skinning forward / backward in HIP
            if not pose2rot:
                raise NotImplementedError("the differentiable FLAME pass takes axis-angle poses (pose2rot=True)")
            vertices = _lbs.lbs_train(betas, full_pose, p["lbs"])
        else:
            vertices, _ = _lbs.lbs(betas, full_pose, self.v_template, self.shapedirs, self.posedirs, self.J_regressor,
                                   self.parents, self.lbs_weights, pose2rot, self.dtype, constants=p["lbs"],
                                   precision=self.lbs_precision)
        landmarks2d = landmarks3d = None
        if return_lm2d:
            row = ops.dynamic_lmk_row(full_pose, p["chain"], pose_is_matrix=not pose2rot).long()
            idx = torch.cat([p["dyn_idx"][row], p["static_idx"].unsqueeze(0).expand(B, -1)], 1).contiguous()
            bary = torch.cat([self.dynamic_lmk_bary_coords[row], self.lmk_bary_coords.unsqueeze(0).expand(B, -1, -1)],
                             1).contiguous()
            landmarks2d = ops.landmarks(vertices, p["faces"], idx, bary)
        if return_lm3d:
            landmarks3d = ops.landmarks(vertices, p["faces"], p["full_idx"], self.full_lmk_bary_coords)
        return vertices, landmarks2d, landmarks3d
