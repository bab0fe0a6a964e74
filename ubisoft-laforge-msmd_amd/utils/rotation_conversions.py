"""Rotation conversions on the MI355X (public names of reference utils/rotation_conversions.py:38-569).

Every function is one launch of csrc/rotations.hip's elementwise kernel; semantics follow the
reference (PyTorch3D): real-first quaternions, small-angle Taylor branch under 1e-6.
"""
from __future__ import annotations

import torch

from .. import ops

_OP = dict(q2m=0, m2q=1, aa2q=2, q2aa=3, aa2m=4, m2aa=5, d62m=6, m2d6=7, aa2d6=8, e2m=9, m2e=10, qstd=11, qinv=12,
           qraw=13, qmul=14, qapply=15)


def _conv_code(convention: str) -> int:
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    if convention[1] in (convention[0], convention[2]):
        raise ValueError(f"Invalid convention {convention}.")
    for letter in convention:
        if letter not in ("X", "Y", "Z"):
            raise ValueError(f"Invalid letter {letter} in convention string.")
    a = ["XYZ".index(c) for c in convention]
    return a[0] | (a[1] << 2) | (a[2] << 4)


def _check_mat(matrix):
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix  shape f{matrix.shape}.")


def quaternion_to_matrix(quaternions):
    return ops.rotation_convert(_OP["q2m"], quaternions, 4, (3, 3))


def matrix_to_quaternion(matrix):
    _check_mat(matrix)
    return ops.rotation_convert(_OP["m2q"], matrix, 9, (4,))


def euler_angles_to_matrix(euler_angles, convention: str):
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    return ops.rotation_convert(_OP["e2m"], euler_angles, 3, (3, 3), conv=_conv_code(convention))


def matrix_to_euler_angles(matrix, convention: str):
    code = _conv_code(convention)
    _check_mat(matrix)
    return ops.rotation_convert(_OP["m2e"], matrix, 9, (3,), conv=code)


def standardize_quaternion(quaternions):
    return ops.rotation_convert(_OP["qstd"], quaternions, 4, (4,))


def quaternion_raw_multiply(a, b):
    a, b = torch.broadcast_tensors(a, b)
    return ops.rotation_convert(_OP["qraw"], a, 4, (4,), x2=b)


def quaternion_multiply(a, b):
    a, b = torch.broadcast_tensors(a, b)
    return ops.rotation_convert(_OP["qmul"], a, 4, (4,), x2=b)


def quaternion_invert(quaternion):
    return ops.rotation_convert(_OP["qinv"], quaternion, 4, (4,))


def quaternion_apply(quaternion, point):
    if point.size(-1) != 3:
        raise ValueError(f"Points are not in 3D, f{point.shape}.")
    lead = torch.broadcast_shapes(quaternion.shape[:-1], point.shape[:-1])
    q = quaternion.expand(*lead, 4)
    p = point.expand(*lead, 3)
    return ops.rotation_convert(_OP["qapply"], q, 4, (3,), x2=p)


def axis_angle_to_matrix(axis_angle):
    return ops.rotation_convert(_OP["aa2m"], axis_angle, 3, (3, 3))


def matrix_to_axis_angle(matrix):
    _check_mat(matrix)
    return ops.rotation_convert(_OP["m2aa"], matrix, 9, (3,))


def axis_angle_to_quaternion(axis_angle):
    return ops.rotation_convert(_OP["aa2q"], axis_angle, 3, (4,))


def quaternion_to_axis_angle(quaternions):
    return ops.rotation_convert(_OP["q2aa"], quaternions, 4, (3,))


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    return ops.rotation_convert(_OP["d62m"], d6, 6, (3, 3))


def matrix_to_rotation_6d(matrix: torch.Tensor) -> torch.Tensor:
    return ops.rotation_convert(_OP["m2d6"], matrix, 9, (6,))


def axis_angle_to_rotation_6d(axis_angle):
    return ops.rotation_convert(_OP["aa2d6"], axis_angle, 3, (6,))


def random_quaternions(n: int, dtype=None, device=None, requires_grad=False):
    """reference utils/rotation_conversions.py:260-281 (host RNG + one normalisation)."""
    o = torch.randn((n, 4), dtype=dtype, device=device, requires_grad=requires_grad)
    s = (o * o).sum(1)
    signs = torch.where(o[:, 0] < 0, -torch.ones_like(s), torch.ones_like(s))
    return o / (signs * torch.sqrt(s))[:, None]


def random_rotations(n: int, dtype=None, device=None, requires_grad=False):
    return quaternion_to_matrix(random_quaternions(n, dtype=dtype, device=device, requires_grad=requires_grad))


def random_rotation(dtype=None, device=None, requires_grad=False):
    return random_rotations(1, dtype, device, requires_grad)[0]
