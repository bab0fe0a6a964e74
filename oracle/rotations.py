"""Oracle: rotation conversions (numpy fp32; test infrastructure).

Follows reference utils/rotation_conversions.py:38-569 (PyTorch3D semantics:
real-first quaternions, small-angle Taylor branch under 1e-6, _sqrt_positive_part,
_copysign).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _f(x):
    return np.asarray(x, dtype=F32)


def quaternion_to_matrix(q):
    q = _f(q)
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = F32(2.0) / (q * q).sum(-1, dtype=F32)
    o = np.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3)).astype(F32)


def _copysign(a, b):
    return np.where((a < 0) != (b < 0), -a, a)


def _sqrt_positive_part(x):
    return np.where(x > 0, np.sqrt(np.maximum(x, 0)), 0).astype(F32)


def matrix_to_quaternion(m):
    m = _f(m)
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    o0 = F32(0.5) * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = F32(0.5) * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = F32(0.5) * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = F32(0.5) * _sqrt_positive_part(1 - m00 - m11 + m22)
    o1 = _copysign(x, m[..., 2, 1] - m[..., 1, 2])
    o2 = _copysign(y, m[..., 0, 2] - m[..., 2, 0])
    o3 = _copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return np.stack([o0, o1, o2, o3], -1).astype(F32)


def _axis_angle_rotation(axis, angle):
    c, s = np.cos(angle).astype(F32), np.sin(angle).astype(F32)
    one, zero = np.ones_like(angle), np.zeros_like(angle)
    if axis == "X":
        R = (one, zero, zero, zero, c, -s, zero, s, c)
    elif axis == "Y":
        R = (c, zero, s, zero, one, zero, -s, zero, c)
    else:
        R = (c, -s, zero, s, c, zero, zero, zero, one)
    return np.stack(R, -1).reshape(angle.shape + (3, 3)).astype(F32)


def euler_angles_to_matrix(e, convention):
    e = _f(e)
    if e.ndim == 0 or e.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    _check_convention(convention)
    ms = [_axis_angle_rotation(c, e[..., i]) for i, c in enumerate(convention)]
    return np.matmul(np.matmul(ms[0], ms[1]), ms[2]).astype(F32)


def _check_convention(convention):
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    if convention[1] in (convention[0], convention[2]):
        raise ValueError(f"Invalid convention {convention}.")
    for letter in convention:
        if letter not in ("X", "Y", "Z"):
            raise ValueError(f"Invalid letter {letter} in convention string.")


def _angle_from_tan(axis, other_axis, data, horizontal, tait_bryan):
    i1, i2 = {"X": (2, 1), "Y": (0, 2), "Z": (1, 0)}[axis]
    if horizontal:
        i2, i1 = i1, i2
    even = (axis + other_axis) in ["XY", "YZ", "ZX"]
    if horizontal == even:
        return np.arctan2(data[..., i1], data[..., i2])
    if tait_bryan:
        return np.arctan2(-data[..., i2], data[..., i1])
    return np.arctan2(data[..., i2], -data[..., i1])


def matrix_to_euler_angles(m, convention):
    m = _f(m)
    _check_convention(convention)
    i0 = "XYZ".index(convention[0])
    i2 = "XYZ".index(convention[2])
    tait_bryan = i0 != i2
    if tait_bryan:
        central = np.arcsin(m[..., i0, i2] * (F32(-1.0) if i0 - i2 in [-1, 2] else F32(1.0)))
    else:
        central = np.arccos(m[..., i0, i0])
    o = (_angle_from_tan(convention[0], convention[1], m[..., i2], False, tait_bryan), central,
         _angle_from_tan(convention[2], convention[1], m[..., i0, :], True, tait_bryan))
    return np.stack(o, -1).astype(F32)


def standardize_quaternion(q):
    q = _f(q)
    return np.where(q[..., 0:1] < 0, -q, q).astype(F32)


def quaternion_raw_multiply(a, b):
    a, b = _f(a), _f(b)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1).astype(F32)


def quaternion_multiply(a, b):
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def quaternion_invert(q):
    return (_f(q) * np.array([1, -1, -1, -1], F32)).astype(F32)


def quaternion_apply(q, p):
    p = _f(p)
    if p.shape[-1] != 3:
        raise ValueError(f"Points are not in 3D, f{p.shape}.")
    pq = np.concatenate([np.zeros(p.shape[:-1] + (1,), F32), p], -1)
    return quaternion_raw_multiply(quaternion_raw_multiply(q, pq), quaternion_invert(q))[..., 1:]


def axis_angle_to_quaternion(aa):
    aa = _f(aa)
    angles = np.sqrt((aa * aa).sum(-1, keepdims=True, dtype=F32)).astype(F32)
    half = F32(0.5) * angles
    small = np.abs(angles) < F32(1e-6)
    with np.errstate(divide="ignore", invalid="ignore"):
        big = np.sin(half) / angles
    soa = np.where(small, F32(0.5) - (angles * angles) / F32(48), big).astype(F32)
    return np.concatenate([np.cos(half), aa * soa], -1).astype(F32)


def quaternion_to_axis_angle(q):
    q = _f(q)
    norms = np.sqrt((q[..., 1:] * q[..., 1:]).sum(-1, keepdims=True, dtype=F32)).astype(F32)
    half = np.arctan2(norms, q[..., :1]).astype(F32)
    angles = F32(2) * half
    small = np.abs(angles) < F32(1e-6)
    with np.errstate(divide="ignore", invalid="ignore"):
        big = np.sin(half) / angles
    soa = np.where(small, F32(0.5) - (angles * angles) / F32(48), big).astype(F32)
    return (q[..., 1:] / soa).astype(F32)


def axis_angle_to_matrix(aa):
    return quaternion_to_matrix(axis_angle_to_quaternion(aa))


def matrix_to_axis_angle(m):
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


def _normalize(x):
    n = np.sqrt((x * x).sum(-1, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(1e-12))).astype(F32)


def rotation_6d_to_matrix(d6):
    d6 = _f(d6)
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = _normalize(a1)
    b2 = _normalize(a2 - (b1 * a2).sum(-1, keepdims=True, dtype=F32) * b1)
    b3 = np.cross(b1, b2, axis=-1)
    return np.stack([b1, b2, b3], axis=-2).astype(F32)


def matrix_to_rotation_6d(m):
    m = _f(m)
    return m[..., :2, :].reshape(m.shape[:-2] + (6,)).copy()


def axis_angle_to_rotation_6d(aa):
    return matrix_to_rotation_6d(axis_angle_to_matrix(aa))
