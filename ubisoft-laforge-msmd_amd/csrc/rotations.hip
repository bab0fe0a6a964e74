// Rotation conversions (reference utils/rotation_conversions.py:38-569; PyTorch3D semantics):
// real-first quaternions, small-angle Taylor branch below 1e-6, _sqrt_positive_part, _copysign.
// Elementwise, one item per thread; HBM-bound.
#include "common.h"

__device__ __forceinline__ void quat_to_mat(const float* q, float* o) {
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  o[0] = 1 - two_s * (j * j + k * k); o[1] = two_s * (i * j - k * r); o[2] = two_s * (i * k + j * r);
  o[3] = two_s * (i * j + k * r); o[4] = 1 - two_s * (i * i + k * k); o[5] = two_s * (j * k - i * r);
  o[6] = two_s * (i * k - j * r); o[7] = two_s * (j * k + i * r); o[8] = 1 - two_s * (i * i + j * j);
}
__device__ __forceinline__ float sqrt_pos(float x) { return x > 0.f ? sqrtf(x) : 0.f; }
__device__ __forceinline__ float copysign_like(float a, float b) { return ((a < 0.f) != (b < 0.f)) ? -a : a; }

__device__ __forceinline__ void mat_to_quat(const float* m, float* o) {
  const float m00 = m[0], m11 = m[4], m22 = m[8];
  o[0] = 0.5f * sqrt_pos(1 + m00 + m11 + m22);
  const float x = 0.5f * sqrt_pos(1 + m00 - m11 - m22);
  const float y = 0.5f * sqrt_pos(1 - m00 + m11 - m22);
  const float z = 0.5f * sqrt_pos(1 - m00 - m11 + m22);
  o[1] = copysign_like(x, m[7] - m[5]);
  o[2] = copysign_like(y, m[2] - m[6]);
  o[3] = copysign_like(z, m[3] - m[1]);
}
__device__ __forceinline__ void aa_to_quat(const float* a, float* o) {
  const float angle = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const float half = 0.5f * angle;
  const float soa = (fabsf(angle) < 1e-6f) ? (0.5f - (angle * angle) / 48.0f) : (sinf(half) / angle);
  o[0] = cosf(half); o[1] = a[0] * soa; o[2] = a[1] * soa; o[3] = a[2] * soa;
}
__device__ __forceinline__ void quat_to_aa(const float* q, float* o) {
  const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float half = atan2f(n, q[0]);
  const float angle = 2.0f * half;
  const float soa = (fabsf(angle) < 1e-6f) ? (0.5f - (angle * angle) / 48.0f) : (sinf(half) / angle);
  o[0] = q[1] / soa; o[1] = q[2] / soa; o[2] = q[3] / soa;
}
__device__ __forceinline__ void quat_raw_mul(const float* a, const float* b, float* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
__device__ __forceinline__ void normalize3(const float* a, float* o) {
  const float n = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-12f);  // F.normalize eps
  o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}
__device__ __forceinline__ void axis_rot(int axis, float ang, float* R) {
  const float c = cosf(ang), s = sinf(ang);
  if (axis == 0) { R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = c; R[5] = -s; R[6] = 0; R[7] = s; R[8] = c; }
  else if (axis == 1) { R[0] = c; R[1] = 0; R[2] = s; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = -s; R[7] = 0; R[8] = c; }
  else { R[0] = c; R[1] = -s; R[2] = 0; R[3] = s; R[4] = c; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1; }
}
__device__ __forceinline__ void mat3mul(const float* a, const float* b, float* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
// _angle_from_tan(axis, other_axis, data(3 values), horizontal, tait_bryan)
__device__ __forceinline__ float angle_from_tan(int axis, int other, const float* data, bool horizontal, bool tb) {
  int i1 = axis == 0 ? 2 : (axis == 1 ? 0 : 1);
  int i2 = axis == 0 ? 1 : (axis == 1 ? 2 : 0);
  if (horizontal) { const int t = i1; i1 = i2; i2 = t; }
  const bool even = (axis == 0 && other == 1) || (axis == 1 && other == 2) || (axis == 2 && other == 0);
  if (horizontal == even) return atan2f(data[i1], data[i2]);
  if (tb) return atan2f(-data[i2], data[i1]);
  return atan2f(data[i2], -data[i1]);
}

__global__ void rotation_kernel(int op, const float* __restrict__ in, const float* __restrict__ in2,
                                float* __restrict__ out, long n, int conv) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= n) return;
  float a[9], b[4], o[9], q[4];
  const int c0 = conv & 3, c1 = (conv >> 2) & 3, c2 = (conv >> 4) & 3;
  switch (op) {
    case MSMD_ROT_QUAT_TO_MAT:
      for (int k = 0; k < 4; ++k) a[k] = in[t * 4 + k];
      quat_to_mat(a, o);
      for (int k = 0; k < 9; ++k) out[t * 9 + k] = o[k];
      break;
    case MSMD_ROT_MAT_TO_QUAT:
      for (int k = 0; k < 9; ++k) a[k] = in[t * 9 + k];
      mat_to_quat(a, o);
      for (int k = 0; k < 4; ++k) out[t * 4 + k] = o[k];
      break;
    case MSMD_ROT_AA_TO_QUAT:
      for (int k = 0; k < 3; ++k) a[k] = in[t * 3 + k];
      aa_to_quat(a, o);
      for (int k = 0; k < 4; ++k) out[t * 4 + k] = o[k];
      break;
    case MSMD_ROT_QUAT_TO_AA:
      for (int k = 0; k < 4; ++k) a[k] = in[t * 4 + k];
      quat_to_aa(a, o);
      for (int k = 0; k < 3; ++k) out[t * 3 + k] = o[k];
      break;
    case MSMD_ROT_AA_TO_MAT:
    case MSMD_ROT_AA_TO_6D:
      for (int k = 0; k < 3; ++k) a[k] = in[t * 3 + k];
      aa_to_quat(a, q);
      quat_to_mat(q, o);
      if (op == MSMD_ROT_AA_TO_MAT) for (int k = 0; k < 9; ++k) out[t * 9 + k] = o[k];
      else for (int k = 0; k < 6; ++k) out[t * 6 + k] = o[k];
      break;
    case MSMD_ROT_MAT_TO_AA:
      for (int k = 0; k < 9; ++k) a[k] = in[t * 9 + k];
      mat_to_quat(a, q);
      quat_to_aa(q, o);
      for (int k = 0; k < 3; ++k) out[t * 3 + k] = o[k];
      break;
    case MSMD_ROT_6D_TO_MAT: {
      for (int k = 0; k < 6; ++k) a[k] = in[t * 6 + k];
      float b1[3], b2[3], tmp[3];
      normalize3(a, b1);
      const float dot = b1[0] * a[3] + b1[1] * a[4] + b1[2] * a[5];
      for (int k = 0; k < 3; ++k) tmp[k] = a[3 + k] - dot * b1[k];
      normalize3(tmp, b2);
      o[0] = b1[0]; o[1] = b1[1]; o[2] = b1[2]; o[3] = b2[0]; o[4] = b2[1]; o[5] = b2[2];
      o[6] = b1[1] * b2[2] - b1[2] * b2[1]; o[7] = b1[2] * b2[0] - b1[0] * b2[2]; o[8] = b1[0] * b2[1] - b1[1] * b2[0];
      for (int k = 0; k < 9; ++k) out[t * 9 + k] = o[k];
      break;
    }
    case MSMD_ROT_MAT_TO_6D:
      for (int k = 0; k < 6; ++k) out[t * 6 + k] = in[t * 9 + k];
      break;
    case MSMD_ROT_EULER_TO_MAT: {
      float R0[9], R1[9], R2[9], R01[9];
      axis_rot(c0, in[t * 3 + 0], R0);
      axis_rot(c1, in[t * 3 + 1], R1);
      axis_rot(c2, in[t * 3 + 2], R2);
      mat3mul(R0, R1, R01);
      mat3mul(R01, R2, o);
      for (int k = 0; k < 9; ++k) out[t * 9 + k] = o[k];
      break;
    }
    case MSMD_ROT_MAT_TO_EULER: {
      for (int k = 0; k < 9; ++k) a[k] = in[t * 9 + k];
      const bool tb = c0 != c2;
      float central;
      if (tb) {
        const int df = c0 - c2;
        central = asinf(a[c0 * 3 + c2] * ((df == -1 || df == 2) ? -1.0f : 1.0f));
      } else {
        central = acosf(a[c0 * 3 + c0]);
      }
      const float col[3] = {a[0 * 3 + c2], a[1 * 3 + c2], a[2 * 3 + c2]};  // matrix[..., i2] (column i2)
      const float rowv[3] = {a[c0 * 3 + 0], a[c0 * 3 + 1], a[c0 * 3 + 2]};  // matrix[..., i0, :]
      out[t * 3 + 0] = angle_from_tan(c0, c1, col, false, tb);
      out[t * 3 + 1] = central;
      out[t * 3 + 2] = angle_from_tan(c2, c1, rowv, true, tb);
      break;
    }
    case MSMD_ROT_QUAT_STANDARDIZE:
      for (int k = 0; k < 4; ++k) a[k] = in[t * 4 + k];
      for (int k = 0; k < 4; ++k) out[t * 4 + k] = a[0] < 0.f ? -a[k] : a[k];
      break;
    case MSMD_ROT_QUAT_INVERT:
      out[t * 4] = in[t * 4];
      for (int k = 1; k < 4; ++k) out[t * 4 + k] = -in[t * 4 + k];
      break;
    case MSMD_ROT_QUAT_RAW_MUL:
    case MSMD_ROT_QUAT_MUL:
      for (int k = 0; k < 4; ++k) { a[k] = in[t * 4 + k]; b[k] = in2[t * 4 + k]; }
      quat_raw_mul(a, b, o);
      if (op == MSMD_ROT_QUAT_MUL && o[0] < 0.f) for (int k = 0; k < 4; ++k) o[k] = -o[k];
      for (int k = 0; k < 4; ++k) out[t * 4 + k] = o[k];
      break;
    case MSMD_ROT_QUAT_APPLY: {
      for (int k = 0; k < 4; ++k) a[k] = in[t * 4 + k];
      const float pq[4] = {0.f, in2[t * 3], in2[t * 3 + 1], in2[t * 3 + 2]};
      const float inv[4] = {a[0], -a[1], -a[2], -a[3]};
      float tmp[4];
      quat_raw_mul(a, pq, tmp);
      quat_raw_mul(tmp, inv, o);
      for (int k = 0; k < 3; ++k) out[t * 3 + k] = o[1 + k];
      break;
    }
    default: break;
  }
}

extern "C" int msmd_rotation_convert(int op, const float* in, const float* in2, float* out, long n, int conv,
                                     msmd_stream_t stream) {
  if (n <= 0 || op < 0 || op > MSMD_ROT_QUAT_APPLY || !in || !out) return 1;
  if ((op == MSMD_ROT_QUAT_RAW_MUL || op == MSMD_ROT_QUAT_MUL || op == MSMD_ROT_QUAT_APPLY) && !in2) return 1;
  hipLaunchKernelGGL(rotation_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, in, in2,
                     out, n, conv);
  MSMD_RETURN_LAST();
}
