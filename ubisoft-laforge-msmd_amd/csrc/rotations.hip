// Rotation conversions (reference utils/rotation_conversions.py:38-569; PyTorch3D semantics):
// real-first quaternions, small-angle Taylor branch below 1e-6, _sqrt_positive_part, _copysign.
// Elementwise and HBM-bound: one item per thread in registers, LDS-staged linear 16-byte global accesses (see rotation_kernel).
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
  const float d1 = i1 == 0 ? data[0] : (i1 == 1 ? data[1] : data[2]);   // selects: a run-time-indexed local array lives in scratch
  const float d2 = i2 == 0 ? data[0] : (i2 == 1 ? data[1] : data[2]);
  if (horizontal == even) return atan2f(d1, d2);
  if (tb) return atan2f(-d2, d1);
  return atan2f(d2, -d1);
}

// Array-of-structures items (3 / 4 / 6 / 9 floats each) never meet HBM through per-thread strided accesses: a workgroup owns
// 256 consecutive items, i.e. ONE contiguous span of the input and of the output.  The span is staged through LDS with 16-byte
// per-lane loads / stores in linear order (whole 64-byte sectors per 4 lanes, 1 KiB per wave instruction); each thread then
// takes its item from LDS (row stride 3 / 9 words is conflict-free, 4 words reads as one ds_read_b128), computes in registers,
// and puts the result back into the same LDS buffer for the linear store pass.  Unaligned bases (a view into a larger
// tensor) fall back to dword pieces in the same linear order.
constexpr int ROT_WG = 256;

__device__ __forceinline__ void rot_stage_in(float* __restrict__ dst, const float* __restrict__ src, int count) {
  if (((uintptr_t)src & 15) == 0) {
    const int nv = count >> 2;
    for (int i = threadIdx.x; i < nv; i += ROT_WG) ((float4*)dst)[i] = ((const float4*)src)[i];
    for (int i = (nv << 2) + threadIdx.x; i < count; i += ROT_WG) dst[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < count; i += ROT_WG) dst[i] = src[i];
  }
}
__device__ __forceinline__ void rot_stage_out(float* __restrict__ dst, const float* __restrict__ src, int count) {
  if (((uintptr_t)dst & 15) == 0) {
    const int nv = count >> 2;
    for (int i = threadIdx.x; i < nv; i += ROT_WG) ((float4*)dst)[i] = ((const float4*)src)[i];
    for (int i = (nv << 2) + threadIdx.x; i < count; i += ROT_WG) dst[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < count; i += ROT_WG) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(ROT_WG) void rotation_kernel(int op, const float* __restrict__ in, const float* __restrict__ in2,
                                                          float* __restrict__ out, long n, int conv, int in_w, int in2_w,
                                                          int out_w) {
  __shared__ __attribute__((aligned(16))) float s_io[ROT_WG * 9];
  __shared__ __attribute__((aligned(16))) float s_b[ROT_WG * 4];
  const long base = blockIdx.x * (long)ROT_WG;
  const int cnt = (int)min((long)ROT_WG, n - base);
  rot_stage_in(s_io, in + base * in_w, cnt * in_w);
  if (in2) rot_stage_in(s_b, in2 + base * in2_w, cnt * in2_w);
  __syncthreads();
  const int t = threadIdx.x;
  const bool live = t < cnt;
  float a[9], b[4], o[9], q[4];
  const int c0 = conv & 3, c1 = (conv >> 2) & 3, c2 = (conv >> 4) & 3;
  // ---- this thread's item out of LDS (compile-time widths per op: the arrays stay in registers)
  if (live) {
    switch (op) {
      case MSMD_ROT_QUAT_TO_MAT: case MSMD_ROT_QUAT_TO_AA: case MSMD_ROT_QUAT_STANDARDIZE: case MSMD_ROT_QUAT_INVERT:
      case MSMD_ROT_QUAT_RAW_MUL: case MSMD_ROT_QUAT_MUL: case MSMD_ROT_QUAT_APPLY: {
        const float4 v = ((const float4*)s_io)[t];
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        break;
      }
      case MSMD_ROT_AA_TO_QUAT: case MSMD_ROT_AA_TO_MAT: case MSMD_ROT_AA_TO_6D: case MSMD_ROT_EULER_TO_MAT:
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] = s_io[t * 3 + k];
        break;
      case MSMD_ROT_6D_TO_MAT:
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] = s_io[t * 6 + k];
        break;
      default:
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = s_io[t * 9 + k];
        break;
    }
    if (op == MSMD_ROT_QUAT_RAW_MUL || op == MSMD_ROT_QUAT_MUL) {
      const float4 v = ((const float4*)s_b)[t];
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else if (op == MSMD_ROT_QUAT_APPLY) {
      b[0] = 0.f; b[1] = s_b[t * 3]; b[2] = s_b[t * 3 + 1]; b[3] = s_b[t * 3 + 2];
    }
  }
  __syncthreads();      // every item is in registers: the buffer becomes the output span
  if (live) {
    switch (op) {
      case MSMD_ROT_QUAT_TO_MAT:
        quat_to_mat(a, o);
#pragma unroll
        for (int k = 0; k < 9; ++k) s_io[t * 9 + k] = o[k];
        break;
      case MSMD_ROT_MAT_TO_QUAT:
        mat_to_quat(a, o);
        ((float4*)s_io)[t] = float4{o[0], o[1], o[2], o[3]};
        break;
      case MSMD_ROT_AA_TO_QUAT:
        aa_to_quat(a, o);
        ((float4*)s_io)[t] = float4{o[0], o[1], o[2], o[3]};
        break;
      case MSMD_ROT_QUAT_TO_AA:
        quat_to_aa(a, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) s_io[t * 3 + k] = o[k];
        break;
      case MSMD_ROT_AA_TO_MAT:
        aa_to_quat(a, q);
        quat_to_mat(q, o);
#pragma unroll
        for (int k = 0; k < 9; ++k) s_io[t * 9 + k] = o[k];
        break;
      case MSMD_ROT_AA_TO_6D:
        aa_to_quat(a, q);
        quat_to_mat(q, o);
#pragma unroll
        for (int k = 0; k < 6; ++k) s_io[t * 6 + k] = o[k];
        break;
      case MSMD_ROT_MAT_TO_AA:
        mat_to_quat(a, q);
        quat_to_aa(q, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) s_io[t * 3 + k] = o[k];
        break;
      case MSMD_ROT_6D_TO_MAT: {
        float b1[3], b2[3], tmp[3];
        normalize3(a, b1);
        const float dot = b1[0] * a[3] + b1[1] * a[4] + b1[2] * a[5];
#pragma unroll
        for (int k = 0; k < 3; ++k) tmp[k] = a[3 + k] - dot * b1[k];
        normalize3(tmp, b2);
        o[0] = b1[0]; o[1] = b1[1]; o[2] = b1[2]; o[3] = b2[0]; o[4] = b2[1]; o[5] = b2[2];
        o[6] = b1[1] * b2[2] - b1[2] * b2[1]; o[7] = b1[2] * b2[0] - b1[0] * b2[2]; o[8] = b1[0] * b2[1] - b1[1] * b2[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) s_io[t * 9 + k] = o[k];
        break;
      }
      case MSMD_ROT_MAT_TO_6D:
#pragma unroll
        for (int k = 0; k < 6; ++k) s_io[t * 6 + k] = a[k];
        break;
      case MSMD_ROT_EULER_TO_MAT: {
        float R0[9], R1[9], R2[9], R01[9];
        axis_rot(c0, a[0], R0);
        axis_rot(c1, a[1], R1);
        axis_rot(c2, a[2], R2);
        mat3mul(R0, R1, R01);
        mat3mul(R01, R2, o);
#pragma unroll
        for (int k = 0; k < 9; ++k) s_io[t * 9 + k] = o[k];
        break;
      }
      case MSMD_ROT_MAT_TO_EULER: {
        const bool tb = c0 != c2;
        // run-time row / column picks as selects over the nine registers (a run-time-indexed local array lives in scratch)
        auto at = [&](int r, int c) {
          const int i = r * 3 + c;
          return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : i == 3 ? a[3] : i == 4 ? a[4] : i == 5 ? a[5]
                 : i == 6 ? a[6] : i == 7 ? a[7] : a[8];
        };
        float central;
        if (tb) {
          const int df = c0 - c2;
          central = asinf(at(c0, c2) * ((df == -1 || df == 2) ? -1.0f : 1.0f));
        } else {
          central = acosf(at(c0, c0));
        }
        const float col[3] = {at(0, c2), at(1, c2), at(2, c2)};   // matrix[..., i2] (column i2)
        const float rowv[3] = {at(c0, 0), at(c0, 1), at(c0, 2)};  // matrix[..., i0, :]
        s_io[t * 3 + 0] = angle_from_tan(c0, c1, col, false, tb);
        s_io[t * 3 + 1] = central;
        s_io[t * 3 + 2] = angle_from_tan(c2, c1, rowv, true, tb);
        break;
      }
      case MSMD_ROT_QUAT_STANDARDIZE: {
        ((float4*)s_io)[t] = a[0] < 0.f ? float4{-a[0], -a[1], -a[2], -a[3]} : float4{a[0], a[1], a[2], a[3]};
        break;
      }
      case MSMD_ROT_QUAT_INVERT:
        ((float4*)s_io)[t] = float4{a[0], -a[1], -a[2], -a[3]};
        break;
      case MSMD_ROT_QUAT_RAW_MUL:
      case MSMD_ROT_QUAT_MUL:
        quat_raw_mul(a, b, o);
        if (op == MSMD_ROT_QUAT_MUL && o[0] < 0.f) { o[0] = -o[0]; o[1] = -o[1]; o[2] = -o[2]; o[3] = -o[3]; }
        ((float4*)s_io)[t] = float4{o[0], o[1], o[2], o[3]};
        break;
      case MSMD_ROT_QUAT_APPLY: {
        const float inv[4] = {a[0], -a[1], -a[2], -a[3]};
        float tmp[4];
        quat_raw_mul(a, b, tmp);
        quat_raw_mul(tmp, inv, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) s_io[t * 3 + k] = o[1 + k];
        break;
      }
      default: break;
    }
  }
  __syncthreads();
  rot_stage_out(out + base * out_w, s_io, cnt * out_w);
}

extern "C" int msmd_rotation_convert(int op, const float* in, const float* in2, float* out, long n, int conv,
                                     msmd_stream_t stream) {
  if (n <= 0 || op < 0 || op > MSMD_ROT_QUAT_APPLY || !in || !out) return 1;
  if ((op == MSMD_ROT_QUAT_RAW_MUL || op == MSMD_ROT_QUAT_MUL || op == MSMD_ROT_QUAT_APPLY) && !in2) return 1;
  // item widths (floats) by op: input, second input, output
  static const signed char W[MSMD_ROT_QUAT_APPLY + 1][3] = {
      {4, 0, 9}, {9, 0, 4}, {3, 0, 4}, {4, 0, 3}, {3, 0, 9}, {9, 0, 3}, {6, 0, 9}, {9, 0, 6},
      {3, 0, 6}, {3, 0, 9}, {9, 0, 3}, {4, 0, 4}, {4, 0, 4}, {4, 4, 4}, {4, 4, 4}, {4, 3, 3}};
  const bool two = W[op][1] != 0;
  hipLaunchKernelGGL(rotation_kernel, dim3((unsigned)((n + ROT_WG - 1) / ROT_WG)), dim3(ROT_WG), 0, (hipStream_t)stream, op, in,
                     two ? in2 : nullptr, out, n, conv, (int)W[op][0], (int)W[op][1], (int)W[op][2]);
  MSMD_RETURN_LAST();
}
