// Shared device helpers for the MSMD gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msmd_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define MSMD_WAVE 64

// Launch-error plumbing: every extern "C" launcher returns a hipError_t-compatible int.
// Philox4x32-10 (Salmon et al. 2011): counter-based, so a dropout mask is a pure function of
// (seed, step, call site, element index) and the backward pass regenerates it instead of storing it.
struct Philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
    const unsigned lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// rng_state (device): [0] = seed, [1] = step counter (advanced by the host side once per iteration / graph replay)
__device__ __forceinline__ Philox4 dropout_bits(const unsigned long* __restrict__ rng_state, unsigned site, unsigned long idx) {
  const unsigned long seed = rng_state[0], step = rng_state[1];
  return philox4x32((unsigned)idx, (unsigned)(idx >> 32), site, (unsigned)step, (unsigned)seed,
                    (unsigned)(seed >> 32) ^ (unsigned)(step >> 32));
}
// keep iff uniform [0,1) >= p  <=>  bits >= p * 2^32
__device__ __forceinline__ unsigned dropout_threshold(float p) { return (unsigned)fminf(p * 4294967296.0f, 4294967295.0f); }

extern int g_tuning[8];  // msmd_set_tuning knobs (gemm.hip)
#define MSMD_RETURN_LAST() return (int)hipGetLastError()

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // RNE, v_cvt_pk_bf16_f32
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return (f16_t)x; }     // RNE, v_cvt_f16_f32

// 4- and 8-wide vectors of a storage type (8 / 16 bytes for the 16-bit types)
template <typename T> struct Vec4T;
template <> struct Vec4T<float> { typedef f32x4 type; };
template <> struct Vec4T<bf16_t> { typedef bf16x4 type; };
template <> struct Vec4T<f16_t> { typedef f16x4 type; };
template <typename T> struct Vec8T;
template <> struct Vec8T<bf16_t> { typedef bf16x8 type; };
template <> struct Vec8T<f16_t> { typedef f16x8 type; };
template <typename T> __device__ __forceinline__ typename Vec4T<T>::type pack4(float a, float b, float c, float d) {
  return typename Vec4T<T>::type{(T)a, (T)b, (T)c, (T)d};
}
// 16x16x32 MFMA on 8 x 16-bit operands held as 4 dwords
template <typename T> __device__ __forceinline__ f32x4 mfma16(const u32x4 a, const u32x4 b, const f32x4 c) {
  if constexpr (sizeof(T) == 2 && __is_same(T, f16_t))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Exact erf GELU (HF ACT2FN['gelu'] / torch F.gelu(approximate='none')).
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): 1 rcp + 1 exp + 6 FMA instead of libm's branchy erff.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == MSMD_ACT_GELU) return gelu_erf(x);
  if (act == MSMD_ACT_ELU) return elu1(x);
  return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); red must hold >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Typed 16-byte / scalar loads with fp32 widening.
template <typename T> struct Elems16;                     // elements per 16-byte chunk
template <> struct Elems16<float> { static constexpr int n = 4; };
template <> struct Elems16<bf16_t> { static constexpr int n = 8; };
template <> struct Elems16<f16_t> { static constexpr int n = 8; };
