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
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
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
// Attention-probability dropout draws 16-bit values: one Philox4 block serves the lane's 8 keys of a fragment PAIR (key
// fragment f even: low halves of the 4 words, odd: high halves), counter = row * 128 + 4 * (f / 2) + fq -- half the
// generator work of a 32-bit value per key (the generator was as much VALU work as the rest of the forward kernel).
// keep <=> value >= threshold16(p); P(drop) = floor(p * 65536) / 65536.
__device__ __forceinline__ unsigned dropout_threshold16(float p) { return (unsigned)fminf(p * 65536.0f, 65535.0f); }
__device__ __forceinline__ unsigned dropout_value16(const Philox4& r, int e, int odd) {
  const unsigned w = e == 0 ? r.x : e == 1 ? r.y : e == 2 ? r.z : r.w;
  return odd ? w >> 16 : w & 0xffffu;
}

// Developer knobs: only the experimental build (make EXP=1, -DMSMD_EXPERIMENTAL) has them; in the product library every
// MSMD_TUNE(k) is the constant 0 and the code it guards folds away (no process-global state: re-entrant per stream).
#ifdef MSMD_EXPERIMENTAL
extern int g_tuning[16];
#define MSMD_TUNE(k) (g_tuning[k])
#else
#define MSMD_TUNE(k) 0
#endif
#define MSMD_RETURN_LAST() return (int)hipGetLastError()

// Zero a small device workspace with a kernel launch instead of hipMemsetAsync: every node of a captured hipGraph is then
// a kernel node (memset nodes were one suspect while hunting the second-replay NaN of the segmented training graphs,
// DESIGN.md 5c -- the cause turned out to be the host library's multi-block reduction; this stays because it costs nothing).
static __global__ void msmd_zero_kernel(unsigned* __restrict__ p, long n_words) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i < n_words) p[i] = 0u;
}
static inline hipError_t msmd_zero_async(void* p, size_t bytes, hipStream_t st) {   // bytes % 4 == 0
  const long n = (long)(bytes / 4);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(msmd_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned*)p, n);
  return hipGetLastError();
}

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
// Four fp32 values -> four stored values.  16-bit types: TWO packed conversions (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round to
// nearest even, both new on gfx950).  Written as four scalar casts, hipcc emitted one conversion per VALUE inside the GEMM
// epilogues (second operand a dummy), a NaN select behind each bf16 one and v_perm_b32 to pair the halves: 10 vector
// instructions per 4 outputs where 2 do -- a fifth of the vector work of a 128 x 128 tile's epilogue.  Same bits for every
// finite value (the instruction is the one the casts lower to); asm without `volatile`, so the scheduler stays free.
template <typename T> __device__ __forceinline__ typename Vec4T<T>::type pack4(float a, float b, float c, float d) {
  if constexpr (__is_same(T, bf16_t)) {
    u32x2 r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[0]) : "v"(a), "v"(b));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[1]) : "v"(c), "v"(d));
    return __builtin_bit_cast(bf16x4, r);
  } else if constexpr (__is_same(T, f16_t)) {
    u32x2 r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[0]) : "v"(a), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[1]) : "v"(c), "v"(d));
    return __builtin_bit_cast(f16x4, r);
  } else {
    return typename Vec4T<T>::type{(T)a, (T)b, (T)c, (T)d};
  }
}
// 16x16x32 MFMA on 8 x 16-bit operands held as 4 dwords
template <typename T> __device__ __forceinline__ f32x4 mfma16(const u32x4 a, const u32x4 b, const f32x4 c) {
  if constexpr (sizeof(T) == 2 && __is_same(T, f16_t))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------
// MSMD_F16X2 ("split pair") storage: a logical fp32 value x is kept as two fp16 numbers
//   hi = RN_f16(x),  lo = RN_f16((x - hi) * 2^11)          x ~= hi + lo * 2^-11   (|err| <= 2^-22 |x| + 2^-35, |x| < 65504)
// A logical row of C values (C % 32 == 0) is a row of 2 C fp16 numbers in 32-element blocks [hi x 32 | lo x 32]:
// one 128-byte line = one 32-deep MFMA k-step of both planes, so tiles are staged by the same LDS-DMA / LDS image as
// plain 16-bit tiles.  A product sum is three f16 MFMAs per k-step (hi.hi -> acc0; hi.lo + lo.hi -> acc1) and
// acc0 + acc1 * 2^-11 at the end: fp32-grade results at one third of the f16 MFMA rate instead of the 1/16 of
// v_mfma_f32_16x16x4_f32 (the dropped lo.lo term is <= 2^-22 relative).
#define MSMD_SPLIT_SCALE 2048.0f
#define MSMD_SPLIT_INV 4.8828125e-4f
__device__ __forceinline__ void split_f16x2(float x, f16_t& hi, f16_t& lo) {
  // Pin x to ONE register value first.  Under -ffp-contract=fast hipcc may duplicate the expression that produced x
  // into both uses below and contract the copies differently (a (1 + b) as a + a b in one, fma(a, b, a) in the other):
  // two values one fp32 ulp apart.  When x sits on an fp16 rounding tie, hi then comes from one neighbour and the
  // residual from the other: lo gets the wrong sign and the pair is off by a whole fp16 ulp (found by the conv0 test:
  // 14 of 1.3 M GELU outputs).
  asm volatile("" : "+v"(x));
  hi = (f16_t)x;
  lo = (f16_t)((x - (float)hi) * MSMD_SPLIT_SCALE);
}
__device__ __forceinline__ float unsplit_f16x2(f16_t hi, f16_t lo) { return fmaf((float)lo, MSMD_SPLIT_INV, (float)hi); }
// physical fp16 offset of logical column c inside a split row (hi; lo is 32 further)
__device__ __forceinline__ long split_col(int c) { return ((long)(c >> 5) << 6) + (c & 31); }
// 4 consecutive logical columns c .. c+3 (c % 4 == 0) of a split row
__device__ __forceinline__ void store4_split(f16_t* row, int c, const float* v) {
  f16_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split_f16x2(v[e], h[e], l[e]);
  f16_t* p = row + split_col(c);
  *(f16x4*)p = f16x4{h[0], h[1], h[2], h[3]};
  *(f16x4*)(p + 32) = f16x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ void load4_split(const f16_t* row, int c, float* v) {
  const f16_t* p = row + split_col(c);
  const f16x4 h = *(const f16x4*)p, l = *(const f16x4*)(p + 32);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = unsplit_f16x2(h[e], l[e]);
}
// 8 consecutive logical columns (c % 8 == 0): two 16-byte stores
__device__ __forceinline__ void store8_split(f16_t* row, int c, const float* v) {
  f16_t h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split_f16x2(v[e], h[e], l[e]);
  f16_t* p = row + split_col(c);
  *(f16x8*)p = f16x8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
  *(f16x8*)(p + 32) = f16x8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
}

// Exact erf GELU (HF ACT2FN['gelu'] / torch F.gelu(approximate='none')).
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): 1 rcp + 1 exp + 6 FMA instead of libm's branchy erff.
// The reciprocal is the bare v_rcp_f32 (1 ulp; its argument is in [1, inf)): __frcp_rn expands to the ten-instruction
// correctly-rounded division sequence, which was a third of this function and ~25 % of the conv0 + GroupNorm + GELU kernel.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// GELU for 16-BIT OUTPUTS without transcendentals: Phi(t) = 0.5 + t P(t^2) on |t| <= 4.25 (near-minimax fit of the normal
// CDF, |Phi error| <= 8.5e-6), t = clamp(x): |error| <= 5.5e-5 everywhere, <= 1.9e-4 relative for x > -2 -- below half an
// ulp of bf16 and under fp16's own rounding.  12 full-rate vector instructions against 12 + v_rcp_f32 + v_exp_f32 (quarter
// rate: 20 issue slots) for gelu_fast: the conv0 kernel and the GELU GEMM epilogues are bound by exactly these.
__device__ __forceinline__ float gelu_poly16(float x) {
  const float t = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
  const float u = t * t;
  float p = 5.564819092e-11f;
  p = fmaf(p, u, -5.327732033e-09f);
  p = fmaf(p, u, 2.255421094e-07f);
  p = fmaf(p, u, -5.626418897e-06f);
  p = fmaf(p, u, 9.341863915e-05f);
  p = fmaf(p, u, -1.108560711e-03f);
  p = fmaf(p, u, 9.815970436e-03f);
  p = fmaf(p, u, -6.634449214e-02f);
  p = fmaf(p, u, 3.989023268e-01f);
  return x * fmaf(t, p, 0.5f);
}

// exp(x) for x <= 0 (softmax numerators) at ~1 ulp from ONE v_exp_f32: t = x log2(e) with the product's rounding error
// and log2(e)'s own fp32 rounding carried separately (r) and applied as exp2(t) (1 + r ln 2).  libm's expf is ~25
// VALU instructions, this is 7; __expf alone (x * log2e rounded once) is off by up to |x| * 2^-24 relative.
__device__ __forceinline__ float exp_neg_accurate(float x) {
  x = fmaxf(x, -200.0f);                       // -inf (masked keys) -> exp2(-288) = 0 without NaNs in the residual
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
  const float t = x * L2E_HI;
  const float r = fmaf(x, L2E_HI, -t) + x * L2E_LO;
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e * r, 0.693147180559945309f, e);
}

// d act(z) / dz for the GEMM epilogue of the fused data-gradient (msmd_gemm_actbwd): the 12-instruction erf and one v_exp_f32
// (|error| <= 2e-7 on a factor the 16-bit gradient rounds to 8 / 11 bits); backward.hip's act_grad keeps libm for fp32.
__device__ __forceinline__ float act_grad_fast(float z, int act) {
  if (act == MSMD_ACT_GELU) return 0.5f * (1.0f + erf_fast(z * 0.70710678118654752440f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
  if (act == MSMD_ACT_ELU) return z > 0.f ? 1.0f : __expf(z);
  return 1.0f;
}

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
