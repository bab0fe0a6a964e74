// Row-wise LayerNorm and small reductions (HBM-bound; one wavefront per row, values kept in registers).
#include "common.h"

template <typename T> __device__ __forceinline__ void load4(const T* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
  const f32x4 t = *(const f32x4*)p;
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
  const bf16x4 t = *(const bf16x4*)p;
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float* v) {
  const f16x4 t = *(const f16x4*)p;
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
  *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) {
  *(bf16x4*)p = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float* v) {
  *(f16x4*)p = f16x4{(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
}

// cols % 4 == 0, cols <= 4 * 64 * MAXC
// SPLIT2: additionally (y may then be NULL) write the row in MSMD_F16X2 split storage to y2 (cols % 32 == 0)
// pre_g / pre_b (msmd_layernorm_pre): x holds UN-normalised rows and is first normalised with (pre_g, pre_b) -- rounded to
// the storage type, as the separate LayerNorm launch it replaces would have stored it -- before the residual is added:
//   y = LN_{gamma, beta}( LN_{pre}(x) + residual )       one launch for norm1 -> (+ cross-attention branch) -> norm2
template <typename TI, typename TO, int MAXC, bool SPLIT2 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, const TI* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ post, TO* __restrict__ y, int rows,
                                                        int cols, float eps, int act, f16_t* __restrict__ y2 = nullptr,
                                                        const float* __restrict__ pre_g = nullptr,
                                                        const float* __restrict__ pre_b = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = cols >> 2;
  float v[MAXC][4];
  float s = 0.f;
  if (pre_g) {      // wave-uniform: the pre-normalisation pass over the row (its own mean / variance)
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < nchunk) {
        load4<TI>(x + (long)row * cols + c * 4, v[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s0 += v[i][e];
      }
    }
    const float mean0 = wave_sum(s0) / (float)cols;
    float q0 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean0; q0 += d * d; }
      }
    }
    const float rstd0 = 1.0f / sqrtf(wave_sum(q0) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < nchunk) {
        float g0[4], b0[4];
        load4<float>(pre_g + c * 4, g0);
        load4<float>(pre_b + c * 4, b0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (float)(TI)((v[i][e] - mean0) * rstd0 * g0[e] + b0[e]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
      if (!pre_g) load4<TI>(x + (long)row * cols + c * 4, v[i]);
      if (res) {
        float r[4];
        load4<TI>(res + (long)row * cols + c * 4, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] += r[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = apply_act(v[i][e], act & 0xff);
        s += v[i][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
      float g[4], b[4], o[4];
      load4<float>(gamma + c * 4, g);
      load4<float>(beta + c * 4, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if (act >> 8) {  // activation AFTER the affine (LayerNorm -> GELU of the layer-norm conv stack)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = sizeof(TO) == 2 ? (((act >> 8) == MSMD_ACT_GELU) ? gelu_fast(o[e]) : apply_act(o[e], act >> 8)) : apply_act(o[e], act >> 8);
      }
      if (post) {
        float pa[4];
        load4<float>(post + c * 4, pa);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += pa[e];
      }
      if (!SPLIT2 || y) store4<TO>(y + (long)row * cols + c * 4, o);
      if constexpr (SPLIT2) store4_split(y2 + (long)row * 2 * cols, c * 4, o);
    }
  }
}


// 16-bit rows in, 16-bit rows out, cols % 8 == 0, no activation / post-add: the same arithmetic with ONE 16-byte access per
// lane and 512-column stretch (a d = 512 row is a single access: the decoder / sampler rows; 768 and 1024 take two).  The
// 8-byte accesses of layernorm_kernel leave a third of the achievable bandwidth on the table on these rows (65 MB of
// norm1 -> + branch -> norm2 traffic per decoder layer and sampler step: 21 us there, 3.1 TB/s).
template <typename T, int MAXC8>
__global__ __launch_bounds__(256) void layernorm16_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ y, int rows, int cols, float eps,
                                                          const float* __restrict__ pre_g, const float* __restrict__ pre_b) {
  typedef typename Vec8T<T>::type V8;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = cols >> 3;
  const long base = (long)row * cols;
  V8 xr[MAXC8], rr[MAXC8];
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {          // every load of the row pair before the first use
    const int c = i * 64 + lane;
    if (c < nchunk) {
      xr[i] = *(const V8*)(x + base + c * 8);
      if (res) rr[i] = *(const V8*)(res + base + c * 8);
    }
  }
  float v[MAXC8][8];
  const float inv = 1.0f / (float)cols;
  if (pre_g) {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i)
      if (i * 64 + lane < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = (float)xr[i][e]; s0 += v[i][e]; }
      }
    const float mean0 = wave_sum(s0) * inv;
    float q0 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i)
      if (i * 64 + lane < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean0; q0 += d * d; }
      }
    const float rstd0 = 1.0f / sqrtf(wave_sum(q0) * inv + eps);
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
      const int c = i * 64 + lane;
      if (c < nchunk) {
        const f32x4 g0a = *(const f32x4*)(pre_g + c * 8), g0b = *(const f32x4*)(pre_g + c * 8 + 4);
        const f32x4 b0a = *(const f32x4*)(pre_b + c * 8), b0b = *(const f32x4*)(pre_b + c * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e] = (float)(T)((v[i][e] - mean0) * rstd0 * g0a[e] + b0a[e]);
          v[i][4 + e] = (float)(T)((v[i][4 + e] - mean0) * rstd0 * g0b[e] + b0b[e]);
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    if (i * 64 + lane < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!pre_g) v[i][e] = (float)xr[i][e];
        if (res) v[i][e] += (float)rr[i][e];
        s += v[i][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) * inv;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC8; ++i)
    if (i * 64 + lane < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * inv + eps);
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int c = i * 64 + lane;
    if (c < nchunk) {
      const f32x4 ga = *(const f32x4*)(gamma + c * 8), gb = *(const f32x4*)(gamma + c * 8 + 4);
      const f32x4 ba = *(const f32x4*)(beta + c * 8), bb = *(const f32x4*)(beta + c * 8 + 4);
      V8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (T)((v[i][e] - mean) * rstd * ga[e] + ba[e]);
        o[4 + e] = (T)((v[i][4 + e] - mean) * rstd * gb[e] + bb[e]);
      }
      *(V8*)(y + base + c * 8) = o;
    }
  }
}

// -> true when the 16-byte form took the call
template <typename T>
static bool launch_ln16(const void* x, const void* res, const float* g, const float* b, void* y, int rows, int cols, float eps,
                        hipStream_t st, const float* pre_g, const float* pre_b) {
  if ((cols & 7) || cols > 8 * 64 * 4 ||
      (((uintptr_t)x | (uintptr_t)res | (uintptr_t)y | (uintptr_t)g | (uintptr_t)b | (uintptr_t)pre_g | (uintptr_t)pre_b) & 15))
    return false;
  dim3 grid((rows + 3) / 4), block(256);
  if (cols <= 512)
    hipLaunchKernelGGL((layernorm16_kernel<T, 1>), grid, block, 0, st, (const T*)x, (const T*)res, g, b, (T*)y, rows, cols, eps, pre_g, pre_b);
  else if (cols <= 1024)
    hipLaunchKernelGGL((layernorm16_kernel<T, 2>), grid, block, 0, st, (const T*)x, (const T*)res, g, b, (T*)y, rows, cols, eps, pre_g, pre_b);
  else
    hipLaunchKernelGGL((layernorm16_kernel<T, 4>), grid, block, 0, st, (const T*)x, (const T*)res, g, b, (T*)y, rows, cols, eps, pre_g, pre_b);
  return true;
}

template <typename TI, typename TO>
static int launch_ln(const void* x, const void* res, const float* g, const float* b, const float* post, void* y,
                     int rows, int cols, float eps, int act, hipStream_t st, const float* pre_g = nullptr,
                     const float* pre_b = nullptr) {
  dim3 grid((rows + 3) / 4), block(256);
  f16_t* none = nullptr;
  if (cols <= 4 * 64 * 2)
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, 2>), grid, block, 0, st, (const TI*)x, (const TI*)res, g, b, post,
                       (TO*)y, rows, cols, eps, act, none, pre_g, pre_b);
  else if (cols <= 4 * 64 * 4)
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, 4>), grid, block, 0, st, (const TI*)x, (const TI*)res, g, b, post,
                       (TO*)y, rows, cols, eps, act, none, pre_g, pre_b);
  else if (cols <= 4 * 64 * 16)
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, 16>), grid, block, 0, st, (const TI*)x, (const TI*)res, g, b, post,
                       (TO*)y, rows, cols, eps, act, none, pre_g, pre_b);
  else
    return 1;
  MSMD_RETURN_LAST();
}

// fp32 in; y (fp32, may be NULL) and / or y2 (split storage): the LayerNorm output is both the next GEMM's A operand
// (split) and the residual of the block after it (fp32) -- one pass writes both.
extern "C" int msmd_layernorm_f16x2(const float* x, const float* residual, const float* gamma, const float* beta,
                                    const float* post_add, float* y, void* y_split, int rows, int cols, float eps,
                                    int act, msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || (cols & 31) || !x || !y_split || !gamma || !beta || ((uintptr_t)y_split & 15)) return 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
  f16_t* y2 = (f16_t*)y_split;
  if (cols <= 4 * 64 * 2)
    hipLaunchKernelGGL((layernorm_kernel<float, float, 2, true>), grid, block, 0, st, x, residual, gamma, beta, post_add,
                       y, rows, cols, eps, act, y2);
  else if (cols <= 4 * 64 * 4)
    hipLaunchKernelGGL((layernorm_kernel<float, float, 4, true>), grid, block, 0, st, x, residual, gamma, beta, post_add,
                       y, rows, cols, eps, act, y2);
  else if (cols <= 4 * 64 * 16)
    hipLaunchKernelGGL((layernorm_kernel<float, float, 16, true>), grid, block, 0, st, x, residual, gamma, beta,
                       post_add, y, rows, cols, eps, act, y2);
  else
    return 1;
  MSMD_RETURN_LAST();
}

extern "C" int msmd_layernorm(const void* x, const void* residual, const float* gamma, const float* beta,
                              const float* post_add, void* y, int rows, int cols, float eps, int act, int in_dtype,
                              int out_dtype, msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || (cols & 3) || !x || !y || !gamma || !beta) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_F32)
    return launch_ln<float, float>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  if (in_dtype == out_dtype && (in_dtype == MSMD_BF16 || in_dtype == MSMD_F16) && act == 0 && !post_add) {   // any other code (MSMD_F16X2, junk) falls through to `return 1`
    const bool took = in_dtype == MSMD_BF16 ? launch_ln16<bf16_t>(x, residual, gamma, beta, y, rows, cols, eps, st, nullptr, nullptr)
                                            : launch_ln16<f16_t>(x, residual, gamma, beta, y, rows, cols, eps, st, nullptr, nullptr);
    if (took) MSMD_RETURN_LAST();
  }
  if (in_dtype == MSMD_BF16 && out_dtype == MSMD_BF16)
    return launch_ln<bf16_t, bf16_t>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  if (in_dtype == MSMD_BF16 && out_dtype == MSMD_F32)
    return launch_ln<bf16_t, float>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_BF16)
    return launch_ln<float, bf16_t>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  if (in_dtype == MSMD_F16 && out_dtype == MSMD_F16)
    return launch_ln<f16_t, f16_t>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  if (in_dtype == MSMD_F16 && out_dtype == MSMD_F32)
    return launch_ln<f16_t, float>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_F16)
    return launch_ln<float, f16_t>(x, residual, gamma, beta, post_add, y, rows, cols, eps, act, st);
  return 1;
}

// y = LN_{gamma, beta}( LN_{pre_gamma, pre_beta}(x) + residual ): two consecutive post-LN LayerNorms with a branch added in
// between (nn.TransformerDecoderLayer: norm1 -> + cross-attention branch -> norm2) in ONE launch; 16-bit rows.
extern "C" int msmd_layernorm_pre(const void* x, const float* pre_gamma, const float* pre_beta, const void* residual,
                                  const float* gamma, const float* beta, void* y, int rows, int cols, float eps, int dtype,
                                  msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || (cols & 3) || !x || !y || !gamma || !beta || !pre_gamma || !pre_beta) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSMD_BF16 || dtype == MSMD_F16) {
    const bool took = dtype == MSMD_BF16 ? launch_ln16<bf16_t>(x, residual, gamma, beta, y, rows, cols, eps, st, pre_gamma, pre_beta)
                                         : launch_ln16<f16_t>(x, residual, gamma, beta, y, rows, cols, eps, st, pre_gamma, pre_beta);
    if (took) MSMD_RETURN_LAST();
  }
  if (dtype == MSMD_BF16)
    return launch_ln<bf16_t, bf16_t>(x, residual, gamma, beta, nullptr, y, rows, cols, eps, 0, st, pre_gamma, pre_beta);
  if (dtype == MSMD_F16)
    return launch_ln<f16_t, f16_t>(x, residual, gamma, beta, nullptr, y, rows, cols, eps, 0, st, pre_gamma, pre_beta);
  return 1;
}

// mean over time of a channels-last (B, T, C) tensor -> (B, C) fp32
template <typename T>
__global__ void mean_time_kernel(const T* __restrict__ x, float* __restrict__ y, int Tn, int C) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const T* p = x + (long)b * Tn * C + c;
  float s = 0.f;
  for (int t = 0; t < Tn; ++t) s += to_f32(p[(long)t * C]);
  y[(long)b * C + c] = s / (float)Tn;
}

extern "C" int msmd_mean_time(const void* x, float* y, int B, int T, int C, int dtype, msmd_stream_t stream) {
  if (B <= 0 || T <= 0 || C <= 0) return 1;
  dim3 grid((C + 255) / 256, B), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(mean_time_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, y, T, C);
  else if (dtype == MSMD_F16)
    hipLaunchKernelGGL(mean_time_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const f16_t*)x, y, T, C);
  else
    hipLaunchKernelGGL(mean_time_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, y, T, C);
  MSMD_RETURN_LAST();
}
