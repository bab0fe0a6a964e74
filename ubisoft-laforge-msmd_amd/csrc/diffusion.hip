// Denoiser glue and the CFG + DDPM update: small elementwise kernels (HBM/launch-bound), written so
// that one sampler step is a fixed sequence of launches with no host round trip (hipGraph-capturable).
#include "common.h"

// feats rows 1.. = [prev_motion ; x_t] ++ indicator ++ zero pad; row 0 (person token slot) is zero-filled.
template <typename TO>
__global__ void pack_input_kernel(const float* __restrict__ motion, const float* __restrict__ eps,
                                  const float* __restrict__ c0, const float* __restrict__ c1,
                                  const float* __restrict__ prev, const float* __restrict__ ind,
                                  TO* __restrict__ feats, int L, int Lp, int dm, int Kpad, int motion_batch) {
  const int n = blockIdx.y;
  const int Tn = 1 + Lp + L;
  const int nm = n % motion_batch;  // CFG entries share one x_t (reference model.py:389)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Tn * Kpad; i += gridDim.x * blockDim.x) {
    const int r = i / Kpad - 1, k = i % Kpad;  // r == -1: person-token row, zero-filled here
    float v = 0.f;
    if (r >= 0 && r < Lp) {
      if (k < dm) v = prev[((long)n * Lp + r) * dm + k];
    } else if (r >= Lp) {
      const int t = r - Lp;
      if (k < dm) {
        v = motion[((long)nm * L + t) * dm + k];
        if (eps) v = c0[nm] * v + c1[nm] * eps[((long)nm * L + t) * dm + k];
      } else if (k == dm && ind) {
        v = ind[(long)n * L + t];
      }
    }
    feats[(long)n * Tn * Kpad + i] = from_f32<TO>(v);
  }
}

extern "C" int msmd_denoiser_pack_input(const float* motion, const float* eps, const float* c0, const float* c1,
                                        const float* prev_motion, const float* indicator, void* feats, int N, int L,
                                        int Lp, int dm, int Kpad, int motion_batch, int out_dtype,
                                        msmd_stream_t stream) {
  if (N <= 0 || L <= 0 || Lp < 0 || dm <= 0 || Kpad < dm + (indicator ? 1 : 0) || motion_batch <= 0) return 1;
  dim3 grid(((1 + Lp + L) * Kpad + 255) / 256, N), block(256);
  if (out_dtype == MSMD_F32)
    hipLaunchKernelGGL(pack_input_kernel<float>, grid, block, 0, (hipStream_t)stream, motion, eps, c0, c1, prev_motion,
                       indicator, (float*)feats, L, Lp, dm, Kpad, motion_batch);
  else if (out_dtype == MSMD_F16)
    hipLaunchKernelGGL(pack_input_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, motion, eps, c0, c1,
                       prev_motion, indicator, (f16_t*)feats, L, Lp, dm, Kpad, motion_batch);
  else
    hipLaunchKernelGGL(pack_input_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, motion, eps, c0, c1,
                       prev_motion, indicator, (bf16_t*)feats, L, Lp, dm, Kpad, motion_batch);
  MSMD_RETURN_LAST();
}

// x (N, T, d) += pe (T, d); row 0 = tok0 (N, d) + row0_add (d, optional) + pe[0]  (row 0 of x is overwritten)
template <typename T>
__global__ void add_pe_token_kernel(T* __restrict__ x, const float* __restrict__ pe, const T* __restrict__ tok0,
                                    const T* __restrict__ row0_add, int Tn, int d) {
  const int n = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Tn * d; i += gridDim.x * blockDim.x) {
    const int t = i / d, c = i % d;
    float base;
    if (t == 0) {
      base = to_f32(tok0[(long)n * d + c]);
      if (row0_add) base += to_f32(row0_add[c]);
    } else {
      base = to_f32(x[(long)n * Tn * d + i]);
    }
    x[(long)n * Tn * d + i] = from_f32<T>(base + pe[i]);
  }
}

// the same on 16-bit rows with d % 8 == 0: 8 elements (one 16-byte access) per thread, the same arithmetic per element
template <typename T>
__global__ __launch_bounds__(256) void add_pe_token_v8_kernel(T* __restrict__ x, const float* __restrict__ pe,
                                                              const T* __restrict__ tok0, const T* __restrict__ row0_add,
                                                              int Tn, int d) {
  typedef typename Vec8T<T>::type V8;
  const int n = blockIdx.y, per_row = d >> 3;
  const int i8 = blockIdx.x * blockDim.x + threadIdx.x;
  if (i8 >= Tn * per_row) return;
  const int t = i8 / per_row, c = (i8 - t * per_row) << 3;
  T* xp = x + ((long)n * Tn + t) * d + c;
  float base[8];
  if (t == 0) {
    const V8 tk = *(const V8*)(tok0 + (long)n * d + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) base[e] = (float)tk[e];
    if (row0_add) {
      const V8 ra = *(const V8*)(row0_add + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) base[e] += (float)ra[e];
    }
  } else {
    const V8 xv = *(const V8*)xp;
#pragma unroll
    for (int e = 0; e < 8; ++e) base[e] = (float)xv[e];
  }
  const f32x4 p0 = *(const f32x4*)(pe + (long)t * d + c), p1 = *(const f32x4*)(pe + (long)t * d + c + 4);
  V8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = (T)(base[e] + p0[e]); o[4 + e] = (T)(base[4 + e] + p1[e]); }
  *(V8*)xp = o;
}

extern "C" int msmd_add_pe_token(void* x, const float* pe, const void* tok0, const void* row0_add, int N, int T, int d,
                                 int dtype, msmd_stream_t stream) {
  if (N <= 0 || T <= 0 || d <= 0 || !tok0) return 1;
  if (dtype != MSMD_F32 && (d & 7) == 0 && !(((uintptr_t)x | (uintptr_t)pe | (uintptr_t)tok0 | (uintptr_t)row0_add) & 15)) {
    dim3 grid8((T * (d >> 3) + 255) / 256, N);
    if (dtype == MSMD_F16)
      hipLaunchKernelGGL(add_pe_token_v8_kernel<f16_t>, grid8, dim3(256), 0, (hipStream_t)stream, (f16_t*)x, pe,
                         (const f16_t*)tok0, (const f16_t*)row0_add, T, d);
    else
      hipLaunchKernelGGL(add_pe_token_v8_kernel<bf16_t>, grid8, dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, pe,
                         (const bf16_t*)tok0, (const bf16_t*)row0_add, T, d);
    MSMD_RETURN_LAST();
  }
  dim3 grid((T * d + 255) / 256, N), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(add_pe_token_kernel<float>, grid, block, 0, (hipStream_t)stream, (float*)x, pe,
                       (const float*)tok0, (const float*)row0_add, T, d);
  else if (dtype == MSMD_F16)
    hipLaunchKernelGGL(add_pe_token_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (f16_t*)x, pe,
                       (const f16_t*)tok0, (const f16_t*)row0_add, T, d);
  else
    hipLaunchKernelGGL(add_pe_token_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (bf16_t*)x, pe,
                       (const bf16_t*)tok0, (const bf16_t*)row0_add, T, d);
  MSMD_RETURN_LAST();
}

// out = dyn + sum_b alpha_b * static_b (dims < dm-3, or all dims when use_head_alpha & 1) ; + sum_b static_b (last 3
// dims); use_head_alpha & 2: alpha_b = sigmoid(raw alpha_b)
template <typename T>
__global__ void heads_mix_kernel(const T* __restrict__ dec, long ld_dec, const T* __restrict__ stat,
                                 float* __restrict__ out, int L, int dm, int nb, int stat_batch, int use_head_alpha) {
  const int n = blockIdx.y;
  const int ns = n % stat_batch;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L * dm; i += gridDim.x * blockDim.x) {
    const int t = i / dm, k = i % dm;
    const T* row = dec + ((long)n * L + t) * ld_dec;
    float v = 0.f;
    const bool weighted = (use_head_alpha & 1) || (k < dm - 3);
    for (int b = 0; b < nb; ++b) {
      const float s = to_f32(stat[((long)ns * nb + b) * dm + k]);
      float a = to_f32(row[dm + b]);
      if (use_head_alpha & 2) a = 1.0f / (1.0f + expf(-a));   // regularize_alpha = 'sigmoid' (model.py:973-974)
      v += weighted ? s * a : s;
    }
    out[((long)n * L + t) * dm + k] = to_f32(row[k]) + v;
  }
}

extern "C" int msmd_heads_static_mix(const void* dec, long ld_dec, const void* stat, float* out, int N, int L, int dm,
                                     int nb, int stat_batch, int use_head_alpha, int dtype, msmd_stream_t stream) {
  if (N <= 0 || L <= 0 || dm <= 3 || nb <= 0 || stat_batch <= 0) return 1;
  dim3 grid((L * dm + 255) / 256, N), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(heads_mix_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)dec, ld_dec,
                       (const float*)stat, out, L, dm, nb, stat_batch, use_head_alpha);
  else if (dtype == MSMD_F16)
    hipLaunchKernelGGL(heads_mix_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const f16_t*)dec, ld_dec,
                       (const f16_t*)stat, out, L, dm, nb, stat_batch, use_head_alpha);
  else
    hipLaunchKernelGGL(heads_mix_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dec, ld_dec,
                       (const bf16_t*)stat, out, L, dm, nb, stat_batch, use_head_alpha);
  MSMD_RETURN_LAST();
}

// CFG combine + DDPM posterior step, in place on x (B, L, dm).  The reference accumulates theta IN PLACE
// into entry 0's slice (model.py:407-415): `results[0]` IS the running theta, so
//   incremental: theta += s_e * (r[e+1] - r[e])   with r[0] := running theta when e == 0,
//   independent: theta += s_e * (r[e+1] - theta)  (later terms see the already-updated entry 0).
__global__ void cfg_ddpm_kernel(float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ z,
                                const float* __restrict__ scales, int n_entries, int B, int L, int Lp, int dm,
                                int mode, int target, float c0, float c1, float sigma) {
  const long total = (long)B * L * dm;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long)L * dm));
    const int rem = (int)(i % ((long)L * dm));
    const int t = rem / dm, k = rem % dm;
    const long stride_e = (long)B * (Lp + L) * dm;
    const long off = ((long)b * (Lp + L) + Lp + t) * dm + k;
    float theta = res[off];
    for (int e = 0; e < n_entries - 1; ++e) {
      const float hi = res[(e + 1) * stride_e + off];
      const float lo = (mode == 1 || e == 0) ? theta : res[e * stride_e + off];
      theta += scales[e] * (hi - lo);
    }
    const float xt = x[i];
    const float zz = z ? z[i] : 0.f;
    x[i] = (target == 0) ? (c0 * xt + c1 * theta + sigma * zz) : (c0 * (xt - c1 * theta) + sigma * zz);
  }
}

extern "C" int msmd_cfg_ddpm_step(float* x, const float* res, const float* z, const float* scales, int n_entries,
                                  int B, int L, int Lp, int dm, int mode, int target, float c0, float c1, float sigma,
                                  msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || dm <= 0 || n_entries < 1 || (n_entries > 1 && !scales)) return 1;
  const long total = (long)B * L * dm;
  dim3 grid((unsigned)min((total + 255) / 256, (long)2048)), block(256);
  hipLaunchKernelGGL(cfg_ddpm_kernel, grid, block, 0, (hipStream_t)stream, x, res, z, scales, n_entries, B, L, Lp, dm,
                     mode, target, c0, c1, sigma);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// hipGraph support for the sampler: the captured step body must not take per-step host scalars, so the step
// index lives on the device.  step_select copies row t of the step-embedding table and the (c0, c1, sigma)
// triple of step t into fixed buffers and then decrements t; cfg_ddpm_dev reads the triple from there.
template <typename T>
__global__ void step_select_kernel(const T* __restrict__ emb_all, const float* __restrict__ coef_table,
                                   int* __restrict__ t_dev, T* __restrict__ emb_row, float* __restrict__ coefs, int d) {
  const int t = *t_dev;
  __syncthreads();
  for (int i = threadIdx.x; i < d; i += blockDim.x) emb_row[i] = emb_all[(long)t * d + i];
  if (threadIdx.x < 3) coefs[threadIdx.x] = coef_table[t * 3 + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) *t_dev = t - 1;
}

extern "C" int msmd_sampler_step_select(const void* emb_all, const float* coef_table, int* t_dev, void* emb_row,
                                        float* coefs, int d, int dtype, msmd_stream_t stream) {
  if (d <= 0 || !emb_all || !coef_table || !t_dev || !emb_row || !coefs) return 1;
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(step_select_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)emb_all,
                       coef_table, t_dev, (float*)emb_row, coefs, d);
  else if (dtype == MSMD_F16)
    hipLaunchKernelGGL(step_select_kernel<f16_t>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const f16_t*)emb_all,
                       coef_table, t_dev, (f16_t*)emb_row, coefs, d);
  else
    hipLaunchKernelGGL(step_select_kernel<bf16_t>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)emb_all,
                       coef_table, t_dev, (bf16_t*)emb_row, coefs, d);
  MSMD_RETURN_LAST();
}

__global__ void cfg_ddpm_dev_kernel(float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ z,
                                    const float* __restrict__ scales, const float* __restrict__ coefs, int n_entries,
                                    int B, int L, int Lp, int dm, int mode, int target) {
  const float c0 = coefs[0], c1 = coefs[1], sigma = coefs[2];
  const long total = (long)B * L * dm;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long)L * dm));
    const int rem = (int)(i % ((long)L * dm));
    const int t = rem / dm, k = rem % dm;
    const long stride_e = (long)B * (Lp + L) * dm;
    const long off = ((long)b * (Lp + L) + Lp + t) * dm + k;
    float theta = res[off];
    for (int e = 0; e < n_entries - 1; ++e) {
      const float hi = res[(e + 1) * stride_e + off];
      const float lo = (mode == 1 || e == 0) ? theta : res[e * stride_e + off];
      theta += scales[e] * (hi - lo);
    }
    const float xt = x[i];
    const float zz = z ? z[i] : 0.f;
    x[i] = (target == 0) ? (c0 * xt + c1 * theta + sigma * zz) : (c0 * (xt - c1 * theta) + sigma * zz);
  }
}

extern "C" int msmd_cfg_ddpm_step_dev(float* x, const float* res, const float* z, const float* scales,
                                      const float* coefs, int n_entries, int B, int L, int Lp, int dm, int mode,
                                      int target, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || dm <= 0 || n_entries < 1 || (n_entries > 1 && !scales) || !coefs) return 1;
  const long total = (long)B * L * dm;
  dim3 grid((unsigned)min((total + 255) / 256, (long)2048)), block(256);
  hipLaunchKernelGGL(cfg_ddpm_dev_kernel, grid, block, 0, (hipStream_t)stream, x, res, z, scales, coefs, n_entries, B,
                     L, Lp, dm, mode, target);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void pad_cols_kernel(const TI* __restrict__ x, TO* __restrict__ y, long rows, int ci, int co) {
  const long total = rows * co;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / co;
    const int c = (int)(i % co);
    y[i] = from_f32<TO>(c < ci ? to_f32(x[r * ci + c]) : 0.f);
  }
}

extern "C" int msmd_pad_cols(const void* x, void* y, long rows, int cols_in, int cols_out, int in_dtype, int out_dtype,
                             msmd_stream_t stream) {
  if (rows <= 0 || cols_in <= 0 || cols_out <= 0) return 1;
  const long total = rows * cols_out;
  dim3 grid((unsigned)min((total + 255) / 256, (long)4096)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_F32)
    hipLaunchKernelGGL((pad_cols_kernel<float, float>), grid, block, 0, st, (const float*)x, (float*)y, rows, cols_in,
                       cols_out);
  else if (in_dtype == MSMD_F32 && out_dtype == MSMD_BF16)
    hipLaunchKernelGGL((pad_cols_kernel<float, bf16_t>), grid, block, 0, st, (const float*)x, (bf16_t*)y, rows,
                       cols_in, cols_out);
  else if (in_dtype == MSMD_BF16 && out_dtype == MSMD_F32)
    hipLaunchKernelGGL((pad_cols_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)x, (float*)y, rows,
                       cols_in, cols_out);
  else if (in_dtype == MSMD_F32 && out_dtype == MSMD_F16)
    hipLaunchKernelGGL((pad_cols_kernel<float, f16_t>), grid, block, 0, st, (const float*)x, (f16_t*)y, rows,
                       cols_in, cols_out);
  else if (in_dtype == MSMD_F16 && out_dtype == MSMD_F32)
    hipLaunchKernelGGL((pad_cols_kernel<f16_t, float>), grid, block, 0, st, (const f16_t*)x, (float*)y, rows,
                       cols_in, cols_out);
  else if (in_dtype == MSMD_F16 && out_dtype == MSMD_F16)
    hipLaunchKernelGGL((pad_cols_kernel<f16_t, f16_t>), grid, block, 0, st, (const f16_t*)x, (f16_t*)y, rows,
                       cols_in, cols_out);
  else if (in_dtype == MSMD_BF16 && out_dtype == MSMD_BF16)
    hipLaunchKernelGGL((pad_cols_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, rows,
                       cols_in, cols_out);
  else
    return 1;
  MSMD_RETURN_LAST();
}

extern "C" int msmd_cast(const void* x, void* y, long n, int in_dtype, int out_dtype, msmd_stream_t stream) {
  if (n <= 0 || n > 0x7fffffffL) return 1;
  return msmd_pad_cols(x, y, 1, (int)n, (int)n, in_dtype, out_dtype, stream);
}

extern "C" int msmd_abi_version(void) { return 2; }   // 2: + msmd_comm_* / msmd_allreduce_bucket (round 4), msmd_comm_version

// ---------------------------------------------------------------------------------------------------
// Dynamic thresholding of the denoiser output (reference model.py:396-402, 578-584):
//   s_n = clamp(quantile_q(|res[n, -L:, :]|), dt_min, dt_max);   res[n] <- clamp(res[n], -s_n, s_n)
// torch.quantile's default 'linear' rule: rank = q (m - 1), lerp between the floor / ceil order statistics.
// One 1024-thread workgroup per sequence keeps the m = L C magnitudes in registers and finds the order statistic by
// bisection on the (non-negative) float bit pattern: 31 rounds of count(x <= mid) instead of a sort.
template <int MAXE>
__global__ __launch_bounds__(1024) void dynamic_threshold_kernel(float* __restrict__ res, int T_all, int L, int C,
                                                                  float q, float dt_min, float dt_max) {
  __shared__ int red[16];
  __shared__ unsigned redu[16];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float* base = res + (long)n * T_all * C;
  const float* tail = base + (long)(T_all - L) * C;
  const int m = L * C;
  unsigned v[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = e * 1024 + tid;
    v[e] = i < m ? (__float_as_uint(tail[i]) & 0x7FFFFFFFu) : 0xFFFFFFFFu;  // padding sorts last
  }
  auto count_le = [&](unsigned x) {
    int c = 0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) c += (v[e] <= x) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __syncthreads();
    if (lane == 0) red[wid] = c;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
  };
  const float rank = q * (float)(m - 1);
  const int k = (int)floorf(rank);
  const float w = rank - (float)k;
  unsigned lo = 0u, hi = 0x7F800000u;  // smallest x with count(<= x) >= k + 1
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    if (count_le(mid) >= k + 1) hi = mid; else lo = mid + 1;
  }
  const unsigned b_lo = lo;
  unsigned b_hi = b_lo;
  if (w > 0.f && count_le(b_lo) < k + 2) {  // next order statistic = smallest value above b_lo
    unsigned mn = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) mn = (v[e] > b_lo && v[e] < mn) ? v[e] : mn;
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64));
    __syncthreads();
    if (lane == 0) redu[wid] = mn;
    __syncthreads();
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) mn = min(mn, redu[ww]);
    b_hi = mn;
  }
  const float a = __uint_as_float(b_lo), b = __uint_as_float(b_hi);
  // ATen lerp: a + w (b - a) for w < 0.5, else b - (b - a)(1 - w)
  float sq = (w < 0.5f) ? a + w * (b - a) : b - (b - a) * (1.0f - w);
  sq = fminf(fmaxf(sq, dt_min), dt_max);
  for (int i = tid; i < T_all * C; i += 1024) base[i] = fminf(fmaxf(base[i], -sq), sq);
}

extern "C" int msmd_dynamic_threshold(float* res, int N, int T_all, int L, int C, float ratio, float dt_min,
                                      float dt_max, msmd_stream_t stream) {
  if (N <= 0 || L <= 0 || T_all < L || C <= 0 || !(ratio >= 0.f && ratio <= 1.f)) return 1;
  const long m = (long)L * C;
  hipStream_t st = (hipStream_t)stream;
  if (m <= 8 * 1024)
    hipLaunchKernelGGL(dynamic_threshold_kernel<8>, dim3(N), dim3(1024), 0, st, res, T_all, L, C, ratio, dt_min, dt_max);
  else if (m <= 20 * 1024)
    hipLaunchKernelGGL(dynamic_threshold_kernel<20>, dim3(N), dim3(1024), 0, st, res, T_all, L, C, ratio, dt_min, dt_max);
  else
    return 1;
  MSMD_RETURN_LAST();
}
