// Audio front end: pad_audio gather map, conv0 + GroupNorm + GELU, linear resampling, group regrouping.
// All HBM-bound streaming kernels; the raw audio is tiny, the conv0 activation (B, 12815, 512) is the
// only large tensor and is written exactly once, channels-last, already normalised and activated.
#include "common.h"

// Source sample of padded position j for pad_audio = reflect(r) o reflect(r) o replicate(rep)
// (reference utils/model_common.py:110-123).  Pure integer maths, bit-exact.
__device__ __forceinline__ int pad_src(int j, int L, int r, int rep) {
  const int L1 = L + 2 * r, L2 = L1 + 2 * r;
  j -= rep;
  j = j < 0 ? 0 : (j >= L2 ? L2 - 1 : j);  // replicate
  if (r > 0) {
    j -= r;                                  // second reflect (input length L1)
    j = j < 0 ? -j : (j >= L1 ? 2 * (L1 - 1) - j : j);
    j -= r;                                  // first reflect (input length L)
    j = j < 0 ? -j : (j >= L ? 2 * (L - 1) - j : j);
  }
  return j;
}

__global__ void pad_audio_kernel(const float* __restrict__ a, float* __restrict__ out, int L, int Lp, int r, int rep) {
  const int b = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Lp; j += gridDim.x * blockDim.x)
    out[(long)b * Lp + j] = a[(long)b * L + pad_src(j, L, r, rep)];
}

extern "C" int msmd_pad_audio(const float* audio, float* out, int B, int L, int reflect_len, int replicate_len,
                              msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || reflect_len < 0 || replicate_len < 0 || reflect_len >= L) return 1;
  const int Lp = L + 4 * reflect_len + 2 * replicate_len;
  dim3 grid(min((Lp + 255) / 256, 1024), B), block(256);
  hipLaunchKernelGGL(pad_audio_kernel, grid, block, 0, (hipStream_t)stream, audio, out, L, Lp, reflect_len,
                     replicate_len);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// conv0 statistics: partial (mean, M2) per (batch, time split, channel), merged with Chan's formula.
#define C0_K 10
#define C0_S 5
#define C0_SPLITS 16

__global__ __launch_bounds__(256) void conv0_stats_partial(const float* __restrict__ audio,
                                                           const float* __restrict__ w0, float* __restrict__ ws,
                                                           int L, int r, int rep, int C, int T0) {
  // block = 64 channels x 4 time lanes; blockIdx = (channel group, split, batch)
  __shared__ float xs[4096];
  __shared__ float red[4][64][2];
  const int b = blockIdx.z, split = blockIdx.y, cg = blockIdx.x;
  const int c = cg * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const int per = (T0 + C0_SPLITS - 1) / C0_SPLITS;
  const int t_begin = split * per, t_end = min(T0, t_begin + per);
  float w[C0_K];
#pragma unroll
  for (int k = 0; k < C0_K; ++k) w[k] = (c < C) ? w0[c * C0_K + k] : 0.f;
  // Welford-free two-pass inside the split, processed in LDS chunks of 800 frames (4005 samples)
  const int CH = 800;
  float sum = 0.f;
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) {
    float acc = 0.f;
    float mean = 0.f;
    if (pass == 1) {
      // combine the four time lanes' sums -> split mean (all lanes of a channel see the same value)
      __syncthreads();
      red[sl][threadIdx.x & 63][0] = sum;
      __syncthreads();
      const int cnt = t_end - t_begin;
      mean = (red[0][threadIdx.x & 63][0] + red[1][threadIdx.x & 63][0] + red[2][threadIdx.x & 63][0] +
              red[3][threadIdx.x & 63][0]) / (float)max(cnt, 1);
    }
    for (int t0 = t_begin; t0 < t_end; t0 += CH) {
      const int nt = min(CH, t_end - t0);
      const int ns = nt * C0_S + (C0_K - C0_S);
      __syncthreads();
      for (int i = threadIdx.x; i < ns; i += 256) xs[i] = audio[(long)b * L + pad_src(t0 * C0_S + i, L, r, rep)];
      __syncthreads();
      for (int t = sl; t < nt; t += 4) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < C0_K; ++k) y = fmaf(w[k], xs[t * C0_S + k], y);
        if (pass == 0) acc += y; else { const float d = y - mean; acc = fmaf(d, d, acc); }
      }
    }
    if (pass == 0) sum = acc;
    else {
      __syncthreads();
      red[sl][threadIdx.x & 63][1] = acc;
      __syncthreads();
      if (sl == 0 && c < C) {
        const float m2 = red[0][threadIdx.x][1] + red[1][threadIdx.x][1] + red[2][threadIdx.x][1] + red[3][threadIdx.x][1];
        float* o = ws + (((long)b * C0_SPLITS + split) * C + c) * 2;
        o[0] = mean;
        o[1] = m2;
      }
    }
  }
  (void)n;
}

__global__ void conv0_stats_merge(const float* __restrict__ ws, float* __restrict__ stats, int C, int T0, float eps) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int per = (T0 + C0_SPLITS - 1) / C0_SPLITS;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int s = 0; s < C0_SPLITS; ++s) {
    const int cnt = min(T0, (s + 1) * per) - min(T0, s * per);
    if (cnt <= 0) continue;
    const float* p = ws + (((long)b * C0_SPLITS + s) * C + c) * 2;
    const double mb = p[0], m2b = p[1], nb = cnt;
    const double delta = mb - mean, tot = n + nb;
    mean += delta * nb / tot;
    m2 += m2b + delta * delta * n * nb / tot;
    n = tot;
  }
  stats[((long)b * C + c) * 2 + 0] = (float)mean;
  stats[((long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(m2 / n + (double)eps));
}

extern "C" int msmd_conv0_stats(const float* audio, const float* w0, float* stats, float* ws, int B, int L,
                                int reflect_len, int replicate_len, int C, float eps, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || C <= 0 || !ws) return 1;
  const int Lp = L + 4 * reflect_len + 2 * replicate_len;
  const int T0 = (Lp - C0_K) / C0_S + 1;
  if (T0 <= 0) return 1;
  dim3 grid((C + 63) / 64, C0_SPLITS, B);
  hipLaunchKernelGGL(conv0_stats_partial, grid, dim3(256), 0, (hipStream_t)stream, audio, w0, ws, L, reflect_len,
                     replicate_len, C, T0);
  hipLaunchKernelGGL(conv0_stats_merge, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, ws, stats, C,
                     T0, eps);
  MSMD_RETURN_LAST();
}

// out (B, T0, C) = GELU(GN(conv0)); thread = 4 consecutive channels, block = 2 frame rows x 128 channel quads
template <typename TO>
__global__ __launch_bounds__(256) void conv0_gn_gelu_kernel(const float* __restrict__ audio,
                                                            const float* __restrict__ w0,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ out,
                                                            int L, int r, int rep, int C, int T0) {
  constexpr int FR = 64;  // frames per block
  __shared__ float xs[FR * C0_S + C0_K];
  const int b = blockIdx.y, t0 = blockIdx.x * FR;
  const int nt = min(FR, T0 - t0);
  const int ns = nt * C0_S + (C0_K - C0_S);
  for (int i = threadIdx.x; i < ns; i += 256) xs[i] = audio[(long)b * L + pad_src(t0 * C0_S + i, L, r, rep)];
  __syncthreads();
  const int half = threadIdx.x >> 7, q = threadIdx.x & 127;
  for (int c0 = q * 4; c0 < C; c0 += 512) {
    float w[4][C0_K], sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
#pragma unroll
      for (int k = 0; k < C0_K; ++k) w[e][k] = w0[c * C0_K + k];
      const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
      sc[e] = rstd * gamma[c];
      sh[e] = beta[c] - mean * sc[e];
    }
    for (int t = half; t < nt; t += 2) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < C0_K; ++k) y = fmaf(w[e][k], xs[t * C0_S + k], y);
        o[e] = gelu_erf(fmaf(y, sc[e], sh[e]));
      }
      TO* op = out + ((long)b * T0 + t0 + t) * C + c0;
      if constexpr (sizeof(TO) == 4) *(f32x4*)op = f32x4{o[0], o[1], o[2], o[3]};
      else *(bf16x4*)op = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
    }
  }
}

extern "C" int msmd_conv0_gn_gelu(const float* audio, const float* w0, const float* stats, const float* gamma,
                                  const float* beta, void* out, int B, int L, int reflect_len, int replicate_len,
                                  int C, int out_dtype, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || C <= 0 || (C & 3)) return 1;
  const int Lp = L + 4 * reflect_len + 2 * replicate_len;
  const int T0 = (Lp - C0_K) / C0_S + 1;
  dim3 grid((T0 + 63) / 64, B), block(256);
  if (out_dtype == MSMD_F32)
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<float>, grid, block, 0, (hipStream_t)stream, audio, w0, stats, gamma, beta,
                       (float*)out, L, reflect_len, replicate_len, C, T0);
  else
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, audio, w0, stats, gamma,
                       beta, (bf16_t*)out, L, reflect_len, replicate_len, C, T0);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// F.interpolate(mode='linear', align_corners=False) along time, channels-last.
// src = fmaf(scale, dst + 0.5, -0.5) clamped at 0 (ATen's fused form; see oracle/nn.py), i0 = floor, w1 = src - i0.
template <typename T>
__global__ void interp_linear_kernel(const T* __restrict__ x, T* __restrict__ y, int T_in, int T_crop, int T_out,
                                     int C) {
  const int b = blockIdx.z, j = blockIdx.y;
  const float scale = (float)T_crop / (float)T_out;
  float src = fmaf(scale, (float)j + 0.5f, -0.5f);
  src = src < 0.f ? 0.f : src;
  int i0 = (int)src;
  i0 = i0 < T_crop - 1 ? i0 : T_crop - 1;
  const int i1 = i0 + 1 < T_crop ? i0 + 1 : T_crop - 1;
  const float w1 = src - (float)i0, w0 = 1.0f - w1;
  const T* p0 = x + ((long)b * T_in + i0) * C;
  const T* p1 = x + ((long)b * T_in + i1) * C;
  T* q = y + ((long)b * T_out + j) * C;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x)
    q[c] = from_f32<T>(w0 * to_f32(p0[c]) + w1 * to_f32(p1[c]));
}

extern "C" int msmd_interp_linear(const void* x, void* y, int B, int T_in, int T_crop, int T_out, int C, int dtype,
                                  msmd_stream_t stream) {
  if (B <= 0 || T_in <= 0 || T_crop <= 0 || T_crop > T_in || T_out <= 0 || C <= 0) return 1;
  dim3 grid((C + 255) / 256, T_out, B), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(interp_linear_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)y,
                       T_in, T_crop, T_out, C);
  else
    hipLaunchKernelGGL(interp_linear_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, T_in, T_crop, T_out, C);
  MSMD_RETURN_LAST();
}

// (B, T, G*Cg) -> zero-padded group-major (B, G, T + 2*pad, Cg)
template <typename T>
__global__ void group_pad_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, int G, int Cg, int pad) {
  const int b = blockIdx.z, g = blockIdx.y;
  const int Tp = Tn + 2 * pad;
  const long n = (long)Tp * Cg;
  T* yo = y + ((long)b * G + g) * n;
  for (long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tp = (int)(i / Cg), c = (int)(i % Cg);
    const int t = tp - pad;
    yo[i] = (t >= 0 && t < Tn) ? x[((long)b * Tn + t) * (G * Cg) + g * Cg + c] : from_f32<T>(0.f);
  }
}

extern "C" int msmd_group_pad(const void* x, void* y, int B, int T, int G, int Cg, int pad, int dtype,
                              msmd_stream_t stream) {
  if (B <= 0 || T <= 0 || G <= 0 || Cg <= 0 || pad < 0) return 1;
  const long n = (long)(T + 2 * pad) * Cg;
  dim3 grid((unsigned)min((n + 255) / 256, (long)64), G, B), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(group_pad_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)y, T, G,
                       Cg, pad);
  else
    hipLaunchKernelGGL(group_pad_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, T,
                       G, Cg, pad);
  MSMD_RETURN_LAST();
}
