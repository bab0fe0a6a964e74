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
// conv0 statistics for GroupNorm(C groups, C channels) WITHOUT evaluating the conv per channel:
// y_c[t] = sum_k w_c[k] x[5t+k] is linear in x, so over the T0 frames of a clip
//   sum_t y_c     = w_c . S          S[k]     = sum_t xc[5t+k]
//   sum_t y_c^2   = w_c^T R w_c      R[k][k'] = sum_t xc[5t+k] xc[5t+k']      (10 + 55 numbers per clip)
// with xc = x - mu the clip-centred signal (a constant shift leaves the variance unchanged and removes the
// E[y^2] - mean^2 cancellation for signals with a DC offset).  190x fewer FLOPs than the direct form; the
// partial sums are fp32 per thread over <= 64 frames, then fp64 across threads/splits.
#define C0_K 10
#define C0_S 5
#define C0_SPLITS 16
#define C0_NMOM 66  // 1 (sum x) + 10 (S) + 55 (upper triangle of R)

__global__ __launch_bounds__(256) void conv0_moments_partial(const float* __restrict__ audio, float* __restrict__ ws,
                                                             int L, int r, int rep, int T0, int pass) {
  // ws[b][split][0] = the shift mu; [1..] = S and R of the shifted signal over this split's frames.
  __shared__ double red[4][C0_NMOM];
  const int b = blockIdx.y, split = blockIdx.x;
  const int per = (T0 + C0_SPLITS - 1) / C0_SPLITS;
  const int t_begin = split * per, t_end = min(T0, t_begin + per);
  float* out = ws + ((long)b * C0_SPLITS + split) * C0_NMOM;
  const float* xa = audio + (long)b * L;
  // Centre of the moments: the mean of the clip's first min(256, n) padded samples, computed by EVERY block in the same
  // order (all splits of a clip must shift by the same constant; any constant near the mean removes the
  // E[y^2] - mean^2 cancellation, the finish kernel's algebra is exact for whatever shift was used).  One launch instead
  // of the former mean pass + moments pass.
  const int n_used = T0 * C0_S + C0_K - C0_S;
  {
    double v = threadIdx.x < min(256, n_used) ? (double)xa[pad_src(threadIdx.x, L, r, rep)] : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0] = v;
    __syncthreads();
  }
  const float mu = (float)((red[0][0] + red[1][0] + red[2][0] + red[3][0]) / (double)min(256, n_used));
  if (threadIdx.x == 0) out[0] = mu;
  float m[C0_NMOM];
#pragma unroll
  for (int i = 0; i < C0_NMOM; ++i) m[i] = 0.f;
  for (int t = t_begin + threadIdx.x; t < t_end; t += 256) {
    float x[C0_K];
#pragma unroll
    for (int k = 0; k < C0_K; ++k) x[k] = xa[pad_src(t * C0_S + k, L, r, rep)] - mu;
#pragma unroll
    for (int k = 0; k < C0_K; ++k) {
      m[1 + k] += x[k];
#pragma unroll
      for (int k2 = k; k2 < C0_K; ++k2) {
        const int idx = 1 + C0_K + k * C0_K - (k * (k - 1)) / 2 + (k2 - k);  // upper-triangle index, compile time
        m[idx] = fmaf(x[k], x[k2], m[idx]);
      }
    }
  }
#pragma unroll
  for (int i = 1; i < C0_NMOM; ++i) {
    double v = (double)m[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x >= 1 && threadIdx.x < C0_NMOM)
    out[threadIdx.x] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void conv0_stats_finish(const float* __restrict__ ws, const float* __restrict__ w0,
                                   float* __restrict__ stats, int C, int T0, float eps) {
  __shared__ double mom[C0_NMOM];
  const int b = blockIdx.y;
  if (threadIdx.x < C0_NMOM) {
    double v = 0.0;
    for (int s = 0; s < C0_SPLITS; ++s) v += (double)ws[((long)b * C0_SPLITS + s) * C0_NMOM + threadIdx.x];
    mom[threadIdx.x] = v;
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = (double)ws[(long)b * C0_SPLITS * C0_NMOM];      // the shift every split used
  double w[C0_K], wsum = 0.0, ws1 = 0.0, q = 0.0;
  for (int k = 0; k < C0_K; ++k) { w[k] = (double)w0[c * C0_K + k]; wsum += w[k]; ws1 += w[k] * mom[1 + k]; }
  int idx = 1 + C0_K;
  for (int k = 0; k < C0_K; ++k)
    for (int k2 = k; k2 < C0_K; ++k2) q += (k2 == k ? 1.0 : 2.0) * w[k] * w[k2] * mom[idx++];
  const double mc = ws1 / T0;                 // mean of the centred response
  const double var = q / T0 - mc * mc;
  stats[((long)b * C + c) * 2 + 0] = (float)(mc + mu * wsum);
  stats[((long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(fmax(var, 0.0) + (double)eps));
}

extern "C" int msmd_conv0_stats(const float* audio, const float* w0, float* stats, float* ws, int B, int L,
                                int reflect_len, int replicate_len, int C, float eps, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || C <= 0 || !ws) return 1;
  const int Lp = L + 4 * reflect_len + 2 * replicate_len;
  const int T0 = (Lp - C0_K) / C0_S + 1;
  if (T0 <= 0) return 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(C0_SPLITS, B);
  hipLaunchKernelGGL(conv0_moments_partial, grid, dim3(256), 0, st, audio, ws, L, reflect_len, replicate_len, T0, 1);
  hipLaunchKernelGGL(conv0_stats_finish, dim3((C + 255) / 256, B), dim3(256), 0, st, ws, w0, stats, C, T0, eps);
  MSMD_RETURN_LAST();
}

// out (B, T0, C) = GELU(GN(conv0)); thread = 4 consecutive channels, block = 2 frame rows x 128 channel quads
// SPLIT: out is MSMD_F16X2 split storage (TO = f16_t): hi / lo 16-byte stores
template <typename TO, bool SPLIT = false>
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
  // 16-byte stores: a lane owns 8 consecutive channels, a wave one full 512-channel frame row (1 KB contiguous)
  const int grp = threadIdx.x >> 6, q = threadIdx.x & 63;
  for (int c0 = q * 8; c0 < C; c0 += 512) {
    float w[8][C0_K], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
#pragma unroll
      for (int k = 0; k < C0_K; ++k) w[e][k] = w0[c * C0_K + k];
      const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
      sc[e] = rstd * gamma[c];
      sh[e] = beta[c] - mean * sc[e];
    }
    for (int t = grp; t < nt; t += 4) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < C0_K; ++k) y = fmaf(w[e][k], xs[t * C0_S + k], y);
        // 16-bit and split outputs: the 12-instruction erf (|abs err| <= 1.5e-7, as in the split GEMM epilogues); fp32: libm
        o[e] = sizeof(TO) == 2 ? gelu_fast(fmaf(y, sc[e], sh[e])) : gelu_erf(fmaf(y, sc[e], sh[e]));
      }
      TO* op = out + ((long)b * T0 + t0 + t) * C + c0;
      if constexpr (SPLIT) {
        store8_split((f16_t*)out + ((long)b * T0 + t0 + t) * 2 * C, c0, o);
      } else if constexpr (sizeof(TO) == 4) {
        *(f32x4*)op = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4*)(op + 4) = f32x4{o[4], o[5], o[6], o[7]};
      } else {
        *(typename Vec8T<TO>::type*)op = typename Vec8T<TO>::type{(TO)o[0], (TO)o[1], (TO)o[2], (TO)o[3],
                                                                 (TO)o[4], (TO)o[5], (TO)o[6], (TO)o[7]};
      }
    }
  }
}

// 16-bit outputs: the conv on the matrix pipe.  The kernel above is bound by its vector instructions (about 30 per output:
// 10 conv FMAs, scale / shift, the erf GELU, the conversion) -- 210 M outputs x 30 / 39 T lane-ops/s = 160 us at B = 32.
// Here one v_mfma_f32_16x16x32 yields 16 channels x 16 frames: the 10 taps sit in the 32-deep K as hi / lo pairs of
// BOTH operands (x = xh + xl, w = wh + wl in the output's 16-bit type; xh.wh + xl.wh + xh.wl = 30 slots, the dropped
// xl.wl term is 2^-16 relative: far below the output's own rounding), so the vector ALU is left with scale / shift +
// GELU (gelu_poly16: no transcendentals) + conversion (about 14 per output).
//   K slots (8 per lane group fq):  g0 = xh[0..7].wh[0..7]   g1 = xl[0..7].wh[0..7]   g2 = xh[0..7].wl[0..7]
//                                   g3 = [xh8 xh9 xl8 xl9 xh8 xh9 0 0] . [wh8 wh9 wh8 wh9 wl8 wl9 0 0]
//   Weight operand rows are permuted so that a lane ends up with 8 CONSECUTIVE channels of its frame after two MFMAs
//   (fragment pair P, Q over 32 channels: row 4 q + e of P = channel 8 q + e, of Q = channel 8 q + 4 + e): one 16-byte
//   store per lane and pair, 64-byte runs per frame.
// Block = 4 waves x 32 frames; the weight image (C x 64 B) and the per-channel scale / shift live in LDS.
template <typename TO>
__global__ __launch_bounds__(256) void conv0_mfma_kernel(const float* __restrict__ audio, const float* __restrict__ w0,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, TO* __restrict__ out, int L, int r,
                                                         int rep, int C, int T0) {
  constexpr int FR = 128, FW = 2;           // frames per block, frame fragments (of 16) per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char c0_smem[];
  float* xs = (float*)c0_smem;                                  // FR * 5 + 5 samples (+ pad to 16 B)
  constexpr int XS_BYTES = ((FR * C0_S + C0_K) * 4 + 15) & ~15;
  unsigned char* wimg = c0_smem + XS_BYTES;                     // C rows x 64 B: the MFMA weight operand, fragment-major
  float* ssc = (float*)(wimg + (long)C * 64);                   // C scale, C shift
  float* ssh = ssc + C;
  const int b = blockIdx.y, t0 = blockIdx.x * FR;
  const int nt = min(FR, T0 - t0);
  const int ns = nt * C0_S + (C0_K - C0_S);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < ns; i += 256) xs[i] = audio[(long)b * L + pad_src(t0 * C0_S + i, L, r, rep)];
  for (int i = ns + tid; i < FR * C0_S + C0_K; i += 256) xs[i] = 0.f;
  // weight image: image row R = 16 j + rho (fragment j, operand row rho = 4 q + e) holds channel
  //   32 (j / 2) + 8 q + 4 (j & 1) + e
  for (int R = tid; R < C; R += 256) {
    const int j = R >> 4, rho = R & 15, c = 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
    TO wh[C0_K], wl[C0_K];
#pragma unroll
    for (int k = 0; k < C0_K; ++k) {
      const float w = w0[c * C0_K + k];
      wh[k] = (TO)w;
      wl[k] = (TO)(w - (float)wh[k]);
    }
    typedef typename Vec8T<TO>::type V8;
    V8* dst = (V8*)(wimg + (long)R * 64);
    dst[0] = V8{wh[0], wh[1], wh[2], wh[3], wh[4], wh[5], wh[6], wh[7]};
    dst[1] = dst[0];
    dst[2] = V8{wl[0], wl[1], wl[2], wl[3], wl[4], wl[5], wl[6], wl[7]};
    dst[3] = V8{wh[8], wh[9], wh[8], wh[9], wl[8], wl[9], (TO)0.f, (TO)0.f};
  }
  for (int c = tid; c < C; c += 256) {
    const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
    const float sc = rstd * gamma[c];
    ssc[c] = sc;
    ssh[c] = beta[c] - mean * sc;
  }
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  // signal operands of this wave's FW frame fragments
  u32x4 xop[FW];
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    const float* xp = xs + ((wid * FW + f) * 16 + fr) * C0_S;
    TO xh[C0_K], xl[C0_K];
#pragma unroll
    for (int k = 0; k < C0_K; ++k) {
      const float x = xp[k];
      xh[k] = (TO)x;
      xl[k] = (TO)(x - (float)xh[k]);
    }
    typedef typename Vec8T<TO>::type V8;
    const V8 g0 = V8{xh[0], xh[1], xh[2], xh[3], xh[4], xh[5], xh[6], xh[7]};
    const V8 g1 = V8{xl[0], xl[1], xl[2], xl[3], xl[4], xl[5], xl[6], xl[7]};
    const V8 g3 = V8{xh[8], xh[9], xl[8], xl[9], xh[8], xh[9], (TO)0.f, (TO)0.f};
    const V8 sel = fq == 0 ? g0 : (fq == 1 ? g1 : (fq == 2 ? g0 : g3));
    xop[f] = __builtin_bit_cast(u32x4, sel);
  }
  for (int jj = 0; jj < C / 32; ++jj) {
    const u32x4 wP = *(const u32x4*)(wimg + (long)((2 * jj) * 16 + fr) * 64 + fq * 16);
    const u32x4 wQ = *(const u32x4*)(wimg + (long)((2 * jj + 1) * 16 + fr) * 64 + fq * 16);
    const int c0 = 32 * jj + 8 * fq;
    const f32x4 sc0 = *(const f32x4*)(ssc + c0), sc1 = *(const f32x4*)(ssc + c0 + 4);
    const f32x4 sh0 = *(const f32x4*)(ssh + c0), sh1 = *(const f32x4*)(ssh + c0 + 4);
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 yP = mfma16<TO>(wP, xop[f], z), yQ = mfma16<TO>(wQ, xop[f], z);
      const int t = (wid * FW + f) * 16 + fr;
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = gelu_poly16(fmaf(yP[e], sc0[e], sh0[e]));
        o[4 + e] = gelu_poly16(fmaf(yQ[e], sc1[e], sh1[e]));
      }
      if (t < nt)
        *(typename Vec8T<TO>::type*)(out + ((long)b * T0 + t0 + t) * C + c0) =
            typename Vec8T<TO>::type{(TO)o[0], (TO)o[1], (TO)o[2], (TO)o[3], (TO)o[4], (TO)o[5], (TO)o[6], (TO)o[7]};
    }
  }
}

extern "C" int msmd_conv0_gn_gelu(const float* audio, const float* w0, const float* stats, const float* gamma,
                                  const float* beta, void* out, int B, int L, int reflect_len, int replicate_len,
                                  int C, int out_dtype, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || C <= 0 || (C & 7)) return 1;
  const int Lp = L + 4 * reflect_len + 2 * replicate_len;
  const int T0 = (Lp - C0_K) / C0_S + 1;
  dim3 grid((T0 + 63) / 64, B), block(256);
  if (out_dtype == MSMD_F16X2) {
    if (C & 31) return 1;
    hipLaunchKernelGGL((conv0_gn_gelu_kernel<f16_t, true>), grid, block, 0, (hipStream_t)stream, audio, w0, stats, gamma,
                       beta, (f16_t*)out, L, reflect_len, replicate_len, C, T0);
  } else if (out_dtype == MSMD_F32)
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<float>, grid, block, 0, (hipStream_t)stream, audio, w0, stats, gamma, beta,
                       (float*)out, L, reflect_len, replicate_len, C, T0);
  else if ((C & 31) == 0 && C <= 1024) {
    // 16-bit outputs: the conv on the matrix pipe (conv0_mfma_kernel)
    const size_t lds = (((128 * C0_S + C0_K) * 4 + 15) & ~15) + (size_t)C * 64 + (size_t)C * 8;
    dim3 g2((T0 + 127) / 128, B);
    if (out_dtype == MSMD_F16) {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)conv0_mfma_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
      hipLaunchKernelGGL(conv0_mfma_kernel<f16_t>, g2, block, lds, (hipStream_t)stream, audio, w0, stats, gamma, beta,
                         (f16_t*)out, L, reflect_len, replicate_len, C, T0);
    } else {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)conv0_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
      hipLaunchKernelGGL(conv0_mfma_kernel<bf16_t>, g2, block, lds, (hipStream_t)stream, audio, w0, stats, gamma, beta,
                         (bf16_t*)out, L, reflect_len, replicate_len, C, T0);
    }
  } else if (out_dtype == MSMD_F16)
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, audio, w0, stats, gamma,
                       beta, (f16_t*)out, L, reflect_len, replicate_len, C, T0);
  else
    hipLaunchKernelGGL(conv0_gn_gelu_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, audio, w0, stats, gamma,
                       beta, (bf16_t*)out, L, reflect_len, replicate_len, C, T0);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// conv0 of the feat_extract_norm="layer" stack (hubert-large / wav2vec2-large: HF HubertLayerNormConvLayer):
//   y[t][c] = GELU(LayerNorm_c(bias[c] + sum_k w[c][k] x[5 t + k]))      LayerNorm over the C = 512 channels of a frame
// One wave per frame: a lane owns 8 consecutive channels (weights in registers, loaded once per wave), the LN
// statistics are two wave reductions, the frame leaves as one 16-byte (bf16) or two 16-byte (fp32) stores per lane.
// pad_audio is fused into the signal loads exactly as in the GroupNorm variant.
template <typename TO>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ audio, const float* __restrict__ w0,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ out, int L,
                                                            int r, int rep, int T0, float eps) {
  constexpr int FR = 64, C = 512;
  __shared__ float xs[FR * C0_S + C0_K];
  const int b = blockIdx.y, t0 = blockIdx.x * FR;
  const int nt = min(FR, T0 - t0);
  const int ns = nt * C0_S + (C0_K - C0_S);
  for (int i = threadIdx.x; i < ns; i += 256) xs[i] = audio[(long)b * L + pad_src(t0 * C0_S + i, L, r, rep)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c0 = lane * 8;
  float w[8][C0_K], bs[8], g[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int k = 0; k < C0_K; ++k) w[e][k] = w0[(c0 + e) * C0_K + k];
    bs[e] = bias ? bias[c0 + e] : 0.f;
    g[e] = gamma[c0 + e];
    be[e] = beta[c0 + e];
  }
  for (int t = wid; t < nt; t += 4) {
    float y[8], s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = bs[e];
#pragma unroll
      for (int k = 0; k < C0_K; ++k) a = fmaf(w[e][k], xs[t * C0_S + k], a);
      y[e] = a;
      s += a;
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) q += (y[e] - mean) * (y[e] - mean);
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (y[e] - mean) * rstd * g[e] + be[e];
      o[e] = sizeof(TO) == 2 ? gelu_poly16(v) : gelu_erf(v);
    }
    TO* op = out + ((long)b * T0 + t0 + t) * C + c0;
    if constexpr (sizeof(TO) == 4) {
      *(f32x4*)op = f32x4{o[0], o[1], o[2], o[3]};
      *(f32x4*)(op + 4) = f32x4{o[4], o[5], o[6], o[7]};
    } else {
      *(typename Vec8T<TO>::type*)op = typename Vec8T<TO>::type{(TO)o[0], (TO)o[1], (TO)o[2], (TO)o[3],
                                                               (TO)o[4], (TO)o[5], (TO)o[6], (TO)o[7]};
    }
  }
}

extern "C" int msmd_conv0_ln_gelu(const float* audio, const float* w0, const float* bias, const float* gamma,
                                  const float* beta, void* out, int B, int L, int reflect_len, int replicate_len,
                                  int C, float eps, int out_dtype, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || C != 512) return 1;
  const int Lp = L + 4 * reflect_len + 2 * replicate_len;
  const int T0 = (Lp - C0_K) / C0_S + 1;
  dim3 grid((T0 + 63) / 64, B), block(256);
  if (out_dtype == MSMD_F32)
    hipLaunchKernelGGL(conv0_ln_gelu_kernel<float>, grid, block, 0, (hipStream_t)stream, audio, w0, bias, gamma, beta,
                       (float*)out, L, reflect_len, replicate_len, T0, eps);
  else if (out_dtype == MSMD_F16)
    hipLaunchKernelGGL(conv0_ln_gelu_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, audio, w0, bias, gamma, beta,
                       (f16_t*)out, L, reflect_len, replicate_len, T0, eps);
  else
    hipLaunchKernelGGL(conv0_ln_gelu_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, audio, w0, bias, gamma, beta,
                       (bf16_t*)out, L, reflect_len, replicate_len, T0, eps);
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
  else if (dtype == MSMD_F16)
    hipLaunchKernelGGL(interp_linear_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const f16_t*)x,
                       (f16_t*)y, T_in, T_crop, T_out, C);
  else
    hipLaunchKernelGGL(interp_linear_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, T_in, T_crop, T_out, C);
  MSMD_RETURN_LAST();
}

// (B, T, G*Cg) -> zero-padded group-major (B, G, T + 2*pad, Cgo) with Cgo >= Cg (extra channels zero).  Output in
// the input's dtype, or (fp32 input, Cgo % 32 == 0) in MSMD_F16X2 split storage for the parity-grade speed mode, where
// the grouped positional conv reads its overlapping windows in place and 32-element blocks must stay whole (48 -> 64).
template <typename T, bool SPLIT>
__global__ void group_pad_kernel(const T* __restrict__ x, void* __restrict__ yv, int Tn, int G, int Cg, int Cgo,
                                 int pad) {
  const int b = blockIdx.z, g = blockIdx.y;
  const int Tp = Tn + 2 * pad;
  const long n = (long)Tp * Cgo;
  for (long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tp = (int)(i / Cgo), c = (int)(i % Cgo);
    const int t = tp - pad;
    const bool in = t >= 0 && t < Tn && c < Cg;
    if constexpr (SPLIT) {
      f16_t hi, lo;
      split_f16x2(in ? to_f32(x[((long)b * Tn + t) * (G * Cg) + g * Cg + c]) : 0.f, hi, lo);
      f16_t* row = (f16_t*)yv + (((long)b * G + g) * Tp + tp) * 2 * Cgo + split_col(c);
      row[0] = hi;
      row[32] = lo;
    } else {
      T* yo = (T*)yv + ((long)b * G + g) * n;
      yo[i] = in ? x[((long)b * Tn + t) * (G * Cg) + g * Cg + c] : from_f32<T>(0.f);
    }
  }
}

extern "C" int msmd_group_pad(const void* x, void* y, int B, int T, int G, int Cg, int Cg_out, int pad, int dtype,
                              int out_dtype, msmd_stream_t stream) {
  if (B <= 0 || T <= 0 || G <= 0 || Cg <= 0 || Cg_out < Cg || pad < 0) return 1;
  const long n = (long)(T + 2 * pad) * Cg_out;
  dim3 grid((unsigned)min((n + 255) / 256, (long)64), G, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MSMD_F16X2) {
    if (dtype != MSMD_F32 || (Cg_out & 31)) return 1;
    hipLaunchKernelGGL((group_pad_kernel<float, true>), grid, block, 0, st, (const float*)x, y, T, G, Cg, Cg_out, pad);
  } else if (out_dtype != dtype) {
    return 1;
  } else if (dtype == MSMD_F32)
    hipLaunchKernelGGL((group_pad_kernel<float, false>), grid, block, 0, st, (const float*)x, y, T, G, Cg, Cg_out, pad);
  else if (dtype == MSMD_F16)
    hipLaunchKernelGGL((group_pad_kernel<f16_t, false>), grid, block, 0, st, (const f16_t*)x, y, T, G, Cg, Cg_out, pad);
  else
    hipLaunchKernelGGL((group_pad_kernel<bf16_t, false>), grid, block, 0, st, (const bf16_t*)x, y, T, G, Cg, Cg_out, pad);
  MSMD_RETURN_LAST();
}
