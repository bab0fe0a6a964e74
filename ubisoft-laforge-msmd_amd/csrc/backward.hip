// Backward-pass building blocks (training path, reference training_script.py:195 `loss.backward()`):
// transposes feeding the MFMA GEMM for dgrad / wgrad, bias-gradient column sums, activation and LayerNorm
// backward, row softmax forward / backward for the explicit (materialised-P) training attention.
// All HBM-bound streaming kernels except the GEMMs they feed.
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// Batched 2-D transpose through a padded LDS tile: y[b][c][r] = x[b][r][c].
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, T* __restrict__ y, int rows, int cols,
                                                        long ldx, long ldy, long sx, long sy, int inner, long sx2,
                                                        long sy2) {
  __shared__ T tile[64][65];
  const int zo = blockIdx.z / inner, zi = blockIdx.z % inner;
  const T* xb = x + zo * sx + zi * sx2;
  T* yb = y + zo * sy + zi * sy2;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + i * 4, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + i * 4][tx] = xb[(long)r * ldx + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + i * 4, r = r0 + tx;
    if (r < rows && c < cols) yb[(long)c * ldy + r] = tile[tx][ty + i * 4];
  }
}

extern "C" int msmd_transpose(const void* x, void* y, int rows, int cols, long ldx, long ldy, int batch, long stride_x,
                              long stride_y, int batch_inner, long stride_x_i, long stride_y_i, int dtype,
                              msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || batch <= 0 || batch_inner <= 0) return 1;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch * batch_inner), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)y, rows,
                       cols, ldx, ldy, stride_x, stride_y, batch_inner, stride_x_i, stride_y_i);
  else
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y,
                       rows, cols, ldx, ldy, stride_x, stride_y, batch_inner, stride_x_i, stride_y_i);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// out[c] (+)= sum_r x[r][c]   (bias gradients, weight-norm reductions; fp32 accumulation).  With a workspace the sum is
// DETERMINISTIC: row blocks write partial sums (no atomics), a second launch adds them in block order; without one the
// row blocks meet through float atomics (order-dependent last bits).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows,
                                                     int cols, long ld, int rows_per_block, float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  // four interleaved running sums (rows r0 + 4 i + u), added in a fixed order: the loop is a chain of loads, not bandwidth
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += to_f32(x[r * ld + c]);
    a1 += to_f32(x[(r + 1) * ld + c]);
    a2 += to_f32(x[(r + 2) * ld + c]);
    a3 += to_f32(x[(r + 3) * ld + c]);
  }
  for (; r < r1; ++r) a0 += to_f32(x[r * ld + c]);
  const float s = (a0 + a1) + (a2 + a3);
  if (partial) partial[(long)blockIdx.y * cols + c] = s;
  else atomicAdd(&out[c], s);
}
// 64 columns x 4 interleaved groups of partial rows per workgroup, the four group sums added in a fixed order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            int nblocks, int cols, int accumulate) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < cols) {
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    int b = g;
    for (; b + 12 < nblocks; b += 16) {
      b0 += partial[(long)b * cols + c];
      b1 += partial[(long)(b + 4) * cols + c];
      b2 += partial[(long)(b + 8) * cols + c];
      b3 += partial[(long)(b + 12) * cols + c];
    }
    for (; b < nblocks; b += 4) b0 += partial[(long)b * cols + c];
    s = (b0 + b1) + (b2 + b3);
  }
  red[g][cx] = s;
  __syncthreads();
  if (g == 0 && c < cols) out[c] = (accumulate ? out[c] : 0.f) + ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]));
}

extern "C" long msmd_colsum_workspace(long rows, int cols) { return ((rows + 127) / 128) * (long)cols * (long)sizeof(float); }

extern "C" int msmd_colsum(const void* x, float* out, long rows, int cols, long ld, int accumulate, int dtype,
                           void* ws, long ws_bytes, msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0) return 1;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 128;
  const int nblocks = (int)((rows + rpb - 1) / rpb);
  float* partial = (ws && ws_bytes >= msmd_colsum_workspace(rows, cols)) ? (float*)ws : nullptr;
  if (!accumulate && !partial) {
    hipError_t e = msmd_zero_async(out, (size_t)cols * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid((cols + 255) / 256, (unsigned)nblocks), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, st, (const float*)x, out, rows, cols, ld, rpb, partial);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, out, rows, cols, ld, rpb, partial);
  if (partial)
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, partial, out, nblocks, cols, accumulate ? 1 : 0);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// y = act(z)  /  dz = dy * act'(z)
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == MSMD_ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    return cdf + z * 0.3989422804014327f * expf(-0.5f * z * z);
  }
  if (act == MSMD_ACT_ELU) return z > 0.f ? 1.0f : expf(z);
  return 1.0f;
}

template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ z, T* __restrict__ y, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = from_f32<T>(apply_act(to_f32(z[i]), act));
}
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z, T* __restrict__ dz, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dz[i] = from_f32<T>(to_f32(dy[i]) * act_grad(to_f32(z[i]), act));
}

extern "C" int msmd_act_fwd(const void* z, void* y, long n, int act, int dtype, msmd_stream_t stream) {
  if (n <= 0) return 1;
  dim3 grid((unsigned)min((n + 255) / 256, (long)8192)), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(act_fwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)z, (float*)y, n, act);
  else
    hipLaunchKernelGGL(act_fwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)z, (bf16_t*)y, n, act);
  MSMD_RETURN_LAST();
}
extern "C" int msmd_act_bwd(const void* dy, const void* z, void* dz, long n, int act, int dtype, msmd_stream_t stream) {
  if (n <= 0) return 1;
  dim3 grid((unsigned)min((n + 255) / 256, (long)8192)), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(act_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)dy, (const float*)z,
                       (float*)dz, n, act);
  else
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)z, (bf16_t*)dz, n, act);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row; x is the LN INPUT (already x + residual), y = xhat*gamma + beta.
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//   dgamma += dy * xhat, dbeta += dy  (fp32 atomics; dgamma/dbeta must be zeroed or hold the running sum)
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, T* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int cols, float eps, int rows_per_block,
                                                            float* __restrict__ partial) {
  __shared__ float sg[4 * 64 * MAXC], sb[4 * 64 * MAXC];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ag[MAXC], abt[MAXC];  // this wave's running dgamma / dbeta partials over its rows
#pragma unroll
  for (int i = 0; i < MAXC; ++i) ag[i] = abt[i] = 0.f;
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
  for (int row = r_begin + wid; row < r_end; row += 4) {
    float xv[MAXC], gv[MAXC], dyv[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      xv[i] = (c < cols) ? to_f32(x[(long)row * cols + c]) : 0.f;
      dyv[i] = (c < cols) ? to_f32(dy[(long)row * cols + c]) : 0.f;
      s += xv[i];
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      const float d = (c < cols) ? xv[i] - mean : 0.f;
      q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
    float sg1 = 0.f, sg2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      xv[i] = (c < cols) ? (xv[i] - mean) * rstd : 0.f;  // xhat
      gv[i] = (c < cols) ? dyv[i] * gamma[c] : 0.f;
      sg1 += gv[i];
      sg2 += gv[i] * xv[i];
    }
    const float m1 = wave_sum(sg1) / (float)cols, m2 = wave_sum(sg2) / (float)cols;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < cols) dx[(long)row * cols + c] = from_f32<T>(rstd * (gv[i] - m1 - xv[i] * m2));
      ag[i] += dyv[i] * xv[i];
      abt[i] += dyv[i];
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    sg[(wid * MAXC + i) * 64 + lane] = ag[i];
    sb[(wid * MAXC + i) * 64 + lane] = abt[i];
  }
  __syncthreads();
  // combine the 4 waves, then ONE atomic per column per block
  for (int k = threadIdx.x; k < MAXC * 64; k += 256) {
    if (k < cols) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += sg[w * MAXC * 64 + k];
        b += sb[w * MAXC * 64 + k];
      }
      if (partial) {  // [block][2][cols]: summed by ln_partial_reduce_kernel (no same-address atomic contention)
        partial[((long)blockIdx.x * 2 + 0) * cols + k] = a;
        partial[((long)blockIdx.x * 2 + 1) * cols + k] = b;
      } else {
        atomicAdd(&dgamma[k], a);
        atomicAdd(&dbeta[k], b);
      }
    }
  }
}

__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int nblocks, int cols) {
  // 32 columns x 8 row groups per workgroup: coalesced 128-byte rows, 8-way split of the nblocks-long sum
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int which = blockIdx.y;
  // eight independent running sums per thread: the loop is a chain of L2 round trips (1 600 partial rows at 12 800 x 768: 100
  // dependent iterations with two sums = 13.8 us per launch, 55 launches per training step), not bandwidth
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (c < cols) {
    int b = rg;
    for (; b + 56 < nblocks; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += partial[((long)(b + 8 * u) * 2 + which) * cols + c];
    }
    for (; b < nblocks; b += 8) acc[0] += partial[((long)b * 2 + which) * cols + c];
  }
  red[rg][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (rg == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][cl];
    (which ? dbeta : dgamma)[c] += s;
  }
}

// 4-wide variant: lane owns columns (i * 64 + lane) * 4 + {0..3}: 8-byte (bf16) / 16-byte (fp32) accesses, and two
// rounds of two independent wave reductions (sum x, sum x^2 | sum g, sum g xhat) instead of four dependent ones.
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };
template <> struct Vec4<bf16_t> { typedef __bf16 type __attribute__((ext_vector_type(4))); };

template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd4_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ gamma, T* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int rows, int cols, float eps, int rows_per_block,
                                                             float* __restrict__ partial, T* __restrict__ dx_drop,
                                                             float p_drop, const unsigned long* __restrict__ rng_state,
                                                             unsigned site) {
  typedef typename Vec4<T>::type V4;
  __shared__ float sg[4][NCH * 256 + 4], sb[4][NCH * 256 + 4];
  const unsigned thr = dropout_threshold(p_drop);
  const float drop_c = 1.0f / (1.0f - p_drop);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ag[NCH][4], abt[NCH][4], gam[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ag[i][j] = abt[i][j] = 0.f;
      gam[i][j] = (c + j < cols) ? gamma[c + j] : 0.f;
    }
  }
  const float inv_n = 1.0f / (float)cols;
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
  for (int row = r_begin + wid; row < r_end; row += 4) {
    float xv[NCH][4], dyv[NCH][4];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < cols) {
        const V4 a = *(const V4*)(x + (long)row * cols + c);
        const V4 d = *(const V4*)(dy + (long)row * cols + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xv[i][j] = (float)a[j];
          dyv[i][j] = (float)d[j];
          s += xv[i][j];
          ss += xv[i][j] * xv[i][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[i][j] = dyv[i][j] = 0.f;
      }
    }
    // two independent butterfly reductions interleave in the pipeline
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      ss += __shfl_xor(ss, o, 64);
    }
    const float mean = s * inv_n;
    const float var = fmaxf(ss * inv_n - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    float g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = (i * 64 + lane) * 4 + j < cols;
        xv[i][j] = ok ? (xv[i][j] - mean) * rstd : 0.f;  // xhat
        const float gv = dyv[i][j] * gam[i][j];
        g1 += gv;
        g2 += gv * xv[i][j];
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      g1 += __shfl_xor(g1, o, 64);
      g2 += __shfl_xor(g2, o, 64);
    }
    const float m1 = g1 * inv_n, m2 = g2 * inv_n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 4;
      V4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = from_f32<T>(rstd * (dyv[i][j] * gam[i][j] - m1 - xv[i][j] * m2));
        ag[i][j] += dyv[i][j] * xv[i][j];
        abt[i][j] += dyv[i][j];
      }
      if (c < cols) {
        *(V4*)(dx + (long)row * cols + c) = o;
        if (dx_drop) {   // the dropped copy the producing Linear's backward wants: mask of msmd_dropout / the GEMM epilogue
                         // (index = element / 4), applied to the ROUNDED dx as the separate msmd_dropout launch would
          const Philox4 r = dropout_bits(rng_state, site, (unsigned long)(((long)row * cols + c) >> 2));
          V4 od;
          od[0] = from_f32<T>(r.x >= thr ? to_f32(o[0]) * drop_c : 0.f);
          od[1] = from_f32<T>(r.y >= thr ? to_f32(o[1]) * drop_c : 0.f);
          od[2] = from_f32<T>(r.z >= thr ? to_f32(o[2]) * drop_c : 0.f);
          od[3] = from_f32<T>(r.w >= thr ? to_f32(o[3]) * drop_c : 0.f);
          *(V4*)(dx_drop + (long)row * cols + c) = od;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sg[wid][(i * 64 + lane) * 4 + j] = ag[i][j];
      sb[wid][(i * 64 + lane) * 4 + j] = abt[i][j];
    }
  __syncthreads();
  for (int k = threadIdx.x; k < NCH * 256; k += 256) {
    if (k < cols) {
      const float a = sg[0][k] + sg[1][k] + sg[2][k] + sg[3][k];
      const float b = sb[0][k] + sb[1][k] + sb[2][k] + sb[3][k];
      if (partial) {
        partial[((long)blockIdx.x * 2 + 0) * cols + k] = a;
        partial[((long)blockIdx.x * 2 + 1) * cols + k] = b;
      } else {
        atomicAdd(&dgamma[k], a);
        atomicAdd(&dbeta[k], b);
      }
    }
  }
}

extern "C" long msmd_layernorm_bwd_workspace(int rows, int cols) {
  const int rpb = rows >= 2048 ? 16 : 8;
  return (long)((rows + rpb - 1) / rpb) * 2 * cols * (long)sizeof(float);
}

static int layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma,
                              float* dbeta, int rows, int cols, float eps, int dtype, void* ws, long ws_bytes,
                              msmd_stream_t stream, void* dx_drop, float p_drop, const unsigned long* rng_state,
                              unsigned site) {
  if (rows <= 0 || cols <= 0 || cols > 1024) return 1;
  if (dx_drop && (!(p_drop > 0.f && p_drop < 1.f) || !rng_state || (cols & 3) || ((uintptr_t)dx_drop & 15) ||
                  ((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dx & 15)))
    return 1;   // the dropped copy exists in the 4-wide kernels only
  float* partial = (ws && ws_bytes >= msmd_layernorm_bwd_workspace(rows, cols)) ? (float*)ws : nullptr;
  const int rpb = rows >= 2048 ? 16 : 8;  // rows per workgroup: 1 atomic per column per workgroup
  dim3 grid((rows + rpb - 1) / rpb), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = ((uintptr_t)dy % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dx % 16 == 0);
  if (cols % 4 == 0 && aligned) {
#define LN4(T, NCH)                                                                                                  \
  hipLaunchKernelGGL((layernorm_bwd4_kernel<T, NCH>), grid, block, 0, st, (const T*)dy, (const T*)x, gamma, (T*)dx, \
                     dgamma, dbeta, rows, cols, eps, rpb, partial, (T*)dx_drop, p_drop, rng_state, site)
    if (dtype == MSMD_F32) {
      if (cols <= 256) LN4(float, 1); else if (cols <= 512) LN4(float, 2); else if (cols <= 768) LN4(float, 3); else LN4(float, 4);
    } else {
      if (cols <= 256) LN4(bf16_t, 1); else if (cols <= 512) LN4(bf16_t, 2); else if (cols <= 768) LN4(bf16_t, 3); else LN4(bf16_t, 4);
    }
#undef LN4
  } else if (dtype == MSMD_F32) {
    if (cols <= 512)
      hipLaunchKernelGGL((layernorm_bwd_kernel<float, 8>), grid, block, 0, st, (const float*)dy, (const float*)x, gamma,
                         (float*)dx, dgamma, dbeta, rows, cols, eps, rpb, partial);
    else
      hipLaunchKernelGGL((layernorm_bwd_kernel<float, 16>), grid, block, 0, st, (const float*)dy, (const float*)x,
                         gamma, (float*)dx, dgamma, dbeta, rows, cols, eps, rpb, partial);
  } else {
    if (cols <= 512)
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 8>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                         gamma, (bf16_t*)dx, dgamma, dbeta, rows, cols, eps, rpb, partial);
    else if (cols <= 768)
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 12>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                         gamma, (bf16_t*)dx, dgamma, dbeta, rows, cols, eps, rpb, partial);
    else
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, 16>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                         gamma, (bf16_t*)dx, dgamma, dbeta, rows, cols, eps, rpb, partial);
  }
  if (partial)
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((cols + 31) / 32, 2), dim3(256), 0, st, partial, dgamma, dbeta,
                       (int)grid.x, cols);
  MSMD_RETURN_LAST();
}

extern "C" int msmd_layernorm_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma,
                                  float* dbeta, int rows, int cols, float eps, int dtype, void* ws, long ws_bytes,
                                  msmd_stream_t stream) {
  return layernorm_bwd_impl(dy, x, gamma, dx, dgamma, dbeta, rows, cols, eps, dtype, ws, ws_bytes, stream, nullptr, 0.f,
                            nullptr, 0u);
}

// msmd_layernorm_bwd that ALSO writes dx_drop = dropout_mask(dx) / (1 - p): when the LayerNorm's input is
// residual + dropout_p(Linear(..)) (every post-LN transformer block), the Linear's backward needs exactly that tensor,
// and this saves its msmd_dropout launch (mask = Philox(rng_state, site, element / 4), as the forward GEMM epilogue drew it).
extern "C" int msmd_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, void* dx, void* dx_drop,
                                          float* dgamma, float* dbeta, int rows, int cols, float eps, float p_drop,
                                          const unsigned long* rng_state, unsigned int site, int dtype, void* ws,
                                          long ws_bytes, msmd_stream_t stream) {
  if (!dx_drop) return 1;
  return layernorm_bwd_impl(dy, x, gamma, dx, dgamma, dbeta, rows, cols, eps, dtype, ws, ws_bytes, stream, dx_drop,
                            p_drop, rng_state, site);
}

// ---------------------------------------------------------------------------------------------------
// Row softmax over a (rows, cols) score matrix with scale and optional (Tq, Tk) byte mask shared by all
// batches (row r uses mask row r % Tq), in place;   and its backward  dS = scale * P o (dP - rowsum(dP o P)).
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ s, const uint8_t* __restrict__ mask, long rows,
                                                           int cols, int ld, int Tq, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  T* p = s + row * ld;
  for (int c = cols + lane; c < ld; c += 64) p[c] = from_f32<T>(0.f);  // padding columns contribute nothing
  const uint8_t* m = mask ? mask + (long)(row % Tq) * cols : nullptr;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) {
    const float v = (m && m[c]) ? -INFINITY : to_f32(p[c]) * scale;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float v = (m && m[c]) ? -INFINITY : to_f32(p[c]) * scale;
    sum += expf(v - mx);
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) {
    const float v = (m && m[c]) ? -INFINITY : to_f32(p[c]) * scale;
    p[c] = from_f32<T>(expf(v - mx) * inv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const T* __restrict__ P, T* __restrict__ dP, long rows,
                                                               int cols, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* p = P + row * ld;
  T* d = dP + row * ld;
  for (int c = cols + lane; c < ld; c += 64) d[c] = from_f32<T>(0.f);
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += to_f32(p[c]) * to_f32(d[c]);
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) d[c] = from_f32<T>(scale * to_f32(p[c]) * (to_f32(d[c]) - dot));
}

extern "C" int msmd_softmax_rows(void* s, const uint8_t* mask, long rows, int cols, int ld, int Tq, float scale,
                                 int dtype, msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || Tq <= 0 || ld < cols) return 1;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, (float*)s, mask, rows, cols, ld,
                       Tq, scale);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (bf16_t*)s, mask, rows, cols,
                       ld, Tq, scale);
  MSMD_RETURN_LAST();
}
extern "C" int msmd_softmax_bwd_rows(const void* P, void* dP, long rows, int cols, int ld, float scale, int dtype,
                                     msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || ld < cols) return 1;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)P,
                       (float*)dP, rows, cols, ld, scale);
  else
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)P,
                       (bf16_t*)dP, rows, cols, ld, scale);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Transposed unfold for the positional-conv weight gradient: out[g][kk*Cg + ci][b*T + t] = xp[b][g][t + kk][ci]
// (xp: zero-padded group-major (B, G, Tp, Cg)).  dW_g = dZ_g^T . unfold_g is then one batched MFMA GEMM.
template <typename T>
__global__ __launch_bounds__(256) void unfold_t_kernel(const T* __restrict__ xp, T* __restrict__ out, int B, int Tn,
                                                       int Tp, int G, int Cg, int Kk, long ld_out) {
  const int g = blockIdx.z;
  const int row = blockIdx.y;  // kk * Cg + ci
  const int kk = row / Cg, ci = row % Cg;
  T* o = out + ((long)g * Kk * Cg + row) * ld_out;
  const long n = (long)B * Tn;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < ld_out; i += (long)gridDim.x * blockDim.x) {
    T v = from_f32<T>(0.f);
    if (i < n) {
      const int b = (int)(i / Tn), t = (int)(i % Tn);
      v = xp[(((long)b * G + g) * Tp + t + kk) * Cg + ci];
    }
    o[i] = v;
  }
}

extern "C" int msmd_unfold_t(const void* xp, void* out, int B, int T, int Tp, int G, int Cg, int Kk, long ld_out,
                             int dtype, msmd_stream_t stream) {
  if (B <= 0 || T <= 0 || Tp < T + Kk - 1 || G <= 0 || Cg <= 0 || Kk <= 0 || ld_out < (long)B * T) return 1;
  dim3 grid((unsigned)min((ld_out + 255) / 256, (long)64), Kk * Cg, G), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(unfold_t_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)xp, (float*)out, B, T,
                       Tp, G, Cg, Kk, ld_out);
  else
    hipLaunchKernelGGL(unfold_t_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)xp, (bf16_t*)out, B,
                       T, Tp, G, Cg, Kk, ld_out);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// y = x * keep / (1 - p) (+ residual), keep ~ Bernoulli(1 - p) from Philox(seed, step, site, index / 4).
// The same call with x = dy (no residual) is the backward.  n % 4 need not hold.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ residual,
                                                      T* __restrict__ y, long n, float p,
                                                      const unsigned long* __restrict__ rng_state, unsigned site) {
  const unsigned thr = dropout_threshold(p);
  const float c = 1.0f / (1.0f - p);
  const long nq = (n + 3) >> 2;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    const Philox4 r = dropout_bits(rng_state, site, (unsigned long)q);
    const unsigned bits[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = q * 4 + j;
      if (i < n) {
        float v = bits[j] >= thr ? to_f32(x[i]) * c : 0.f;
        if (residual) v += to_f32(residual[i]);
        y[i] = from_f32<T>(v);
      }
    }
  }
}

extern "C" int msmd_dropout(const void* x, const void* residual, void* y, long n, float p,
                            const unsigned long* rng_state, unsigned int site, int dtype, msmd_stream_t stream) {
  if (n <= 0 || !(p >= 0.f && p < 1.f) || !rng_state) return 1;
  dim3 grid((unsigned)min(((n + 3) / 4 + 255) / 256, (long)8192)), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x,
                       (const float*)residual, (float*)y, n, p, rng_state, site);
  else
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)residual, (bf16_t*)y, n, p, rng_state, site);
  MSMD_RETURN_LAST();
}

// dz = dropout_mask(dy) * act'(z): the backward of y = dropout(act(z)) in ONE pass (mask regenerated from the same
// Philox stream as the forward GEMM epilogue / msmd_dropout: index = element / 4).
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_dropout_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                              T* __restrict__ dz, long n, int act, float p,
                                                              const unsigned long* __restrict__ rng, unsigned site) {
  const unsigned thr = dropout_threshold(p);
  const float c = 1.0f / (1.0f - p);
  const long nq = (n + 3) >> 2;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    const Philox4 r = dropout_bits(rng, site, (unsigned long)q);
    const unsigned bits[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = q * 4 + j;
      if (i < n) dz[i] = from_f32<T>(bits[j] >= thr ? to_f32(dy[i]) * c * act_grad(to_f32(z[i]), act) : 0.f);
    }
  }
}

extern "C" int msmd_act_bwd_dropout(const void* dy, const void* z, void* dz, long n, int act, float p,
                                    const unsigned long* rng_state, unsigned int site, int dtype, msmd_stream_t stream) {
  if (n <= 0 || !(p > 0.f && p < 1.f) || !rng_state) return 1;
  dim3 grid((unsigned)min(((n + 3) / 4 + 255) / 256, (long)8192)), block(256);
  if (dtype == MSMD_F32)
    hipLaunchKernelGGL(act_bwd_dropout_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)dy,
                       (const float*)z, (float*)dz, n, act, p, rng_state, site);
  else
    hipLaunchKernelGGL(act_bwd_dropout_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)z, (bf16_t*)dz, n, act, p, rng_state, site);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Once-per-step weight refresh for mixed-precision training: every trainable (N, K) Linear weight of the flat fp32
// parameter arena is cast to bf16 AND written transposed (K, N) (the data-gradient GEMM's operand) by ONE launch --
// the ~240 per-weight cast / transpose launches of an iteration collapse into this.  meta (n_w, 6) int64:
// [src offset in the fp32 arena, N, K, dst offset in the cast arena, dst offset in the transposed arena, first tile];
// N, K multiples of 8; 32 x 32 tiles through LDS (coalesced on both sides).
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const float* __restrict__ base,
                                                                   const long* __restrict__ meta, int n_w,
                                                                   bf16_t* __restrict__ cast, bf16_t* __restrict__ tr) {
  __shared__ float tile[32][33];
  // which weight owns this tile: binary search over the tile prefix
  int lo = 0, hi = n_w - 1;
  const long t = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (meta[mid * 6 + 5] <= t) lo = mid; else hi = mid - 1;
  }
  const long* m = meta + lo * 6;
  const long src = m[0], N = m[1], K = m[2], dc = m[3], dt = m[4];
  const long lt = t - m[5];
  const int tk = (int)((K + 31) / 32);
  const int n0 = (int)(lt / tk) * 32, k0 = (int)(lt % tk) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + ty + 8 * r, k = k0 + tx;
    float v = 0.f;
    if (n < N && k < K) {
      v = base[src + (long)n * K + k];
      cast[dc + (long)n * K + k] = (bf16_t)v;
    }
    tile[ty + 8 * r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + ty + 8 * r, n = n0 + tx;
    if (n < N && k < K) tr[dt + (long)k * N + n] = (bf16_t)tile[tx][ty + 8 * r];
  }
}

extern "C" int msmd_cast_transpose_multi(const float* base, const long* meta, int n_weights, long total_tiles,
                                         void* cast_arena, void* transposed_arena, msmd_stream_t stream) {
  if (n_weights <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffL) return 1;
  hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, base,
                     meta, n_weights, (bf16_t*)cast_arena, (bf16_t*)transposed_arena);
  MSMD_RETURN_LAST();
}
