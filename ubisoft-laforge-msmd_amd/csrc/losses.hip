// Masked sequence losses (reference utils/common.py:198-620): masked MSE / L1 on a sequence, its first
// temporal difference (velocity) or second difference (smoothness), averaged over a channel range and over the
// valid frames.  One generic reduction kernel serves the parameter-space losses (C = 67) and the vertex-space
// ones (C = 15069); HBM-bound streaming reads, fp64 accumulation across workgroups.
#include "common.h"

struct LossArgs {
  const float* gt; const float* pred; const int* end_idx; double* acc;
  int N, T, C, c_lo, c_hi, order, prefix, criterion, vs_zero, mode;
};

__device__ __forceinline__ float diff_at(const float* p, long row_stride, int order) {
  if (order == 0) return p[0];
  if (order == 1) return p[row_stride] - p[0];
  return (p[2 * row_stride] - p[row_stride]) - (p[row_stride] - p[0]);
}

// mode 0: value = crit(D gt, D pred); mode 1 (smooth): crit(D pred, 0) (reference compares against zeros, or,
// in the vertex-space variant, vel_pred[1:] against vel_pred[:-1], which is the same quantity).
__global__ __launch_bounds__(256) void masked_seq_loss_kernel(const LossArgs p) {
  __shared__ float red[16];
  const int Td = p.T - p.order;
  const int nc = p.c_hi - p.c_lo;
  const long rows = (long)p.N * Td;
  double local_sum = 0.0;
  double local_cnt = 0.0;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = (int)(row / Td), t = (int)(row % Td);
    const int tm = t + p.order;  // mask index: mask[:, order:]
    // prefix >= 0: the first `prefix` frames (previous window) always count; prefix < 0: they never do
    // (--no_constrain_prev, reference utils/common.py:382-385)
    const int pf = p.prefix < 0 ? -p.prefix : p.prefix;
    bool valid = p.prefix > 0 && tm < pf;
    if (!valid && tm >= pf) {
      const int e = p.end_idx ? p.end_idx[n] : (p.T - pf);
      valid = (tm - pf) < e;
    }
    if (!valid) continue;  // block-uniform
    const float* g = p.gt + ((long)n * p.T + t) * p.C + p.c_lo;
    const float* q = p.pred + ((long)n * p.T + t) * p.C + p.c_lo;
    float s = 0.f;
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
      const float dp = diff_at(q + c, p.C, p.order);
      const float dg = p.mode == 1 ? 0.f : diff_at(g + c, p.C, p.order);
      const float d = dg - dp;
      s += p.criterion == 0 ? d * d : fabsf(d);
    }
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) {
      local_sum += (double)tot / (double)nc;
      local_cnt += 1.0;
    }
  }
  if (threadIdx.x == 0 && local_cnt > 0.0) {
    atomicAdd(&p.acc[0], local_sum);
    atomicAdd(&p.acc[1], local_cnt);
  }
}

// Narrow rows (coefficient space: 4-67 columns): four waves per workgroup, each takes rows of its own and every lane keeps ONE
// running sum over all its (row, column) elements -- no shuffle and no barrier inside the row loop, so the loads of several rows
// are in flight at once; one wave reduction and one workgroup reduction at the end, then the two atomics.  (The kernel above
// spends a barrier-delimited block reduction per row: 44 us for 3 168 rows of 50 columns, almost all of it latency.)
__global__ __launch_bounds__(256) void masked_seq_loss_narrow_kernel(const LossArgs p) {
  __shared__ double sred[4][2];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int Td = p.T - p.order;
  const int nc = p.c_hi - p.c_lo;
  const long rows = (long)p.N * Td;
  const int pf = p.prefix < 0 ? -p.prefix : p.prefix;
  float s = 0.f;
  int cnt = 0;
#pragma unroll 4
  for (long row = blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
    const int n = (int)(row / Td), t = (int)(row % Td);
    const int tm = t + p.order;
    bool valid = p.prefix > 0 && tm < pf;
    if (!valid && tm >= pf) {
      const int e = p.end_idx ? p.end_idx[n] : (p.T - pf);
      valid = (tm - pf) < e;
    }
    if (!valid) continue;  // wave-uniform
    cnt += 1;
    const float* g = p.gt + ((long)n * p.T + t) * p.C + p.c_lo;
    const float* q = p.pred + ((long)n * p.T + t) * p.C + p.c_lo;
    for (int c = lane; c < nc; c += 64) {
      const float dp = diff_at(q + c, p.C, p.order);
      const float dg = p.mode == 1 ? 0.f : diff_at(g + c, p.C, p.order);
      const float d = dg - dp;
      s += p.criterion == 0 ? d * d : fabsf(d);
    }
  }
  const float tot = wave_sum(s);
  if (lane == 0) { sred[wid][0] = (double)tot / (double)nc; sred[wid][1] = (double)cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a = (sred[0][0] + sred[1][0]) + (sred[2][0] + sred[3][0]), c = (sred[0][1] + sred[1][1]) + (sred[2][1] + sred[3][1]);
    if (c > 0.0) {
      atomicAdd(&p.acc[0], a);
      atomicAdd(&p.acc[1], c);
    }
  }
}

// No valid row: the plain term is the mean of an empty selection (NaN, as torch's); the velocity / smoothness terms are
// None -> 0 in the reference when every sample is too short for a difference (utils/common.py:571-583), e.g. a window
// whose samples are all truncated at end_idx == 1.
__global__ void loss_finish_kernel(const double* acc, float* out, float scale, int empty_is_zero) {
  out[0] = acc[1] > 0.0 ? (float)(acc[0] / acc[1] * (double)scale) : (empty_is_zero ? 0.f : nanf(""));
}

extern "C" int msmd_masked_seq_loss(const float* gt, const float* pred, const int* end_idx, float* out, double* acc_ws,
                                    int N, int T, int C, int c_lo, int c_hi, int order, int prefix, int criterion,
                                    int mode, float scale, msmd_stream_t stream) {
  if (N <= 0 || T <= order || C <= 0 || c_lo < 0 || c_hi > C || c_hi <= c_lo || order < 0 || order > 2 || !acc_ws)
    return 1;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = msmd_zero_async(acc_ws, 2 * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  LossArgs p{gt, pred, end_idx, acc_ws, N, T, C, c_lo, c_hi, order, prefix, criterion, 0, mode};
  const long rows = (long)N * (T - order);
  const int threads = (c_hi - c_lo) >= 1024 ? 256 : 64;
  // every workgroup ends with two double atomics on ONE address pair: size the grid by the work (about 4096 elements
  // per workgroup), not by the rows -- coefficient-space calls (67 columns) took 75 us with one workgroup per row
  const long want = rows * (long)(c_hi - c_lo) / 4096;
  dim3 grid((unsigned)max((long)1, min(rows, min(max(want, (long)64), (long)4096)))), block(threads);
  if (c_hi - c_lo <= 256)      // coefficient space
    hipLaunchKernelGGL(masked_seq_loss_narrow_kernel, dim3((unsigned)max((long)1, min((long)64, (rows + 3) / 4))), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(masked_seq_loss_kernel, grid, block, 0, st, p);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, acc_ws, out, scale, order > 0 ? 1 : 0);
  MSMD_RETURN_LAST();
}

// Gradient of msmd_masked_seq_loss with respect to `pred`, ADDED into grad_pred (N, T, C; the channels outside
// [c_lo, c_hi) are untouched):  loss = scale / (cnt nc) sum_{valid rows t} sum_c crit(D gt - D pred)  (mode 1: D gt = 0), D =
// order-th temporal difference with stencil s = (1), (-1, 1), (1, -2, 1), so
//   d loss / d pred[n, tau, c] = - scale up / (cnt nc) * sum_k s_k crit'(d[n, tau - k, c])   over the valid rows tau - k.
// cnt is read from the forward's workspace (acc_ws[1]); `upstream` is the incoming 0-dim gradient on the device.
__global__ __launch_bounds__(256) void masked_seq_loss_bwd_kernel(const LossArgs p, float* __restrict__ grad,
                                                                  const float* __restrict__ upstream, float scale,
                                                                  int rows_per_wg) {
  // narrow rows (rows_per_wg = 4, launched when nc <= 256): one wave per (n, tau) row, four rows per workgroup -- 3 520 workgroups
  // of one wave each were dispatch-bound (18 us for 235 k elements); wide rows (vertex space, nc = 15 069): rows_per_wg = 1,
  // all 256 threads stride one row
  const int n = blockIdx.y, tau = blockIdx.x * rows_per_wg + (rows_per_wg > 1 ? (int)(threadIdx.x >> 6) : 0);
  if (tau >= p.T) return;
  const int c_first = rows_per_wg > 1 ? (threadIdx.x & 63) : threadIdx.x, c_step = rows_per_wg > 1 ? 64 : blockDim.x;
  const int nc = p.c_hi - p.c_lo, Td = p.T - p.order;
  const double cnt = p.acc[1];
  if (!(cnt > 0.0)) return;
  const float g0 = -scale * upstream[0] / ((float)cnt * (float)nc);
  const int pf = p.prefix < 0 ? -p.prefix : p.prefix;
  // stencil (1) / (-1, 1) / (1, -2, 1) by order, as three scalars: a local array indexed by the run-time order lives in scratch
  // memory (every use a scratch load: 18-31 us for 235 k elements)
  const float st0 = p.order == 1 ? -1.f : 1.f, st1 = p.order == 0 ? 0.f : (p.order == 1 ? 1.f : -2.f), st2 = p.order == 2 ? 1.f : 0.f;
  bool valid[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int t = tau - k;
    bool v = k <= p.order && t >= 0 && t < Td;
    if (v) {
      const int tm = t + p.order;
      bool ok = p.prefix > 0 && tm < pf;
      if (!ok && tm >= pf) {
        const int e = p.end_idx ? p.end_idx[n] : (p.T - pf);
        ok = (tm - pf) < e;
      }
      v = ok;
    }
    valid[k] = v;
  }
  if (!valid[0] && !valid[1] && !valid[2]) return;
  for (int c = c_first; c < nc; c += c_step) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!valid[k]) continue;
      const long off = ((long)n * p.T + (tau - k)) * p.C + p.c_lo + c;
      const float dp = diff_at(p.pred + off, p.C, p.order);
      const float dg = p.mode == 1 ? 0.f : diff_at(p.gt + off, p.C, p.order);
      const float d = dg - dp;
      const float dc = p.criterion == 0 ? 2.0f * d : (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f));
      a = fmaf(k == 0 ? st0 : (k == 1 ? st1 : st2), dc, a);
    }
    grad[((long)n * p.T + tau) * p.C + p.c_lo + c] += g0 * a;
  }
}

extern "C" int msmd_masked_seq_loss_bwd(const float* gt, const float* pred, const int* end_idx, const double* acc_ws,
                                        const float* upstream, float* grad_pred, int N, int T, int C, int c_lo, int c_hi,
                                        int order, int prefix, int criterion, int mode, float scale,
                                        msmd_stream_t stream) {
  if (N <= 0 || T <= order || C <= 0 || c_lo < 0 || c_hi > C || c_hi <= c_lo || order < 0 || order > 2 || !acc_ws ||
      !upstream || !grad_pred)
    return 1;
  LossArgs p{gt, pred, end_idx, (double*)acc_ws, N, T, C, c_lo, c_hi, order, prefix, criterion, 0, mode};
  if (c_hi - c_lo <= 256)
    hipLaunchKernelGGL(masked_seq_loss_bwd_kernel, dim3((T + 3) / 4, N), dim3(256), 0, (hipStream_t)stream, p, grad_pred, upstream,
                       scale, 4);
  else
    hipLaunchKernelGGL(masked_seq_loss_bwd_kernel, dim3(T, N), dim3((c_hi - c_lo) >= 1024 ? 256 : 64), 0, (hipStream_t)stream, p,
                       grad_pred, upstream, scale, 1);
  MSMD_RETURN_LAST();
}

// KL divergence of the style VAE (reference utils/common.py:443-454): -0.5 * sum(1 + logvar - mu^2 - exp(logvar)).
__global__ void kl_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, double* acc, long n) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += 1.0f + logvar[i] - mu[i] * mu[i] - expf(logvar[i]);
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(&acc[0], (double)tot);
}
__global__ void kl_finish_kernel(const double* acc, float* out) { out[0] = (float)(-0.5 * acc[0]); }

extern "C" int msmd_kl_loss(const float* mu, const float* logvar, float* out, double* acc_ws, long n,
                            msmd_stream_t stream) {
  if (n <= 0 || !acc_ws) return 1;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = msmd_zero_async(acc_ws, 2 * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kl_kernel, dim3((unsigned)min((n + 255) / 256, (long)256)), dim3(256), 0, st, mu, logvar, acc_ws, n);
  hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(1), 0, st, acc_ws, out);
  MSMD_RETURN_LAST();
}

// Truncation augmentation (reference utils/common.py:769-798): x[n, end[n]*unit:] = 0 or the last kept value.
__global__ void truncate_rows_kernel(float* __restrict__ x, const int* __restrict__ end_idx, int L, int inner, int unit,
                                     int replicate) {
  const int n = blockIdx.y;
  const long e = (long)end_idx[n] * unit;  // first overwritten index along L
  const long total = (long)L * inner;
  float* row = x + (long)n * total;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long l = i / inner;
    if (l >= e) row[i] = replicate ? row[(e - 1) * inner + i % inner] : 0.f;
  }
}

extern "C" int msmd_truncate_rows(float* x, const int* end_idx, int N, int L, int inner, int unit, int replicate,
                                  msmd_stream_t stream) {
  if (N <= 0 || L <= 0 || inner <= 0 || unit <= 0) return 1;
  const long total = (long)L * inner;
  dim3 grid((unsigned)min((total + 255) / 256, (long)1024), N), block(256);
  hipLaunchKernelGGL(truncate_rows_kernel, grid, block, 0, (hipStream_t)stream, x, end_idx, L, inner, unit, replicate);
  MSMD_RETURN_LAST();
}

// Fused multi-tensor Adam step on a flat parameter / gradient / moment arena (torch.optim.Adam semantics,
// reference training_script.py:548-551: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                         float bc1, float bc2, float grad_scale) {
  const float gi = g * grad_scale;
  const float mi = b1 * m + (1.0f - b1) * gi;
  const float vi = b2 * v + (1.0f - b2) * gi * gi;
  m = mi;
  v = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p -= (lr / bc1) * (mi / denom);
}
// 16 bytes per lane and stream (7 streams: read p, g, m, v; write p, m, v = 28 B per parameter): HBM-bound
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2, float grad_scale) {
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = ((f32x4*)p)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
    const f32x4 gg = ((const f32x4*)g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      adam_one(pe, gg[e], me, ve, lr, b1, b2, eps, bc1, bc2, grad_scale);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    adam_one(p[i], g[i], m[i], v[i], lr, b1, b2, eps, bc1, bc2, grad_scale);
}

extern "C" int msmd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                              float beta1, float beta2, float eps, int step, float grad_scale, msmd_stream_t stream) {
  if (n <= 0 || step <= 0) return 1;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return 1;
  dim3 grid((unsigned)min((n / 4 + 255) / 256 + 1, (long)8192)), block(256);
  hipLaunchKernelGGL(adam_kernel, grid, block, 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, bc1, bc2, grad_scale);
  MSMD_RETURN_LAST();
}
