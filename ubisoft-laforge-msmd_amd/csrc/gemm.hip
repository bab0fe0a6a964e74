// MFMA GEMM for gfx950:  C = act(A . W^T + bias) + residual
//
// One kernel template serves every dense contraction on the audio->motion path: HF conv stack as a
// windowed GEMM over the channels-last signal (no im2col), Linear layers, attention projections,
// grouped positional conv (batched), style-encoder convs.  Both operands are K-contiguous
// ((M,K) activations, (N,K) torch-layout weights), so A and W tiles are staged identically.
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2x2), K step = 128 BYTES per row
// (64 bf16 / 32 fp32) so the LDS image and the staging code are byte-identical for both dtypes.
// LDS rows are 128 B with the 16-B chunk index XOR-swizzled by ((row>>1)&7): conflict-free for the
// real ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS table).  Global->register
// prefetch of tile k+1 is issued before the MFMAs of tile k; two LDS buffers, one barrier per K step.
// Operands are swapped (D = W_tile . X_tile^T) so each lane ends up with 4 CONSECUTIVE output
// columns of one row: 16-B (fp32) / 8-B (bf16) epilogue stores and a float4 bias load.
// bf16: v_mfma_f32_16x16x32_bf16; fp32 parity mode: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include "common.h"

struct GemmArgs {
  const void* A; const void* W; const float* bias; const void* R; void* C;
  int M, N, K;
  long lda; int rows_per_batch; long a_batch_stride; long ldw, ldc, ldr;
  int act; int vec_ok;
  long strideA, strideW, strideC, strideBias, strideR;
  int mt, nt;
};

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  // one 16-B chunk = 8 bf16 = the lane's K-slice of one 16x16x32 MFMA
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  // one 16-B chunk = 4 fp32; lane (l>>4) owns k = 4*(l>>4)+e in MFMA step e (K is permuted
  // identically for both operands, which leaves the contraction unchanged).
  // NB: __builtin_bit_cast on a vector ELEMENT reads element 0 (hipcc 7.2); convert by value instead.
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
  }
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, typename TO, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
  constexpr int E = 16 / sizeof(T);     // elements per 16-B chunk
  constexpr int BKE = 128 / sizeof(T);  // K elements per tile
  constexpr int LA = BM / 32, LW = BN / 32;  // 16-B loads per thread per tile
  constexpr int FM = BM / 32, FN = BN / 32;  // 16x16 fragments per wave (wave tile = BM/2 x BN/2)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * 128];

  // XCD-aware tile order: the nt column tiles of one row panel run back to back on ONE XCD
  // (block ids congruent mod 8 share an XCD's L2), so the A panel is fetched from HBM once.
  const int pid = blockIdx.x;
  const int xcd = pid & 7, slot = pid >> 3;
  const int m_tile = (slot / p.nt) * 8 + xcd, n_tile = slot % p.nt;
  if (m_tile >= p.mt) return;
  const int z = blockIdx.z;
  const T* __restrict__ A = (const T*)p.A + z * p.strideA;
  const T* __restrict__ W = (const T*)p.W + z * p.strideW;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // per-thread staging assignments (fixed across K steps)
  const T* a_ptr[LA]; bool a_ok[LA]; int a_lds[LA];
  const T* w_ptr[LW]; bool w_ok[LW]; int w_lds[LW];
  const int cc = tid & 7;
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int row = (tid >> 3) + i * 32;
    const int m = m0 + row;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    a_ptr[i] = A + (long)(mm / p.rows_per_batch) * p.a_batch_stride + (long)(mm % p.rows_per_batch) * p.lda + cc * E;
    a_lds[i] = lds_off(row, cc);
  }
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int row = (tid >> 3) + i * 32;
    const int n = n0 + row;
    w_ok[i] = n < p.N;
    w_ptr[i] = W + (long)(w_ok[i] ? n : 0) * p.ldw + cc * E;
    w_lds[i] = BM * 128 + lds_off(row, cc);
  }

  u32x4 ra[LA], rw[LW];
  auto load_global = [&](int k0) {
    const bool kin = (k0 + cc * E) < p.K;
#pragma unroll
    for (int i = 0; i < LA; ++i)
      ra[i] = (a_ok[i] && kin) ? *(const u32x4*)(a_ptr[i] + k0) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < LW; ++i)
      rw[i] = (w_ok[i] && kin) ? *(const u32x4*)(w_ptr[i] + k0) : u32x4{0, 0, 0, 0};
  };
  auto store_lds = [&](int buf) {
    unsigned char* base = smem + buf * (BM + BN) * 128;
#pragma unroll
    for (int i = 0; i < LA; ++i) *(u32x4*)(base + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < LW; ++i) *(u32x4*)(base + w_lds[i]) = rw[i];
  };

  const int wm = (wid >> 1) * (BM / 2), wn = (wid & 1) * (BN / 2);
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BKE - 1) / BKE;
  load_global(0);
  store_lds(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_global((kt + 1) * BKE);
    const unsigned char* sa = smem + cur * (BM + BN) * 128;
    const unsigned char* sw = sa + BM * 128;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      u32x4 fx[FM], fw[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) fx[j] = *(const u32x4*)(sa + lds_off(wm + j * 16 + fr, g * 4 + fq));
#pragma unroll
      for (int i = 0; i < FN; ++i) fw[i] = *(const u32x4*)(sw + lds_off(wn + i * 16 + fr, g * 4 + fq));
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) Mfma<T>::run(fw[i], fx[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_lds(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds row m = ..+fr, columns n = ..+fq*4 + {0..3}
  TO* __restrict__ C = (TO*)p.C + z * p.strideC;
  const TO* __restrict__ R = p.R ? (const TO*)p.R + z * p.strideR : nullptr;
  const float* __restrict__ bias = p.bias ? p.bias + z * p.strideBias : nullptr;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0 + wn + i * 16 + fq * 4;
    if (n >= p.N) continue;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? bias[n + e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm + j * 16 + fr;
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][e] + bv[e], p.act);
      TO* cp = C + (long)m * p.ldc + n;
      if (p.vec_ok && n + 3 < p.N) {
        if (R) {
          const TO* rp = R + (long)m * p.ldr + n;
          if constexpr (sizeof(TO) == 4) {
            const f32x4 r = *(const f32x4*)rp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          } else {
            const bf16x4 r = *(const bf16x4*)rp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
          }
        }
        if constexpr (sizeof(TO) == 4) {
          *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          *(bf16x4*)cp = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e < p.N) {
            float o = v[e];
            if (R) o += to_f32(R[(long)m * p.ldr + n + e]);
            cp[e] = from_f32<TO>(o);
          }
        }
      }
    }
  }
}

template <typename T, typename TO>
static int launch_gemm(GemmArgs& p, int batch, hipStream_t st) {
  // Small-N / small-M problems use the 64x64 tile (less padding waste, more workgroups).
  const bool small = (p.N <= 64) || ((long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch < 128);
  if (small) {
    p.mt = (p.M + 63) / 64; p.nt = (p.N + 63) / 64;
    dim3 grid(((p.mt + 7) / 8) * 8 * p.nt, 1, batch);
    hipLaunchKernelGGL((gemm_kernel<T, TO, 64, 64>), grid, dim3(256), 0, st, p);
  } else {
    p.mt = (p.M + 127) / 128; p.nt = (p.N + 127) / 128;
    dim3 grid(((p.mt + 7) / 8) * 8 * p.nt, 1, batch);
    hipLaunchKernelGGL((gemm_kernel<T, TO, 128, 128>), grid, dim3(256), 0, st, p);
  }
  MSMD_RETURN_LAST();
}

extern "C" int msmd_gemm(const void* A, const void* W, const float* bias, const void* residual, void* C, int M,
                         int N, int K, int in_dtype, int out_dtype, long lda, int rows_per_batch,
                         long a_batch_stride, long ldw, long ldc, long ldr, int act, int batch, long strideA,
                         long strideW, long strideC, long strideBias, long strideR, msmd_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || !A || !W || !C) return 1;
  const int E = in_dtype == MSMD_BF16 ? 8 : 4;
  if (K % E || lda % E || ldw % E || a_batch_stride % E || strideA % E || strideW % E) return 1;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return 1;
  if (rows_per_batch <= 0) rows_per_batch = M;
  GemmArgs p;
  p.A = A; p.W = W; p.bias = bias; p.R = residual; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.rows_per_batch = rows_per_batch; p.a_batch_stride = a_batch_stride;
  p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
  p.strideA = strideA; p.strideW = strideW; p.strideC = strideC; p.strideBias = strideBias; p.strideR = strideR;
  const int osz = out_dtype == MSMD_BF16 ? 2 : 4;
  p.vec_ok = (ldc % 4 == 0) && (strideC % 4 == 0) && (((uintptr_t)C % (4 * osz)) == 0) &&
             (!residual || ((ldr % 4 == 0) && (strideR % 4 == 0) && (((uintptr_t)residual % (4 * osz)) == 0)));
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == MSMD_BF16 && out_dtype == MSMD_BF16) return launch_gemm<bf16_t, bf16_t>(p, batch, st);
  if (in_dtype == MSMD_BF16 && out_dtype == MSMD_F32) return launch_gemm<bf16_t, float>(p, batch, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_F32) return launch_gemm<float, float>(p, batch, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_BF16) return launch_gemm<float, bf16_t>(p, batch, st);
  return 1;
}
