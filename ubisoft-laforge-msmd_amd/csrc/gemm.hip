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
#include <utility>

struct GemmArgs {
  const void* A; const void* W; const float* bias; const void* R; void* C;
  int M, N, K;
  long lda; int rows_per_batch; long a_batch_stride; long ldw, ldc, ldr;
  int act; int vec_ok; float inv_rpb;
  long strideA, strideW, strideC, strideBias, strideR;
  int mt, nt;
  int batch_inner; long strideA2, strideW2, strideC2;  // z = outer * batch_inner + inner (attention: batch x heads)
  // training epilogue (msmd_gemm_ex): optional pre-activation copy Z (layout of C) and dropout on act(.) before the
  // residual add; the keep mask is Philox(rng, site, (m * N + n) / 4), i.e. msmd_dropout's on a contiguous (M, N) C
  void* Z; float p_drop; const unsigned long* rng; unsigned site;
  int xn;  // XCDs along N (1, 2 or 4): the 8 XCDs form an (8 / xn) x xn grid over (M tiles, N tiles)
  int flags;  // MSMD_GEMM_* bits 16.. of `act`, shifted down: 1 = write-through (sc1) output stores, 2 = paired 16-B stores;
              // 8 (internal, msmd_gemm_actbwd) = Z is an INPUT: C = keep_mask / (1 - p) * act'(Z) * (A W^T), no bias / residual
  // LayerNorm folded into the GEMMs around it (msmd_gemm_ln; all NULL for plain calls):
  //   a_stats (a_nt, M, 2): A holds UN-normalised rows u, the partial (sum, sum of squares) of each row over 64-column
  //     slabs; W carries gamma folded in, w_colsum[n] = sum_k W'[n][k], bias carries beta . W:  y = rstd (acc - mu s[n]) + c[n]
  //   r_stats (r_nt, M, 2) + r_gamma / r_beta (N): the residual operand holds un-normalised rows: R <- LN(R) on the fly
  //   stats_out (N / slab, M, 2): partial (sum, sum of squares) of the STORED (rounded) output rows, per 64-column slab
  const float* a_stats = nullptr; int a_nt = 0; const float* w_colsum = nullptr;
  const float* r_stats = nullptr; int r_nt = 0; const float* r_gamma = nullptr; const float* r_beta = nullptr;
  float* stats_out = nullptr; float ln_eps = 1e-5f;
  // flags bit 2 (MSMD_GEMM_STAGGER): the workgroups dispatched second onto their CU start `stagger_ticks` (100 MHz) late
  int stagger_ticks = 0;
  // developer build only (msmd_exp_set_stamps): per-workgroup time stamps of the plain-epilogue path, 8 longs per workgroup
  long* stamps = nullptr;
};

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  // one 16-B chunk = 8 bf16 = the lane's K-slice of one 16x16x32 MFMA
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mfma<f16_t> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
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

// Row m of the (possibly windowed) A operand -> element offset.  Plain GEMMs skip the division; windowed
// (conv) rows use an fp32 reciprocal with an exact fix-up (m < 2^24), not a 40-instruction integer divide.
__device__ __forceinline__ long a_row_offset(const GemmArgs& p, int m) {
  if (p.rows_per_batch >= p.M) return (long)m * p.lda;
  int q = (int)((float)m * p.inv_rpb);
  int r = m - q * p.rows_per_batch;
  if (r < 0) { q -= 1; r += p.rows_per_batch; }
  if (r >= p.rows_per_batch) { q += 1; r -= p.rows_per_batch; }
  return (long)q * p.a_batch_stride + (long)r * p.lda;
}

// Epilogue shared by both kernels: lane holds row m = ..+fr, columns n = ..+fq*4 + {0..3} of each 16x16 fragment.
template <typename TO> __device__ __forceinline__ float act_out(float x, int act) {
  // 16-bit outputs: the transcendental-free GELU (|err| <= 5.5e-5, common.h) is below their own rounding;
  // fp32 outputs (parity mode) use the exact erff.
  if (act == MSMD_ACT_GELU) return sizeof(TO) == 2 ? gelu_poly16(x) : gelu_erf(x);
  if (act == MSMD_ACT_ELU) return elu1(x);
  return x;
}
// ACT >= 0: the activation is a compile-time constant of the KERNEL (the launcher picks the instantiation from p.act).  With
// the run-time code tested per ELEMENT the compiler keeps a scalar compare-and-branch around each of a lane's 32 outputs
// (737 branches in the 128 x 128 kernel; per-workgroup stamps put its bias-only epilogue at 2.7-3.0 us of a 15 us tile, almost
// all of it taken branches and instruction fetch); three copies behind ONE test inside the kernel spill.  ACT < 0: run-time.
template <typename TO, int ACT> __device__ __forceinline__ float act_out_c(float x, int act) {
  if constexpr (ACT == MSMD_ACT_GELU) return sizeof(TO) == 2 ? gelu_poly16(x) : gelu_erf(x);
  else if constexpr (ACT == MSMD_ACT_ELU) return elu1(x);
  else if constexpr (ACT == MSMD_ACT_NONE) return x;
  else return act_out<TO>(x, act);
}

// Output stores.  wt = write-through (`sc1`): the bytes leave the XCD's L2 for memory as they are stored instead of
// waiting dirty for the end-of-kernel write-back (MI355X_MICROARCH.md, "stores of each flavour"), so a consumer kernel
// on another XCD never depends on that write-back having completed.
__device__ __forceinline__ void store16_wt(void* p, const u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store8_wt(void* p, const u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <typename TO> __device__ __forceinline__ void store4_out(TO* p, const float* v, bool wt) {
  if constexpr (sizeof(TO) == 4) {
    const f32x4 o = f32x4{v[0], v[1], v[2], v[3]};
    if (wt) store16_wt(p, __builtin_bit_cast(u32x4, o));
    else *(f32x4*)p = o;
  } else {
    const typename Vec4T<TO>::type o = pack4<TO>(v[0], v[1], v[2], v[3]);
    if (wt) store8_wt(p, __builtin_bit_cast(u32x2, o));
    else *(typename Vec4T<TO>::type*)p = o;
  }
}

// Interior tiles (fully inside M x N, vector-aligned): straight-line code, no per-element bounds checks.
// AB: the kernel also serves msmd_gemm_actbwd (flags bit 3).  Only the LDS-DMA 16-bit kernels carry that code: in the v1 /
// fp32 kernels it cost 272 bytes of scratch (fp32 mode 23.8 -> 34 ms).
// LEAN: the inference epilogue only (no pre-activation copy, no dropout, no activation backward): the launcher sends calls that
// carry those to the kernels that compile them in.  On the 16-fragment wave tiles of gemm4_kernel the full epilogue is
// tens of thousands of instructions and pushed the accumulators into scratch.
template <typename TO, int FM, int FN, bool AB, bool LEAN, int ACT>
__device__ __forceinline__ void gemm_epilogue_interior_a(const GemmArgs& p, const f32x4 (&acc)[FN][FM], int z, int m_base,
                                                         int n_base, int fr, int fq) {
  TO* __restrict__ C = (TO*)p.C + (z / p.batch_inner) * p.strideC + (z % p.batch_inner) * p.strideC2 +
                       (long)(m_base + fr) * p.ldc + n_base + fq * 4;
  const TO* __restrict__ R = p.R ? (const TO*)p.R + z * p.strideR + (long)(m_base + fr) * p.ldr + n_base + fq * 4 : nullptr;
  const float* __restrict__ bias = p.bias ? p.bias + z * p.strideBias + n_base + fq * 4 : nullptr;
  const bool wt = p.flags & 1;
  f32x4 bv[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) bv[i] = bias ? *(const f32x4*)(bias + i * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
  // every residual fragment is requested BEFORE the first output store: loads issued between the stores are serialised
  // into one L2 round trip per fragment (load, wait, store, load, ...), 8 of them on a 128 x 128 tile
  // (fp32 outputs: one fragment row at a time -- all of them would take the 128 x 128 kernel past 128 registers)
  typedef typename Vec4T<TO>::type V4;
  constexpr int RJ = sizeof(TO) == 2 ? FM : 1;
  V4 rr[RJ][FN];
  auto load_residual = [&](int j0) {
    if (R) {
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int i = 0; i < FN; ++i) rr[j][i] = *(const V4*)(R + (long)(j0 + j) * 16 * p.ldr + i * 16);
    }
  };
  if constexpr (RJ == FM) load_residual(0);
  // the finished 4 values of fragment (i, j): bias, optional pre-activation copy, activation, dropout, residual
  auto finish = [&](int i, int j, float (&v)[4]) {
    TO* crow = C + (long)j * 16 * p.ldc;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bv[i][e];
    if (AB && (p.flags & 8)) {    // the backward of y = dropout(act(z)) applied to this data gradient: z read where C goes
      const V4 z4 = *(const V4*)((const TO*)p.Z + (crow + i * 16 - (TO*)p.C));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= act_grad_fast((float)z4[e], ACT >= 0 ? ACT : p.act);
    } else {
      if (!LEAN && p.Z) store4_out<TO>((TO*)p.Z + (crow + i * 16 - (TO*)p.C), v, wt);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_out_c<TO, ACT>(v[e], p.act);
    }
    if (!LEAN && p.p_drop > 0.f) {
      const long idx = (long)(m_base + j * 16 + fr) * p.N + (n_base + i * 16 + fq * 4);
      const Philox4 rb = dropout_bits(p.rng, p.site, (unsigned long)(idx >> 2));
      const unsigned thr = dropout_threshold(p.p_drop);
      const float c = 1.0f / (1.0f - p.p_drop);
      v[0] = rb.x >= thr ? v[0] * c : 0.f;
      v[1] = rb.y >= thr ? v[1] * c : 0.f;
      v[2] = rb.z >= thr ? v[2] * c : 0.f;
      v[3] = rb.w >= thr ? v[3] * c : 0.f;
    }
    if (R) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)rr[j % RJ][i][e];
    }
  };
  if constexpr (sizeof(TO) == 2 && FM % 2 == 0) {
    if (p.flags & 2) {
      // Paired 16-byte stores: lanes l and l ^ 16 hold columns 4 fq .. 4 fq + 3 and the next four of the same rows.
      // They swap one fragment row each (even fq keeps row j0 and takes the partner's half of it, odd fq keeps row
      // j0 + 1), so every lane issues ONE 16-byte store per fragment-row pair instead of two 8-byte ones.
      const bool odd = fq & 1;
#pragma unroll
      for (int jp = 0; jp < FM / 2; ++jp) {
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          float v0[4], v1[4];
          finish(i, 2 * jp, v0);
          finish(i, 2 * jp + 1, v1);
          const u32x2 a = __builtin_bit_cast(u32x2, pack4<TO>(v0[0], v0[1], v0[2], v0[3]));
          const u32x2 b = __builtin_bit_cast(u32x2, pack4<TO>(v1[0], v1[1], v1[2], v1[3]));
          const u32x2 send = odd ? a : b;
          u32x2 recv;
          recv[0] = __shfl_xor(send[0], 16, 64);
          recv[1] = __shfl_xor(send[1], 16, 64);
          const u32x4 o = odd ? u32x4{recv[0], recv[1], b[0], b[1]} : u32x4{a[0], a[1], recv[0], recv[1]};
          TO* dst = C + (long)(2 * jp + (odd ? 1 : 0)) * 16 * p.ldc + i * 16 - (odd ? 4 : 0);
          if (wt) store16_wt(dst, o);
          else *(u32x4*)dst = o;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    if constexpr (RJ != FM) load_residual(j);
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      float v[4];
      finish(i, j, v);
      store4_out<TO>(C + (long)j * 16 * p.ldc + i * 16, v, wt);
    }
  }
}

template <typename TO, int FM, int FN, bool AB = false, bool LEAN = false, int ACT = -1>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, const f32x4 (&acc)[FN][FM], int z, int m_base,
                                              int n_base, int fr, int fq) {
  if (p.vec_ok && m_base + FM * 16 <= p.M && n_base + FN * 16 <= p.N) {
    gemm_epilogue_interior_a<TO, FM, FN, AB, LEAN, ACT>(p, acc, z, m_base, n_base, fr, fq);
    return;
  }
  TO* __restrict__ C = (TO*)p.C + (z / p.batch_inner) * p.strideC + (z % p.batch_inner) * p.strideC2;
  const TO* __restrict__ R = p.R ? (const TO*)p.R + z * p.strideR : nullptr;
  const float* __restrict__ bias = p.bias ? p.bias + z * p.strideBias : nullptr;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n_base + i * 16 + fq * 4;
    if (n >= p.N) continue;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? bias[n + e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m_base + j * 16 + fr;
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bv[e];
      TO* cp = C + (long)m * p.ldc + n;
      if (AB && (p.flags & 8)) {
        const TO* zp = (const TO*)p.Z + (cp - (TO*)p.C);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) v[e] *= act_grad_fast((float)zp[e], ACT >= 0 ? ACT : p.act);
      } else {
        if (!LEAN && p.Z) {
          TO* zp = (TO*)p.Z + (cp - (TO*)p.C);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) zp[e] = (TO)v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_out_c<TO, ACT>(v[e], p.act);
      }
      if (!LEAN && p.p_drop > 0.f) {   // launcher guarantees N % 4 == 0 and ldc == N here
        const Philox4 rb = dropout_bits(p.rng, p.site, (unsigned long)(((long)m * p.N + n) >> 2));
        const unsigned thr = dropout_threshold(p.p_drop);
        const float c = 1.0f / (1.0f - p.p_drop);
        v[0] = rb.x >= thr ? v[0] * c : 0.f;
        v[1] = rb.y >= thr ? v[1] * c : 0.f;
        v[2] = rb.z >= thr ? v[2] * c : 0.f;
        v[3] = rb.w >= thr ? v[3] * c : 0.f;
      }
      if (p.vec_ok && n + 3 < p.N) {
        if (R) {
          const TO* rp = R + (long)m * p.ldr + n;
          if constexpr (sizeof(TO) == 4) {
            const f32x4 r = *(const f32x4*)rp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          } else {
            const typename Vec4T<TO>::type r = *(const typename Vec4T<TO>::type*)rp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
          }
        }
        store4_out<TO>(cp, v, p.flags & 1);
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

// LayerNorm folded into the GEMM (see GemmArgs).  Statistics travel as per-row partial (sum, sum of squares) over
// column slabs = what ONE wave of a tile holds of a row (64 columns in the 128 x 128 kernel), so writing them needs no LDS
// and no barrier; the layout is slab-major, (slabs, rows, 2): the 16 rows of a fragment are one 128-byte line of a slab
// (row-major partials cost 16 lines per load instruction -- +13 % on HuBERT-large's FFN1, measured).  Reading them is latency, not bandwidth: the producer ran on other XCDs, so the first touch of a row's partials
// misses this XCD's L2.  Measured on the encoder's FFN1 (49 us): a loop over the slabs in the epilogue +5.5 us (serialised
// round trips), all slabs loaded and reduced before the K loop +8.5 us (every workgroup stalls on the miss before its first
// tile).  So: ln_issue only ISSUES the loads before the K loop -- lane (fr, fq) takes slabs fq, fq + 4, ... of the FM rows
// it owns, 16 registers -- and ln_finish reduces them (two cross-lane adds) in the epilogue, a whole K loop later.
template <int FM>
__device__ __forceinline__ void ln_issue(const float* stats, int nt, int M, int m_base, int fr, int fq, f32x2 (&raw)[FM][4]) {
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const f32x2* sp = (const f32x2*)stats + min(m_base + j * 16 + fr, M - 1);     // slab-major: 16 rows = one 128-byte line
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[j][q] = fq + 4 * q < nt ? sp[(long)(fq + 4 * q) * M] : f32x2{0.f, 0.f};
  }
}

template <int FM>
__device__ __forceinline__ void ln_finish(const float* stats, int nt, int M, int m_base, int fr, int fq, const f32x2 (&raw)[FM][4],
                                          float inv_cols, float eps, float (&mu)[FM], float (&rs)[FM]) {
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    float S = (raw[j][0][0] + raw[j][1][0]) + (raw[j][2][0] + raw[j][3][0]);
    float Q = (raw[j][0][1] + raw[j][1][1]) + (raw[j][2][1] + raw[j][3][1]);
    if (nt > 16) {     // more than 16 slabs per row: the rest, serially
      const f32x2* sp = (const f32x2*)stats + min(m_base + j * 16 + fr, M - 1);
      for (int t = 16 + fq; t < nt; t += 4) { const f32x2 w = sp[(long)t * M]; S += w[0]; Q += w[1]; }
    }
    S += __shfl_xor(S, 16, 64); S += __shfl_xor(S, 32, 64);
    Q += __shfl_xor(Q, 16, 64); Q += __shfl_xor(Q, 32, 64);
    mu[j] = S * inv_cols;
    rs[j] = rsqrtf(fmaxf(Q * inv_cols - mu[j] * mu[j], 0.f) + eps);
  }
}

// Epilogue of a wave whose FN fragments span exactly one statistics slab (64 columns in the 128 x 128 kernel, 32 in the
// 64 x 64 one).  MODE 1: the operand was LayerNorm'ed
// (folded weights; no residual).  MODE 2: residual add, the residual LayerNorm'ed on the fly when r_stats is given, and
// the statistics of the stored rows written when stats_out is given.  Every load of the epilogue (bias, column sums or
// gamma / beta, the residual fragments) is issued up front, unconditionally: with run-time "is this pointer set" tests
// inside the fragment loops the compiler serialises them into one L2 round trip per fragment column (+4 us per launch).
template <typename TO, int FM, int FN, int MODE, int ACT>
__device__ __forceinline__ void gemm_epilogue_ln_a(const GemmArgs& p, const f32x4 (&acc)[FN][FM], int m_base, int n_base,
                                                   int fr, int fq, const f32x2 (&raw)[FM][4]) {
  static_assert((FN == 4 || FN == 2) && (FM >= 2 && FM <= 4), "one wave = one statistics slab of 16 FN columns, 2 to 4 fragment rows");
  constexpr int SLAB = FN * 16;
  // fragment columns whose operands are requested together: all of them for the two-row tiles; two at a time for the
  // 192-row tile, whose kernel has 128 registers for everything (all four would need 166: one workgroup per CU)
  constexpr int IB = FM == 3 ? 2 : FN;
  typedef typename Vec4T<TO>::type V4;
  const int n = n_base + fq * 4;
  float mu[FM], rs[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) { mu[j] = 0.f; rs[j] = 1.f; }      // (r - 0) * 1 * 1 + 0 == r exactly: the plain residual
  if constexpr (MODE == 1) ln_finish<FM>(p.a_stats, p.a_nt, p.M, m_base, fr, fq, raw, 1.0f / (float)p.K, p.ln_eps, mu, rs);
  else if (p.r_stats) ln_finish<FM>(p.r_stats, p.r_nt, p.M, m_base, fr, fq, raw, 1.0f / (float)p.N, p.ln_eps, mu, rs);
  // Statistics of the stored rows, in ONE association for every kernel that writes them (this epilogue on its 128 x 128 /
  // 192 x 128 / 64 x 64 tiles, gemm8_kernel on 256 x 256): a row's result must not depend on the tile the launch's row count
  // routed it to (a clip alone and the same clip in a batch of 32).  Per 16-column fragment i and lane group fq:
  // p_i = the serial sum of its four stored values; (p_0 + p_1) + (p_2 + p_3) over the slab's fragments; then the lane
  // groups fq ^ 1 and fq ^ 2.  (tS / prS: the odd-man-out and the first pair while the fragments go by.)
  float rowS[FM], rowQ[FM], tS[FM], tQ[FM], prS[FM], prQ[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) rowS[j] = rowQ[j] = tS[j] = tQ[j] = prS[j] = prQ[j] = 0.f;
  const bool odd = fq & 1;
  const bool pair = sizeof(TO) == 2 && (p.flags & 2);
#pragma unroll
  for (int i0 = 0; i0 < FN; i0 += IB) {
    f32x4 bv[IB], xv[IB], yv[IB];
    V4 rr[FM][IB];
#pragma unroll
    for (int ii = 0; ii < IB; ++ii) bv[ii] = *(const f32x4*)(p.bias + n + (i0 + ii) * 16);
    if constexpr (MODE == 1) {
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) xv[ii] = *(const f32x4*)(p.w_colsum + n + (i0 + ii) * 16);
    } else {
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const TO* rp = (const TO*)p.R + (long)min(m_base + j * 16 + fr, p.M - 1) * p.ldr + n;
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) rr[j][ii] = *(const V4*)(rp + (i0 + ii) * 16);
      }
      if (p.r_stats) {
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) { xv[ii] = *(const f32x4*)(p.r_gamma + n + (i0 + ii) * 16); yv[ii] = *(const f32x4*)(p.r_beta + n + (i0 + ii) * 16); }
      } else {
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) { xv[ii] = f32x4{1.f, 1.f, 1.f, 1.f}; yv[ii] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    }
#pragma unroll
    for (int ii = 0; ii < IB; ++ii) {
      const int i = i0 + ii;
      V4 o[FM];
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][j][e];
          if constexpr (MODE == 1) x = rs[j] * (x - mu[j] * xv[ii][e]);
          v[e] = act_out_c<TO, ACT>(x + bv[ii][e], p.act);
          if constexpr (MODE == 2) v[e] += fmaf(((float)rr[j][ii][e] - mu[j]) * rs[j], xv[ii][e], yv[ii][e]);
        }
        if constexpr (sizeof(TO) == 4) o[j] = V4{v[0], v[1], v[2], v[3]};
        else o[j] = pack4<TO>(v[0], v[1], v[2], v[3]);
        if constexpr (MODE == 2) {
          float pS = 0.f, pQ = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float w = (float)o[j][e]; pS += w; pQ = fmaf(w, w, pQ); }   // of what the consumer will read
          if ((i & 1) == 0) { tS[j] = pS; tQ[j] = pQ; }
          else {
            const float aS = tS[j] + pS, aQ = tQ[j] + pQ;
            if (FN == 2) { rowS[j] = aS; rowQ[j] = aQ; }
            else if ((i & 2) == 0) { prS[j] = aS; prQ[j] = aQ; }
            else { rowS[j] = prS[j] + aS; rowQ[j] = prQ[j] + aQ; }
          }
        }
      }
      bool stored = false;
      if constexpr (sizeof(TO) == 2) {
        if (pair) {     // lanes l and l ^ 16 swap one fragment row: one 16-byte store each (see gemm_epilogue_interior)
#pragma unroll
          for (int jp = 0; jp + 1 < FM; jp += 2) {
            const u32x2 a = __builtin_bit_cast(u32x2, o[jp]), b = __builtin_bit_cast(u32x2, o[jp + 1]);
            const u32x2 send = odd ? a : b;
            u32x2 recv;
            recv[0] = __shfl_xor(send[0], 16, 64);
            recv[1] = __shfl_xor(send[1], 16, 64);
            const u32x4 w = odd ? u32x4{recv[0], recv[1], b[0], b[1]} : u32x4{a[0], a[1], recv[0], recv[1]};
            const int m = m_base + jp * 16 + (odd ? 16 : 0) + fr;
            if (m < p.M) *(u32x4*)((TO*)p.C + (long)m * p.ldc + n + i * 16 - (odd ? 4 : 0)) = w;
          }
          if constexpr (FM % 2 == 1) {   // the last fragment row of the 192-row tile has no partner: plain 8-byte stores
            const int m2 = m_base + (FM - 1) * 16 + fr;
            if (m2 < p.M) *(V4*)((TO*)p.C + (long)m2 * p.ldc + n + i * 16) = o[FM - 1];
          }
          stored = true;
        }
      }
      if (!stored) {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int m = m_base + j * 16 + fr;
          if (m < p.M) *(V4*)((TO*)p.C + (long)m * p.ldc + n + i * 16) = o[j];
        }
      }
    }
  }
  if constexpr (MODE == 2) {
    if (p.stats_out) {
      // row sums over this wave's slab: the 4 lane groups (fq) hold 4 columns of every fragment each
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        rowS[j] += __shfl_xor(rowS[j], 16, 64); rowS[j] += __shfl_xor(rowS[j], 32, 64);
        rowQ[j] += __shfl_xor(rowQ[j], 16, 64); rowQ[j] += __shfl_xor(rowQ[j], 32, 64);
        const int m = m_base + j * 16 + fr;
        if (fq == 0 && m < p.M) *(f32x2*)(p.stats_out + ((long)(n_base / SLAB) * p.M + m) * 2) = f32x2{rowS[j], rowQ[j]};
      }
    }
  }
}

template <typename TO, int FM, int FN, int MODE, int ACT = -1>
__device__ __forceinline__ void gemm_epilogue_ln(const GemmArgs& p, const f32x4 (&acc)[FN][FM], int m_base, int n_base,
                                                 int fr, int fq, const f32x2 (&raw)[FM][4]) {
  gemm_epilogue_ln_a<TO, FM, FN, MODE, ACT>(p, acc, m_base, n_base, fr, fq, raw);
}

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
  const int zo = z / p.batch_inner, zi = z % p.batch_inner;
  const T* __restrict__ A = (const T*)p.A + zo * p.strideA + zi * p.strideA2;
  const T* __restrict__ W = (const T*)p.W + zo * p.strideW + zi * p.strideW2;
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
    a_ptr[i] = A + a_row_offset(p, mm) + cc * E;
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

  gemm_epilogue<TO, FM, FN>(p, acc, z, m0 + wm, n0 + wn, fr, fq);
}

// ---------------------------------------------------------------------------------------------------
// bf16 kernel v2: LDS-DMA staging (global_load_lds, 16 B/lane) into an NSTAGE-deep LDS ring, counted
// s_waitcnt vmcnt so NSTAGE-2 K tiles stay in flight ACROSS the (single, raw) barrier of each K step.
// Same LDS image / swizzle / fragment reads / epilogue as v1; the swizzle moves to the per-lane SOURCE
// address because an LDS-DMA wave-instruction writes 1 KiB linearly (cdna_hip_programming.md rule 21).
// Requires K % 64 == 0 (no K tail: LDS-DMA cannot zero-fill); out-of-range rows are clamped to a valid
// row (their products are never stored).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// EPI: which epilogues this instantiation carries.  0 = all of them behind run-time tests (training: pre-activation copy,
// dropout, activation backward, both LayerNorm forms); 1 = the plain inference epilogue only (bias, activation, residual);
// 2 / 3 = the LayerNorm-operand / LayerNorm-residual form only.  One instantiation with everything is 127 KB of code of which a
// launch executes a few KB scattered among the branches it does not take: per-workgroup stamps (tools/gemm_stamps.py) put the
// epilogue of a 128 x 128 tile at 3.0 us for 32 outputs per lane (bias only), most of it instruction fetch.
// EPIA = EPI + 10 * (1 + activation) when the activation is compiled in as well (11 / 21 plain with none / GELU, 12 / 22
// LayerNorm-operand with none / GELU, 13 LayerNorm-residual with none): see act_out_c.
template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, bool PIPE = false, typename TI = bf16_t, bool STAG = false, int EPIA = 0>
// 8-wave workgroups whose ring fits twice into a CU's LDS are MEANT to run two per CU: 4 waves per SIMD = 128 registers
// (the 192-row tile's LayerNorm epilogues drifted to 145 once, i.e. to one workgroup per CU: HuBERT-large 18.7 -> 21.4 ms).
// Its everything-epilogue instantiation (EPI 0: dropout / pre-activation copies on a tall grid, no caller on the path) does not
// fit 128 without spilling and keeps the register count the compiler picks.
__global__ __launch_bounds__(WM * WN * 64, (STAG || (WM * WN == 8 && NSTAGE * (BM + BN) * 128 <= 80 * 1024 && (EPIA % 10 != 0 || BM * BN <= 128 * 128))) ? 4 : 1) void gemm2_kernel(const GemmArgs p) {
  constexpr int EPI = EPIA % 10, ACTK = EPIA / 10 - 1;
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int LPT = (BM + BN) * 8 / NT;  // LDS-DMA instructions per thread per K tile
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  static_assert((BM + BN) * 8 % NT == 0, "tile chunks must divide over the threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int pid = blockIdx.x;
  const int xcd = pid & 7, slot = pid >> 3;
  // XCD grid (8 / xn) x xn over (M, N) tiles: an XCD's L2 then holds 1 / xn of the weight matrix instead of all of it
  const int xm_n = 8 / p.xn, ntx = (p.nt + p.xn - 1) / p.xn;
  const int m_tile = (slot / ntx) * xm_n + (xcd % xm_n), n_tile = (slot % ntx) * p.xn + xcd / xm_n;
  if (m_tile >= p.mt || n_tile >= p.nt) return;
  const int z = blockIdx.z;
  const int zo = z / p.batch_inner, zi = z % p.batch_inner;
  const bf16_t* __restrict__ A = (const bf16_t*)p.A + zo * p.strideA + zi * p.strideA2;
  const bf16_t* __restrict__ W = (const bf16_t*)p.W + zo * p.strideW + zi * p.strideW2;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#ifdef MSMD_EXPERIMENTAL
  long stamp[5] = {0, 0, 0, 0, 0};
  if (p.stamps) stamp[0] = (long)__builtin_amdgcn_s_memrealtime();
#endif

  const bf16_t* src[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int id = (i * NW + wid) * 64 + lane;  // 16-B slot of the tile image
    const int row = id >> 3, phys = id & 7;
    const int c = phys ^ ((row >> 1) & 7);      // logical chunk stored at this physical slot
    if (row < BM) {
      const int m = min(m0 + row, p.M - 1);
      src[i] = A + a_row_offset(p, m) + c * 8;
    } else {
      const int n = min(n0 + row - BM, p.N - 1);
      src[i] = W + (long)n * p.ldw + c * 8;
    }
  }
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + kt * 64),
                                       (lds_void_t*)(smem + stage * STAGE + (i * NW + wid) * 1024), 16, 0, 0);
  };

  const int wm = (wid / WN) * (BM / WM), wn = (wid % WN) * (BN / WN);
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / 64;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);
  if (p.stagger_ticks > 0 && ((pid >> 8) & 1) && pid < 512) {
    // Two co-resident workgroups that start together stay in lockstep for the whole launch (both multiply, then both run
    // their epilogues, then their two successors start together): the second one of each CU waits half a tile period once,
    // with its first stage already in flight, so that from then on one workgroup's prologue / epilogue runs beside the
    // other's K loop
    const unsigned long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < p.stagger_ticks) __builtin_amdgcn_s_sleep(16);
  }
  // LayerNorm-folding mode (msmd_gemm_ln; the 4 x 2-wave 128 x 128 tile only): the row statistics' loads go out now, consumed in the epilogue
  constexpr bool LNK = (BN == 128 || BN == 64) && WN == 2 && (FM == 2 || FM == 3) && !STAG && sizeof(TO) == 2;   // slab = BN / 2
  // the 192-row tile has no registers to spare during the K loop (124 of 128): its statistics are loaded in the epilogue instead
  // (one batched round trip; with two workgroups per CU and grids of many rounds it hides under the neighbour's K loop)
  constexpr bool LN_LATE = FM == 3;
  f32x2 lnraw[(LNK && !LN_LATE) ? FM : 1][4];
  if constexpr (LNK && !LN_LATE && EPI != 1) {
    if (EPI == 2 || (EPI == 0 && p.a_stats)) ln_issue<FM>(p.a_stats, p.a_nt, p.M, m0 + wm, fr, fq, lnraw);
    else if (p.r_stats) ln_issue<FM>(p.r_stats, p.r_nt, p.M, m0 + wm, fr, fq, lnraw);
  }
  int stage = 0;
  if constexpr (STAG) {
    // Staggered halves (MI355X_MICROARCH.md, "Two waves per SIMD", item 9): the eight waves run the same program with one
    // barrier per K tile, so SIMD partners reach their fragment reads, their MFMAs and the barrier together.  Waves 4-7
    // (the younger wave of every SIMD) multiply tile k AFTER the barrier of tile k + 1, from fragments they read one
    // iteration earlier: while waves 0-3 read the new tile, waves 4-7 keep the matrix pipe busy, and vice versa.  Same
    // products in the same order per output element: results are bit-identical to the unstaggered kernel.
    // MEASURED (round 3): 25-30 % SLOWER on every shape and 4.76 -> 5.64 ms on the forward step -- with two workgroups per CU
    // the co-resident workgroup already fills the other's read phase, and the late half lengthens every tile's critical
    // path.  Experimental build only (variant 41).
    static_assert(NSTAGE == 2 && NW == 8, "stagger is written for the 2-stage, 8-wave kernel");
    const bool late = __builtin_amdgcn_readfirstlane(wid) >= 4;
    u32x4 fx[2][FM], fw[2][FN];
    auto read_frags = [&](int st) {
      const unsigned char* sa = smem + st * STAGE;
      const unsigned char* sw = sa + BM * 128;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int j = 0; j < FM; ++j) fx[g][j] = *(const u32x4*)(sa + lds_off(wm + j * 16 + fr, g * 4 + fq));
#pragma unroll
        for (int i = 0; i < FN; ++i) fw[g][i] = *(const u32x4*)(sw + lds_off(wn + i * 16 + fr, g * 4 + fq));
      }
    };
    auto multiply = [&]() {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) Mfma<TI>::run(fw[g][i], fx[g][j], acc[i][j]);
    };
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) issue(kt + 1, stage ^ 1);
      if (late && kt > 0) multiply();           // waves 4-7: tile kt - 1, from the fragments read before this barrier
      __builtin_amdgcn_sched_barrier(0);
      read_frags(stage);
      __builtin_amdgcn_sched_barrier(0);
      if (!late) multiply();                     // waves 0-3: this tile
      // waves 4-7: the reads must have RETURNED before the wave arrives at the next barrier (after it any wave may re-stage
      // this buffer); waves 0-3 have consumed theirs in the MFMAs above
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1;
    }
    if (late) multiply();
    gemm_epilogue<TO, FM, FN, sizeof(TO) == 2>(p, acc, z, m0 + wm, n0 + wn, fr, fq);
    return;
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NSTAGE - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef MSMD_EXPERIMENTAL
    if (p.stamps && kt == 0) stamp[1] = (long)__builtin_amdgcn_s_memrealtime();
#endif
    if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, (stage + NSTAGE - 1) % NSTAGE);
    const unsigned char* sa = smem + stage * STAGE;
    const unsigned char* sw = sa + BM * 128;
    if constexpr (PIPE) {
      // Explicit software pipeline: issue the fragment reads of BOTH 32-deep k-steps, then the MFMAs.  Left to
      // itself hipcc re-uses 16 fragment registers and emits read / lgkmcnt(0) / 4 MFMA groups, exposing the LDS
      // latency four times per K tile; pinned like this the second k-step's reads land behind the first's MFMAs.
      u32x4 fx[2][FM], fw[2][FN];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int j = 0; j < FM; ++j) fx[g][j] = *(const u32x4*)(sa + lds_off(wm + j * 16 + fr, g * 4 + fq));
#pragma unroll
        for (int i = 0; i < FN; ++i) fw[g][i] = *(const u32x4*)(sw + lds_off(wn + i * 16 + fr, g * 4 + fq));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) Mfma<TI>::run(fw[g][i], fx[g][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
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
          for (int j = 0; j < FM; ++j) Mfma<TI>::run(fw[i], fx[j], acc[i][j]);
      }
    }
    stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
  }
#ifdef MSMD_EXPERIMENTAL
  if (p.stamps) stamp[2] = (long)__builtin_amdgcn_s_memrealtime();
#endif
  // the 192-row tile has no register to spare: the lane's fragment coordinates are derived again here instead of living
  // through the K loop (one of them went to scratch otherwise)
  int tid_e = tid;
  if constexpr (LN_LATE) asm volatile("" : "+v"(tid_e));
  const int wm_e = ((tid_e >> 6) / WN) * (BM / WM), wn_e = ((tid_e >> 6) % WN) * (BN / WN);
  const int fr_e = tid_e & 15, fq_e = (tid_e & 63) >> 4;
  auto run_epilogue = [&]() {
    if constexpr (EPI == 1) { gemm_epilogue<TO, FM, FN, false, true, ACTK>(p, acc, z, m0 + wm_e, n0 + wn_e, fr_e, fq_e); return; }
    if constexpr (LNK && !LN_LATE && EPI == 2) { gemm_epilogue_ln<TO, FM, FN, 1, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, lnraw); return; }
    if constexpr (LNK && !LN_LATE && EPI == 3) { gemm_epilogue_ln<TO, FM, FN, 2, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, lnraw); return; }
    if constexpr (LNK && LN_LATE && (EPI == 2 || EPI == 3)) {      // the 192-row tile: statistics loaded here (no registers to spare in the K loop)
      f32x2 late[FM][4];
      if constexpr (EPI == 2) ln_issue<FM>(p.a_stats, p.a_nt, p.M, m0 + wm_e, fr_e, fq_e, late);
      else if (p.r_stats) ln_issue<FM>(p.r_stats, p.r_nt, p.M, m0 + wm_e, fr_e, fq_e, late);
      gemm_epilogue_ln<TO, FM, FN, EPI - 1, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, late);
      return;
    }
    if constexpr (LNK && !LN_LATE) {
      if (p.a_stats) { gemm_epilogue_ln<TO, FM, FN, 1, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, lnraw); return; }
      if (p.r_stats || p.stats_out) { gemm_epilogue_ln<TO, FM, FN, 2, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, lnraw); return; }
    }
    if constexpr (LNK && LN_LATE) {
      if (p.a_stats || p.r_stats || p.stats_out) {
        f32x2 late[FM][4];
        if (p.a_stats) ln_issue<FM>(p.a_stats, p.a_nt, p.M, m0 + wm_e, fr_e, fq_e, late);
        else if (p.r_stats) ln_issue<FM>(p.r_stats, p.r_nt, p.M, m0 + wm_e, fr_e, fq_e, late);
        if (p.a_stats) gemm_epilogue_ln<TO, FM, FN, 1, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, late);
        else gemm_epilogue_ln<TO, FM, FN, 2, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, late);
        return;
      }
    }
    gemm_epilogue<TO, FM, FN, sizeof(TO) == 2, false, ACTK>(p, acc, z, m0 + wm_e, n0 + wn_e, fr_e, fq_e);
  };
  run_epilogue();
#ifdef MSMD_EXPERIMENTAL
  if (p.stamps) {      // per-workgroup stamps (tools/gemm_stamps.py): entry, first tile landed, K loop done, stores issued, stores drained
    stamp[3] = (long)__builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[4] = (long)__builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      long* o = p.stamps + (long)pid * 8;
      unsigned hw = 0, xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[0] = stamp[0]; o[1] = stamp[1]; o[2] = stamp[2]; o[3] = stamp[3]; o[4] = stamp[4]; o[5] = hw; o[6] = xcc; o[7] = m_tile * 1000 + n_tile;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Persistent form of gemm2_kernel<.., 2 stages, pipelined> for launches of MORE THAN ONE ROUND (more tiles than the 512
// workgroups the chip holds at two per CU).  The grid is 512 workgroups; workgroup b multiplies the tiles gemm2_kernel's blocks
// b, b + 512, b + 1024, ... would (same XCD: 512 % 8 == 0) -- and the operand stream never drains between them: the LAST K
// tile of a tile issues the FIRST stage of the next one, so the next tile's prologue (addresses, the first operand round
// trip: 1.2-1.9 us per workgroup on the stamps of DESIGN.md 5d) runs under this tile's last multiply and its epilogue, and
// there is no workgroup retirement / dispatch between rounds (0.8 us on the same stamps).  Per tile the products, their
// order and the epilogue are gemm2_kernel's: results are bit-identical.
template <typename TO, int BM, int BN, int WM, int WN, typename TI, int EPIA>
__global__ __launch_bounds__(WM * WN * 64, 4) void gemm2p_kernel(const GemmArgs p) {
  constexpr int EPI = EPIA % 10, ACTK = EPIA / 10 - 1;
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int LPT = (BM + BN) * 8 / NT;
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  static_assert(EPI >= 1 && EPI <= 3 && sizeof(TO) == 2 && NW == 8 && WN == 2 && BN == 128 && (FM == 2 || FM == 3), "the hot 8-wave tiles, inference epilogues");
  static_assert((BM + BN) * 8 % NT == 0, "tile chunks must divide over the threads");
  constexpr bool LN_LATE = FM == 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int xm_n = 8 / p.xn, ntx = (p.nt + p.xn - 1) / p.xn;
  const int vgrid = ((p.mt + xm_n - 1) / xm_n) * ntx * 8;      // gemm2_kernel's grid
  const TI* __restrict__ A = (const TI*)p.A;
  const TI* __restrict__ W = (const TI*)p.W;
  // first block id >= vb of this workgroup's sequence that maps to a tile (the XCD grid pads), or -1
  auto next_tile = [&](int vb, int& m0, int& n0) -> int {
    for (; vb < vgrid; vb += gridDim.x) {
      const int xcd = vb & 7, slot = vb >> 3;
      const int m_tile = (slot / ntx) * xm_n + (xcd % xm_n), n_tile = (slot % ntx) * p.xn + xcd / xm_n;
      if (m_tile < p.mt && n_tile < p.nt) { m0 = m_tile * BM; n0 = n_tile * BN; return vb; }
    }
    return -1;
  };
  // Everything derived from the lane id is derived AGAIN per tile and per phase (t_k for the K loop, t_e for the epilogue) from
  // copies the compiler cannot match: kept across the tile loop, the K loop's fragment / staging offsets would be live in
  // the epilogue and the epilogue's in the K loop (+30 registers: 90-250 bytes of scratch in every LayerNorm form)
  const TI* src[LPT];
  auto set_src = [&](int m0, int n0, int t) {
    const int lane = t & 63, wid = t >> 6;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int id = (i * NW + wid) * 64 + lane;  // 16-B slot of the tile image
      const int row = id >> 3, phys = id & 7;
      const int c = phys ^ ((row >> 1) & 7);      // logical chunk stored at this physical slot
      if (row < BM) src[i] = A + a_row_offset(p, min(m0 + row, p.M - 1)) + c * 8;
      else src[i] = W + (long)min(n0 + row - BM, p.N - 1) * p.ldw + c * 8;
    }
  };
  auto issue = [&](int kt, int stage, int t) {
    const int wid = t >> 6;
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + kt * 64),
                                       (lds_void_t*)(smem + stage * STAGE + (i * NW + wid) * 1024), 16, 0, 0);
  };
  int m0 = 0, n0 = 0;
  int vb = next_tile(blockIdx.x, m0, n0);
  if (vb < 0) return;
  const int nk = p.K / 64;
  set_src(m0, n0, tid);
  int stage = 0;
  issue(0, 0, tid);
  for (;;) {
    int t_k = tid;
    asm volatile("" : "+v"(t_k));
    const int wm = ((t_k >> 6) / WN) * (BM / WM), wn = ((t_k >> 6) % WN) * (BN / WN);
    const int fr = t_k & 15, fq = (t_k & 63) >> 4;
    f32x2 lnraw[LN_LATE ? 1 : FM][4];
    if constexpr (!LN_LATE && EPI != 1) {
      if constexpr (EPI == 2) ln_issue<FM>(p.a_stats, p.a_nt, p.M, m0 + wm, fr, fq, lnraw);
      else if (p.r_stats) ln_issue<FM>(p.r_stats, p.r_nt, p.M, m0 + wm, fr, fq, lnraw);
    }
    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int m0n = 0, n0n = 0, vbn = -1;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) {
        issue(kt + 1, stage ^ 1, t_k);
      } else {
        // the other stage is free (every wave is past this barrier, i.e. done reading it): the next tile's first stage
        vbn = next_tile(vb + (int)gridDim.x, m0n, n0n);
        if (vbn >= 0) { set_src(m0n, n0n, t_k); issue(0, stage ^ 1, t_k); }
      }
      const unsigned char* sa = smem + stage * STAGE;
      const unsigned char* sw = sa + BM * 128;
      u32x4 fx[2][FM], fw[2][FN];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int j = 0; j < FM; ++j) fx[g][j] = *(const u32x4*)(sa + lds_off(wm + j * 16 + fr, g * 4 + fq));
#pragma unroll
        for (int i = 0; i < FN; ++i) fw[g][i] = *(const u32x4*)(sw + lds_off(wn + i * 16 + fr, g * 4 + fq));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) Mfma<TI>::run(fw[g][i], fx[g][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      stage ^= 1;
    }
    int t_e = tid;
    asm volatile("" : "+v"(t_e));
    const int wm_e = ((t_e >> 6) / WN) * (BM / WM), wn_e = ((t_e >> 6) % WN) * (BN / WN);
    const int fr_e = t_e & 15, fq_e = (t_e & 63) >> 4;
    if constexpr (EPI == 1) {
      gemm_epilogue<TO, FM, FN, false, true, ACTK>(p, acc, 0, m0 + wm_e, n0 + wn_e, fr_e, fq_e);
    } else if constexpr (!LN_LATE) {
      gemm_epilogue_ln<TO, FM, FN, EPI - 1, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, lnraw);
    } else {      // the 192-row tile: statistics loaded here (no registers to spare in the K loop)
      f32x2 late[FM][4];
      if constexpr (EPI == 2) ln_issue<FM>(p.a_stats, p.a_nt, p.M, m0 + wm_e, fr_e, fq_e, late);
      else if (p.r_stats) ln_issue<FM>(p.r_stats, p.r_nt, p.M, m0 + wm_e, fr_e, fq_e, late);
      gemm_epilogue_ln<TO, FM, FN, EPI - 1, ACTK>(p, acc, m0 + wm_e, n0 + wn_e, fr_e, fq_e, late);
    }
    if (vbn < 0) break;
    vb = vbn; m0 = m0n; n0 = n0n;
    // the operand addresses again, from values the compiler cannot match with the ones above: they are NOT kept in
    // registers across the epilogue (8 registers the LayerNorm epilogues do not have)
    asm volatile("" : "+s"(m0), "+s"(n0));
    int t_s = tid;
    asm volatile("" : "+v"(t_s));
    set_src(m0, n0, t_s);
  }
}

// ---------------------------------------------------------------------------------------------------
// MSMD_F16X2 ("split pair", common.h) kernel: the parity-grade speed mode.  Operands are rows of 2K fp16 numbers in
// 32-element blocks [hi x 32 | lo x 32], so ONE 128-byte line = one 32-deep k-step of both planes and the LDS-DMA
// ring, LDS image, swizzle and fragment reads are those of gemm2_kernel (chunks 0-3 of a row = hi, 4-7 = lo).  Per
// k-step and fragment pair: acc0 += Wh.Ah, acc1 += Wh.Al + Wl.Ah (three v_mfma_f32_16x16x32_f16), result
// acc0 + acc1 / 2048.  The launcher passes lda / ldw / strides of A and W already doubled (fp16 units); K stays logical.
// TO = float (fp32 C and residual) or f16_t (C and residual in split storage).
// ACT: see act_out_c (>= 0: a constant of the kernel).  GELU through the 12-instruction erf (|abs err| <= 1.5e-7, i.e.
// <= 0.5 |x| 1.5e-7 on the output: the size of an fp32 rounding error at these magnitudes); libm's erff would be ~15 % of an
// FFN1 launch.
template <int ACT> __device__ __forceinline__ float act_split(float x, int act) {
  if constexpr (ACT == MSMD_ACT_GELU) return gelu_fast(x);
  else if constexpr (ACT == MSMD_ACT_NONE) return x;
  else return act == MSMD_ACT_GELU ? gelu_fast(x) : apply_act(x, act);
}

template <int FM, int FN, int ACT = -1>
__device__ __forceinline__ void gemm_epilogue_split(const GemmArgs& p, const f32x4 (&acc)[FN][FM], int z, int m_base,
                                                    int n_base, int fr, int fq) {
  f16_t* __restrict__ C = (f16_t*)p.C + 2 * ((z / p.batch_inner) * p.strideC + (z % p.batch_inner) * p.strideC2);
  const f16_t* __restrict__ R = p.R ? (const f16_t*)p.R + 2 * (z * p.strideR) : nullptr;
  const float* __restrict__ bias = p.bias ? p.bias + z * p.strideBias : nullptr;
  if (m_base + FM * 16 <= p.M && n_base + FN * 16 <= p.N) {
    // interior tiles: straight-line code, every load (bias, both halves of every residual fragment) requested before the
    // first store -- between the stores they are serialised into one L2 round trip per fragment (see gemm_epilogue_interior)
    const int n = n_base + fq * 4;
    f32x4 bv[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bv[i] = bias ? *(const f32x4*)(bias + n + i * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    f16x4 rh[FM][FN], rl[FM][FN];
    if (R) {
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const f16_t* rrow = R + (long)(m_base + j * 16 + fr) * 2 * p.ldr;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const f16_t* q = rrow + split_col(n + i * 16);
          rh[j][i] = *(const f16x4*)q;
          rl[j][i] = *(const f16x4*)(q + 32);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      f16_t* crow = C + (long)(m_base + j * 16 + fr) * 2 * p.ldc;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_split<ACT>(acc[i][j][e] + bv[i][e], p.act);
        if (R) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += unsplit_f16x2(rh[j][i][e], rl[j][i][e]);
        }
        store4_split(crow, n + i * 16, v);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n_base + i * 16 + fq * 4;
    if (n >= p.N) continue;   // N % 4 == 0 (launcher)
    const f32x4 bv = bias ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m_base + j * 16 + fr;
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_split<ACT>(acc[i][j][e] + bv[e], p.act);
      if (R) {
        float r[4];
        load4_split(R + (long)m * 2 * p.ldr, n, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      store4_split(C + (long)m * 2 * p.ldc, n, v);
    }
  }
}

// WS (the call carries MSMD_GEMM_W_BELOW_32, i.e. W is a model weight): gemm8_kernel's fold-free form -- W's hi fragments x 2^11
// in registers, ONE accumulator in units of 2^-11, per K tile and output element the products Wh.AL, WL.Ah, (2^11 Wh).Ah in
// this order -- so that a launch returns the same bits whether its row count routes it to this kernel or to the 256 x 256 one.
template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, int ACTK = -1, bool WS = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm2s_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int LPT = (BM + BN) * 8 / NT;
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  static_assert((BM + BN) * 8 % NT == 0, "tile chunks must divide over the threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int pid = blockIdx.x;
  const int xcd = pid & 7, slot = pid >> 3;
  const int xm_n = 8 / p.xn, ntx = (p.nt + p.xn - 1) / p.xn;
  const int m_tile = (slot / ntx) * xm_n + (xcd % xm_n), n_tile = (slot % ntx) * p.xn + xcd / xm_n;
  if (m_tile >= p.mt || n_tile >= p.nt) return;
  const int z = blockIdx.z;
  const int zo = z / p.batch_inner, zi = z % p.batch_inner;
  const f16_t* __restrict__ A = (const f16_t*)p.A + zo * p.strideA + zi * p.strideA2;
  const f16_t* __restrict__ W = (const f16_t*)p.W + zo * p.strideW + zi * p.strideW2;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  const f16_t* src[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int id = (i * NW + wid) * 64 + lane;
    const int row = id >> 3, phys = id & 7;
    const int c = phys ^ ((row >> 1) & 7);
    if (row < BM) src[i] = A + a_row_offset(p, min(m0 + row, p.M - 1)) + c * 8;
    else src[i] = W + (long)min(n0 + row - BM, p.N - 1) * p.ldw + c * 8;
  }
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + kt * 64),
                                       (lds_void_t*)(smem + stage * STAGE + (i * NW + wid) * 1024), 16, 0, 0);
  };

  const int wm = (wid / WN) * (BM / WM), wn = (wid % WN) * (BN / WN);
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc0[FN][FM], acc1[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc0[i][j] = acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / 32;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NSTAGE - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, (stage + NSTAGE - 1) % NSTAGE);
    const unsigned char* sa = smem + stage * STAGE;
    const unsigned char* sw = sa + BM * 128;
    u32x4 ah[FM], al[FM], wh[FN], wl[FN];
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      ah[j] = *(const u32x4*)(sa + lds_off(wm + j * 16 + fr, fq));
      al[j] = *(const u32x4*)(sa + lds_off(wm + j * 16 + fr, 4 + fq));
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      wh[i] = *(const u32x4*)(sw + lds_off(wn + i * 16 + fr, fq));
      wl[i] = *(const u32x4*)(sw + lds_off(wn + i * 16 + fr, 4 + fq));
    }
    if constexpr (WS) {
      u32x4 ws[FN];
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        f16x8 h = __builtin_bit_cast(f16x8, wh[i]);
        h = h * (f16_t)MSMD_SPLIT_SCALE;
        ws[i] = __builtin_bit_cast(u32x4, h);
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) Mfma<f16_t>::run(wh[i], al[j], acc0[i][j]);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) Mfma<f16_t>::run(wl[i], ah[j], acc0[i][j]);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) Mfma<f16_t>::run(ws[i], ah[j], acc0[i][j]);
    } else {
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          Mfma<f16_t>::run(wh[i], ah[j], acc0[i][j]);
          Mfma<f16_t>::run(wh[i], al[j], acc1[i][j]);
          Mfma<f16_t>::run(wl[i], ah[j], acc1[i][j]);
        }
    }
    stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
  }
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc0[i][j][e] = WS ? acc0[i][j][e] * MSMD_SPLIT_INV : fmaf(acc1[i][j][e], MSMD_SPLIT_INV, acc0[i][j][e]);   // (x 2^-11: exact)
  if constexpr (sizeof(TO) == 4) gemm_epilogue<float, FM, FN, false, false, ACTK>(p, acc0, z, m0 + wm, n0 + wn, fr, fq);
  else gemm_epilogue_split<FM, FN, ACTK>(p, acc0, z, m0 + wm, n0 + wn, fr, fq);
}

// ---------------------------------------------------------------------------------------------------
// bf16 / f16 kernel v4 (round 4): the same LDS image, LDS-DMA ring and epilogues as gemm2_kernel, with the fragment reads
// software-pipelined INSIDE each wave and one workgroup barrier per K tile placed between its two 32-deep k-steps.
//
// Why (tools/lds_share_probe.hip, tools/intake_probe.hip): with LDS-DMA writes, fragment reads and MFMAs
// running free in the 128 x 128 tile's per-K-tile ratio one CU sustains 0.37 us per K tile (1 460 TFLOP/s chip-wide);
// gemm2_kernel takes 0.52 us hot and 0.68 us in the step.  Its waves run  wait / barrier / 4 DMA / 12 reads / lgkmcnt(0) /
// 16 MFMA  back to back: nothing inside a wave overlaps, and a 128 x 128 K tile needs all of the CU's 64 B/clk vector-memory
// path for as long as its MFMAs take (32 KB staged per 512 MFMA cycles), so every bubble is lost.  Here:
//   * a wave holds TWO fragment sets; while the 16-32 MFMAs of one k-step issue, the reads of the next k-step are in flight
//     (also across the K-tile boundary), so the matrix pipe never waits for LDS latency;
//   * the barrier sits between the k-steps: before it  reads(g1) + MFMA(g0),  after it  DMA(tile + NSTAGE) + reads(next g0) +
//     MFMA(g1) -- the stage a wave refills is the one whose last reads the barrier has just retired, and the DMA has
//     NSTAGE - 1 whole K tiles to land;
//   * bigger tiles at ONE workgroup per CU (256 x 128: 48 KB per 1024 MFMA cycles = 3/4 of the bytes per FLOP; 256 VGPRs):
//     8 waves of 64 x 64.  Same products in the same order per output element as gemm2_kernel: bit-identical results.
constexpr int waitcnt_lgkm0() { return 0xC07F; }                                              // lgkmcnt(0), vmcnt / expcnt untouched
constexpr int waitcnt_vm(int n) { return (n & 0xF) | 0x70 | 0xF00 | ((n >> 4) << 14); }        // vmcnt(n), lgkmcnt / expcnt untouched

template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, typename TI = bf16_t, bool ILV = false>
__global__ __launch_bounds__(WM * WN * 64)
__attribute__((amdgpu_waves_per_eu(NSTAGE * (BM + BN) * 128 > 80 * 1024 ? WM * WN / 4 : WM * WN / 2, NSTAGE * (BM + BN) * 128 > 80 * 1024 ? WM * WN / 4 : WM * WN / 2)))
void gemm4_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int LPT = (BM + BN) * 8 / NT;  // LDS-DMA instructions per thread per K tile
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  static_assert((BM + BN) * 8 % NT == 0, "tile chunks must divide over the threads");
  static_assert(NSTAGE == 2 || NSTAGE == 3, "ring of 2 or 3 stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int pid = blockIdx.x;
  const int xcd = pid & 7, slot = pid >> 3;
  const int xm_n = 8 / p.xn, ntx = (p.nt + p.xn - 1) / p.xn;
  const int m_tile = (slot / ntx) * xm_n + (xcd % xm_n), n_tile = (slot % ntx) * p.xn + xcd / xm_n;
  if (m_tile >= p.mt || n_tile >= p.nt) return;
  const int z = blockIdx.z;
  const int zo = z / p.batch_inner, zi = z % p.batch_inner;
  const bf16_t* __restrict__ A = (const bf16_t*)p.A + zo * p.strideA + zi * p.strideA2;
  const bf16_t* __restrict__ W = (const bf16_t*)p.W + zo * p.strideW + zi * p.strideW2;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  const bf16_t* src[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int id = (i * NW + wid) * 64 + lane;  // 16-B slot of the tile image
    const int row = id >> 3, phys = id & 7;
    const int c = phys ^ ((row >> 1) & 7);      // logical chunk stored at this physical slot
    if (row < BM) src[i] = A + a_row_offset(p, min(m0 + row, p.M - 1)) + c * 8;
    else src[i] = W + (long)min(n0 + row - BM, p.N - 1) * p.ldw + c * 8;
  }
  auto issue_piece = [&](int i, int kt, int stage) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + kt * 64), (lds_void_t*)(smem + stage * STAGE + (i * NW + wid) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) issue_piece(i, kt, stage);
  };

  const int wm = (wid / WN) * (BM / WM), wn = (wid % WN) * (BN / WN);
  const int fr = lane & 15, fq = lane >> 4;
  // fragment addresses inside a stage: rows wm + 16 j + fr (A) / BM + wn + 16 i + fr (W); the swizzle (row >> 1) & 7 only
  // depends on fr because the row bases are multiples of 16
  const int sw = (fr >> 1) & 7;
  const unsigned offA = (wm + fr) * 128, offW = (BM + wn + fr) * 128;
  const unsigned ch[2] = {(unsigned)((fq ^ sw) << 4), (unsigned)(((4 + fq) ^ sw) << 4)};
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fx[2][FM], fw[2][FN];     // two fragment sets: k-step g of the tile being multiplied lives in set g
  auto read0 = [&](unsigned sbase) {
#pragma unroll
    for (int j = 0; j < FM; ++j) fx[0][j] = *(const u32x4*)(smem + sbase + offA + ch[0] + j * 2048);
#pragma unroll
    for (int i = 0; i < FN; ++i) fw[0][i] = *(const u32x4*)(smem + sbase + offW + ch[0] + i * 2048);
  };
  auto read1 = [&](unsigned sbase) {
#pragma unroll
    for (int j = 0; j < FM; ++j) fx[1][j] = *(const u32x4*)(smem + sbase + offA + ch[1] + j * 2048);
#pragma unroll
    for (int i = 0; i < FN; ++i) fw[1][i] = *(const u32x4*)(smem + sbase + offW + ch[1] + i * 2048);
  };

  const int nk = p.K / 64;
  // LayerNorm-folding mode: the row statistics' loads go out before the K loop, consumed in the epilogue (as gemm2_kernel)
  constexpr bool LNK = (BN / WN == 64 || BN / WN == 32) && (FM >= 2 && FM <= 4) && sizeof(TO) == 2;
  f32x2 lnraw[LNK ? FM : 1][4];
  if constexpr (LNK) {
    if (p.a_stats) ln_issue<FM>(p.a_stats, p.a_nt, p.M, m0 + wm, fr, fq, lnraw);
    else if (p.r_stats) ln_issue<FM>(p.r_stats, p.r_nt, p.M, m0 + wm, fr, fq, lnraw);
  }
  // prologue: NSTAGE - 1 tiles in flight, tile 0 landed, the last stage filled, fragment set 0 of tile 0 read
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);
  if (NSTAGE - 1 <= nk) __builtin_amdgcn_s_waitcnt(waitcnt_vm((NSTAGE - 2) * LPT));
  else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
  __builtin_amdgcn_s_barrier();
  if (NSTAGE - 1 < nk) issue(NSTAGE - 1, NSTAGE - 1);
  read0(0);
  __builtin_amdgcn_s_waitcnt(waitcnt_lgkm0());
  int st = 0;
  if constexpr (ILV) {
    // Branch-free body, instruction stream interleaved by hand: every fragment read and every DMA piece sits BETWEEN two
    // MFMAs (pinned with sched_barrier: sched_group_barrier left the DMA pieces -- chained through M0 -- in one burst).  With one workgroup per CU the
    // eight waves move in lockstep from barrier to barrier; bursts (48 DMA pieces, then 64 fragment reads, then the MFMAs)
    // serialise the vector-memory queue, the LDS and the matrix pipe one after the other -- measured 2.1 x the free-running
    // time per K tile.  Past the last K tile the reads and the refill are repeated on clamped indices (the stage they
    // touch is never read again) instead of being branched around: one basic block per half.
    static_assert(FM * FN >= FM + FN + LPT, "one MFMA slot per fragment read and DMA piece");
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* sa = smem + st * STAGE;
      // half 1: MFMAs of k-step 0; the 8 fragment reads of k-step 1 go out one per MFMA slot
#pragma unroll
      for (int q = 0; q < FM * FN; ++q) {
        Mfma<TI>::run(fw[0][q / FM], fx[0][q % FM], acc[q / FM][q % FM]);
        if (q < FM) fx[1][q] = *(const u32x4*)(sa + offA + ch[1] + q * 2048);
        else if (q < FM + FN) fw[1][q - FM] = *(const u32x4*)(sa + offW + ch[1] + (q - FM) * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_waitcnt(waitcnt_vm((NSTAGE - 2) * LPT));
      __builtin_amdgcn_s_waitcnt(waitcnt_lgkm0());
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const int sn = st + 1 == NSTAGE ? 0 : st + 1;
      const unsigned char* sna = smem + sn * STAGE;
      const int kr = min(kt + NSTAGE, nk - 1);
      // half 2: MFMAs of k-step 1; the next tile's k-step-0 reads first (they are needed first), then the refill's DMA pieces
#pragma unroll
      for (int q = 0; q < FM * FN; ++q) {
        Mfma<TI>::run(fw[1][q / FM], fx[1][q % FM], acc[q / FM][q % FM]);
        if (q < FM) fx[0][q] = *(const u32x4*)(sna + offA + ch[0] + q * 2048);
        else if (q < FM + FN) fw[0][q - FM] = *(const u32x4*)(sna + offW + ch[0] + (q - FM) * 2048);
        else if (q - FM - FN < LPT) issue_piece(q - FM - FN, kr, st);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_waitcnt(waitcnt_lgkm0());
      st = sn;
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));     // the clamped refills of the last iterations must not outlive the workgroup's LDS
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned sb = st * STAGE;
    read1(sb);                                   // k-step 1 of this tile: in flight under the MFMAs of k-step 0
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) Mfma<TI>::run(fw[0][i], fx[0][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    // tile kt + 1 has landed (this wave's share; the barrier makes it everyone's); the newer tile may stay in flight
    if (kt + NSTAGE - 1 < nk) __builtin_amdgcn_s_waitcnt(waitcnt_vm((NSTAGE - 2) * LPT));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __builtin_amdgcn_s_waitcnt(waitcnt_lgkm0());  // set 1 is in registers: this wave has finished reading stage st
    __builtin_amdgcn_s_barrier();
    const int sn = st + 1 == NSTAGE ? 0 : st + 1;
    if (kt + NSTAGE < nk) issue(kt + NSTAGE, st);   // stage st is free: every wave has its fragments of tile kt in registers
    if (kt + 1 < nk) read0(sn * STAGE);          // k-step 0 of the next tile: in flight under the MFMAs of k-step 1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) Mfma<TI>::run(fw[1][i], fx[1][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(waitcnt_lgkm0());
    st = sn;
  }
  }
  if constexpr (LNK) {
    if (p.a_stats) { gemm_epilogue_ln<TO, FM, FN, 1>(p, acc, m0 + wm, n0 + wn, fr, fq, lnraw); return; }
    if (p.r_stats || p.stats_out) { gemm_epilogue_ln<TO, FM, FN, 2>(p, acc, m0 + wm, n0 + wn, fr, fq, lnraw); return; }
  }
  // the 16-fragment wave tiles carry the inference epilogue only (launch_gemm4 refuses Z / dropout / actbwd calls for them)
  gemm_epilogue<TO, FM, FN, sizeof(TO) == 2 && FM * FN <= 8, (FM * FN > 8)>(p, acc, z, m0 + wm, n0 + wn, fr, fq);
}

// ---------------------------------------------------------------------------------------------------
// gemm8_kernel: 256 x 256 x 64 tiles, ONE 8-wave workgroup per CU, the 8-phase schedule of cdna_hip_programming.md section 5
// ("The 256^2 8-phase template") on this library's operand layout.  tools/gemm8_probe.hip is the bare schedule: 1 305 TFLOP/s at
// 4096^3 and 1 303 at 16384 x 4096 x 3072 on random operands (the guide quotes 1 320-1 340 / 1 470 at 4096^3 / 8192^3) against
// 1 012-1 105 for the 128 x 128 / 192 x 128 kernels above -- round 3's first attempt at this schedule (exp/gemm_variants.inc,
// gemm8p: 902) lost a third of that to a branch around every staging step, per-fragment epilogues that fell into 528 bytes of
// scratch and 8-byte stores in 32-byte runs.  What is different here:
//   * the steady-state K tile is branch-free (the last two K tiles are separate copies without staging);
//   * the epilogue goes through LDS: accumulators are written as fp32 into a 64-row x 256-column block (two blocks of 64 KB =
//     the operand ring, alternating), then every wave takes whole 1 KB rows back -- a lane owns 4 consecutive columns of one
//     row per step -- so bias / activation / both LayerNorm forms / residual are a few registers of per-column constants and
//     per-row scalars, residual rows are read and output rows written as whole 512-byte runs, and the statistics of the
//     stored rows (64-column slabs, the layout the 128 x 128 kernels write) are four 16-lane reductions.
//   waves: 2 (wr, along M) x 4 (wc, along N); a wave owns 64 rows of EACH 128-row half of the X tile and 32 columns of EACH
//   128-column half of the W tile: four 64 x 32 quadrants (h, g), 16 MFMAs each per K tile = one PHASE; the two wave rows run
//   one barrier apart, so one multiplies while the other reads fragments and issues the next loads.
//   LDS: 2 K-tile buffers x {X0, X1, W0, W1} half-tiles of 128 rows x 128 bytes (16 KB, swizzled as everywhere here) = 128 KB.
//   phase p of K tile t (buffer t & 1):  fragment reads | ONE half-tile of LDS-DMA (2 pieces per thread) | [counted vmcnt,
//   phase 4 only] | s_barrier | 16 MFMAs | s_barrier
//     ph1: reads W0 (4, issued first) + X0 (8), lgkmcnt(8) retires the W0 reads before the barrier;  quadrant (0,0);  stages X1(t+1)
//     ph2: reads W1 (4);                                                                         quadrant (0,1);  stages W0(t+2)
//     ph3: reads X1 (8);                                                                         quadrant (1,1);  stages X0(t+2)
//     ph4: no reads (W0 is still in registers);  vmcnt(6) -> K tile t+1 has landed;              quadrant (1,0);  stages W1(t+2)
// Per output element the products and their order are gemm2_kernel's (k ascending, one 16x16x32 MFMA per 32): the plain
// outputs are bit-identical to the 128 x 128 kernels'; the row statistics of the LayerNorm forms are summed in another order.
// Calls it takes (launch_gemm8): 16-bit operands and output, N % 256 == 0, K % 64 == 0, K >= 128, one problem per launch,
// inference epilogues (EPIA 11 / 21 plain, 12 / 22 LayerNorm-operand, 13 LayerNorm-residual + statistics).
//
// SPLIT (round 6): the same tile, ring and schedule on MSMD_F16X2 split-pair operands (TI = f16_t; gemm2s_kernel's storage: a
// 128-byte row piece = [hi x 32 | lo x 32] of 32 logical k).  A K tile is then 32 logical k, its "k-step 0" fragments (chunks
// 0-3) are the hi planes and its "k-step 1" fragments (chunks 4-7) the lo planes: staging, LDS image and every fragment address
// are unchanged, only the multiplies differ -- per fragment pair  t = Wh.Al + Wl.Ah  (two MFMAs from a zero accumulator),
// acc = fma(t, 2^-11, acc)  (four vector FMAs, issued under the neighbouring MFMAs),  acc += Wh.Ah:  24 MFMAs per phase
// instead of 16 on the same LDS bytes.  The cross terms are folded once per K tile because a second accumulator set for the
// whole K loop (gemm2s_kernel's acc1) would be another 128 registers.  SPLIT = 1: C and the residual in split storage; 2: fp32.
// SPLIT + 4 (the caller states |W| < 32, MSMD_GEMM_W_BELOW_32): no fold at all -- the wave multiplies its hi-plane W fragments
// by 2^11 in registers (four v_pk_mul_f16 per fragment, exact below 32 in magnitude), so that  (2^11 Wh).Ah + Wh.AL + WL.Ah  is
// ONE sum in units of 2^-11 (AL / WL = the stored lo planes, already x 2^11), scaled back in the epilogue: 16 vector
// instructions per K tile instead of 128, one accumulator set, the matrix pipe 11-17 % busier (conv1 875 -> 779 us, QKV 81 ->
// 68, same box).  LayerNorm forms are not built for it (the split path runs LayerNorm as its own kernel): EPI = 1 only.
template <typename TI, int EPIA, int SPLIT = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8_kernel(const GemmArgs p) {
  typedef typename std::conditional<(SPLIT & 3) == 2, float, TI>::type TO;
  constexpr bool WS = (SPLIT & 4) != 0;   // W's hi plane scaled in registers, single accumulator
  constexpr int EPI = EPIA % 10, ACTK = EPIA / 10 - 1;
  static_assert(SPLIT == 0 || (EPI == 1 && __is_same(TI, f16_t)), "split operands: fp16 planes, plain epilogue");
  constexpr int HALF = 128 * 128;        // bytes of one half-tile
  constexpr int BUF = 4 * HALF;          // X0 X1 W0 W1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // Tile order: workgroups pid, pid + 8, ... share an XCD (one L2); each XCD takes one contiguous run of the row-major tile
  // list -- counts differ by at most one, which matters at ONE workgroup per CU (the (8 / xn) x xn XCD grid of the kernels
  // above leaves XCDs with 39 and 26 tiles of the 225 of 6400 x 2304: two rounds).  The grid is min(tiles, 256) workgroups:
  // a workgroup walks its XCD's run with the stride of that XCD's workgroup count (persistent: see the tile loop below).
  const int pid = blockIdx.x, ntiles = p.mt * p.nt, nwg = gridDim.x;
  const int xcd = pid & 7;
  const int run0 = (xcd < (ntiles & 7) ? xcd * ((ntiles >> 3) + 1) : (ntiles & 7) * ((ntiles >> 3) + 1) + (xcd - (ntiles & 7)) * (ntiles >> 3));
  const int run_n = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
  const int stride = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
  int tl = pid >> 3;                     // position in the XCD's run
  const bf16_t* __restrict__ A = (const bf16_t*)p.A;
  const bf16_t* __restrict__ W = (const bf16_t*)p.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // LDS-DMA sources: piece i (0, 1) of this wave inside a half-tile is its 1 KB chunk i * 8 + wid = rows 8 chunk .. + 7
  unsigned src[4][2];        // [X0 X1 W0 W1][piece]: byte offsets from A / W (the launcher refuses operands past 4 GB)
  int m0 = 0, n0 = 0;
  // (called twice per tile -- before the epilogue of the previous one for the first K tile, after it for the rest -- from a
  // lane id the compiler cannot match, so that the eight pointers are not kept across the epilogue: with the 128 accumulators
  // and the epilogue's rows live they were the spill)
  auto set_tile = [&](int t) {
    const int m_tile = (run0 + t) / p.nt, n_tile = (run0 + t) - m_tile * p.nt;
    m0 = m_tile * 256; n0 = n_tile * 256;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wid) * 8 + (ln >> 3), phys = ln & 7;
      const int c = phys ^ ((row >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        src[h][i] = (unsigned)((a_row_offset(p, min(m0 + h * 128 + row, p.M - 1)) + c * 8) * 2);
        src[2 + h][i] = (unsigned)(((long)min(n0 + h * 128 + row, p.N - 1) * p.ldw + c * 8) * 2);
      }
    }
  };
  const unsigned dma_base = wid * 1024;     // + buffer + half + piece * 8192
  auto stage = [&](int which, int kt, unsigned bufoff) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)(which < 2 ? A : W) + (size_t)(src[which][i] + (unsigned)kt * 128u)),
                                       (lds_void_t*)(smem + (dma_base + bufoff + which * HALF + i * 8192)), 16, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const unsigned ch0 = (fq ^ sw) << 4;
  // fragment addresses (k-step 0 / 1) inside buffer 0; the buffer is toggled by XOR BUF
  unsigned xa0 = (wr * 64 + fr) * 128 + ch0, xa1 = xa0 ^ 64;
  unsigned wa0 = 2 * HALF + (wc * 32 + fr) * 128 + ch0, wa1 = wa0 ^ 64;

  f32x4 acc[2][2][2][4];   // [h][g][i (W fragment)][j (X fragment)]
  u32x4 fx[2][4], fw0[2][2], fw1[2][2];   // [k-step][fragment]
  u32x4 fs0[2], fs1[2];                   // WS: the hi-plane W fragments x 2^11
  auto scale_w = [&](const u32x4 (&fw)[2][2], u32x4 (&fs)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f16x8 h = __builtin_bit_cast(f16x8, fw[0][i]);
      h = h * (f16_t)MSMD_SPLIT_SCALE;
      fs[i] = __builtin_bit_cast(u32x4, h);
    }
  };

  auto read_x = [&](int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fx[0][j] = *(const u32x4*)(smem + xa0 + h * HALF + j * 2048);
      fx[1][j] = *(const u32x4*)(smem + xa1 + h * HALF + j * 2048);
    }
  };
  auto read_w = [&](int g, u32x4 (&fw)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fw[0][i] = *(const u32x4*)(smem + wa0 + g * HALF + i * 2048);
      fw[1][i] = *(const u32x4*)(smem + wa1 + g * HALF + i * 2048);
    }
  };
  auto quadrant = [&](f32x4 (&a)[2][4], const u32x4 (&fw)[2][2], const u32x4 (&fs)[2]) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (SPLIT == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mfma<TI>::run(fw[ks][i], fx[ks][j], a[i][j]);
    } else if constexpr (WS) {
      // one sum in units of 2^-11: Wh.AL, WL.Ah, (2^11 Wh).Ah -- each accumulator is touched every 8th MFMA
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mfma<TI>::run(fw[0][i], fx[1][j], a[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mfma<TI>::run(fw[1][i], fx[0][j], a[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mfma<TI>::run(fs[i], fx[0][j], a[i][j]);
    } else {
      // [0] = hi planes, [1] = lo planes (x 2^11) of one 32-deep k-step; products in gemm2s_kernel's order (Wh.Al, then Wl.Ah)
      f32x4 t[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; Mfma<TI>::run(fw[0][i], fx[1][j], t[i][j]); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mfma<TI>::run(fw[1][i], fx[0][j], t[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[i][j][e] = fmaf(t[i][j][e], MSMD_SPLIT_INV, a[i][j][e]);
          Mfma<TI>::run(fw[0][i], fx[0][j], a[i][j]);
        }
      // issue order: the four FMAs of fold q sit in the shadow of an MFMA issued after the one that finished t[q] (left to
      // itself the scheduler put 17 of the 32 FMAs in one run with the matrix pipe idle behind an s_nop 5)
      __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
#define G8_BAR()                        \
  do {                                  \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)
#define G8_EBAR()                                      \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
  } while (0)

  // LayerNorm forms: the row statistics of the 32 rows this wave finishes (rows m0 + 64 ps + 8 wid + it) come in by LDS-DMA --
  // no registers across the K loop, and requested a whole K loop before they are read: the producer ran on other XCDs, the
  // first touch misses this L2.  One 16-byte piece = rows (2 rp, 2 rp + 1) of one 64-column slab (slab-major (slabs, M, 2)
  // fp32, M even); lane = 16 so + 4 ps + rp takes slab 4 q + so in piece q; 4 KB per wave behind the operand ring.
  const float* ln_stats = EPI == 2 ? p.a_stats : p.r_stats;
  const int ln_nt = EPI == 2 ? p.a_nt : p.r_nt;
  auto ln_request = [&]() {
    if constexpr (EPI >= 2) {
      if (ln_stats) {
        int ll = lane;
        asm volatile("" : "+v"(ll));
        const int m = min(m0 + ((ll >> 2) & 3) * 64 + wid * 8 + (ll & 3) * 2, p.M - 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q < ln_nt) {
            const int t = min(4 * q + (ll >> 4), ln_nt - 1);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(ln_stats + ((long)t * p.M + m) * 2),
                                             (lds_void_t*)(smem + (2 * BUF + wid * 4096 + q * 1024)), 16, 0, 0);
          }
        }
      }
    }
  };

  const int nk = p.K / (SPLIT ? 32 : 64);       // K tiles of 128 bytes per operand row
  typedef typename Vec4T<TO>::type V4;
  const bool has_r = p.R != nullptr;
  const bool wt = p.flags & 1;     // write-through stores (MSMD_GEMM_WRITE_THROUGH): the rows leave the L2 as they are stored

  // prologue of the first tile: K tile 0 whole, K tile 1 except X1 (which phase 1 of tile 0 stages)
  set_tile(tl);
  ln_request();
  stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
  stage(2, 1, BUF); stage(0, 1, BUF); stage(3, 1, BUF);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: wave row 1 runs one barrier behind row 0
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[h][g][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned cur = 0;   // byte offset of the buffer of K tile t
    // MODE 0: steady state; 1: K tile nk - 2 (only X1 of the last tile is left to stage); 2: the last K tile
    auto ktile = [&](int t, auto MODE_) {
      constexpr int MODE = decltype(MODE_)::value;
      // ---- phase 1
      read_w(0, fw0);
      __builtin_amdgcn_sched_barrier(0);
      read_x(0);
      if (MODE <= 1) stage(1, t + 1, cur ^ BUF);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the four W0 reads (issued first) have returned
      if constexpr (WS) scale_w(fw0, fs0);                 // ... in this wave's read segment, the other wave row is multiplying
      G8_BAR();
      quadrant(acc[0][0], fw0, fs0);
      G8_BAR();
      // ---- phase 2
      read_w(1, fw1);
      if (MODE == 0) stage(2, t + 2, cur);
      if constexpr (WS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        scale_w(fw1, fs1);
      }
      G8_BAR();
      quadrant(acc[0][1], fw1, fs1);
      G8_BAR();
      // ---- phase 3
      read_x(1);
      if (MODE == 0) stage(0, t + 2, cur);
      G8_BAR();
      quadrant(acc[1][1], fw1, fs1);
      G8_BAR();
      // ---- phase 4
      if (MODE == 0) stage(3, t + 2, cur);
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but the three youngest half-tiles: tile t + 1 is in
      if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      G8_BAR();
      quadrant(acc[1][0], fw0, fs0);
      G8_BAR();
      cur ^= BUF; xa0 ^= BUF; xa1 ^= BUF; wa0 ^= BUF; wa1 ^= BUF;
    };
    for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
    ktile(nk - 2, std::integral_constant<int, 1>{});
    ktile(nk - 1, std::integral_constant<int, 2>{});
    if (wr == 0) __builtin_amdgcn_s_barrier();   // row 0 pays back the stagger: every wave has read its last fragments
    if (nk & 1) { xa0 ^= BUF; xa1 ^= BUF; wa0 ^= BUF; wa1 ^= BUF; }   // the next tile starts in buffer 0 again

    // ---- epilogue through LDS, in the upper half of the ring (the lower half already takes the next tile's first K tile).
    // Pass ps = 2 h + wr' covers tile rows 64 ps .. 64 ps + 63: the four waves of wave row wr' write their two quadrants of
    // half h as fp32 (16-byte chunk c of row r at r * 1024 + ((c ^ (r & 15)) << 4): conflict-free for the fragment writes and
    // the row reads), then wave w finishes rows 8 w .. 8 w + 7 of the block.
    const int em0 = m0, en0 = n0;                    // this tile (m0 / n0 move on to the next one below)
    // everything the epilogue derives from the lane id comes from a copy the compiler cannot match with the K loop's: hoisted
    // out of the tile loop those values lived across the K loop and pushed its staging pointers into scratch
    int el = lane;
    asm volatile("" : "+v"(el));
    const int efr = el & 15, efq = el >> 4;
    const int ncol = en0 + el * 4;                   // this lane's four columns in every row it finishes
    f32x4 cbias = f32x4{0.f, 0.f, 0.f, 0.f}, cx = f32x4{1.f, 1.f, 1.f, 1.f}, cy = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) cbias = *(const f32x4*)(p.bias + ncol);
    if constexpr (EPI == 2) cx = *(const f32x4*)(p.w_colsum + ncol);
    if constexpr (EPI == 3) {
      if (p.r_stats) { cx = *(const f32x4*)(p.r_gamma + ncol); cy = *(const f32x4*)(p.r_beta + ncol); }
    }
    float mu_l = 0.f, rs_l = 1.f;                    // of row (lane & 31) of this wave's 32
    if constexpr (EPI >= 2) {
      if (ln_stats) {
        // ln_finish's association (the 128 x 128 kernels: lane group f holds slabs f, f + 4, f + 8, f + 12 and the tail
        // 16 + f, 20 + f, ...; groups combine as (0 + 1) + (2 + 3)), so that a row's moments are the same bits whichever kernel
        // its launch was routed to.  Both halves of the wave compute the same 32 rows.
        const int r32 = el & 31;
        const unsigned base = 2 * BUF + wid * 4096 + (((r32 >> 3) * 4 + ((r32 & 7) >> 1)) << 4) + ((r32 & 1) << 3);
        float Sf[4], Qf[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          f32x2 w[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t = f + 4 * q;
            w[q] = *(const f32x2*)(smem + base + (t >> 2) * 1024 + ((t & 3) << 8));
            if (t >= ln_nt) w[q] = f32x2{0.f, 0.f};
          }
          Sf[f] = (w[0][0] + w[1][0]) + (w[2][0] + w[3][0]);
          Qf[f] = (w[0][1] + w[1][1]) + (w[2][1] + w[3][1]);
        }
        if (ln_nt > 16) {             // more than 16 slabs per row: the rest from memory, serially
          const f32x2* rp = (const f32x2*)ln_stats + min(em0 + (r32 >> 3) * 64 + wid * 8 + (r32 & 7), p.M - 1);
#pragma unroll
          for (int f = 0; f < 4; ++f)
            for (int t = 16 + f; t < ln_nt; t += 4) { const f32x2 w = rp[(long)t * p.M]; Sf[f] += w[0]; Qf[f] += w[1]; }
        }
        const float S = (Sf[0] + Sf[1]) + (Sf[2] + Sf[3]), Q = (Qf[0] + Qf[1]) + (Qf[2] + Qf[3]);
        const float inv_cols = 1.0f / (float)(EPI == 2 ? p.K : p.N);
        mu_l = S * inv_cols;
        rs_l = rsqrtf(fmaxf(Q * inv_cols - mu_l * mu_l, 0.f) + p.ln_eps);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's statistics are in registers: its 4 KB may be refilled
      }
    }
    // the next tile of this workgroup: its first K tile goes out now and lands under the epilogue
    tl += stride;
    const bool more = tl < run_n;
    if (more) {
      set_tile(tl);
      stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
      ln_request();
    }
    auto write_block = [&](const f32x4 (&q0)[2][4], const f32x4 (&q1)[2][4]) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = j * 16 + efr, c = g * 32 + wc * 8 + i * 4 + efq;
            *(f32x4*)(smem + BUF + row * 1024 + ((c ^ efr) << 4)) = g == 0 ? q0[i][j] : q1[i][j];
          }
    };
    auto finish_block = [&](int ps) {
      if constexpr ((SPLIT & 3) == 2) {
        // fp32 C / residual: a lane's four columns are 16 bytes, a row is one 1 KB run
        f32x4 rr[8];
        if (has_r) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int m = min(em0 + ps * 64 + wid * 8 + it, p.M - 1);
            rr[it] = *(const f32x4*)((const float*)p.R + (long)m * p.ldr + ncol);
          }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = wid * 8 + it, m = em0 + ps * 64 + row;
          const f32x4 a = *(const f32x4*)(smem + BUF + row * 1024 + ((el ^ (row & 15)) << 4));
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = act_out_c<float, ACTK>(WS ? fmaf(a[e], MSMD_SPLIT_INV, cbias[e]) : a[e] + cbias[e], p.act);   // gemm2s_kernel's fp32 epilogue
            if (has_r) o[e] += rr[it][e];
          }
          if (m < p.M) {
            if (wt) store16_wt((float*)p.C + (long)m * p.ldc + ncol, __builtin_bit_cast(u32x4, o));
            else *(f32x4*)((float*)p.C + (long)m * p.ldc + ncol) = o;
          }
        }
      } else if constexpr ((SPLIT & 3) == 1) {
        // split C / residual rows ([hi x 32 | lo x 32] blocks, 2 ldc / 2 ldr fp16 per row).  Lanes 2 q and 2 q + 1 own the
        // eight logical columns 8 q .. 8 q + 7 of a row between them: the even lane moves their hi plane (16 bytes), the odd
        // lane their lo plane (the 16 bytes 64 further), the two exchange halves by DPP -- every wave instruction reads or
        // writes one whole 1 KB row piece instead of 8-byte pieces in 64-byte runs.
        const bool odd = el & 1;
        const long coff = 2L * en0 + ((el >> 3) << 6) + (((el >> 1) & 3) << 3) + (odd ? 32 : 0);
        auto swap1 = [](unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false); };   // quad_perm [1, 0, 3, 2]
        u32x4 rr[8];
        if (has_r) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int m = min(em0 + ps * 64 + wid * 8 + it, p.M - 1);
            rr[it] = *(const u32x4*)((const f16_t*)p.R + (long)m * 2 * p.ldr + coff);
          }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = wid * 8 + it, m = em0 + ps * 64 + row;
          const f32x4 a = *(const f32x4*)(smem + BUF + row * 1024 + ((el ^ (row & 15)) << 4));
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_split<ACTK>(WS ? fmaf(a[e], MSMD_SPLIT_INV, cbias[e]) : a[e] + cbias[e], p.act);
          if (has_r) {
            // even lane holds [h own | h partner], odd lane [l partner | l own]
            const unsigned g0 = swap1(odd ? rr[it][0] : rr[it][2]), g1 = swap1(odd ? rr[it][1] : rr[it][3]);
            const f16x2 h01 = __builtin_bit_cast(f16x2, odd ? g0 : rr[it][0]), h23 = __builtin_bit_cast(f16x2, odd ? g1 : rr[it][1]);
            const f16x2 l01 = __builtin_bit_cast(f16x2, odd ? rr[it][2] : g0), l23 = __builtin_bit_cast(f16x2, odd ? rr[it][3] : g1);
            v[0] += unsplit_f16x2(h01[0], l01[0]); v[1] += unsplit_f16x2(h01[1], l01[1]);
            v[2] += unsplit_f16x2(h23[0], l23[0]); v[3] += unsplit_f16x2(h23[1], l23[1]);
          }
          f16_t h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_f16x2(v[e], h[e], l[e]);
          const unsigned hp0 = __builtin_bit_cast(unsigned, f16x2{h[0], h[1]}), hp1 = __builtin_bit_cast(unsigned, f16x2{h[2], h[3]});
          const unsigned lp0 = __builtin_bit_cast(unsigned, f16x2{l[0], l[1]}), lp1 = __builtin_bit_cast(unsigned, f16x2{l[2], l[3]});
          const unsigned g0 = swap1(odd ? hp0 : lp0), g1 = swap1(odd ? hp1 : lp1);
          const u32x4 o = odd ? u32x4{g0, g1, lp0, lp1} : u32x4{hp0, hp1, g0, g1};
          if (m < p.M) {
            if (wt) store16_wt((f16_t*)p.C + (long)m * 2 * p.ldc + coff, o);
            else *(u32x4*)((f16_t*)p.C + (long)m * 2 * p.ldc + coff) = o;
          }
        }
      } else {
      // residual rows first: eight 512-byte runs per wave, in flight while the block is read back
      V4 rr[8];
      if (EPI != 2 && has_r) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int m = min(em0 + ps * 64 + wid * 8 + it, p.M - 1);
          rr[it] = *(const V4*)((const TO*)p.R + (long)m * p.ldr + ncol);
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = wid * 8 + it, m = em0 + ps * 64 + row;
        const f32x4 a = *(const f32x4*)(smem + BUF + row * 1024 + ((el ^ (row & 15)) << 4));
        float mu = 0.f, rs = 1.f;
        if constexpr (EPI >= 2) {
          mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu_l), ps * 8 + it));
          rs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs_l), ps * 8 + it));
        }
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = a[e];
          if constexpr (EPI == 2) x = rs * (x - mu * cx[e]);
          v[e] = act_out_c<TO, ACTK>(x + cbias[e], p.act);
          if constexpr (EPI == 3) v[e] += fmaf(((float)rr[it][e] - mu) * rs, cx[e], cy[e]);
          else if (has_r) v[e] += (float)rr[it][e];
        }
        const V4 o = pack4<TO>(v[0], v[1], v[2], v[3]);
        if (m < p.M) {
          if (wt) store8_wt((TO*)p.C + (long)m * p.ldc + ncol, __builtin_bit_cast(u32x2, o));
          else *(V4*)((TO*)p.C + (long)m * p.ldc + ncol) = o;
        }
        if constexpr (EPI == 3) {
          if (p.stats_out) {      // sums of what the consumer will read, per 64-column slab = 16 lanes
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float w = (float)o[e]; S += w; Q = fmaf(w, w, Q); }
            // the association of gemm_epilogue_ln_a: a lane's four columns are fragment i = lane bits 2-3, lane group fq = bits 0-1
#pragma unroll
            for (int d : {4, 8, 1, 2}) { S += __shfl_xor(S, d, 64); Q += __shfl_xor(Q, d, 64); }
            if ((el & 15) == 0 && m < p.M) *(f32x2*)(p.stats_out + ((long)((en0 >> 6) + (el >> 4)) * p.M + m) * 2) = f32x2{S, Q};
          }
        }
      }
      }
    };
    if (wr == 0) write_block(acc[0][0], acc[0][1]);
    G8_EBAR();
    finish_block(0);
    G8_EBAR();
    if (wr == 1) write_block(acc[0][0], acc[0][1]);
    G8_EBAR();
    finish_block(1);
    G8_EBAR();
    if (wr == 0) write_block(acc[1][0], acc[1][1]);
    G8_EBAR();
    finish_block(2);
    G8_EBAR();
    if (wr == 1) write_block(acc[1][0], acc[1][1]);
    G8_EBAR();
    finish_block(3);
    if (!more) break;
    // the upper half of the ring is free again once every wave has read its rows of the last block; the first K tile has
    // been in flight since before the epilogue: K tile 1 except X1 follows, and the loop's own counted waits take over
    G8_EBAR();
    set_tile(tl);
    stage(2, 1, BUF); stage(0, 1, BUF); stage(3, 1, BUF);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef G8_BAR
#undef G8_EBAR
}

#ifdef MSMD_EXPERIMENTAL
// Developer knobs exist ONLY in the experimental build (make EXP=1 -> libmsmd_hip_exp.so): the product library has no
// process-global state -- kernel variant and epilogue flags travel per call in `act` (include/msmd_hip.h).
int g_tuning[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
long* g_gemm_stamps = nullptr;
extern "C" int msmd_exp_set_stamps(long* p) { g_gemm_stamps = p; return 0; }
extern "C" int msmd_exp_set_tuning(int key, int value) {
  if (key < 0 || key >= 16) return 1;
  g_tuning[key] = value;
  return 0;
}
#include "exp/gemm_variants.inc"
#endif

// XCD grid over (M, N) tiles (p.mt, p.nt set).  With all 8 XCDs striped along M every L2 streams its own copy of the whole weight
// matrix from HBM, while the activation rows (the previous kernel's output) are still warm: an XCD of an
// (8 / xn) x xn grid reads 1 / xn of W and xn / 8 of A, so xn is picked per problem from  0.7 xn |A| + (8 / xn) |W|
// (the 0.7 fitted on the qkv shape, where 2 x 4 ties with 8 x 1 and both trail 4 x 2).  Forward step, same-graph A/B
// in both orders: 8 x 1 4.96 ms, 4 x 2 everywhere 4.89, this rule 4.88.  tuning key 7 forces xn = 1 / 2 / 4.
template <int BM, int BN>
static void gemm2_xcd_grid(GemmArgs& p) {
  int want_xn = 1;
  if (MSMD_TUNE(7) == 1 || MSMD_TUNE(7) == 2 || MSMD_TUNE(7) == 4) {
    want_xn = MSMD_TUNE(7);
  } else {
    const double a_bytes = 2.0 * p.M * (double)(p.rows_per_batch < p.M ? p.lda : p.K), w_bytes = 2.0 * p.N * (double)p.K;
    double best = 1e30;
    for (int xn = 1; xn <= 4; xn *= 2) {
      const double c = 0.7 * xn * a_bytes + (8.0 / xn) * w_bytes;
      if (c < best && p.nt >= xn) { best = c; want_xn = xn; }
    }
  }
  p.xn = p.nt >= want_xn ? want_xn : 1;
}

// gemm2p_kernel: 512 persistent workgroups (two per CU) over a grid of more than 512 tiles
template <typename TO, int BM, int BN, int WM, int WN, typename TI, int EPIA>
static int launch_gemm2p(GemmArgs& p, hipStream_t st) {
  constexpr int lds = 2 * (BM + BN) * 128;
  static bool attr_done = false;
  auto kfn = gemm2p_kernel<TO, BM, BN, WM, WN, TI, EPIA>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  p.stagger_ticks = 0;
#ifdef MSMD_EXPERIMENTAL
  p.stamps = nullptr;
#endif
  hipLaunchKernelGGL(kfn, dim3(512, 1, 1), dim3(WM * WN * 64), lds, st, p);
  MSMD_RETURN_LAST();
}

template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, bool PIPE = false, typename TI = bf16_t, bool STAG = false, int EPI = 0>
static int launch_gemm2(GemmArgs& p, int batch, hipStream_t st) {
  constexpr int lds = NSTAGE * (BM + BN) * 128;
  static bool attr_done = false;
  auto kfn = gemm2_kernel<TO, BM, BN, WM, WN, NSTAGE, PIPE, TI, STAG, EPI>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  p.mt = (p.M + BM - 1) / BM; p.nt = (p.N + BN - 1) / BN;
  gemm2_xcd_grid<BM, BN>(p);
  const int xm_n = 8 / p.xn;
  dim3 grid(((p.mt + xm_n - 1) / xm_n) * ((p.nt + p.xn - 1) / p.xn) * 8, 1, batch);
  // stagger (flags bit 2): only where the launch runs more than one round of two workgroups per CU
  p.stagger_ticks = ((p.flags & 4) && batch == 1 && NSTAGE * (BM + BN) * 128 <= 80 * 1024 && (long)p.mt * p.nt > 640)
                        ? (int)(100.0 * 0.5 * ((p.K / 64) * 0.5 + 3.0)) : 0;
#ifdef MSMD_EXPERIMENTAL
  p.stamps = g_gemm_stamps;
#endif
  hipLaunchKernelGGL(kfn, grid, dim3(WM * WN * 64), lds, st, p);
  MSMD_RETURN_LAST();
}

// One instantiation per epilogue family, picked per call (see gemm2_kernel's EPI): the plain inference epilogue, the two
// LayerNorm forms, and the everything-kernel for training calls (pre-activation copy / dropout / activation backward).
template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, bool PIPE = false, typename TI = bf16_t>
static int launch_gemm2_epi(GemmArgs& p, int batch, hipStream_t st) {
  constexpr bool LNK = (BN == 128 || BN == 64) && WN == 2 && (BM / WM / 16 == 2 || BM / WM / 16 == 3) && sizeof(TO) == 2;
  // every family also with the activation as a constant of the kernel (none / GELU: what the path launches; ELU and the
  // LayerNorm-residual form with an activation stay run-time)
  int epi = 0;
  if (LNK && p.a_stats) epi = 2;
  else if (LNK && (p.r_stats || p.stats_out)) epi = 3;
  else if (!p.a_stats && !p.r_stats && !p.stats_out && !p.Z && !(p.p_drop > 0.f) && !(p.flags & 8)) epi = 1;
  const int a = p.act == MSMD_ACT_NONE ? 10 : (p.act == MSMD_ACT_GELU && epi != 3) ? 20 : 0;
  if constexpr (LNK && PIPE && NSTAGE == 2 && WM * WN == 8 && BN == 128 && BM == 128) {
    // more than one round of two workgroups per CU: the persistent form (gemm2p_kernel); flags bit 4 = caller opts out (A/B).
    // The 128 x 128 tile's plain and LayerNorm-operand epilogues: the LayerNorm-residual one (its launches on the path are
    // single-round) and the 192-row tile do not fit 128 registers in this form (20-128 bytes of scratch).
    if (batch == 1 && epi >= 1 && a && !(p.flags & 16)) {
      p.mt = (p.M + BM - 1) / BM; p.nt = (p.N + BN - 1) / BN;
      gemm2_xcd_grid<BM, BN>(p);
      const int xm_n = 8 / p.xn;
      if (((p.mt + xm_n - 1) / xm_n) * ((p.nt + p.xn - 1) / p.xn) * 8 > 512) {
#define MSMD_EPI_CASE(E) case E: return launch_gemm2p<TO, BM, BN, WM, WN, TI, E>(p, st)
        switch (epi + a) { MSMD_EPI_CASE(11); MSMD_EPI_CASE(21); MSMD_EPI_CASE(12); MSMD_EPI_CASE(22); default: break; }
#undef MSMD_EPI_CASE
      }
    }
  }
#define MSMD_EPI_CASE(E) case E: return launch_gemm2<TO, BM, BN, WM, WN, NSTAGE, PIPE, TI, false, E>(p, batch, st)
  if constexpr (LNK) {
    switch (epi + a) { MSMD_EPI_CASE(2); MSMD_EPI_CASE(12); MSMD_EPI_CASE(22); MSMD_EPI_CASE(3); MSMD_EPI_CASE(13); default: break; }
  }
  switch (epi + a) { MSMD_EPI_CASE(1); MSMD_EPI_CASE(11); MSMD_EPI_CASE(21); default: break; }
  if constexpr (sizeof(TI) == 2 && !__is_same(TI, f16_t)) {     // the everything-epilogue is the training step's: bf16 only
    switch (epi + a) { MSMD_EPI_CASE(10); MSMD_EPI_CASE(20); default: break; }
  }
#undef MSMD_EPI_CASE
  return launch_gemm2<TO, BM, BN, WM, WN, NSTAGE, PIPE, TI, false, 0>(p, batch, st);
}

template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, typename TI = bf16_t, bool ILV = false>
static int launch_gemm4(GemmArgs& p, int batch, hipStream_t st) {
  constexpr int lds = NSTAGE * (BM + BN) * 128;
  static bool attr_done = false;
  auto kfn = gemm4_kernel<TO, BM, BN, WM, WN, NSTAGE, TI, ILV>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  if ((BM / WM / 16) * (BN / WN / 16) > 8 && (p.Z || p.p_drop > 0.f || (p.flags & 8))) return -1;   // lean epilogue: not a training call
  p.mt = (p.M + BM - 1) / BM; p.nt = (p.N + BN - 1) / BN;
  int want_xn = 1;     // XCD grid over (M, N) tiles: the cost model of launch_gemm2
  if (MSMD_TUNE(7) == 1 || MSMD_TUNE(7) == 2 || MSMD_TUNE(7) == 4) {
    want_xn = MSMD_TUNE(7);
  } else {
    const double a_bytes = 2.0 * p.M * (double)(p.rows_per_batch < p.M ? p.lda : p.K), w_bytes = 2.0 * p.N * (double)p.K;
    double best = 1e30;
    for (int xn = 1; xn <= 4; xn *= 2) {
      const double c = 0.7 * xn * a_bytes + (8.0 / xn) * w_bytes;
      if (c < best && p.nt >= xn) { best = c; want_xn = xn; }
    }
  }
  p.xn = p.nt >= want_xn ? want_xn : 1;
  const int xm_n = 8 / p.xn;
  dim3 grid(((p.mt + xm_n - 1) / xm_n) * ((p.nt + p.xn - 1) / p.xn) * 8, 1, batch);
  hipLaunchKernelGGL(kfn, grid, dim3(WM * WN * 64), lds, st, p);
  MSMD_RETURN_LAST();
}

// One persistent workgroup per CU: the grid cap is the device's CU count (256 on MI355X; the tile-run logic of the kernel
// works for any workgroup count, the shape rules below are fitted on 256).
static int gemm8_workgroup_cap() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

// gemm8_kernel: which calls it takes, and the launch.  Returns -1 for a call it does not take.
static bool gemm8_takes(const GemmArgs& p, int batch, int osz) {
  if (osz != 2 || batch != 1 || p.batch_inner != 1 || (p.N % 256) || (p.K % 64) || p.K < 128 || !p.vec_ok) return false;
  if (p.Z || p.p_drop > 0.f || (p.flags & 8)) return false;               // inference epilogues only
  if (((uintptr_t)p.bias & 15) || ((uintptr_t)p.C & 7) || ((uintptr_t)p.R & 7)) return false;
  {   // the staging offsets are 32-bit byte offsets from A / W
    const long a_rows = p.rows_per_batch < p.M ? ((long)((p.M - 1) / p.rows_per_batch) * p.a_batch_stride + (long)(p.rows_per_batch - 1) * p.lda) : (long)(p.M - 1) * p.lda;
    if ((a_rows + p.K) * 2 >= (1L << 32) || ((long)(p.N - 1) * p.ldw + p.K) * 2 >= (1L << 32)) return false;
  }
  const int epi = p.a_stats ? 2 : (p.r_stats || p.stats_out) ? 3 : 1;
  if ((p.a_stats || p.r_stats) && ((p.M & 1) || ((uintptr_t)p.a_stats & 15) || ((uintptr_t)p.r_stats & 15))) return false;   // row pairs by 16-byte LDS-DMA
  if (epi == 2 && (p.R || !p.bias || !p.w_colsum)) return false;
  if (epi == 3 && (!p.R || !p.bias || p.act != MSMD_ACT_NONE)) return false;
  return p.act == MSMD_ACT_NONE || p.act == MSMD_ACT_GELU;
}

// ... and which of those it wins (hot, isolated, tools/bench_gemm_variants.py SHAPES=guide|conv|sampler|encoder; us for the
// 128 x 128 / 192 x 128 kernels -> this one): the launch must fill its last round of 256 one-per-CU workgroups and, with a short K
// (12 K tiles or fewer: 13 us of prologue + epilogue per round against 1.4 us per K tile), nearly completely.
//   wins   conv1-4 (204768 ... 25568 x 512 x 1536: 375 -> 323, 198 -> 186, 92 -> 85, 52 -> 44), 6400 x 2304 x 768 (33 -> 29.4),
//          21312 x 1536 x 512 (50 -> 47), 12800 x 2304 x 768 (59 -> 57), HuBERT-large 15968 x {3072, 1024, 4096} x 1024 (114 -> 109,
//          43 -> 37, 153 -> 143) and 15968 x 1024 x 4096 (133 -> 103), 16384 x 4096 x 3072 (364 -> 311 = 1 326 TFLOP/s)
//   loses  under-filled rounds: 6400 x 768 x {768, 3072} (75 tiles), 6400 x 3072 x 768 (300: 40 -> 52), 12800 x 3072 x 768 (600: 80 -> 86),
//          21312 x 512 x {512, 2048} (168: 22.6 -> 26, 49 -> 51), 21312 x 2048 x 512 (672 = 2.6 rounds, K = 512: 67 -> 69)
static bool gemm8_wins(int M, int N, int K) {
  const long tiles = (long)((M + 255) / 256) * (N / 256);
  if (tiles < 192) return false;
  const double fill = (double)tiles / (256.0 * (double)((tiles + 255) / 256));
  return fill >= 0.75 && (K >= 1024 || fill >= 0.878);
}

template <typename TI, int EPIA>
static int launch_gemm8_e(GemmArgs& p, hipStream_t st) {
  constexpr int lds = 2 * 4 * 128 * 128 + (EPIA % 10 >= 2 ? 8 * 4096 : 0);   // 128 KB operand ring (its upper half = the epilogue's row block) + the LayerNorm forms' row statistics
  static bool attr_done = false;
  auto kfn = gemm8_kernel<TI, EPIA>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  p.mt = (p.M + 255) / 256; p.nt = p.N / 256;
  const int tiles = p.mt * p.nt;
  // Up to two rounds the launch ends in one burst of output rows: with write-through stores they leave the L2s while the
  // epilogue is still running instead of at the end-of-kernel write-back (6400 x 2304 x 768: 30.3-31.2 -> 28.1-29.3 us, 21312 x
  // 1536 x 512: 50.4 -> 47.3; many-round launches are indifferent or lose: 15968 x 3072 x 1024 101 -> 110)
  if (tiles <= 512) p.flags |= 1;
  hipLaunchKernelGGL(kfn, dim3(tiles < gemm8_workgroup_cap() ? tiles : gemm8_workgroup_cap(), 1, 1), dim3(512), lds, st, p);
  MSMD_RETURN_LAST();
}

template <typename TO, typename TI>
static int launch_gemm8(GemmArgs& p, int batch, hipStream_t st) {
  if constexpr (sizeof(TO) != 2) {
    return -1;
  } else {
    if (!gemm8_takes(p, batch, 2)) return -1;
    const int epi = p.a_stats ? 2 : (p.r_stats || p.stats_out) ? 3 : 1;
    switch (epi + (p.act == MSMD_ACT_GELU ? 20 : 10)) {
      case 11: return launch_gemm8_e<TI, 11>(p, st);
      case 21: return launch_gemm8_e<TI, 21>(p, st);
      case 12: return launch_gemm8_e<TI, 12>(p, st);
      case 22: return launch_gemm8_e<TI, 22>(p, st);
      case 13: return launch_gemm8_e<TI, 13>(p, st);
      default: return -1;
    }
  }
}

// gemm8_kernel on split-pair operands (SPLIT): the calls it takes, the rule, the launch.  p carries fp16 strides for A / W
// (doubled by gemm_impl) and logical ldc / ldr.
static bool gemm8s_takes(const GemmArgs& p, int batch, bool split_out) {
  if (batch != 1 || p.batch_inner != 1 || (p.N % 256) || (p.K % 32) || p.K < 64 || !p.vec_ok) return false;
  if (((uintptr_t)p.bias & 15) || ((uintptr_t)p.C & 15) || ((uintptr_t)p.R & 15)) return false;
  if (split_out ? ((p.ldc % 32) || (p.R && (p.ldr % 32))) : ((p.ldc % 4) || (p.R && (p.ldr % 4)))) return false;
  const long a_rows = p.rows_per_batch < p.M ? ((long)((p.M - 1) / p.rows_per_batch) * p.a_batch_stride + (long)(p.rows_per_batch - 1) * p.lda) : (long)(p.M - 1) * p.lda;
  if ((a_rows + 2L * p.K) * 2 >= (1L << 32) || ((long)(p.N - 1) * p.ldw + 2L * p.K) * 2 >= (1L << 32)) return false;   // 32-bit staging offsets
  return p.act == MSMD_ACT_NONE || p.act == MSMD_ACT_GELU;
}
// ... and which of those it wins (tools/bench_gemm_split.py, us, 128 x 128 gemm2s_kernel -> this kernel folding -> fold-free with
// MSMD_GEMM_W_BELOW_32; fp32 output, GELU).  Three MFMAs per fragment pair on the same staged bytes: a K tile (32 logical k)
// takes 1.5 x the 16-bit kernel's (64 k) while prologue / epilogue cost per round is the same, so under-filled rounds cost
// relatively less than gemm8_wins() charges them.
//   conv1-4 204768 ... 25568 x 512 x 1536 (1 600 ... 200 tiles): 951 -> 870 -> 791, 481 -> 482 -> 438, 254 -> 240 -> 214, 133 -> 114 -> 100
//   6400 x 2304 x 768 (225): 78 -> 76 -> 66;  21312 x 1536 x 512 (504): 118 -> 110 -> 101;  21312 x 2048 x 512 (672): 158 -> 153 -> 139
//   21312 x 512 x 2048 (168, fill 0.66): 135 -> 141 -> 120;  21312 x 512 x 512 (168): 47.9 -> 48.8 -> 44.0
//   loses: 6400 x 3072 x 768 (300, fill 0.59): 100 -> 125 -> 114;  75-100 tiles: 6400 x 768 x {768, 3072} 38 -> 58 -> 52, 114 -> 171 -> 148,
//          12768 x 512 x 1024 49 -> 73 -> 64
static bool gemm8s_wins(int M, int N, int K, bool w_below_32) {
  const long tiles = (long)((M + 255) / 256) * (N / 256);
  if (tiles < 160) return false;
  const double fill = (double)tiles / (256.0 * (double)((tiles + 255) / 256));
  return fill >= (w_below_32 ? 0.65 : 0.75);
}
template <typename TO, int ACTK, int WS>
static int launch_gemm8s(GemmArgs& p, hipStream_t st) {
  constexpr int lds = 2 * 4 * 128 * 128;
  constexpr int SPLIT = (sizeof(TO) == 4 ? 2 : 1) + 4 * WS;
  static bool attr_done = false;
  auto kfn = gemm8_kernel<f16_t, 11 + 10 * ACTK, SPLIT>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  p.mt = (p.M + 255) / 256; p.nt = p.N / 256;
  const int tiles = p.mt * p.nt;
  if (tiles <= 512) p.flags |= 1;      // write-through stores for launches of up to two rounds (launch_gemm8_e)
  hipLaunchKernelGGL(kfn, dim3(tiles < gemm8_workgroup_cap() ? tiles : gemm8_workgroup_cap(), 1, 1), dim3(512), lds, st, p);
  MSMD_RETURN_LAST();
}

template <typename TO, int BM, int BN, int WM, int WN, int NSTAGE, int ACTK = -1, int WSK = -1>
static int launch_gemm2s(GemmArgs& p, int batch, hipStream_t st) {
  if constexpr (WSK == -1) {
    return (p.flags & 64) ? launch_gemm2s<TO, BM, BN, WM, WN, NSTAGE, ACTK, 1>(p, batch, st)
                          : launch_gemm2s<TO, BM, BN, WM, WN, NSTAGE, ACTK, 0>(p, batch, st);
  } else {
  if constexpr (ACTK == -1 && NSTAGE * (BM + BN) * 128 <= 80 * 1024) {      // the routed tiles: the activation as a constant of the kernel
    if (p.act == MSMD_ACT_NONE) return launch_gemm2s<TO, BM, BN, WM, WN, NSTAGE, MSMD_ACT_NONE, WSK>(p, batch, st);
    if (p.act == MSMD_ACT_GELU) return launch_gemm2s<TO, BM, BN, WM, WN, NSTAGE, MSMD_ACT_GELU, WSK>(p, batch, st);
  }
  constexpr int lds = NSTAGE * (BM + BN) * 128;
  static bool attr_done = false;
  auto kfn = gemm2s_kernel<TO, BM, BN, WM, WN, NSTAGE, ACTK, WSK != 0>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  p.mt = (p.M + BM - 1) / BM; p.nt = (p.N + BN - 1) / BN;
  int want_xn = 1;
  if (MSMD_TUNE(7) == 1 || MSMD_TUNE(7) == 2 || MSMD_TUNE(7) == 4) {
    want_xn = MSMD_TUNE(7);
  } else {   // same cost model as launch_gemm2 (bytes are 4 per logical element here; the ratio is what matters)
    const double a_bytes = 4.0 * p.M * (double)(p.rows_per_batch < p.M ? p.lda / 2 : p.K), w_bytes = 4.0 * p.N * (double)p.K;
    double best = 1e30;
    for (int xn = 1; xn <= 4; xn *= 2) {
      const double c = 0.7 * xn * a_bytes + (8.0 / xn) * w_bytes;
      if (c < best && p.nt >= xn) { best = c; want_xn = xn; }
    }
  }
  p.xn = p.nt >= want_xn ? want_xn : 1;
  const int xm_n = 8 / p.xn;
  dim3 grid(((p.mt + xm_n - 1) / xm_n) * ((p.nt + p.xn - 1) / p.xn) * 8, 1, batch);
  hipLaunchKernelGGL(kfn, grid, dim3(WM * WN * 64), lds, st, p);
  MSMD_RETURN_LAST();
  }
}

template <typename TO>
static int dispatch_gemm2s(GemmArgs& p, int batch, hipStream_t st, int variant) {
  switch (variant) {
    case 1: return launch_gemm2s<TO, 128, 128, 4, 2, 2>(p, batch, st);   // 64 KB, 2 workgroups / CU (default)
    case 5: return launch_gemm2s<TO, 64, 64, 2, 2, 4>(p, batch, st);     // small grids
    case 14: return launch_gemm2s<TO, 256, 64, 8, 1, 2>(p, batch, st);   // narrow outputs (N <= 64): the positional conv
    case 80:                                                             // 256 x 256 tiles, 8-phase schedule (gemm8_kernel, SPLIT)
      if (!gemm8s_takes(p, batch, sizeof(TO) == 2)) return -1;
      if (p.flags & 64) return p.act == MSMD_ACT_GELU ? launch_gemm8s<TO, 1, 1>(p, st) : launch_gemm8s<TO, 0, 1>(p, st);   // MSMD_GEMM_W_BELOW_32
      return p.act == MSMD_ACT_GELU ? launch_gemm8s<TO, 1, 0>(p, st) : launch_gemm8s<TO, 0, 0>(p, st);
#ifdef MSMD_EXPERIMENTAL
    case 2: return launch_gemm2s<TO, 128, 128, 4, 2, 4>(p, batch, st);   // 128 KB, deep ring
    case 3: return launch_gemm2s<TO, 128, 128, 2, 2, 2>(p, batch, st);   // 4 waves of 64 x 64
    case 4: return launch_gemm2s<TO, 256, 128, 4, 2, 3>(p, batch, st);   // 144 KB, 64 x 64 wave tiles
    case 6: return launch_gemm2s<TO, 128, 64, 2, 2, 3>(p, batch, st);
    case 7: return launch_gemm2s<TO, 64, 128, 2, 2, 3>(p, batch, st);
    case 8: return launch_gemm2s<TO, 128, 128, 2, 2, 4>(p, batch, st);
    case 9: return launch_gemm2s<TO, 128, 128, 4, 2, 3>(p, batch, st);   // 96 KB
    case 10: return launch_gemm2s<TO, 128, 256, 2, 4, 3>(p, batch, st);  // 144 KB
    case 11: return launch_gemm3<TO, f16_t, 2, 128, 128, 4, 2, 4>(p, batch, st);   // 1 workgroup / CU, pipelined reads
    case 12: return launch_gemm3<TO, f16_t, 2, 128, 128, 4, 2, 3>(p, batch, st);
    case 13: return launch_gemm3<TO, f16_t, 2, 128, 128, 2, 4, 4>(p, batch, st);
    case 15: return launch_gemm3<TO, f16_t, 2, 128, 128, 4, 2, 4, 4>(p, batch, st);   // + 4 loader waves
#endif
    default: return -1;
  }
}

// Product variants: 14 = 256 x 64 for narrow outputs; 17 = 128 x 128, 8 waves (4 x 2), 2-stage ring, fragment reads of both k-steps issued first (default
// once the grid fills the chip); 13 = the same tile with the compiler's own read / multiply interleave; 9 / 12 = 64 x 64
// tiles with a 4- / 2-stage ring for grids that would not fill the chip.  Every other family that was built and measured
//  lives in exp/gemm_variants.inc and is compiled only with -DMSMD_EXPERIMENTAL.
template <typename TO>
static int dispatch_gemm2(GemmArgs& p, int batch, hipStream_t st, int variant) {
  switch (variant) {
    case 9: return launch_gemm2_epi<TO, 64, 64, 2, 2, 4>(p, batch, st);
    case 12: return launch_gemm2_epi<TO, 64, 64, 2, 2, 2>(p, batch, st);
    case 13: return launch_gemm2<TO, 128, 128, 4, 2, 2>(p, batch, st);
    case 14: return launch_gemm2_epi<TO, 256, 64, 8, 1, 2, true>(p, batch, st);   // narrow outputs (N <= 64): 8 waves of 32 x 64
    case 15: return launch_gemm2_epi<TO, 192, 128, 4, 2, 2, true>(p, batch, st);  // tall grids (M >= 16 k): 80 KB, still 2 workgroups / CU
    case 17: return launch_gemm2_epi<TO, 128, 128, 4, 2, 2, true>(p, batch, st);
    case 66: return launch_gemm2<TO, 128, 128, 4, 2, 2, true>(p, batch, st);       // 17 as ONE kernel with every epilogue (round 3's form)
    case 80: return launch_gemm8<TO, bf16_t>(p, batch, st);                        // 256 x 256, 8-phase schedule, one workgroup per CU
    // v4 kernels (fragment reads pipelined inside the wave, barrier between the k-steps):
    case 60: return launch_gemm4<TO, 256, 128, 4, 2, 3>(p, batch, st);                   // 144 KB, 1 workgroup / CU, 8 waves of 64 x 64
    case 61: return launch_gemm4<TO, 256, 128, 4, 2, 2>(p, batch, st);                   // 96 KB
    case 62: return launch_gemm4<TO, 128, 128, 4, 2, 2>(p, batch, st);                   // 64 KB, 2 workgroups / CU: variant 17's tile
    case 63: return launch_gemm4<TO, 256, 128, 4, 2, 3, bf16_t, true>(p, batch, st);     // 60 with the refill's DMA between the MFMAs
    case 64: return launch_gemm4<TO, 256, 128, 4, 2, 2, bf16_t, true>(p, batch, st);     // 63 with a 2-stage ring (96 KB)
#ifdef MSMD_EXPERIMENTAL
    case 41: return launch_gemm2<TO, 128, 128, 4, 2, 2, true, bf16_t, true>(p, batch, st);   // 17 with waves 4-7 staggered: -25 % (5c)
    case 40: return launch_gemm8p<TO>(p, batch, st);   // 256 x 256, 8-phase schedule, one workgroup per CU (round 3: slower, see exp/)
    case 1: return launch_gemm2<TO, 128, 128, 2, 2, 2>(p, batch, st);
    case 2: return launch_gemm2<TO, 128, 128, 2, 2, 3>(p, batch, st);
    case 3: return launch_gemm2<TO, 128, 128, 2, 2, 4>(p, batch, st);
    case 4: return launch_gemm2<TO, 256, 128, 4, 2, 3>(p, batch, st);
    case 5: return launch_gemm2<TO, 256, 256, 2, 4, 2>(p, batch, st);
    case 6: return launch_gemm2<TO, 128, 256, 2, 4, 3>(p, batch, st);
    case 7: return launch_gemm2<TO, 64, 128, 1, 4, 4>(p, batch, st);
    // more, smaller tiles with a deeper ring for the M = 6400 encoder shapes (round 3: see DESIGN 5c for the numbers)
    case 48: return launch_gemm2<TO, 128, 64, 4, 2, 3, true>(p, batch, st);   // 72 KB: 2 workgroups / CU, 2 stages in flight each
    case 49: return launch_gemm2<TO, 128, 64, 4, 2, 2, true>(p, batch, st);   // 48 KB: 3 workgroups / CU
    case 50: return launch_gemm2<TO, 64, 128, 2, 4, 3, true>(p, batch, st);
    case 51: return launch_gemm2<TO, 128, 64, 4, 2, 4, true>(p, batch, st);   // 96 KB: 1 workgroup / CU, 3 stages in flight
    // positional-conv candidates (N = 48 per group, 6400 x 48 x 6144 x 16 groups; variant 9: 140 us, 12: 120 us):
    case 42: return launch_gemm2<TO, 256, 64, 4, 1, 2>(p, batch, st);         // 94 us
    case 43: return launch_gemm2<TO, 256, 64, 4, 1, 2, true>(p, batch, st);   // 86 us   (product variant 14, 8 waves: 85 us)
    case 44: return launch_gemm2<TO, 128, 64, 4, 1, 2, true>(p, batch, st);   // 124 us
    case 46: return launch_gemm2<TO, 256, 64, 4, 1, 3, true>(p, batch, st);   // 148 us (one workgroup per CU)
    case 8: return launch_gemm2<TO, 128, 64, 4, 1, 4>(p, batch, st);          // 190 us
    case 10: return launch_gemm2<TO, 128, 64, 2, 2, 3>(p, batch, st);
    case 11: return launch_gemm2<TO, 64, 128, 2, 2, 3>(p, batch, st);
    case 47: return launch_gemm2<TO, 128, 64, 2, 2, 2>(p, batch, st);
    case 56: return launch_gemm2<TO, 64, 128, 2, 2, 2>(p, batch, st);
    case 16: return launch_gemm2<TO, 128, 128, 2, 4, 2>(p, batch, st);
    case 18: return launch_gemm2<TO, 128, 128, 2, 4, 2, true>(p, batch, st);
    case 19: return launch_gemm2<TO, 64, 64, 2, 2, 2, true>(p, batch, st);
    case 20: return launch_gemm2<TO, 64, 64, 2, 2, 4, true>(p, batch, st);
    case 21: return launch_gemm2<TO, 128, 128, 2, 2, 2, true>(p, batch, st);
    case 22: return launch_gemm2<TO, 256, 128, 4, 2, 3, true>(p, batch, st);
    case 23: return launch_gemm2<TO, 128, 128, 4, 2, 3, true>(p, batch, st);
    case 24: return launch_gemm2<TO, 256, 256, 2, 4, 2, true>(p, batch, st);
    case 25: return launch_gemm2<TO, 256, 256, 4, 2, 2, true>(p, batch, st);
    case 26: return launch_gemm2<TO, 128, 256, 2, 4, 3, true>(p, batch, st);
    case 27: return launch_gemm2<TO, 256, 128, 4, 2, 2, true>(p, batch, st);
    case 28: return launch_gemm2k<TO>(p, batch, st);
    case 37: return launch_gemm2<TO, 128, 96, 2, 2, 2, true>(p, batch, st);
    case 38: return launch_gemm2<TO, 128, 96, 2, 2, 3, true>(p, batch, st);
    case 30: return launch_gemm3<TO, bf16_t, 1, 256, 128, 4, 2, 3>(p, batch, st);
    case 31: return launch_gemm3<TO, bf16_t, 1, 256, 256, 2, 4, 2>(p, batch, st);
    case 32: return launch_gemm3<TO, bf16_t, 1, 128, 128, 4, 2, 4>(p, batch, st);
    case 33: return launch_gemm3<TO, bf16_t, 1, 128, 256, 2, 4, 3>(p, batch, st);
    case 34: return launch_gemm3<TO, bf16_t, 1, 256, 128, 4, 2, 3, 4>(p, batch, st);   // + 4 loader waves
    case 35: return launch_gemm3<TO, bf16_t, 1, 128, 128, 4, 2, 4, 4>(p, batch, st);
    case 36: return launch_gemm3<TO, bf16_t, 1, 128, 256, 2, 4, 3, 4>(p, batch, st);
#endif
    default: return -1;
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

template <typename TO>
static int dispatch_gemm2_f16(GemmArgs& p, int batch, hipStream_t st, int variant) {
  switch (variant) {
    case 9: return launch_gemm2_epi<TO, 64, 64, 2, 2, 4, false, f16_t>(p, batch, st);
    case 12: return launch_gemm2_epi<TO, 64, 64, 2, 2, 2, false, f16_t>(p, batch, st);
    case 14: return launch_gemm2_epi<TO, 256, 64, 8, 1, 2, true, f16_t>(p, batch, st);
    case 15: return launch_gemm2_epi<TO, 192, 128, 4, 2, 2, true, f16_t>(p, batch, st);
    case 17: return launch_gemm2_epi<TO, 128, 128, 4, 2, 2, true, f16_t>(p, batch, st);
    case 80: return launch_gemm8<TO, f16_t>(p, batch, st);
    default: return -1;
  }
}

// Between 9 600 and 16 000 rows (the training step's M = 12 800: both windows in one batch) the tile follows how full the
// LAST round of 512 workgroups is: 12800 x 768 is 600 tiles of 128 x 128 (one full round + 88 stragglers) but 402 of
// 192 x 128 (one round).  Measured (tools/bench_gemm_variants.py, SHAPES=train): x 768 x 3072 78.9 -> 68.0 us, x 2304 x 768
// 66.1 -> 60.7, x 768 x 768 a tie, x 3072 x 768 79.6 -> 82.4 (kept on 128 x 128 by this rule); 1.2 = the 192-row tile's
// advantage per round-slot on those shapes.
static bool tall_rounds_favour_192(int M, long tiles128, long tiles192) {
  if (M < 9600 || tiles192 < 256) return false;
  const double fill128 = (double)tiles128 / (512.0 * (double)((tiles128 + 511) / 512));
  const double fill192 = (double)tiles192 / (512.0 * (double)((tiles192 + 511) / 512));
  return 1.2 * fill192 > 1.03 * fill128;
}

static int gemm_impl(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N, int K,
                     int in_dtype, int out_dtype, long lda, int rows_per_batch, long a_batch_stride, long ldw, long ldc,
                     long ldr, int act, int batch, long strideA, long strideW, long strideC, long strideBias,
                     long strideR, int batch_inner, long strideA2, long strideW2, long strideC2, msmd_stream_t stream,
                     void* z_out = nullptr, float p_drop = 0.f, const unsigned long* rng = nullptr, unsigned site = 0,
                     int internal_flags = 0) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch_inner <= 0 || !A || !W || !C) return 1;
  const int hint = (act >> 8) & 0xff;  // caller-chosen kernel variant (host-side autotune), 0 = heuristic below
  const int flags = ((act >> 16) & 0x7) | (((act >> 19) & 1) << 4) | (((act >> 20) & 1) << 5) | (((act >> 21) & 1) << 6) | internal_flags;  // MSMD_GEMM_WRITE_THROUGH / MSMD_GEMM_PAIRED_STORES (include/msmd_hip.h)
  act &= 0xff;
  if (in_dtype == MSMD_F16X2) {
    // split-pair operands: logical sizes in, fp16 strides (x 2) into the kernel; 32-element blocks must stay whole
    if (out_dtype != MSMD_F32 && out_dtype != MSMD_F16X2) return 1;
    if (K % 32 || lda % 32 || ldw % 32 || a_batch_stride % 32 || strideA % 32 || strideW % 32 || strideA2 % 32 ||
        strideW2 % 32 || z_out || p_drop != 0.f)
      return 1;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15)) return 1;
    if (rows_per_batch <= 0) rows_per_batch = M;
    GemmArgs p;
    p.A = A; p.W = W; p.bias = bias; p.R = residual; p.C = C;
    p.M = M; p.N = N; p.K = K;
    p.lda = 2 * lda; p.rows_per_batch = rows_per_batch; p.a_batch_stride = 2 * a_batch_stride;
    p.ldw = 2 * ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
    p.inv_rpb = 1.0f / (float)rows_per_batch;
    p.strideA = 2 * strideA; p.strideW = 2 * strideW; p.strideC = strideC; p.strideBias = strideBias; p.strideR = strideR;
    p.batch_inner = batch_inner; p.strideA2 = 2 * strideA2; p.strideW2 = 2 * strideW2; p.strideC2 = strideC2;
    p.Z = nullptr; p.p_drop = 0.f; p.rng = nullptr; p.site = 0; p.xn = 1; p.flags = flags & (1 | 64);
    if (out_dtype == MSMD_F16X2) {
      if ((N & 3) || ldc % 32 || strideC % 32 || strideC2 % 32 || (residual && (ldr % 32 || strideR % 32))) return 1;
      if (bias && (((uintptr_t)bias & 15) || (strideBias & 3))) return 1;
      p.vec_ok = 1;
    } else {
      p.vec_ok = (ldc % 4 == 0) && (strideC % 4 == 0) && (strideC2 % 4 == 0) && (((uintptr_t)C % 16) == 0) &&
                 (!residual || ((ldr % 4 == 0) && (strideR % 4 == 0) && (((uintptr_t)residual % 16) == 0)));
    }
    hipStream_t st = (hipStream_t)stream;
    const int nz = batch * batch_inner;
    int variant = MSMD_TUNE(3) > 0 ? MSMD_TUNE(3) : hint;
    if (variant == 80 && !gemm8s_takes(p, nz, out_dtype == MSMD_F16X2)) variant = 0;   // a hint the call cannot follow
    if (variant == 0) {
      const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * nz;
      variant = (N > 64 && tiles128 >= 192) ? 1 : 5;
      if (N <= 64 && (long)((M + 255) / 256) * nz >= 256) variant = 14;
      // the library's own choice takes the 256 x 256 kernel only under MSMD_GEMM_W_BELOW_32: there it returns the bits of
      // gemm2s_kernel's WS form, so the row count of a launch never changes a row's result (its folding form, reachable by the
      // variant hint, sums the cross terms in another order than gemm2s_kernel's two accumulators)
      if (!(flags & 32) && (flags & 64) && gemm8s_takes(p, nz, out_dtype == MSMD_F16X2) && gemm8s_wins(M, N, K, true)) variant = 80;
    }
    const int r = out_dtype == MSMD_F32 ? dispatch_gemm2s<float>(p, nz, st, variant)
                                        : dispatch_gemm2s<f16_t>(p, nz, st, variant);
    return r >= 0 ? r : 1;
  }
  const int E = in_dtype == MSMD_F32 ? 4 : 8;
  if (K % E || lda % E || ldw % E || a_batch_stride % E || strideA % E || strideW % E || strideA2 % E || strideW2 % E)
    return 1;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return 1;
  if (rows_per_batch <= 0) rows_per_batch = M;
  GemmArgs p;
  p.A = A; p.W = W; p.bias = bias; p.R = residual; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.rows_per_batch = rows_per_batch; p.a_batch_stride = a_batch_stride;
  p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
  p.inv_rpb = 1.0f / (float)rows_per_batch;
  p.strideA = strideA; p.strideW = strideW; p.strideC = strideC; p.strideBias = strideBias; p.strideR = strideR;
  p.batch_inner = batch_inner; p.strideA2 = strideA2; p.strideW2 = strideW2; p.strideC2 = strideC2;
  p.Z = z_out; p.p_drop = p_drop; p.rng = rng; p.site = site; p.xn = 1; p.flags = flags;
  if (p_drop != 0.f && (!(p_drop > 0.f && p_drop < 1.f) || !rng || (N & 3) || ldc != N || batch != 1 || batch_inner != 1))
    return 1;  // the mask index assumes one contiguous (M, N) output
  const int osz = out_dtype == MSMD_F32 ? 4 : 2;
  p.vec_ok = (ldc % 4 == 0) && (strideC % 4 == 0) && (strideC2 % 4 == 0) && (((uintptr_t)C % (4 * osz)) == 0) &&
             (!residual || ((ldr % 4 == 0) && (strideR % 4 == 0) && (((uintptr_t)residual % (4 * osz)) == 0)));
  if ((p.flags & 2) && (osz != 2 || (ldc % 8) || (strideC % 8) || (strideC2 % 8) || ((uintptr_t)C % 16) || !p.vec_ok))
    p.flags &= ~2;   // paired stores need 16-byte aligned row pairs
  hipStream_t st = (hipStream_t)stream;
  const int nz = batch * batch_inner;
  if (in_dtype == MSMD_BF16 && (K % 64) == 0 && MSMD_TUNE(0) >= 0) {
    // Measured on MI355X (tools/bench_gemm.py): the 128x128 LDS-DMA kernel wins once the grid fills the
    // chip at 2 workgroups per CU; below that, 64x64 tiles (deep ring for long K) keep more CUs busy.
    int variant = MSMD_TUNE(0) ? MSMD_TUNE(0) : hint;
    if (variant == 80 && !gemm8_takes(p, nz, osz)) variant = 0;     // a hint the call cannot follow: the library's own choice
    if (variant == 0) {
      const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * nz;
      const long tiles192 = (long)((M + 191) / 192) * ((N + 127) / 128) * nz;
      if (!(flags & 32) && gemm8_takes(p, nz, osz) && gemm8_wins(M, N, K)) {
        variant = 80;      // 256 x 256 tiles, 8-phase schedule (flags bit 5 = caller opts out: A/B)
      } else if (N > 64 && M >= 16000 && tiles192 >= 400) {
        // tall grids: 192 x 128 tiles (76.8 FLOP per staged byte instead of 64, still two workgroups per CU).  Measured against
        // the 128 x 128 tile: conv1 454 -> 379 us, 21312 x 512 x 2048 58.7 -> 49.6, 21312 x 2048 x 512 76.6 -> 64.7; worse
        // below ~16 k rows (12800 x 512 x 1024: 22 -> 28 us) and mixed at M = 6400
        variant = 15;
      } else if (N > 64 && tall_rounds_favour_192(M, tiles128, tiles192) && !z_out && !(p_drop > 0.f) && !(flags & 8) && out_dtype == MSMD_BF16) {
        variant = 15;      // inference epilogues only: the 192-row tile's everything-epilogue runs one workgroup per CU
      } else if (N > 64 && tiles128 >= 192) {
        variant = MSMD_TUNE(4) ? 13 : 17;  // 128x128, 8 waves (4x2), 2-stage ring, 2 workgroups/CU, fragment reads pipelined
        // experiment knobs (tools/ab_graph.py): 5 = variant for M >= 20000 (conv stack), 6 = variant for the rest
        if (M >= 20000 && MSMD_TUNE(5) > 0) variant = MSMD_TUNE(5);
        if (M < 20000 && MSMD_TUNE(6) > 0) variant = MSMD_TUNE(6);
      }
      else if (N <= 64 && (long)((M + 255) / 256) * nz >= 256) variant = 14;   // the positional conv: 256 x 64 tiles, 140 -> 85 us
      else variant = (K >= 1024) ? 9 : 12;
    }
    const int r = out_dtype == MSMD_BF16 ? dispatch_gemm2<bf16_t>(p, nz, st, variant)
                                         : dispatch_gemm2<float>(p, nz, st, variant);
    if (r >= 0) return r;
  }
  if (in_dtype == MSMD_F16 && (out_dtype == MSMD_F16 || out_dtype == MSMD_F32) && (K % 64) == 0 && MSMD_TUNE(0) >= 0) {
    // fp16 storage: same LDS-DMA kernels with v_mfma_f32_16x16x32_f16 (the heuristic's variants only)
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * nz;
    const bool no_hint = !hint || (hint == 80 && !gemm8_takes(p, nz, osz));
    int variant = !no_hint ? hint : ((N > 64 && tiles128 >= 192) ? 17 : ((K >= 1024) ? 9 : 12));
    if (no_hint && N <= 64 && (long)((M + 255) / 256) * nz >= 256) variant = 14;
    if (no_hint && N > 64 && M >= 16000 && (long)((M + 191) / 192) * ((N + 127) / 128) * nz >= 400) variant = 15;
    if (no_hint && N > 64 && out_dtype == MSMD_F16 && !z_out && !(p_drop > 0.f) && !(flags & 8) &&
        tall_rounds_favour_192(M, tiles128, (long)((M + 191) / 192) * ((N + 127) / 128) * nz))
      variant = 15;
    if (no_hint && !(flags & 32) && gemm8_takes(p, nz, osz) && gemm8_wins(M, N, K)) variant = 80;
    const int r = out_dtype == MSMD_F16 ? dispatch_gemm2_f16<f16_t>(p, nz, st, variant)
                                        : dispatch_gemm2_f16<float>(p, nz, st, variant);
    if (r >= 0) return r;
  }
  if (in_dtype == MSMD_F16 && out_dtype == MSMD_F16) return launch_gemm<f16_t, f16_t>(p, nz, st);
  if (in_dtype == MSMD_F16 && out_dtype == MSMD_F32) return launch_gemm<f16_t, float>(p, nz, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_F16) return launch_gemm<float, f16_t>(p, nz, st);
  if (in_dtype == MSMD_BF16 && out_dtype == MSMD_BF16) return launch_gemm<bf16_t, bf16_t>(p, nz, st);
  if (in_dtype == MSMD_BF16 && out_dtype == MSMD_F32) return launch_gemm<bf16_t, float>(p, nz, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_F32) return launch_gemm<float, float>(p, nz, st);
  if (in_dtype == MSMD_F32 && out_dtype == MSMD_BF16) return launch_gemm<float, bf16_t>(p, nz, st);
  return 1;
}

extern "C" int msmd_gemm_256_tile_rule(int M, int N, int K) {
  return (M > 0 && N > 0 && K >= 128 && (N % 256) == 0 && (K % 64) == 0 && gemm8_wins(M, N, K)) ? 1 : 0;
}

extern "C" int msmd_gemm_256_tile_rule_f16x2(int M, int N, int K, int w_below_32) {
  return (M > 0 && N > 0 && K >= 64 && (N % 256) == 0 && (K % 32) == 0 && w_below_32 && gemm8s_wins(M, N, K, true)) ? 1 : 0;
}

extern "C" int msmd_gemm(const void* A, const void* W, const float* bias, const void* residual, void* C, int M,
                         int N, int K, int in_dtype, int out_dtype, long lda, int rows_per_batch,
                         long a_batch_stride, long ldw, long ldc, long ldr, int act, int batch, long strideA,
                         long strideW, long strideC, long strideBias, long strideR, msmd_stream_t stream) {
  return gemm_impl(A, W, bias, residual, C, M, N, K, in_dtype, out_dtype, lda, rows_per_batch, a_batch_stride, ldw, ldc,
                   ldr, act, batch, strideA, strideW, strideC, strideBias, strideR, 1, 0, 0, 0, stream);
}

// C = act(LN_A(A) . W^T + bias) + LN_R(residual), with the LayerNorms folded into this GEMM's epilogue and (optionally)
// the row statistics of C written for the next consumer: see GemmArgs and include/msmd_hip.h.  Plain row-major operands,
// no batch, 16-bit operands and output.  The statistics' slab width names the kernel that writes them: 64 = the
// 128 x 128 tile, 32 = the 64 x 64 tile (grids that would not fill the chip with 128 x 128 tiles).
extern "C" int msmd_gemm_ln(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N,
                            int K, int in_dtype, int out_dtype, long lda, long ldw, long ldc, long ldr, int act,
                            const float* a_stats, const float* w_colsum, const float* r_stats, const float* r_gamma,
                            const float* r_beta, float* stats_out, int slab_in, int slab_out, float eps,
                            msmd_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C || (K % 64) || (N % 64)) return 1;
  if ((in_dtype != MSMD_BF16 && in_dtype != MSMD_F16) || out_dtype != in_dtype) return 1;   // 16-bit rows in and out
  if ((lda % 8) || (ldw % 8) || (ldc % 4) || (residual && (ldr % 4))) return 1;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15) || ((uintptr_t)residual & 7) || ((uintptr_t)bias & 15)) return 1;
  if ((a_stats != nullptr) != (w_colsum != nullptr)) return 1;
  if (!bias || (a_stats && (residual || stats_out)) || (!a_stats && !residual)) return 1;   // the two epilogue modes
  if (r_stats && (!residual || !r_gamma || !r_beta || a_stats)) return 1;   // one side per call
  if (((uintptr_t)w_colsum & 15) || ((uintptr_t)r_gamma & 15) || ((uintptr_t)r_beta & 15) || ((uintptr_t)a_stats & 7) ||
      ((uintptr_t)r_stats & 7) || ((uintptr_t)stats_out & 7))
    return 1;
  if ((a_stats || r_stats) && slab_in != 32 && slab_in != 64) return 1;
  if (stats_out && slab_out != 32 && slab_out != 64) return 1;
  if ((a_stats && (K % slab_in)) || (r_stats && (N % slab_in))) return 1;
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const bool big = stats_out ? slab_out == 64 : (tiles128 >= 192 && (N % 128) == 0);
  if (big && (N % 128)) return 1;
  GemmArgs p;
  p.A = A; p.W = W; p.bias = bias; p.R = residual; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.rows_per_batch = M; p.a_batch_stride = 0; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act & 0xff;
  p.inv_rpb = 1.0f / (float)M; p.vec_ok = 1;
  p.strideA = p.strideW = p.strideC = p.strideBias = p.strideR = 0;
  p.batch_inner = 1; p.strideA2 = p.strideW2 = p.strideC2 = 0;
  p.Z = nullptr; p.p_drop = 0.f; p.rng = nullptr; p.site = 0; p.xn = 1;
  p.flags = ((ldc % 8) == 0 ? 2 : 0) | (((act >> 19) & 1) << 4);   // paired 16-byte stores; MSMD_GEMM_ONE_TILE_PER_WORKGROUP
  p.a_stats = a_stats; p.a_nt = a_stats ? K / slab_in : 0; p.w_colsum = w_colsum;
  p.r_stats = r_stats; p.r_nt = r_stats ? N / slab_in : 0; p.r_gamma = r_gamma; p.r_beta = r_beta;
  p.stats_out = stats_out; p.ln_eps = eps;
  hipStream_t st = (hipStream_t)stream;
  int variant = big ? 17 : (K >= 1024 ? 9 : 12);
  if (big && M >= 16000 && (long)((M + 191) / 192) * (N / 128) >= 400) variant = 15;    // tall grids: the 192 x 128 tile (same 64-column slabs)
  if (big && tall_rounds_favour_192(M, tiles128, (long)((M + 191) / 192) * (N / 128))) variant = 15;
  {   // caller's tile hint for the big-tile family (same 64-column statistics slabs): 15 = 192 x 128, 17 = 128 x 128, 66 = A/B form of 17
    const int hint = (act >> 8) & 0xff;
    if (big && (hint == 15 || hint == 17 || hint == 66)) variant = hint;
    // 80 = the 256 x 256 kernel: writes the same 64-column statistics slabs, reads either width
    const bool can8 = gemm8_takes(p, 1, 2) && (!stats_out || slab_out == 64);
    if (can8 && (hint == 80 || (hint == 0 && !(act & MSMD_GEMM_NO_256_TILE) && gemm8_wins(M, N, K)))) variant = 80;
  }
  const int r = in_dtype == MSMD_BF16 ? dispatch_gemm2<bf16_t>(p, 1, st, variant) : dispatch_gemm2_f16<f16_t>(p, 1, st, variant);
  return r >= 0 ? r : 1;
}

extern "C" int msmd_gemm_ex(const void* A, const void* W, const float* bias, const void* residual, void* C, int M,
                            int N, int K, int in_dtype, int out_dtype, long lda, int rows_per_batch,
                            long a_batch_stride, long ldw, long ldc, long ldr, int act, int batch, long strideA,
                            long strideW, long strideC, long strideBias, long strideR, void* z_out, float p_drop,
                            const unsigned long* rng_state, unsigned int site, msmd_stream_t stream) {
  return gemm_impl(A, W, bias, residual, C, M, N, K, in_dtype, out_dtype, lda, rows_per_batch, a_batch_stride, ldw, ldc,
                   ldr, act, batch, strideA, strideW, strideC, strideBias, strideR, 1, 0, 0, 0, stream, z_out, p_drop,
                   rng_state, site);
}

// dZ = keep_mask / (1 - p) * act'(Z) * (A . W^T): the data gradient of a Linear whose INPUT was dropout(act(Z)) -- the product
// and the backward of the activation + dropout in one launch (A = the upstream gradient, W = the transposed weight cast,
// Z = the forward's pre-activation, C and Z contiguous (M, N)).  Replaces msmd_gemm + msmd_act_bwd_dropout / msmd_act_bwd.
extern "C" int msmd_gemm_actbwd(const void* A, const void* W, const void* Z, void* C, int M, int N, int K, int in_dtype,
                                int out_dtype, long lda, long ldw, int act, float p_drop, const unsigned long* rng_state,
                                unsigned int site, msmd_stream_t stream) {
  if (!Z || in_dtype != out_dtype || (in_dtype != MSMD_BF16 && in_dtype != MSMD_F16) || (N & 3) || ((uintptr_t)Z & 7)) return 1;
  if (K % 64) return 1;      // the LDS-DMA kernels only (the others do not carry this epilogue)
  return gemm_impl(A, W, nullptr, nullptr, C, M, N, K, in_dtype, out_dtype, lda, 0, 0, ldw, N, 0, act & 0x3ffff, 1, 0, 0, 0, 0,
                   0, 1, 0, 0, 0, stream, const_cast<void*>(Z), p_drop, rng_state, site, 8);
}

extern "C" int msmd_gemm_batched2(const void* A, const void* W, void* C, int M, int N, int K, int in_dtype,
                                  int out_dtype, long lda, long ldw, long ldc, int batch_outer, long strideA_o,
                                  long strideW_o, long strideC_o, int batch_inner, long strideA_i, long strideW_i,
                                  long strideC_i, msmd_stream_t stream) {
  return gemm_impl(A, W, nullptr, nullptr, C, M, N, K, in_dtype, out_dtype, lda, 0, 0, ldw, ldc, 0, MSMD_ACT_NONE,
                   batch_outer, strideA_o, strideW_o, strideC_o, 0, 0, batch_inner, strideA_i, strideW_i, strideC_i,
                   stream);
}
