// Decoder self-attention block in ONE launch per sequence (reference model.py:874-878: nn.TransformerDecoderLayer's
// `x = norm1(x + self_attn(x))`, the block the sampler runs 8 x 3 x 64 x 500 times per call, model.py:368-440):
//
//   x1[n] = LN1( resid[n] + softmax(scale Q_h K_h^T) V_h  (all 8 heads)  . Wo^T + bo ),     resid = x or LN3(u_prev)
//
// from the fused Q | K | V rows the QKV GEMM stored.  One workgroup = one sequence (T <= 112 tokens, 7 waves x 16 queries),
// heads one after the other:
//   * K_h, V_h (T x 64 each) are staged into LDS (the row-major images and swizzles of attn_whole_kernel: conflict-free
//     ds_read_b128 for K, ds_read_b64_tr_b16 for V^T); S^T = K Q^T, the two-pass softmax and O^T = V^T P^T run as there,
//     the query on the MFMA lane;
//   * O_h never leaves the registers: its accumulator layout (head dims 16 d + 4 fq + e on the rows, query on the lane) IS a
//     B operand of v_mfma_f32_16x16x32 up to a permutation of the 32 K slots, and the out-projection's A operand -- the head's
//     512 x 64 slice of Wo, streamed into a double-buffered LDS image by LDS-DMA one head ahead -- is read in the same
//     permuted order (two 8-byte reads per fragment): acc^T[n][query] += Wo[n, 64 h + k] O_h[query][k], 64 MFMAs per wave
//     and head into 128 accumulator registers;
//   * every wave ends up with COMPLETE output rows of its 16 queries (4 lanes x 128 registers per row), so the bias, the
//     residual -- LayerNorm'ed on the fly from the un-normalised rows the previous layer's FFN stored when (g3, b3) are
//     given: its statistics are two shuffles away -- and norm1 itself are applied in registers and the normalised rows are
//     stored.  No attention output, no out-projection launch, no row statistics in HBM.
// Against the two launches it replaces in the sampler's step (attn_whole_kernel 29 us + the out-projection GEMM with its
// LayerNorm-residual epilogue 28-31 us at 192 sequences) it moves the same Q | K | V and residual bytes once and nothing else.
// d = 512, H = 8 (the decoder's geometry), 16-bit storage.
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct SabArgs {
  const void* qkv; long qkv_b, qkv_t;            // rows [Q | K | V] of 3 d elements: batch / token strides in elements
  const void* Wo; const float* bo;               // (d, d) row-major (out feature, in feature), bias
  const void* resid; const float* g3; const float* b3;   // residual rows (N, T, d); LayerNorm'ed with (g3, b3) when g3 != NULL
  const float* g1; const float* b1;              // norm1
  void* out;                                     // (N, T, d)
  int N, T;
  float scale, eps;
};

// raw workgroup barrier fenced for the compiler on both sides (no vmcnt / lgkmcnt wait of its own: the callers wait by hand)
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// LN3: the residual rows are un-normalised and LayerNorm'ed here with (g3, b3)
template <typename T, bool LN3>
__global__ __launch_bounds__(448) void self_attn_block_kernel(const SabArgs p) {
  constexpr int D = 512, H = 8, NF = 7, ROWS = 16 * NF, NT = 448;
  constexpr int KV_BYTES = ROWS * 128;                 // one K or V image: 112 rows x 128 B
  constexpr int W_BYTES = D * 128;                     // one head's slice of Wo: 512 rows x 64 elements
  typedef typename Vec8T<T>::type V8;
  typedef typename Vec4T<T>::type V4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + KV_BYTES;
  unsigned char* sW = smem + 2 * KV_BYTES;             // two slices (head parity)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int n = blockIdx.x;
  const int query = wid * 16 + fr;
  const int qrow = query < p.T ? query : p.T - 1;
  const T* base = (const T*)p.qkv + (long)n * p.qkv_b;
  const T* Qrow = base + (long)qrow * p.qkv_t;
  const int nf = (p.T + 15) >> 4;

  // ---- staging helpers
  constexpr int NPF = (ROWS * 8 + NT - 1) / NT;        // 16-byte chunks of one image per thread (2)
  u32x4 pk[NPF], pv[NPF];
  auto load_kv = [&](int h) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int c = tid + i * NT, row = c >> 3, ch = c & 7;
      pk[i] = pv[i] = u32x4{0, 0, 0, 0};
      if (row < p.T) {
        const T* r = base + (long)row * p.qkv_t + h * 64 + ch * 8;
        pk[i] = *(const u32x4*)(r + D);
        pv[i] = *(const u32x4*)(r + 2 * D);
      }
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int c = tid + i * NT, row = c >> 3, ch = c & 7;
      if (row < ROWS) {
        *(u32x4*)(sK + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = pk[i];
        *(u32x4*)(sV + row * 128 + ((((ch >> 1) ^ ((row >> 1) & 3)) << 5) + ((ch & 1) << 4))) = pv[i];
      }
    }
  };
  // Wo[:, 64 h .. 64 h + 63] -> sW[h & 1] by LDS-DMA: piece k = rows 8 k .. 8 k + 7 (1 KiB, linear in LDS); the 16-byte chunk
  // index is XOR-swizzled with (row >> 1) & 7 on the SOURCE side (a DMA wave-instruction writes linearly).  Inline asm: through
  // the builtin the compiler would put s_waitcnt vmcnt(0) in front of every later LDS read (attention.hip, prefetch note).
  typedef __attribute__((address_space(3))) void lds_t;
  const unsigned sw_lds = (unsigned)(uintptr_t)(lds_t*)sW;
  auto issue_w = [&](int h) {
    const T* W = (const T*)p.Wo + h * 64;
    const unsigned dst0 = sw_lds + (h & 1) * W_BYTES;
    for (int k = wid; k < 64; k += 7) {
      const int row = 8 * k + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      const T* src = W + (long)row * D + c * 8;
      const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + k * 1024);
      unsigned m0_keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_keep) : "v"(src), "s"(dst) : "memory");
    }
  };

  f32x4 acc[32];                                       // out^T[n = 16 nb + 4 fq + e][query = fr]
#pragma unroll
  for (int nb = 0; nb < 32; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_w(0);
  load_kv(0);
  u32x4 qf[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) qf[g] = *(const u32x4*)(Qrow + 32 * g + 8 * fq);
  store_kv();
  const float c2 = p.scale * 1.4426950408889634f;
  const int qp = fr >> 2, pp = fr & 3;

  for (int h = 0; h < H; ++h) {
    // K_h, V_h written (by every thread, above / at the end of the last pass) and W_h landed: visible to all
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wg_barrier();
    u32x4 qn[2] = {qf[0], qf[1]};
    if (h + 1 < H) {                                   // one head ahead: the weight slice (other buffer: last read a head ago,
      issue_w(h + 1);                                  // every wave is past that), then K / V / Q of the next head into registers
      load_kv(h + 1);
#pragma unroll
      for (int g = 0; g < 2; ++g) qn[g] = *(const u32x4*)(Qrow + (h + 1) * 64 + 32 * g + 8 * fq);
    }
    // ---- S^T = K Q^T, softmax over the whole row, O^T = V^T P^T   (attn_whole_kernel's layouts)
    f32x4 s[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      s[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (f >= nf) continue;
      const int row = 16 * f + fr;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const u32x4 a = *(const u32x4*)(sK + row * 128 + (((4 * g + fq) ^ ((row >> 1) & 7)) << 4));
        s[f] = mfma16<T>(a, qf[g], s[f]);
      }
      if (f & 1) __builtin_amdgcn_sched_barrier(0);      // two key fragments of reads in flight (the 128 accumulators stay put)
    }
    if (p.T & 15) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f != nf - 1) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (16 * f + 4 * fq + e >= p.T) s[f][e] = -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f >= nf) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[f][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c2;
    float ps = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f >= nf) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = __builtin_amdgcn_exp2f(fmaf(s[f][e], c2, -mc));
        s[f][e] = pe;
        ps += pe;
      }
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    f32x4 acc_o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) acc_o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < (NF + 1) / 2; ++pr) {
      const int f0 = 2 * pr, f1 = 2 * pr + 1;
      if (f0 >= nf) continue;
      const bool has1 = f1 < NF && f1 < nf;
      const f32x4 s1 = has1 ? s[f1 < NF ? f1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
      const V8 pbv = V8{(T)s[f0][0], (T)s[f0][1], (T)s[f0][2], (T)s[f0][3], (T)s1[0], (T)s1[1], (T)s1[2], (T)s1[3]};
      const u32x4 pb = __builtin_bit_cast(u32x4, pbv);
      const int r0 = 16 * f0 + 4 * fq + qp, r1 = has1 ? r0 + 16 : r0;   // no second fragment: any staged row, p = 0
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sV + r0 * 128 + ((d ^ ((r0 >> 1) & 3)) << 5) + pp * 8));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sV + r1 * 128 + ((d ^ ((r1 >> 1) & 3)) << 5) + pp * 8));
        const s16x8 va = s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        acc_o[d] = mfma16<T>(__builtin_bit_cast(u32x4, va), pb, acc_o[d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- every wave is done with the K / V images (its reads fed MFMAs that were issued): the next head's go in (their loads
    // were issued a whole attention ago)
    wg_barrier();
    if (h + 1 < H) store_kv();
    qf[0] = qn[0]; qf[1] = qn[1];
    // ---- out-projection of this head: B operand = O_h straight from the accumulators (rounded to storage as a stored
    // attention output would be), K slots in accumulator order: slot (fq, e) = 32 g + 4 fq + e, slot (fq, 4 + e) = 32 g + 16 + 4 fq + e
    const float inv = 1.0f / ps;
    u32x4 ob[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const V8 o8 = V8{(T)(acc_o[2 * g][0] * inv), (T)(acc_o[2 * g][1] * inv), (T)(acc_o[2 * g][2] * inv), (T)(acc_o[2 * g][3] * inv),
                       (T)(acc_o[2 * g + 1][0] * inv), (T)(acc_o[2 * g + 1][1] * inv), (T)(acc_o[2 * g + 1][2] * inv),
                       (T)(acc_o[2 * g + 1][3] * inv)};
      ob[g] = __builtin_bit_cast(u32x4, o8);
    }
    const unsigned char* sw = sW + (h & 1) * W_BYTES;
#pragma unroll
    for (int nb = 0; nb < 32; ++nb) {
      const int row = 16 * nb + fr, x = (row >> 1) & 7;
      const unsigned char* rp = sw + row * 128 + (fq & 1) * 8;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const u32x2 lo = *(const u32x2*)(rp + (((4 * g + (fq >> 1)) ^ x) << 4));
        const u32x2 hi = *(const u32x2*)(rp + (((4 * g + 2 + (fq >> 1)) ^ x) << 4));
        acc[nb] = mfma16<T>(u32x4{lo[0], lo[1], hi[0], hi[1]}, ob[g], acc[nb]);
      }
      if ((nb & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four fragment rows of reads in flight, not all 32 (registers)
    }
  }

  // ---- epilogue: complete rows.  Lane (fr, fq) holds out[query = fr][16 nb + 4 fq + e]; a row = the 4 lanes sharing fr.
  const T* rrow = (const T*)p.resid + ((long)n * p.T + qrow) * D + 4 * fq;
  float mu3 = 0.f, rs3 = 1.f;
  if constexpr (LN3) {   // statistics of the un-normalised residual row: a pass of its own (the row is re-read below, from cache:
    float S = 0.f, Q = 0.f;  // keeping it would cost 64 registers beside the 128 accumulators)
#pragma unroll 4
    for (int nb = 0; nb < 32; ++nb) {
      const V4 r4 = *(const V4*)(rrow + 16 * nb);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = (float)r4[e]; S += v; Q = fmaf(v, v, Q); }
    }
    S += __shfl_xor(S, 16, 64); S += __shfl_xor(S, 32, 64);
    Q += __shfl_xor(Q, 16, 64); Q += __shfl_xor(Q, 32, 64);
    mu3 = S * (1.0f / D);
    rs3 = rsqrtf(fmaxf(Q * (1.0f / D) - mu3 * mu3, 0.f) + p.eps);
  }
  asm volatile("" : "+v"(rrow));     // (the re-read below must not be merged with the statistics pass's loads: 64 registers)
  float S1 = 0.f;
  V4 uh[32];                                           // the block's un-normalised output rows, rounded to storage (as a stored
#pragma unroll                                         // GEMM output would be): 64 registers instead of the 128 accumulators
  for (int nb = 0; nb < 32; ++nb) {
    const int c0 = 16 * nb + 4 * fq;
    const f32x4 bv = *(const f32x4*)(p.bo + c0);
    f32x4 gv = f32x4{1.f, 1.f, 1.f, 1.f}, be = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LN3) { gv = *(const f32x4*)(p.g3 + c0); be = *(const f32x4*)(p.b3 + c0); }
    const V4 r4 = *(const V4*)(rrow + 16 * nb);
    float u[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // the residual as the consumer of a stored LayerNorm output would have read it: rounded to storage
      const float r = LN3 ? (float)(T)fmaf(((float)r4[e] - mu3) * rs3, gv[e], be[e]) : (float)r4[e];
      u[e] = acc[nb][e] + bv[e] + r;
    }
    uh[nb] = pack4<T>(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) S1 += (float)uh[nb][e];
    if ((nb & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // the bias / gamma / beta loads of two column blocks at a time
  }
  S1 += __shfl_xor(S1, 16, 64); S1 += __shfl_xor(S1, 32, 64);
  const float mu1 = S1 * (1.0f / D);
  float Q1 = 0.f;
#pragma unroll
  for (int nb = 0; nb < 32; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float dd = (float)uh[nb][e] - mu1; Q1 = fmaf(dd, dd, Q1); }
  Q1 += __shfl_xor(Q1, 16, 64); Q1 += __shfl_xor(Q1, 32, 64);
  const float rs1 = 1.0f / sqrtf(Q1 * (1.0f / D) + p.eps);
  {   // (a lane past the last token computed token T - 1 again, from the same operands: it stores the same bytes)
    T* orow = (T*)p.out + ((long)n * p.T + qrow) * D + 4 * fq;
#pragma unroll
    for (int nb = 0; nb < 32; ++nb) {
      const int c0 = 16 * nb + 4 * fq;
      const f32x4 gv = *(const f32x4*)(p.g1 + c0), be = *(const f32x4*)(p.b1 + c0);
      *(V4*)(orow + 16 * nb) = pack4<T>(((float)uh[nb][0] - mu1) * rs1 * gv[0] + be[0], ((float)uh[nb][1] - mu1) * rs1 * gv[1] + be[1],
                                        ((float)uh[nb][2] - mu1) * rs1 * gv[2] + be[2], ((float)uh[nb][3] - mu1) * rs1 * gv[3] + be[3]);
      if ((nb & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

}  // namespace

// x1 (N, T, 512) = LN_{g1,b1}( resid' + MHA(qkv) Wo^T + bo ), resid' = resid or LN_{g3,b3}(resid): see the kernel.  qkv rows are
// [Q | K | V] (3 x 512 elements) at qkv + n qkv_bstride + t qkv_tstride (elements); T <= 112; d = 512, H = 8; dtype MSMD_F16 |
// MSMD_BF16 for qkv, Wo, resid, out; bo / g* / b* fp32.  out must not alias resid.
extern "C" int msmd_self_attn_block(const void* qkv, long qkv_bstride, long qkv_tstride, const void* Wo, const float* bo,
                                    const void* resid, const float* g3, const float* b3, const float* g1, const float* b1,
                                    void* out, int N, int T, int d, int H, float scale, float eps, int dtype,
                                    msmd_stream_t stream) {
  if (N <= 0 || T <= 0 || T > 112 || d != 512 || H != 8 || !qkv || !Wo || !bo || !resid || !g1 || !b1 || !out || out == resid ||
      (g3 != nullptr) != (b3 != nullptr) || (qkv_bstride & 7) || (qkv_tstride & 7) || qkv_tstride < 3 * d)
    return 1;
  if (((uintptr_t)qkv | (uintptr_t)Wo | (uintptr_t)bo | (uintptr_t)resid | (uintptr_t)g3 | (uintptr_t)b3 | (uintptr_t)g1 |
       (uintptr_t)b1 | (uintptr_t)out) & 15)
    return 1;
  SabArgs p{qkv, qkv_bstride, qkv_tstride, Wo, bo, resid, g3, b3, g1, b1, out, N, T, scale, eps};
  constexpr int lds = 2 * 112 * 128 + 2 * 512 * 128;     // K, V images + two weight slices = 156 KB: one workgroup per CU
  hipStream_t st = (hipStream_t)stream;
#define MSMD_SAB(TT, L3)                                                                                              \
  do {                                                                                                                 \
    auto kfn = self_attn_block_kernel<TT, L3>;                                                                         \
    static bool attr = false;                                                                                          \
    if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
    hipLaunchKernelGGL(kfn, dim3(N), dim3(448), lds, st, p);                                                           \
  } while (0)
  if (dtype == MSMD_F16) { if (g3) MSMD_SAB(f16_t, true); else MSMD_SAB(f16_t, false); }
  else if (dtype == MSMD_BF16) { if (g3) MSMD_SAB(bf16_t, true); else MSMD_SAB(bf16_t, false); }
  else return 1;
#undef MSMD_SAB
  MSMD_RETURN_LAST();
}
