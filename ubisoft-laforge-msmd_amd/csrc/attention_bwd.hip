// Fused attention backward for short sequences (Tk <= 256, head_dim 64, bf16): one workgroup per (batch, head)
// recomputes P = softmax(scale Q K^T, masked) tile by tile and produces dQ, dK, dV without ever writing P, dP or
// any transposed operand to HBM (the unfused path: 6 transposes + 4 batched GEMMs + 2 softmax passes per call).
//
//   per 64-row q tile (wave w owns q rows 16w..16w+15, all keys):
//     S^T  = K Q^T            (row reads of the K / Q images)          -> softmax in registers (lane = one q)
//     dP^T = V dO^T           (row reads of V / dO)
//     dS^T = scale P^T o (dP^T - rowsum(P o dP))
//     dQ^T = K^T dS^T         (K^T via ds_read_b64_tr_b16, dS^T straight from the accumulator registers)
//     P, dS -> LDS [q][key] images (bf16)
//   then (wave w owns key fragments w, w+4, ...):
//     dV^T += dO^T P,  dK^T += Q^T dS   (both operands via transposing reads; fp32 accumulators live in registers
//                                        across all q tiles)
// K, V stay resident in LDS for the whole workgroup; Q / dO are staged per tile.  All [rows][64] images share one
// XOR swizzle of 32-byte blocks that serves row reads and transposed reads.
//
// Reference semantics: autograd of HF Wav2Vec2Attention (utils/wav2vec2.py:111 -> transformers) and of
// nn.MultiheadAttention inside nn.TransformerDecoderLayer / nn.TransformerEncoderLayer (model.py:874-878,
// style_encoder.py:158) under loss.backward() (training_script.py:196); eval-mode (no attention dropout).
#include "common.h"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct AbArgs {
  const bf16_t *Q, *K, *V, *dO;
  bf16_t *dQ, *dK, *dV;
  const uint8_t* mask;
  int B, H, Tq, Tk;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, do_bs, do_ts, dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
  float scale;
  float p_drop;
  const unsigned long* rng_state;
  unsigned site;
};

// byte offset of 16-byte chunk ch (0..7) of row `row` in a [rows][64] bf16 image
__device__ __forceinline__ int img_off(int row, int ch) {
  return row * 128 + (((((ch >> 1) ^ ((row >> 1) & 3)) << 1) | (ch & 1)) << 4);
}
// address a lane supplies for a transposed read of rows r0..r0+3 (lane's row = r0 + q'), 16-column block db
__device__ __forceinline__ int img_tr(int row, int db, int pp) {
  return row * 128 + ((db ^ ((row >> 1) & 3)) << 5) + pp * 8;
}
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  return __builtin_bit_cast(u32x2, v);
}
__device__ __forceinline__ bf16x8 frag8(const u32x2 lo, const u32x2 hi) {
  return __builtin_bit_cast(bf16x8, (u32x4{lo[0], lo[1], hi[0], hi[1]}));
}

// global (rows x 64 bf16, row stride ts) -> swizzled LDS image of ROWS rows, rows >= valid zero-filled.  Two halves so
// that a caller can put ALL its loads in flight before the first LDS write (one memory latency per staging step instead
// of one per 16-byte chunk: the rolled loop this replaces waited for every load before its store).
template <int ROWS>
struct RowStage {
  static constexpr int N = ROWS * 8 / 256;
  static_assert(ROWS * 8 % 256 == 0, "whole passes of the 256 threads");
  u32x4 v[N];
  __device__ __forceinline__ void load(const bf16_t* src, long ts, int valid, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
      v[i] = *(const u32x4*)(src + (long)min(row, valid - 1) * ts + ch * 8);   // valid >= 1: always an address of the tensor
    }
  }
  __device__ __forceinline__ void store(unsigned char* img, int valid, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
      *(u32x4*)(img + img_off(row, ch)) = row < valid ? v[i] : u32x4{0u, 0u, 0u, 0u};
    }
  }
};

template <int NKF>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AbArgs p) {
  constexpr int NKP = (NKF + 1) / 2;             // key-fragment pairs (32-key MFMA k-steps)
  constexpr int KROWS = NKP * 32;                // K / V image rows (zero padded)
  constexpr int RS = NKF * 32 + ((NKF & 1) ? 0 : 32);  // P / dS image row stride in bytes (odd multiple of 32 B)
  constexpr int MAXF = (NKF + 3) / 4;            // key fragments a wave owns in the dV / dK phase
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = sK + KROWS * 128;
  unsigned char* sQ = sV + KROWS * 128;
  unsigned char* sdO = sQ + 64 * 128;
  unsigned char* sP = sdO + 64 * 128;
  unsigned char* sdS = sP + 64 * RS;

  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, qp = fr >> 2, pp = fr & 3;
  const bf16_t* Qg = p.Q + b * p.q_bs + h * 64;
  const bf16_t* Kg = p.K + b * p.k_bs + h * 64;
  const bf16_t* Vg = p.V + b * p.v_bs + h * 64;
  const bf16_t* dOg = p.dO + b * p.do_bs + h * 64;

  {
    RowStage<KROWS> rk, rv;
    rk.load(Kg, p.k_ts, p.Tk, tid);
    rv.load(Vg, p.v_ts, p.Tk, tid);
    rk.store(sK, p.Tk, tid);
    rv.store(sV, p.Tk, tid);
  }

  f32x4 accV[MAXF][4], accK[MAXF][4];
#pragma unroll
  for (int i = 0; i < MAXF; ++i)
#pragma unroll
    for (int d = 0; d < 4; ++d) accV[i][d] = accK[i][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nqt = (p.Tq + 63) / 64;
  constexpr bool AHEAD = NKF < 16;   // 16 key fragments leave no registers for rows in flight across phase B
  RowStage<64> rq, rdo;   // the NEXT q tile's Q / dO rows, in flight while the current tile is computed
  if (AHEAD) {
    rq.load(Qg, p.q_ts, p.Tq, tid);
    rdo.load(dOg, p.do_ts, p.Tq, tid);
  }
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 64;
    __syncthreads();  // previous tile's readers of sQ / sdO / sP / sdS are done (and K / V staged on the first pass)
    if (!AHEAD) {
      rq.load(Qg + (long)q0 * p.q_ts, p.q_ts, p.Tq - q0, tid);
      rdo.load(dOg + (long)q0 * p.do_ts, p.do_ts, p.Tq - q0, tid);
    }
    rq.store(sQ, p.Tq - q0, tid);
    rdo.store(sdO, p.Tq - q0, tid);
    __syncthreads();

    // ---------------- phase A: this wave's 16 q rows against all keys
    // a wave whose 16 rows all lie past Tq (the ragged last q tile: T = 200 -> 8 live rows of 64) only zeroes its rows of
    // the P / dS images; phase B skips the 32-row k-steps that hold no live row
    const int live_rows = p.Tq - q0;
    if (16 * w >= live_rows) {
      if (16 * w < ((live_rows + 31) & ~31)) {   // rows of a k-step phase B still reads
#pragma unroll
        for (int f = 0; f < NKF; ++f) {
          *(u32x2*)(sP + (16 * w + fr) * RS + f * 32 + fq * 8) = u32x2{0u, 0u};
          *(u32x2*)(sdS + (16 * w + fr) * RS + f * 32 + fq * 8) = u32x2{0u, 0u};
        }
      }
    } else {
    const int qrow = 16 * w + fr;      // tile-local q of this lane (operand column)
    const int qglob = q0 + qrow;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(sQ + img_off(qrow, 4 * ks + fq)));
      dof[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(sdO + img_off(qrow, 4 * ks + fq)));
    }
    // the lane's 4 keys of a fragment are 4 consecutive mask bytes: ONE 4-byte load per fragment, all of them issued
    // before the score MFMAs (the byte loads this replaces were each waited for: 4 NKF round trips per q tile)
    unsigned m4[NKF];
    int fq4 = 4 * fq;
    asm volatile("" : "+v"(fq4));   // opaque per tile: the NKF offsets / shifts below are recomputed, not kept in registers across tiles
    if (p.mask) {
      const uint8_t* mrow4 = p.mask + (long)min(qglob, p.Tq - 1) * p.Tk;
      if (p.Tk >= 4) {
#pragma unroll
        for (int f = 0; f < NKF; ++f) __builtin_memcpy(&m4[f], mrow4 + min(16 * f + fq4, p.Tk - 4), 4);
      } else {   // every live key is one of the row's first Tk < 4 bytes
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (e < p.Tk) m |= (unsigned)mrow4[e] << (8 * e);
#pragma unroll
        for (int f = 0; f < NKF; ++f) m4[f] = m;
      }
    }
    f32x4 s[NKF], dp[NKF];
    unsigned long keep = ~0ul;  // 4 keep bits per key fragment
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
      s[f] = dp[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const u32x4*)(sK + img_off(16 * f + fr, 4 * ks + fq)));
        const bf16x8 vf = __builtin_bit_cast(bf16x8, *(const u32x4*)(sV + img_off(16 * f + fr, 4 * ks + fq)));
        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[f], 0, 0, 0);
        dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp[f], 0, 0, 0);
      }
    }
    // s[f][e] = S^T[key = 16 f + 4 fq + e][q = fr]: softmax over keys = over (f, e) in-lane and over fq across lanes
    // (selects only: the short-circuit form of these tests compiled to three branches per element)
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[f][e] = 16 * f + 4 * fq + e >= p.Tk ? -3.0e38f : s[f][e] * p.scale;
    if (p.mask) {
      const int tk4 = max(p.Tk - 4, 0);
#pragma unroll
      for (int f = 0; f < NKF; ++f) {
        // the 4 bytes were loaded from min(k0, Tk - 4): the byte of key k0 + e sits (k0 - that) bytes further up
        // (3 at most for a live key; a shift that runs off the word belongs to keys >= Tk, dead already)
        const unsigned m = m4[f] >> ((8 * max(16 * f + fq4 - tk4, 0)) & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[f][e] = ((m >> (8 * e)) & 0xffu) ? -3.0e38f : s[f][e];
      }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[f][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = s[f][e] > -1.0e38f ? __expf(s[f][e] - mx) : 0.f;
        s[f][e] = pv;
        l += pv;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = l > 0.f ? 1.0f / l : 0.f;
    float delta = 0.f;
    if (p.p_drop > 0.f) {  // same Philox mask as the forward: dP <- dP o keep / (1 - p)
      const unsigned thr = dropout_threshold16(p.p_drop);
      const float c = 1.0f / (1.0f - p.p_drop);
      const unsigned long rowbase = ((unsigned long)blockIdx.x * p.Tq + min(qglob, p.Tq - 1)) * 128ul;
#pragma unroll
      for (int pr = 0; pr < NKP; ++pr) {   // one generator block per fragment pair, as the forward kernels (common.h)
        const Philox4 r = dropout_bits(p.rng_state, p.site, rowbase + (unsigned long)(4 * pr + fq));
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int f = 2 * pr + o;
          if (f >= NKF) continue;
          unsigned long k4 = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool kp = dropout_value16(r, e, o) >= thr;
            dp[f][e] = kp ? dp[f][e] * c : 0.f;
            k4 |= (unsigned long)kp << e;
          }
          keep = (keep & ~(15ul << (4 * f))) | (k4 << (4 * f));
        }
      }
    }
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[f][e] *= inv_l;
        delta += s[f][e] * dp[f][e];
      }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float drop_c = 1.0f / (1.0f - p.p_drop);
    // P and dS (bf16) -> registers for dQ and -> the [q][key] LDS images for the dV / dK phase
    bf16x4 pb[NKF], dsb[NKF];
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pb[f][e] = (bf16_t)(((keep >> (4 * f + e)) & 1ul) ? s[f][e] * drop_c : 0.f);  // P o keep / (1 - p) feeds dV
        dsb[f][e] = (bf16_t)(s[f][e] * (dp[f][e] - delta) * p.scale);
      }
      *(bf16x4*)(sP + qrow * RS + f * 32 + fq * 8) = pb[f];
      *(bf16x4*)(sdS + qrow * RS + f * 32 + fq * 8) = dsb[f];
    }
    // dQ^T[d][q] = sum_key K[key][d] dS^T[key][q]
    f32x4 accq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) accq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
      const int f0 = 2 * kp, f1 = 2 * kp + 1;
      bf16x8 bds;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bds[e] = dsb[f0][e];
        bds[4 + e] = (f1 < NKF) ? dsb[f1 < NKF ? f1 : f0][e] : (bf16_t)0.f;
      }
      const int r0 = 16 * f0 + 4 * fq + qp, r1 = r0 + 16;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 ka = frag8(tr_read(sK + img_tr(r0, d, pp)), tr_read(sK + img_tr(r1, d, pp)));
        accq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, bds, accq[d], 0, 0, 0);
      }
    }
    if (qglob < p.Tq) {
      bf16_t* dst = p.dQ + b * p.dq_bs + (long)qglob * p.dq_ts + h * 64;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)accq[d][e];
        *(bf16x4*)(dst + 16 * d + 4 * fq) = o;
      }
    }
    }
    __syncthreads();  // P / dS images complete
    if (AHEAD && qt + 1 < nqt) {   // issued HERE: phase B holds few registers, and its MFMAs cover the loads' latency
      rq.load(Qg + (long)(q0 + 64) * p.q_ts, p.q_ts, p.Tq - q0 - 64, tid);
      rdo.load(dOg + (long)(q0 + 64) * p.do_ts, p.do_ts, p.Tq - q0 - 64, tid);
    }

    // ---------------- phase B: dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (32 * ks >= live_rows) continue;   // no live q row in this k-step (wave-uniform)
      const int r0 = 32 * ks + 4 * fq + qp, r1 = r0 + 16;  // tile-local q rows this lane addresses
      bf16x8 ado[4], aq[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        ado[d] = frag8(tr_read(sdO + img_tr(r0, d, pp)), tr_read(sdO + img_tr(r1, d, pp)));
        aq[d] = frag8(tr_read(sQ + img_tr(r0, d, pp)), tr_read(sQ + img_tr(r1, d, pp)));
      }
#pragma unroll
      for (int i = 0; i < MAXF; ++i) {
        const int f = w + 4 * i;
        if (f < NKF) {  // wave-uniform
          const bf16x8 bp = frag8(tr_read(sP + r0 * RS + f * 32 + pp * 8), tr_read(sP + r1 * RS + f * 32 + pp * 8));
          const bf16x8 bd = frag8(tr_read(sdS + r0 * RS + f * 32 + pp * 8), tr_read(sdS + r1 * RS + f * 32 + pp * 8));
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            accV[i][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado[d], bp, accV[i][d], 0, 0, 0);
            accK[i][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[d], bd, accK[i][d], 0, 0, 0);
          }
        }
      }
    }
  }

  // acc*[i][d][e] = d{V,K}[key = 16 f + fr][dim = 16 d + 4 fq + e]
#pragma unroll
  for (int i = 0; i < MAXF; ++i) {
    const int f = w + 4 * i;
    const int key = 16 * f + fr;
    if (f < NKF && key < p.Tk) {
      bf16_t* dv = p.dV + b * p.dv_bs + (long)key * p.dv_ts + h * 64;
      bf16_t* dk = p.dK + b * p.dk_bs + (long)key * p.dk_ts + h * 64;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        bf16x4 ov, ok;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ov[e] = (bf16_t)accV[i][d][e];
          ok[e] = (bf16_t)accK[i][d][e];
        }
        *(bf16x4*)(dv + 16 * d + 4 * fq) = ov;
        *(bf16x4*)(dk + 16 * d + 4 * fq) = ok;
      }
    }
  }
}

template <int NKF>
int launch_attn_bwd(const AbArgs& p, hipStream_t st) {
  constexpr int NKP = (NKF + 1) / 2, KROWS = NKP * 32, RS = NKF * 32 + ((NKF & 1) ? 0 : 32);
  constexpr int lds = 2 * KROWS * 128 + 2 * 64 * 128 + 2 * 64 * RS;
  static bool attr_done = false;
  auto kfn = attn_bwd_kernel<NKF>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(p.B * p.H), dim3(256), lds, st, p);
  MSMD_RETURN_LAST();
}

}  // namespace

extern "C" int msmd_attention_bwd(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK,
                                  void* dV, int B, int H, int Tq, int Tk, long q_bstride, long q_tstride,
                                  long k_bstride, long k_tstride, long v_bstride, long v_tstride, long do_bstride,
                                  long do_tstride, long dq_bstride, long dq_tstride, long dk_bstride, long dk_tstride,
                                  long dv_bstride, long dv_tstride, float scale, const uint8_t* mask, float p_drop,
                                  const unsigned long* rng_state, unsigned int site, msmd_stream_t stream) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || Tk > 256) return 1;
  if (!(p_drop >= 0.f && p_drop < 1.f) || (p_drop > 0.f && !rng_state)) return 1;
  const long strides[] = {q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride, do_bstride, do_tstride,
                          dq_bstride, dq_tstride, dk_bstride, dk_tstride, dv_bstride, dv_tstride};
  for (long s : strides)
    if (s & 7) return 1;  // 16-byte row chunks
  const void* ptrs[] = {Q, K, V, dO, dQ, dK, dV};
  for (const void* q : ptrs)
    if (!q || ((uintptr_t)q & 15)) return 1;
  AbArgs p;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.dO = (const bf16_t*)dO;
  p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV; p.mask = mask;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  p.q_bs = q_bstride; p.q_ts = q_tstride; p.k_bs = k_bstride; p.k_ts = k_tstride; p.v_bs = v_bstride; p.v_ts = v_tstride;
  p.do_bs = do_bstride; p.do_ts = do_tstride; p.dq_bs = dq_bstride; p.dq_ts = dq_tstride;
  p.dk_bs = dk_bstride; p.dk_ts = dk_tstride; p.dv_bs = dv_bstride; p.dv_ts = dv_tstride;
  p.scale = scale; p.p_drop = p_drop; p.rng_state = rng_state; p.site = site;
  hipStream_t st = (hipStream_t)stream;
  const int nkf = (Tk + 15) / 16;
  if (nkf <= 7) return launch_attn_bwd<7>(p, st);
  if (nkf <= 13) return launch_attn_bwd<13>(p, st);
  return launch_attn_bwd<16>(p, st);
}
