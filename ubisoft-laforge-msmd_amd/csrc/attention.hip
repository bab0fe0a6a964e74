// Fused softmax attention, head_dim 64, any Tq/Tk (flash-style online softmax over 64-key tiles).
//
// Workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16 queries.
// S is computed SWAPPED (S^T = K . Q^T) so that the query sits on the MFMA lane (l & 15) and the keys
// in the accumulator registers: row max / row sum are lane-local plus two cross-lane steps (xor 16, 32),
// and P^T is already the B operand of the second product O^T = V^T . P^T with no LDS round trip.
// K-slot permutation: the PV MFMA consumes keys in the order the S accumulators hold them
// (slot (q, e) = key 16 f + 4 q + e); the V^T operand is fetched in that same order
//   bf16: ds_read_b64_tr_b16 transposed reads from a row-major [key][d] LDS image (no transpose pass),
//   fp32: ds_read_b32 from a padded [key][d] image (parity mode, exact fp32 MFMA 16x16x4).
#include "common.h"

struct AttnArgs {
  const void* Q; const void* K; const void* V; void* O;
  int B, H, Tq, Tk;
  long qb, qt, kb, kt, vb, vt, ob, ot;
  float scale;
  const uint8_t* mask;
  float p_drop;                     // attention-probability dropout (training); 0 = off
  const unsigned long* rng_state;   // device [seed, step]
  unsigned site;
  // msmd_attention_prefetch: byte ranges this launch also READS (and discards) so that they sit in the memory-side cache
  // when the following kernels want them (the layer's remaining weights); all NULL / 0 otherwise
  const void* pf_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
  long pf_bytes[4] = {0, 0, 0, 0};
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// NW waves per workgroup = 16 NW query rows sharing every staged K / V tile (short sequences fit one workgroup per
// (batch, head): T = 111 -> 7 waves, T = 200 -> 13 waves, so K / V are read from HBM / L2 once instead of once per
// 64 queries).
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_kernel(const AttnArgs p) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int KROW = BF ? 128 : 256;  // bytes per K row in LDS (64 elements)
  constexpr int VROW = BF ? 128 : 272;  // fp32 V rows padded to 68 floats
  constexpr int NCH = BF ? 8 : 16;      // 16-B chunks per 64-element row
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * KROW + 64 * VROW + 1024];   // K, V tiles + 1 KB prefetch sink
  unsigned char* sK = smem;
  unsigned char* sV = smem + 64 * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  constexpr int NT = 64 * NW;
  const int query = blockIdx.x * (16 * NW) + wid * 16 + fr;
  const int qrow = query < p.Tq ? query : p.Tq - 1;
  const T* Qp = (const T*)p.Q + (long)b * p.qb + (long)qrow * p.qt + h * 64;
  const T* Kb = (const T*)p.K + (long)b * p.kb + h * 64;
  const T* Vb = (const T*)p.V + (long)b * p.vb + h * 64;

  // Q fragments (B operand of S^T = K Q^T): chunk (4g + fq) of the lane's query row
  u32x4 qf[BF ? 2 : 4];
#pragma unroll
  for (int g = 0; g < (BF ? 2 : 4); ++g) qf[g] = *(const u32x4*)(Qp + (BF ? 32 : 16) * g + (BF ? 8 : 4) * fq);

  f32x4 acc_o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) acc_o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // K / V tiles are prefetched one tile ahead into registers: the global loads of tile t + 1 are in flight while tile
  // t is being multiplied, so the staging latency (~2 us per 64-key tile) is off the critical path.
  constexpr int NPF = (64 * NCH + NT - 1) / NT;   // 16-byte chunks of each of K, V a thread carries per tile
  u32x4 pk[NPF], pv[NPF];
  auto prefetch = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      // unconditional loads of clamped keys (zeroed on the way into LDS): loads under a run-time test go out one piece at a time
      const int c = min(tid + i * NT, 64 * NCH - 1);
      const int row = c / NCH, ch = c % NCH;
      const int key = min(kv0 + row, p.Tk - 1);
      pk[i] = *(const u32x4*)(Kb + (long)key * p.kt + ch * (16 / sizeof(T)));
      pv[i] = *(const u32x4*)(Vb + (long)key * p.vt + ch * (16 / sizeof(T)));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  prefetch(0);
  // msmd_attention_prefetch / training: byte ranges (the weights of the GEMMs that follow) pulled through the memory-side
  // cache by this launch -- LDS-DMA loads into a sink nobody reads, issued next to the first K / V tile's loads (see
  // attn_whole_kernel)
  if (p.pf_bytes[0] > 0) {
    typedef __attribute__((address_space(3))) void lds_sink_t;
    typedef __attribute__((address_space(1))) const void gbl_src_t;
    const long stride = (long)gridDim.x * gridDim.y * gridDim.z * NT * 16;
    const long wave0 = ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NT + (tid & ~63)) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const char* base = (const char*)p.pf_ptr[r];
      const long nb = p.pf_bytes[r] & ~1023L;
      for (long off = wave0; off < nb; off += stride)
        __builtin_amdgcn_global_load_lds((gbl_src_t*)(base + off + lane * 16), (lds_sink_t*)(smem + 64 * KROW + 64 * VROW), 16, 0, 0);
    }
  }
  for (int kv0 = 0; kv0 < p.Tk; kv0 += 64) {
    __syncthreads();
    // ---- stage the prefetched K and V tiles (rows beyond Tk are zero-filled)
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int c = tid + i * NT;
      if (c < 64 * NCH) {
        const int row = c / NCH, ch = c % NCH;
        const bool live = kv0 + row < p.Tk;
        const u32x4 kk = live ? pk[i] : u32x4{0, 0, 0, 0}, vv = live ? pv[i] : u32x4{0, 0, 0, 0};
        if constexpr (BF) {
          *(u32x4*)(sK + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kk;
          *(u32x4*)(sV + row * 128 + ((((ch >> 1) ^ ((row >> 1) & 3)) << 5) + ((ch & 1) << 4))) = vv;
        } else {
          *(u32x4*)(sK + row * 256 + ((ch ^ (row & 15)) << 4)) = kk;
          *(u32x4*)(sV + row * 272 + (ch << 4)) = vv;
        }
      }
    }
    __syncthreads();
    if (kv0 + 64 < p.Tk) prefetch(kv0 + 64);

    // ---- S^T tile: 4 fragments of 16 keys x 16 queries; the last, ragged tile only touches the nf fragments that
    // hold keys (T = 200: one of four -- 19 % of the kernel's products and softmax work were on padding)
    const int nf = min(4, (p.Tk - kv0 + 15) >> 4);
    f32x4 s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (f >= nf) continue;
      const int row = 16 * f + fr;
      if constexpr (BF) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const u32x4 a = *(const u32x4*)(sK + row * 128 + (((4 * g + fq) ^ ((row >> 1) & 7)) << 4));
          s[f] = mfma16<T>(a, qf[g], s[f]);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 a = *(const u32x4*)(sK + row * 256 + (((4 * g + fq) ^ (row & 15)) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e)
            s[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(qf[g][e]), s[f], 0, 0, 0);
        }
      }
    }

    // ---- mask, online softmax (query = lane & 15; keys 16 f + 4 fq + e).  Keys are only tested where a test can fail:
    // in the last, ragged tile and under an explicit mask (both wave-uniform conditions).
    if (kv0 + 64 > p.Tk || p.mask) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kv0 + 16 * f + 4 * fq + e;
          bool dead = key >= p.Tk;
          if (p.mask && !dead) dead = p.mask[(long)qrow * p.Tk + key] != 0;
          if (dead) s[f][e] = -INFINITY;    // (fragments f >= nf are all dead: their exp below is skipped, p = 0)
        }
    }
    float mx = -INFINITY, ps = 0.f, alpha;
    if constexpr (BF) {
      // 16-bit storage mode: p = exp2(s c - m c) with c = scale log2(e) > 0 -- ONE fma + ONE v_exp_f32 per score (was
      // scale, subtract, scale, exp) and the running maximum taken on the raw scores (the kernel is bound by these
      // vector instructions, not by its 16 MFMAs per tile: DESIGN.md 5b)
      const float c = p.scale * 1.4426950408889634f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[f][e]);      // -inf for dead keys
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float mc = (m_new == -INFINITY) ? 0.f : m_new * c;
      alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc));
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (f >= nf) { s[f] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[f][e], c, -mc));
          s[f][e] = pv;
          ps += pv;
        }
      }
      m_run = m_new;
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[f][e] *= p.scale;                  // -inf stays -inf
          mx = fmaxf(mx, s[f][e]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_use);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = expf(s[f][e] - m_use);
          s[f][e] = pv;
          ps += pv;
        }
      m_run = m_new;
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    if (p.p_drop > 0.f) {  // drop probabilities AFTER the softmax denominator; the 1/(1-p) factor is applied to O
      const unsigned thr = dropout_threshold16(p.p_drop);
      const unsigned long rowbase = ((unsigned long)(b * p.H + h) * p.Tq + qrow) * 128ul;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {   // one generator block per fragment pair (common.h: dropout_value16)
        const Philox4 r = dropout_bits(p.rng_state, p.site, rowbase + (unsigned long)(4 * ((kv0 >> 5) + pr) + fq));
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[2 * pr + o][e] = dropout_value16(r, e, o) >= thr ? s[2 * pr + o][e] : 0.f;
      }
    }
    if (!__all(alpha == 1.0f)) {   // the running maximum moved for some query of this wave: rescale the accumulators
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc_o[d][e] *= alpha;
    }

    // ---- O^T += V^T . P^T
    if constexpr (BF) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int f0 = 2 * pr, f1 = 2 * pr + 1;
        if (f0 >= nf) continue;      // this pair of key fragments is padding
        typedef typename Vec8T<T>::type V8;
        const V8 pbv = V8{(T)s[f0][0], (T)s[f0][1], (T)s[f0][2], (T)s[f0][3],
                          (T)s[f1][0], (T)s[f1][1], (T)s[f1][2], (T)s[f1][3]};
        const u32x4 pb = __builtin_bit_cast(u32x4, pbv);
        // transposed read: lane i = 4 q' + p' of each 16-lane group addresses row key0 + q', cols d0 + 4 p'
        const int qp = fr >> 2, pp = fr & 3;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int r0 = 16 * f0 + 4 * fq + qp, r1 = 16 * f1 + 4 * fq + qp;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sV + r0 * 128 + ((d ^ ((r0 >> 1) & 3)) << 5) + pp * 8));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sV + r1 * 128 + ((d ^ ((r1 >> 1) & 3)) << 5) + pp * 8));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 va = s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          acc_o[d] = mfma16<T>(__builtin_bit_cast(u32x4, va), pb, acc_o[d]);
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int krow = 16 * f + 4 * fq + e;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float a = *(const float*)(sV + krow * 272 + (16 * d + fr) * 4);
            acc_o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s[f][e], acc_o[d], 0, 0, 0);
          }
        }
    }
  }

  if (query < p.Tq) {
    const float inv = 1.0f / (l_run * (1.0f - p.p_drop));
    T* Op = (T*)p.O + (long)b * p.ob + (long)query * p.ot + h * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = acc_o[d][e] * inv;
      if constexpr (BF)
        *(typename Vec4T<T>::type*)(Op + 16 * d + 4 * fq) = pack4<T>(o[0], o[1], o[2], o[3]);
      else
        *(f32x4*)(Op + 16 * d + 4 * fq) = f32x4{o[0], o[1], o[2], o[3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Short sequences, 16-bit storage, no dropout (the whole inference path: T = 200 encoder frames, 111 / 110 decoder
// tokens): ALL keys and values of the (batch, head) are staged once -- Tk <= 208 rows x 128 B x 2 = 52 KB, three
// workgroups per CU -- behind ONE barrier; each wave then holds the complete score row block of its 16 queries in
// registers (13 fragments), so the softmax is the plain two-pass one (no running maximum, no accumulator rescale) and
// the loop over key tiles with its two barriers per tile is gone.  Same operand layouts, same MFMA order per output as
// attn_kernel -- but exp2 arguments are taken against the global row maximum instead of the running one, so results
// differ from attn_kernel in the last bit of a few probabilities (both within the 16-bit modes' tolerance).
constexpr int ATTN_WHOLE_NF = 13;   // key fragments of 16 of the default instantiation: Tk <= 208 (52 KB: three workgroups per CU)
// NF = 17 (Tk <= 272: the decoder at n_motions = 250; 68 KB, two workgroups per CU) takes the same kernel: 25 us against 39 for the
// tiled kernel.  NF = 32 (Tk <= 512: HuBERT-large's 10 s clips; 128 KB, ONE workgroup per CU, 152 registers) was built and
// measured: 118 us against the tiled kernel's 85 -- not kept.  (Round 6: the 13-wave form forced to 72 registers so that TWO workgroups
// share a CU -- 384 workgroups in one round instead of 1.5 -- spilled 28 bytes and ran 18.9 us against 16.3: not kept either.)

template <typename T, int NW, int NF = ATTN_WHOLE_NF>
__global__ __launch_bounds__(64 * NW) void attn_whole_kernel(const AttnArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage modes");
  constexpr int ROWS = 16 * NF, NT = 64 * NW;
  // K, V images of `rows` rows each (the launcher sizes the allocation by Tk, not by NF: T = 111 takes 29 KB and four
  // workgroups share a CU where the full 208-row images allowed three) + 1 KB prefetch sink
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int query = blockIdx.x * (16 * NW) + wid * 16 + fr;
  const int qrow = query < p.Tq ? query : p.Tq - 1;
  const T* Qp = (const T*)p.Q + (long)b * p.qb + (long)qrow * p.qt + h * 64;
  const T* Kb = (const T*)p.K + (long)b * p.kb + h * 64;
  const T* Vb = (const T*)p.V + (long)b * p.vb + h * 64;
  const int nf = (p.Tk + 15) >> 4;            // fragments that hold keys
  const int rows = min(((nf + 1) & ~1) * 16, ROWS);   // staged rows: whole fragment PAIRS (the PV product takes 32 keys), zero-filled
  unsigned char* sV = smem + rows * 128;

  // every load of the kernel goes out here: K, V chunks (16 B), then this lane's Q fragments
  // (unconditional loads of CLAMPED rows, zeroed on the way into LDS: with the loads under `if (row < Tk)` the compiler closed
  // every conditional block with s_waitcnt vmcnt(0) -- the NPF pieces went out one pair at a time, 2-4 serial HBM round trips
  // in front of the first MFMA of a 17-28 us launch)
  constexpr int NPF = (ROWS * 8 + NT - 1) / NT;
  u32x4 pk[NPF], pv[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int c = tid + i * NT, row = min(c >> 3, p.Tk - 1), ch = c & 7;
    pk[i] = *(const u32x4*)(Kb + (long)row * p.kt + ch * 8);
    pv[i] = *(const u32x4*)(Vb + (long)row * p.vt + ch * 8);
  }
  u32x4 qf[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) qf[g] = *(const u32x4*)(Qp + 32 * g + 8 * fq);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int c = tid + i * NT, row = c >> 3, ch = c & 7;
    if (row < rows) {
      const bool live = row < p.Tk;
      *(u32x4*)(sK + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = live ? pk[i] : u32x4{0, 0, 0, 0};
      *(u32x4*)(sV + row * 128 + ((((ch >> 1) ^ ((row >> 1) & 3)) << 5) + ((ch & 1) << 4))) = live ? pv[i] : u32x4{0, 0, 0, 0};
    }
  }
  __syncthreads();
  // ---- prefetch ranges (msmd_attention_prefetch): 16 bytes per lane and load, the whole grid sweeps each range.  LDS-DMA
  // into a 1 KB sink every wave overwrites -- no register waits for data nobody wants.  The point is HBM -> Infinity
  // Cache traffic under this compute-light kernel instead of stalls in the next GEMMs' K loops.  Measured placements
  // (encoder launch, 14 MB of weights, 16.4 us without): here with the builtin 19.8 us, with the inline-asm DMA below 18.2;
  // dedicated prefetch workgroups in an extra blockIdx.x 26 us; the layer's GEMMs give back 14-17 us in every case.
  // the Q fragments must have LANDED before the first prefetch load goes out: vmcnt retires in order, so a wait for Q
  // issued after them would be a wait for all of them (the compiler emits vmcnt(0) in front of the first MFMA otherwise)
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]));
  if (p.pf_bytes[0] > 0) {
    typedef __attribute__((address_space(3))) void lds_sink_t;
    const unsigned sink_lds = (unsigned)(uintptr_t)(lds_sink_t*)smem + 2 * rows * 128;   // LDS byte address of the sink (M0 for the DMA)
    unsigned m0_keep;                                                         // M0 is restored around every DMA
    const long stride = (long)gridDim.x * gridDim.y * gridDim.z * NT * 16;
    const long wave0 = ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NT + (tid & ~63)) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const char* base = (const char*)p.pf_ptr[r];
      const long nb = p.pf_bytes[r] & ~1023L;       // whole 1 KB wave-loads
      // (inline asm, not the builtin: the compiler orders EVERY later LDS read behind a builtin LDS-DMA -- vmcnt(0) in front
      // of the first K fragment read, i.e. the prefetch latency in the critical path: 16.4 -> 19.8 us.  Its own vmcnt
      // arithmetic stays safe: loads it does not know about can only make its waits longer, never shorter.)
      for (long off = wave0; off < nb; off += stride)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_keep) : "v"(base + off + lane * 16), "s"(sink_lds) : "memory");
    }
  }

  // ---- S^T = K . Q^T: fragment f = keys 16 f .. 16 f + 15 (accumulator rows 4 fq + e) x this wave's 16 queries
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
  }
  // ---- mask: the ragged last fragment and the explicit mask (wave-uniform conditions).  The lane's 4 keys of a
  // fragment are 4 consecutive mask bytes: one (unaligned) 4-byte load instead of four byte loads where the row has them
  if ((p.Tk & 15) || p.mask) {
    const uint8_t* mrow = p.mask ? p.mask + (long)qrow * p.Tk : nullptr;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f >= nf || !(p.mask || f == nf - 1)) continue;
      const int k0 = 16 * f + 4 * fq;
      unsigned m4 = 0;
      if (mrow) {
        if (k0 + 4 <= p.Tk) __builtin_memcpy(&m4, mrow + k0, 4);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + e < p.Tk) m4 |= (unsigned)(mrow[k0 + e] != 0) << (8 * e);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e >= p.Tk || ((m4 >> (8 * e)) & 0xff)) s[f][e] = -INFINITY;
    }
  }
  // ---- softmax over the whole row: p = exp2(s c - m c), c = scale log2(e)
  const float c = p.scale * 1.4426950408889634f;
  float mx = -INFINITY;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (f >= nf) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[f][e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mc = (mx == -INFINITY) ? 0.f : mx * c;
  float ps = 0.f;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (f >= nf) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pe = __builtin_amdgcn_exp2f(fmaf(s[f][e], c, -mc));
      s[f][e] = pe;
      ps += pe;
    }
  }
  ps += __shfl_xor(ps, 16, 64);
  ps += __shfl_xor(ps, 32, 64);
  if (p.p_drop > 0.f) {   // training: drop probabilities AFTER the denominator (same Philox stream as attn_kernel and the
                          // backward: common.h dropout_value16); the 1 / (1 - p) factor goes on O
    const unsigned thr = dropout_threshold16(p.p_drop);
    const unsigned long rowbase = ((unsigned long)(b * p.H + h) * p.Tq + qrow) * 128ul;
#pragma unroll
    for (int pr = 0; pr < (NF + 1) / 2; ++pr) {   // one generator block per fragment pair (common.h: dropout_value16)
      if (2 * pr >= nf) continue;
      const Philox4 r = dropout_bits(p.rng_state, p.site, rowbase + (unsigned long)(4 * pr + fq));
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        if (2 * pr + o >= NF) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[2 * pr + o][e] = dropout_value16(r, e, o) >= thr ? s[2 * pr + o][e] : 0.f;
      }
    }
  }

  // ---- O^T = V^T . P^T over fragment pairs (32 keys per MFMA)
  f32x4 acc_o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) acc_o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef typename Vec8T<T>::type V8;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const int qp = fr >> 2, pp = fr & 3;
#pragma unroll
  for (int pr = 0; pr < (NF + 1) / 2; ++pr) {
    const int f0 = 2 * pr, f1 = 2 * pr + 1;
    if (f0 >= nf) continue;
    const f32x4 s1 = f1 < NF ? s[f1 < NF ? f1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};   // (f1 >= nf: zeros from the S loop)
    const V8 pbv = V8{(T)s[f0][0], (T)s[f0][1], (T)s[f0][2], (T)s[f0][3], (T)s1[0], (T)s1[1], (T)s1[2], (T)s1[3]};
    const u32x4 pb = __builtin_bit_cast(u32x4, pbv);
    const int r0 = 16 * f0 + 4 * fq + qp, r1 = f1 < NF ? r0 + 16 : r0;   // f1's rows are staged (zero-filled); past NF: any valid row, p = 0
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (s16x4 __attribute__((address_space(3)))*)(sV + r0 * 128 + ((d ^ ((r0 >> 1) & 3)) << 5) + pp * 8));
      const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (s16x4 __attribute__((address_space(3)))*)(sV + r1 * 128 + ((d ^ ((r1 >> 1) & 3)) << 5) + pp * 8));
      const s16x8 va = s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      acc_o[d] = mfma16<T>(__builtin_bit_cast(u32x4, va), pb, acc_o[d]);
    }
  }
  if (query < p.Tq) {
    const float inv = 1.0f / (ps * (1.0f - p.p_drop));
    T* Op = (T*)p.O + (long)b * p.ob + (long)query * p.ot + h * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *(typename Vec4T<T>::type*)(Op + 16 * d + 4 * fq) = pack4<T>(acc_o[d][0] * inv, acc_o[d][1] * inv, acc_o[d][2] * inv, acc_o[d][3] * inv);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the prefetch loads land in this workgroup's LDS: wait before it is freed
}

// ---------------------------------------------------------------------------------------------------
// MSMD_F16X2 (split pair, common.h) attention for the parity-grade speed mode: Q / K / V arrive in split storage
// straight from the QKV GEMM's epilogue (a 64-wide head = two 32-element blocks = one 256-byte row [hi0|lo0|hi1|lo1]),
// both products run as three f16 MFMAs per k-step (S = Kh.Qh + (Kh.Ql + Kl.Qh) / 2048; the fp32 probabilities are
// split in registers for O = Vh.Ph + (Vh.Pl + Vl.Ph) / 2048), softmax in fp32 (exp_neg_accurate: ~1 ulp, 7 VALU).
// Same structure as attn_kernel (S computed swapped, V^T operand by ds_read_b64_tr_b16 from the row-major image);
// LDS rows are 256 bytes: K chunks XOR (row & 15) (conflict-free ds_read_b128), V chunks XOR the dual-use pattern of
// cdna_hip_programming.md T10 (b).  TO = float (fp32 O) or f16_t (O in split storage for the out-projection GEMM).
__device__ __forceinline__ int vsw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <typename TO, int NW>
__global__ __launch_bounds__(64 * NW) void attn_split_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 64 * 256];
  unsigned char* sK = smem;
  unsigned char* sV = smem + 64 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  constexpr int NT = 64 * NW;
  const int query = blockIdx.x * (16 * NW) + wid * 16 + fr;
  const int qrow = query < p.Tq ? query : p.Tq - 1;
  const f16_t* Qp = (const f16_t*)p.Q + 2 * ((long)b * p.qb + (long)qrow * p.qt) + h * 128;
  const f16_t* Kb = (const f16_t*)p.K + 2 * (long)b * p.kb + h * 128;
  const f16_t* Vb = (const f16_t*)p.V + 2 * (long)b * p.vb + h * 128;

  u32x4 qh[2], ql[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    qh[g] = *(const u32x4*)(Qp + 64 * g + 8 * fq);
    ql[g] = *(const u32x4*)(Qp + 64 * g + 32 + 8 * fq);
  }
  f32x4 o0[4], o1[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o0[d] = o1[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  constexpr int NPF = (64 * 16 + NT - 1) / NT;
  u32x4 pk[NPF], pv[NPF];
  // (unconditional loads of clamped keys, zeroed on the way into LDS: loads under a run-time test are closed with
  // s_waitcnt vmcnt(0) per block, i.e. they go out one piece at a time -- see attn_whole_kernel)
  auto prefetch = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int c = min(tid + i * NT, 64 * 16 - 1);
      const int row = c >> 4, ch = c & 15;
      const int key = min(kv0 + row, p.Tk - 1);
      pk[i] = *(const u32x4*)(Kb + 2 * (long)key * p.kt + ch * 8);
      pv[i] = *(const u32x4*)(Vb + 2 * (long)key * p.vt + ch * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  prefetch(0);
  for (int kv0 = 0; kv0 < p.Tk; kv0 += 64) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int c = tid + i * NT;
      if (c < 64 * 16) {
        const int row = c >> 4, ch = c & 15;
        const bool live = kv0 + row < p.Tk;
        *(u32x4*)(sK + row * 256 + ((ch ^ (row & 15)) << 4)) = live ? pk[i] : u32x4{0, 0, 0, 0};
        *(u32x4*)(sV + row * 256 + ((ch ^ vsw(row)) << 4)) = live ? pv[i] : u32x4{0, 0, 0, 0};
      }
    }
    __syncthreads();
    if (kv0 + 64 < p.Tk) prefetch(kv0 + 64);

    const int nf = min(4, (p.Tk - kv0 + 15) >> 4);   // key fragments of this tile that hold keys (ragged last tile)
    f32x4 s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
      s[f] = s0;
      if (f >= nf) continue;
      const int row = 16 * f + fr;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const u32x4 kh = *(const u32x4*)(sK + row * 256 + (((8 * g + fq) ^ (row & 15)) << 4));
        const u32x4 kl = *(const u32x4*)(sK + row * 256 + (((8 * g + 4 + fq) ^ (row & 15)) << 4));
        s0 = mfma16<f16_t>(kh, qh[g], s0);
        s1 = mfma16<f16_t>(kh, ql[g], s1);
        s1 = mfma16<f16_t>(kl, qh[g], s1);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[f][e] = fmaf(s1[e], MSMD_SPLIT_INV, s0[e]) * p.scale;
    }

    // keys are only tested where a test can fail: the ragged last tile and an explicit mask (wave-uniform conditions)
    if (kv0 + 64 > p.Tk || p.mask) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kv0 + 16 * f + 4 * fq + e;
          bool dead = key >= p.Tk;
          if (p.mask && !dead) dead = p.mask[(long)qrow * p.Tk + key] != 0;
          if (dead) s[f][e] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[f][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = (m_run == -INFINITY) ? 0.f : exp_neg_accurate(m_run - m_use);
    float ps = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f >= nf) { s[f] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = exp_neg_accurate(s[f][e] - m_use);
        s[f][e] = pe;
        ps += pe;
      }
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[d][e] *= alpha; o1[d][e] *= alpha; }
    }

    const int qp = fr >> 2, pp = fr & 3;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int f0 = 2 * pr, f1 = 2 * pr + 1;
      if (f0 >= nf) continue;        // padding
      f16x8 phv, plv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f16_t hh, ll;
        split_f16x2(s[f0][e], hh, ll); phv[e] = hh; plv[e] = ll;
        split_f16x2(s[f1][e], hh, ll); phv[4 + e] = hh; plv[4 + e] = ll;
      }
      const u32x4 ph = __builtin_bit_cast(u32x4, phv), pl = __builtin_bit_cast(u32x4, plv);
      const int r0 = 16 * f0 + 4 * fq + qp, r1 = 16 * f1 + 4 * fq + qp;
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      typedef s16x4 __attribute__((address_space(3)))* lds_s16x4_ptr;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ch = (d >> 1) * 8 + (d & 1) * 2 + (pp >> 1);   // hi chunk of this d block; lo is 4 chunks further
        const int b8 = (pp & 1) * 8;
        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sV + r0 * 256 + ((ch ^ vsw(r0)) << 4) + b8));
        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sV + r1 * 256 + ((ch ^ vsw(r1)) << 4) + b8));
        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sV + r0 * 256 + (((ch + 4) ^ vsw(r0)) << 4) + b8));
        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sV + r1 * 256 + (((ch + 4) ^ vsw(r1)) << 4) + b8));
        const u32x4 vh = __builtin_bit_cast(u32x4, (s16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}));
        const u32x4 vl = __builtin_bit_cast(u32x4, (s16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]}));
        o0[d] = mfma16<f16_t>(vh, ph, o0[d]);
        o1[d] = mfma16<f16_t>(vh, pl, o1[d]);
        o1[d] = mfma16<f16_t>(vl, ph, o1[d]);
      }
    }
  }

  if (query < p.Tq) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaf(o1[d][e], MSMD_SPLIT_INV, o0[d][e]) * inv;
      if constexpr (sizeof(TO) == 4) {
        float* Op = (float*)p.O + (long)b * p.ob + (long)query * p.ot + h * 64;
        *(f32x4*)(Op + 16 * d + 4 * fq) = f32x4{o[0], o[1], o[2], o[3]};
      } else {
        f16_t* Op = (f16_t*)p.O + 2 * ((long)b * p.ob + (long)query * p.ot);
        store4_split(Op, h * 64 + 16 * d + 4 * fq, o);
      }
    }
  }
}

// Waves per workgroup (16 queries each; a workgroup shares every staged K / V tile): the candidate that gives the most
// loaded CU the fewest wave-tiles -- ceil(workgroups / 256 CUs) x waves -- ties to the larger workgroup (K / V read once
// per more queries).  T = 200, 12 heads, B = 32: 13 waves = 384 workgroups = two per CU on half of the chip (26); 7 waves
// = 768 workgroups = three per CU everywhere (21).
static int attn_waves(int Tq, int H, int B) {
  const int cand[4] = {16, 13, 7, 4};
  int best = 4;
  long best_cost = 1L << 60;
  for (int i = 0; i < 4; ++i) {
    const int nw = cand[i];
    const long blocks = (long)((Tq + 16 * nw - 1) / (16 * nw)) * H * B;
    const long cost = ((blocks + 255) / 256) * nw;
    if (cost < best_cost) { best_cost = cost; best = nw; }
  }
  return best;
}

static int attention_impl(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                          long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                          long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                          float p_drop, const unsigned long* rng_state, unsigned site, int dtype,
                          msmd_stream_t stream, const void* const* pf_ptrs = nullptr, const long* pf_bytes = nullptr,
                          int n_pf = 0) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || !Q || !K || !V || !O || !(scale > 0.f)) return 1;
  if (!(p_drop >= 0.f && p_drop < 1.f) || (p_drop > 0.f && (!rng_state || Tk > 512))) return 1;
  const int E = dtype == MSMD_F32 ? 4 : 8;
  if (q_tstride % E || k_tstride % E || v_tstride % E || o_tstride % 4 || q_bstride % E || k_bstride % E ||
      v_bstride % E || o_bstride % 4)
    return 1;
  if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)V & 15) || ((uintptr_t)O & 15)) return 1;
  AttnArgs p{Q, K, V, O, B, H, Tq, Tk, q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride,
             o_bstride, o_tstride, scale, mask, p_drop, rng_state, site};
  if (n_pf < 0 || n_pf > 4 || (n_pf > 0 && (!pf_ptrs || !pf_bytes))) return 1;
  for (int i = 0, j = 0; i < n_pf; ++i)
    if (pf_ptrs[i] && pf_bytes[i] >= 16 && ((uintptr_t)pf_ptrs[i] & 15) == 0) { p.pf_ptr[j] = pf_ptrs[i]; p.pf_bytes[j] = pf_bytes[i]; ++j; }
  int nw = attn_waves(Tq, H, B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype != MSMD_F32 && Tk <= 272) {
    // short sequences: all keys staged once, plain softmax (attn_whole_kernel).  One workgroup per (batch, head) when the
    // queries fit (K / V read once): in the forward step T = 200 runs 16.4 us with 13 waves against 17.5 with 7 (two
    // workgroups per head) and 19.8 for attn_kernel; T = 111 6.6 us with 7 waves against 7.8.
#define MSMD_ATTN_W(T, NWV, NFV)                                                                                     \
  do {                                                                                                               \
    constexpr int lds_max_ = 2 * 16 * NFV * 128 + 1024;                                                              \
    const int rows_ = std::min((((Tk + 15) / 16 + 1) & ~1) * 16, 16 * NFV);                                           \
    const int lds_ = 2 * rows_ * 128 + 1024;                                                                         \
    static bool attr_ = false;                                                                                       \
    auto k_ = attn_whole_kernel<T, NWV, NFV>;                                                                         \
    if (!attr_) { (void)hipFuncSetAttribute((const void*)k_, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max_); attr_ = true; } \
    dim3 grid_((Tq + 16 * NWV - 1) / (16 * NWV), H, B);                                                               \
    hipLaunchKernelGGL(k_, grid_, dim3(64 * NWV), lds_, st, p);                                                       \
  } while (0)
#define MSMD_ATTN_WT(T)                                                                  \
  do {                                                                                   \
    if (Tk <= 16 * ATTN_WHOLE_NF) {                                                      \
      if (Tq <= 112) MSMD_ATTN_W(T, 7, ATTN_WHOLE_NF);                                   \
      else MSMD_ATTN_W(T, 13, ATTN_WHOLE_NF);                                            \
    } else MSMD_ATTN_W(T, 9, 17);                                                        \
  } while (0)
    if (dtype == MSMD_BF16) MSMD_ATTN_WT(bf16_t);
    else MSMD_ATTN_WT(f16_t);
#undef MSMD_ATTN_WT
#undef MSMD_ATTN_W
    MSMD_RETURN_LAST();
  }
  dim3 grid((Tq + 16 * nw - 1) / (16 * nw), H, B), block(64 * nw);
#define MSMD_ATTN(T)                                                                                   \
  do {                                                                                                 \
    if (nw == 4) hipLaunchKernelGGL((attn_kernel<T, 4>), grid, block, 0, st, p);                       \
    else if (nw == 7) hipLaunchKernelGGL((attn_kernel<T, 7>), grid, block, 0, st, p);                  \
    else if (nw == 13) hipLaunchKernelGGL((attn_kernel<T, 13>), grid, block, 0, st, p);                \
    else hipLaunchKernelGGL((attn_kernel<T, 16>), grid, block, 0, st, p);                              \
  } while (0)
  if (dtype == MSMD_BF16) MSMD_ATTN(bf16_t);
  else if (dtype == MSMD_F16) MSMD_ATTN(f16_t);
  else if (dtype == MSMD_F32) MSMD_ATTN(float);
  else return 1;
#undef MSMD_ATTN
  MSMD_RETURN_LAST();
}

extern "C" int msmd_attention(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                              long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                              long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                              int dtype, msmd_stream_t stream) {
  return attention_impl(Q, K, V, O, B, H, Tq, Tk, q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride,
                        o_bstride, o_tstride, scale, mask, 0.f, nullptr, 0u, dtype, stream);
}

// msmd_attention that also pulls up to four byte ranges (the weights the NEXT kernels will read) through the memory-side
// cache while it runs (16-bit modes, Tk <= 208: the whole-sequence kernel; other shapes ignore the ranges).
extern "C" int msmd_attention_prefetch(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                                       long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                                       long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                                       int dtype, const void* const* prefetch_ptrs, const long* prefetch_bytes,
                                       int n_prefetch, msmd_stream_t stream) {
  return attention_impl(Q, K, V, O, B, H, Tq, Tk, q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride,
                        o_bstride, o_tstride, scale, mask, 0.f, nullptr, 0u, dtype, stream, prefetch_ptrs, prefetch_bytes,
                        n_prefetch);
}

// Training-mode forward: probabilities are dropped with probability p_drop (mask = Philox(rng_state, site,
// (b, h, q, key / 4)), regenerated by msmd_attention_bwd with the same arguments).  Tk <= 512.
extern "C" int msmd_attention_dropout(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq,
                                      int Tk, long q_bstride, long q_tstride, long k_bstride, long k_tstride,
                                      long v_bstride, long v_tstride, long o_bstride, long o_tstride, float scale,
                                      const uint8_t* mask, float p_drop, const unsigned long* rng_state,
                                      unsigned int site, int dtype, msmd_stream_t stream) {
  return attention_impl(Q, K, V, O, B, H, Tq, Tk, q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride,
                        o_bstride, o_tstride, scale, mask, p_drop, rng_state, site, dtype, stream);
}

// msmd_attention_dropout with msmd_attention_prefetch's byte ranges (the training forward: the layer's bf16 weight casts).
extern "C" int msmd_attention_dropout_prefetch(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq,
                                               int Tk, long q_bstride, long q_tstride, long k_bstride, long k_tstride,
                                               long v_bstride, long v_tstride, long o_bstride, long o_tstride, float scale,
                                               const uint8_t* mask, float p_drop, const unsigned long* rng_state,
                                               unsigned int site, int dtype, const void* const* prefetch_ptrs,
                                               const long* prefetch_bytes, int n_prefetch, msmd_stream_t stream) {
  return attention_impl(Q, K, V, O, B, H, Tq, Tk, q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride,
                        o_bstride, o_tstride, scale, mask, p_drop, rng_state, site, dtype, stream, prefetch_ptrs,
                        prefetch_bytes, n_prefetch);
}

// Split-pair attention (inference): Q / K / V in MSMD_F16X2 storage, O in fp32 (out_dtype MSMD_F32) or split storage.
// Strides in logical elements, multiples of 32.
extern "C" int msmd_attention_f16x2(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                                    long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                                    long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                                    int out_dtype, msmd_stream_t stream) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || !Q || !K || !V || !O) return 1;
  if (out_dtype != MSMD_F32 && out_dtype != MSMD_F16X2) return 1;
  if (q_tstride % 32 || k_tstride % 32 || v_tstride % 32 || q_bstride % 32 || k_bstride % 32 || v_bstride % 32) return 1;
  if (out_dtype == MSMD_F16X2 ? (o_tstride % 32 || o_bstride % 32) : (o_tstride % 4 || o_bstride % 4)) return 1;
  if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)V & 15) || ((uintptr_t)O & 15)) return 1;
  AttnArgs p{Q, K, V, O, B, H, Tq, Tk, q_bstride, q_tstride, k_bstride, k_tstride, v_bstride, v_tstride,
             o_bstride, o_tstride, scale, mask, 0.f, nullptr, 0u};
  const int nw = attn_waves(Tq, H, B);
  dim3 grid((Tq + 16 * nw - 1) / (16 * nw), H, B), block(64 * nw);
  hipStream_t st = (hipStream_t)stream;
#define MSMD_ATTN_S(TO)                                                                                \
  do {                                                                                                 \
    if (nw == 4) hipLaunchKernelGGL((attn_split_kernel<TO, 4>), grid, block, 0, st, p);                \
    else if (nw == 7) hipLaunchKernelGGL((attn_split_kernel<TO, 7>), grid, block, 0, st, p);           \
    else if (nw == 13) hipLaunchKernelGGL((attn_split_kernel<TO, 13>), grid, block, 0, st, p);         \
    else hipLaunchKernelGGL((attn_split_kernel<TO, 16>), grid, block, 0, st, p);                       \
  } while (0)
  if (out_dtype == MSMD_F32) MSMD_ATTN_S(float);
  else MSMD_ATTN_S(f16_t);
#undef MSMD_ATTN_S
  MSMD_RETURN_LAST();
}



// ---------------------------------------------------------------------------------------------------
// Person-token cross-attention query (sampler fast path, diagonal alignment mask): row 0 of every sequence is the only
// row of the denoiser's cross-attention with a real softmax (rows t >= 1 see exactly one key; model.py:874-878 mask),
// so per layer and denoising step the path needs  a0[n] = softmax(scale (x[n,0] Wq^T + bq)_h K_h[n]^T) V_h[n]  for N
// sequences: a 1-row projection GEMM + a Tq = 1 attention launch, both latency-bound (16 us together inside the
// sampler's graph).  Here ONE wave per (sequence, head) does both on the vector ALU in fp32, all loads issued up front.
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  } else {
    const typename Vec8T<T>::type v = *(const typename Vec8T<T>::type*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
  }
}
__device__ __forceinline__ float group8_sum(float v) {   // over the 8 lanes that share lane / 8
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

// Lane = (g = lane / 8, c = lane % 8): every global access is 8 rows x 128 contiguous bytes (8 lanes x 8 elements).
//   projection: row block rb -> row rb*8+g of Wq_h, the lane's chunks c, c+8, ... of the row against x0's; group sum
//   scores:     iteration it -> key it*8+g, chunk c of the 64-wide head; group sum -> all 8 lanes hold the score
//   output:     the same lanes hold p for their key and read chunk c of its V row; sum over g at the end
// NIT = key iterations held in registers (Tk <= 8 NIT).
template <typename T, int NIT>
__global__ __launch_bounds__(256) void person_query_attention_kernel(const T* __restrict__ x, long x_seq_stride,
                                                                     const T* __restrict__ Wq, const float* __restrict__ bq,
                                                                     const T* __restrict__ K, const T* __restrict__ V,
                                                                     long kv_bstride, long kv_tstride, T* __restrict__ out,
                                                                     int N, int H, int Tk, int d, float scale,
                                                                     const float* __restrict__ wq_colsum = nullptr,
                                                                     float ln_eps = 1e-5f) {
  // wq_colsum (msmd_person_query_attention_ln): x row 0 is UN-normalised; Wq carries the LayerNorm weight folded in, bq the
  // folded bias, wq_colsum[r] = sum_k Wq'[r][k]: q = rstd (Wq' x0 - mu s) + bq with mu / rstd of x0 computed here
  __shared__ float sq[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 3, c = lane & 7;
  const int w = blockIdx.x * 4 + wid;
  if (w >= N * H) return;
  const int n = w / H, h = w - n * H;
  const T* x0 = x + (long)n * x_seq_stride;
  const int nch = d >> 6;                      // chunks of 8 per lane along a d-long row (d = 64 H)
  const T* kbase = K + (long)n * kv_bstride + h * 64 + c * 8;
  const T* vbase = V + (long)n * kv_bstride + h * 64 + c * 8;
  // 2-byte storage, Tk <= 128: the K / V stream (the only HBM-sized traffic here) does not depend on the query, so all
  // of it is requested before the projection starts and lands behind it
  // (K before the projection, V behind it: holding both through the projection left registers for ONE 9-load batch of the
  // projection in flight -- eight serial L2 round trips, half of the launch's 22 us in the sampler step; with V requested after
  // the projection two batches fit under 256 registers, i.e. two workgroups per CU: four round trips)
  constexpr bool PRE = sizeof(T) == 2 && NIT <= 16;
  u32x4 kraw[PRE ? NIT : 1], vraw[PRE ? NIT : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it * 8 < Tk) kraw[it] = *(const u32x4*)(kbase + (long)min(it * 8 + g, Tk - 1) * kv_tstride);
  }
  float qv[8];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) qv[rb] = 0.f;
  // the lane's 8 bias / column-sum values, requested here (under the projection): read where they are used they were 16
  // dependent single loads behind run-time tests, one L2 round trip each
  float bqv[8], csv[8];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) {
    const int r = h * 64 + rb * 8 + g;
    bqv[rb] = bq ? bq[r] : 0.f;
    csv[rb] = wq_colsum ? wq_colsum[r] : 0.f;
  }
  float xs1 = 0.f, xs2 = 0.f;          // sum and sum of squares of this lane's part of x0 (the 8 lanes of a group cover it)
  int i_done = 0;
  if constexpr (PRE) {
    // two chunks of the row per batch: 2 + 16 sixteen-byte loads in flight, then their 128 FMAs per lane
    constexpr int UB = 2;
#pragma unroll 1
    for (; i_done + UB <= nch; i_done += UB) {
      typedef typename Vec8T<T>::type V8;
      V8 xr[UB], wr[UB][8];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        xr[u] = *(const V8*)(x0 + (c + 8 * (i_done + u)) * 8);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) wr[u][rb] = *(const V8*)(Wq + (long)(h * 64 + rb * 8 + g) * d + (c + 8 * (i_done + u)) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);      // all 18 requests go out before the first use (the scheduler otherwise sinks each
#pragma unroll                                // load to its FMAs to save registers: one L2 round trip per load)
      for (int u = 0; u < UB; ++u) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[e] = (float)xr[u][e]; xs1 += xv[e]; xs2 = fmaf(xv[e], xv[e], xs2); }
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
          for (int e = 0; e < 8; ++e) qv[rb] = fmaf(xv[e], (float)wr[u][rb][e], qv[rb]);
      }
    }
  }
  for (int i = i_done; i < nch; ++i) {
    float xv[8];
    load8<T>(x0 + (c + 8 * i) * 8, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) { xs1 += xv[e]; xs2 = fmaf(xv[e], xv[e], xs2); }
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
      float wv[8];
      load8<T>(Wq + (long)(h * 64 + rb * 8 + g) * d + (c + 8 * i) * 8, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[rb] = fmaf(xv[e], wv[e], qv[rb]);
    }
  }
  float ln_mu = 0.f, ln_rs = 1.f;
  if (wq_colsum) {
    const float inv = 1.0f / (float)d;
    ln_mu = group8_sum(xs1) * inv;
    ln_rs = rsqrtf(fmaxf(group8_sum(xs2) * inv - ln_mu * ln_mu, 0.f) + ln_eps);
  }
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) {
    float t = group8_sum(qv[rb]);
    if (wq_colsum) t = ln_rs * (t - ln_mu * csv[rb]);
    if (c == 0) sq[wid][rb * 8 + g] = (t + bqv[rb]) * scale;
  }
  if constexpr (PRE) {      // V goes out now and lands behind the scores and the softmax (unconditional, clamped rows: loads
#pragma unroll              // under a branch are sunk to their use, one L2 round trip each)
    for (int it = 0; it < NIT; ++it) vraw[it] = *(const u32x4*)(vbase + (long)min(it * 8 + g, Tk - 1) * kv_tstride);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_wave_barrier();
  float q8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q8[e] = sq[wid][c * 8 + e];

  float sc[NIT];
  float m = -INFINITY;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int key = it * 8 + g;
    sc[it] = -INFINITY;
    if (it * 8 < Tk) {                         // wave-uniform
      float kvv[8];
      if constexpr (PRE) {
        const typename Vec8T<T>::type v8 = __builtin_bit_cast(typename Vec8T<T>::type, kraw[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) kvv[e] = (float)v8[e];
      } else {
        load8<T>(kbase + (long)min(key, Tk - 1) * kv_tstride, kvv);
      }
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a = fmaf(q8[e], kvv[e], a);
      a = group8_sum(a);
      if (key < Tk) { sc[it] = a; m = fmaxf(m, a); }
    }
  }
  m = wave_max(m);
  float l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int key = it * 8 + g;
    if (it * 8 < Tk) {
      const float pv = key < Tk ? (sizeof(T) == 4 ? expf(sc[it] - m) : __expf(sc[it] - m)) : 0.f;
      l += pv;
      float vv[8];
      if constexpr (PRE) {
        const typename Vec8T<T>::type v8 = __builtin_bit_cast(typename Vec8T<T>::type, vraw[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = (float)v8[e];
      } else {
        load8<T>(vbase + (long)min(key, Tk - 1) * kv_tstride, vv);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(pv, vv[e], o[e]);
    }
  }
  l = wave_sum(l) * 0.125f;                    // every key was counted by its 8 lanes
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = o[e];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    o[e] = t / l;
  }
  if (g == 0) {
    T* op = out + (long)n * d + h * 64 + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) op[e] = from_f32<T>(o[e]);
  }
}


// (A weight-stationary form -- a workgroup owns one head and 8 sequences, the head's 64 x d slice of Wq staged into LDS once
// instead of being read per (sequence, head) wave: 12.6 MB instead of 100 MB through the L2s at N = 192 -- was built in round 5:
// 27.6 us against this kernel's 22.0 in the sampler step (one 8-wave workgroup on 192 CUs, the staging round trip in front of
// every chain).  The chain is latency, not L2 traffic; not kept.)
static int person_query_impl(const void* x, long x_seq_stride, const void* Wq, const float* bq, const void* K,
                             const void* V, long kv_bstride, long kv_tstride, void* out, int N, int H, int Tk, int d,
                             float scale, int dtype, msmd_stream_t stream, const float* wq_colsum, float ln_eps) {
  if (N <= 0 || H <= 0 || Tk <= 0 || Tk > 512 || d != H * 64 || !x || !Wq || !K || !V || !out) return 1;
  const int E = 8;
  if (x_seq_stride % E || kv_bstride % E || kv_tstride % E || ((uintptr_t)x & 15) || ((uintptr_t)Wq & 15) ||
      ((uintptr_t)K & 15) || ((uintptr_t)V & 15))
    return 1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((N * H + 3) / 4), block(256);
#define LAUNCH_PQA(T)                                                                                                  \
  do {                                                                                                                 \
    if (Tk <= 128)                                                                                                     \
      hipLaunchKernelGGL((person_query_attention_kernel<T, 16>), grid, block, 0, st, (const T*)x, x_seq_stride,         \
                         (const T*)Wq, bq, (const T*)K, (const T*)V, kv_bstride, kv_tstride, (T*)out, N, H, Tk, d, scale, wq_colsum, ln_eps); \
    else                                                                                                               \
      hipLaunchKernelGGL((person_query_attention_kernel<T, 64>), grid, block, 0, st, (const T*)x, x_seq_stride,         \
                         (const T*)Wq, bq, (const T*)K, (const T*)V, kv_bstride, kv_tstride, (T*)out, N, H, Tk, d, scale, wq_colsum, ln_eps); \
  } while (0)
  if (dtype == MSMD_F32) LAUNCH_PQA(float);
  else if (dtype == MSMD_BF16) LAUNCH_PQA(bf16_t);
  else if (dtype == MSMD_F16) LAUNCH_PQA(f16_t);
  else return 1;
#undef LAUNCH_PQA
  MSMD_RETURN_LAST();
}

extern "C" int msmd_person_query_attention(const void* x, long x_seq_stride, const void* Wq, const float* bq,
                                           const void* K, const void* V, long kv_bstride, long kv_tstride, void* out,
                                           int N, int H, int Tk, int d, float scale, int dtype, msmd_stream_t stream) {
  return person_query_impl(x, x_seq_stride, Wq, bq, K, V, kv_bstride, kv_tstride, out, N, H, Tk, d, scale, dtype, stream,
                           nullptr, 0.f);
}

// The same with the LayerNorm in front of the query projection folded in: x row 0 un-normalised, Wq / bq the gamma / beta
// folded operands (msmd_gemm_ln's operand form), wq_colsum the row sums of the folded Wq.
extern "C" int msmd_person_query_attention_ln(const void* x, long x_seq_stride, const void* Wq, const float* bq,
                                              const float* wq_colsum, float ln_eps, const void* K, const void* V,
                                              long kv_bstride, long kv_tstride, void* out, int N, int H, int Tk, int d,
                                              float scale, int dtype, msmd_stream_t stream) {
  if (!wq_colsum || !bq) return 1;
  return person_query_impl(x, x_seq_stride, Wq, bq, K, V, kv_bstride, kv_tstride, out, N, H, Tk, d, scale, dtype, stream,
                           wq_colsum, ln_eps);
}

