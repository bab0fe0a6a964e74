// Weight-gradient GEMM ("TN"): C[n][k] = sum_m A[m][n] * B[m][k], both operands row-major with the CONTRACTION
// index m as their slow axis (A = dZ (M, N), B = X (M, K): exactly what a Linear's backward holds), so no
// transposed copies of the activations are ever written to HBM.
//
// gfx950 mapping: 128 x 128 output tile per 512-thread workgroup (8 waves, 2 (n) x 4 (k)), contraction consumed in
// 64-row stages streamed global -> LDS with LDS-DMA (global_load_lds, 16 B / lane) into an NSTAGE ring; the MFMA
// operands need 8 consecutive m for a fixed column, i.e. a COLUMN of the LDS image: read with ds_read_b64_tr_b16
// (the CDNA4 transposing LDS read).  32-byte column blocks are XOR-swizzled with (row & 7) on the source side of
// the DMA so that the 16 rows one tr-read touches fall in distinct banks.  Operands are swapped (D = X^T-frag .
// dZ-frag) so that each lane owns 4 consecutive k of one output row -> 16-byte stores.  The contraction is split
// over blockIdx.y (partial slabs in a caller workspace, summed by a second tiny kernel) because weight-gradient outputs are small (768 x 768 = 36 tiles)
// while M = batch x time is long.  Optional fused bias gradient: colsum[n] = sum_m A[m][n] via one extra MFMA
// against an all-ones fragment in the k-tile-0 workgroups.
//
// Reference semantics: the weight / bias gradients autograd produces for nn.Linear / F.linear in the reference's
// training step (training_script.py:163-201 -> loss.backward()).
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct TnArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  float* colsum;
  int M, N, K;
  long lda, ldb, ldc;
  int nt_n, nt_k, splits, steps_per_split;
  long strideA, strideB, strideC;
  float* ws;       // partial-product slabs [split][batch][N][K] when splits > 1
  long slab;       // elements per slab
  int b_rpw;       // B operand windowed rows: row r -> (r / b_rpw) * b_wstride + (r % b_rpw) * ldb  (0 = plain)
  long b_wstride;
  float inv_rpw;
  int accumulate;  // C += product, colsum += sums (gradient accumulation straight into .grad)
  int plain_order; // tuning key 1 = 1: items in (tile, split) launch order instead of XCD-contiguous eighths
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ds_read_b64_tr_b16 as inline asm: issued through the builtin, the compiler orders every transposed read behind
// ALL outstanding LDS-DMA loads (s_waitcnt vmcnt(0) in front of each group), which serialises the ring on HBM
// latency.  The hand-placed counted vmcnt + barrier below is the real dependency; lgkmcnt is waited by hand too.
template <int OFF>
__device__ __forceinline__ u32x2 tr_read(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

struct Frags {
  u32x2 a[4][2], b[2][2];  // [frag][row-half]: dZ (n) fragments and X (k) fragments of one 32-row k-step
};

template <int G>
__device__ __forceinline__ void read_frags(Frags& f, const unsigned (&aa)[4], const unsigned (&ab)[2]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.a[j][0] = tr_read<G * 8192>(aa[j]);
    f.a[j][1] = tr_read<G * 8192 + 4096>(aa[j]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f.b[i][0] = tr_read<16384 + G * 8192>(ab[i]);
    f.b[i][1] = tr_read<16384 + G * 8192 + 4096>(ab[i]);
  }
}

// wait until at most N LDS reads are outstanding, and tie the wait to the fragments so nothing using them moves above
template <int N>
__device__ __forceinline__ void wait_frags(Frags& f) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]),
                 "+v"(f.a[3][0]), "+v"(f.a[3][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])
               : "n"(N)
               : "memory");
}

__device__ __forceinline__ bf16x8 frag8(const u32x2 lo, const u32x2 hi) {
  return __builtin_bit_cast(bf16x8, (u32x4{lo[0], lo[1], hi[0], hi[1]}));
}

template <int NSTAGE>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const TnArgs p) {
  constexpr int STAGE = 2 * 64 * 256;  // A image [64 m][128 n] + B image [64 m][128 k], bf16
  constexpr int LPT = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Work item = (n tile, k tile, M split).  Workgroups are dealt to the 8 XCDs round-robin by blockIdx.x, so XCD x takes
  // the x-th CONTIGUOUS eighth of the item list ordered with the larger operand's tile index slowest: the items that
  // re-read one 128-column slab of that operand then share one L2 instead of pulling it into all eight.
  int n_tile, k_tile, split;
  {
    const int items = p.nt_n * p.nt_k * p.splits;
    const int per_xcd = (items + 7) >> 3;
    const int lin = p.plain_order ? (int)blockIdx.x : (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (lin >= items || (!p.plain_order && (int)(blockIdx.x >> 3) >= per_xcd)) return;
    split = lin % p.splits;
    const int t = lin / p.splits;
    if (p.nt_n >= p.nt_k) { n_tile = t / p.nt_k; k_tile = t % p.nt_k; }
    else { k_tile = t / p.nt_n; n_tile = t % p.nt_n; }
  }
  const int z = blockIdx.z;
  const bf16_t* __restrict__ A = p.A + z * p.strideA;
  const bf16_t* __restrict__ B = p.B + z * p.strideB;
  const int n0 = n_tile * 128, k0 = k_tile * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nk_total = (p.M + 63) / 64;
  const int kt_begin = split * p.steps_per_split;
  const int kt_end = min(nk_total, kt_begin + p.steps_per_split);
  if (kt_begin >= kt_end) return;

  // per-thread DMA sources: slot id -> (operand, row, physical chunk); logical chunk = swizzle(physical, row)
  const bf16_t* base[LPT];
  long ld[LPT];
  int rowi[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int id = (i * 8 + wid) * 64 + lane;
    const int opnd = id >> 10, idp = id & 1023;
    const int row = idp >> 4, phys = idp & 15;
    const int c = ((((phys >> 1) ^ (row & 7)) << 1) | (phys & 1));
    rowi[i] = row;
    if (opnd == 0) {
      base[i] = A + min(n0 + c * 8, p.N - 8);
      ld[i] = p.lda;
    } else {
      base[i] = B + min(k0 + c * 8, p.K - 8);
      ld[i] = p.ldb;
    }
  }
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int r = min(kt * 64 + rowi[i], p.M - 1);
      long off = (long)r * ld[i];
      if (p.b_rpw > 0 && i >= 2) {  // slots 2, 3 hold the B operand (id >> 10 == 1); fp32 reciprocal + exact fix-up
        int q = (int)((float)r * p.inv_rpw);
        int rem = r - q * p.b_rpw;
        if (rem < 0) { q -= 1; rem += p.b_rpw; }
        if (rem >= p.b_rpw) { q += 1; rem -= p.b_rpw; }
        off = (long)q * p.b_wstride + (long)rem * p.ldb;
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(base[i] + off),
                                       (lds_void_t*)(smem + stage * STAGE + (i * 8 + wid) * 1024), 16, 0, 0);
    }
  };

  const int wn = (wid >> 2) * 64, wk = (wid & 3) * 32;  // wave's n / k offsets inside the tile
  const int fr = lane & 15, fq = lane >> 4;
  const int qp = fr >> 2, pp = fr & 3;
  // transposed-read addresses inside one stage: lane 4q'+p' of a 16-lane group addresses row 4 fq + q' (+16 for the
  // second half, +32 per k-step: immediates), 8 bytes at column block blk ^ (row & 7)
  const int rr = 4 * fq + qp;
  unsigned aa[4], ab[2];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
#pragma unroll
  for (int j = 0; j < 4; ++j) aa[j] = rr * 256 + ((((wn >> 4) + j) ^ (rr & 7)) << 5) + pp * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) ab[i] = rr * 256 + ((((wk >> 4) + i) ^ (rr & 7)) << 5) + pp * 8;

  f32x4 acc[2][4];  // [k frag][n frag]
  f32x4 acs[4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) acs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = (p.colsum != nullptr) && k_tile == 0 && (wid & 3) == 0;
  const bf16x8 ones = bf16x8{(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f,
                             (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};

  auto mma = [&](const Frags& f, int m_base) {
    bf16x8 fa[4], fb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = frag8(f.a[j][0], f.a[j][1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) fb[i] = frag8(f.b[i][0], f.b[i][1]);
    if (m_base + 32 > p.M) {  // contraction rows >= M were clamped duplicates: zero them in the dZ operand
      const int mb = m_base + 4 * fq;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (mb + (e >> 2) * 16 + (e & 3) >= p.M) fa[j][e] = (bf16_t)0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    if (do_cs) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[j], acs[j], 0, 0, 0);
    }
  };

  const int nk = kt_end - kt_begin;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(kt_begin + s, s);
  int stage = 0;
  for (int it = 0; it < nk; ++it) {
    if (it + NSTAGE - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + NSTAGE - 1 < nk) issue(kt_begin + it + NSTAGE - 1, (stage + NSTAGE - 1) % NSTAGE);
    const unsigned sbase = lds0 + stage * STAGE;
    unsigned ra[4], rb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = sbase + aa[j];
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = sbase + ab[i];
    const int m_stage = (kt_begin + it) * 64;
    Frags f0, f1;
    read_frags<0>(f0, ra, rb);
    read_frags<1>(f1, ra, rb);
    wait_frags<12>(f0);
    mma(f0, m_stage);
    wait_frags<0>(f1);
    mma(f1, m_stage + 32);
    stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
  }

  // epilogue: acc[i][j][e] = C[n = n0 + wn + 16 j + fr][k = k0 + wk + 16 i + 4 fq + e]
  const bool partial = p.splits > 1;
  float* __restrict__ C = partial ? p.ws + split * p.slab + (long)z * p.N * p.K : p.C + z * p.strideC;
  const long ldc = partial ? p.K : p.ldc;
  const bool vec = (ldc % 4 == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn + 16 * j + fr;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = k0 + wk + 16 * i + 4 * fq;
      float* dst = C + (long)n * ldc + k;
      const bool add = p.accumulate && !partial;
      if (vec && k + 3 < p.K) {
        *(f32x4*)dst = add ? *(const f32x4*)dst + acc[i][j] : acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < p.K) dst[e] = add ? dst[e] + acc[i][j][e] : acc[i][j][e];
      }
    }
    if (do_cs && fq == 0) {
      if (partial || p.accumulate) atomicAdd(p.colsum + n, acs[j][0]);
      else p.colsum[n] = acs[j][0];
    }
  }
}

// (Round 6 built this contraction on 256 x 256 output tiles with gemm.hip's 8-phase schedule -- gemm_tn8_kernel, commit fbbdd5f:
// correct on every test, no scratch, 229 registers -- and removed it: one workgroup per CU needs 7-28 contraction splits to fill
// the chip where this kernel takes 4-6, so the partial slabs double (64 MB written and read back per weight gradient), and its K
// tile ran at 1.8 us per workgroup against the 1.0 us of a saturated matrix pipe (48 transposed 8-byte LDS reads per phase where
// the forward kernel issues 24 of 16 bytes).  us per call with the slab reduction, M = 12800: 768 x 768 35.6 -> 39.6,
// 2304 x 768 69.4 -> 69.3, 3072 x 768 82.5 -> 81.4, 768 x 3072 82.8 -> 83.6; M = 7040 decoder shapes 20-34 -> 32-40.  What the
// numbers ask for instead is one GROUPED launch per layer's four weight gradients -- 108 tiles, two splits -- DESIGN.md section 5.)
// C[z][n][k] = sum_s ws[s][z][n][k]
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int splits,
                                                        long slab, int N, int K, long ldc, long strideC, int accumulate) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;  // float4 index inside one batch's (N, K)
  const int z = blockIdx.y;
  const int kq = K >> 2;
  if (q >= (long)N * kq) return;
  const int n = (int)(q / kq), k = (int)(q % kq) * 4;
  const float* src = ws + (long)z * N * K + (long)n * K + k;
  f32x4 a = *(const f32x4*)src;
  for (int s = 1; s < splits; ++s) {
    const f32x4 b = *(const f32x4*)(src + s * slab);
    a += b;
  }
  float* dst = C + z * strideC + (long)n * ldc + k;
  if (accumulate) { a[0] += dst[0]; a[1] += dst[1]; a[2] += dst[2]; a[3] += dst[3]; }
  if (ldc % 4 == 0) *(f32x4*)dst = a;
  else { dst[0] = a[0]; dst[1] = a[1]; dst[2] = a[2]; dst[3] = a[3]; }
}

}  // namespace

// Contraction splits: fill the 512 workgroup slots of the chip (2 per CU at 64 KB of LDS), never more; the slab
// reduction costs splits x N x K x 4 B of traffic, so large outputs take at most 4 (6 while 4 would leave more than half
// of the slots empty: 768 x 768 = 36 tiles, M = 6400: 30.2 us with 4, 24.8 with 6, 26.2 with 8 -- tools/bench_gemm_tn.py).
static long tn_auto_splits(long tiles, long nk_elems) {
  long s = 512 / (tiles > 0 ? tiles : 1);
  const long cap = nk_elems >= (1 << 19) ? (tiles * 4 < 256 ? 6 : 4) : 8;
  return max(1L, min(s, cap));
}

// C (N, K) fp32 = A^T . B with A (M, N) bf16, B (M, K) bf16 (contraction over rows).  N % 8 == 0, K % 8 == 0,
// lda / ldb multiples of 8 elements, pointers 16-byte aligned.  colsum (N) fp32 or NULL.  batch >= 1 with element
// strides.  b_rows_per_window / b_window_stride: B may be a windowed view (row r -> (r / rpw) * stride + (r % rpw) * ldb:
// overlapping conv windows of a padded signal, so a conv weight gradient needs no unfold); 0 = plain.
// accumulate: C += product and colsum += sums instead of overwriting (gradient accumulation into .grad).
// ws / ws_bytes: optional workspace for split-contraction partial slabs (msmd_gemm_tn_workspace gives the
// size that lets the launch fill the chip; smaller or NULL just means fewer / no splits).
extern "C" long msmd_gemm_tn_workspace(int M, int N, int K, int batch) {
  const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128) * batch;
  const int nk = (M + 63) / 64;
  long splits = tn_auto_splits(tiles, (long)N * K);
  splits = max(1L, min(splits, (long)max(1, nk / 4)));
  return splits > 1 ? splits * batch * (long)N * K * (long)sizeof(float) : 0;
}

extern "C" int msmd_gemm_tn(const void* A, const void* B, float* C, float* colsum, int M, int N, int K, long lda,
                            long ldb, long ldc, int batch, long strideA, long strideB, long strideC,
                            int b_rows_per_window, long b_window_stride, int accumulate, void* ws,
                            long ws_bytes, msmd_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || batch < 1) return 1;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || ((uintptr_t)ws & 15)) return 1;
  if (colsum && batch != 1) return 1;
  hipStream_t st = (hipStream_t)stream;
  TnArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.colsum = colsum;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.nt_n = (N + 127) / 128; p.nt_k = (K + 127) / 128;
  p.strideA = strideA; p.strideB = strideB; p.strideC = strideC;
  const long tiles = (long)p.nt_n * p.nt_k * batch;
  const int nk = (M + 63) / 64;
  long splits = tn_auto_splits(tiles, (long)N * K);
  const int forced = (accumulate >> 8) & 0xff;   // per-call override of the contraction split count (tests, tuning)
  accumulate &= 1;
  if (forced > 0) splits = forced;
  if (MSMD_TUNE(2) > 0) splits = MSMD_TUNE(2);
  splits = max(1L, min(splits, (long)max(1, nk / 4)));
  const long slab = (long)batch * N * K;
  if (!ws) splits = 1;
  else splits = min(splits, ws_bytes / (slab * (long)sizeof(float)));
  splits = max(1L, splits);
  p.steps_per_split = (int)((nk + splits - 1) / splits);
  p.splits = (nk + p.steps_per_split - 1) / p.steps_per_split;
  p.ws = (float*)ws; p.slab = slab;
  p.b_rpw = (b_rows_per_window > 0 && b_rows_per_window < M) ? b_rows_per_window : 0;
  p.b_wstride = b_window_stride; p.inv_rpw = p.b_rpw ? 1.0f / (float)p.b_rpw : 0.f;
  if (p.b_rpw && ((b_window_stride & 7) || M >= (1 << 24))) return 1;
  p.accumulate = accumulate ? 1 : 0;
  p.plain_order = MSMD_TUNE(1) == 1;
  if (p.splits > 1 && colsum && !accumulate) {
    hipError_t e = msmd_zero_async(colsum, sizeof(float) * N, st);
    if (e != hipSuccess) return (int)e;
  }
  // 2-stage ring = 64 KB of LDS = TWO workgroups per CU: a K tile costs a workgroup ~0.7 us whatever the ring depth
  // (measured with 2 / 3 / 4 stages), so throughput comes from co-residency, as in the forward GEMM
  constexpr int NSTAGE = 2;
  constexpr int lds = NSTAGE * 2 * 64 * 256;
  static bool attr_done = false;
  auto kfn = gemm_tn_kernel<NSTAGE>;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(((p.nt_n * p.nt_k * p.splits + 7) / 8) * 8, 1, batch), dim3(512), lds, st, p);
  if (p.splits > 1) {
    const long quads = (long)N * (K / 4);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((quads + 255) / 256), batch), dim3(256), 0, st, p.ws, C,
                       p.splits, slab, N, K, ldc, strideC, p.accumulate);
  }
  MSMD_RETURN_LAST();
}
