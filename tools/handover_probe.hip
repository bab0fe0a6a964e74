// Probe for DESIGN.md section 7.1 / VERDICT round 5 item 4: does handing rows from one GEMM to the next INSIDE one persistent launch beat
// the kernel boundary between two launches?  Workload: Y = X W1^T, Z = Y W2^T with X (M x K), W1 (K x K), W2 (K x K) -- the encoder's
// M = 6400, d = 768 sub-round launches (75 tiles of 256 x 256 per GEMM on 256 CUs), bf16, the bare 8-phase schedule of
// tools/gemm8_probe.hip (no epilogue work), random operands.
//   mode 0: two launches on one stream (what the product does);
//   mode 1: ONE launch: workgroups 0 .. 74 compute Y tiles, store them, drain (s_waitcnt vmcnt(0)), barrier, one lane agent-release
//           + relaxed agent atomic add on the row block's counter; workgroups 80 .. 154 compute Z tiles: they stage their W2
//           half-tiles first, then one lane polls the counter of their row block (3 producers: the three column tiles of Y's
//           256-row block; bounded spin), agent-acquire, barrier, and only then stage Y rows (cdna_hip_programming.md section 6,
//           Guideline 16 R1: plain stores + release / acquire);
//   mode 2: as 1 with write-through (sc1) Y stores and no release fence (the guide's publish-large row);
//   mode 3: mode 1's launch WITHOUT the wait (the consumers read whatever Y holds): the floor any hand-over could reach.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/handover_probe.hip -o tools/_bin/handover_probe
//   tools/_bin/handover_probe [M=6400 (rounded up to 256)] [K=768] [reps=50]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int HALF = 128 * 128;   // bytes of one half-tile: 128 rows x 64 bf16
constexpr int BUF = 4 * HALF;     // X0 X1 W0 W1

#define BAR()                           \
  do {                                  \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)

// MODE 0: one GEMM per launch (A, W, C = the operands of that launch).  MODE 1-3: the fused launch: workgroups below `cons0` produce
// Y = X W1^T, workgroups from `cons0` (a multiple of 8: the same XCD labels) consume it for Z = Y W2^T.
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void chain(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W1, bf16_t* __restrict__ Y, const bf16_t* __restrict__ W2,
           bf16_t* __restrict__ Z, int M, int N, int K, int mt, int nt, int cons0, unsigned* __restrict__ cnt, unsigned expect,
           unsigned* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwg = mt * nt;
  const bool consumer = MODE != 0 && (int)blockIdx.x >= cons0;
  const int pid = consumer ? (int)blockIdx.x - cons0 : (int)blockIdx.x;
  if (pid >= nwg) return;
  const bf16_t* __restrict__ A = consumer ? Y : X;
  const bf16_t* __restrict__ W = consumer ? W2 : W1;
  bf16_t* __restrict__ C = consumer ? Z : Y;
  int tile;
  {
    const int xcd = pid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pid >> 3);
  }
  const int m_tile = tile / nt, n_tile = tile % nt;
  const int m0 = m_tile * 256, n0 = n_tile * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // LDS-DMA sources: piece i (0, 1) of this wave inside a half-tile is its 1 KB chunk q = i * 8 + wid = rows 8 q .. 8 q + 7
  const bf16_t* src[4][2];   // [X0 X1 W0 W1][piece]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 8 + wid) * 8 + (lane >> 3), phys = lane & 7;
    const int c = phys ^ ((row >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      src[h][i] = A + (long)(m0 + h * 128 + row) * K + c * 8;
      src[2 + h][i] = W + (long)(n0 + h * 128 + row) * K + c * 8;
    }
  }
  unsigned dma_base = wid * 1024;     // + buffer + half + piece * 8192
  auto stage = [&](int which, int kt, unsigned bufoff) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[which][i] + (long)kt * 64),
                                       (lds_void_t*)(smem + (dma_base + bufoff + which * HALF + i * 8192)), 16, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const unsigned ch0 = (fq ^ sw) << 4;
  // fragment addresses (k-step 0 / 1) inside buffer 0; the buffer is toggled by XOR BUF
  unsigned xa0 = (wr * 64 + fr) * 128 + ch0, xa1 = xa0 ^ 64;
  unsigned wa0 = 2 * HALF + (wc * 32 + fr) * 128 + ch0, wa1 = wa0 ^ 64;

  f32x4 acc[2][2][2][4];   // [h][g][i (W fragment)][j (X fragment)]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][g][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fx[2][4], fw0[2][2], fw1[2][2];   // [k-step][fragment]

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
  auto quadrant = [&](f32x4 (&a)[2][4], const u32x4 (&fw)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[ks][i]), __builtin_bit_cast(bf16x8, fx[ks][j]), a[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nk = K / 64;
  // prologue: K tile 0 whole, K tile 1 except X1 (which phase 1 of tile 0 stages)
  if (!consumer) {
    stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
    stage(2, 1, BUF); stage(0, 1, BUF); stage(3, 1, BUF);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    // the weight half-tiles do not depend on the producers: they go out before the wait
    stage(2, 0, 0); stage(3, 0, 0); stage(2, 1, BUF); stage(3, 1, BUF);
    if (MODE != 3) {
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt + m_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) { *err = 1u; break; }     // bounded: a lost signal must not hang the box
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    stage(0, 0, 0); stage(1, 0, 0); stage(0, 1, BUF);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // everything but X0 of K tile 1
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: wave row 1 runs one barrier behind row 0

  unsigned cur = 0;   // byte offset of the buffer of K tile t
  // KMODE 0: steady state; 1: K tile nk - 2 (only X1 of the last tile is left to stage); 2: last K tile
  auto ktile = [&](int t, auto MODE_) {
    constexpr int KMODE = decltype(MODE_)::value;
    // ---- phase 1
    read_w(0, fw0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(0);
    if (KMODE <= 1) stage(1, t + 1, cur ^ BUF);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the four W0 reads (issued first) have returned
    BAR();
    quadrant(acc[0][0], fw0);
    BAR();
    // ---- phase 2
    read_w(1, fw1);
    if (KMODE == 0) stage(2, t + 2, cur);
    BAR();
    quadrant(acc[0][1], fw1);
    BAR();
    // ---- phase 3
    read_x(1);
    if (KMODE == 0) stage(0, t + 2, cur);
    BAR();
    quadrant(acc[1][1], fw1);
    BAR();
    // ---- phase 4
    if (KMODE == 0) stage(3, t + 2, cur);
    if (KMODE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but the three youngest half-tiles: tile t + 1 is in
    if (KMODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();
    quadrant(acc[1][0], fw0);
    BAR();
    cur ^= BUF; xa0 ^= BUF; xa1 ^= BUF; wa0 ^= BUF; wa1 ^= BUF;
  };
  for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
  ktile(nk - 2, std::integral_constant<int, 1>{});
  ktile(nk - 1, std::integral_constant<int, 2>{});
  if (wr == 0) __builtin_amdgcn_s_barrier();   // row 0 pays back the stagger

#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m0 + h * 128 + wr * 64 + j * 16 + fr, n = n0 + g * 128 + wc * 32 + i * 16 + fq * 4;
          const f32x4 v = acc[h][g][i][j];
          bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          if (MODE == 2 && !consumer) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(C + (long)m * N + n), "v"(__builtin_bit_cast(u32x2, o)) : "memory");
          else *(bf16x4*)(C + (long)m * N + n) = o;
        }
  if (MODE != 0 && !consumer) {
    // publish this Y tile: every wave's stores have left, then ONE lane releases (plain stores) and signals the row block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE != 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(cnt + m_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


static unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 6400;
  const int K = argc > 2 ? atoi(argv[2]) : 768, reps = argc > 3 ? atoi(argv[3]) : 50;
  M = (M + 255) / 256 * 256;
  if (K % 256 || K < 256) { printf("K a multiple of 256\n"); return 1; }
  const int N = K, mt = M / 256, nt = N / 256, tiles = mt * nt, cons0 = (tiles + 7) / 8 * 8;
  std::vector<unsigned short> hX((size_t)M * K), hW1((size_t)K * K), hW2((size_t)K * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.0f - 1.0f; };
  for (auto& v : hX) v = f2bf(rnd());
  const float ws = 1.0f / sqrtf((float)K) * 1.7f;
  for (auto& v : hW1) v = f2bf(rnd() * ws);
  for (auto& v : hW2) v = f2bf(rnd() * ws);
  bf16_t *X, *W1, *W2, *Y, *Z, *Yref, *Zref;
  unsigned *cnt, *err;
  CK(hipMalloc(&X, hX.size() * 2)); CK(hipMalloc(&W1, hW1.size() * 2)); CK(hipMalloc(&W2, hW2.size() * 2));
  CK(hipMalloc(&Y, (size_t)M * N * 2)); CK(hipMalloc(&Z, (size_t)M * N * 2));
  CK(hipMalloc(&Yref, (size_t)M * N * 2)); CK(hipMalloc(&Zref, (size_t)M * N * 2));
  CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W1, hW1.data(), hW1.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W2, hW2.data(), hW2.size() * 2, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto attr = [&](const void* k) { CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF)); };
  attr((const void*)chain<0>); attr((const void*)chain<1>); attr((const void*)chain<2>); attr((const void*)chain<3>);
  unsigned launches = 0;
  auto two = [&](bf16_t* y, bf16_t* z) {
    hipLaunchKernelGGL(chain<0>, dim3(tiles), dim3(512), 2 * BUF, st, X, W1, y, W2, z, M, N, K, mt, nt, 0, cnt, 0u, err);
    hipLaunchKernelGGL(chain<0>, dim3(tiles), dim3(512), 2 * BUF, st, (const bf16_t*)y, W2, z, W2, z, M, N, K, mt, nt, 0, cnt, 0u, err);
  };
  auto fused = [&](int mode) {
    ++launches;
    const unsigned expect = (unsigned)nt * launches;
    if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(cons0 + tiles), dim3(512), 2 * BUF, st, X, W1, Y, W2, Z, M, N, K, mt, nt, cons0, cnt, expect, err);
    if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(cons0 + tiles), dim3(512), 2 * BUF, st, X, W1, Y, W2, Z, M, N, K, mt, nt, cons0, cnt, expect, err);
    if (mode == 3) hipLaunchKernelGGL(chain<3>, dim3(cons0 + tiles), dim3(512), 2 * BUF, st, X, W1, Y, W2, Z, M, N, K, mt, nt, cons0, cnt, expect, err);
  };
  // reference: two launches
  two(Yref, Zref);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned short> zr((size_t)M * N), zz((size_t)M * N);
  CK(hipMemcpy(zr.data(), Zref, zr.size() * 2, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto&& f) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) f();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / reps < best) best = ms / reps;
    }
    return best * 1e3f;
  };
  const double gf = 2.0 * 2.0 * M * (double)N * K / 1e9;
  printf("Y = X W1^T, Z = Y W2^T: M = %d, K = N = %d, %d tiles of 256 x 256 per GEMM, %.1f GFLOP for the pair\n", M, K, tiles, gf);
  const float t_one = timeit([&]() { hipLaunchKernelGGL(chain<0>, dim3(tiles), dim3(512), 2 * BUF, st, X, W1, Yref, W2, Zref, M, N, K, mt, nt, 0, cnt, 0u, err); });
  printf("  one GEMM, one launch                        %7.1f us\n", t_one);
  const float t_two = timeit([&]() { two(Yref, Zref); });
  printf("  mode 0: two launches                        %7.1f us   (%.0f TFLOP/s)\n", t_two, gf / t_two * 1e3);
  for (int mode = 1; mode <= 3; ++mode) {
    CK(hipMemsetAsync(cnt, 0, 4096 * 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
    CK(hipMemsetAsync(Y, 0xff, (size_t)M * N * 2, st)); CK(hipMemsetAsync(Z, 0xff, (size_t)M * N * 2, st));   // NaN poison: a stale read shows
    launches = 0;
    fused(mode);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(zz.data(), Z, zz.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < zz.size(); ++i) diff += zz[i] != zr[i];
    const float t = timeit([&]() { fused(mode); });
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(zz.data(), Z, zz.size() * 2, hipMemcpyDeviceToHost));
    size_t diff2 = 0;
    for (size_t i = 0; i < zz.size(); ++i) diff2 += zz[i] != zr[i];
    printf("  mode %d: %-36s %7.1f us   Z differs from the two-launch result in %zu (first, poisoned launch) / %zu (last) of %zu values%s\n", mode,
           mode == 1 ? "one launch, release / acquire" : mode == 2 ? "one launch, sc1 stores, acquire" : "one launch, NO wait (floor)", t, diff, diff2,
           zz.size(), herr ? "   SPIN LIMIT HIT" : "");
  }
  return 0;
}
