// Stand-alone probe of the 256 x 256 x 64 "8-phase" GEMM schedule of cdna_hip_programming.md section 5 on this library's operand
// layout (C[M][N] = A[M][K] . W[N][K]^T, bf16 in, fp32 accumulate, bf16 out), stripped of everything the product kernel carries
// (windowed rows, ragged edges, epilogue families): what does the bare schedule reach on this machine, and which details matter?
//   hipcc --offload-arch=gfx950 -O3 tools/gemm8_probe.hip -o tools/_bin/gemm8_probe
//   tools/_bin/gemm8_probe M N K [variant] [reps]        (M, N multiples of 256; K a multiple of 64, K >= 128)
// Prints TFLOP/s (random operands in [-1, 1)) and the worst deviation from an fp32 dot product on 4096 sampled outputs.
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

// VAR bit 0: no s_setprio pair; bit 1: no stagger of the wave rows; bit 2: plain tile order (no XCD chunks);
// bit 3: every tile reads the first 4096 rows of A (an L2-resident operand: what does streaming A from HBM cost a tall grid?);
// bit 4: no output stores (only a lane whose result is NaN stores): what does draining the output cost a one-round launch?
template <int VAR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm8(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K, int mt, int nt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pid = blockIdx.x, nwg = mt * nt;
  int tile = pid;
  if (!(VAR & 4)) {
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
      src[h][i] = A + (long)(((VAR & 8) ? (m0 & 4095) : m0) + h * 128 + row) * K + c * 8;
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
    if (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[ks][i]), __builtin_bit_cast(bf16x8, fx[ks][j]), a[i][j], 0, 0, 0);
    if (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
  };

  const int nk = K / 64;
  // prologue: K tile 0 whole, K tile 1 except X1 (which phase 1 of tile 0 stages)
  stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
  stage(2, 1, BUF); stage(0, 1, BUF); stage(3, 1, BUF);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!(VAR & 2) && wr == 1) __builtin_amdgcn_s_barrier();   // stagger: wave row 1 runs one barrier behind row 0

  unsigned cur = 0;   // byte offset of the buffer of K tile t
  // MODE 0: steady state; 1: K tile nk - 2 (only X1 of the last tile is left to stage); 2: last K tile
  auto ktile = [&](int t, auto MODE_) {
    constexpr int MODE = decltype(MODE_)::value;
    // ---- phase 1
    read_w(0, fw0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(0);
    if (MODE <= 1) stage(1, t + 1, cur ^ BUF);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the four W0 reads (issued first) have returned
    BAR();
    quadrant(acc[0][0], fw0);
    BAR();
    // ---- phase 2
    read_w(1, fw1);
    if (MODE == 0) stage(2, t + 2, cur);
    BAR();
    quadrant(acc[0][1], fw1);
    BAR();
    // ---- phase 3
    read_x(1);
    if (MODE == 0) stage(0, t + 2, cur);
    BAR();
    quadrant(acc[1][1], fw1);
    BAR();
    // ---- phase 4
    if (MODE == 0) stage(3, t + 2, cur);
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but the three youngest half-tiles: tile t + 1 is in
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();
    quadrant(acc[1][0], fw0);
    BAR();
    cur ^= BUF; xa0 ^= BUF; xa1 ^= BUF; wa0 ^= BUF; wa1 ^= BUF;
  };
  for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
  ktile(nk - 2, std::integral_constant<int, 1>{});
  ktile(nk - 1, std::integral_constant<int, 2>{});
  if (!(VAR & 2) && wr == 0) __builtin_amdgcn_s_barrier();   // row 0 pays back the stagger

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
          if (!(VAR & 16) || v[0] != v[0]) *(bf16x4*)(C + (long)m * N + n) = o;
        }
}

__global__ void ref_samples(const bf16_t* A, const bf16_t* W, const int* mn, float* out, int K, int n_s) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_s) return;
  const bf16_t* a = A + (long)mn[2 * s] * K;
  const bf16_t* w = W + (long)mn[2 * s + 1] * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += (float)a[k] * (float)w[k];
  out[s] = acc;
}

static unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int VAR>
static void run(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int N, int K, hipStream_t st) {
  auto k = gemm8<VAR>;
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF)); done = true; }
  const int mt = M / 256, nt = N / 256;
  hipLaunchKernelGGL(k, dim3(mt * nt), dim3(512), 2 * BUF, st, A, W, C, M, N, K, mt, nt);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const int var = argc > 4 ? atoi(argv[4]) : 0, reps = argc > 5 ? atoi(argv[5]) : 20;
  if (M % 256 || N % 256 || K % 64 || K < 128) { printf("M, N multiples of 256; K a multiple of 64, >= 128\n"); return 1; }
  std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.0f - 1.0f; };
  for (auto& v : hA) v = f2bf(rnd());
  for (auto& v : hW) v = f2bf(rnd());
  bf16_t *A, *W, *C;
  CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(C, 0xff, (size_t)M * N * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto launch = [&]() {
    switch (var) {
      case 0: run<0>(A, W, C, M, N, K, st); break;
      case 1: run<1>(A, W, C, M, N, K, st); break;
      case 2: run<2>(A, W, C, M, N, K, st); break;
      case 4: run<4>(A, W, C, M, N, K, st); break;
      case 8: run<8>(A, W, C, M, N, K, st); break;
      case 16: run<16>(A, W, C, M, N, K, st); break;
      default: printf("variant?\n"); exit(1);
    }
  };
  launch();
  CK(hipStreamSynchronize(st));
  // check 4096 sampled outputs
  const int n_s = 4096;
  std::vector<int> mn(2 * n_s);
  for (int i = 0; i < n_s; ++i) { s = s * 1664525u + 1013904223u; mn[2 * i] = (s >> 4) % M; s = s * 1664525u + 1013904223u; mn[2 * i + 1] = (s >> 4) % N; }
  int* dmn; float* dref;
  CK(hipMalloc(&dmn, mn.size() * 4)); CK(hipMalloc(&dref, n_s * 4));
  CK(hipMemcpy(dmn, mn.data(), mn.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ref_samples, dim3(n_s / 256), dim3(256), 0, st, A, W, dmn, dref, K, n_s);
  std::vector<float> ref(n_s);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ref.data(), dref, n_s * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned short> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0; int bad = 0;
  for (int i = 0; i < n_s; ++i) {
    const float c = bf2f(hC[(size_t)mn[2 * i] * N + mn[2 * i + 1]]);
    const double e = fabs(c - ref[i]), tol = 0.01 * fabs(ref[i]) + 0.02;
    if (!(e <= tol) && !(var & 24)) ++bad;
    if (e > worst) worst = e;
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / reps < best) best = ms / reps;
  }
  printf("%d x %d x %d variant %d: %.1f us  %.1f TFLOP/s   sampled check: worst |err| %.4f, %d of %d outside tolerance\n", M, N, K, var,
         best * 1e3, 2.0 * M * N * K / best / 1e9, worst, bad, n_s);
  return bad ? 2 : 0;
}
