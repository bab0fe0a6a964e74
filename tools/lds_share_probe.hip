// What can one CU sustain when LDS-DMA writes, LDS fragment reads and MFMAs run free (no barriers, no data dependence
// between them), in the per-K-tile ratios of the GEMM tiles?  Tells whether the 128 x 128 kernel's ~0.52 us per K tile and
// CU (hot) is a scheduling problem or the LDS array's / matrix pipe's own limit.  (round 4)
//   hipcc --offload-arch=gfx950 -O3 tools/lds_share_probe.hip -o tools/_bin/lds_share_probe
// Per wave and iteration ("K tile"): NDMA x global_load_lds_dwordx4 (1 KB each, L2-resident source, LDS ring nobody reads),
// NVEC x global_load_dwordx4 full-line loads to registers, NREAD x ds_read_b128 (conflict-free), NMFMA x
// v_mfma_f32_16x16x32_bf16 on registers.  8 waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NDMA, int NVEC, int NREAD, int NMFMA>
__global__ __launch_bounds__(512) void probe(const unsigned char* __restrict__ buf, unsigned window, int T, unsigned* sink,
                                             long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr unsigned TILE = (NDMA + NVEC) * 8 * 1024 > 0 ? (NDMA + NVEC) * 8 * 1024 : 1024;
  constexpr unsigned RING = 2 * NDMA * 8 * 1024;
  // fragment-read addresses: 128-byte rows, chunk XOR-swizzled by (row >> 1) & 7 (the GEMM's conflict-free image)
  const int fr = lane & 15, fq = lane >> 4;
  unsigned raddr[NREAD > 0 ? NREAD : 1];
#pragma unroll
  for (int i = 0; i < NREAD; ++i) {
    const int row = ((wid * 16 + i * 16) & 111) + fr, chunk = (fq + 4 * (i & 1)) ^ ((row >> 1) & 7);
    raddr[i] = RING + row * 128 + chunk * 16;
  }
  for (int i = tid; i < 4096; i += 512) ((unsigned*)(smem + RING))[i] = i * 2654435761u;
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fa = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, fb = {0x3f803f80u, 0x3f003f00u + lane, 0x3f803f80u, 0x3f803f80u};
  u32x4 xacc = {0, 0, 0, 0};
  unsigned base = (blockIdx.x * 13u * TILE) % window;
  const long long t0 = __builtin_amdgcn_s_memtime();
  u32x4 v[2][NVEC > 0 ? NVEC : 1];
#pragma unroll
  for (int i = 0; i < NVEC; ++i) v[0][i] = v[1][i] = u32x4{0, 0, 0, 0};
  // one iteration; `st` is a compile-time constant at both call sites (register arrays stay registers).  The vector loads of
  // iteration t land in v[st] and are only touched after the wait of iteration t + 1 (which retires them): an asm load's
  // destination must stay allocated until its data has arrived
  auto iter = [&](auto ST) {
    constexpr int st = decltype(ST)::value;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(buf + base + (i * 8 + wid) * 1024 + lane * 16),
                                       (lds_void_t*)(smem + (st * NDMA + i) * 8 * 1024 + wid * 1024), 16, 0, 0);
    // plain loads: the compiler tracks their landing itself (an asm load's destination may be a temporary that is copied
    // and re-used while the data is still in flight -- that corrupted address registers in the first version of this probe)
#pragma unroll
    for (int i = 0; i < NVEC; ++i) v[st][i] = *(const u32x4*)(buf + base + ((NDMA + i) * 8 + wid) * 1024 + lane * 16);
    u32x4 r[NREAD > 0 ? NREAD : 1];
#pragma unroll
    for (int i = 0; i < NREAD; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i]) : "v"(raddr[i]) : "memory");
#pragma unroll
    for (int i = 0; i < NMFMA; ++i)
      acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i & 3], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NREAD; ++i) { asm volatile("" : "+v"(r[i])); if (i % 6 == 0) xacc ^= r[i]; }
    // leave this iteration's vector-memory operations in flight: wait only for the previous iteration's
    if constexpr (NVEC == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
#pragma unroll
    for (int i = 0; i < NVEC; ++i) xacc ^= v[st ^ 1][i];      // the previous iteration's loads (the compiler waits as it sees fit)
    base += TILE; if (base + TILE > window) base = 0;
  };
  for (int t = 0; t < T; t += 2) { iter(std::integral_constant<int, 0>{}); iter(std::integral_constant<int, 1>{}); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NVEC; ++i) xacc ^= v[0][i] ^ v[1][i];
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345f || (xacc[0] ^ xacc[1] ^ xacc[2] ^ xacc[3]) == 0x12345678u) sink[0] = 1;
}

template <int NDMA, int NVEC, int NREAD, int NMFMA>
static void run(const char* name, const unsigned char* buf, unsigned window, unsigned* sink, long long* cyc, int wgpc) {
  const int T = 3000;
  auto k = probe<NDMA, NVEC, NREAD, NMFMA>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int need = 2 * NDMA * 8 * 1024 + 16384;
  int lds = wgpc == 2 ? 72 * 1024 : 90 * 1024;
  if (need > lds) lds = need;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * wgpc;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, buf, window, T, sink, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<long long> h(grid);
  CK(hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
  long long med; { std::vector<long long> s = h; std::sort(s.begin(), s.end()); med = s[grid / 2]; }
  const double us_per_iter_cu = best * 1e3 / T / wgpc;     // time per K tile and CU
  // s_memtime ticks at 100 MHz on gfx950; shader cycles come from the MFMA-only calibration run (printed first)
  printf("%-40s wg/CU %d  %7.3f ms  %6.3f us per K tile per CU  (memtime ticks/iter %.1f)  DMA %5.1f GB/s/CU  MFMA %5.0f TF/s chip\n", name, wgpc,
         best, us_per_iter_cu, (double)med / T, (NDMA + NVEC) * 8192.0 / us_per_iter_cu / 1e3,
         NMFMA * 8 * 2.0 * 16 * 16 * 32 * 256 / us_per_iter_cu / 1e6);
}

#include <algorithm>
int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const unsigned window = 2u << 20;
  unsigned char* buf; unsigned* sink; long long* cyc;
  CK(hipMalloc(&buf, window + (1 << 20))); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 1024 * 8));
  std::vector<unsigned> h((window + (1 << 20)) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  CK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int wgpc = 1; wgpc <= 2; ++wgpc) {
    run<0, 0, 0, 16>("MFMA only (16)", buf, window, sink, cyc, wgpc);
    run<0, 0, 12, 0>("reads only (12)", buf, window, sink, cyc, wgpc);
    run<4, 0, 0, 0>("DMA only (4)", buf, window, sink, cyc, wgpc);
    run<4, 0, 12, 0>("DMA 4 + reads 12", buf, window, sink, cyc, wgpc);
    run<0, 0, 12, 16>("reads 12 + MFMA 16", buf, window, sink, cyc, wgpc);
    run<4, 0, 0, 16>("DMA 4 + MFMA 16", buf, window, sink, cyc, wgpc);
    run<4, 0, 12, 16>("128x128 tile: DMA 4 + reads 12 + MFMA 16", buf, window, sink, cyc, wgpc);
    run<5, 0, 14, 24>("192x128 tile: DMA 5 + reads 14 + MFMA 24", buf, window, sink, cyc, wgpc);
    run<6, 0, 16, 32>("256x128 tile: DMA 6 + reads 16 + MFMA 32", buf, window, sink, cyc, wgpc);
    run<8, 0, 24, 64>("256x256 tile: DMA 8 + reads 24 + MFMA 64", buf, window, sink, cyc, wgpc);
    run<2, 4, 8, 16>("128x128, W direct: DMA 2 + vec 4 + reads 8 + MFMA 16", buf, window, sink, cyc, wgpc);
    run<4, 8, 8, 32>("256x128, W direct: DMA 4 + vec 8 + reads 8 + MFMA 32", buf, window, sink, cyc, wgpc);
    run<0, 4, 12, 16>("vec 4 (no DMA) + reads 12 + MFMA 16", buf, window, sink, cyc, wgpc);
  }
  return 0;
}
