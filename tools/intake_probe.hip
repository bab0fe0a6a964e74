// How fast can one CU take in L2-resident operand bytes, by path?  (round 4: decides whether a GEMM whose W fragments go
// global -> VGPR while A rides the LDS-DMA ring can beat the all-LDS-DMA 128 x 128 kernel, which sits at ~63 GB/s per CU.)
//   hipcc --offload-arch=gfx950 -O3 tools/intake_probe.hip -o tools/_bin/intake_probe && tools/_bin/intake_probe
// Every workgroup (512 threads) streams T "K tiles" out of a 2 MB window (L2-resident on every XCD); per tile and wave:
//   NDMA  x global_load_lds_dwordx4 (1 KB each, into a 2-stage LDS ring)
//   NVEC  x global_load_dwordx4 to VGPRs, PATTERN 0 = full 128-B lines (8 rows x 128 B per instruction),
//                                         PATTERN 1 = MFMA-fragment shaped (16 rows x 64 B per instruction)
//   SHARE = number of waves that read the SAME vector addresses (L1 reuse: 1, 2, 4)
// Two tiles in flight per wave (counted vmcnt).  Prints GB/s per CU of REQUESTED bytes (redundant ones included) and of
// distinct bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NDMA, int NVEC, int PATTERN, int SHARE>
__global__ __launch_bounds__(512) void probe(const unsigned char* __restrict__ buf, unsigned window, int T, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr unsigned DMA_BYTES = NDMA * 8 * 1024, VEC_BYTES = NVEC * (8 / SHARE) * 1024;
  constexpr unsigned TILE = DMA_BYTES + VEC_BYTES > 0 ? DMA_BYTES + VEC_BYTES : 1024;
  unsigned voff[NVEC > 0 ? NVEC : 1];
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const unsigned slice = DMA_BYTES + (wid / SHARE) * NVEC * 1024;
    if (PATTERN == 0) voff[i] = slice + i * 1024 + lane * 16;
    else voff[i] = slice + (i >> 1) * 2048 + (lane & 15) * 128 + (i & 1) * 64 + (lane >> 4) * 16;
  }
  u32x4 acc = {0, 0, 0, 0};
  u32x4 r[2][NVEC > 0 ? NVEC : 1];
  unsigned base = (blockIdx.x * 13u * TILE) % window;
  auto issue = [&](int st, unsigned b) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(buf + b + (i * 8 + wid) * 1024 + lane * 16),
                                       (lds_void_t*)(smem + st * DMA_BYTES + (i * 8 + wid) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NVEC; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[st][i]) : "v"(buf + b + voff[i]) : "memory");
  };
  issue(0, base);
  for (int t = 0; t < T; t += 2) {
    unsigned b1 = base + TILE; if (b1 + TILE > window) b1 = 0;
    issue(1, b1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NVEC) : "memory");
#pragma unroll
    for (int i = 0; i < NVEC; ++i) { asm volatile("" : "+v"(r[0][i])); acc ^= r[0][i]; }
    unsigned b2 = b1 + TILE; if (b2 + TILE > window) b2 = 0;
    issue(0, b2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NVEC) : "memory");
#pragma unroll
    for (int i = 0; i < NVEC; ++i) { asm volatile("" : "+v"(r[1][i])); acc ^= r[1][i]; }
    base = b2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (NDMA > 0) { __syncthreads(); acc[0] ^= *(const unsigned*)(smem + tid * 4); }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int NDMA, int NVEC, int PATTERN, int SHARE>
static void run(const char* name, const unsigned char* buf, unsigned window, unsigned* sink, int wgpc) {
  constexpr unsigned DMA_BYTES = NDMA * 8 * 1024, VEC_BYTES = NVEC * (8 / SHARE) * 1024;
  const int T = 4000;
  const int lds = 2 * DMA_BYTES + 2048;
  auto k = probe<NDMA, NVEC, PATTERN, SHARE>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * wgpc;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), wgpc == 2 ? (lds > 72 * 1024 ? lds : 72 * 1024) : (lds > 90 * 1024 ? lds : 90 * 1024), 0, buf, window, T, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double tiles = (double)grid * (T + 1);
  const double req = tiles * (DMA_BYTES + (double)NVEC * 8 * 1024), distinct = tiles * (DMA_BYTES + VEC_BYTES);
  printf("%-44s wg/CU %d  %7.3f ms  requested %6.1f GB/s/CU (%5.2f TB/s)  distinct %6.1f GB/s/CU\n", name, wgpc, best,
         req / best / 1e6 / 256, req / best / 1e9, distinct / best / 1e6 / 256);
}

int main() {
  const unsigned window = 2u << 20;
  unsigned char* buf; unsigned* sink;
  CK(hipMalloc(&buf, window + (1 << 20))); CK(hipMalloc(&sink, 64));
  std::vector<unsigned> h((window + (1 << 20)) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  CK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int wgpc = 1; wgpc <= 2; ++wgpc) {
    run<4, 0, 0, 1>("LDS-DMA 32 KB/tile", buf, window, sink, wgpc);
    run<2, 0, 0, 1>("LDS-DMA 16 KB/tile", buf, window, sink, wgpc);
    run<0, 4, 0, 1>("VGPR full-line 32 KB/tile", buf, window, sink, wgpc);
    run<0, 4, 1, 1>("VGPR fragment-shaped 32 KB/tile", buf, window, sink, wgpc);
    run<0, 4, 1, 2>("VGPR fragment-shaped, pairs share (16 KB)", buf, window, sink, wgpc);
    run<0, 4, 1, 4>("VGPR fragment-shaped, quads share (8 KB)", buf, window, sink, wgpc);
    run<0, 4, 0, 2>("VGPR full-line, pairs share (16 KB)", buf, window, sink, wgpc);
    run<2, 2, 1, 1>("DMA 16 KB + VGPR frag 16 KB", buf, window, sink, wgpc);
    run<2, 4, 1, 2>("DMA 16 KB + VGPR frag 32 KB req / 16 distinct", buf, window, sink, wgpc);
    run<2, 4, 1, 4>("DMA 16 KB + VGPR frag 32 KB req / 8 distinct", buf, window, sink, wgpc);
    run<2, 2, 0, 1>("DMA 16 KB + VGPR full-line 16 KB", buf, window, sink, wgpc);
    run<2, 4, 0, 2>("DMA 16 KB + VGPR full 32 KB req / 16 distinct", buf, window, sink, wgpc);
    run<4, 4, 1, 1>("DMA 32 KB + VGPR frag 32 KB", buf, window, sink, wgpc);
  }
  return 0;
}
