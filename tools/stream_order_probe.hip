// Does a kernel always see what its SAME-STREAM predecessor wrote once a second HIP stream is busy?  (gfx950, ROCm 7.2)
//
// with two streams of one process active the forward pass gives results that differ from serial runs
// in 10-50 % of trials.  Tests built from the real pipeline are nearly blind to a stale read: every pass recomputes the
// SAME values into the SAME recycled buffers, so a consumer that reads its input too early reads identical bytes.  This
// probe makes staleness visible: each stream runs a chain  producer(it) -> consumer(it)  on its own buffer, the producer
// writes the iteration number (its late blocks finish last: uneven spin), the consumer -- a block that runs on ANOTHER
// XCD than the writer of its slice -- counts every word that is not `it`.  Consumer forms: plain vector loads, `sc1`
// loads, LDS-DMA (global_load_lds: how the GEMMs stage their operands).  Launch forms: eager, and hipGraph replays.
//
//   hipcc --offload-arch=gfx950 -O3 tools/stream_order_probe.hip -o tools/_bin/stream_order_probe
//   tools/_bin/stream_order_probe [iters=2000] [streams=2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NB = 1024, NT = 256, WPT = 4;   // 1024 blocks x 256 threads x 16 bytes = 4 MiB per buffer
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void spin_us(int us) {
  const unsigned long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long)us * 100) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(NT) void producer(int4* __restrict__ buf, int it, int slow_from, int us) {
  const int b = blockIdx.x;
  if (b >= slow_from) spin_us(us);            // the LAST blocks finish last
  else spin_us((b * 37) % 7);
  buf[b * NT + threadIdx.x] = int4{it, it, it, it};
}

// mode 0: plain loads; 1: sc1 loads (bypass this CU's L1); 2: LDS-DMA then ds_read
template <int MODE>
__global__ __launch_bounds__(NT) void consumer(const int4* __restrict__ buf, int it, unsigned* __restrict__ errors,
                                               unsigned* __restrict__ first_bad) {
  __shared__ int4 stage[NT];
  const int src = (blockIdx.x * 5 + 3) % NB;  // NB % 8 == 0 and 5 b + 3 is odd-shifted: another XCD than the writer's
  const int4* p = buf + src * NT + threadIdx.x;
  int4 v;
  if (MODE == 0) {
    v = *p;
  } else if (MODE == 1) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  } else {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)p, (lds_void_t*)(stage + (threadIdx.x & ~63)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    v = stage[threadIdx.x];
  }
  const int bad = (v.x != it) + (v.y != it) + (v.z != it) + (v.w != it);
  if (bad) {
    atomicAdd(errors, (unsigned)bad);
    atomicMin(first_bad, (unsigned)src);
  }
}

// unrelated load on the other queue: streams through a big buffer (keeps CUs / memory busy unevenly)
__global__ __launch_bounds__(256) void noise(float4* __restrict__ x, long n, float a) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    float4 v = x[i];
    v.x = v.x * a + 1.f; v.y = v.y * a + 1.f; v.z = v.z * a + 1.f; v.w = v.w * a + 1.f;
    x[i] = v;
  }
}

template <int MODE>
static void enqueue_pair(hipStream_t st, int4* buf, int it, unsigned* err, unsigned* fb, int slow_from, int us) {
  hipLaunchKernelGGL(producer, dim3(NB), dim3(NT), 0, st, buf, it, slow_from, us);
  hipLaunchKernelGGL(consumer<MODE>, dim3(NB), dim3(NT), 0, st, (const int4*)buf, it, err, fb);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int S = argc > 2 ? atoi(argv[2]) : 2;
  std::vector<hipStream_t> st(S);
  std::vector<int4*> buf(S);
  std::vector<unsigned*> err(S), fb(S);
  for (int s = 0; s < S; ++s) {
    CHECK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    CHECK(hipMalloc(&buf[s], sizeof(int4) * NB * NT));
    CHECK(hipMemset(buf[s], 0, sizeof(int4) * NB * NT));
    CHECK(hipMalloc(&err[s], 3 * sizeof(unsigned)));
    CHECK(hipMalloc(&fb[s], 3 * sizeof(unsigned)));
  }
  float4* big; const long nbig = 64L << 20;   // 1 GiB
  CHECK(hipMalloc(&big, nbig * sizeof(float4)));
  CHECK(hipMemset(big, 0, nbig * sizeof(float4)));
  hipStream_t ns; CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));

  const char* mode_name[3] = {"plain loads", "sc1 loads", "LDS-DMA (global_load_lds)"};
  for (int with_noise = 0; with_noise < 2; ++with_noise) {
    for (int use_graph = 0; use_graph < 2; ++use_graph) {
      for (int s = 0; s < S; ++s) { CHECK(hipMemset(err[s], 0, 3 * sizeof(unsigned))); CHECK(hipMemset(fb[s], 0xff, 3 * sizeof(unsigned))); }
      CHECK(hipDeviceSynchronize());
      if (!use_graph) {
        for (int it = 1; it <= iters; ++it) {
          for (int s = 0; s < S; ++s) {
            const int us = 8 + 5 * s + (it % 3) * 4, slow_from = NB - 64 - 32 * s;
            enqueue_pair<0>(st[s], buf[s], 3 * it, err[s] + 0, fb[s] + 0, slow_from, us);
            enqueue_pair<1>(st[s], buf[s], 3 * it + 1, err[s] + 1, fb[s] + 1, slow_from, us);
            enqueue_pair<2>(st[s], buf[s], 3 * it + 2, err[s] + 2, fb[s] + 2, slow_from, us);
          }
          if (with_noise && it % 4 == 0) hipLaunchKernelGGL(noise, dim3(512), dim3(256), 0, ns, big, nbig / 8, 1.0001f);
        }
      } else {
        // one graph per stream holding 8 iterations' worth of pairs with FIXED values 1..24 per replay round: the value
        // written is (replay parity * 1000 + k) via two graphs alternated, so a stale read of the previous replay shows
        std::vector<hipGraphExec_t> ge(2 * S);
        for (int s = 0; s < S; ++s)
          for (int par = 0; par < 2; ++par) {
            hipGraph_t g;
            CHECK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < 8; ++k) {
              const int us = 8 + 5 * s + (k % 3) * 4, slow_from = NB - 64 - 32 * s, v = par * 1000 + 3 * k;
              enqueue_pair<0>(st[s], buf[s], v, err[s] + 0, fb[s] + 0, slow_from, us);
              enqueue_pair<1>(st[s], buf[s], v + 1, err[s] + 1, fb[s] + 1, slow_from, us);
              enqueue_pair<2>(st[s], buf[s], v + 2, err[s] + 2, fb[s] + 2, slow_from, us);
            }
            CHECK(hipStreamEndCapture(st[s], &g));
            CHECK(hipGraphInstantiate(&ge[2 * s + par], g, nullptr, nullptr, 0));
          }
        for (int r = 0; r < iters / 8; ++r) {
          for (int s = 0; s < S; ++s) CHECK(hipGraphLaunch(ge[2 * s + (r & 1)], st[s]));
          if (with_noise && r % 2 == 0) hipLaunchKernelGGL(noise, dim3(512), dim3(256), 0, ns, big, nbig / 8, 1.0001f);
        }
      }
      CHECK(hipDeviceSynchronize());
      for (int s = 0; s < S; ++s) {
        unsigned e[3], f[3];
        CHECK(hipMemcpy(e, err[s], sizeof(e), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(f, fb[s], sizeof(f), hipMemcpyDeviceToHost));
        for (int m = 0; m < 3; ++m)
          printf("RESULT streams=%d noise=%d launch=%s stream %d consumer=%-26s stale words %u of %ld (lowest stale slice %d)\n", S,
                 with_noise, use_graph ? "graph" : "eager", s, mode_name[m], e[m], (long)iters * NB * NT * WPT, e[m] ? (int)f[m] : -1);
      }
    }
  }
  return 0;
}
