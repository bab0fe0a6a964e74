// Measurement aid (not on the hot path): a kernel of KNOWN duration.  bench.py brackets individual launches of the
// dominant kernel with HIP events on the launch stream; an event pair also measures the queue's record / dispatch
// latency, which is not additive with an empty pair (two back-to-back records sit ~5 us apart, a record behind a
// kernel completes with it).  Timing this kernel at two known lengths gives the pair's true overhead as the intercept
// of (measured - known), so per-launch figures agree with rocprofv3 --kernel-trace durations (profiles/).
#include "common.h"

__global__ void spin_kernel(long ticks, long* out) {
  const long t0 = (long)__builtin_amdgcn_s_memrealtime();   // 100 MHz constant clock
  long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
    t = (long)__builtin_amdgcn_s_memrealtime();
  }
  if (out && threadIdx.x == 0) out[0] = t - t0;
}

extern "C" int msmd_spin_us(float us, long* ticks_out, msmd_stream_t stream) {
  if (!(us > 0.f) || us > 1.0e6f) return 1;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)(us * 100.0f), ticks_out);
  MSMD_RETURN_LAST();
}
