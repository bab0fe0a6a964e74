// Probe the lane/register layout of v_mfma_f32_16x16x4_f32 on the device (debug tool, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int la, float* out) {
  const int l = threadIdx.x;
  float a = (l == la) ? 1.0f : 0.0f;
  float b = 1.0f + l;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = acc[e];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  float h[256];
  for (int la = 0; la < 64; la += 1) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    if (la < 3 || la == 16 || la == 17 || la == 32 || la == 63) {
      printf("A one-hot lane %d:", la);
      int cnt = 0;
      for (int i = 0; i < 256; ++i) if (h[i] != 0) { if (cnt < 20) printf(" (lane %d reg %d)=Blane%d", i / 4, i % 4, (int)h[i] - 1); cnt++; }
      printf("  [%d nonzero]\n", cnt);
    }
  }
  return 0;
}
