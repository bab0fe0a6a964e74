// Probe (debug tool, not product code): numerics of v_mfma_f32_16x16x32_f16 on gfx950 for the f16x2 split scheme.
//  (1) are fp16 SUBNORMAL inputs honoured by the MFMA (CDNA2 flushed them)?
//  (2) error of one 16x16x32 f16 MFMA accumulation against an fp64 reference on random data
//  (3) error of a K = 768 dot product computed as  x ~ hi + lo / 2048  with three MFMAs per k-step
//      (hi.hi -> acc0; hi.lo + lo.hi -> acc1; result acc0 + acc1 / 2048) against fp64, next to the exact-fp32
//      v_mfma_f32_16x16x4_f32 chain and the plain 1-MFMA fp16 / bf16 products.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_split.hip -o tools/probe_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A (16 x K) row-major, B (16 x K) row-major (D = A . B^T), K % 32 == 0.  mode 0: f32 exact, 1: f16 plain,
// 2: f16x2 split (3 MFMA), 3: bf16 plain, 4: f16x2 split with ALL FOUR products (lo.lo kept)
__global__ void dot_kernel(const float* A, const float* B, float* D, int K, int mode) {
  const int l = threadIdx.x, fr = l & 15, fq = l >> 4;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[fr * K + k0 + 8 * fq + e]; b[e] = B[fr * K + k0 + 8 * fq + e]; }
    if (mode == 0) {
      // 16x16x4 f32: lane supplies k = fq for step s -> use k = k0 + 4 s + fq over 8 steps
      for (int s = 0; s < 8; ++s)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[fr * K + k0 + 4 * s + fq], B[fr * K + k0 + 4 * s + fq], acc0, 0, 0, 0);
    } else if (mode == 3) {
      bf16x8 ah, bh;
      for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)a[e]; bh[e] = (__bf16)b[e]; }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc0, 0, 0, 0);
    } else {
      f16x8 ah, al, bh, bl;
      for (int e = 0; e < 8; ++e) {
        ah[e] = (_Float16)a[e]; al[e] = (_Float16)((a[e] - (float)ah[e]) * 2048.0f);
        bh[e] = (_Float16)b[e]; bl[e] = (_Float16)((b[e] - (float)bh[e]) * 2048.0f);
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc0, 0, 0, 0);
      if (mode >= 2) {
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc1, 0, 0, 0);
      }
      if (mode == 4) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, acc2, 0, 0, 0);
    }
  }
  // D[i][j]: i = row of A = 4 fq + e ... (C/D map: col = lane & 15 -> B row, row = 4 (lane >> 4) + e -> A row)
  for (int e = 0; e < 4; ++e) {
    float v = acc0[e];
    if (mode == 2) v = acc0[e] + acc1[e] * (1.0f / 2048.0f);
    if (mode == 4) v = acc0[e] + (acc1[e] + acc2[e] * (1.0f / 2048.0f)) * (1.0f / 2048.0f);
    D[(4 * fq + e) * 16 + fr] = v;
  }
}

__global__ void subnormal_kernel(float* out) {
  const int l = threadIdx.x;
  f16x8 a, b;
  const _Float16 tiny = (_Float16)9.5367431640625e-07f;  // 2^-20: subnormal in fp16 (min normal 2^-14)
  for (int e = 0; e < 8; ++e) { a[e] = tiny; b[e] = (_Float16)1.0f; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (l == 0) out[0] = acc[0];   // expect 32 * 2^-20 = 3.0517578125e-05 if subnormals are honoured, 0 if flushed
  // bf16 subnormal: 2^-130
  bf16x8 c, d;
  for (int e = 0; e < 8; ++e) { c[e] = (__bf16)7.3468396926392969e-40f; d[e] = (__bf16)1.0f; }
  f32x4 acc2 = {0, 0, 0, 0};
  acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, d, acc2, 0, 0, 0);
  if (l == 0) out[1] = acc2[0];
}

static double urand() { return rand() / (double)RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

int main() {
  float* dout; hipMalloc(&dout, 16);
  hipLaunchKernelGGL(subnormal_kernel, dim3(1), dim3(64), 0, 0, dout);
  float ho[2]; hipMemcpy(ho, dout, 8, hipMemcpyDeviceToHost);
  printf("subnormal f16 input: got %.10e (expect 3.0517578125e-05 if honoured)\n", ho[0]);
  printf("subnormal bf16 input: got %.10e (expect 2.3509887e-38 if honoured)\n", ho[1]);
  srand(1234);
  for (int K : {32, 768, 3072}) {
    for (int dist = 0; dist < 3; ++dist) {
      std::vector<float> A(16 * K), B(16 * K);
      for (auto& v : A) v = dist == 0 ? (float)nrand() : dist == 1 ? (float)(nrand() * exp(3.0 * nrand())) : (float)(urand() + 0.5);
      for (auto& v : B) v = dist == 0 ? (float)(0.03 * nrand()) : dist == 1 ? (float)(0.03 * nrand() * exp(2.0 * nrand())) : (float)(urand() + 0.5);
      float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      printf("K=%d dist=%d (0 gauss, 1 heavy-tailed, 2 all-positive):", K, dist);
      for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode);
        float D[256]; hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
          double ref = 0, mag = 0;
          for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[j * K + k]; mag += fabs((double)A[i * K + k] * B[j * K + k]); }
          worst = fmax(worst, fabs(D[i * 16 + j] - ref) / mag);
        }
        printf("  m%d %.2e", mode, worst);
      }
      printf("   (max |err| / sum|a b|; m0 f32-exact, m1 f16, m2 f16x2-3mfma, m3 bf16, m4 f16x2-4mfma)\n");
      hipFree(dA); hipFree(dB); hipFree(dD);
    }
  }
  return 0;
}
