// MSMD_F16X2 ("split pair") storage conversions (common.h): fp32 <-> rows of 32-element [hi | lo] fp16 blocks.
// HBM-bound streaming kernels; the hot producers (LayerNorm, GEMM epilogues, conv0) write split rows themselves, so
// these run only where an fp32 tensor from elsewhere feeds a split GEMM.
#include "common.h"

// x (rows, cols) fp32 with leading dimension ldx -> y (rows, 2 * cols_out) fp16 split rows; columns >= cols are zero.
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, f16_t* __restrict__ y, long rows,
                                                    int cols, long ldx, int cols_out) {
  const int quads = cols_out >> 2;
  const long total = rows * quads;
  const bool vec = ((ldx & 3) == 0) && (((uintptr_t)x & 15) == 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / quads;
    const int c = (int)(i - r * quads) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xp = x + r * ldx + c;
    if (vec && c + 3 < cols) {
      const f32x4 t = *(const f32x4*)xp;
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < cols) v[e] = xp[e];
    }
    store4_split(y + r * 2 * cols_out, c, v);
  }
}

__global__ __launch_bounds__(256) void unsplit_kernel(const f16_t* __restrict__ x, float* __restrict__ y, long rows,
                                                      int cols, int cols_in, long ldy) {
  const int quads = (cols + 3) >> 2;
  const long total = rows * quads;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / quads;
    const int c = (int)(i - r * quads) * 4;
    float v[4];
    load4_split(x + r * 2 * cols_in, c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < cols) y[r * ldy + c + e] = v[e];
  }
}

extern "C" int msmd_split_f16x2(const float* x, void* y, long rows, int cols, long ldx, int cols_out,
                                msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || cols_out < cols || (cols_out & 31) || ldx < cols || !x || !y || ((uintptr_t)y & 15))
    return 1;
  const long total = rows * (cols_out >> 2);
  const unsigned grid = (unsigned)min((total + 255) / 256, (long)8192);
  hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (f16_t*)y, rows, cols, ldx,
                     cols_out);
  MSMD_RETURN_LAST();
}

extern "C" int msmd_unsplit_f16x2(const void* x, float* y, long rows, int cols, int cols_in, long ldy,
                                  msmd_stream_t stream) {
  if (rows <= 0 || cols <= 0 || cols_in < cols || (cols_in & 31) || ldy < cols || !x || !y || ((uintptr_t)x & 15))
    return 1;
  const long total = rows * ((cols + 3) >> 2);
  const unsigned grid = (unsigned)min((total + 255) / 256, (long)8192);
  hipLaunchKernelGGL(unsplit_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, y, rows, cols,
                     cols_in, ldy);
  MSMD_RETURN_LAST();
}
