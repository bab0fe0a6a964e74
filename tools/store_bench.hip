// Dev microbenchmark: how fast can 8 waves per CU write a (B, V, 3) fp32 tensor in the skinning kernel's tile order?
// Patterns: 0 = lane (vertex i, frames 4q+e): 4 dwordx3 per 16x16 tile (192-B runs); 1 = lane (frame i, vertices 4q..4q+3):
// 3 dwordx4 (48-B pieces); 2 = the workgroup's 16 rows x 1536 B as whole-row dwordx4 (what an LDS transposition would give);
// 3 = pattern 2 with nontemporal stores; 4 = pattern 0 nontemporal; 5 = pattern 0 with sc1 (write-through) stores; 6 = pattern 2 with sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

template <int P>
__global__ __launch_bounds__(512) void k(float* __restrict__ out, int B, int V, int fpb, int vtn) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 15, q = lane >> 4;
  const int vt8 = (vtn + 7) >> 3, L = blockIdx.x;
  const int v_tile = (L & 7) * vt8 + ((L >> 3) % vt8);
  if (v_tile >= vtn) return;
  const int f_begin = ((L >> 3) / vt8) * fpb;
  if (f_begin >= B) return;
  const int f_end = min(B, f_begin + fpb), ntiles = (f_end - f_begin + 15) >> 4;
  const int v0w = v_tile * 128 + wid * 16;
  float val = (float)tid;
  for (int t = 0; t < ntiles; ++t) {
    const int f0 = f_begin + 16 * t;
    val += 1.0f;
    if constexpr (P == 0 || P == 4 || P == 5) {
      const int ve = min(v0w + i, V - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = min(f0 + 4 * q + e, B - 1);
        F3 o{val, val + e, val - e};
        F3* d = (F3*)(out + ((long)f * V + ve) * 3);
        if constexpr (P == 4) { __builtin_nontemporal_store(o.x, &d->x); __builtin_nontemporal_store(o.y, &d->y); __builtin_nontemporal_store(o.z, &d->z); }
        else if constexpr (P == 5) {
          typedef float f32x3 __attribute__((ext_vector_type(3)));
          const f32x3 o3 = {o.x, o.y, o.z};
          asm volatile("global_store_dwordx3 %0, %1, off sc1" ::"v"(d), "v"(o3) : "memory");
        }
        else *d = o;
      }
    } else if constexpr (P == 1) {
      const int vb = min(v0w + 4 * q, V - 4), f = min(f0 + i, B - 1);
      f32x4u* d = (f32x4u*)(out + ((long)f * V + vb) * 3);
      d[0] = f32x4{val, val, val, val}; d[1] = f32x4{val, val + 1, val, val}; d[2] = f32x4{val, val, val + 2, val};
    } else {
      // rows r = 2 wid, 2 wid + 1 of the tile; 1536 B = 96 x 16 B each; 3 instructions cover 192 pieces
      const int vbase = min(v_tile * 128, V - 128);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int p = s * 64 + lane, r = 2 * wid + p / 96, c = p % 96;
        const int f = min(f0 + r, B - 1);
        f32x4u* d = (f32x4u*)(out + ((long)f * V + vbase) * 3) + c;
        const f32x4 o = f32x4{val, val + s, val, val};
        if constexpr (P == 3) __builtin_nontemporal_store(o, (f32x4*)d);
        else if constexpr (P == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(o) : "memory");
        else *d = o;
      }
    }
  }
}

// W adjacent 128-vertex slices per workgroup visit: 16 rows x (1536 W) B as whole-row dwordx4 (pattern 2 is W = 1); ALIGN = 1:
// every run starts on the 64-byte boundary at or below its first byte and ends on the one at or above its last (the extra bytes
// are the neighbours': only the rate matters here) -- what would the edges' partial lines cost?
template <int W, int ALIGN>
__global__ __launch_bounds__(512) void kw(float* __restrict__ out, int B, int V, int fpb, int vgn) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int vg8 = (vgn + 7) >> 3, L = blockIdx.x;
  const int v_grp = (L & 7) * vg8 + ((L >> 3) % vg8);
  if (v_grp >= vgn) return;
  const int f_begin = ((L >> 3) / vg8) * fpb;
  if (f_begin >= B) return;
  const int f_end = min(B, f_begin + fpb), ntiles = (f_end - f_begin + 15) >> 4;
  const int vbase = min(v_grp * 128 * W, V - 128 * W);
  float val = (float)tid;
  for (int t = 0; t < ntiles; ++t) {
    const int f0 = f_begin + 16 * t;
    val += 1.0f;
#pragma unroll
    for (int s = 0; s < 3 * W; ++s) {
      const int p = s * 64 + lane, r = 2 * wid + p / (96 * W), c = p % (96 * W);
      const int f = min(f0 + r, B - 1);
      char* d = (char*)(out + ((long)f * V + vbase) * 3) + c * 16;
      if (ALIGN) d = (char*)((unsigned long)d & ~63ul) + (c & 3) * 16;
      *(f32x4u*)d = f32x4{val, val + s, val, val};
    }
  }
}

template <int W, int ALIGN> float runw(float* out, int B, int V, int reps) {
  const int vg = (V + 128 * W - 1) / (128 * W);
  int splits = (1024 + vg - 1) / vg;
  int fpb = (((B + splits - 1) / splits) + 15) / 16 * 16;
  splits = (B + fpb - 1) / fpb;
  dim3 grid(((vg + 7) / 8) * 8 * splits);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((kw<W, ALIGN>), grid, dim3(512), 0, 0, out, B, V, fpb, vg);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((kw<W, ALIGN>), grid, dim3(512), 0, 0, out, B, V, fpb, vg);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

// plain linear stream of the same bytes (16 B per lane, grid-stride)
__global__ __launch_bounds__(512) void klin(f32x4* __restrict__ out, long n16) {
  for (long i = blockIdx.x * 512l + threadIdx.x; i < n16; i += gridDim.x * 512l) out[i] = f32x4{1.f, 2.f, 3.f, (float)i};
}

template <int P> float run(float* out, int B, int V, int reps) {
  const int vt = (V + 127) / 128;
  int splits = (1024 + vt - 1) / vt;
  int fpb = (((B + splits - 1) / splits) + 15) / 16 * 16;
  splits = (B + fpb - 1) / fpb;
  dim3 grid(((vt + 7) / 8) * 8 * splits);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<P>, grid, dim3(512), 0, 0, out, B, V, fpb, vt);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<P>, grid, dim3(512), 0, 0, out, B, V, fpb, vt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 25600, V = 5023;
  float* out; hipMalloc(&out, (size_t)B * V * 12 + 4096);
  const double gb = (double)B * V * 12 / 1e9;
  float t;
  t = run<0>(out, B, V, 10); printf("pattern 0 (4 x dwordx3, 192-B runs)        %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = run<1>(out, B, V, 10); printf("pattern 1 (3 x dwordx4, 48-B pieces)        %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = run<2>(out, B, V, 10); printf("pattern 2 (whole 1536-B rows, dwordx4)      %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = run<3>(out, B, V, 10); printf("pattern 3 (whole rows, nontemporal)         %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = run<4>(out, B, V, 10); printf("pattern 4 (dword x 3 nontemporal)           %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = run<5>(out, B, V, 10); printf("pattern 5 (pattern 0, sc1 write-through)    %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = run<6>(out, B, V, 10); printf("pattern 6 (whole rows, sc1 write-through)   %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = runw<1, 0>(out, B, V, 10); printf("W = 1 (1536-B runs)                          %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = runw<2, 0>(out, B, V, 10); printf("W = 2 (3072-B runs)                          %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = runw<4, 0>(out, B, V, 10); printf("W = 4 (6144-B runs)                          %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = runw<1, 1>(out, B, V, 10); printf("W = 1, 64-B aligned pieces                   %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  t = runw<4, 1>(out, B, V, 10); printf("W = 4, 64-B aligned pieces                   %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  {
    const long n16 = (long)B * V * 12 / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(klin, dim3(2048), dim3(512), 0, 0, (f32x4*)out, n16);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(klin, dim3(2048), dim3(512), 0, 0, (f32x4*)out, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); t = ms / 10;
    printf("linear stream, 16 B per lane                 %8.1f us  %7.0f GB/s\n", t * 1e3, gb / t * 1e3);
  }
  return 0;
}
