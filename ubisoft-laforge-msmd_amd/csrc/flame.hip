// FLAME vertex pass: blendshapes + pose correctives + skinning fused in ONE kernel, plus the tiny
// per-frame kinematics, landmarks and the dynamic-contour LUT row.
// Reference: utils/lbs.py:141-371, utils/flame.py:126-244.
//
// Skinning kernel (the HBM-relevant one): per frame it must write 5023*3 fp32 (60 276 B) and read 186
// coefficients; everything else (dirs 11.6 MB, weights, template) is frame-invariant and cache resident.
// The contraction v_posed = template + coef(186) . dirs(186 x 15069) runs on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32): a workgroup owns 64 vertices (16 per wave), keeps their three coordinate
// planes of `dirs` in REGISTERS (3 x 48 VGPRs per lane, loaded once) and streams frame tiles of 16
// through it, so the reference's materialised (B, V, 4, 4) transforms, W.expand and homogeneous
// coordinates (about 10x the algorithmic bytes) never exist.  The vertex sits on the MFMA lane, so the
// per-vertex blend T = sum_j w_j A_j and the 3x4 transform are lane-local; 16 lanes store 16 consecutive
// vertices (192 contiguous bytes) per frame.
#include "common.h"

__device__ __forceinline__ void rodrigues(const float* r, float* R) {
  // reference utils/lbs.py:285-300: angle = ||r + 1e-8||, dir = r / angle (un-shifted r)
  const float x = r[0] + 1e-8f, y = r[1] + 1e-8f, z = r[2] + 1e-8f;
  const float angle = sqrtf(x * x + y * y + z * z);
  const float rx = r[0] / angle, ry = r[1] / angle, rz = r[2] / angle;
  const float s = sinf(angle), c1 = 1.0f - cosf(angle);
  // K = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]];  R = I + s K + (1-c) K K
  const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float kk = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
      R[i * 3 + j] = (i == j ? 1.0f : 0.0f) + s * K[i * 3 + j] + c1 * kk;
    }
}

__global__ void rodrigues_kernel(const float* __restrict__ rv, float* __restrict__ R, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float r[3] = {rv[i * 3], rv[i * 3 + 1], rv[i * 3 + 2]}, o[9];
  rodrigues(r, o);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(long)i * 9 + k] = o[k];
}

extern "C" int msmd_batch_rodrigues(const float* rot_vecs, float* R, int N, msmd_stream_t stream) {
  if (N <= 0) return 1;
  hipLaunchKernelGGL(rodrigues_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rot_vecs, R, N);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Per-frame kinematics: 16 frames per 256-thread workgroup, 16 lanes per frame.  (The first version ran one 64-thread
// workgroup per frame with the joint chain serial in lane 0: latency bound, 110 us for 25 600 frames -- a sixth of the
// whole FLAME pass.)  The joint-regression table (9 KB) is staged in LDS once per workgroup and shared by its frames;
// the kinematic chain runs joint by joint (parents[i] < i) with 12 lanes computing the 3x4 entries of a transform.
#define LBS_MAXJ 5     // the launchers take J <= 5 (FLAME: 5 joints)
#define LBS_FPB 16
#define LBS_TILE_COEF_BYTES 12288   // 2 (hi, lo) x 192 K x 16 frames x 2 B
#define LBS_TILE_BYTES 18432        // + 12 components x 16 slots x 16 frames x 2 B of blend rows
__global__ __launch_bounds__(256) void lbs_prepare_kernel(const float* __restrict__ betas,
                                                          const float* __restrict__ pose,
                                                          const float* __restrict__ JS,
                                                          const int* __restrict__ parents, float* __restrict__ coef,
                                                          bf16_t* __restrict__ coef_hl, float* __restrict__ A,
                                                          float* __restrict__ joints_out, int NB, int J, int Kp,
                                                          int pose_is_matrix, f16_t* __restrict__ at_tiles, int B,
                                                          const float* __restrict__ betas2 = nullptr, int NB1 = 0,
                                                          const float* __restrict__ eye = nullptr, int pose_mode = 0,
                                                          int* __restrict__ shape_varies = nullptr, int kfold = 0) {
  // betas2 != NULL: the coefficient row is [betas (B, NB1) | betas2 (B, NB - NB1)] (FLAME's shape | expression, no
  // concatenated copy).  pose_mode 1 / 2: `pose` is FLAME's (B, 6) [global | jaw] axis-angle input, the neck is the
  // identity, `eye` (B, 6) or NULL = identity; 2 also ignores the global rotation (utils/flame.py:199-207).
  // LDS sized by the launcher for THIS model (lbs_prepare_lds): round 6 found the static arrays for the largest admissible one
  // (NB = 256, J = 8: 49.8 KB) held the launch to three workgroups per CU -- three rounds at 25 600 frames, 58 us of latency.
  // The joint-regression table is dead once the joints are known: the chain's transforms sT / sAo live in its space.
  extern __shared__ __attribute__((aligned(16))) float lbs_prep_smem[];
  const int nbp = NB + 4;
  float* sJS = lbs_prep_smem;                                          // (NB + 1) x 3 J, later sT | sAo
  const int region0 = max((NB + 1) * J * 3, 2 * LBS_FPB * LBS_MAXJ * 12);
  float (*sT)[LBS_MAXJ * 12] = (float (*)[LBS_MAXJ * 12])lbs_prep_smem;
  float (*sAo)[LBS_MAXJ * 12] = (float (*)[LBS_MAXJ * 12])(lbs_prep_smem + LBS_FPB * LBS_MAXJ * 12);
  float* sBf = lbs_prep_smem + region0;                                // [frame][NB + 4]
  float (*sJ)[16] = (float (*)[16])(sBf + LBS_FPB * nbp);
  float (*sR)[LBS_MAXJ * 9] = (float (*)[LBS_MAXJ * 9])(sBf + LBS_FPB * nbp + LBS_FPB * 16);
#define sB(f, k) sBf[(f) * nbp + (k)]
  const int tid = threadIdx.x, fl = tid >> 4, ln = tid & 15;
  const int b = blockIdx.x * LBS_FPB + fl;
  const bool valid = b < B;
  const int bb = valid ? b : B - 1;
  const int J3 = J * 3;
  for (int k = tid; k < (NB + 1) * J3; k += 256) sJS[k] = JS[k];
  if (betas2) {
    const float* b1 = betas + (long)bb * NB1;
    const float* b2 = betas2 + (long)bb * (NB - NB1);
    for (int k = ln; k < NB; k += 16) sB(fl, k) = k < NB1 ? b1[k] : b2[k - NB1];
    if (shape_varies) {   // do the first kfold shape coefficients of this frame differ from frame 0's?  (see lbs_shape_fold)
      bool diff = false;
      for (int k = ln; k < kfold; k += 16) diff |= b1[k] != betas[k];
      if (diff) atomicOr(shape_varies, 1);
    }
  } else {
    const float* be = betas + (long)bb * NB;
    for (int k = ln; k < NB; k += 16) sB(fl, k) = be[k];
  }
  if (ln < J) {
    float R[9];
    if (pose_mode) {
      float r[3] = {0.f, 0.f, 0.f};
      const float* src = nullptr;
      if (ln == 0 && pose_mode == 1) src = pose + (long)bb * 6;
      else if (ln == 2) src = pose + (long)bb * 6 + 3;
      else if (ln >= 3 && eye) src = eye + (long)bb * 6 + (ln - 3) * 3;
      if (src) { r[0] = src[0]; r[1] = src[1]; r[2] = src[2]; }
      rodrigues(r, R);
    } else if (pose_is_matrix) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = pose[((long)bb * J + ln) * 9 + k];
    } else {
      const float r[3] = {pose[((long)bb * J + ln) * 3], pose[((long)bb * J + ln) * 3 + 1], pose[((long)bb * J + ln) * 3 + 2]};
      rodrigues(r, R);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) sR[fl][ln * 9 + k] = R[k];
  }
  __syncthreads();
  // joints = J_regressor . (template + sum_l beta_l shapedirs_l): linear in the coefficients -> the JS table
  if (ln < J3) {
    float a0 = sJS[ln], a1 = 0.f;
    int l = 0;
    for (; l + 1 < NB; l += 2) {
      a0 = fmaf(sB(fl, l), sJS[(1 + l) * J3 + ln], a0);
      a1 = fmaf(sB(fl, l + 1), sJS[(2 + l) * J3 + ln], a1);
    }
    if (l < NB) a0 = fmaf(sB(fl, l), sJS[(1 + l) * J3 + ln], a0);
    sJ[fl][ln] = a0 + a1;
  }
  __syncthreads();
  // kinematic chain (utils/lbs.py:317-371): T_0 = [R_0 | J_0]; T_i = T_parent . [R_i | J_i - J_parent]; entry (r, c) = lane
  for (int i = 0; i < J; ++i) {
    if (ln < 12) {
      const int r = ln >> 2, c = ln & 3;
      const int pa = i == 0 ? -1 : parents[i];
      auto loc = [&](int rr) { return c < 3 ? sR[fl][i * 9 + rr * 3 + c] : sJ[fl][i * 3 + rr] - (pa >= 0 ? sJ[fl][pa * 3 + rr] : 0.f); };
      float v;
      if (pa < 0) {
        v = loc(r);
      } else {
        const float* Tp = &sT[fl][pa * 12 + r * 4];
        v = Tp[0] * loc(0) + Tp[1] * loc(1) + Tp[2] * loc(2);
        if (c == 3) v += Tp[3];
      }
      sT[fl][i * 12 + ln] = v;
    }
    __syncthreads();
  }
  // A_i = [R | t - R J_i] (relative to the rest pose); posed joints = translation column of T_i
  for (int e = ln; e < J * 12; e += 16) {
    const int i = e / 12, rc = e - i * 12, r = rc >> 2, c = rc & 3;
    const float* Tr = &sT[fl][i * 12 + r * 4];
    float v = Tr[c];
    if (c == 3) v -= Tr[0] * sJ[fl][i * 3] + Tr[1] * sJ[fl][i * 3 + 1] + Tr[2] * sJ[fl][i * 3 + 2];
    sAo[fl][e] = v;
    if (valid) {
      if (A) A[(long)b * J * 12 + e] = v;
      if (joints_out && c == 3) joints_out[((long)b * J + i) * 3 + r] = Tr[3];
    }
  }
  // coefficient row = [betas | pose_feature = (R[1:] - I) | 0 pad]
  // Skin tiles (msmd_lbs_skin_v2's input, one 18 KB record per 16 frames, read by contiguous LDS-DMA):
  //   bytes [0, 12288): coefficients, bf16 hi then lo, chunk-major [chunk = (hi|lo) * Kp/8 + k/8][frame % 16][k % 8]
  //   bytes [12288, 18432): blend rows, see below.
  // Frames beyond B - 1 of the last tile carry frame B - 1's values (the kernel clamps them to that frame and stores
  // identical bytes).
  unsigned char* tile = at_tiles ? (unsigned char*)at_tiles + (long)blockIdx.x * LBS_TILE_BYTES : nullptr;
  {
    float* crow = coef + (long)b * Kp;
    for (int k = ln; k < Kp; k += 16) {
      float v = 0.f;
      if (k < NB) v = sB(fl, k);
      else if (k < NB + (J - 1) * 9) {
        const int pf = k - NB, jj = 1 + pf / 9, rc = pf % 9;
        v = sR[fl][jj * 9 + rc] - ((rc == 0 || rc == 4 || rc == 8) ? 1.0f : 0.0f);
      }
      if (valid && coef) crow[k] = v;
      if (coef_hl || tile) {  // bf16 hi/lo split for the 3-product MFMA form: v ~= hi + lo with 16 significant bits
        asm volatile("" : "+v"(v));
        const bf16_t hi = (bf16_t)v, lo = (bf16_t)(v - (float)hi);
        if (coef_hl && valid) {
          coef_hl[((long)b * 2 + 0) * Kp + k] = hi;
          coef_hl[((long)b * 2 + 1) * Kp + k] = lo;
        }
        if (tile) {
          bf16_t* tc = (bf16_t*)tile;
          tc[((k >> 3) * 16 + fl) * 8 + (k & 7)] = hi;
          tc[((Kp / 8 + (k >> 3)) * 16 + fl) * 8 + (k & 7)] = lo;
        }
      }
    }
  }
  if (tile) {
    // Frame-side operand of the skinning kernel's blend MFMA (msmd_lbs_skin_v2): for component m of the 3x4 transforms,
    // 16 fp16 K slots per frame = [Ah_0..4 | Al_0..4 | Ah_0..4 | 0] (A = Ah + Al, unscaled: |A| = O(1), so the lo
    // halves keep 2^-25 absolute).  Layout [m][slot / 8][frame % 16][slot % 8]: one 16-byte chunk per
    // (m, slot octet, frame), frames of a tile contiguous -> conflict-free ds_read_b128.
    __syncthreads();
    f16_t* ta = (f16_t*)(tile + LBS_TILE_COEF_BYTES);
    for (int idx = ln; idx < 12 * 16; idx += 16) {
      const int m = idx >> 4, slot = idx & 15;
      float a = slot < 15 ? sAo[fl][(slot % 5) * 12 + m] : 0.f;
      asm volatile("" : "+v"(a));
      const f16_t ah = (f16_t)a, al = (f16_t)(a - (float)ah);
      const f16_t val = slot >= 15 ? (f16_t)0.f : ((slot >= 5 && slot < 10) ? al : ah);
      ta[((m * 2 + (slot >> 3)) * 16 + fl) * 8 + (slot & 7)] = val;
    }
  }
}
#undef sB
static size_t lbs_prepare_lds(int NB, int J) {
  const int region0 = max((NB + 1) * J * 3, 2 * LBS_FPB * LBS_MAXJ * 12);
  return sizeof(float) * (size_t)(region0 + LBS_FPB * (NB + 4) + LBS_FPB * 16 + LBS_FPB * LBS_MAXJ * 9);
}

extern "C" int msmd_lbs_prepare(const float* betas, const float* pose, const float* JS, const int* parents,
                                float* coef, void* coef_hl, float* A, float* joints, void* at_tiles, int B, int NB, int J,
                                int Kp, int pose_is_matrix, msmd_stream_t stream) {
  if (B <= 0 || NB <= 0 || NB > 256 || J <= 0 || J > 5 || Kp < NB + (J - 1) * 9 || (at_tiles && (J != 5 || Kp != 192))) return 1;
  hipLaunchKernelGGL(lbs_prepare_kernel, dim3((B + LBS_FPB - 1) / LBS_FPB), dim3(256), lbs_prepare_lds(NB, J), (hipStream_t)stream, betas, pose,
                     JS, parents, coef, (bf16_t*)coef_hl, A, joints, NB, J, Kp, pose_is_matrix, (f16_t*)at_tiles, B);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Fused blendshape + skinning.  KS = Kp / 4 MFMA steps; lane (q = l>>4, i = l&15) owns k = q*KS + s at
// step s for BOTH operands (a K permutation leaves the contraction unchanged).
template <int KS, int J>
__global__ __launch_bounds__(256) void lbs_skin_kernel(const float* __restrict__ coef, const float* __restrict__ A,
                                                       const float* __restrict__ tmpl, const float* __restrict__ dirs,
                                                       const float* __restrict__ wts, float* __restrict__ verts, int B,
                                                       int V, int Vp, int frames_per_block) {
  constexpr int Kp = KS * 4;
  __shared__ __attribute__((aligned(16))) float sA[16 * J * 12];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int v = blockIdx.x * 64 + wid * 16 + i;  // < Vp by construction
  const int f_begin = blockIdx.y * frames_per_block;
  const int f_end = min(B, f_begin + frames_per_block);

  // frame-invariant operands -> registers
  float d[3][KS];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int s = 0; s < KS; ++s) d[c][s] = dirs[((long)c * Kp + q * KS + s) * Vp + v];
  float t3[3], w[J];
#pragma unroll
  for (int c = 0; c < 3; ++c) t3[c] = tmpl[(long)c * Vp + v];
#pragma unroll
  for (int j = 0; j < J; ++j) w[j] = wts[(long)j * Vp + v];

  for (int f0 = f_begin; f0 < f_end; f0 += 16) {
    __syncthreads();
    for (int k = tid; k < 16 * J * 12; k += 256) {
      const int f = f0 + k / (J * 12);
      sA[k] = f < f_end ? A[(long)f * J * 12 + k % (J * 12)] : 0.f;
    }
    // A operand: row = frame f0 + i, k = q*KS + s  (16-B loads, 4 steps each)
    const int fa = min(f0 + i, B - 1);
    const float* crow = coef + (long)fa * Kp + q * KS;
    f32x4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      const f32x4 a = *(const f32x4*)(crow + s4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], d[c][s4 * 4 + e], acc[c], 0, 0, 0);
    }
    __syncthreads();
    // epilogue: lane = vertex v, frames f0 + 4q + e
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int fl = 4 * q + e, f = f0 + fl;
      float T[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const f32x4* ap = (const f32x4*)(sA + (fl * J + j) * 12);
        const f32x4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          T[k] = fmaf(w[j], a0[k], T[k]);
          T[4 + k] = fmaf(w[j], a1[k], T[4 + k]);
          T[8 + k] = fmaf(w[j], a2[k], T[8 + k]);
        }
      }
      const float px = t3[0] + acc[0][e], py = t3[1] + acc[1][e], pz = t3[2] + acc[2][e];
      if (f < f_end && v < V) {
        float* o = verts + ((long)f * V + v) * 3;
        o[0] = T[0] * px + T[1] * py + T[2] * pz + T[3];
        o[1] = T[4] * px + T[5] * py + T[6] * pz + T[7];
        o[2] = T[8] * px + T[9] * py + T[10] * pz + T[11];
      }
    }
  }
}

// FLAME.forward's inputs straight into the skinning kernel's tile records: shape (B, NS) | expression (B, NE) as the
// coefficient row, pose6 (B, 6) = [global | jaw] axis-angle with the identity neck, eye (B, 6) or NULL (identity eyes):
// no concatenated betas / full_pose tensors.  coef / A / joints are optional outputs.
// One subject, many frames (the usual call: every frame carries the same shape row, or zeros): the first LBS_KFOLD shape
// directions can be added to the template ONCE, v_folded = template + sum_{k < LBS_KFOLD} shape_0[k] dirs[k], and the
// skinning kernel then skips those K groups (27 of its 54 blendshape MFMAs per tile).  Whether the frames really share
// the row is decided on the device (shape_varies, set by the kinematics kernel), so nothing reads back to the host and
// the general path stays one uniform branch away.
#define LBS_KFOLD 96
__global__ __launch_bounds__(256) void lbs_shape_fold_kernel(const float* __restrict__ shape0, const float* __restrict__ dirs,
                                                             const float* __restrict__ tmpl, float* __restrict__ out,
                                                             int Vp, int Kp, int kfold) {
  // 64 vertices x 4 K slices per workgroup (the first version walked all 96 directions serially in 60 workgroups: 24 us)
  __shared__ float part[4][64];
  const int vl = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int v = blockIdx.x * 64 + vl, c = blockIdx.y;
  const int kn = kfold / 4, k0 = ks * kn;
  float a = 0.f;
  if (v < Vp) {
    const float* d = dirs + ((long)c * Kp + k0) * Vp + v;
#pragma unroll 8
    for (int k = 0; k < kn; ++k) a = fmaf(shape0[k0 + k], d[(long)k * Vp], a);
  }
  part[ks][vl] = a;
  __syncthreads();
  if (ks == 0 && v < Vp) out[(long)c * Vp + v] = tmpl[(long)c * Vp + v] + ((part[0][vl] + part[1][vl]) + (part[2][vl] + part[3][vl]));
}

extern "C" int msmd_flame_prepare(const float* shape, const float* expr, const float* pose6, const float* eye,
                                  const float* JS, const int* parents, float* coef, float* A, float* joints,
                                  void* skin_tiles, int B, int NS, int NE, int ignore_global_rot, int* shape_varies,
                                  float* v_template_folded, const float* dirs, const float* v_template, int Vp,
                                  msmd_stream_t stream) {
  const int NB = NS + NE, J = 5, Kp = 192;
  if (B <= 0 || NS <= 0 || NE <= 0 || NB > 256 || Kp < NB + (J - 1) * 9 || !shape || !expr || !pose6 || !skin_tiles) return 1;
  hipStream_t st = (hipStream_t)stream;
  const bool fold = shape_varies && v_template_folded && dirs && v_template && NS >= LBS_KFOLD;
  if (fold) {
    hipError_t e = msmd_zero_async(shape_varies, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(lbs_prepare_kernel, dim3((B + LBS_FPB - 1) / LBS_FPB), dim3(256), lbs_prepare_lds(NB, J), st, shape, pose6,
                     JS, parents, coef, (bf16_t*)nullptr, A, joints, NB, J, Kp, 0, (f16_t*)skin_tiles, B, expr, NS, eye,
                     ignore_global_rot ? 2 : 1, fold ? shape_varies : (int*)nullptr, LBS_KFOLD);
  if (fold)
    hipLaunchKernelGGL(lbs_shape_fold_kernel, dim3((Vp + 63) / 64, 3), dim3(256), 0, st, shape, dirs, v_template,
                       v_template_folded, Vp, Kp, LBS_KFOLD);
  MSMD_RETURN_LAST();
}

extern "C" int msmd_lbs_skin(const float* coef, const float* A, const float* v_template, const float* dirs,
                             const float* lbs_weights, float* verts, int B, int J, int V, int Vp, int Kp,
                             msmd_stream_t stream) {
  if (B <= 0 || V <= 0 || Vp < V || (Vp & 63) || J != 5 || Kp != 192) return 1;  // FLAME2020 geometry
  // enough frame splits to fill 256 CUs, at least 64 frames per workgroup to amortise the dirs load
  const int vt = Vp / 64;
  int splits = max(1, min((B + 63) / 64, (2048 + vt - 1) / vt));
  int fpb = (((B + splits - 1) / splits) + 15) / 16 * 16;
  splits = (B + fpb - 1) / fpb;
  dim3 grid(vt, splits), block(256);
  hipLaunchKernelGGL((lbs_skin_kernel<48, 5>), grid, block, 0, (hipStream_t)stream, coef, A, v_template, dirs,
                     lbs_weights, verts, B, V, Vp, fpb);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Fused blendshape + skinning, split-bf16 form: coef = c_hi + c_lo, dirs = d_hi + d_lo (bf16 each), and
//   coef . dirs ~= c_hi.d_hi + c_hi.d_lo + c_lo.d_hi      (the dropped lo.lo term is 2^-16 relative)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 54 MFMAs x 16 cycles per 16x16 tile instead of 144 x 32 on
// the fp32 MFMA, which moves the kernel from matrix-bound to the epilogue / HBM-write side.  Same register
// residency of dirs (144 VGPRs), same lane-local epilogue.  dirs_hl: (2, 3, Kp/8, Vp, 8) bf16.
template <int KG, int J>  // KG = Kp / 32 MFMA K groups
__global__ __launch_bounds__(256) void lbs_skin_bf16x3_kernel(const bf16_t* __restrict__ coef_hl,
                                                              const float* __restrict__ A,
                                                              const float* __restrict__ tmpl,
                                                              const bf16_t* __restrict__ dirs_hl,
                                                              const float* __restrict__ wts, float* __restrict__ verts,
                                                              int B, int V, int Vp, int frames_per_block) {
  constexpr int Kp = KG * 32;
  constexpr int NCH = 2 * Kp / 8;               // 16-B chunks per frame row of coef_hl (hi then lo)
  constexpr int COEF_BYTES = NCH * 16 * 16;     // one 16-frame tile, chunk-major [chunk][frame] -> conflict-free reads
  constexpr int A_BYTES = 16 * J * 12 * 4;      // 16 frames x J x 3x4 fp32
  constexpr int A_CHUNKS = A_BYTES / 16;
  constexpr int STAGE = COEF_BYTES + ((A_BYTES + 1023) / 1024) * 1024;
  // Two stages, filled by LDS-DMA one tile ahead of the MFMAs (no register staging, one barrier per tile).
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  // 1-D grid, vertex tile = (L % 8) + 8 * ((L / 8) % ceil(vt / 8)): every frame split of one 64-vertex slice runs on
  // the SAME XCD (workgroups go to XCDs round-robin), so the slice's 147 KB of blendshape directions is fetched into one
  // L2 once and re-read from there by the other splits instead of bouncing through all eight
  const int vt8 = (Vp / 64 + 7) >> 3;
  const int L = blockIdx.x;
  const int v_tile = (L & 7) + 8 * ((L >> 3) % vt8);
  if (v_tile * 64 >= Vp) return;
  const int v = v_tile * 64 + wid * 16 + i;
  const int f_begin = ((L >> 3) / vt8) * frames_per_block;
  const int f_end = min(B, f_begin + frames_per_block);

  u32x4 dh[3][KG], dl[3][KG];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const long off = (((long)c * (Kp / 8) + 4 * g + q) * Vp + v) * 8;
      dh[c][g] = *(const u32x4*)(dirs_hl + off);
      dl[c][g] = *(const u32x4*)(dirs_hl + (long)3 * (Kp / 8) * Vp * 8 + off);
    }
  float t3[3], w[J];
#pragma unroll
  for (int c = 0; c < 3; ++c) t3[c] = tmpl[(long)c * Vp + v];
#pragma unroll
  for (int j = 0; j < J; ++j) w[j] = wts[(long)j * Vp + v];

  typedef __attribute__((address_space(3))) void lds_t;
  typedef __attribute__((address_space(1))) const void gbl_t;
  auto issue = [&](int f0, int stage) {
    unsigned char* base = smem + stage * STAGE;
    // coefficient tile: LDS slot p = chunk*16 + frame  <-  global (frame f0 + p%16, chunk p/16)
#pragma unroll
    for (int k = 0; k < NCH * 16 / 256; ++k) {
      const int p = (k * 4 + wid) * 64 + lane;
      const int fr = min(f0 + (p & 15), B - 1), ch = p >> 4;
      __builtin_amdgcn_global_load_lds((gbl_t*)(coef_hl + (long)fr * 2 * Kp + ch * 8),
                                       (lds_t*)(base + (k * 4 + wid) * 1024), 16, 0, 0);
    }
    // rigid transforms of the 16 frames: contiguous in global memory
    {
      const int p = wid * 64 + lane;
      const long gofs = min((long)f0 * J * 12 + (long)p * 4, (long)B * J * 12 - 4);
      if (wid * 64 < A_CHUNKS)
        __builtin_amdgcn_global_load_lds((gbl_t*)(A + gofs), (lds_t*)(base + COEF_BYTES + wid * 1024), 16, 0, 0);
    }
  };

  int stage = 0;
  issue(f_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: nothing younger than the loads yet
  // a wave issues exactly 4 vertex-store instructions per tile unless ALL its vertices are padding
  const bool wave_stores = __builtin_amdgcn_readfirstlane(v_tile * 64 + wid * 16) < V;
  for (int f0 = f_begin; f0 < f_end; f0 += 16) {
    // vmcnt counts stores too: waiting for 0 would serialise on the previous tile's vertex stores (the 4 youngest
    // VMEM ops of this wave).  Leave them in flight and wait only for this tile's 4 LDS-DMA loads, which are older.
    if (wave_stores) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (f0 + 16 < f_end) issue(f0 + 16, stage ^ 1);
    const unsigned char* sc = smem + stage * STAGE;
    const float* sA = (const float*)(sc + COEF_BYTES);
    f32x4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(sc + ((4 * g + q) * 16 + i) * 16));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(sc + ((Kp / 8 + 4 * g + q) * 16 + i) * 16));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, dh[c][g]), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, dl[c][g]), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, dh[c][g]), acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int fl = 4 * q + e, f = f0 + fl;
      // packed fp32 maths (v_pk_fma_f32): the epilogue is VALU-issue bound, so halve its instruction count
      f32x2 T2[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) T2[k] = f32x2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const f32x4* ap = (const f32x4*)(sA + (fl * J + j) * 12);
        const f32x4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        const f32x2 wj = f32x2{w[j], w[j]};
        T2[0] = __builtin_elementwise_fma(wj, f32x2{a0[0], a0[1]}, T2[0]);
        T2[1] = __builtin_elementwise_fma(wj, f32x2{a0[2], a0[3]}, T2[1]);
        T2[2] = __builtin_elementwise_fma(wj, f32x2{a1[0], a1[1]}, T2[2]);
        T2[3] = __builtin_elementwise_fma(wj, f32x2{a1[2], a1[3]}, T2[3]);
        T2[4] = __builtin_elementwise_fma(wj, f32x2{a2[0], a2[1]}, T2[4]);
        T2[5] = __builtin_elementwise_fma(wj, f32x2{a2[2], a2[3]}, T2[5]);
      }
      const f32x2 pxy = f32x2{t3[0] + acc[0][e], t3[1] + acc[1][e]};
      const f32x2 pz1 = f32x2{t3[2] + acc[2][e], 1.0f};
      if (f < f_end && v < V) {
        struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
        F3 o;
        const f32x2 s0 = __builtin_elementwise_fma(T2[0], pxy, T2[1] * pz1);
        const f32x2 s1 = __builtin_elementwise_fma(T2[2], pxy, T2[3] * pz1);
        const f32x2 s2 = __builtin_elementwise_fma(T2[4], pxy, T2[5] * pz1);
        o.x = s0[0] + s0[1];
        o.y = s1[0] + s1[1];
        o.z = s2[0] + s2[1];
        *(F3*)(verts + ((long)f * V + v) * 3) = o;  // one 12-byte store per lane: 16 lanes = 192 contiguous bytes
      }
    }
    stage ^= 1;
  }
}

extern "C" int msmd_lbs_skin_bf16x3(const void* coef_hl, const float* A, const float* v_template, const void* dirs_hl,
                                    const float* lbs_weights, float* verts, int B, int J, int V, int Vp, int Kp,
                                    msmd_stream_t stream) {
  if (B <= 0 || V <= 0 || Vp < V || (Vp & 63) || J != 5 || Kp != 192) return 1;
  const int vt = Vp / 64;
  int splits = max(1, min((B + 63) / 64, (2048 + vt - 1) / vt));
  int fpb = (((B + splits - 1) / splits) + 15) / 16 * 16;
  splits = (B + fpb - 1) / fpb;
  dim3 grid(((vt + 7) / 8) * 8 * splits), block(256);
  hipLaunchKernelGGL((lbs_skin_bf16x3_kernel<6, 5>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)coef_hl, A,
                     v_template, (const bf16_t*)dirs_hl, lbs_weights, verts, B, V, Vp, fpb);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Skinning, second form (msmd_lbs_skin_v2): the per-(vertex, frame) blend  T = sum_j w_j A_j  moves from the vector
// ALU (60 FMAs + 60 broadcast LDS reads per lane and frame: the epilogue that bounded lbs_skin_bf16x3 at ~20 % of HBM)
// onto the matrix pipe.  For component m of the 3x4 transforms, T_m(frame, vertex) = sum_j A_j[m](frame) w_j(vertex) is
// a contraction over the FIVE joints: with both sides split into fp16 hi + lo it is 15 K slots of ONE
// v_mfma_f32_16x16x32_f16 (frame-side rows [Ah | Al | Ah] written by lbs_prepare, vertex-side column [wh | wh | wl] held
// in one VGPR quad), whose accumulator lands exactly where the blendshape product puts p(vertex, frame): vertex on the
// lane, frames 4q + e in the registers.  12 MFMAs replace ~240 vector instructions per 16 x 16 tile; what is left on
// the vector ALU is out_c = T_c0 px + T_c1 py + T_c2 pz + T_c3 (36 FMAs per lane and tile).
// Workgroup = NWV waves = 16 NWV vertices (their `dirs` planes in registers, 144 VGPRs, as before); frame tiles of 16
// arrive by LDS-DMA through an NS-deep ring (coefficients 12 KB + blend rows 6 KB per tile = 18 one-KiB pieces dealt
// round-robin to the waves) behind counted vmcnt waits that leave younger tiles' loads AND the vertex stores in flight.
// Stores: 12 bytes (x, y, z of the lane's vertex) per lane and frame, 16 lanes = 192 contiguous bytes.  (Turning each
// wave's 16 x 48 floats through an LDS patch into plain dword stores of whole row slices was built and measured: the
// 28 extra LDS instructions per lane and tile cost more than the narrower stores saved, 765 vs 695 us at 25 600 frames.)
// No lane is ever masked: padding vertices (v >= V) and padding frames (f >= B) recompute and re-store vertex V - 1 /
// frame B - 1 (identical bytes), so every wave issues exactly 4 store instructions per tile and the vmcnt arithmetic
// is uniform.
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void vm_wait_n(int n) {   // n = (pieces per load group) a + (stores per tile) b, see the kernel
#define VMW(k) case k: vm_wait<k>(); break;
  switch (n) {
    VMW(1) VMW(2) VMW(3) VMW(4) VMW(5) VMW(6) VMW(7) VMW(8) VMW(9) VMW(10) VMW(11) VMW(12) VMW(13) VMW(14) VMW(15) VMW(16)
    VMW(17) VMW(18) VMW(19) VMW(20) VMW(21) VMW(22) VMW(23) VMW(24) VMW(25) VMW(26) VMW(27) VMW(28) VMW(29) VMW(30) VMW(31)
    VMW(32) VMW(33) VMW(34) VMW(35) VMW(36) VMW(37) VMW(38) VMW(39) VMW(40)
    default: vm_wait<0>(); break;
  }
#undef VMW
}

// NWV waves per workgroup (16 vertices each).  ABL (developer ablations, tuning key 8; 0 in the product): 1 = no vertex
// stores, 2 = no blend MFMAs, 4 = no blendshape MFMAs, 8 = no LDS-DMA after the prologue (stale tiles) -- outputs are
// then wrong by construction.
// WP: also store the un-skinned vertices p = template + coef . dirs (B, V, 3) -- what the backward of the skinning needs
// (training through the vertex-space loss); 8 instead of 4 store instructions per tile and wave.
// (Whole-row stores through an LDS image of the tile -- every wave puts its 16 x 16 x 3 results into [16 frames][128 vertices
// x 3] and, one barrier later, the waves write the image out as three wave-wide 16-byte stores of whole 1 536-byte rows, the
// shape that reaches 3.80 instead of 2.99 TB/s as a PURE store stream -- were built again in round 5 with the image double
// buffered on the tile loop's own barrier: 0.703 ms against 0.592 at 25 600 frames, same box, alternating.  The image costs
// 12 LDS writes + 3 reads per lane and tile and ties the waves to one barrier per tile, i.e. it removes the drift between the
// waves that the two-tiles-per-barrier schedule below lives on; the store shape is not what is left to win here.)
// OUT16 (msmd_lbs_skin_v2_f16, round 6): fp16 vertices in rows of V_ld (even, >= V) vertices -- the kernel is bound by its store
// stream, this halves it.  Lanes 2 p and 2 p + 1 (vertices v, v + 1) trade two frames each by DPP so that every lane stores the
// 12 contiguous bytes [x y z x y z] of its vertex PAIR for two of the tile's four frames: 2 store instructions of 96-byte runs
// per 8 lanes instead of 4 of 192-byte runs per 16.  (V = 5023 is odd: a row of V * 3 halves would put every other frame on a
// 2-byte boundary; hence the padded row.  Slot V of a row receives a copy of vertex V - 1.)
// F16P (with OUT16; msmd_lbs_skin_v2_f16's default form): fp16 vertices do not need 16-bit-split operands -- the blendshape
// product runs on ONE fp16 plane of `dirs` (72 registers instead of 144) and ONE fp16 coefficient plane (12 KB tile records
// [coef fp16 6 KB | blend rows 6 KB] written by msmd_lbs_tiles_f16): 1 MFMA per K group and coordinate instead of 3.  The
// operands' own rounding (2^-12 relative on offsets of <= ~0.03) stays under the fp16 step of the stored vertex.
template <int KG, int NS, int NWV, int ABL = 0, bool WP = false, int RB = 0, bool OUT16 = false, bool F16P = false>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(2, 2)))
void lbs_skin_v2_kernel(const unsigned char* __restrict__ tiles,
                        const float* __restrict__ tmpl, const bf16_t* __restrict__ dirs_hl,
                        const float* __restrict__ wts, float* __restrict__ verts, int B, int V, int Vp,
                        int frames_per_block, int vtn, float* __restrict__ vposed = nullptr, int xcd_adj = 0,
                        const int* __restrict__ shape_varies = nullptr, const float* __restrict__ tmpl_folded = nullptr,
                        int V_ld = 0) {
  static_assert(!(OUT16 && WP), "the training form stores fp32");
  constexpr int SPT = WP ? 8 : (OUT16 ? 2 : 4);   // store instructions per tile and wave
  // all frames share their first LBS_KFOLD shape coefficients (msmd_flame_prepare): folded template, skip those K groups
  const bool uni = shape_varies != nullptr && tmpl_folded != nullptr && *shape_varies == 0;
  const int g_first = uni ? LBS_KFOLD / 32 : 0;
  if (uni) tmpl = tmpl_folded;
  static_assert(!F16P || OUT16, "the single-plane form is for fp16 vertices");
  constexpr int Kp = KG * 32, J = 5, VPB = 16 * NWV;
  constexpr int NCH = (F16P ? 1 : 2) * Kp / 8;   // 16-byte chunks per frame of the coefficients (hi then lo; F16P: one fp16 plane)
  constexpr int COEF_BYTES = NCH * 256;      // [chunk][frame] 16 B
  constexpr int AT_BYTES = 12 * 2 * 256;     // [m][slot octet][frame] 16 B
  constexpr int STAGE = COEF_BYTES + AT_BYTES;
  constexpr int NP = STAGE / 1024;           // 18 (F16P: 12) one-KiB LDS-DMA pieces per tile
  static_assert(COEF_BYTES == (F16P ? 6 : 12) * 1024 && AT_BYTES == 6 * 1024, "piece map below");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int vt8 = (vtn + 7) >> 3;
  const int L = blockIdx.x;
  // workgroups go to the XCDs round-robin (L & 7).  Every frame split of a vertex slice runs on ONE XCD (its dirs stay
  // in that L2); each XCD owns vt8 ADJACENT slices and walks them fastest, so the 128-byte lines two neighbouring slices
  // share in a frame row meet in one L2 (554 vs 588 us at 25 600 frames against dealing the slices one by one).
  const int v_tile = xcd_adj ? (L & 7) * vt8 + ((L >> 3) % vt8) : (L & 7) + 8 * ((L >> 3) % vt8);
  if (v_tile >= vtn) return;
  const int f_begin = ((L >> 3) / vt8) * frames_per_block;
  if (f_begin >= B) return;
  const int f_end = min(B, f_begin + frames_per_block);
  const int ntiles = (f_end - f_begin + 15) >> 4;
  const int ve = min(v_tile * VPB + wid * 16 + i, V - 1);

  u32x4 dh[3][KG], dl[3][F16P ? 1 : KG];     // F16P: dirs_hl points at the ONE fp16 plane (3, Kp / 8, Vp, 8)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const long off = (((long)c * (Kp / 8) + 4 * g + q) * Vp + ve) * 8;
      dh[c][g] = *(const u32x4*)(dirs_hl + off);
      if constexpr (!F16P) dl[c][g] = *(const u32x4*)(dirs_hl + (long)3 * (Kp / 8) * Vp * 8 + off);
    }
  float t3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) t3[c] = tmpl[(long)c * Vp + ve];
  // vertex-side operand of the blend MFMA: K slots [wh_0..4 | wh_0..4 | wl_0..4 | 0 ...]; lane (i, q) holds slots 8q .. 8q+7
  u32x4 wB;
  {
    f16_t wh[J], wl[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      float w = wts[(long)j * Vp + ve];
      asm volatile("" : "+v"(w));
      wh[j] = (f16_t)w;
      wl[j] = (f16_t)(w - (float)wh[j]);
    }
    const f16_t z = (f16_t)0.f;
    const f16x8 s0 = f16x8{wh[0], wh[1], wh[2], wh[3], wh[4], wh[0], wh[1], wh[2]};
    const f16x8 s1 = f16x8{wh[3], wh[4], wl[0], wl[1], wl[2], wl[3], wl[4], z};
    const f16x8 zz = f16x8{z, z, z, z, z, z, z, z};
    wB = __builtin_bit_cast(u32x4, q == 0 ? s0 : (q == 1 ? s1 : zz));
  }

  typedef __attribute__((address_space(3))) void lds_t;
  typedef __attribute__((address_space(1))) const void gbl_t;
  // piece k of a tile -> wave k % NWV; a wave stages pieces wid, wid + NWV, ... (my_np of them: the vmcnt unit)
  const int my_np = (NP - wid + NWV - 1) / NWV;
  auto issue = [&](int t) {   // one 18 KB tile record = 18 contiguous one-KiB LDS-DMA pieces
    unsigned char* base = smem + (t % NS) * STAGE;
    const unsigned char* src = tiles + (long)((f_begin >> 4) + t) * STAGE + lane * 16;
#pragma unroll
    for (int r = 0; r < (NP + NWV - 1) / NWV; ++r) {
      const int k = wid + NWV * r;
      if (k < NP) __builtin_amdgcn_global_load_lds((gbl_t*)(src + k * 1024), (lds_t*)(base + k * 1024), 16, 0, 0);
    }
  };

  auto do_tile = [&](int t) {
    const unsigned char* sc = smem + (t % NS) * STAGE;
    const unsigned char* sat = sc + COEF_BYTES;
    const int f0 = f_begin + 16 * t;

    f32x4 accp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) accp[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      if (g < g_first) continue;   // wave-uniform
      if constexpr (F16P) {
        const f16x8 a16 = __builtin_bit_cast(f16x8, *(const u32x4*)(sc + ((4 * g + q) * 16 + i) * 16));
#pragma unroll
        for (int c = 0; c < 3; ++c)
          accp[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, __builtin_bit_cast(f16x8, dh[c][g]), accp[c], 0, 0, 0);
        continue;
      }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(sc + ((4 * g + q) * 16 + i) * 16));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(sc + (((F16P ? 0 : Kp / 8) + 4 * g + q) * 16 + i) * 16));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if constexpr (ABL & 4) {
          if (g == 0) accp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, dh[c][g]), accp[c], 0, 0, 0);
          asm volatile("" :: "v"(dh[c][g]), "v"(dl[c][F16P ? 0 : g]), "v"(ah), "v"(al));
        } else {
          accp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, dh[c][g]), accp[c], 0, 0, 0);
          accp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, dl[c][F16P ? 0 : g]), accp[c], 0, 0, 0);
          accp[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, dh[c][g]), accp[c], 0, 0, 0);
        }
      }
    }
    float px[4], py[4], pz[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { px[e] = t3[0] + accp[0][e]; py[e] = t3[1] + accp[1][e]; pz[e] = t3[2] + accp[2][e]; }
    float out[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 T[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = 4 * c + k;
        // K slots 16 .. 31 are zero on the vertex side (wB), so lanes q >= 2 may feed any finite values: they re-read
        // octet q & 1 (no divergent branch, no select)
        const u32x4 fa = *(const u32x4*)(sat + ((m * 2 + (q & 1)) * 16 + i) * 16);
        if constexpr (ABL & 2) {
          T[k] = __builtin_bit_cast(f32x4, fa);
          asm volatile("" :: "v"(wB));
        } else {
          T[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, wB),
                                                        f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[c][e] = fmaf(T[0][e], px[e], fmaf(T[1][e], py[e], fmaf(T[2][e], pz[e], T[3][e])));
    }
    if constexpr (OUT16) {
      // even lane keeps frames e = 0, 1 and gives 2, 3; odd lane the other way round; lower vertex of the pair = the even lane's
      const bool oddl = i & 1;
      auto swap1 = [](float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, false)); };
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float own[3], got[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          own[c] = oddl ? out[c][2 + s2] : out[c][s2];
          got[c] = swap1(oddl ? out[c][s2] : out[c][2 + s2]);
        }
        const float lo0 = oddl ? got[0] : own[0], lo1 = oddl ? got[1] : own[1], lo2 = oddl ? got[2] : own[2];
        const float up0 = oddl ? own[0] : got[0], up1 = oddl ? own[1] : got[1], up2 = oddl ? own[2] : got[2];
        struct __attribute__((aligned(4))) U3 { unsigned a, b, c; };
        U3 o;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o.a) : "v"(lo0), "v"(lo1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o.b) : "v"(lo2), "v"(up0));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o.c) : "v"(up1), "v"(up2));
        const int f = min(f0 + 4 * q + (oddl ? 2 : 0) + s2, B - 1);
        if constexpr (ABL & 1) asm volatile("" :: "v"(o.a), "v"(o.b), "v"(o.c), "v"(f));
        else *(U3*)((f16_t*)verts + ((long)f * V_ld + (ve & ~1)) * 3) = o;   // 8 lanes = 96 contiguous bytes per frame
      }
      return;
    }
    struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = min(f0 + 4 * q + e, B - 1);
      F3 o;
      o.x = out[0][e]; o.y = out[1][e]; o.z = out[2][e];
      if constexpr (ABL & 1) asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(f));
      else *(F3*)(verts + ((long)f * V + ve) * 3) = o;   // 16 lanes = 192 contiguous bytes per frame
      if constexpr (WP) {
        F3 pp;
        pp.x = px[e]; pp.y = py[e]; pp.z = pz[e];
        *(F3*)(vposed + ((long)f * V + ve) * 3) = pp;
      }
    }
  };

  if constexpr (RB == 0) {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < ntiles) issue(s);
    for (int t = 0; t < ntiles; ++t) {
      // ops younger than tile t's loads, in issue order: min(NS-2, ntiles-1-t) load groups (my_np each) and
      // min(NS-1, t) store groups (SPT each)
      vm_wait_n((ABL & 9) ? 0 : my_np * min(NS - 2, ntiles - 1 - t) + SPT * min(NS - 1, t));
      __builtin_amdgcn_s_barrier();
      if (t + NS - 1 < ntiles && !(ABL & 8)) issue(t + NS - 1);
      do_tile(t);
    }
  } else {
    // One workgroup barrier per RB tiles (NS = 2 RB slots: the group being read and the group in flight): between
    // barriers the waves drift apart, so one wave's vector / store phase can run under its SIMD partner's MFMA phase
    // instead of all eight waves meeting the matrix pipe, then the store path, in lockstep; s_setprio on half the waves
    // keeps the pairs apart.  770 -> 690 us at 25 600 frames.
    static_assert(RB == 0 || NS == 2 * RB, "ring = two barrier groups");
    const int ngroups = (ntiles + RB - 1) / RB;
    if (wid >= NWV / 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (r < ntiles) issue(r);
    for (int g = 0; g < ngroups; ++g) {
      // younger than group g's loads: the stores of group g-1's RB tiles
      if (g == 0) vm_wait<0>(); else vm_wait_n(SPT * RB);
      __builtin_amdgcn_s_barrier();
      if (g + 1 < ngroups) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
          if ((g + 1) * RB + r < ntiles) issue((g + 1) * RB + r);
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int t = g * RB + r;
        if (t < ntiles) do_tile(t);
      }
    }
  }
}

static int lbs_skin_v2_impl(const void* skin_tiles, const float* v_template, const void* dirs_hl,
                            const float* lbs_weights, float* verts, float* vposed, int B, int J, int V, int Vp, int Kp,
                            msmd_stream_t stream, const int* shape_varies = nullptr, const float* tmpl_folded = nullptr,
                            int V_ld16 = 0, bool single_plane = false) {
  if (B <= 0 || V <= 0 || Vp < V || J != 5 || Kp != 192 || !skin_tiles) return 1;
  if (V_ld16 && (V_ld16 < V || (V_ld16 & 1) || vposed || ((uintptr_t)verts & 3))) return 1;
  // tuning key 9: 1 = two 4-wave workgroups per CU (64 vertices each, 3-stage rings) instead of one 8-wave workgroup
  // (128 vertices, 4-stage ring).  Measured at 25 600 frames: 783 vs 732 us -- the smaller workgroups double the
  // staged bytes per vertex and their phase drift buys less than that costs.
  const bool big = MSMD_TUNE(9) != 1;
  const int vpb = big ? 128 : 64;
  const int vt = (V + vpb - 1) / vpb;
  int splits = max(1, min((B + 63) / 64, ((big ? 1024 : 2048) + vt - 1) / vt));   // >= 64 frames per workgroup amortise the dirs load
  int fpb = (((B + splits - 1) / splits) + 15) / 16 * 16;
  splits = (B + fpb - 1) / fpb;
  dim3 grid(((vt + 7) / 8) * 8 * splits);
  const int xcd_adj = MSMD_TUNE(10) != 1;   // key 10 = 1: the round-1 dealing
#define LBS_V2_LAUNCH(NS, NWV, ABL, ...)                                                                                    \
  do {                                                                                                                 \
    constexpr int lds = NS * 18 * 1024;                                                                                \
    auto kfn = lbs_skin_v2_kernel<6, NS, NWV, ABL, false, ##__VA_ARGS__>;                                                                 \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                       \
    hipLaunchKernelGGL(kfn, grid, dim3(64 * NWV), lds, (hipStream_t)stream, (const unsigned char*)skin_tiles,           \
                       v_template, (const bf16_t*)dirs_hl, lbs_weights, verts, B, V, Vp, fpb, vt, \
                       (float*)nullptr, xcd_adj, shape_varies, tmpl_folded, 0);                                                                           \
  } while (0)
  if (vposed) {
    constexpr int lds = 4 * 18 * 1024;
    auto kfn = lbs_skin_v2_kernel<6, 4, 8, 0, true>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int vt8w = (V + 127) / 128;
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, (hipStream_t)stream, (const unsigned char*)skin_tiles,
                       v_template, (const bf16_t*)dirs_hl, lbs_weights, verts, B, V, Vp, fpb, vt8w, vposed, xcd_adj, (const int*)nullptr, (const float*)nullptr, 0);
#ifdef MSMD_EXPERIMENTAL
  } else if (!big) {
    LBS_V2_LAUNCH(3, 4, 0);
  } else if (MSMD_TUNE(8)) {
    switch (MSMD_TUNE(8)) {   // ablation builds (tools/lbs_ablate.py)
      case 1: LBS_V2_LAUNCH(4, 8, 1); break;
      case 2: LBS_V2_LAUNCH(4, 8, 2); break;
      case 4: LBS_V2_LAUNCH(4, 8, 4); break;
      case 8: LBS_V2_LAUNCH(4, 8, 8); break;
      case 7: LBS_V2_LAUNCH(4, 8, 7); break;
      case 15: LBS_V2_LAUNCH(4, 8, 15); break;
      default: LBS_V2_LAUNCH(4, 8, 0); break;         // 100: one barrier per tile (the round-2a schedule)
    }
#endif
  } else if (V_ld16 && single_plane) {
    // (143 registers.  Twelve-wave workgroups of 192 vertices at three waves per SIMD measured the same as these eight-wave ones:
    // 0.387-0.392 against 0.388-0.394 ms at 25 600 frames; two 8-wave workgroups per CU need 128 registers and spilled 24 bytes
    // into the counted-vmcnt loop.)
    constexpr int lds = 4 * 12 * 1024;
    auto kfn = lbs_skin_v2_kernel<6, 4, 8, 0, false, 2, true, true>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, (hipStream_t)stream, (const unsigned char*)skin_tiles, v_template,
                       (const bf16_t*)dirs_hl, lbs_weights, verts, B, V, Vp, fpb, vt, (float*)nullptr, xcd_adj, shape_varies,
                       tmpl_folded, V_ld16);
  } else if (V_ld16) {
    constexpr int lds = 4 * 18 * 1024;
    auto kfn = lbs_skin_v2_kernel<6, 4, 8, 0, false, 2, true>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, (hipStream_t)stream, (const unsigned char*)skin_tiles, v_template,
                       (const bf16_t*)dirs_hl, lbs_weights, verts, B, V, Vp, fpb, vt, (float*)nullptr, xcd_adj, shape_varies,
                       tmpl_folded, V_ld16);
  } else {
    LBS_V2_LAUNCH(4, 8, 0, 2);                        // one barrier per two tiles + wave priorities
  }
#undef LBS_V2_LAUNCH
  MSMD_RETURN_LAST();
}

extern "C" int msmd_lbs_skin_v2(const void* skin_tiles, const float* v_template, const void* dirs_hl,
                                const float* lbs_weights, float* verts, int B, int J, int V, int Vp, int Kp,
                                const int* shape_varies, const float* v_template_folded, msmd_stream_t stream) {
  return lbs_skin_v2_impl(skin_tiles, v_template, dirs_hl, lbs_weights, verts, nullptr, B, J, V, Vp, Kp, stream,
                          shape_varies, v_template_folded);
}

extern "C" int msmd_lbs_skin_v2_f16(const void* skin_tiles, const float* v_template, const void* dirs, const float* lbs_weights,
                                    void* verts16, int B, int J, int V, int V_ld, int Vp, int Kp, const int* shape_varies,
                                    const float* v_template_folded, int single_plane, msmd_stream_t stream) {
  if (V_ld <= 0) return 1;
  return lbs_skin_v2_impl(skin_tiles, v_template, dirs, lbs_weights, (float*)verts16, nullptr, B, J, V, Vp, Kp, stream,
                          shape_varies, v_template_folded, V_ld, single_plane != 0);
}

// msmd_lbs_skin_v2's 18 KB tile records -> the 12 KB records of its single-plane fp16 form: coefficients hi + lo as ONE fp16
// number, [chunk = k / 8][frame % 16][k % 8] (6 KB), then the blend rows unchanged (6 KB).  One workgroup per tile.
__global__ __launch_bounds__(256) void lbs_tiles_f16_kernel(const unsigned char* __restrict__ tiles, unsigned char* __restrict__ out) {
  const unsigned char* src = tiles + (long)blockIdx.x * LBS_TILE_BYTES;
  unsigned char* dst = out + (long)blockIdx.x * 12288;
  for (int id = threadIdx.x; id < 24 * 16; id += 256) {       // one 16-byte chunk = 8 coefficients of one frame
    const bf16x8 h = *(const bf16x8*)(src + id * 16), l = *(const bf16x8*)(src + (24 * 16 + id) * 16);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16_t)((float)h[e] + (float)l[e]);
    *(f16x8*)(dst + id * 16) = o;
  }
  for (int id = threadIdx.x; id < 6144 / 16; id += 256)
    *(u32x4*)(dst + 6144 + id * 16) = *(const u32x4*)(src + LBS_TILE_COEF_BYTES + id * 16);
}

extern "C" int msmd_lbs_tiles_f16(const void* skin_tiles, void* tiles16, int B, msmd_stream_t stream) {
  if (B <= 0 || !skin_tiles || !tiles16 || ((uintptr_t)skin_tiles & 15) || ((uintptr_t)tiles16 & 15)) return 1;
  hipLaunchKernelGGL(lbs_tiles_f16_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)skin_tiles,
                     (unsigned char*)tiles16);
  MSMD_RETURN_LAST();
}

// Training form: the same kernel, additionally writing the un-skinned vertices v_posed (B, V, 3) for msmd_lbs_skin_bwd.
extern "C" int msmd_lbs_skin_v2_train(const void* skin_tiles, const float* v_template,
                                      const void* dirs_hl, const float* lbs_weights, float* verts, float* v_posed, int B,
                                      int J, int V, int Vp, int Kp, msmd_stream_t stream) {
  if (!v_posed || MSMD_TUNE(9) == 1) return 1;
  return lbs_skin_v2_impl(skin_tiles, v_template, dirs_hl, lbs_weights, verts, v_posed, B, J, V, Vp, Kp, stream);
}

// ---------------------------------------------------------------------------------------------------
// (coef (B, Kp), A (B, 5, 12)) fp32 -> the skinning kernel's operand formats: coef_hl (B, 2, Kp) bf16 hi / lo and the
// blend rows at_tiles (see lbs_prepare_kernel).  Used where the per-frame kinematics come from somewhere else than
// msmd_lbs_prepare: the differentiable FLAME pass computes them with autograd on (B, 5, 3, 3)-sized tensors.
__global__ __launch_bounds__(256) void lbs_pack_kernel(const float* __restrict__ coef, const float* __restrict__ A,
                                                       bf16_t* __restrict__ coef_hl, unsigned char* __restrict__ tiles,
                                                       int B, int Kp) {
  const int fl = threadIdx.x >> 4, ln = threadIdx.x & 15;
  const int b = blockIdx.x * 16 + fl;
  const bool valid = b < B;
  const int bb = valid ? b : B - 1;
  unsigned char* tile = tiles + (long)blockIdx.x * LBS_TILE_BYTES;
  bf16_t* tc = (bf16_t*)tile;
  for (int k = ln; k < Kp; k += 16) {
    float v = coef[(long)bb * Kp + k];
    asm volatile("" : "+v"(v));
    const bf16_t hi = (bf16_t)v, lo = (bf16_t)(v - (float)hi);
    if (valid && coef_hl) {
      coef_hl[((long)b * 2 + 0) * Kp + k] = hi;
      coef_hl[((long)b * 2 + 1) * Kp + k] = lo;
    }
    tc[((k >> 3) * 16 + fl) * 8 + (k & 7)] = hi;
    tc[((Kp / 8 + (k >> 3)) * 16 + fl) * 8 + (k & 7)] = lo;
  }
  f16_t* ta = (f16_t*)(tile + LBS_TILE_COEF_BYTES);
  for (int idx = ln; idx < 12 * 16; idx += 16) {
    const int m = idx >> 4, slot = idx & 15;
    float a = slot < 15 ? A[((long)bb * 5 + (slot % 5)) * 12 + m] : 0.f;
    asm volatile("" : "+v"(a));
    const f16_t ah = (f16_t)a, al = (f16_t)(a - (float)ah);
    const f16_t val = slot >= 15 ? (f16_t)0.f : ((slot >= 5 && slot < 10) ? al : ah);
    ta[((m * 2 + (slot >> 3)) * 16 + fl) * 8 + (slot & 7)] = val;
  }
}

extern "C" int msmd_lbs_pack(const float* coef, const float* A, void* coef_hl, void* at_tiles, int B, int Kp,
                             msmd_stream_t stream) {
  if (B <= 0 || Kp != 192 || !coef || !A || !at_tiles) return 1;
  hipLaunchKernelGGL(lbs_pack_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, coef, A, (bf16_t*)coef_hl,
                     (unsigned char*)at_tiles, B, Kp);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// Backward of the skinning  v = sum_j w_j (R_j p + t_j),  p = template + coef . dirs  (reference utils/lbs.py:185-221
// differentiated; the reference gets this from autograd through its materialised (B, V, 4, 4) transforms).  Given
// g = dL/dv (B, V, 3) and the forward's p (B, V, 3):
//   dp(b, v)      = (sum_j w_j R_j)^T g                       -> written as coordinate planes (B, 3, Vp) = the A operand of
//                                                                 the contraction  dcoef = dp . dirs^T  (msmd_gemm)
//   dA(b, j, r, :) = sum_v w_j(v) g_r(v) [p(v) ; 1]            -> (B, 5, 12), reduced over the vertices in registers
// One wave per frame: a lane walks vertices lane, lane + 64, ... (768 contiguous bytes per wave-load of g and p), keeps
// the 60 partial sums of dA in registers and folds them across the wave once at the end.
__global__ __launch_bounds__(64) void lbs_skin_bwd_kernel(const float* __restrict__ gverts, const float* __restrict__ vposed,
                                                          const float* __restrict__ A, const float* __restrict__ wts,
                                                          float* __restrict__ dp, float* __restrict__ dA, int V, int Vp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  __shared__ float sA[60];
  if (lane < 60) sA[lane] = A[(long)b * 60 + lane];
  __syncthreads();
  float acc[5][12];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[j][k] = 0.f;
  float* dpb = dp + (long)b * 3 * Vp;
  for (int v = lane; v < Vp; v += 64) {
    float dpx = 0.f, dpy = 0.f, dpz = 0.f;
    if (v < V) {
      const float* gp = gverts + ((long)b * V + v) * 3;
      const float* pp = vposed + ((long)b * V + v) * 3;
      const float g[3] = {gp[0], gp[1], gp[2]};
      const float ph[4] = {pp[0], pp[1], pp[2], 1.0f};
      float R[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float w = wts[(long)j * Vp + v];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int c = 0; c < 3; ++c) R[r * 3 + c] = fmaf(w, sA[j * 12 + r * 4 + c], R[r * 3 + c]);
          const float wg = w * g[r];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[j][r * 4 + c] = fmaf(wg, ph[c], acc[j][r * 4 + c]);
        }
      }
      dpx = R[0] * g[0] + R[3] * g[1] + R[6] * g[2];
      dpy = R[1] * g[0] + R[4] * g[1] + R[7] * g[2];
      dpz = R[2] * g[0] + R[5] * g[1] + R[8] * g[2];
    }
    dpb[v] = dpx;                 // padding vertices: zeros (the contraction runs over Vp)
    dpb[Vp + v] = dpy;
    dpb[2 * Vp + v] = dpz;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float t = wave_sum(acc[j][k]);
      if (lane == 0) dA[(long)b * 60 + j * 12 + k] = t;
    }
}

extern "C" int msmd_lbs_skin_bwd(const float* grad_verts, const float* v_posed, const float* A, const float* lbs_weights,
                                 float* dp_planes, float* dA, int B, int J, int V, int Vp, msmd_stream_t stream) {
  if (B <= 0 || V <= 0 || Vp < V || J != 5 || !grad_verts || !v_posed || !A || !dp_planes || !dA) return 1;
  hipLaunchKernelGGL(lbs_skin_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, grad_verts, v_posed, A, lbs_weights,
                     dp_planes, dA, V, Vp);
  MSMD_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------------
// One thread per (frame, landmark): face -> three vertex ids -> three 12-byte vertex reads (ONE dwordx3 each: the gathers are
// what this kernel is made of, and a scalar-per-component version issued nine single-dword requests per thread -- 15.7 M L2
// requests for 25 600 frames, which is what its 139 us were) -> one 12-byte store.  The index chain (idx, faces) is read
// through the read-only path; per-landmark tables are cache resident.
__global__ __launch_bounds__(256) void landmarks_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                        const int* __restrict__ idx, long idx_bs, const float* __restrict__ bary,
                                                        long bary_bs, float* __restrict__ out, int B, int V, int L) {
  struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
  struct __attribute__((packed, aligned(4))) I3 { int a, b, c; };
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * L) return;
  const int b = t / L, l = t - b * L;
  const int face = __ldg(idx + b * idx_bs + l);
  const I3 vi = *(const I3*)(faces + (long)face * 3);
  const F3 bc = *(const F3*)(bary + b * bary_bs + (long)l * 3);
  const float* vb = verts + (long)b * V * 3;
  const F3 p0 = *(const F3*)(vb + (long)vi.a * 3), p1 = *(const F3*)(vb + (long)vi.b * 3), p2 = *(const F3*)(vb + (long)vi.c * 3);
  F3 o;      // the reference's order of accumulation (utils/lbs.py:136: sum over the three corners, corner 0 first)
  o.x = p0.x * bc.x; o.y = p0.y * bc.x; o.z = p0.z * bc.x;
  o.x += p1.x * bc.y; o.y += p1.y * bc.y; o.z += p1.z * bc.y;
  o.x += p2.x * bc.z; o.y += p2.y * bc.z; o.z += p2.z * bc.z;
  *(F3*)(out + (long)t * 3) = o;
}

extern "C" int msmd_landmarks(const float* verts, const int* faces, const int* lmk_faces_idx, long idx_bstride,
                              const float* bary, long bary_bstride, float* out, int B, int V, int L,
                              msmd_stream_t stream) {
  if (B <= 0 || V <= 0 || L <= 0) return 1;
  hipLaunchKernelGGL(landmarks_kernel, dim3((B * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, faces,
                     lmk_faces_idx, idx_bstride, bary, bary_bstride, out, B, V, L);
  MSMD_RETURN_LAST();
}

// LUT row of utils/flame.py:126-172: relative neck rotation -> yaw degrees -> row in [0, 78].  The pose is axis-angle
// (pose2rot=True, J x 3) or row-major rotation matrices (pose2rot=False, J x 9).
__global__ void dyn_lmk_row_kernel(const float* __restrict__ full_pose, const int* __restrict__ chain, int n_chain,
                                   int* __restrict__ row, int B, int J, int pose_is_matrix) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float rel[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  for (int n = 0; n < n_chain; ++n) {
    const int j = chain[n];
    float r[3] = {0.f, 0.f, 0.f};
    if (!pose_is_matrix) {
      r[0] = full_pose[((long)b * J + j) * 3];
      r[1] = full_pose[((long)b * J + j) * 3 + 1];
      r[2] = full_pose[((long)b * J + j) * 3 + 2];
    }
    float R[9], o[9];
    if (pose_is_matrix) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = full_pose[((long)b * J + j) * 9 + k];
    } else {
      rodrigues(r, R);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) o[i * 3 + k] = R[i * 3] * rel[k] + R[i * 3 + 1] * rel[3 + k] + R[i * 3 + 2] * rel[6 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) rel[k] = o[k];
  }
  const float sy = sqrtf(rel[0] * rel[0] + rel[3] * rel[3]);
  const float ang = atan2f(-rel[6], sy) * 180.0f / 3.14159265358979323846f;
  long y = (long)rintf(fminf(ang, 39.0f));  // torch.round = round-half-to-even
  const long neg = y < 0 ? 1 : 0, m = y < -39 ? 1 : 0;
  const long neg_vals = m * 78 + (1 - m) * (39 - y);
  row[b] = (int)(neg * neg_vals + (1 - neg) * y);
}

extern "C" int msmd_dynamic_lmk_row(const float* full_pose, const int* neck_chain, int n_chain, int* row, int B, int J,
                                    int pose_is_matrix, msmd_stream_t stream) {
  if (B <= 0 || n_chain <= 0) return 1;
  hipLaunchKernelGGL(dyn_lmk_row_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, full_pose,
                     neck_chain, n_chain, row, B, J, pose_is_matrix);
  MSMD_RETURN_LAST();
}
