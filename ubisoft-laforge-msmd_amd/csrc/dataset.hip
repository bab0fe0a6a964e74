// HBM-resident training-corpus loader: the reference's DatasetPickle.__getitem__ + collate (datasets.py:251-368,
// 424-503) for a whole batch in ONE launch.  The corpus (all clips' audio and coefficient tracks) lives in two flat
// device arrays; per sample the host only decides the crop (a few integers, same numpy draws as the reference) and
// this kernel gathers both 100-frame windows: audio z-normalised with the clip's own mean / std (computed before any
// padding, datasets.py:258-260), zero padding where the reference pads (front / back of short clips, collate's
// pad-or-trim to 64000 samples), coefficient rows normalised with the corpus statistics.
// HBM-bound copy: 2 x 64000 x 4 B of audio + 2 x L x C x 4 B of coefficients per sample, coalesced along time.
#include "common.h"

namespace {
struct BwArgs {
  const float* audio;
  const float* coef;
  const long* desc;        // (B, 8): a_off, a_len, c_off, c_len, start1, pad_front_frames, pad_front_audio, clip
  const float* clip_stats; // (n_clips, 2): audio mean, std
  const float* cmean;      // (C) or NULL
  const float* cstd;
  float* out_audio;        // (2, B, n_audio)
  float* out_motion;       // (2, B, L, C)
  int B, L, C, n_audio;
  double unit;
};

__global__ __launch_bounds__(256) void batch_windows_kernel(const BwArgs p) {
  const int b = blockIdx.y, w = blockIdx.z;
  const long* d = p.desc + (long)b * 8;
  const long a_off = d[0], a_len = d[1], c_off = d[2], c_len = d[3], start1 = d[4], pf = d[5], pfa = d[6], clip = d[7];
  const long start = start1 + (long)w * p.L, end = start + p.L;
  const long a0 = (long)((double)start * p.unit), a1 = (long)((double)end * p.unit);
  const long n_w = a1 - a0;
  const float mean = p.clip_stats[clip * 2], sd = p.clip_stats[clip * 2 + 1] + 1e-5f;
  float* oa = p.out_audio + ((long)w * p.B + b) * p.n_audio;
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < p.n_audio; j += (long)gridDim.x * 256) {
    float v = 0.f;
    const long s = a0 + j - pfa;
    if (j < n_w && s >= 0 && s < a_len) v = (p.audio[a_off + s] - mean) / sd;
    oa[j] = v;
  }
  if (blockIdx.x == 0) {
    float* om = p.out_motion + ((long)w * p.B + b) * p.L * p.C;
    for (int i = threadIdx.x; i < p.L * p.C; i += 256) {
      const int t = i / p.C, c = i % p.C;
      const long r = start + t - pf;
      float x = (r >= 0 && r < c_len) ? p.coef[(c_off + r) * p.C + c] : 0.f;
      if (p.cmean) x = (x - p.cmean[c]) / (p.cstd[c] + 1e-9f);
      om[i] = x;
    }
  }
}
}  // namespace

extern "C" int msmd_batch_windows(const float* audio_flat, const float* coef_flat, const long* desc,
                                  const float* clip_stats, const float* coef_mean, const float* coef_std,
                                  float* out_audio, float* out_motion, int B, int L, int C, int n_audio,
                                  double audio_unit, msmd_stream_t stream) {
  if (B <= 0 || L <= 0 || C <= 0 || n_audio <= 0 || !(audio_unit > 0.0) || (coef_mean == nullptr) != (coef_std == nullptr))
    return 1;
  BwArgs p{audio_flat, coef_flat, desc, clip_stats, coef_mean, coef_std, out_audio, out_motion, B, L, C, n_audio,
           audio_unit};
  hipLaunchKernelGGL(batch_windows_kernel, dim3(32, B, 2), dim3(256), 0, (hipStream_t)stream, p);
  MSMD_RETURN_LAST();
}
