// sss_segcat.h - log-probability of the recorded action and entropy of a categorical distribution per SEGMENT of a flat score
// array (the PPO update's `evaluate_actions`). What the reference runs here: schedulers/decima/utils.py:26-41
// `evaluate` - a segment softmax out of scatter / gather operations (max, exp, sum, division), torch.distributions' clamp of the
// probabilities to [eps, 1 - eps], their logs, the chosen entry, -sum p log p - for the stage scores (segments = the observation's
// schedulable stages), and torch.softmax + the same for the executor counts a job allows (scheduler.py:101-139). As tensor operations
// that is ~25 launches over 5 M-element arrays forward and as many backward; here one thread walks one segment (a few rows, an
// observation's neighbours next to it in memory: the 20 MB of scores stay in L2) forward, and again backward.
//   forward:   m = max s_i; e_i = exp(s_i - m); D = sum e_i + den_eps; p_i = clamp(e_i / D, eps, 1 - eps); lp_i = log p_i
//              lg = lp_chosen; ent = -sum lp_i p_i                                  (an empty segment: lg = ent = 0)
//   backward:  d lp_i = g_lg [i == chosen] - g_ent p_i; d p_i = -g_ent lp_i + d lp_i / p_i; through the clamp (0 where it bit),
//              d e_i = d p_i / D, d D = -sum d p_i e_i / D^2, d s_i = e_i (d e_i + d D)    (m carries no gradient)
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define SSS_SEGCAT_FN __host__ __device__ inline
#else
#define SSS_SEGCAT_FN static inline
#endif

struct SssSegcatArgs {
  int64_t n_seg;
  const float* scores;    // f32[rows]
  const int64_t* ptr;     // i64[n_seg + 1]: segment s = rows ptr[s] .. ptr[s + 1] - 1
  const int64_t* chosen;  // i64[n_seg]: index of the recorded action inside its segment
  float den_eps;          // added to the sum of exponentials (decima/utils.py:35: 1e-16; torch.softmax: 0)
  float* lg;              // forward out f32[n_seg]
  float* ent;             // forward out f32[n_seg]
  const float* g_lg;      // backward in f32[n_seg]
  const float* g_ent;     // backward in f32[n_seg]
  float* g_scores;        // backward out f32[rows]
};

#define SSS_SEGCAT_EPS 1.1920928955078125e-07f  // torch.finfo(torch.float32).eps

SSS_SEGCAT_FN void segcat_segment(const SssSegcatArgs& a, int64_t s, bool backward) {
  const int64_t lo = a.ptr[s], hi = a.ptr[s + 1];
  if (hi <= lo) {
    if (!backward) a.lg[s] = 0.0f, a.ent[s] = 0.0f;
    return;
  }
  float m = a.scores[lo];
  for (int64_t i = lo + 1; i < hi; i++) m = fmaxf(m, a.scores[i]);
  float D = 0.0f;
  for (int64_t i = lo; i < hi; i++) D += expf(a.scores[i] - m);
  D += a.den_eps;
  const int64_t c = lo + a.chosen[s];
  if (!backward) {
    float ent = 0.0f, lg = 0.0f;
    for (int64_t i = lo; i < hi; i++) {
      float p = expf(a.scores[i] - m) / D;
      p = fminf(fmaxf(p, SSS_SEGCAT_EPS), 1.0f - SSS_SEGCAT_EPS);
      const float lp = logf(p);
      ent -= lp * p;
      if (i == c) lg = lp;
    }
    a.lg[s] = lg, a.ent[s] = ent;
    return;
  }
  const float g_lg = a.g_lg[s], g_ent = a.g_ent[s];
  float dD = 0.0f;
  for (int64_t i = lo; i < hi; i++) {  // first the sum that every row's gradient needs ...
    const float e = expf(a.scores[i] - m), raw = e / D;
    const float p = fminf(fmaxf(raw, SSS_SEGCAT_EPS), 1.0f - SSS_SEGCAT_EPS);
    const float lp = logf(p);
    const float dlp = (i == c ? g_lg : 0.0f) - g_ent * p;
    const float dp = (raw >= SSS_SEGCAT_EPS && raw <= 1.0f - SSS_SEGCAT_EPS) ? (-g_ent * lp + dlp / p) : 0.0f;
    dD -= dp * e / (D * D);
  }
  for (int64_t i = lo; i < hi; i++) {  // ... then the rows
    const float e = expf(a.scores[i] - m), raw = e / D;
    const float p = fminf(fmaxf(raw, SSS_SEGCAT_EPS), 1.0f - SSS_SEGCAT_EPS);
    const float lp = logf(p);
    const float dlp = (i == c ? g_lg : 0.0f) - g_ent * p;
    const float dp = (raw >= SSS_SEGCAT_EPS && raw <= 1.0f - SSS_SEGCAT_EPS) ? (-g_ent * lp + dlp / p) : 0.0f;
    a.g_scores[i] = e * (dp / D + dD);
  }
}

#if defined(__HIPCC__)
__global__ __launch_bounds__(256) void sss_segcat_kernel(SssSegcatArgs a, int backward) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s < a.n_seg) segcat_segment(a, s, backward != 0);
}
static int sss_segcat_launch(const SssSegcatArgs& a, int backward, void* stream) {
  hipLaunchKernelGGL(sss_segcat_kernel, dim3((unsigned)((a.n_seg + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, backward);
  return (int)hipGetLastError();
}
#endif
