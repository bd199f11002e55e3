// sss_returns.h - what the trainer computes from the collected rollouts before the PPO epochs:
//   discounted returns   trainers/utils/returns_calculator.py:67-76: R_k = r_k + exp(-beta * 1e-3 * dt_k) * R_{k+1}, per rollout, from its end
//   baselines            trainers/utils/baselines.py:12-37 with trainer.py:206-207: for the rollouts of one job sequence, every
//                        rollout's value curve (step times -> returns) interpolated (numpy.interp) at each rollout's own step
//                        times; the baseline is the mean over the sequence's rollouts
// over the [T, B] record of `RolloutCollector` (row = step, column = env; `active` marks the rows an env recorded: a prefix of its
// column). As tensor operations these were a Python loop over T (7 441 rows at BASELINE config 5: 0.30 s) and twenty operations on
// [sequences, R, R, T] tensors (0.25 s) - a fifth of an update. One thread per env (returns: the recurrence is sequential in k,
// the loads are not) and one per (step, env) query (baselines); the arithmetic is the tensor form's, operation by operation - the
// mean over a sequence's rollouts in numpy's own summation order (baseline_pairwise).
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define SSS_ANY __host__ __device__ inline
#else
#define SSS_ANY static inline
#endif

struct SssReturnsArgs {
  int64_t T, B;
  const uint8_t* active;   // [T][B]
  const double* t_before;  // [T][B]
  const double* t_after;   // [T][B]
  const double* rewards;   // [T][B]
  double beta;
  double* out;             // [T][B]
};

SSS_ANY void returns_env(const SssReturnsArgs& a, int64_t b) {
  const double c = -a.beta * 1e-3;
  double R = 0.0;
  for (int64_t k = a.T - 1; k >= 0; k--) {
    const int64_t i = k * a.B + b;
    const bool on = a.active[i] != 0;
    if (on) R = a.rewards[i] + exp(c * (a.t_after[i] - a.t_before[i])) * R;
    a.out[i] = R * (on ? 1.0 : 0.0);
  }
}

struct SssBaselineArgs {
  int64_t T, B;
  int32_t R;               // rollouts per job sequence: envs g * R .. g * R + R - 1 belong together
  int32_t skip_empty;      // != 0: rollouts that recorded nothing are left out of their sequence's mean
  const uint8_t* active;   // [T][B]
  const double* times;     // [T][B] step times (non-decreasing along a column's active prefix)
  const double* values;    // [T][B]
  const int64_t* n;        // [B] recorded steps per env
  double* out;             // [T][B]
};

// numpy.interp's case analysis on column `col` with n knots: the LAST knot j with xp[j] <= x (0 if none), clamp at the last knot,
// exact hit -> fp[j], else slope * (x - xp[j]) + fp[j]
SSS_ANY double baseline_interp(const SssBaselineArgs& a, int64_t col, double x) {
  const int64_t n = a.n[col], last = n > 0 ? n - 1 : 0;
  int64_t lo = 0, hi = n;  // first knot > x (searchsorted right)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a.times[mid * a.B + col] <= x) lo = mid + 1; else hi = mid;
  }
  int64_t j = lo - 1;
  if (j < 0) j = 0;
  if (j > last) j = last;
  const int64_t j1 = j + 1 < last ? j + 1 : last;
  const double x0 = a.times[j * a.B + col], x1 = a.times[j1 * a.B + col], y0 = a.values[j * a.B + col], y1 = a.values[j1 * a.B + col];
  if (x == x0 || j == last || x < x0) return y0;
  const double slope = (y1 - y0) / (x1 - x0);
  return slope * (x - x0) + y0;
}
// one term of a sequence's mean: rollout j's curve at x (an empty rollout contributes 0 with skip_empty)
SSS_ANY double baseline_term(const SssBaselineArgs& a, int64_t g0, int j, double x) {
  const double y = baseline_interp(a, g0 + j, x);
  return a.skip_empty ? y * (a.n[g0 + j] > 0 ? 1.0 : 0.0) : y;
}
// The sum of terms [lo, lo + n) in the order numpy's add.reduce takes over a float64 axis - what `y_hat.mean()` does
// (baselines.py:33; numpy/core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum): fewer than 8 terms one after the other from
// 0.0; up to 128 terms eight strided partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail; above that the
// range is halved (the first half rounded down to a multiple of 8). With fewer than 8 rollouts per sequence (decima_tpch.yaml: 4)
// this is the plain running sum. Pinned against numpy itself in tests/test_emu_training.py.
SSS_ANY double baseline_pairwise(const SssBaselineArgs& a, int64_t g0, double x, int lo, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; i++) res = res + baseline_term(a, g0, lo + i, x);
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = baseline_term(a, g0, lo + k, x);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; k++) r[k] = r[k] + baseline_term(a, g0, lo + i + k, x);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + baseline_term(a, g0, lo + i, x);
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return baseline_pairwise(a, g0, x, lo, n2) + baseline_pairwise(a, g0, x, lo + n2, n - n2);
}
SSS_ANY void baseline_query(const SssBaselineArgs& a, int64_t t, int64_t b) {
  const int64_t g0 = (b / a.R) * a.R;
  const double x = a.times[t * a.B + b];
  const double acc = baseline_pairwise(a, g0, x, 0, a.R);
  double cnt = (double)a.R;
  if (a.skip_empty) {
    cnt = 0.0;
    for (int j = 0; j < a.R; j++) cnt = cnt + (a.n[g0 + j] > 0 ? 1.0 : 0.0);
  }
  const double mean = a.skip_empty ? acc / (cnt < 1.0 ? 1.0 : cnt) : acc / (double)a.R;
  a.out[t * a.B + b] = mean * (a.active[t * a.B + b] ? 1.0 : 0.0);
}

#if defined(__HIPCC__)
__global__ __launch_bounds__(64) void sss_returns_kernel(SssReturnsArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b < a.B) returns_env(a, b);
}
__global__ __launch_bounds__(256) void sss_baseline_kernel(SssBaselineArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < a.T * a.B) baseline_query(a, i / a.B, i % a.B);
}
#endif
