/* exported probes of oracle/np_random.h for tests/test_oracle_rng.py (test infrastructure) */
#include "np_random.h"

void sso_t_seed_state(uint64_t seed, uint64_t out[4]) {
  sso_rng r;
  sso_rng_seed(&r, seed);
  out[0] = (uint64_t)(r.state >> 64);
  out[1] = (uint64_t)r.state;
  out[2] = (uint64_t)(r.inc >> 64);
  out[3] = (uint64_t)r.inc;
}

/* replay a script of draws: op 0 = random(), 1 = integers(arg), 2 = exponential(1.0) */
void sso_t_replay(uint64_t seed, int n, const int32_t *op, const uint32_t *arg, double *out) {
  sso_rng r;
  sso_rng_seed(&r, seed);
  for (int i = 0; i < n; i++) {
    if (op[i] == 0) out[i] = sso_random(&r);
    else if (op[i] == 1) out[i] = (double)sso_integers(&r, arg[i]);
    else out[i] = sso_exponential(&r, 1.0);
  }
}

void sso_t_log1p(int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) y[i] = sso_log1p(x[i]);
}
void sso_t_exp(int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) y[i] = sso_exp(x[i]);
}
