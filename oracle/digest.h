/* oracle/digest.h - C mirror of spark_sched_sim_amd/digest.py (test infrastructure) */
#ifndef SSS_ORACLE_DIGEST_H
#define SSS_ORACLE_DIGEST_H
#include <stddef.h>
#include <stdint.h>
static inline uint64_t sss_digest_words(const uint32_t *w, size_t n) {
  if (n == 0) return 0;
  const uint64_t P = 0x9E3779B97F4A7C15ull, Q = 0xC2B2AE3D27D4EB4Full;
  uint64_t h = 0, pw = 1;
  for (size_t i = 0; i < n; i++) {
    pw *= P;
    h += ((uint64_t)w[i] + 1) * pw;
  }
  return h ^ ((uint64_t)n * Q);
}
#endif
