/* oracle/np_random.h - CPU restatement of the numpy random stream the reference env consumes.
 *
 * TEST INFRASTRUCTURE (see oracle/README.md): only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * The reference seeds `self.np_random = Generator(PCG64(SeedSequence(seed)))` in
 * `gymnasium.Env.reset(seed=...)` (reference spark_sched_sim/spark_sched_sim.py:130) and then
 * draws, in this order and nowhere else:
 *     integers(22)            tpch.py:177   (query number)
 *     choice(7 strings)       tpch.py:178   (== integers(7))
 *     exponential(1/rate)     tpch.py:70    (inter-arrival time)
 *     random()                tpch.py:225   (executor-level interpolation)
 *     choice(list of ints)    tpch.py:211   (== list[integers(len)])
 * numpy is a third-party dependency of the reference that is not vendored in it
 * (requirements.txt:21 pins numpy==1.26.0; the image has numpy 2.2.6). What follows restates
 * numpy's published algorithms (numpy/random/bit_generator.pyx SeedSequence,
 * src/pcg64/pcg64.h, src/distributions/distributions.c); it is pinned against the installed
 * numpy in tests/test_oracle_rng.py and, transitively, by every golden trajectory.
 */
#ifndef SSS_ORACLE_NP_RANDOM_H
#define SSS_ORACLE_NP_RANDOM_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 sso_u128;

typedef struct {
  sso_u128 state;
  sso_u128 inc;
  int has_uint32;
  uint32_t uinteger;
} sso_rng;

/* ---- SeedSequence(entropy=int seed).generate_state(4, uint64) ---- */

static inline uint32_t sso_ss_hashmix(uint32_t value, uint32_t *hash_const) {
  value ^= *hash_const;
  *hash_const *= 0x931e8875u;
  value *= *hash_const;
  value ^= value >> 16;
  return value;
}

static inline uint32_t sso_ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
  r ^= r >> 16;
  return r;
}

static inline void sso_seed_sequence_state(uint64_t seed, uint64_t out[4]) {
  /* entropy as little-endian uint32 words; int 0 -> [0]; < 2^32 -> 1 word, else 2 words */
  uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  int n_ent = (seed >> 32) ? 2 : 1;
  uint32_t pool[4];
  uint32_t hc = 0x43b0d7e5u;
  for (int i = 0; i < 4; i++) pool[i] = sso_ss_hashmix(i < n_ent ? ent[i] : 0u, &hc);
  for (int s = 0; s < 4; s++)
    for (int d = 0; d < 4; d++)
      if (s != d) pool[d] = sso_ss_mix(pool[d], sso_ss_hashmix(pool[s], &hc));
  /* generate_state: 8 uint32 words viewed as 4 little-endian uint64 */
  uint32_t w[8];
  uint32_t hb = 0x8b51f9ddu;
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= 0x58f38dedu;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  for (int i = 0; i < 4; i++) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

/* ---- PCG64 (XSL-RR 128/64, setseq) ---- */

#define SSO_PCG_MULT ((((sso_u128)0x2360ED051FC65DA4ULL) << 64) | (sso_u128)0x4385DF649FCCF645ULL)

static inline void sso_rng_seed(sso_rng *r, uint64_t seed) {
  uint64_t s[4];
  sso_seed_sequence_state(seed, s);
  sso_u128 initstate = ((sso_u128)s[0] << 64) | s[1];
  sso_u128 initseq = ((sso_u128)s[2] << 64) | s[3];
  r->inc = (initseq << 1) | 1u;
  r->state = 0;
  r->state = r->state * SSO_PCG_MULT + r->inc;
  r->state += initstate;
  r->state = r->state * SSO_PCG_MULT + r->inc;
  r->has_uint32 = 0;
  r->uinteger = 0;
}

static inline uint64_t sso_next64(sso_rng *r) {
  r->state = r->state * SSO_PCG_MULT + r->inc;
  uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}

static inline uint32_t sso_next32(sso_rng *r) {
  if (r->has_uint32) {
    r->has_uint32 = 0;
    return r->uinteger;
  }
  uint64_t n = sso_next64(r);
  r->has_uint32 = 1;
  r->uinteger = (uint32_t)(n >> 32);
  return (uint32_t)n;
}

/* Generator.random(): does not touch the 32-bit buffer */
static inline double sso_random(sso_rng *r) { return (double)(sso_next64(r) >> 11) * (1.0 / 9007199254740992.0); }

/* Generator.integers(n) / choice(len-n list) for 0 < n <= 2^32-1: buffered Lemire rejection */
static inline uint32_t sso_integers(sso_rng *r, uint32_t n) {
  uint32_t rng = n - 1;
  if (rng == 0) return 0; /* consumes nothing */
  uint64_t m = (uint64_t)sso_next32(r) * n;
  uint32_t leftover = (uint32_t)m;
  if (leftover < n) {
    uint32_t threshold = (0xFFFFFFFFu - rng) % n;
    while (leftover < threshold) {
      m = (uint64_t)sso_next32(r) * n;
      leftover = (uint32_t)m;
    }
  }
  return (uint32_t)(m >> 32);
}

/* ---- libm pieces the exponential ziggurat's slow path needs ----
 * numpy calls the C library's log1p()/exp() there. To be independent of which libm a box has,
 * these are restatements of the FDLIBM algorithms as evaluated by glibc 2.35 (the image's libm;
 * sysdeps/ieee754/dbl-64/s_log1p.c, polynomial in the split R1..R4 form). tests/test_oracle_rng.py
 * checks sso_log1p bit-for-bit against this box's libm on the domain used here, (-1, 0].
 * sso_exp feeds only a comparison (accept/reject of a wedge sample): a last-bit difference from
 * libm's exp flips the outcome with probability ~2^-52 per slow-path draw ("RNG").
 * Compile with -ffp-contract=off. */

static inline uint32_t sso_hi32(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return (uint32_t)(u >> 32);
}
static inline double sso_with_hi32(double x, uint32_t hi) {
  uint64_t u;
  memcpy(&u, &x, 8);
  u = (u & 0xFFFFFFFFull) | ((uint64_t)hi << 32);
  memcpy(&x, &u, 8);
  return x;
}

static inline double sso_log1p(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      two54 = 1.80143985094819840000e+16, Lp1 = 6.666666666666735130e-01,
                      Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                      Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01,
                      Lp6 = 1.531383769920937332e-01, Lp7 = 1.479819860511658591e-01;
  double hfsq, f = 0, c = 0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
  int32_t k, hx, hu = 0, ax;
  hx = (int32_t)sso_hi32(x);
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) { /* x < 0.41422 */
    if (ax >= 0x3ff00000) { /* x <= -1.0 */
      if (x == -1.0) return -two54 / 0.0;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) { /* |x| < 2**-29 */
      if (two54 + x > 0.0 && ax < 0x3c900000) return x; /* |x| < 2**-54 */
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) { /* -0.2929 < x < 0.41422 */
      k = 0;
      f = x;
      hu = 1;
    }
  } else if (hx >= 0x7ff00000)
    return x + x;
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = (int32_t)sso_hi32(u);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0); /* correction term */
      c /= u;
    } else {
      u = x;
      hu = (int32_t)sso_hi32(u);
      k = (hu >> 20) - 1023;
      c = 0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = sso_with_hi32(u, (uint32_t)hu | 0x3ff00000u); /* normalize u */
    } else {
      k += 1;
      u = sso_with_hi32(u, (uint32_t)hu | 0x3fe00000u); /* normalize u/2 */
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) { /* |f| < 2**-20 */
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  R1 = z * Lp1;
  z2 = z * z;
  R2 = Lp2 + z * Lp3;
  z4 = z2 * z2;
  R3 = Lp4 + z * Lp5;
  z6 = z4 * z2;
  R4 = Lp6 + z * Lp7;
  R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

/* FDLIBM e_exp.c, restricted to finite |x| < 700 (callers pass -x with 0 <= x < ~45) */
static inline double sso_exp(double x) {
  static const double halF[2] = {0.5, -0.5}, ln2HI[2] = {6.93147180369123816490e-01, -6.93147180369123816490e-01},
                      ln2LO[2] = {1.90821492927058770002e-10, -1.90821492927058770002e-10},
                      invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                      P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double y, hi = 0, lo = 0, c, t;
  int32_t k = 0, xsb;
  uint32_t hx = sso_hi32(x);
  xsb = (int32_t)((hx >> 31) & 1);
  hx &= 0x7fffffff;
  if (hx > 0x3fd62e42) {   /* |x| > 0.5 ln2 */
    if (hx < 0x3FF0A2B2) { /* and |x| < 1.5 ln2 */
      hi = x - ln2HI[xsb];
      lo = ln2LO[xsb];
      k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + halF[xsb]);
      t = k;
      hi = x - t * ln2HI[0];
      lo = t * ln2LO[0];
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) { /* |x| < 2**-28 */
    return 1.0 + x;
  } else
    k = 0;
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    uint32_t yh = sso_hi32(y);
    return sso_with_hi32(y, yh + ((uint32_t)k << 20));
  } else {
    uint32_t yh = sso_hi32(y);
    y = sso_with_hi32(y, yh + ((uint32_t)(k + 1000) << 20));
    return y * 9.33263618503218878990e-302; /* 2^-1000 */
  }
}

/* ---- Generator.standard_exponential (ziggurat) / exponential(scale) ---- */

#include "zig_tables.inc"

static inline double sso_standard_exponential(sso_rng *r) {
  for (;;) {
    uint64_t ri = sso_next64(r);
    ri >>= 3;
    unsigned idx = (unsigned)(ri & 0xFF);
    ri >>= 8;
    double x = (double)ri * ZIG_WE[idx];
    if (ri < ZIG_KE[idx]) return x; /* ~98.9 % */
    if (idx == 0) {
      /* tail: r - log1p(-U) */
      return 7.69711747013104972 - sso_log1p(-sso_random(r));
    }
    if ((ZIG_FE[idx - 1] - ZIG_FE[idx]) * sso_random(r) + ZIG_FE[idx] < sso_exp(-x)) return x;
    /* rejected: start over */
  }
}

static inline double sso_exponential(sso_rng *r, double scale) { return scale * sso_standard_exponential(r); }

#endif
