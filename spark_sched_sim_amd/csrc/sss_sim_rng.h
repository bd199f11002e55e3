// sss_sim_rng.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// the numpy Generator(PCG64) stream: SeedSequence, PCG64, Lemire bounded ints, ziggurat exponential, FDLIBM log1p / exp; wave-wide jump-ahead refill.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 1  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
// ------------------------------------------------------------------------------------------
// numpy Generator(PCG64) stream (lane 0). Restates numpy/random: SeedSequence, pcg64 XSL-RR,
// buffered 32-bit Lemire bounded ints, the exponential ziggurat and the FDLIBM log1p/exp its slow
// path calls (third-party dependency of the reference: requirements.txt:21). Draw sites:
// TPCH:70,177,178,211,225.
// ------------------------------------------------------------------------------------------

#define PCG_MH 0x2360ED051FC65DA4ull
#define PCG_ML 0x4385DF649FCCF645ull

SSS_DEV void rng_step(SssHdr& h) {
  uint64_t lo = h.rng_state_lo, hi = h.rng_state_hi;
  uint64_t plo = lo * PCG_ML;
  uint64_t phi = mul64hi(lo, PCG_ML) + hi * PCG_ML + lo * PCG_MH;
  uint64_t rlo = plo + h.rng_inc_lo;
  uint64_t rhi = phi + h.rng_inc_hi + (rlo < plo ? 1ull : 0ull);
  h.rng_state_lo = rlo, h.rng_state_hi = rhi;
}

SSS_DEV uint64_t pcg_output(uint64_t hi, uint64_t lo) {  // XSL-RR 128/64
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}

// (a_hi:a_lo) * (b_hi:b_lo) mod 2^128
SSS_DEV void mul128(uint64_t a_hi, uint64_t a_lo, uint64_t b_hi, uint64_t b_lo, uint64_t& r_hi, uint64_t& r_lo) {
  r_lo = a_lo * b_lo;
  r_hi = mul64hi(a_lo, b_lo) + a_lo * b_hi + a_hi * b_lo;
}

// the generator's state k steps away (k in [-64, 64]) from (s_hi:s_lo): A_k * state + C_k * inc
SSS_DEV void pcg_jump(int k, uint64_t s_hi, uint64_t s_lo, uint64_t inc_hi, uint64_t inc_lo, uint64_t& r_hi, uint64_t& r_lo) {
  const uint64_t* row = g_c.pk.pcg_jump + (size_t)(k + 64) * 4;
  uint64_t a_hi, a_lo, c_hi, c_lo;
  mul128(row[0], row[1], s_hi, s_lo, a_hi, a_lo);
  mul128(row[2], row[3], inc_hi, inc_lo, c_hi, c_lo);
  r_lo = a_lo + c_lo;
  r_hi = a_hi + c_hi + (r_lo < a_lo ? 1ull : 0ull);
}

// All lanes: the next 64 raw outputs of the stream into g_sc.rng_buf, one per lane. While outputs
// are buffered the header holds the state BEHIND the last buffered output; lane l produces the
// output (rng_pos + l + 1 - 64) steps from there, so unconsumed outputs are simply produced again.
SSS_DEV void rng_refill() {
  int lane = wave_lane();
  int p = g_sc.rng_pos;
  uint64_t s_hi, s_lo;
  pcg_jump(p + lane + 1 - 64, g_hot.h.rng_state_hi, g_hot.h.rng_state_lo, g_hot.h.rng_inc_hi, g_hot.h.rng_inc_lo, s_hi, s_lo);
  wave_sync();  // every lane has read the old state
  g_sc.rng_buf[lane] = pcg_output(s_hi, s_lo);
  if (lane == 63) g_hot.h.rng_state_hi = s_hi, g_hot.h.rng_state_lo = s_lo, g_sc.rng_pos = 0;
  wave_sync();
}

// lane 0: the header's state becomes the state numpy's generator would have now (HBM image)
SSS_DEV void rng_canonicalize() {
  int p = g_sc.rng_pos;
  if (p < 64) {
    uint64_t s_hi, s_lo;
    pcg_jump(p - 64, g_hot.h.rng_state_hi, g_hot.h.rng_state_lo, g_hot.h.rng_inc_hi, g_hot.h.rng_inc_lo, s_hi, s_lo);
    g_hot.h.rng_state_hi = s_hi, g_hot.h.rng_state_lo = s_lo;
    g_sc.rng_pos = 64;
  }
}

// lane 0: one raw output - from the buffer while it lasts, else by stepping the generator
SSS_DEV uint64_t rng_next64() {
  int p = g_sc.rng_pos;
  if (p < 64) {
    g_sc.rng_pos = p + 1;
    return g_sc.rng_buf[p];
  }
  rng_step(g_hot.h);
  return pcg_output(g_hot.h.rng_state_hi, g_hot.h.rng_state_lo);
}

SSS_DEV uint32_t rng_next32() {
  SssHdr& h = g_hot.h;
  if (h.rng_has32) {
    h.rng_has32 = 0;
    return h.rng_u32;
  }
  uint64_t n = rng_next64();
  h.rng_has32 = 1;
  h.rng_u32 = (uint32_t)(n >> 32);
  return (uint32_t)n;
}

SSS_DEV double u64_to_unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
SSS_DEV double rng_random() { return u64_to_unit(rng_next64()); }

SSS_DEV uint32_t rng_integers(uint32_t n) {
  uint32_t rng = n - 1;
  if (rng == 0) return 0;
  uint64_t m = (uint64_t)rng_next32() * n;
  uint32_t leftover = (uint32_t)m;
  if (leftover < n) {
    uint32_t threshold = (0xFFFFFFFFu - rng) % n;
    while (leftover < threshold) {
      m = (uint64_t)rng_next32() * n;
      leftover = (uint32_t)m;
    }
  }
  return (uint32_t)(m >> 32);
}

SSS_DEV uint32_t ss_hashmix(uint32_t value, uint32_t& hash_const) {
  value ^= hash_const;
  hash_const *= 0x931e8875u;
  value *= hash_const;
  value ^= value >> 16;
  return value;
}
SSS_DEV uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
  r ^= r >> 16;
  return r;
}

// Generator(PCG64(SeedSequence(seed))): gymnasium's Env.reset(seed) (ENV:130)
SSS_DEV void rng_seed(SssHdr& h, uint64_t seed) {
  uint32_t ent0 = (uint32_t)seed, ent1 = (uint32_t)(seed >> 32);
  int n_ent = ent1 ? 2 : 1;
  uint32_t pool[4];
  uint32_t hc = 0x43b0d7e5u;
  pool[0] = ss_hashmix(ent0, hc);
  pool[1] = ss_hashmix(n_ent > 1 ? ent1 : 0u, hc);
  pool[2] = ss_hashmix(0u, hc);
  pool[3] = ss_hashmix(0u, hc);
  for (int s = 0; s < 4; s++)
    for (int d = 0; d < 4; d++)
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
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
  uint64_t s0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), s1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  uint64_t s2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), s3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
  // initstate = (s0 << 64) | s1 ; initseq = (s2 << 64) | s3 ; inc = (initseq << 1) | 1
  h.rng_inc_hi = (s2 << 1) | (s3 >> 63);
  h.rng_inc_lo = (s3 << 1) | 1ull;
  h.rng_state_hi = 0, h.rng_state_lo = 0;
  rng_step(h);
  uint64_t lo = h.rng_state_lo + s1;
  h.rng_state_hi = h.rng_state_hi + s0 + (lo < s1 ? 1ull : 0ull);
  h.rng_state_lo = lo;
  rng_step(h);
  h.rng_has32 = 0, h.rng_u32 = 0;
}

// FDLIBM s_log1p.c as evaluated by glibc 2.35 (split polynomial); domain here is (-1, 0]
SSS_DEV double fd_log1p(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
               Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double hfsq, f = 0, cc = 0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
  int32_t k, hx, hu = 0, ax;
  hx = (int32_t)f64_hi32(x);
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) {
      if (x == -1.0) return -two54 / 0.0;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) {
      if (two54 + x > 0.0 && ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) {
      k = 0;
      f = x;
      hu = 1;
    }
  } else if (hx >= 0x7ff00000)
    return x + x;
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = (int32_t)f64_hi32(u);
      k = (hu >> 20) - 1023;
      cc = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      cc /= u;
    } else {
      u = x;
      hu = (int32_t)f64_hi32(u);
      k = (hu >> 20) - 1023;
      cc = 0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = f64_with_hi32(u, (uint32_t)hu | 0x3ff00000u);
    } else {
      k += 1;
      u = f64_with_hi32(u, (uint32_t)hu | 0x3fe00000u);
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      cc += k * ln2_lo;
      return k * ln2_hi + cc;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + cc)) - f);
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
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + cc))) - f);
}

// FDLIBM e_exp.c for finite x <= 0 (wedge test of the ziggurat; discounted rewards)
SSS_DEV double fd_exp(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
               P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x < -745.2) return 0.0;
  double y, hi = 0, lo = 0, cc, t;
  int32_t k = 0;
  uint32_t hx = f64_hi32(x);
  int xsb = (int)((hx >> 31) & 1);
  hx &= 0x7fffffff;
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      hi = xsb ? x + ln2HI : x - ln2HI;
      lo = xsb ? -ln2LO : ln2LO;
      k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
      t = k;
      hi = x - t * ln2HI;
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) {
    return 1.0 + x;
  } else
    k = 0;
  t = x * x;
  cc = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * cc) / (cc - 2.0) - x);
  y = 1.0 - ((lo - (x * cc) / (2.0 - cc)) - hi);
  if (k >= -1021) return f64_with_hi32(y, f64_hi32(y) + ((uint32_t)k << 20));
  y = f64_with_hi32(y, f64_hi32(y) + ((uint32_t)(k + 1000) << 20));
  return y * 9.33263618503218878990e-302;
}

SSS_DEV double rng_standard_exponential() {
  for (;;) {
    uint64_t ri = rng_next64();
    ri >>= 3;
    unsigned idx = (unsigned)(ri & 0xFF);
    ri >>= 8;
    double x = (double)ri * g_c.pk.zig_we[idx];
    if (ri < g_c.pk.zig_ke[idx]) return x;
    if (idx == 0) return 7.69711747013104972 - fd_log1p(-rng_random());
    if ((g_c.pk.zig_fe[idx - 1] - g_c.pk.zig_fe[idx]) * rng_random() + g_c.pk.zig_fe[idx] < fd_exp(-x)) return x;
  }
}
