// tests/emu/wave_rt.h - CPU emulation of spark_sched_sim_amd/csrc/wave_rt.h (TEST INFRASTRUCTURE).
//
// Lets the unmodified kernel source (csrc/sss_sim.h) be compiled with g++ and run on the host:
// the 64 lanes of a "wave" are 64 fibers executed strictly round-robin; a fiber runs until its
// next collective (wave_sync / ballot / min / bcast), deposits its operand and yields to the
// next lane. Because every lane executes the same sequence of collectives (they are only legal
// in wave-uniform control flow), all 64 operands are present when a lane resumes. The emulator
// aborts if lanes disagree on which collective they are at, or if some lanes finish the kernel
// while others still wait - i.e. it also checks the uniformity the real hardware silently assumes.
//
// This is how the HIP code path is debugged and run under ASan/UBSan without a GPU. It is never
// loaded by the product: only tests build tests/emu (see tests/emu/Makefile).
#pragma once
#include <stdint.h>
#include <string.h>

#define SSS_DEV static inline
#define SSS_KERNEL extern "C"
#define SSS_SHARED static
#define SSS_SHARED_DYN(name) alignas(16) static uint8_t name[65536]

namespace emu {
enum Op { OP_SYNC = 1, OP_BALLOT, OP_BCAST, OP_MIN32, OP_MIN64, OP_SUM32, OP_SCAN32, OP_SUMF32 };
int lane();
int env();
// deposits (op, value), yields round-robin, returns once all 64 lanes have deposited; the
// operands of all lanes are then readable through slot(i)
void collective(int op, uint64_t value);
uint64_t slot(int lane);
}  // namespace emu

namespace emu { extern const void* g_kernargs; }  // set by the launch wrappers (emu_backend.cpp)
#define SSS_KERNARG_PTR() (emu::g_kernargs)

SSS_DEV int wave_lane() { return emu::lane(); }
SSS_DEV int wave_env() { return emu::env(); }
SSS_DEV void wave_sync() { emu::collective(emu::OP_SYNC, 0); }
SSS_DEV void wave_sync_local() { emu::collective(emu::OP_SYNC, 0); }
SSS_DEV uint64_t wave_ballot(bool p) {
  emu::collective(emu::OP_BALLOT, p ? 1 : 0);
  uint64_t m = 0;
  for (int i = 0; i < 64; i++) m |= (emu::slot(i) & 1ull) << i;
  return m;
}
SSS_DEV uint32_t wave_bcast_u32(uint32_t v, int src);
SSS_DEV double wave_bcast_f64(double v, int src);
SSS_DEV uint32_t wave_lane0_u32(uint32_t v) { return wave_bcast_u32(v, 0); }
SSS_DEV double wave_lane0_f64(double v) { return wave_bcast_f64(v, 0); }
SSS_DEV uint32_t wave_bcast_u32(uint32_t v, int src) {
  emu::collective(emu::OP_BCAST, v);
  return (uint32_t)emu::slot(src & 63);
}
SSS_DEV double wave_bcast_f64(double v, int src) {
  uint64_t u;
  memcpy(&u, &v, 8);
  emu::collective(emu::OP_BCAST, u);
  u = emu::slot(src & 63);
  memcpy(&v, &u, 8);
  return v;
}
SSS_DEV uint32_t wave_min_u32(uint32_t v) {
  emu::collective(emu::OP_MIN32, v);
  uint32_t m = 0xFFFFFFFFu;
  for (int i = 0; i < 64; i++) m = (uint32_t)emu::slot(i) < m ? (uint32_t)emu::slot(i) : m;
  return m;
}
SSS_DEV uint32_t wave_min_u32_row0(uint32_t v) {  // lanes 16..63 must not hold anything below row 0's minimum (checked)
  uint32_t m = wave_min_u32(v);
  uint32_t m0 = wave_min_u32(emu::lane() < 16 ? v : 0xFFFFFFFFu);
  if (m != m0) __builtin_trap();
  return m;
}
SSS_DEV uint64_t wave_min_u64(uint64_t v) {
  emu::collective(emu::OP_MIN64, v);
  uint64_t m = ~0ull;
  for (int i = 0; i < 64; i++) m = emu::slot(i) < m ? emu::slot(i) : m;
  return m;
}
SSS_DEV double wave_min_f64_nonneg(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  emu::collective(emu::OP_MIN64, u);
  uint64_t m = ~0ull;  // non-negative doubles order like their bit patterns
  for (int i = 0; i < 64; i++) m = emu::slot(i) < m ? emu::slot(i) : m;
  memcpy(&x, &m, 8);
  return x;
}
SSS_DEV double wave_min_f64_nonneg_row0(double x) {  // lanes 16..63 must hold +inf (checked)
  double m = wave_min_f64_nonneg(emu::lane() < 16 ? x : __builtin_inf());
  if (emu::lane() >= 16 && x != __builtin_inf()) __builtin_trap();
  return m;
}
SSS_DEV uint32_t wave_sum_u32(uint32_t v) {
  emu::collective(emu::OP_SUM32, v);
  uint32_t s = 0;
  for (int i = 0; i < 64; i++) s += (uint32_t)emu::slot(i);
  return s;
}
SSS_DEV float wave_sum_f32(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  emu::collective(emu::OP_SUMF32, u);
  float s = 0.0f;
  for (int i = 0; i < 64; i++) {
    uint32_t w = (uint32_t)emu::slot(i);
    float f;
    memcpy(&f, &w, 4);
    s += f;
  }
  return s;
}
SSS_DEV uint32_t wave_readlane_u32(uint32_t v, int l) { return wave_bcast_u32(v, l); }
SSS_DEV double wave_readlane_f64(double v, int l) { return wave_bcast_f64(v, l); }
SSS_DEV uint64_t wave_readlane_u64(uint64_t v, int l) {
  emu::collective(emu::OP_BCAST, v);
  return emu::slot(l & 63);
}
// fibers run one at a time between collectives: plain read-modify-write is atomic enough
SSS_DEV void lane_atomic_add_i32(int32_t* p, int32_t v) { *p += v; }
SSS_DEV void lane_atomic_or_u64(uint64_t* p, uint64_t v) { *p |= v; }
SSS_DEV void lane_atomic_and_u64(uint64_t* p, uint64_t v) { *p &= v; }
SSS_DEV void lane_atomic_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
SSS_DEV void lane_atomic_add_u32(uint32_t* p, uint32_t v) { *p += v; }
SSS_DEV void lane_atomic_add_u64(uint64_t* p, uint64_t v) { *p += v; }
// a counter in global memory shared by the waves of a launch: returns the value before the addition
SSS_DEV int64_t global_fetch_add_i64(int64_t* p, int64_t v) { int64_t o = *p; *p = o + v; return o; }  // (waves run one at a time)
SSS_DEV void global_atomic_max_i64(int64_t* p, int64_t v) { if (*p < v) *p = v; }  // (waves run one at a time)
SSS_DEV void lane_atomic_add_f32(float* p, float v) { *p += v; }
SSS_DEV void lane_atomic_max_i32(int32_t* p, int32_t v) { if (*p < v) *p = v; }
SSS_DEV uint32_t wave_scan_excl_u32(uint32_t v) {
  emu::collective(emu::OP_SCAN32, v);
  uint32_t s = 0;
  for (int i = 0; i < emu::lane(); i++) s += (uint32_t)emu::slot(i);
  return s;
}
SSS_DEV uint64_t wave_clock() { return 0; }
template <class T>
SSS_DEV void nt_store(T* p, T v) { *p = v; }
SSS_DEV uint64_t mul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
SSS_DEV uint64_t bit64(int i) { return 1ull << i; }
SSS_DEV int popc64(uint64_t x) { return __builtin_popcountll(x); }
SSS_DEV int ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : -1; }
SSS_DEV int ctz64_nz(uint64_t x) { if (!x) __builtin_trap(); return __builtin_ctzll(x); }
SSS_DEV uint64_t f64_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
SSS_DEV double bits_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
SSS_DEV uint32_t f64_hi32(double x) { return (uint32_t)(f64_bits(x) >> 32); }
SSS_DEV double f64_with_hi32(double x, uint32_t hi) { return bits_f64((f64_bits(x) & 0xFFFFFFFFull) | ((uint64_t)hi << 32)); }

struct uint4 { uint32_t x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { uint32_t x, y; };
SSS_DEV int4 mk_i4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
SSS_DEV uint2 mk_u2(uint32_t x, uint32_t y) { return uint2{x, y}; }
SSS_DEV uint4 mk_u4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

