// wave_rt.h - the wave64 primitives the simulator is written against (gfx950 build).
//
// The simulator runs ONE wavefront (= one 64-thread workgroup) per environment. Everything
// cross-lane goes through the handful of collectives below; they must be called from
// wave-uniform control flow. tests/emu/wave_rt.h implements the same names on fibers so that
// the very same kernel source can be debugged and sanitised on a CPU (test infrastructure only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// everything is inlined into the kernels: the kernel-argument segment pointer (SSS_KERNARG_PTR) is null in callees
#define SSS_DEV __device__ __forceinline__
#ifndef SSS_WAVES_PER_SIMD
#define SSS_WAVES_PER_SIMD 4
#endif
#define SSS_KERNEL extern "C" __global__ __launch_bounds__(64, SSS_WAVES_PER_SIMD)
#define SSS_SHARED __shared__
#define SSS_SHARED_DYN(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]

// the kernel-argument segment of the running kernel (constant address space: scalar, invariant loads);
// usable from any device function
#define SSS_KERNARG_PTR() ((const void*)__builtin_amdgcn_kernarg_segment_ptr())

#ifdef SSS_OPAQUE_LANE
// (experiment) the lane id through an opaque move: what is computed from it is computed where it is used instead of being
// hoisted out of the event loop and kept - spilled - across it
SSS_DEV int wave_lane() {
  int l = (int)threadIdx.x;
  asm volatile("" : "+v"(l));
  return l;
}
#else
SSS_DEV int wave_lane() { return (int)threadIdx.x; }
#endif
SSS_DEV int wave_env() { return (int)blockIdx.x; }

// Ordering point between the lanes of the env's wave: what any lane wrote to LDS or global memory before it is what every
// lane reads after it. The workgroup IS one wavefront (SSS_KERNEL: 64 threads), so the synchronisation scope is "wavefront":
// a wave's LDS operations execute in issue order and its vector memory operations reach the L1 in issue order, which is why
// the AMDGPU memory model needs no instruction at all for a wavefront-scope release / acquire - the pair below only keeps the
// COMPILER from moving or caching memory accesses across the point. __syncthreads() here meant s_waitcnt vmcnt(0) lgkmcnt(0)
// at every one of the hundreds of ordering points of a step: a full HBM round trip whenever a store was in flight (gfx9
// counts stores in vmcnt) - in the slowest envs of a launch that was most of what the lane-parallel paths waited for.
// -DSSS_SYNC_WORKGROUP restores the workgroup-scope form (A/B comparisons).
#ifdef SSS_SYNC_WORKGROUP
SSS_DEV void wave_sync() { __syncthreads(); }
#else
SSS_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

// ordering point inside ONE wavefront of a larger workgroup: earlier LDS / global writes of any lane
// are visible to every lane afterwards; no workgroup barrier (other wavefronts are not involved)
SSS_DEV void wave_sync_local() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

SSS_DEV uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // v_cmp straight into an SGPR pair

// value of lane 0 on every lane (v_readfirstlane: no LDS round trip); all lanes must be active
SSS_DEV uint32_t wave_lane0_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
SSS_DEV double wave_lane0_f64(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
SSS_DEV uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }

SSS_DEV double wave_bcast_f64(double v, int src) {
  int lo = __shfl(__double2loint(v), src, 64), hi = __shfl(__double2hiint(v), src, 64);
  return __hiloint2double(hi, lo);
}

// Minimum over each 16-lane row on the DPP network (no LDS traffic): quad swaps, half-row and row mirrors
// leave every lane with its row's minimum. `old` = the identity of min lets the compiler fold each DPP move
// into the v_min itself (v_min_u32_dpp: one instruction per step).
SSS_DEV uint32_t dpp_row_min_u32(uint32_t v) {
  uint32_t o;
  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
  v = o < v ? o : v;
  return v;
}
// min over the 64 lanes: the four row results are combined on the scalar unit
SSS_DEV uint32_t wave_min_u32(uint32_t v) {
  v = dpp_row_min_u32(v);
  uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  a = a < b ? a : b, c = c < d ? c : d;
  return a < c ? a : c;
}
// the same when only lanes 0..15 can hold the minimum (<= 16 executors): the cross-row combine is skipped
SSS_DEV uint32_t wave_min_u32_row0(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)dpp_row_min_u32(v), 0); }

// min of non-negative doubles (incl. +inf) over the wave: they order like their bit patterns, so the minimum of
// the high words, then the minimum of the low words among the lanes that hold it (two 32-bit passes: 64-bit
// operands cannot ride the DPP network)
SSS_DEV double wave_min_f64_nonneg(double x) {
  const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
  const uint32_t mh = wave_min_u32(hi);
  const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return __hiloint2double((int)mh, (int)ml);
}
SSS_DEV double wave_min_f64_nonneg_row0(double x) {
  const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
  const uint32_t mh = wave_min_u32_row0(hi);
  const uint32_t ml = wave_min_u32_row0(hi == mh ? lo : 0xFFFFFFFFu);
  return __hiloint2double((int)mh, (int)ml);
}

SSS_DEV uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}

SSS_DEV uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
  return v;
}

// sum of a float over the wave (butterfly order; only used where a tolerance applies)
SSS_DEV float wave_sum_f32(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// value of lane `l` (wave-uniform index) on every lane
SSS_DEV uint32_t wave_readlane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
SSS_DEV double wave_readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
SSS_DEV uint64_t wave_readlane_u64(uint64_t v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
// read-modify-write used when several lanes commit to the same record in one step
SSS_DEV void lane_atomic_add_i32(int32_t* p, int32_t v) { atomicAdd(p, v); }
SSS_DEV void lane_atomic_or_u64(uint64_t* p, uint64_t v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
SSS_DEV void lane_atomic_and_u64(uint64_t* p, uint64_t v) { atomicAnd((unsigned long long*)p, (unsigned long long)v); }
SSS_DEV void lane_atomic_or_u32(uint32_t* p, uint32_t v) { atomicOr(p, v); }
SSS_DEV void lane_atomic_add_u32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
SSS_DEV void lane_atomic_add_u64(uint64_t* p, uint64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
// a counter in global memory shared by the waves of a launch: returns the value before the addition
SSS_DEV int64_t global_fetch_add_i64(int64_t* p, int64_t v) { return (int64_t)atomicAdd((unsigned long long*)p, (unsigned long long)v); }
SSS_DEV void global_atomic_max_i64(int64_t* p, int64_t v) { atomicMax((long long*)p, (long long)v); }
SSS_DEV void lane_atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }  // global_atomic_add_f32
SSS_DEV void lane_atomic_max_i32(int32_t* p, int32_t v) { atomicMax(p, v); }

// exclusive prefix sum over lanes
SSS_DEV uint32_t wave_scan_excl_u32(uint32_t v) {
  uint32_t x = v;
  int lane = (int)threadIdx.x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = (uint32_t)__shfl_up((int)x, d, 64);
    if (lane >= d) x += o;
  }
  return x - v;
}

SSS_DEV int4 mk_i4(int x, int y, int z, int w) { return make_int4(x, y, z, w); }
SSS_DEV uint2 mk_u2(uint32_t x, uint32_t y) { return make_uint2(x, y); }
SSS_DEV uint4 mk_u4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return make_uint4(x, y, z, w); }
SSS_DEV uint64_t wave_clock() { return (uint64_t)clock64(); }
template <class T>
SSS_DEV void nt_store(T* p, T v) { __builtin_nontemporal_store(v, p); }
SSS_DEV uint64_t mul64hi(uint64_t a, uint64_t b) { return __umul64hi(a, b); }
SSS_DEV uint64_t bit64(int i) { return 1ull << i; }
SSS_DEV int popc64(uint64_t x) { return __popcll(x); }
SSS_DEV int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }
SSS_DEV int ctz64_nz(uint64_t x) { return __builtin_ctzll(x); }  // x != 0
SSS_DEV uint64_t f64_bits(double x) { return (uint64_t)__double_as_longlong(x); }
SSS_DEV double bits_f64(uint64_t x) { return __longlong_as_double((long long)x); }
SSS_DEV uint32_t f64_hi32(double x) { return (uint32_t)__double2hiint(x); }
SSS_DEV double f64_with_hi32(double x, uint32_t hi) { return __hiloint2double((int)hi, __double2loint(x)); }
