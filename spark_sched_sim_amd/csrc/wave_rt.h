// wave_rt.h - the wave64 primitives the simulator is written against (gfx950 build).
//
// The simulator runs ONE wavefront (= one 64-thread workgroup) per environment. Everything
// cross-lane goes through the handful of collectives below; they must be called from
// wave-uniform control flow. tests/emu/wave_rt.h implements the same names on fibers so that
// the very same kernel source can be debugged and sanitised on a CPU (test infrastructure only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SSS_DEV __device__ __forceinline__
#define SSS_DEV_NOINLINE __device__ __noinline__
#define SSS_KERNEL extern "C" __global__ __launch_bounds__(64, 4)
#define SSS_SHARED __shared__
#define SSS_SHARED_DYN(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]

SSS_DEV int wave_lane() { return (int)threadIdx.x; }
SSS_DEV int wave_env() { return (int)blockIdx.x; }

// workgroup == one wave: s_barrier is free, what matters is the LDS/global fence
SSS_DEV void wave_sync() { __syncthreads(); }

SSS_DEV uint64_t wave_ballot(bool p) { return __ballot(p); }

SSS_DEV uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }

SSS_DEV double wave_bcast_f64(double v, int src) {
  int lo = __shfl(__double2loint(v), src, 64), hi = __shfl(__double2hiint(v), src, 64);
  return __hiloint2double(hi, lo);
}

SSS_DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    v = o < v ? o : v;
  }
  return v;
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

SSS_DEV uint64_t wave_clock() { return (uint64_t)clock64(); }
SSS_DEV uint64_t mul64hi(uint64_t a, uint64_t b) { return __umul64hi(a, b); }
SSS_DEV int popc64(uint64_t x) { return __popcll(x); }
SSS_DEV int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }
SSS_DEV uint64_t f64_bits(double x) { return (uint64_t)__double_as_longlong(x); }
SSS_DEV double bits_f64(uint64_t x) { return __longlong_as_double((long long)x); }
SSS_DEV uint32_t f64_hi32(double x) { return (uint32_t)__double2hiint(x); }
SSS_DEV double f64_with_hi32(double x, uint32_t hi) { return __hiloint2double((int)hi, __double2loint(x)); }
