// sss_hip_wide.hip - gfx950 build of the WIDE instantiation of the simulator kernels (65..128 executors): the same source,
// sss_sim.h, compiled with SSS_WIDE (two executors per lane in the queue's pop and the staging loops, 128-entry executor arrays,
// every event through the one-at-a-time handlers). Linked into libsss_hip.so next to sss_hip.hip, which holds the C ABI and
// picks the instantiation by num_executors (sss_host.h). Same flags: -O3 -ffp-contract=off.
#define SSS_WIDE 1
#include <hip/hip_runtime.h>

#include "sss_sim.h"
#include "sss_wide.h"

int sss_wide_hot_bytes() { return (int)sizeof(SssHot); }
int sss_wide_static_lds_bytes() { return SSS_STATIC_LDS_BYTES; }

int sss_wide_launch_reset(const SssKernelArgs& a, int num_envs, const uint64_t* seeds, const double* tl, const uint8_t* mask, void* stream) {
  hipLaunchKernelGGL(sss_reset_kernel_wide, dim3(num_envs), dim3(64), (size_t)a.P.pool_bytes, (hipStream_t)stream, a, seeds, tl, mask);
  return (int)hipGetLastError();
}
int sss_wide_launch_step_bounded(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                 int budget, uint8_t* ready, void* stream) {
  hipLaunchKernelGGL(sss_step_bounded_kernel_wide, dim3(num_envs), dim3(64), (size_t)a.P.pool_bytes, (hipStream_t)stream, a, stage_idx, num_exec, auto_reset,
                     seed_stride, budget, ready);
  return (int)hipGetLastError();
}
int sss_wide_launch_step(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                         void* stream) {
  hipLaunchKernelGGL(sss_step_kernel_wide, dim3(num_envs), dim3(64), (size_t)a.P.pool_bytes, (hipStream_t)stream, a, stage_idx, num_exec, auto_reset, seed_stride);
  return (int)hipGetLastError();
}
int sss_wide_launch_policy(const SssKernelArgs& a, int num_envs, int policy, int param, int32_t* stage_idx, int32_t* num_exec, void* stream) {
  hipLaunchKernelGGL(sss_policy_kernel_wide, dim3(num_envs), dim3(64), (size_t)a.P.pool_bytes, (hipStream_t)stream, a, policy, param, stage_idx, num_exec);
  return (int)hipGetLastError();
}
int sss_wide_launch_rollout(const SssKernelArgs& a, int num_envs, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void* stream) {
  hipLaunchKernelGGL(sss_rollout_kernel_wide, dim3(num_envs), dim3(64), (size_t)a.P.pool_bytes, (hipStream_t)stream, a, policy, param, n_steps, auto_reset, seed_stride);
  return (int)hipGetLastError();
}

#ifdef SSS_EVPROF3  // timing builds only (tools/debug/evprof3.py): the scoped profiler's table of THIS instantiation
extern "C" int sss_debug_prof_wide(unsigned long long* out64) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof3), sizeof(unsigned long long) * 96) != hipSuccess) return -1;
  static const unsigned long long zeros[96] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof3), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
extern "C" int sss_debug_prof_min_wide(unsigned long long min_step_ticks) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof3_min), &min_step_ticks, sizeof(min_step_ticks)) == hipSuccess ? 0 : -1;
}
#endif
