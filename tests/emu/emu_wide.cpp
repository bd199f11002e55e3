// tests/emu/emu_wide.cpp - the WIDE instantiation of the simulator kernels (65..128 executors, csrc/sss_sim.h with SSS_WIDE)
// on the CPU wave emulator: the counterpart of csrc/sss_hip_wide.hip (TEST INFRASTRUCTURE, see wave_rt.h).
#define SSS_WIDE 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#include "wave_rt.h"
namespace emu {
void launch(int grid, const std::function<void()>& body);
}
#ifdef SSS_BATCH_STATS
extern "C" { extern long long sss_batch_stats[128]; }
#endif
#include "sss_sim.h"
#include "sss_wide.h"

int sss_wide_hot_bytes() { return (int)sizeof(SssHot); }
int sss_wide_static_lds_bytes() { return SSS_STATIC_LDS_BYTES; }

int sss_wide_launch_reset(const SssKernelArgs& a, int num_envs, const uint64_t* seeds, const double* tl, const uint8_t* mask, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_reset_kernel_wide(a, seeds, tl, mask); });
  return 0;
}
int sss_wide_launch_step_bounded(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                 int budget, uint8_t* ready, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_step_bounded_kernel_wide(a, stage_idx, num_exec, auto_reset, seed_stride, budget, ready); });
  return 0;
}
int sss_wide_launch_step(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_step_kernel_wide(a, stage_idx, num_exec, auto_reset, seed_stride); });
  return 0;
}
int sss_wide_launch_policy(const SssKernelArgs& a, int num_envs, int policy, int param, int32_t* stage_idx, int32_t* num_exec, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_policy_kernel_wide(a, policy, param, stage_idx, num_exec); });
  return 0;
}
int sss_wide_launch_rollout(const SssKernelArgs& a, int num_envs, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_rollout_kernel_wide(a, policy, param, n_steps, auto_reset, seed_stride); });
  return 0;
}
