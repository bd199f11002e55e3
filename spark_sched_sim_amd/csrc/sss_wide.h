// sss_wide.h - what the host side (sss_host.h) needs of the WIDE instantiation of the simulator, the one that runs envs with
// 65..128 executors (sss_sim.h compiled with -DSSS_WIDE in its own translation unit: csrc/sss_hip_wide.hip on gfx950,
// tests/emu/emu_wide.cpp on the CPU wave emulator). Its SssHot / scratch have 128-entry executor arrays, so the layout
// sizes come from that unit; the launchers mirror be_launch_reset / _step / _policy / _rollout of the narrow unit.
#pragma once
#include <stdint.h>

#include "sss_layout.h"

int sss_wide_hot_bytes();         // sizeof(SssHot) with SSS_MAX_EXEC = 128
int sss_wide_static_lds_bytes();  // SSS_STATIC_LDS_BYTES of that instantiation
int sss_wide_launch_reset(const SssKernelArgs& a, int num_envs, const uint64_t* seeds, const double* tl, const uint8_t* mask, void* stream);
int sss_wide_launch_step_bounded(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                 int budget, uint8_t* ready, void* stream);
int sss_wide_launch_step(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride, void* stream);
int sss_wide_launch_policy(const SssKernelArgs& a, int num_envs, int policy, int param, int32_t* stage_idx, int32_t* num_exec, void* stream);
int sss_wide_launch_rollout(const SssKernelArgs& a, int num_envs, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void* stream);
