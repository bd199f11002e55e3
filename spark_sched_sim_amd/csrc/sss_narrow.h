// sss_narrow.h - what the host side (sss_host.h) needs of the instantiation of the simulator kernels for up to 64 executors
// (sss_sim.h in its own translation unit: csrc/sss_hip_sim.hip on gfx950 - the unit the build compiles with machine LICM off,
// see spark_sched_sim_amd/build.py; tests/emu/emu_backend.cpp on the CPU wave emulator). Mirrors sss_wide.h.
#pragma once
#include <stdint.h>

#include "sss_layout.h"

int sss_narrow_hot_bytes();         // sizeof(SssHot) with SSS_MAX_EXEC = 64
int sss_narrow_static_lds_bytes();  // SSS_STATIC_LDS_BYTES of that instantiation
int sss_narrow_launch_reset(const SssKernelArgs& a, int num_envs, const uint64_t* seeds, const double* tl, const uint8_t* mask, void* stream);
int sss_narrow_launch_step_bounded(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                   int budget, uint8_t* ready, void* stream);
int sss_narrow_launch_step(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride, void* stream);
int sss_narrow_launch_policy(const SssKernelArgs& a, int num_envs, int policy, int param, int32_t* stage_idx, int32_t* num_exec, void* stream);
int sss_narrow_launch_rollout(const SssKernelArgs& a, int num_envs, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void* stream);
