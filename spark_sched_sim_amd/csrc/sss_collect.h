// sss_collect.h - the per-step bookkeeping of the rollout workers (trainers/rollout_worker.py:133-159
// RolloutWorkerSync.collect_rollout, :162-206 RolloutWorkerAsync.collect_rollout, wrappers StochasticTimeLimit) for all
// envs at once, as two small launches per step around sss_step instead of ~45 tensor operations (the loop was host-bound:
// profiles/r03_ppo_collect_profile.txt - 75 launches per iteration, 1.5 ms of wall time against 1.07 ms of device time):
//
//   phase 0 (before the step)  the env's action from the policy's sample: stage_idx = active ? stage_sel : SSS_SKIP_ENV,
//                              num_exec = max(1, 1 + exec_sel)                       (env_wrapper.py:33-34; frozen envs skip)
//   phase 1 (after the step)   truncated = wall_time >= time_limit (the wrapper's rule); an env that reported an error leaves
//                              the collection; row t of the rollout record (active, reward, action, log-probability, times,
//                              reset flag); the env's clock / elapsed time / step count; who goes on; three flags for the
//                              host's control flow (an env failed / an episode ended / somebody goes on).
// One thread per env; plain code shared by the gfx950 kernel and the host backend of tests/emu.
#pragma once
#include <stdint.h>

struct SssCollectArgs {
  int32_t num_envs;
  int32_t asynchronous;      // 0: one episode per env (sync worker), 1: `duration` ms of simulated time per env (async worker)
  int64_t t;                 // row of the record this step fills
  double duration;
  // the env's outputs of this step (VecSparkSchedSimEnv.obs_f64 / obs_i32) and the wrapper's limits
  const double* obs_f64;     // [B][2]: reward, wall_time
  const int32_t* obs_i32;    // [B][obs_i32_stride]: .. [6] terminated, [7] error code
  int64_t obs_i32_stride;
  const double* time_limit;  // [B]
  // per-env state of the collection (in / out)
  uint8_t* active;           // in: took part in this step; out: takes part in the next one
  double* wall;              // the env's clock before this step -> after it (0 after an episode ended, async)
  double* elapsed;           // async: simulated time collected so far
  int64_t* step_counts;
  uint8_t* pending_reset;    // envs that failed: they start a new episode at the next collection
  // the policy's sample for this step
  const int64_t* stage_sel;
  const int64_t* job_idx;
  const int64_t* exec_sel;
  const float* lgprob;
  // phase 0 outputs
  int32_t* stage_idx;
  int32_t* num_exec;
  // the record: [T_cap][B] arrays, row t is written
  uint8_t* rec_active;
  double* rec_t_before;
  double* rec_t_after;
  double* rec_rewards;
  int64_t* rec_stage_sel;
  int64_t* rec_job_idx;
  int64_t* rec_exec_sel;
  float* rec_lgprobs;
  uint8_t* rec_resets;
  const uint8_t* in_group;   // nullable: only envs with in_group[b] != 0 take part in this call (their step was launched with
                             // the others skipped) - two groups of envs can then be one step apart on two streams
  int32_t* flags;            // [8], zero on entry: an env failed / an episode ended / somebody goes on / 1 + a failed env / somebody was recorded
};

#define SSS_COLLECT_SKIP_ENV (-2147483647 - 1)  // include/sss.h SSS_SKIP_ENV

// flag_or(i, v): flags[i] |= v; flag_max(i, v): flags[i] = max(flags[i], v) - slot 3 names ONE failed env (the one with
// the largest index when several fail in the same step), so it takes a maximum, not an OR of indices
template <typename OrFn, typename MaxFn>
SSS_DEV void collect_env(const SssCollectArgs& a, int phase, int b, OrFn flag_or, MaxFn flag_max) {
  const bool member = a.in_group == nullptr || a.in_group[b] != 0;
  if (phase == 0) {
    a.stage_idx[b] = (member && a.active[b]) ? (int32_t)a.stage_sel[b] : SSS_COLLECT_SKIP_ENV;
    const int64_t n = 1 + a.exec_sel[b];
    a.num_exec[b] = (int32_t)(n < 1 ? 1 : n);
    return;
  }
  if (!member) return;
  const int64_t row = a.t * (int64_t)a.num_envs + b;
  const bool was_active = a.active[b] != 0;
  const double reward = a.obs_f64[2 * b], wall_time = a.obs_f64[2 * b + 1];
  const int32_t* oi = a.obs_i32 + (int64_t)b * a.obs_i32_stride;
  const bool terminated = oi[6] != 0, truncated = wall_time >= a.time_limit[b];
  const bool bad = oi[7] != 0 && was_active;
  const bool done = (terminated || truncated) && was_active && !bad;
  // an env that failed sits out the rest of this collection; its failing step is not recorded (training.py, "truncate")
  const bool act = was_active && !bad;
  if (bad) a.pending_reset[b] = 1, flag_or(0, 1), flag_max(3, b + 1);
  if (done) flag_or(1, 1);
  const double wall = a.wall[b];
  double new_wall = act ? wall_time : wall;
  if (act) flag_or(4, 1);
  a.rec_active[row] = act ? 1 : 0;
  a.rec_rewards[row] = act ? reward : 0.0;
  a.rec_stage_sel[row] = a.stage_sel[b], a.rec_job_idx[row] = a.job_idx[b], a.rec_exec_sel[row] = a.exec_sel[b];
  a.rec_lgprobs[row] = a.lgprob[b];
  bool next = act;
  if (a.asynchronous) {
    double el = a.elapsed[b];
    a.rec_t_before[row] = el;
    if (act) el = el + (new_wall - wall);
    a.rec_t_after[row] = el;
    a.elapsed[b] = el;
    a.rec_resets[row] = done ? 1 : 0;
    if (done) new_wall = 0.0;  // the env is reset in place (rollout_worker.py:195-199)
    next = act && el < a.duration;
  } else {
    a.rec_t_before[row] = wall;
    a.rec_t_after[row] = new_wall;
    a.rec_resets[row] = 0;
    next = act && !done;
  }
  a.wall[b] = new_wall;
  if (was_active) a.step_counts[b] += 1;
  a.active[b] = next ? 1 : 0;
  if (next) flag_or(2, 1);
}
