// sss_layout.h - how one simulated environment is laid out in HBM (host + device view).
//
// One env = one contiguous, 256-byte aligned block of the state arena (torch-owned). A
// wavefront owns exactly one env for the duration of a launch, so what has to be cheap is
// "one wave streams its env's hot block into LDS with full-width loads" - not "64 envs read
// the same field". The block therefore starts with a fixed-size HOT region (header scalars,
// the per-executor event slots = the event "heap", the commitment list) that is copied
// HBM -> LDS -> HBM verbatim with 16-byte lane accesses, followed by the per-job / per-stage /
// per-pool arrays that are touched sparsely and stay in HBM (L2 / MALL resident).
#pragma once
#include <stdint.h>

// One lane per executor up to 64 executors: event slots, batch ranking, fast runs, executor-pool set images in registers.
// 65..128 executors (the reference takes any num_executors, spark_sched_sim.py:37; its level table reaches 100,
// tpch.py:237-262) run on a second instantiation of the same source compiled with -DSSS_WIDE (csrc/sss_hip_wide.hip):
// two executors per lane - in the queue's pop and the staging loops both, in the lane-parallel event machinery the one whose
// event comes first (sss_sim.h: lane_event).
#ifdef SSS_WIDE
#define SSS_MAX_EXEC 128
#else
#define SSS_MAX_EXEC 64
#endif
#define SSS_MAX_EXEC_ANY 128  // what the host accepts (it picks the instantiation by num_executors)
#define SSS_MAX_STAGES 64     // stage bit masks are 64 bits
#define SSS_MAX_JOBS 1024     // job ids are 16 bit; job-id set scratch lives in LDS (sss_sim.h)
#define SSS_JOBSET_SLOTS 2048  // CPython table for <= 1228 distinct small ints
#define SSS_MAX_LEVELS 16
// bytes of a scratch set image built from scratch (no dummies) or by set.copy(): <= 63 small ints at any resize fit 256 slots
// (CPython grows to > 4 * used, copies to > 2 * used); 128 ints: 512 slots. The two areas are also where pool tables are staged
// (pool_pair_*): up to 256 bytes each below 64 executors, up to 1024 bytes (a pool that held all 128 executors) in the wide instantiation
#ifdef SSS_WIDE
#define SSS_SET_TABLE 1024
#else
#define SSS_SET_TABLE 256
#endif
// bytes per executor-pool image: a pool that holds all 64 executors when its fill (keys + dummies)
// crosses 3/5 of a 128-slot table is rebuilt into 512 slots (set_table_resize(used * 4 = 256)); one that holds
// 128 when a 512-slot table fills up goes to 1024 (used * 4 = 512)
#define sss_pool_table_bytes_any(E) ((E) >= 128 ? 1024 : ((E) >= 64 ? 512 : 256))  // host: layout of either instantiation
#ifdef SSS_WIDE
#define sss_pool_table_bytes(E) ((E) >= 128 ? 1024 : 512)  // device, 65..128 executors
#else
#define sss_pool_table_bytes(E) ((E) >= 64 ? 512 : 256)    // device, <= 64 executors
#endif
#define SSS_DUR_RING 200      // deque(maxlen=200), reference spark_sched_sim.py:83
#define SSS_OBS_I32 8
#define SSS_OBS_F64 2

// obs_i32[env][k]
enum { OBS_N_NODES = 0, OBS_N_EDGES, OBS_N_JOBS, OBS_N_SCHED, OBS_NUM_COMMITTABLE, OBS_SOURCE_JOB_IDX, OBS_TERMINATED, OBS_ERR };
// obs_f64[env][k]
enum { OBS_REWARD = 0, OBS_WALL_TIME };

// error codes (per env, sticky until reset for everything but the three invalid-action codes)
enum {
  SSS_OK = 0,
  SSS_ERR_ACTION_SPACE = 1,  // reference ValueError spark_sched_sim.py:276-277
  SSS_ERR_STAGE_IDX = 2,     // reference KeyError  :284 (stage_idx >= number of schedulable stages)
  SSS_ERR_TOO_MANY = 4,      // reference ValueError :294-295
  SSS_ERR_STALLED = 5,       // reference AssertionError "[step]" :212-215
  SSS_ERR_NO_DURATION = 6,   // KeyError/ValueError escaping tpch.py:88-106
  SSS_ERR_INVARIANT = 7,     // any other reference assert / container error
  SSS_ERR_NEED_RESET = 8,    // stepping a finished or failed episode
  SSS_ERR_NO_LIMIT = 9,      // reference ValueError :137-138
  SSS_ERR_CAPACITY = 10      // more jobs than the arena was sized for (time-limit mode)
};

enum { EV_NONE = 0, EV_TASK_FINISHED = 2, EV_EXECUTOR_READY = 3 };

// pool keys: (job+1) << 8 | (stage+1); common pool = 0; "None" = 0xFFFFFFFF
#define POOL_NONE 0xFFFFFFFFu
#define POOL_COMMON 0u

struct SssHdr {            // 320 bytes
  uint64_t rng_state_hi, rng_state_lo, rng_inc_hi, rng_inc_lo;
  uint32_t rng_has32, rng_u32;
  double wall_time;
  double time_limit;
  uint64_t seed;           // seed of the current episode
  uint64_t n_steps;        // real step() calls completed over the env's lifetime
  uint64_t n_events;       // events popped over the env's lifetime
  uint64_t model_bytes;    // SURVEY 8(d) algorithmic byte counter (sum of B_step)
  uint32_t counter;        // event push sequence (components/event.py:25)
  int32_t next_arrival;    // cursor into the time-sorted arrivals
  int32_t J;               // realised number of jobs this episode
  int32_t n_active;        // len(active_job_ids)
  int32_t n_completed;
  uint32_t curr_source;    // pool key
  int32_t n_sched;         // len(schedulable_stages)
  int32_t obs_n_nodes;     // action_space["stage_idx"].n - 1 at the last _observe
  int32_t obs_n_sched;     // len(stage_selection_map)
  int32_t n_commits;       // live entries in the commitment list
  uint32_t commit_seq;
  int32_t supply_none;     // _total_executor_count[None]
  int32_t terminated;
  int32_t err;
  int32_t need_reset;
  int32_t dur_head, dur_n;
  int32_t episodes;        // completed episodes (auto-reset bookkeeping)
  double last_reward;
  double ep_return;        // sum of rewards of the running episode
  double last_ep_return;   // ... of the last finished episode
  double last_ep_wall;     // wall_time at which the last finished episode ended
  int32_t ep_steps;        // step() calls in the running episode
  int32_t last_ep_steps;
  double next_arrival_t;   // t_arrival[next_arrival] (+inf when exhausted): keeps the pop off HBM
  uint64_t prof[5];        // shader-clock ticks spent in: slow-path event handlers, action+fulfil, event loop, reward, observe
  uint64_t n_fast;         // TASK_FINISHED events that took the "stage has more tasks" path (batched or one at a time)
  uint64_t n_batched;      // ... of which handled by the lane-parallel batch path
  uint64_t n_rounds;       // batch rounds that committed at least one event
  uint64_t err_line;       // diagnostics: source line (csrc/sss_sim.h) of the check that set `err` last (0: none so far); survives resets
  // a step cut at its event budget (sss_step_bounded): the next launch continues its event loop and takes no action
  int32_t mid_step;        // 1: the current step's event loop has not reached its end
  int32_t step_events;     // events of the current step so far
  // The active subgraph (which jobs, in which order, with which stages) changes only when a job arrives or
  // completes or a stage completes; the observation's edge rows are a function of it alone. graph_version counts
  // those changes, obs_* say what the edge rows in the caller's buffer were written from: an unchanged graph's
  // rows are not written again (write_observation).
  uint32_t graph_version;
  uint32_t obs_graph_version;  // 0: nothing written yet
  int32_t obs_n_edges;
  uint32_t obs_bind_gen;       // SssBuffers::gen of the buffer they were written to
  // ... and what the first part of that step left in scratch for its end: the clock and the active jobs at the end of the
  // commitment round (the reward integrates over the jobs of either list, ENV:847-874); the list itself follows the active list in HBM
  double wall_old;
  int32_t n_old_active;
  int32_t pad2_;
};

// one pending event per executor at most: 16 bytes, read with a single LDS access.
// info = kind | stage << 8 | job << 16; t = +inf when the executor has no pending event.
struct SssEvSlot {
  double t;
  uint32_t seq;
  uint32_t info;
};

struct SssHot {            // staged HBM <-> LDS as a flat block
  SssHdr h;
  // event slots: one pending event per executor at most (TASK_FINISHED while busy,
  // EXECUTOR_READY while moving); the queue's pop is a wave-wide arg-min over (t, seq)
  SssEvSlot ev[SSS_MAX_EXEC];
  uint32_t ex_loc[SSS_MAX_EXEC];       // pool key, POOL_NONE while moving
  int16_t ex_job[SSS_MAX_EXEC];        // executor.job_id, -1 = None
  int8_t ex_task_stage[SSS_MAX_EXEC];  // executor.task.stage_id, -1 = task is None
  uint8_t ex_executing[SSS_MAX_EXEC];
  // commitments: insertion-ordered dict-of-dicts flattened; order within a source = c_seq
  uint32_t c_src[SSS_MAX_EXEC];
  uint32_t c_dst[SSS_MAX_EXEC];
  uint32_t c_seq[SSS_MAX_EXEC];
  int16_t c_n[SSS_MAX_EXEC];
  uint8_t pad2[128];
};

struct SssJob {            // 64 bytes, one cache line
  uint64_t active_mask;    // job.active_stages
  uint64_t frontier_mask;  // job.frontier_stages
  uint64_t selected_mask;  // env.selected_stages restricted to this job
  uint64_t sched_mask;     // env.schedulable_stages restricted to this job
  uint64_t sat_mask;       // derived: bit s <=> executor demand of stage s <= 0
  uint64_t local_mask;     // job.local_executors
  int32_t edge_off;        // first edge row of the template in the pack
  int16_t supply;          // exec_tracker._total_executor_count[job]
  int16_t sat_count;       // job.saturated_stage_count
  int16_t completion_order;
  uint8_t n_stages;
  uint8_t n_edges;         // template edges
  int32_t gs_base;         // first stage row of the template in the pack (identifies the template)
};

// 8 bytes, read and written as one access. `remaining` is a stage's task count (any int the reference's Python ints hold that fits
// 31 bits: tpch.py:185-187 is a list length); the other three count executors and are bounded by num_executors <= 128.
// Lane-parallel commits add to the record as words: word 0 = remaining, word 1 = executing | commit_to << 16 | moving_to << 24.
struct alignas(8) SssStage {
  int32_t remaining;
  int16_t executing;
  uint8_t commit_to, moving_to;
};
#define STG_W1_EXECUTING 1u
#define STG_W1_COMMIT_TO (1u << 16)
#define STG_W1_MOVING_TO (1u << 24)
static_assert(SSS_MAX_EXEC_ANY <= 255, "commit_to / moving_to are bytes");

struct SssPoolHdr {        // 16 bytes: CPython set header + outgoing commitment count + the table while it has 8 slots
  uint16_t mask, fill, used;
  int16_t commit_from;
  uint8_t tab8[8];         // the table image when mask == 7 (nearly every pool); larger tables live in the pool's slot of the overflow area
};

// per-env byte offsets, filled on the host by sss_compute_layout
struct SssLayout {
  int32_t num_envs, E, J_cap, SP, L, n_pools, n_cap, ed_cap, max_edges_per_job;
  int32_t pad_;
  int64_t off_active, off_jobs, off_t_arrival, off_t_completed, off_stages, off_durations;
  int64_t off_pool_hdr, off_pool_tab, off_dur_ring, off_old_active, env_stride, state_bytes;
};

// small by-value kernel parameters (live in SGPRs)
struct SssParams {
  int32_t cap_cfg;         // job_arrival_cap, 0 = None
  int32_t n_slots;         // LDS cache slots for active jobs (<= 64)
  int32_t jobset_slots;    // capacity of the job-id set image (power of two)
  int32_t pool_bytes;      // dynamic LDS size
  int32_t off_active, off_old_active, off_slot_of, off_keys, off_jobset, off_cjobs, off_cstages, off_cdur, off_exdesc;  // byte offsets in g_pool
  int32_t off_slot_job, off_slot_ref;
  int32_t max_edges;       // max template edges (flattened edge pass stride)
  int8_t lvl_of[8];        // pack level index of executor levels {5,10,20,40,50,60,80,100}, -1 if absent
  double mean_interarrival, moving_delay, warmup_delay, beta;
};

// read-only workload pack, device pointers (see spark_sched_sim_amd/workload.py)
struct SssPackDev {
  int32_t T, L, s_max, total_stages, total_edges, total_durations;
  int32_t n_queries, n_sizes;  // NUM_QUERIES, len(QUERY_SIZES) of the trace set (tpch.py:14-15): a job's template is query * n_sizes + size
  const int32_t *levels, *tmpl_stage_off, *tmpl_edge_off, *stage_num_tasks;
  const double* stage_rough;
  const uint64_t *stage_parent_mask, *stage_child_mask;
  const uint32_t* stage_first_keymask;
  const int32_t *stage_max_first_lvl, *edges, *desc, *durations;
  const uint64_t* zig_ke;
  const double *zig_we, *zig_fe;
  const int32_t* eff;  // [total_stages][8 executor levels][3 modes][4] = (offset, len | warmup << 30, min duration of the list, 0)
  const int32_t* eff0; // [total_stages][3 modes][4]: the same for an executor key that is in no first_wave dict (always the stage's largest level)
  const uint8_t* common_pool;  // the common pool right after reset, set(range(E)): its 16-byte record, then (tables beyond 8 slots) the table
  const uint64_t* pcg_jump;  // [129][4]: PCG64 jump-ahead by k = -64..64 steps: state' = A * state + C * inc, rows (A_hi, A_lo, C_hi, C_lo)
  // [101], by the job's executor count n: the executor-level draw of TPCH:222-229 picks the upper level exactly when the raw
  // generator output x has (x >> 11) >= lvl_thr[n]; 2^53 (never) when n sits on a level (sss_host.h: sss_build_lvl_thr)
  const uint64_t* lvl_thr;
};

struct SssBuffers {        // raw device pointers of torch-allocated tensors
  void* state;             // uint8[state_bytes]
  float* nodes;            // f32[B][n_cap][3]
  int32_t* edge_links;     // i32[B][ed_cap][2]
  int32_t* dag_ptr;        // i32[B][J_cap + 1]
  int32_t* exec_supplies;  // i32[B][J_cap]
  int32_t* obs_i32;        // i32[B][SSS_OBS_I32]
  double* obs_f64;         // f64[B][SSS_OBS_F64]
  uint32_t gen, gen_pad;   // bumped by every sss_bind_buffers: rows written to other buffers do not count as written
};

// the first argument of every simulator kernel (sss_sim.h reads it back from the kernel-argument segment)
struct SssKernelArgs {
  SssLayout L;
  SssBuffers B;
  SssParams P;
  SssPackDev pk;
};

static inline int64_t sss_align(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

#define SSS_LDS_BUDGET 10240  // bytes of LDS per workgroup that keep 16 workgroups (4 waves/SIMD) on a CU

// carve the dynamic LDS pool; returns 0 on success, -1 if even a minimal cache does not fit
// per-executor cache of the two duration descriptors the executor's stage can need next
// (executor mode "same stage", one per candidate executor level): keeps the descriptor load out
// of the event chain. LDS only, rebuilt lazily after every launch.
struct SssExDesc {          // 32 bytes, two 16-byte LDS accesses
  int32_t gs;              // pack stage row the entries belong to, -1 = invalid
  int8_t li, ri;           // the two candidate executor levels (equal when the level interval is closed)
  int16_t thr_n;           // the executor count `thr` below was fetched for (-1: none): fast_run keeps the level threshold with the entry
  int32_t off_l, lenw_l;   // list of (stage, level li, "same stage" mode)
  uint32_t thr_lo, thr_hi; // SssPackDev::lvl_thr[thr_n]: the raw-output threshold of the executor-level draw (TPCH:222-229)
  int32_t off_r, lenw_r;   // ... level ri
};

static inline int sss_compute_lds_pool(SssParams* P, int J_cap, int SP, int E, int static_bytes) {
  int jobset = J_cap <= 76 ? 128 : (J_cap <= 306 ? 512 : 2048);  // CPython resize thresholds (fill*5 >= mask*3)
  int o = 0;
  P->off_active = o, o += 2 * J_cap;
  P->off_old_active = o, o += 2 * J_cap;
  P->off_slot_of = o, o += J_cap;
  P->off_slot_ref = o, o += 64;
  o = (o + 1) & ~1;
  P->off_slot_job = o, o += 2 * 64;
  P->off_keys = o, o += 2 * ((J_cap > E ? J_cap : E) + 8);  // live keys of a set being rebuilt: job ids or executor ids
  o = (o + 15) & ~15;  // the set image is cleared and counted 16 bytes at a time
  P->off_jobset = o, o += 2 * jobset;
  o = (o + 15) & ~15;
  P->off_exdesc = o, o += (int)sizeof(SssExDesc) * E;
  o = (o + 15) & ~15;
  int per_slot = (int)sizeof(SssJob) + 8 * SP + 4 * SP;
  int budget = SSS_LDS_BUDGET - static_bytes;
  int n = (budget - o) / per_slot;
  // Large job capacities (BASELINE config 3: 200 jobs -> 4.2 KB of lists and maps): the 16-workgroups/CU target
  // is given up rather than the cache - but not by much. Measured at C3 on MI355X (profiles/r02_bench.md: `-DSSS_FALLBACK_SLOTS=n` test builds,
  // env-steps/s step / fused): 6 slots 8.3 / 24.8 M, 8: 9.1 / 23.6, 10: 9.5 / 24.0, 14: 9.4 / 23.4, 16: 9.4 / 23.2,
  // 24: 9.0 / 21.3, 32: 8.8 / 20.3, 48: 8.4 / 18.8 - occupancy is worth more than cache coverage beyond ~10 jobs.
#ifndef SSS_FALLBACK_SLOTS
#define SSS_FALLBACK_SLOTS 10
#endif
  if (n < 8) n = E > 64 ? 24 : SSS_FALLBACK_SLOTS;  // (more than 64 executors: more jobs with pending events, occupancy is gone anyway)
  if (n > 64) n = 64;
  if (n > J_cap) n = J_cap;
  P->n_slots = n, P->jobset_slots = jobset;
  P->off_cjobs = o, o += n * (int)sizeof(SssJob);
  P->off_cstages = o, o += n * 8 * SP;
  P->off_cdur = o, o += n * 4 * SP;
  P->pool_bytes = (o + 15) & ~15;
  return P->pool_bytes + static_bytes <= 65536 ? 0 : -1;
}

// hot_bytes: sizeof(SssHot) of the instantiation that will run the env (its arrays have SSS_MAX_EXEC entries)
static inline void sss_compute_layout(SssLayout* L, int num_envs, int E, int J_cap, int SP, int n_levels, int max_edges_per_job, int hot_bytes) {
  SP = (SP + 1) & ~1;  // even stride: a job's stage row (8 B records) stays 16-byte aligned
  L->num_envs = num_envs, L->E = E, L->J_cap = J_cap, L->SP = SP, L->L = n_levels;
  L->n_pools = 1 + J_cap + J_cap * SP;
  L->n_cap = J_cap * SP;
  L->max_edges_per_job = max_edges_per_job;
  L->ed_cap = J_cap * max_edges_per_job;
  L->pad_ = 0;
  int64_t o = (int64_t)hot_bytes;
  L->off_active = o, o = sss_align(o + 2 * 2 * (int64_t)J_cap, 64);  // active_job_ids, then the old list of a step cut at its event budget
  L->off_jobs = o, o += (int64_t)sizeof(SssJob) * J_cap;
  L->off_t_arrival = o, o += 8 * (int64_t)J_cap;
  L->off_t_completed = o, o += 8 * (int64_t)J_cap;
  L->off_stages = o, o = sss_align(o + 8 * (int64_t)J_cap * SP, 64);
  L->off_durations = o, o = sss_align(o + 4 * (int64_t)J_cap * SP, 64);
  L->off_pool_hdr = o, o = sss_align(o + (int64_t)sizeof(SssPoolHdr) * L->n_pools, 64);
  L->off_pool_tab = o, o += (int64_t)sss_pool_table_bytes_any(E) * L->n_pools;
  L->off_dur_ring = o, o += 8 * SSS_DUR_RING;
  L->off_old_active = o, o += 2 * (int64_t)J_cap;
  L->env_stride = sss_align(o, 256);
  L->state_bytes = L->env_stride * num_envs;
}

static_assert(sizeof(SssHdr) == 320, "SssHdr must be 320 bytes");
static_assert(sizeof(SssHot) % 16 == 0, "SssHot is copied with 16-byte accesses");
static_assert(sizeof(SssJob) == 64, "SssJob must be one 64-byte line");
static_assert(sizeof(SssStage) == 8 && sizeof(SssPoolHdr) == 16, "packed records");
