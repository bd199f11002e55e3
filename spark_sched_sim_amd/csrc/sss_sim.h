// sss_sim.h - the batched Spark-scheduling simulator, device side (gfx950, wave64).
//
// One wavefront (one 64-thread workgroup) simulates one environment. The env's hot block
// (header, per-executor event slots, commitment list) is staged HBM -> LDS at kernel entry and
// written back at exit. Inside a launch the wave alternates between
//   * serial phases, executed by lane 0 only (the discrete-event logic is one dependent chain
//     per env by construction), and
//   * wave-parallel phases: the event-queue pop (arg-min over one lane per executor), the
//     schedulable-stage scan (one lane per stage, ballot), the observation writer (lanes over
//     stages / edges with ballot prefix compaction), state staging and episode initialisation,
// separated by wave_sync(). Collectives are only issued from wave-uniform control flow; the
// branch conditions come from LDS flags lane 0 publishes before the sync.
//
// What each function restates is cited as ENV:n (reference spark_sched_sim/spark_sched_sim.py),
// TRK:n (components/executor_tracker.py), JOB:n (components/job.py), STG:n (components/stage.py),
// TPCH:n (data_samplers/tpch.py), EVQ:n (components/event.py). Behavioural quirks that must be
// kept are listed in SURVEY.md appendix B; the CPython-set and numpy-Generator models are
// described in DESIGN.md ("Third-party semantics").
//
// No oracle code is used here: this is an independent implementation on different data
// structures (bit masks, flat slot arrays, fixed-capacity set images); tests compare the two.
#pragma once
// -DSSS_WIDE: the instantiation for 65..128 executors (csrc/sss_hip_wide.hip, tests/emu/emu_wide.cpp): 128-entry executor arrays,
// job.local_executors kept as a count, set images of up to 1024 slots, and TWO EXECUTORS PER LANE wherever lanes stand for
// executors. Outside the event chain (the queue's pop, staging at launch boundaries, episode initialisation) a lane simply handles
// both. In the lane-parallel event machinery - fast runs, event batches - a lane speaks for the one of its two executors whose
// pending event comes first (lane_event); the other one's event is an outsider that bounds the window like any event of another
// kind. That costs window length, never correctness: a window may always be cut short, and the event that cut it is its lane's
// first event in the next round. The wave-uniform single-event paths, chunked fulfilment and the pair staging take executor ids
// and list positions, not lanes, and only needed their tables and masks widened. -DSSS_NO_BATCH (either instantiation): every
// event through the one-at-a-time handlers - the reference's control flow restated - for the byte-identity tests.
#ifdef SSS_WIDE
#define SSS_KNAME(name) name##_wide
#else
#define SSS_KNAME(name) name
#endif
#include "../../include/sss.h"
#include "sss_layout.h"
#include <wave_rt.h>  // csrc/wave_rt.h (gfx950) or tests/emu/wave_rt.h (CPU emulator), chosen by -I order

// ------------------------------------------------------------------------------------------
// per-wave context
// ------------------------------------------------------------------------------------------

// LDS of one workgroup (= one env). File-scope objects so that every access is a ds_* instruction
// (a pointer passed through a call would degrade to flat_*). Static part: the hot block and the
// scratch below; dynamic part (g_pool, sized by the host, see SssParams::off_*): the ordered
// active-job list, the job -> cache-slot map, the LDS cache of the ACTIVE jobs' records and stage
// counters (what the event chain touches on every event), and scratch for set images.
struct alignas(16) SssScratch {
  uint8_t pool8[8];  // the 8-slot table of the pool that is open (pool_open / pool_close)
  uint8_t pool8b[8]; // ... of a second pool staged next to it (pool_pair_stage)
  // the next raw outputs of the env's PCG64 stream, produced 64 at a time by the whole wave
  // (rng_refill); rng_pos of them are consumed; rng_pos == 64: none buffered, the header holds the
  // generator's state as numpy would have it
  uint64_t rng_buf[64];
  int32_t rng_pos, pad0_[3];
  uint8_t setA[SSS_SET_TABLE];
  uint8_t setB[SSS_SET_TABLE];
  // flags lane 0 publishes for the uniform control flow
  int32_t f_done, f_scan, f_round_continues;
  int32_t m_n_active, m_src_job;  // mailbox for find_schedulable_all
  int32_t n_old_active;
  int32_t events_this_step;
  int32_t events_at_launch;       // ... of which taken by earlier launches (a step cut at its event budget, sss_step_bounded)
  int32_t pending_free;           // job whose cache slot is to be released (-1: none)
  int32_t pinned_job;             // job of the event being handled: its cache slot is not given away
  int32_t sel_job, sel_stage;     // the stage an action names (select_stage_wave -> take_action)
  // the idle executors of pool `idle_key`, found with one ballot right before a lane-0 section that asks for
  // them (publish_idle_mask); consumed by the next get_idle_source_executors, valid for nothing else
  uint32_t idle_key;
  int32_t idle_valid;
  uint64_t idle_mask;
#ifdef SSS_WIDE
  uint64_t idle_mask_hi;          // ... executors 64..127
#endif
  int32_t jobset_mask, f_need_jobtime;
  // the set image of (old active list + active list) only changes when a job arrives or completes:
  // versions of the two lists it was built from (valid within one launch)
  uint32_t active_version, old_version, jobset_old_v, jobset_new_v;
  int32_t jobset_valid;
  int32_t active_dirty;           // the ordered active-job list has changed since it was staged in (env_end writes it back only then)
  uint64_t free_slots;            // bit k set <=> cache slot k is free
  double wall_old;
  uint32_t fc_dst[SSS_MAX_EXEC];  // snapshot of the source's commitments (fulfill_commitments_from_source)
#ifdef SSS_WIDE
  uint32_t fc_seq[SSS_MAX_EXEC];  // their insertion numbers while the snapshot is sorted (fulfil_order_commitments)
#endif
  int16_t fc_num[SSS_MAX_EXEC];
  // the executors that fulfil them, in the order the reference pops them (fulfil_build_list): executor,
  // index of its commitment in the snapshot, and what became of it in a lane-parallel chunk
  uint8_t fi_e[SSS_MAX_EXEC], fi_k[SSS_MAX_EXEC], fi_type[SSS_MAX_EXEC];
  int32_t fi_m, fi_m_par, f_fulfil;
  int32_t fc_n, fc_pad_[3];       // entries of the snapshot
  uint32_t rl_old[SSS_MAX_EXEC];  // batch_released_events: the members' old pools and commitment entries, by rank
  uint8_t rl_idx[SSS_MAX_EXEC];
  uint32_t rl_seq[SSS_MAX_EXEC];
  uint64_t fi_detach;             // executors a chunk detaches from the source's job
  double reset_t;                 // do_reset: arrival time of the next job while the job sequence is drawn in chunks
  int32_t reset_more, reset_pad;
  uint32_t fi_rng_pos, fi_rng_has32, fi_rng_u32, fi_pad;
};

#define SSS_STATIC_LDS_BYTES ((int)(sizeof(SssHot) + sizeof(SssScratch)))

SSS_SHARED SssHot g_hot;
SSS_SHARED SssScratch g_sc;
SSS_SHARED_DYN(g_pool);

// Per-launch context: this env's HBM pointers, the workload-pack pointers and the small parameters.
// All of it is a function of the kernel arguments (every simulator kernel takes SssKernelArgs first)
// and the workgroup id, so any function - inlined or not - reads what it needs from the
// kernel-argument segment: scalar loads the compiler knows to be invariant (they are merged and
// hoisted freely, cost no LDS round trip and no LDS space) plus a little scalar arithmetic.
// `g_c.x` builds the view and uses one member; everything unused folds away.
struct Ctx {
  uint16_t* active_g;
  SssJob* jobs;
  double* t_arrival;
  double* t_completed;
  SssStage* stages;
  float* durations;
  SssPoolHdr* pool_hdr;
  uint8_t* pool_tab;
  double* dur_ring;
  SssPackDev pk;
  SssParams P;
  int E, J_cap, SP;
};
SSS_DEV Ctx ctx_make() {
  const SssKernelArgs* a = (const SssKernelArgs*)SSS_KERNARG_PTR();
  uint8_t* env_base = (uint8_t*)a->B.state + (size_t)wave_env() * (size_t)a->L.env_stride;
  Ctx c;
  c.active_g = (uint16_t*)(env_base + a->L.off_active);
  c.jobs = (SssJob*)(env_base + a->L.off_jobs);
  c.t_arrival = (double*)(env_base + a->L.off_t_arrival);
  c.t_completed = (double*)(env_base + a->L.off_t_completed);
  c.stages = (SssStage*)(env_base + a->L.off_stages);
  c.durations = (float*)(env_base + a->L.off_durations);
  c.pool_hdr = (SssPoolHdr*)(env_base + a->L.off_pool_hdr);
  c.pool_tab = env_base + a->L.off_pool_tab;
  c.dur_ring = (double*)(env_base + a->L.off_dur_ring);
  c.pk = a->pk;
  c.P = a->P;
  c.E = a->L.E, c.J_cap = a->L.J_cap, c.SP = a->L.SP;
  return c;
}
#define g_c (ctx_make())

SSS_DEV void prof3_clear();
SSS_DEV void ctx_init() { prof3_clear(); }

#define SSS_SRC_ID 0  // which source file a failed check sits in: 0 = this file, 1.. = the parts below in include order
#define H (g_hot.h)
#ifdef SSS_CHECK_TRACE  // emulator debugging: say which invariant broke
#include <stdio.h>
#define FAIL(code)                                                                          \
  do {                                                                                      \
    if (H.err == 0) fprintf(stderr, "[FAIL] line %d: code %d\n", __LINE__, (int)(code)), H.err = (code), H.err_line = (uint64_t)SSS_SRC_ID * 100000u + __LINE__; \
  } while (0)
#else
#define FAIL(code)                                         \
  do {                                                     \
    if (H.err == 0) H.err = (code), H.err_line = (uint64_t)SSS_SRC_ID * 100000u + __LINE__; \
  } while (0)
#endif
#ifdef SSS_CHECK_TRACE
#define CHECK(cond)                                                                     \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      if (H.err == 0) fprintf(stderr, "[CHECK] line %d: %s\n", __LINE__, #cond);         \
      FAIL(SSS_ERR_INVARIANT);                                                          \
    }                                                                                   \
  } while (0)
#else
#define CHECK(cond)                          \
  do {                                       \
    if (!(cond)) FAIL(SSS_ERR_INVARIANT);     \
  } while (0)
#endif

#if defined(__HIP_DEVICE_COMPILE__) || defined(__clang__)
#define SSS_UNROLL4 _Pragma("unroll 4")
#define SSS_UNROLL8 _Pragma("unroll 8")
#else
#define SSS_UNROLL4
#define SSS_UNROLL8
#endif

#ifdef SSS_BATCH_STATS  // emulator-only census of why rounds end (tests/emu, never in the product build)
extern "C" { extern long long sss_batch_stats[128]; }
#define STAT(i, v) ((void)(wave_lane() == 0 ? (sss_batch_stats[i] += (v)) : 0))
#else
#define STAT(i, v) ((void)0)
#endif

#include "sss_prof.h"  // PROF3 scopes: empty unless a timing build defines SSS_EVPROF3
#if defined(SSS_CHECK_TRACE) && defined(SSS_UTRACE)
#define UTRACE(tag) do { if (wave_lane() < 2 && getenv("SSS_UTRACE")) fprintf(stderr, "[u] lane %d %s\n", wave_lane(), tag); } while (0)
#else
#define UTRACE(tag) ((void)0)
#endif

// ---- LDS pool views ----
#define LENW_LEN 0x3FFFFFFF  // list length in a duration descriptor (bit 30: warmup_delay is added)
#define SLOT_NONE 255
SSS_DEV uint16_t* lds_active() { return (uint16_t*)(g_pool + g_c.P.off_active); }
SSS_DEV uint8_t* lds_slot_of() { return g_pool + g_c.P.off_slot_of; }
SSS_DEV uint16_t* lds_keys() { return (uint16_t*)(g_pool + g_c.P.off_keys); }
SSS_DEV uint16_t* lds_jobset() { return (uint16_t*)(g_pool + g_c.P.off_jobset); }
SSS_DEV SssJob* lds_cjobs() { return (SssJob*)(g_pool + g_c.P.off_cjobs); }
SSS_DEV SssStage* lds_cstages() { return (SssStage*)(g_pool + g_c.P.off_cstages); }
SSS_DEV float* lds_cdur() { return (float*)(g_pool + g_c.P.off_cdur); }
SSS_DEV SssExDesc* lds_exdesc() { return (SssExDesc*)(g_pool + g_c.P.off_exdesc); }
SSS_DEV uint16_t* lds_old_active() { return (uint16_t*)(g_pool + g_c.P.off_old_active); }
SSS_DEV uint16_t* lds_slot_job() { return (uint16_t*)(g_pool + g_c.P.off_slot_job); }  // job held by a cache slot
SSS_DEV uint8_t* lds_slot_ref() { return g_pool + g_c.P.off_slot_ref; }                // pending events that name the slot

// record of job j: its LDS cache slot if it has one, else the HBM copy
SSS_DEV SssJob* jobp(int j) {
  int s = lds_slot_of()[j];
  return s != SLOT_NONE ? lds_cjobs() + s : g_c.jobs + j;
}
SSS_DEV SssStage* stgp(int j, int st) {
  int s = lds_slot_of()[j];
  return s != SLOT_NONE ? lds_cstages() + s * g_c.SP + st : g_c.stages + j * g_c.SP + st;
}
SSS_DEV float* durp(int j, int st) {  // stage.most_recent_duration (observed as f32, ENV:381)
  int s = lds_slot_of()[j];
  return s != SLOT_NONE ? lds_cdur() + s * g_c.SP + st : g_c.durations + j * g_c.SP + st;
}

// all three at once (one look-up of the job's slot). A view stays valid until the next cache_acquire / cache_release:
// those can move a job's records between HBM and LDS
struct JobView {
  SssJob* job;
  SssStage* st;  // [stage]
  float* dur;    // [stage]
};
SSS_DEV JobView jobview(int j) {
  const int s = lds_slot_of()[j];
  JobView v;
  if (s != SLOT_NONE)
    v.job = lds_cjobs() + s, v.st = lds_cstages() + s * g_c.SP, v.dur = lds_cdur() + s * g_c.SP;
  else
    v.job = g_c.jobs + j, v.st = g_c.stages + j * g_c.SP, v.dur = g_c.durations + j * g_c.SP;
  return v;
}

// pool keys
SSS_DEV uint32_t key_job_pool(int j) { return (uint32_t)(j + 1) << 8; }
SSS_DEV uint32_t key_stage_pool(int j, int s) { return ((uint32_t)(j + 1) << 8) | (uint32_t)(s + 1); }
SSS_DEV int key_job(uint32_t k) { return k == POOL_NONE ? -1 : (int)(k >> 8) - 1; }   // pool_key[0], -1 = None
SSS_DEV int key_stage(uint32_t k) { return k == POOL_NONE ? -1 : (int)(k & 0xFF) - 1; }  // pool_key[1], -1 = None
SSS_DEV int pool_index(uint32_t k) {
  int j = key_job(k), s = key_stage(k);
  if (j < 0) return 0;
  if (s < 0) return 1 + j;
  return 1 + g_c.J_cap + j * g_c.SP + s;
}

// job.local_executors (JOB:81-89) is only ever counted (TPCH:217, the executor-level key). Up to 64 executors it is kept as a
// bit mask (the event batches update it with lane masks); the wide instantiation keeps the count itself in the same field.
#ifdef SSS_WIDE
SSS_DEV int local_count(uint64_t m) { return (int)m; }
SSS_DEV uint64_t local_with(uint64_t m, int) { return m + 1; }
SSS_DEV uint64_t local_without(uint64_t m, int) { return m - 1; }
SSS_DEV bool local_has(uint64_t m, int) { return m != 0; }
// several lanes at once (event batches); the count lives in the low word and never carries or borrows out of it
SSS_DEV void local_atomic_attach(SssJob* jp, int) { lane_atomic_add_u32((uint32_t*)&jp->local_mask, 1u); }
SSS_DEV void local_atomic_detach(SssJob* jp, int) { lane_atomic_add_u32((uint32_t*)&jp->local_mask, 0u - 1u); }
// a group of executors that leaves a job together (lane 0): gathered one by one, taken off the job's record at once
struct LocalGroup { uint32_t n; };
SSS_DEV LocalGroup local_group() { return LocalGroup{0u}; }
SSS_DEV void local_group_add(LocalGroup& g, int) { g.n++; }
SSS_DEV void local_group_detach(SssJob* jp, const LocalGroup& g) {
  CHECK(jp->local_mask >= g.n);
  jp->local_mask -= g.n;
}
#else
SSS_DEV int local_count(uint64_t m) { return popc64(m); }
SSS_DEV uint64_t local_with(uint64_t m, int e) { return m | bit64(e); }
SSS_DEV uint64_t local_without(uint64_t m, int e) { return m & ~bit64(e); }
SSS_DEV bool local_has(uint64_t m, int e) { return (m & bit64(e)) != 0; }
SSS_DEV void local_atomic_attach(SssJob* jp, int e) { lane_atomic_or_u64(&jp->local_mask, bit64(e)); }
SSS_DEV void local_atomic_detach(SssJob* jp, int e) { lane_atomic_and_u64(&jp->local_mask, ~bit64(e)); }
struct LocalGroup { uint64_t m; };
SSS_DEV LocalGroup local_group() { return LocalGroup{0ull}; }
SSS_DEV void local_group_add(LocalGroup& g, int e) { g.m |= bit64(e); }
SSS_DEV void local_group_detach(SssJob* jp, const LocalGroup& g) {
  CHECK((jp->local_mask & g.m) == g.m);
  jp->local_mask &= ~g.m;
}
#endif

// ---- executors per lane ----
// Where lanes stand for executors: one each up to 64 executors; in the wide instantiation lane l holds executors l and l + 64
// (slots beyond num_executors are empty: t = +inf, EV_NONE).
#ifdef SSS_WIDE
#define SSS_EPL 2
#else
#define SSS_EPL 1
#endif
SSS_DEV int head_lane(int ex) { return ex & 63; }  // the lane that speaks for executor `ex`
// The pending event a lane speaks for in the lane-parallel event machinery, its executor, and - wide - the time of the lane's
// OTHER event, which takes no part: it bounds every window like an event of another kind (t_alt = +inf when there is none,
// and always with one executor per lane).
struct LaneEvent {
  SssEvSlot sl;
  int ex;
  double t_alt;
};
SSS_DEV LaneEvent lane_event(int lane) {
  LaneEvent le;
#ifdef SSS_WIDE
  const SssEvSlot a = g_hot.ev[lane], b = g_hot.ev[lane + 64];
  const bool b_first = b.t < a.t || (b.t == a.t && b.seq < a.seq);  // heapq's order (EVQ:35); empty slots hold +inf
  le.sl = b_first ? b : a, le.ex = b_first ? lane + 64 : lane, le.t_alt = b_first ? a.t : b.t;
#else
  le.sl = g_hot.ev[lane], le.ex = lane, le.t_alt = __builtin_inf();
#endif
  return le;
}
SSS_DEV double min_f64(double a, double b) { return b < a ? b : a; }

// ---- the device code by concern (each part cites what it restates; the order is the dependency order) ----
#include "sss_sim_rng.h"  // the numpy Generator(PCG64) stream: SeedSequence, PCG64, Lemire bounded ints, ziggurat exponential, FDLIBM log1p / exp; wave-wide jump-ahead refill
#include "sss_sim_pyset.h"  // CPython 3.10 set images (executor pools, the job-id set of the reward): add / remove / pop / resize / copy, lane-0 and wave-wide forms
#include "sss_sim_tracker.h"  // ExecutorTracker restated: commitments, pool records, moving an executor between two pools (pool_pair_*)
#include "sss_sim_jobs.h"  // jobs / stages, the task-duration sampler, the serial schedulable-stage search, executor movement (lane 0)
#include "sss_sim_fulfil.h"  // lane-parallel fulfilment of commitments
#include "sss_sim_events.h"  // the one-at-a-time event handlers (lane 0), the wave-parallel queue pop, the LDS job cache
#include "sss_sim_fast_run.h"  // the fast run: consecutive "task finished, its stage has more tasks" events in registers
#include "sss_sim_batches.h"  // lane-parallel batches of released and of arriving executors
#include "sss_sim_lean.h"  // one released / one arriving executor with wave-uniform control flow; the flush of a completing job
#include "sss_sim_observe.h"  // the wave-parallel schedulable-stage scan and the observation writer
#include "sss_sim_env.h"  // staging at launch boundaries, the pieces of a step (action, reward), the event loop, episode initialisation, do_step
#undef SSS_SRC_ID
#define SSS_SRC_ID 0
// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// reset envs whose mask byte is non-zero (mask == nullptr: all)
SSS_KERNEL void SSS_KNAME(sss_reset_kernel)(SssKernelArgs a, const uint64_t* seeds, const double* time_limits, const uint8_t* mask) {
  int env = wave_env();
  if (mask && !mask[env]) return;
  uint8_t* base = (uint8_t*)a.B.state + (size_t)env * a.L.env_stride;
  ctx_init();
  env_begin(base);
  do_reset(a.L, seeds[env], time_limits ? time_limits[env] : __builtin_inf());
  write_observation(a.L, a.B, env, 0.0);
  env_end(base);
}

// one step() per env; with auto_reset != 0 an env that is terminated at entry starts its next
// episode instead (seed += seed_stride), like a vector env in "next-step" autoreset mode
SSS_DEV void step_env(const SssKernelArgs& a, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride, int budget, uint8_t* ready) {
  int env = wave_env();
  if (stage_idx[env] == SSS_SKIP_ENV) return;  // wave-uniform: the env is not touched at all
  uint8_t* base = (uint8_t*)a.B.state + (size_t)env * a.L.env_stride;
  ctx_init();
  env_begin(base);
  double reward = 0.0;
  bool yielded = false;
  // the ballot doubles as the barrier between "all lanes read the header" and lane 0 rewriting it
  bool start_next_episode = wave_ballot(auto_reset && g_hot.h.terminated && !g_hot.h.err) != 0;
  if (start_next_episode) {
    do_reset(a.L, g_hot.h.seed + seed_stride, g_hot.h.time_limit);
  } else {
    reward = do_step(stage_idx[env], num_exec[env], budget, &yielded);
  }
  if (!yielded) write_observation(a.L, a.B, env, reward);  // (a step that goes on in the next launch has no observation yet)
  if (ready && wave_lane() == 0) ready[env] = yielded ? 0 : 1;
  env_end(base);
  prof3_flush();
}
SSS_KERNEL void SSS_KNAME(sss_step_kernel)(SssKernelArgs a, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride) {
  step_env(a, stage_idx, num_exec, auto_reset, seed_stride, 0, nullptr);
}
// sss_step_bounded: the same with an event budget per launch. A launch of sss_step_kernel ends with its slowest env - one with
// 170 events in its step where the mean is 21 (profiles/r04_bench.md) - while the other waves' SIMDs idle. Here an env whose step
// has taken `budget` events stops at the top of its next round (mid_step in its header, ready[env] = 0), the launch ends, and the
// next launch continues that step (its action arguments are not looked at) while the other envs take their next steps. Every
// env's trajectory is the one sss_step_kernel produces; what changes is which launch an env's k-th step ends in.
SSS_KERNEL void SSS_KNAME(sss_step_bounded_kernel)(SssKernelArgs a, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                                   int budget, uint8_t* ready) {
  step_env(a, stage_idx, num_exec, auto_reset, seed_stride, budget, ready);
}

// ------------------------------------------------------------------------------------------
// on-device policies: the build's batched counterparts of the reference's heuristic plugins.
// They read the same quantities a `Scheduler.schedule(obs)` plugin gets from the observation
// (per-job schedulable / frontier stages, exec_supplies, num_committable_execs, source_job_idx),
// but straight from the env state, so that no observation round trip is needed.
// ------------------------------------------------------------------------------------------

SSS_DEV int obs_num_committable() {
  uint32_t srck = g_hot.h.curr_source;
  if (srck == POOL_NONE) return 0;
  int p = pool_index(srck);
  return (int)g_c.pool_hdr[p].used - (int)g_c.pool_hdr[p].commit_from;
}

// RoundRobinScheduler.schedule (reference schedulers/heuristics/round_robin.py:14-49 with
// find_stage / preprocess_obs of heuristics/utils.py:5-37). All lanes; results are uniform.
SSS_DEV void policy_fair(bool dynamic_partition, int& stage_idx, int& num_exec) {
  int lane = wave_lane();
  // shared state is read up front; at least one collective follows before anything returns
  int A = g_hot.h.n_active;
  uint32_t srck = g_hot.h.curr_source;
  int ncommit = obs_num_committable();
  int src_job = (srck == POOL_NONE || srck == POOL_COMMON) ? -1 : key_job(srck);
  int denom = A > 1 ? A : 1;
  int cap = dynamic_partition ? (g_c.E + denom - 1) / denom : g_c.E;  // int(ceil(E / max(1, A)))
  int src_rank = -1, first_rank = -1, first_sup = 0;
  uint32_t base = 0;
  int n_chunks = (A + 63) / 64;
  if (n_chunks == 0) n_chunks = 1;
  for (int ch = 0; ch < n_chunks; ch++) {
    int k = ch * 64 + lane;
    bool valid = k < A;
    int j = valid ? (int)lds_active()[k] : 0;
    uint64_t sm = 0, act = 0;
    int sup = 0, gs = 0;
    if (valid) {
      const SssJob& job = (*jobp(j));
      sm = job.sched_mask, act = job.active_mask, sup = job.supply, gs = job.gs_base;
    }
    // find_stage: first schedulable stage with no active parent, else first schedulable stage
    int best = -1;
    uint64_t m = sm;
    while (m) {
      int s = ctz64(m);
      m &= m - 1;
      if ((g_c.pk.stage_parent_mask[gs + s] & act) == 0) {
        best = s;
        break;
      }
    }
    if (best < 0 && sm) best = ctz64(sm);
    uint32_t cnt = (uint32_t)popc64(sm);
    uint32_t excl = wave_scan_excl_u32(cnt);
    uint32_t total = wave_sum_u32(cnt);
    uint32_t abs_rank = base + excl + (best >= 0 ? (uint32_t)popc64(sm & (bit64(best) - 1)) : 0u);
    bool is_src = valid && j == src_job;
    uint64_t m_src = wave_ballot(is_src && best >= 0);
    uint64_t m_el = wave_ballot(valid && best >= 0 && !(sup >= cap || is_src));
    if (m_src != 0 && src_rank < 0) src_rank = (int)wave_bcast_u32(abs_rank, ctz64(m_src));
    if (m_el != 0 && first_rank < 0) {
      int l = ctz64(m_el);
      first_rank = (int)wave_bcast_u32(abs_rank, l);
      first_sup = (int)wave_bcast_u32((uint32_t)sup, l);
    }
    base += total;
  }
  if (src_rank >= 0) {
    stage_idx = src_rank, num_exec = ncommit;
  } else if (first_rank >= 0) {
    int room = cap - first_sup;
    stage_idx = first_rank, num_exec = ncommit < room ? ncommit : room;
  } else {
    stage_idx = -1, num_exec = ncommit;
  }
  if (num_exec < 1) num_exec = 1;
}

SSS_DEV uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// the build's counter-based uniform-random policy, keyed (episode seed, step in episode); mirrors
// hash_policy in tests/golden/make_golden.py. SURVEY 8(d) C2: stage uniform over the schedulable
// stages, num_exec uniform in [1, num_committable]; `p_none_permille` adds stage_idx = -1 draws.
SSS_DEV void policy_hash(int p_none_permille, int& stage_idx, int& num_exec) {
  uint64_t seed = g_hot.h.seed;
  uint64_t step = (uint64_t)g_hot.h.ep_steps;
  int n_sched = g_hot.h.n_sched;
  int ncommit = obs_num_committable();
  wave_sync();  // reads above vs. lane 0's writes in the step that follows
  uint64_t h = splitmix64((seed << 32) ^ step), h2 = splitmix64(h), h3 = splitmix64(h2);
  if (n_sched == 0 || (int)(h3 % 1000) < p_none_permille)
    stage_idx = -1;
  else
    stage_idx = (int)(h % (uint64_t)n_sched);
  num_exec = 1 + (int)(h2 % (uint64_t)(ncommit > 0 ? ncommit : 1));
}

enum { SSS_POLICY_FAIR = 0, SSS_POLICY_FIFO = 1, SSS_POLICY_HASH = 2 };

SSS_DEV void run_policy(int policy, int param, int& stage_idx, int& num_exec) {
  PROF3(29);
  if (policy == SSS_POLICY_HASH)
    policy_hash(param, stage_idx, num_exec);
  else
    policy_fair(policy == SSS_POLICY_FAIR, stage_idx, num_exec);
}

// writes one action per env into stage_idx / num_exec (for sss_step)
SSS_KERNEL void SSS_KNAME(sss_policy_kernel)(SssKernelArgs a, int policy, int param, int32_t* stage_idx, int32_t* num_exec) {
  int env = wave_env();
  uint8_t* base = (uint8_t*)a.B.state + (size_t)env * a.L.env_stride;
  ctx_init();
  env_begin_readonly(base);
  int si, ne;
  run_policy(policy, param, si, ne);
  if (wave_lane() == 0) stage_idx[env] = si, num_exec[env] = ne;
}

// n_steps x (policy -> step -> observe) per env in one launch; the env's hot block and job cache
// stay in LDS in between. Every step still writes the full observation, as the reference's step() does.
SSS_KERNEL void SSS_KNAME(sss_rollout_kernel)(SssKernelArgs a, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride) {
  int env = wave_env();
  uint8_t* base = (uint8_t*)a.B.state + (size_t)env * a.L.env_stride;
  ctx_init();
  env_begin(base);
  for (int it = 0; it < n_steps; it++) {
    bool over = wave_ballot(g_hot.h.terminated || g_hot.h.need_reset) != 0;
    double reward = 0.0;
    if (over) {
      if (!auto_reset || wave_ballot(g_hot.h.err != 0) != 0) break;  // failed envs stay failed
      do_reset(a.L, g_hot.h.seed + seed_stride, g_hot.h.time_limit);
    } else {
      int si, ne;
      run_policy(policy, param, si, ne);
      reward = do_step<true>(si, ne);
    }
    write_observation(a.L, a.B, env, reward);
    wave_sync();
  }
  env_end(base);
  prof3_flush();
}

#undef H
#undef FAIL
#undef CHECK
