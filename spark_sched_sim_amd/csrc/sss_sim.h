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

#define H (g_hot.h)
#ifdef SSS_CHECK_TRACE  // emulator debugging: say which invariant broke
#include <stdio.h>
#define FAIL(code)                                                                          \
  do {                                                                                      \
    if (H.err == 0) fprintf(stderr, "[FAIL] line %d: code %d\n", __LINE__, (int)(code)), H.err = (code), H.err_line = __LINE__; \
  } while (0)
#else
#define FAIL(code)                                         \
  do {                                                     \
    if (H.err == 0) H.err = (code), H.err_line = __LINE__; \
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

// ------------------------------------------------------------------------------------------
// numpy Generator(PCG64) stream (lane 0). Restates numpy/random: SeedSequence, pcg64 XSL-RR,
// buffered 32-bit Lemire bounded ints, the exponential ziggurat and the FDLIBM log1p/exp its slow
// path calls (third-party dependency of the reference: requirements.txt:21). Draw sites:
// TPCH:70,177,178,211,225.
// ------------------------------------------------------------------------------------------

#define PCG_MH 0x2360ED051FC65DA4ull
#define PCG_ML 0x4385DF649FCCF645ull

SSS_DEV void rng_step(SssHdr& h) {
  uint64_t lo = h.rng_state_lo, hi = h.rng_state_hi;
  uint64_t plo = lo * PCG_ML;
  uint64_t phi = mul64hi(lo, PCG_ML) + hi * PCG_ML + lo * PCG_MH;
  uint64_t rlo = plo + h.rng_inc_lo;
  uint64_t rhi = phi + h.rng_inc_hi + (rlo < plo ? 1ull : 0ull);
  h.rng_state_lo = rlo, h.rng_state_hi = rhi;
}

SSS_DEV uint64_t pcg_output(uint64_t hi, uint64_t lo) {  // XSL-RR 128/64
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}

// (a_hi:a_lo) * (b_hi:b_lo) mod 2^128
SSS_DEV void mul128(uint64_t a_hi, uint64_t a_lo, uint64_t b_hi, uint64_t b_lo, uint64_t& r_hi, uint64_t& r_lo) {
  r_lo = a_lo * b_lo;
  r_hi = mul64hi(a_lo, b_lo) + a_lo * b_hi + a_hi * b_lo;
}

// the generator's state k steps away (k in [-64, 64]) from (s_hi:s_lo): A_k * state + C_k * inc
SSS_DEV void pcg_jump(int k, uint64_t s_hi, uint64_t s_lo, uint64_t inc_hi, uint64_t inc_lo, uint64_t& r_hi, uint64_t& r_lo) {
  const uint64_t* row = g_c.pk.pcg_jump + (size_t)(k + 64) * 4;
  uint64_t a_hi, a_lo, c_hi, c_lo;
  mul128(row[0], row[1], s_hi, s_lo, a_hi, a_lo);
  mul128(row[2], row[3], inc_hi, inc_lo, c_hi, c_lo);
  r_lo = a_lo + c_lo;
  r_hi = a_hi + c_hi + (r_lo < a_lo ? 1ull : 0ull);
}

// All lanes: the next 64 raw outputs of the stream into g_sc.rng_buf, one per lane. While outputs
// are buffered the header holds the state BEHIND the last buffered output; lane l produces the
// output (rng_pos + l + 1 - 64) steps from there, so unconsumed outputs are simply produced again.
SSS_DEV void rng_refill() {
  int lane = wave_lane();
  int p = g_sc.rng_pos;
  uint64_t s_hi, s_lo;
  pcg_jump(p + lane + 1 - 64, g_hot.h.rng_state_hi, g_hot.h.rng_state_lo, g_hot.h.rng_inc_hi, g_hot.h.rng_inc_lo, s_hi, s_lo);
  wave_sync();  // every lane has read the old state
  g_sc.rng_buf[lane] = pcg_output(s_hi, s_lo);
  if (lane == 63) g_hot.h.rng_state_hi = s_hi, g_hot.h.rng_state_lo = s_lo, g_sc.rng_pos = 0;
  wave_sync();
}

// lane 0: the header's state becomes the state numpy's generator would have now (HBM image)
SSS_DEV void rng_canonicalize() {
  int p = g_sc.rng_pos;
  if (p < 64) {
    uint64_t s_hi, s_lo;
    pcg_jump(p - 64, g_hot.h.rng_state_hi, g_hot.h.rng_state_lo, g_hot.h.rng_inc_hi, g_hot.h.rng_inc_lo, s_hi, s_lo);
    g_hot.h.rng_state_hi = s_hi, g_hot.h.rng_state_lo = s_lo;
    g_sc.rng_pos = 64;
  }
}

// lane 0: one raw output - from the buffer while it lasts, else by stepping the generator
SSS_DEV uint64_t rng_next64() {
  int p = g_sc.rng_pos;
  if (p < 64) {
    g_sc.rng_pos = p + 1;
    return g_sc.rng_buf[p];
  }
  rng_step(g_hot.h);
  return pcg_output(g_hot.h.rng_state_hi, g_hot.h.rng_state_lo);
}

SSS_DEV uint32_t rng_next32() {
  SssHdr& h = g_hot.h;
  if (h.rng_has32) {
    h.rng_has32 = 0;
    return h.rng_u32;
  }
  uint64_t n = rng_next64();
  h.rng_has32 = 1;
  h.rng_u32 = (uint32_t)(n >> 32);
  return (uint32_t)n;
}

SSS_DEV double u64_to_unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
SSS_DEV double rng_random() { return u64_to_unit(rng_next64()); }

SSS_DEV uint32_t rng_integers(uint32_t n) {
  uint32_t rng = n - 1;
  if (rng == 0) return 0;
  uint64_t m = (uint64_t)rng_next32() * n;
  uint32_t leftover = (uint32_t)m;
  if (leftover < n) {
    uint32_t threshold = (0xFFFFFFFFu - rng) % n;
    while (leftover < threshold) {
      m = (uint64_t)rng_next32() * n;
      leftover = (uint32_t)m;
    }
  }
  return (uint32_t)(m >> 32);
}

SSS_DEV uint32_t ss_hashmix(uint32_t value, uint32_t& hash_const) {
  value ^= hash_const;
  hash_const *= 0x931e8875u;
  value *= hash_const;
  value ^= value >> 16;
  return value;
}
SSS_DEV uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
  r ^= r >> 16;
  return r;
}

// Generator(PCG64(SeedSequence(seed))): gymnasium's Env.reset(seed) (ENV:130)
SSS_DEV void rng_seed(SssHdr& h, uint64_t seed) {
  uint32_t ent0 = (uint32_t)seed, ent1 = (uint32_t)(seed >> 32);
  int n_ent = ent1 ? 2 : 1;
  uint32_t pool[4];
  uint32_t hc = 0x43b0d7e5u;
  pool[0] = ss_hashmix(ent0, hc);
  pool[1] = ss_hashmix(n_ent > 1 ? ent1 : 0u, hc);
  pool[2] = ss_hashmix(0u, hc);
  pool[3] = ss_hashmix(0u, hc);
  for (int s = 0; s < 4; s++)
    for (int d = 0; d < 4; d++)
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
  uint32_t w[8];
  uint32_t hb = 0x8b51f9ddu;
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= 0x58f38dedu;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  uint64_t s0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), s1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  uint64_t s2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), s3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
  // initstate = (s0 << 64) | s1 ; initseq = (s2 << 64) | s3 ; inc = (initseq << 1) | 1
  h.rng_inc_hi = (s2 << 1) | (s3 >> 63);
  h.rng_inc_lo = (s3 << 1) | 1ull;
  h.rng_state_hi = 0, h.rng_state_lo = 0;
  rng_step(h);
  uint64_t lo = h.rng_state_lo + s1;
  h.rng_state_hi = h.rng_state_hi + s0 + (lo < s1 ? 1ull : 0ull);
  h.rng_state_lo = lo;
  rng_step(h);
  h.rng_has32 = 0, h.rng_u32 = 0;
}

// FDLIBM s_log1p.c as evaluated by glibc 2.35 (split polynomial); domain here is (-1, 0]
SSS_DEV double fd_log1p(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
               Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double hfsq, f = 0, cc = 0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
  int32_t k, hx, hu = 0, ax;
  hx = (int32_t)f64_hi32(x);
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) {
      if (x == -1.0) return -two54 / 0.0;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) {
      if (two54 + x > 0.0 && ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) {
      k = 0;
      f = x;
      hu = 1;
    }
  } else if (hx >= 0x7ff00000)
    return x + x;
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = (int32_t)f64_hi32(u);
      k = (hu >> 20) - 1023;
      cc = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      cc /= u;
    } else {
      u = x;
      hu = (int32_t)f64_hi32(u);
      k = (hu >> 20) - 1023;
      cc = 0;
    }
    hu &= 0x000fffff;
    if (hu < 0x6a09e) {
      u = f64_with_hi32(u, (uint32_t)hu | 0x3ff00000u);
    } else {
      k += 1;
      u = f64_with_hi32(u, (uint32_t)hu | 0x3fe00000u);
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      cc += k * ln2_lo;
      return k * ln2_hi + cc;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + cc)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  R1 = z * Lp1;
  z2 = z * z;
  R2 = Lp2 + z * Lp3;
  z4 = z2 * z2;
  R3 = Lp4 + z * Lp5;
  z6 = z4 * z2;
  R4 = Lp6 + z * Lp7;
  R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + cc))) - f);
}

// FDLIBM e_exp.c for finite x <= 0 (wedge test of the ziggurat; discounted rewards)
SSS_DEV double fd_exp(double x) {
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
               P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x < -745.2) return 0.0;
  double y, hi = 0, lo = 0, cc, t;
  int32_t k = 0;
  uint32_t hx = f64_hi32(x);
  int xsb = (int)((hx >> 31) & 1);
  hx &= 0x7fffffff;
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      hi = xsb ? x + ln2HI : x - ln2HI;
      lo = xsb ? -ln2LO : ln2LO;
      k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
      t = k;
      hi = x - t * ln2HI;
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) {
    return 1.0 + x;
  } else
    k = 0;
  t = x * x;
  cc = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * cc) / (cc - 2.0) - x);
  y = 1.0 - ((lo - (x * cc) / (2.0 - cc)) - hi);
  if (k >= -1021) return f64_with_hi32(y, f64_hi32(y) + ((uint32_t)k << 20));
  y = f64_with_hi32(y, f64_hi32(y) + ((uint32_t)(k + 1000) << 20));
  return y * 9.33263618503218878990e-302;
}

SSS_DEV double rng_standard_exponential() {
  for (;;) {
    uint64_t ri = rng_next64();
    ri >>= 3;
    unsigned idx = (unsigned)(ri & 0xFF);
    ri >>= 8;
    double x = (double)ri * g_c.pk.zig_we[idx];
    if (ri < g_c.pk.zig_ke[idx]) return x;
    if (idx == 0) return 7.69711747013104972 - fd_log1p(-rng_random());
    if ((g_c.pk.zig_fe[idx - 1] - g_c.pk.zig_fe[idx]) * rng_random() + g_c.pk.zig_fe[idx] < fd_exp(-x)) return x;
  }
}

// ------------------------------------------------------------------------------------------
// CPython 3.10 set images (lane 0). Slot encoding: 0 = EMPTY, 1 = DUMMY, key + 2 otherwise.
// Restates Objects/setobject.c set_add_entry / set_lookkey / set_insert_clean / set_table_resize
// / set_merge / set_pop for keys with hash(k) == k. Why: SURVEY H1 (ENV:714-741,762,855-864).
// ------------------------------------------------------------------------------------------

template <typename T>
struct SetImg {
  T* tab;
  uint32_t mask, fill, used, finger;
  uint32_t cap;  // slots available at `tab`; a resize beyond it continues in `big` (pool images: 8 inline slots, then the overflow area)
  T* big;
  T* small;      // where an 8-slot table goes (pool images), or nullptr
  uint32_t aux;  // pool images: the record's outgoing commitment count, carried from pool_open to pool_close
  bool wide;     // `tab` is in HBM: probe groups are fetched whole (ProbeGroup)
  bool big_wide; // ... and so is `big` (only read when a resize moves the table there)
};

// One probe group of a byte table (entries i .. i + probes, at most 10) fetched with two accesses
// instead of up to ten dependent ones; used for tables in HBM (`wide`), whose accesses may be unaligned.
struct ProbeGroup {
  uint64_t lo;
  uint32_t hi;
};
SSS_DEV uint32_t probe_group_at(const ProbeGroup& g, uint32_t p) { return p < 8 ? (uint32_t)(g.lo >> (8 * p)) & 0xFFu : (g.hi >> (8 * (p - 8))) & 0xFFu; }
template <typename T>
SSS_DEV ProbeGroup probe_group_load(const T* tab, uint32_t i, uint32_t probes) {
  ProbeGroup g;
  g.lo = 0, g.hi = 0;
  if (probes) {
    uint16_t h;
    __builtin_memcpy(&g.lo, (const uint8_t*)tab + i, 8);
    __builtin_memcpy(&h, (const uint8_t*)tab + i + 8, 2);
    g.hi = h;
  } else
    g.lo = ((const uint8_t*)tab)[i];
  return g;
}

template <typename T>
SSS_DEV void set_insert_clean(T* tab, uint32_t mask, uint32_t key) {
  uint32_t perturb = key;
  uint32_t i = key & mask;
  for (;;) {
    uint32_t probes = (i + 9 <= mask) ? 9 : 0;
    for (uint32_t p = 0; p <= probes; p++) {
      if (tab[i + p] == 0) {
        tab[i + p] = (T)(key + 2);
        return;
      }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

// `keys` is scratch for the live keys (>= used entries)
template <typename T>
SSS_DEV void set_resize(SetImg<T>& s, uint32_t minused, uint16_t* keys) {
  uint32_t newsize = 8;
  while (newsize <= minused) newsize <<= 1;
  uint32_t n = 0;
  for (uint32_t i = 0; i <= s.mask; i++) {
    uint32_t e = s.tab[i];
    if (e >= 2) keys[n++] = (uint16_t)(e - 2);
  }
  if (newsize > s.cap)
    s.tab = s.big, s.cap = 0xFFFFFFFFu, s.wide = s.big_wide;  // the live keys are in `keys`: nothing to copy
  else if (newsize <= 8 && s.small)
    s.tab = s.small, s.cap = 8, s.wide = false;         // a pool image that fits its record again
  for (uint32_t i = 0; i < newsize; i++) s.tab[i] = 0;
  for (uint32_t i = 0; i < n; i++) set_insert_clean(s.tab, newsize - 1, keys[i]);
  s.mask = newsize - 1;
  s.fill = s.used;
}

template <typename T>
SSS_DEV void set_add(SetImg<T>& s, uint32_t key, uint16_t* keys) {
  uint32_t mask = s.mask;
  uint32_t i = key & mask;
  uint32_t perturb = key;
  int freeslot = -1;
  uint32_t idx = 0;
  for (;;) {
    uint32_t probes = (i + 9 <= mask) ? 9 : 0;
    bool found = false;
    ProbeGroup g;
    if (sizeof(T) == 1 && s.wide) g = probe_group_load(s.tab, i, probes);
    for (uint32_t p = 0; p <= probes; p++) {
      uint32_t e = (sizeof(T) == 1 && s.wide) ? probe_group_at(g, p) : (uint32_t)s.tab[i + p];
      if (e == 0) {
        idx = i + p;
        found = true;
        break;
      }
      if (e == key + 2) return;
      if (e == 1) freeslot = (int)(i + p);
    }
    if (found) break;
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
  if (freeslot >= 0) {
    s.used++;
    s.tab[freeslot] = (T)(key + 2);
    return;
  }
  s.fill++;
  s.used++;
  s.tab[idx] = (T)(key + 2);
  if (s.fill * 5 < mask * 3) return;
  set_resize(s, s.used * 4, keys);
}

template <typename T>
SSS_DEV bool set_remove(SetImg<T>& s, uint32_t key) {
  uint32_t mask = s.mask;
  uint32_t i = key & mask;
  uint32_t perturb = key;
  for (;;) {
    uint32_t probes = (i + 9 <= mask) ? 9 : 0;
    ProbeGroup g;
    if (sizeof(T) == 1 && s.wide) g = probe_group_load(s.tab, i, probes);
    for (uint32_t p = 0; p <= probes; p++) {
      uint32_t e = (sizeof(T) == 1 && s.wide) ? probe_group_at(g, p) : (uint32_t)s.tab[i + p];
      if (e == 0) return false;
      if (e == key + 2) {
        s.tab[i + p] = 1;
        s.used--;
        return true;
      }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

// marks key's slot of a byte table in HBM as a dummy (the table half of set_remove); any lane
SSS_DEV bool table_mark_dummy(uint8_t* tab, uint32_t mask, uint32_t key) {
  uint32_t i = key & mask, perturb = key;
  for (;;) {
    uint32_t probes = (i + 9 <= mask) ? 9 : 0;
    ProbeGroup g = probe_group_load(tab, i, probes);
    for (uint32_t p = 0; p <= probes; p++) {
      uint32_t en = probe_group_at(g, p);
      if (en == 0) return false;
      if (en == key + 2) {
        tab[i + p] = 1;
        return true;
      }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

template <typename T>
SSS_DEV uint32_t set_pop(SetImg<T>& s) {
  uint32_t i = s.finger & s.mask;
  while (s.tab[i] < 2) {
    i++;
    if (i > s.mask) i = 0;
  }
  uint32_t key = (uint32_t)s.tab[i] - 2;
  s.tab[i] = 1;
  s.used--;
  s.finger = i + 1;
  return key;
}

// A pool's record is 16 bytes: the set header, the outgoing commitment count and - while the table
// has 8 slots, which is nearly always - the table itself. pool_open fetches the record with one
// access and works on the table in LDS scratch; pool_close stores the record with one access.
// Tables that have grown live in the pool's slot of the overflow area (g_c.pool_tab).
SSS_DEV SetImg<uint8_t> pool_open(uint32_t key) {
  int p = pool_index(key);
  const uint4 rec = *(const uint4*)(g_c.pool_hdr + p);  // mask | fill << 16, used | commit_from << 16, tab8[0..3], tab8[4..7]
  SetImg<uint8_t> s;
  s.mask = rec.x & 0xFFFFu, s.fill = rec.x >> 16, s.used = rec.y & 0xFFFFu, s.finger = 0, s.aux = rec.y >> 16;
  s.big = g_c.pool_tab + (size_t)p * sss_pool_table_bytes(g_c.E), s.big_wide = true;
  s.small = g_sc.pool8;
  if (s.mask == 7) {
    *(uint2*)g_sc.pool8 = mk_u2(rec.z, rec.w);
    s.tab = g_sc.pool8, s.cap = 8, s.wide = false;
  } else
    s.tab = s.big, s.cap = 0xFFFFFFFFu, s.wide = true;
  return s;
}
// nothing else may have been opened in between (one scratch table), no commitment of the pool changed
SSS_DEV void pool_close(uint32_t key, const SetImg<uint8_t>& s) {
  SssPoolHdr* hd = g_c.pool_hdr + pool_index(key);
  const uint32_t w0 = s.mask | (s.fill << 16), w1 = (s.used & 0xFFFFu) | (s.aux << 16);
  if (s.mask == 7) {
    const uint2 t = *(const uint2*)s.small;  // (g_sc.pool8, or the second staging area's 8-slot scratch)
    *(uint4*)hd = mk_u4(w0, w1, t.x, t.y);
  } else
    *(uint4*)hd = mk_u4(w0, w1, 0u, 0u);  // the table lives in the overflow area; the inline bytes are kept clean
}
// The same with the whole wave, for a run of operations on one pool (the event batches): the pool's table comes
// into LDS with one access per lane whatever its size, lane 0 works on it there - dependent LDS accesses
// instead of dependent HBM ones - and it goes back the same way. pool_stage_in (all lanes) .. lane-0 section
// on the image it returns .. wave_sync .. pool_stage_out (all lanes). The staging area is setA + setB.
SSS_DEV uint8_t* pool_table_hbm(uint32_t key) { return g_c.pool_tab + (size_t)pool_index(key) * sss_pool_table_bytes(g_c.E); }
// one access per lane moves a whole pool table between HBM and LDS: 8 bytes each up to 64 executors (tables of at most 512
// bytes), 16 bytes in the wide instantiation (at most 1024)
#ifdef SSS_WIDE
typedef uint4 tabword_t;
SSS_DEV tabword_t tabword_zero() { return mk_u4(0u, 0u, 0u, 0u); }
#else
typedef uint2 tabword_t;
SSS_DEV tabword_t tabword_zero() { return mk_u2(0u, 0u); }
#endif
SSS_DEV bool tabword_differs(const tabword_t& a, const tabword_t& b) {
#ifdef SSS_WIDE
  return a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
#else
  return a.x != b.x || a.y != b.y;
#endif
}
SSS_DEV bool tabword_in(int lane) { return (uint32_t)lane * (uint32_t)sizeof(tabword_t) < (uint32_t)sss_pool_table_bytes(g_c.E); }
// whether two pool tables fit the two staging areas side by side (pool_pair_*): not with exactly 64 executors (512-byte tables
// against 256-byte areas); the wide instantiation's areas hold its largest table
#ifdef SSS_WIDE
SSS_DEV bool pair_staging_fits(int) { return true; }
#else
SSS_DEV bool pair_staging_fits(int E) { return E < 64; }
#endif
// `fetched`: this lane's bytes of the table as they are in HBM - pool_stage_out stores a lane's bytes only if they have changed
SSS_DEV SetImg<uint8_t> pool_stage_in(uint32_t key, tabword_t& fetched) {
  const int lane = wave_lane();
  const uint4 rec = *(const uint4*)(g_c.pool_hdr + pool_index(key));
  const uint32_t bytes = sss_pool_table_bytes(g_c.E);
  static_assert(2 * SSS_SET_TABLE >= 64 * sizeof(tabword_t), "a staged table lies in setA (+ setB, which follows it)");
  fetched = tabword_zero();
  if (tabword_in(lane)) fetched = ((const tabword_t*)pool_table_hbm(key))[lane], ((tabword_t*)g_sc.setA)[lane] = fetched;
  SetImg<uint8_t> s;
  s.mask = rec.x & 0xFFFFu, s.fill = rec.x >> 16, s.used = rec.y & 0xFFFFu, s.finger = 0, s.aux = rec.y >> 16;
  s.big = g_sc.setA, s.big_wide = false, s.small = g_sc.pool8;
  if (s.mask == 7) {
    if (lane == 0) *(uint2*)g_sc.pool8 = mk_u2(rec.z, rec.w);
    s.tab = g_sc.pool8, s.cap = 8, s.wide = false;
  } else
    s.tab = g_sc.setA, s.cap = bytes, s.wide = false;
  wave_sync();
  return s;
}
// `s`: lane 0's image after its operations (the other lanes' copies are stale)
SSS_DEV void pool_stage_out(uint32_t key, const SetImg<uint8_t>& s, const tabword_t& fetched) {
  const int lane = wave_lane();
  // every word of the area that differs from what was fetched goes back, not just the slots in use: the HBM copy then is byte for
  // byte what the one-operation-at-a-time code would have left (it works in place), dead slots included - and a table in which
  // one byte changed costs one store, not its whole area
  if (tabword_in(lane)) {
    const tabword_t now = ((const tabword_t*)g_sc.setA)[lane];
    if (tabword_differs(now, fetched)) ((tabword_t*)pool_table_hbm(key))[lane] = now;
  }
  if (lane == 0) pool_close(key, s);
  wave_sync();
}
// ---- set operations on a STAGED image with the whole wave (all lanes; every lane keeps the same header) ----
// An operation on a table of 16 slots or more examines a probe group - the home slot and the nine after it
// (LINEAR_PROBES) - with one lane per entry: one LDS access for the group, three ballots, and the rules of
// set_add_entry / set_discard_entry on bit masks (a key is found if it comes before the group's first empty slot;
// an addition reuses the LAST dummy seen before the first empty slot). Lane 0 writes the one byte that changes.
// About 20 instructions per operation, where the one-lane code pays an LDS round trip per entry. 8-slot tables (one
// probe per step) and resizes stay with the one-lane code (staged_sync_from_lane0 brings the lanes' headers back in step).
SSS_DEV void staged_fix_location(SetImg<uint8_t>& s) {  // where a staged image lives follows from its size
  const bool small = s.mask == 7;
  s.tab = small ? s.small : s.big, s.cap = small ? 8u : (uint32_t)sss_pool_table_bytes(g_c.E), s.wide = false;
}
SSS_DEV void staged_sync_from_lane0(SetImg<uint8_t>& s) {
  wave_sync();
  s.mask = wave_lane0_u32(s.mask), s.fill = wave_lane0_u32(s.fill), s.used = wave_lane0_u32(s.used);
  staged_fix_location(s);
}
// set_add; returns with the image updated (a resize included)
SSS_DEV void staged_add(SetImg<uint8_t>& s, uint32_t key) {
  const int lane = wave_lane();
  if (s.mask < 15) {  // (wave-uniform)
    if (lane == 0) set_add(s, key, lds_keys());
    staged_sync_from_lane0(s);
    return;
  }
  uint8_t* const tab = s.big;  // (the staging area the image was brought into: setA, or setB for a second pool)
  const uint32_t mask = s.mask;
  uint32_t i = key & mask, perturb = key;
  int freeslot = -1, idx = -1;
  for (;;) {
    const uint32_t probes = (i + 9 <= mask) ? 9u : 0u;
    const bool in = (uint32_t)lane <= probes;
    const uint32_t en = in ? (uint32_t)tab[i + (in ? lane : 0)] : 0xFFu;
    const uint64_t zm = wave_ballot(in && en == 0), mm = wave_ballot(in && en == key + 2), dm = wave_ballot(in && en == 1);
    const uint64_t before = zm ? (bit64(ctz64_nz(zm)) - 1) : ~0ull;  // the entries the scan reaches before it stops
    if (mm & before) return;  // already a member
    if (dm & before) freeslot = (int)i + 63 - __builtin_clzll(dm & before);
    if (zm) {
      idx = (int)i + ctz64_nz(zm);
      break;
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
  bool resize = false;
  if (freeslot >= 0) {
    s.used++;
    if (lane == 0) tab[freeslot] = (uint8_t)(key + 2);
  } else {
    s.fill++, s.used++;
    if (lane == 0) tab[idx] = (uint8_t)(key + 2);
    resize = s.fill * 5 >= mask * 3;
  }
  wave_sync();  // the byte is there before any lane looks at the table again
  if (resize) {
    if (lane == 0) set_resize(s, s.used * 4, lds_keys());
    staged_sync_from_lane0(s);
  }
}
// set_remove; returns whether the key was a member
SSS_DEV bool staged_remove(SetImg<uint8_t>& s, uint32_t key) {
  const int lane = wave_lane();
  if (s.mask < 15) {
    uint32_t was = 0;
    if (lane == 0) was = set_remove(s, key) ? 1u : 0u;
    was = wave_lane0_u32(was);
    staged_sync_from_lane0(s);
    return was != 0;
  }
  uint8_t* const tab = s.big;
  const uint32_t mask = s.mask;
  uint32_t i = key & mask, perturb = key;
  for (;;) {
    const uint32_t probes = (i + 9 <= mask) ? 9u : 0u;
    const bool in = (uint32_t)lane <= probes;
    const uint32_t en = in ? (uint32_t)tab[i + (in ? lane : 0)] : 0xFFu;
    const uint64_t zm = wave_ballot(in && en == 0), mm = wave_ballot(in && en == key + 2);
    const uint64_t before = zm ? (bit64(ctz64_nz(zm)) - 1) : ~0ull;
    if (mm & before) {
      if (lane == 0) tab[i + (uint32_t)ctz64_nz(mm & before)] = 1;
      s.used--;
      wave_sync();
      return true;
    }
    if (zm) return false;
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}
SSS_DEV int pool_size(uint32_t key) { return key == POOL_NONE ? 0 : (int)g_c.pool_hdr[pool_index(key)].used; }
SSS_DEV int pool_commit_from(uint32_t key) { return key == POOL_NONE ? 0 : (int)g_c.pool_hdr[pool_index(key)].commit_from; }

// ------------------------------------------------------------------------------------------
// tracker (lane 0)
// ------------------------------------------------------------------------------------------

SSS_DEV int trk_source_job_id() {  // TRK:101-105
  uint32_t k = H.curr_source;
  if (k == POOL_NONE || k == POOL_COMMON) return -1;
  return key_job(k);
}

SSS_DEV void publish_scan_inputs() {
  g_sc.m_n_active = H.n_active;
  g_sc.m_src_job = trk_source_job_id();
}

SSS_DEV int trk_num_committable() {  // TRK:107-113
  uint32_t k = H.curr_source;
  if (k == POOL_NONE) return 0;
  int p = pool_index(k);
  int n = (int)g_c.pool_hdr[p].used - (int)g_c.pool_hdr[p].commit_from;
  CHECK(n >= 0);
  return n;
}

// executor demand bookkeeping: sat bit of stage (j, s) <=> remaining - (moving_to + commit_to) <= 0 (ENV:566-582)
SSS_DEV void update_sat(const JobView& v, int s) {
  const SssStage st = v.st[s];
  const int demand = (int)st.remaining - ((int)st.moving_to + (int)st.commit_to);
  const uint64_t m = v.job->sat_mask;
  v.job->sat_mask = demand <= 0 ? (m | bit64(s)) : (m & ~bit64(s));
}
SSS_DEV void update_sat(int j, int s) { update_sat(jobview(j), s); }

SSS_DEV void add_supply(int job, int d) {
  if (job < 0) {
    H.supply_none += d;
    CHECK(H.supply_none >= 0);
  } else {
    SssJob* jp = jobp(job);
    int v = (int)jp->supply + d;
    CHECK(v >= 0);
    jp->supply = (int16_t)v;
  }
}

SSS_DEV void trk_add_commitment(int n, uint32_t dst) {  // TRK:148-157, 226-238
  PROF3(1);
  uint32_t src = H.curr_source;
  CHECK(src != POOL_NONE);
  if (src == POOL_NONE) return;
  SssHot& hot = g_hot;
  int i;
  for (i = 0; i < H.n_commits; i++)
    if (hot.c_src[i] == src && hot.c_dst[i] == dst) break;
  if (i < H.n_commits)
    hot.c_n[i] = (int16_t)(hot.c_n[i] + n);
  else {
    CHECK(i < SSS_MAX_EXEC);
    if (i >= SSS_MAX_EXEC) return;
    hot.c_src[i] = src, hot.c_dst[i] = dst, hot.c_n[i] = (int16_t)n, hot.c_seq[i] = H.commit_seq++;
    H.n_commits = i + 1;
  }
  int ps = pool_index(src);
  g_c.pool_hdr[ps].commit_from = (int16_t)(g_c.pool_hdr[ps].commit_from + n);
  CHECK((int)g_c.pool_hdr[ps].used >= (int)g_c.pool_hdr[ps].commit_from);
  int dj = key_job(dst), ds = key_stage(dst);
  if (ds >= 0) {
    const JobView v = jobview(dj);
    v.st[ds].commit_to = (uint8_t)(v.st[ds].commit_to + n);
    update_sat(v, ds);
  }
  if (dj != key_job(src)) add_supply(dj, n);
}

// returns the source pool key (TRK:159-176, 240-251)
SSS_DEV uint32_t trk_remove_commitment(int e, uint32_t dst) {
  PROF3(2);
  SssHot& hot = g_hot;
  uint32_t src = hot.ex_loc[e];
  CHECK(src != POOL_NONE);
  int i;
  for (i = 0; i < H.n_commits; i++)
    if (hot.c_src[i] == src && hot.c_dst[i] == dst) break;
  CHECK(i < H.n_commits);
  if (i >= H.n_commits) return src;
  hot.c_n[i] = (int16_t)(hot.c_n[i] - 1);
  int ps = pool_index(src);
  g_c.pool_hdr[ps].commit_from = (int16_t)(g_c.pool_hdr[ps].commit_from - 1);
  CHECK(g_c.pool_hdr[ps].commit_from >= 0);
  int dj = key_job(dst), ds = key_stage(dst);
  if (ds >= 0) {
    const JobView v = jobview(dj);
    const int c = (int)v.st[ds].commit_to - 1;
    CHECK(c >= 0);
    v.st[ds].commit_to = (uint8_t)c;
    update_sat(v, ds);
  }
  if (hot.c_n[i] == 0) {  // dict.pop(dst): swap-remove, order lives in c_seq
    int last = H.n_commits - 1;
    hot.c_src[i] = hot.c_src[last], hot.c_dst[i] = hot.c_dst[last], hot.c_n[i] = hot.c_n[last], hot.c_seq[i] = hot.c_seq[last];
    H.n_commits = last;
  }
  if (dj != key_job(src)) add_supply(dj, -1);
  return src;
}

// first-inserted live destination of `src`, POOL_NONE if none (TRK:178-183)
SSS_DEV uint32_t trk_peek_commitment(uint32_t src) {
  const SssHot& hot = g_hot;
  uint32_t best = 0xFFFFFFFFu, dst = POOL_NONE;
  for (int i = 0; i < H.n_commits; i++)
    if (hot.c_src[i] == src && hot.c_seq[i] < best) best = hot.c_seq[i], dst = hot.c_dst[i];
  return dst;
}

// the same with the whole wave (all lanes, the same arguments on every lane; `on` = false: no hit): the first-inserted live entry of
// source `src` - among those to the common pool only, with `only_common` - found with one ballot over the list, one entry per lane
// (two in the wide instantiation: the list has one entry per executor at most). ci = -1: none.
struct CommitHit {
  int ci;
  uint32_t dst;
  int num;
};
SSS_DEV CommitHit commit_first_wave(uint32_t src, bool only_common, bool on, int n_commits) {
  const int lane = wave_lane();
  bool mine = on && lane < n_commits && g_hot.c_src[lane] == src && (!only_common || g_hot.c_dst[lane] == POOL_COMMON);
  uint32_t seq = g_hot.c_seq[lane], dst = g_hot.c_dst[lane];
  int num = g_hot.c_n[lane], idx = lane;
#ifdef SSS_WIDE
  {
    const int l2 = lane + 64;
    const bool mine2 = on && l2 < n_commits && g_hot.c_src[l2] == src && (!only_common || g_hot.c_dst[l2] == POOL_COMMON);
    const uint32_t seq2 = g_hot.c_seq[l2];
    if (mine2 && (!mine || seq2 < seq)) seq = seq2, dst = g_hot.c_dst[l2], num = g_hot.c_n[l2], idx = l2;
    mine = mine || mine2;
  }
#endif
  CommitHit h;
  h.ci = -1, h.dst = POOL_NONE, h.num = 0;
  const uint64_t cm = wave_ballot(mine);
  if (cm == 0) return h;
  int wl = ctz64_nz(cm);
  if (cm & (cm - 1)) {
    const uint32_t best = wave_min_u32(mine ? seq : 0xFFFFFFFFu);
    wl = ctz64_nz(wave_ballot(mine && seq == best));
  }
  h.ci = (int)wave_readlane_u32((uint32_t)idx, wl), h.dst = wave_readlane_u32(dst, wl), h.num = (int)wave_readlane_u32((uint32_t)num, wl);
  return h;
}

// ---- 8-slot set images held in a register (mask == 7: LINEAR_PROBES never applies, i + 9 > mask) ----
SSS_DEV uint32_t t8_get(uint64_t t, uint32_t i) { return (uint32_t)(t >> (8 * i)) & 0xFFu; }
SSS_DEV uint64_t t8_set(uint64_t t, uint32_t i, uint32_t v) { return (t & ~(0xFFull << (8 * i))) | ((uint64_t)v << (8 * i)); }
SSS_DEV bool set8_remove(uint64_t& t, uint32_t& used, uint32_t key) {  // set_remove
  uint32_t i = key & 7, perturb = key;
  for (;;) {
    uint32_t e = t8_get(t, i);
    if (e == 0) return false;
    if (e == key + 2) {
      t = t8_set(t, i, 1);
      used--;
      return true;
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & 7;
  }
}
// set_add; returns true when the table has to be resized afterwards (fill * 5 >= mask * 3)
SSS_DEV bool set8_add(uint64_t& t, uint32_t& fill, uint32_t& used, uint32_t key) {
  uint32_t i = key & 7, perturb = key;
  int freeslot = -1;
  for (;;) {
    uint32_t e = t8_get(t, i);
    if (e == 0) break;
    if (e == key + 2) return false;
    if (e == 1) freeslot = (int)i;
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & 7;
  }
  if (freeslot >= 0) {
    used++;
    t = t8_set(t, (uint32_t)freeslot, key + 2);
    return false;
  }
  fill++, used++;
  t = t8_set(t, i, key + 2);
  return fill * 5 >= 7 * 3;
}

// ------------------------------------------------------------------------------------------
// Two pools at once through the LDS staging areas (all lanes). An executor that changes pools touches two set
// images - the one it leaves and the one it enters - and on lane 0 every one of them is a chain of dependent HBM
// round trips: the record, then (tables beyond 8 slots) the probe group, then the stores. Here both records and
// both tables are fetched with ONE round trip - every lane loads 8 bytes of each table whatever the records will
// say about their sizes - land in setA / setB (+ pool8 / pool8b for 8-slot tables), are operated on with the whole
// wave (staged_add / staged_remove) and go back with one store per lane. Tables of up to 256 bytes (fewer than
// 64 executors). pool_pair_fetch (loads in flight) .. pool_pair_stage (in LDS, images ready) .. operations ..
// pool_pair_flush.
// ------------------------------------------------------------------------------------------
struct PoolPairRegs {
  uint4 rec_a, rec_b;      // the two 16-byte records
  tabword_t tab_a, tab_b;  // this lane's bytes of either table
};
SSS_DEV PoolPairRegs pool_pair_fetch(uint32_t key_a, uint32_t key_b, bool has_b) {
  const int lane = wave_lane();
  PoolPairRegs r;
  r.rec_a = *(const uint4*)(g_c.pool_hdr + pool_index(key_a));
  r.rec_b = mk_u4(7u, 0u, 0u, 0u);
  r.tab_a = tabword_zero(), r.tab_b = r.tab_a;
  const bool in = tabword_in(lane);
  if (in) r.tab_a = ((const tabword_t*)pool_table_hbm(key_a))[lane];
  if (has_b) {
    r.rec_b = *(const uint4*)(g_c.pool_hdr + pool_index(key_b));
    if (in) r.tab_b = ((const tabword_t*)pool_table_hbm(key_b))[lane];
  }
  return r;
}
// An image of the pair: the set header, and the table - in a register, the same on every lane, while it has 8 slots (set8_add /
// set8_remove: no LDS, no barrier; at BASELINE config 2 nearly every pool, at config 3 the pools of jobs with few executors),
// else in its staging area, operated on with the whole wave (staged_add / staged_remove).
struct PairImg {
  SetImg<uint8_t> s;
  uint64_t t8;
  uint32_t mask_before;  // the record's mask when it was fetched
  tabword_t fetched;     // this lane's bytes of the table area as they are in HBM
};
SSS_DEV PairImg pool_pair_image(const uint4 rec, uint8_t* area, uint8_t* small8) {
  PairImg p;
  p.s.mask = rec.x & 0xFFFFu, p.s.fill = rec.x >> 16, p.s.used = rec.y & 0xFFFFu, p.s.finger = 0, p.s.aux = rec.y >> 16;
  p.s.big = area, p.s.big_wide = false, p.s.small = small8;
  staged_fix_location(p.s);
  p.t8 = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
  p.mask_before = p.s.mask;
  return p;
}
// (ends with a barrier: the tables beyond 8 slots are in LDS)
SSS_DEV void pool_pair_stage(const PoolPairRegs& r, bool has_b, PairImg& a, PairImg& b) {
  const int lane = wave_lane();
  const bool in = tabword_in(lane);
  if (in) ((tabword_t*)g_sc.setA)[lane] = r.tab_a;
  a = pool_pair_image(r.rec_a, g_sc.setA, g_sc.pool8);
  a.fetched = r.tab_a;
  if (has_b && in) ((tabword_t*)g_sc.setB)[lane] = r.tab_b;
  b = pool_pair_image(r.rec_b, g_sc.setB, g_sc.pool8b);
  b.fetched = r.tab_b;
  wave_sync();
}
// the image has just been through a resize on lane 0 (its header is in step again): an 8-slot result goes to the register
SSS_DEV void pair_after_resize(PairImg& p) {
  if (p.s.mask == 7) {
    const uint2 t = *(const uint2*)p.s.small;
    p.t8 = (uint64_t)t.x | ((uint64_t)t.y << 32);
    wave_sync();  // every lane has read the scratch before the next resize may write it (8-slot operations have no barrier of their own)
  }
}
SSS_DEV void pair_add(PairImg& p, uint32_t key) {  // set_add (all lanes)
  if (p.s.mask == 7) {
    if (!set8_add(p.t8, p.s.fill, p.s.used, key)) return;
    // fill * 5 >= mask * 3: set_table_resize(used * 4) - through the 8-slot scratch, on lane 0; the result may have 8 slots or more
    if (wave_lane() == 0) {
      *(uint2*)p.s.small = mk_u2((uint32_t)p.t8, (uint32_t)(p.t8 >> 32));
      p.s.tab = p.s.small, p.s.cap = 8, p.s.wide = false;
      set_resize(p.s, p.s.used * 4, lds_keys());
    }
    staged_sync_from_lane0(p.s);
    pair_after_resize(p);
    return;
  }
  const uint32_t m0 = p.s.mask;
  staged_add(p.s, key);
  if (p.s.mask != m0) pair_after_resize(p);
}
SSS_DEV bool pair_remove(PairImg& p, uint32_t key) {  // set_remove (all lanes)
  if (p.s.mask == 7) return set8_remove(p.t8, p.s.used, key);
  return staged_remove(p.s, key);
}
// n members leave the pool at once (all lanes): list[from .. to) are their ids. Removals commute - a removal leaves a dummy, no
// probe chain changes - so on a staged table every member's own lane finds and marks its slot (the table is in LDS: the lanes'
// probe loops run side by side); an 8-slot image in the register is walked by every lane alike.
SSS_DEV void pair_remove_many(PairImg& p, const uint8_t* list, int from, int to) {
  const int n = to - from;
  if (p.s.mask == 7) {
    for (int i = from; i < to; i++) {
      bool was = set8_remove(p.t8, p.s.used, (uint32_t)list[i]);
      CHECK(was);
    }
    return;
  }
  for (int q0 = wave_lane(); q0 < n; q0 += 64) {  // (more than 64 members: the wide instantiation)
    uint8_t* const tab = p.s.big;
    const uint32_t key = list[from + q0], mask = p.s.mask;
    uint32_t i = key & mask, perturb = key;
    bool done = false;
    for (int guard = 0; guard < 64 && !done; guard++) {
      const uint32_t probes = (i + 9 <= mask) ? 9u : 0u;
      for (uint32_t q = 0; q <= probes && !done; q++) {
        const uint32_t en = tab[i + q];
        if (en == key + 2) tab[i + q] = 1, done = true;
        else if (en == 0) guard = 64;  // (not a member: reported below)
      }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
    CHECK(done);
  }
  p.s.used -= (uint32_t)n;
  wave_sync();
}
// one image back to HBM: the record, and the table area unless the image had 8 slots before and has 8 slots now (the
// area then holds what was fetched). Like pool_stage_out every word of the area that has changed goes back, so that the HBM bytes
// are what the one-operation-at-a-time code leaves, dead slots included (round 4 stored the whole area: 256 bytes where one byte
// had changed, +3.7 MB per config-2 step launch).
SSS_DEV void pool_pair_flush_one(uint32_t key, const PairImg& p) {
  const int lane = wave_lane();
  if ((p.mask_before != 7 || p.s.mask != 7) && tabword_in(lane)) {
    const tabword_t now = ((const tabword_t*)p.s.big)[lane];
    if (tabword_differs(now, p.fetched)) ((tabword_t*)pool_table_hbm(key))[lane] = now;
  }
  if (lane == 0) {
    const uint32_t w0 = p.s.mask | (p.s.fill << 16), w1 = (p.s.used & 0xFFFFu) | (p.s.aux << 16);
    const bool small = p.s.mask == 7;  // (larger tables live in the overflow area; the inline bytes are kept clean)
    *(uint4*)(g_c.pool_hdr + pool_index(key)) = mk_u4(w0, w1, small ? (uint32_t)p.t8 : 0u, small ? (uint32_t)(p.t8 >> 32) : 0u);
  }
}

SSS_DEV void trk_move_executor_to_pool(int e, uint32_t new_pool, bool send) {  // TRK:188-222
  PROF3(3);
  SssHot& hot = g_hot;
  uint32_t old = hot.ex_loc[e];
  const bool has_old = old != POOL_NONE, has_new = !send;
  const bool same = has_old && has_new && old == new_pool;
  // both records are fetched up front (one round trip); 8-slot images are worked on in registers, larger
  // ones in their table in the overflow area, with the header taken from the record already fetched
  SssPoolHdr* ho = g_c.pool_hdr + (has_old ? pool_index(old) : 0);
  SssPoolHdr* hn = g_c.pool_hdr + (has_new ? pool_index(new_pool) : 0);
  uint4 ro = mk_u4(7u, 0u, 0u, 0u), rn = ro;
  if (has_old) ro = *(const uint4*)ho;
  if (has_new && !same) rn = *(const uint4*)hn;
  STAT(100, 1), STAT(101, has_old), STAT(102, has_old && (ro.x & 0xFFFFu) != 7), STAT(103, has_new), STAT(104, has_new && ((same ? ro.x : rn.x) & 0xFFFFu) != 7), STAT(105, same);
  if (has_old) {
    if ((ro.x & 0xFFFFu) == 7) {
      uint64_t t = (uint64_t)ro.z | ((uint64_t)ro.w << 32);
      uint32_t used = ro.y & 0xFFFFu;
      bool was = set8_remove(t, used, (uint32_t)e);
      CHECK(was);
      ro.y = (ro.y & 0xFFFF0000u) | used, ro.z = (uint32_t)t, ro.w = (uint32_t)(t >> 32);
    } else {
      bool was = table_mark_dummy(g_c.pool_tab + (size_t)pool_index(old) * sss_pool_table_bytes(g_c.E), ro.x & 0xFFFFu, (uint32_t)e);
      CHECK(was);
      ro.y = (ro.y & 0xFFFF0000u) | (((ro.y & 0xFFFFu) - 1u) & 0xFFFFu);  // used--
    }
    if (!same) *(uint4*)ho = ro;
    hot.ex_loc[e] = POOL_NONE;
  }
  if (has_new) {
    if (same) rn = ro;
    hot.ex_loc[e] = new_pool;
    SetImg<uint8_t> s;
    s.small = g_sc.pool8, s.big = g_c.pool_tab + (size_t)pool_index(new_pool) * sss_pool_table_bytes(g_c.E), s.big_wide = true;
    s.mask = rn.x & 0xFFFFu, s.fill = rn.x >> 16, s.used = rn.y & 0xFFFFu, s.finger = 0, s.aux = rn.y >> 16;
    if (s.mask == 7) {
      uint64_t t = (uint64_t)rn.z | ((uint64_t)rn.w << 32);
      if (set8_add(t, s.fill, s.used, (uint32_t)e)) {
        // set_table_resize(used * 4): through the scratch table, the result may have more than 8 slots
        *(uint2*)g_sc.pool8 = mk_u2((uint32_t)t, (uint32_t)(t >> 32));
        s.tab = g_sc.pool8, s.cap = 8, s.wide = false;
        set_resize(s, s.used * 4, lds_keys());
        pool_close(new_pool, s);
      } else
        *(uint4*)hn = mk_u4(7u | (s.fill << 16), (rn.y & 0xFFFF0000u) | s.used, (uint32_t)t, (uint32_t)(t >> 32));
    } else {
      s.tab = s.big, s.cap = 0xFFFFFFFFu, s.wide = true;
      set_add(s, (uint32_t)e, lds_keys());
      pool_close(new_pool, s);
    }
    return;
  }
  int nj = key_job(new_pool), ns = key_stage(new_pool);
  CHECK(nj >= 0 && ns >= 0);  // "can only send executors to stages"
  {
    const JobView v = jobview(nj);
    v.st[ns].moving_to = (uint8_t)(v.st[ns].moving_to + 1);
    update_sat(v, ns);
    const int sup = (int)v.job->supply + 1;  // add_supply(nj, 1)
    v.job->supply = (int16_t)sup;
  }
  int oj = key_job(old);
  CHECK(oj != nj);
  if (oj >= 0) add_supply(oj, -1);
}

// ------------------------------------------------------------------------------------------
// jobs / stages (lane 0)
// ------------------------------------------------------------------------------------------

SSS_DEV void job_attach_executor(int j, int e) {  // JOB:81-84
  CHECK(g_hot.ex_task_stage[e] < 0);
  SssJob* jp = jobp(j);
  jp->local_mask = local_with(jp->local_mask, e);
  g_hot.ex_job[e] = (int16_t)j;
}
SSS_DEV void job_detach_executor(int j, int e) {  // JOB:86-89
  SssJob* jp = jobp(j);
  CHECK(local_has(jp->local_mask, e));
  jp->local_mask = local_without(jp->local_mask, e);
  g_hot.ex_job[e] = -1;
  g_hot.ex_task_stage[e] = -1;
}
SSS_DEV bool stage_completed(const SssStage& st) { return st.remaining == 0 && st.executing == 0; }  // STG:41-43

// JOB:65-73,100-128: stage s of job j completed; returns whether the frontier gained stages
SSS_DEV bool job_record_stage_completion(int j, int s) {
  PROF3(4);
  SssJob& job = (*jobp(j));
  CHECK((job.active_mask & bit64(s)) && (job.frontier_mask & bit64(s)));
  uint64_t active = job.active_mask & ~bit64(s);
  job.active_mask = active;
  uint64_t frontier = job.frontier_mask & ~bit64(s);
  // completed stages == stages that are no longer active
  uint64_t all = job.n_stages >= 64 ? ~0ull : (bit64(job.n_stages) - 1);
  uint64_t completed = all & ~active;
  uint64_t children = g_c.pk.stage_child_mask[job.gs_base + s];
  uint64_t newm = 0;
  uint64_t cand = children & active;
  while (cand) {
    int ch = ctz64(cand);
    cand &= cand - 1;
    uint64_t parents = g_c.pk.stage_parent_mask[job.gs_base + ch];
    if ((parents & ~completed) == 0) newm |= bit64(ch);
  }
  job.frontier_mask = frontier | newm;
  H.graph_version++;  // a node left the active subgraph
  return newm != 0;
}

// ------------------------------------------------------------------------------------------
// data sampler: task durations (lane 0)
// ------------------------------------------------------------------------------------------

// TPCHDataSampler._init_executor_intervals (TPCH:237-262) in closed form for exec_cap <= 100: the
// row of `num_local_executors` = n is (5,5) for n <= 5, (n,n) when n is one of the executor
// levels, else the two enclosing levels. Returns indices into {5,10,20,40,50,60,80,100}.
SSS_DEV int exec_level_value(int i) {
  const uint64_t packed = 5ull | (10ull << 8) | (20ull << 16) | (40ull << 24) | (50ull << 32) | (60ull << 40) | (80ull << 48) | (100ull << 56);
  return (int)((packed >> (8 * i)) & 0xFF);
}
SSS_DEV void executor_interval(int n, int& li, int& ri) {
  // index of the first level >= n (levels above 80 only matter for n > 80)
  ri = (n > 5) + (n > 10) + (n > 20) + (n > 40) + (n > 50) + (n > 60) + (n > 80);
  li = (n <= 5 || n == exec_level_value(ri)) ? ri : ri - 1;
#ifdef SSS_WIDE
  // exec_cap > 100 (TPCH:258-260): rows 101 .. exec_cap - 1 are (100, 100); row exec_cap itself keeps np.zeros' (0, 0), and
  // key 0 is in no first_wave dict, so the stage's largest level is taken (TPCH:231-233): "level" 8, served by SssPackDev::eff0
  if (n > 100) li = ri = (n == g_c.E ? 8 : 7);
#endif
}
// the resolved duration list of (pack stage, executor level index, executor mode): sss_host.h sss_build_eff
SSS_DEV const int32_t* eff_row(const int32_t* eff, int gs, int li, int mode) {
#ifdef SSS_WIDE
  if (li == 8) return g_c.pk.eff0 + ((size_t)gs * 3 + mode) * 4;
#endif
  return eff + (((size_t)gs * 8 + li) * 3 + mode) * 4;
}

// TPCH:75-106, 216-235. Which list is sampled is a pure function of (stage, executor level, executor
// mode): the level substitution (`executor_key not in first_wave` -> max key, TPCH:231-233) and the
// exception-driven fallback chain (TPCH:88-106; a missing key or an empty list raises before any
// draw) are resolved once per template on the host into `eff` (sss_host.h: sss_build_eff), so the
// device does one descriptor load, the draw, and one value load.
SSS_DEV double task_duration(const SssJob* job, int s, int e) {
  PROF3(5);
  int gs = job->gs_base + s;
  int n_local = local_count(job->local_mask);
  CHECK(n_local > 0 && n_local <= g_c.E);
  if (n_local <= 0 || n_local > g_c.E) return 0.0;
  int li, ri;
  executor_interval(n_local, li, ri);
  if (li != ri) {
    double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
    int rand_pt = 1 + (int)(rng_random() * (right - left));
    if (!((double)rand_pt <= (double)n_local - left)) li = ri;
  }
  int task_stage = g_hot.ex_task_stage[e];
  int mode = task_stage < 0 ? 0 : (task_stage == s ? 1 : 2);  // idle / same stage id (TPCH:95) / other
  const int32_t* d = eff_row(g_c.pk.eff, gs, li, mode);
  int off = d[0], lenw = d[1];
  int len = lenw & 0x3FFFFFFF;
  if (len == 0) {
    FAIL(SSS_ERR_NO_DURATION);
    return 0.0;
  }
  uint32_t i = rng_integers((uint32_t)len);
  double v = (double)g_c.pk.durations[off + (int)i];
  if (lenw >> 30) v += g_c.P.warmup_delay;
  return v;
}

// ------------------------------------------------------------------------------------------
// schedulable-stage search, serial flavour (lane 0): single jobs and the backup search
// ------------------------------------------------------------------------------------------

// stages of job j that are active, not selected this round and ready (ENV:533-555); `pass`
// filter (job == source or supply < E, ENV:526-531) applied by the caller
SSS_DEV uint64_t ready_mask_of_job(const SssJob& job, bool first_only) {
  uint64_t cand = job.active_mask & ~job.selected_mask & ~job.sat_mask;
  uint64_t m = 0;
  while (cand) {
    int s = ctz64(cand);
    cand &= cand - 1;
    uint64_t parents = g_c.pk.stage_parent_mask[job.gs_base + s];
    if ((parents & ~job.sat_mask) == 0) {
      m |= bit64(s);
      if (first_only) break;
    }
  }
  return m;
}

SSS_DEV bool job_passes_filter(int j, int source_job_id) {
  return j == source_job_id || (int)(*jobp(j)).supply < g_c.E;
}

// ENV:821-845 -> (job, stage) or job = -1
SSS_DEV void find_backup_stage(int e, int& out_j, int& out_s) {
  PROF3(6);
  out_j = -1, out_s = -1;
  int ejob = g_hot.ex_job[e];
  CHECK(ejob >= 0);
  if (ejob < 0) return;
  // `if not source_job_id` (ENV:521): job id 0 is falsy and gets replaced by the tracker's source
  int src = ejob <= 0 ? trk_source_job_id() : ejob;
  if (job_passes_filter(ejob, src)) {
    uint64_t m = ready_mask_of_job((*jobp(ejob)), true);
    if (m) {
      out_j = ejob, out_s = ctz64(m);
      return;
    }
  }
  // other jobs; an empty list is falsy and means "all active jobs" (ENV:518-519)
  bool ejob_active = (*jobp(ejob)).active_mask != 0;
  int n_others = H.n_active - (ejob_active ? 1 : 0);
  for (int a = 0; a < H.n_active; a++) {
    int j = lds_active()[a];
    if (n_others > 0 && j == ejob) continue;
    if (!job_passes_filter(j, src)) continue;
    uint64_t m = ready_mask_of_job((*jobp(j)), true);
    if (m) {
      out_j = j, out_s = ctz64(m);
      return;
    }
  }
}

// ------------------------------------------------------------------------------------------
// executor movement (lane 0)
// ------------------------------------------------------------------------------------------

// event word: kind (bits 0-7) | stage (8-13) | LDS slot of the job, valid within a launch, 127 = none
// (14-20) | job (21-31). The slot rides along so that the handler of a popped TASK_FINISHED does not
// have to look it up (one dependent LDS round trip less per event); env_begin fills it for the events
// that are pending when a launch starts, env_end clears it so that the HBM image does not depend on
// how slots were handed out.
#define INFO_SLOT_NONE 127u
SSS_DEV uint32_t ev_info(int kind, int j, int s, uint32_t slot) {
  return (uint32_t)kind | ((uint32_t)s << 8) | ((slot > 63u ? INFO_SLOT_NONE : slot) << 14) | ((uint32_t)j << 21);
}
SSS_DEV int info_kind(uint32_t i) { return (int)(i & 0xFF); }
SSS_DEV int info_stage(uint32_t i) { return (int)((i >> 8) & 0x3F); }
SSS_DEV uint32_t info_slot(uint32_t i) { return (i >> 14) & 0x7F; }
SSS_DEV int info_job(uint32_t i) { return (int)(i >> 21); }
SSS_DEV uint32_t info_with_slot(uint32_t i, uint32_t slot) { return (i & ~(0x7Fu << 14)) | ((slot > 63u ? INFO_SLOT_NONE : slot) << 14); }

SSS_DEV int cache_acquire(int j);
SSS_DEV void push_event(int e, double t, int kind, int j, int s) {  // EVQ:34-35
  SssHot& hot = g_hot;
  CHECK((hot.ev[e].info & 0xFF) == EV_NONE);
  int slot = cache_acquire(j);  // a job with a pending event holds a cache slot (if there is one to have)
  if (slot != SLOT_NONE) lds_slot_ref()[slot]++;
  SssEvSlot sl;
  sl.t = t, sl.seq = H.counter++, sl.info = ev_info(kind, j, s, (uint32_t)slot);
  hot.ev[e] = sl;
}

SSS_DEV void execute_next_task(int e, int j, int s) {  // ENV:584-615
  PROF3(7);
  const JobView v = jobview(j);  // (valid up to push_event, which may hand the job a cache slot)
  SssStage st = v.st[s];
  CHECK(st.remaining > 0 && g_hot.ex_job[e] == j && !g_hot.ex_executing[e]);
  st.remaining = (st.remaining - 1);  // STG:53-58
  st.executing = (int16_t)(st.executing + 1);
  v.st[s] = st;
  if (st.remaining == 0) v.job->sat_count = (int16_t)(v.job->sat_count + 1);
  {
    const int demand = (int)st.remaining - ((int)st.moving_to + (int)st.commit_to);  // update_sat on the values at hand
    const uint64_t m = v.job->sat_mask;
    v.job->sat_mask = demand <= 0 ? (m | bit64(s)) : (m & ~bit64(s));
  }
  double d = task_duration(v.job, s, e);
  g_hot.ex_task_stage[e] = (int8_t)s;
  g_hot.ex_executing[e] = 1;
  v.dur[s] = (float)d;
  push_event(e, H.wall_time + d, EV_TASK_FINISHED, j, s);
}

SSS_DEV void send_executor(int e, int j, int s) {  // ENV:617-637
  PROF3(8);
  CHECK(!g_hot.ex_executing[e] && g_hot.ex_job[e] != j);
  trk_move_executor_to_pool(e, key_stage_pool(j, s), true);
  int oj = g_hot.ex_job[e];
  if (oj >= 0) job_detach_executor(oj, e);
  push_event(e, H.wall_time + g_c.P.moving_delay, EV_EXECUTOR_READY, j, s);
}

// ENV:745-782 for an explicit executor list of one
SSS_DEV void move_idle_executor(uint32_t src, int e) {
  if (src == POOL_NONE) src = H.curr_source;
  CHECK(src != POOL_NONE);
  if (src == POOL_NONE || src == POOL_COMMON) return;
  int j = key_job(src), s = key_stage(src);
  const SssJob* jp = jobp(j);
  bool is_sat = (int)jp->sat_count == (int)jp->n_stages;  // JOB:53-55
  if (s < 0 && !is_sat) return;
  uint32_t dst = is_sat ? POOL_COMMON : key_job_pool(j);
  trk_move_executor_to_pool(e, dst, false);
  if (dst == POOL_COMMON) job_detach_executor(j, e);
}

// set(id for id in pool.copy() if not executing) into sc->setB (ENV:714-728)
// all lanes: which executors sit idle in the source pool (a pool's members are the executors located in it)
SSS_DEV void publish_idle_mask() {
  int lane = wave_lane();
  uint32_t key = g_hot.h.curr_source;
  uint64_t m = wave_ballot(lane < g_c.E && key != POOL_NONE && g_hot.ex_loc[lane] == key && !g_hot.ex_executing[lane]);
#ifdef SSS_WIDE
  uint64_t mh = wave_ballot(lane + 64 < g_c.E && key != POOL_NONE && g_hot.ex_loc[lane + 64] == key && !g_hot.ex_executing[lane + 64]);
  if (lane == 0) g_sc.idle_mask_hi = mh;
#endif
  if (lane == 0) g_sc.idle_key = key, g_sc.idle_mask = m, g_sc.idle_valid = 1;
}
SSS_DEV SetImg<uint8_t> get_idle_source_executors(uint32_t key) {
  PROF3(9);
  SetImg<uint8_t> out;
  out.tab = g_sc.setB;
  for (int i = 0; i < 8; i++) out.tab[i] = 0;
  out.mask = 7, out.fill = 0, out.used = 0, out.finger = 0, out.cap = 0xFFFFFFFFu, out.big = nullptr, out.small = nullptr, out.wide = false;
  if (key == POOL_NONE) return out;
  if (g_sc.idle_valid && g_sc.idle_key == key) {
    // at most one idle executor: the set built from the pool's copy is {e} whatever the iteration order
    // (one add into a fresh 8-slot table) - the usual case when executors are released one at a time
    uint64_t m = g_sc.idle_mask;
    g_sc.idle_valid = 0;
#ifdef SSS_WIDE
    const uint64_t mh = g_sc.idle_mask_hi;
#else
    const uint64_t mh = 0;
#endif
    const uint32_t n_idle = (uint32_t)(popc64(m) + popc64(mh));
    if (n_idle <= 1) {
      if (n_idle) {
        int e = m ? ctz64(m) : 64 + ctz64(mh);
        out.tab[e & 7] = (uint8_t)(e + 2);
        out.fill = out.used = 1;
      }
      return out;
    }
    // 19 or more: whatever order they are added in, the set grows 8 -> 32 (5th key) -> 128 slots (19th key,
    // set_table_resize(76)) - and on to 512 slots with the 77th (set_table_resize(308)) - where every executor id sits in
    // its home slot: the image is the same for every order
    if (n_idle >= 19) {
      const int slots = n_idle >= 77 ? 512 : 128;
      for (int i = 0; i < slots / 8; i++) ((uint2*)out.tab)[i] = mk_u2(0u, 0u);
      for (uint64_t r = m; r; r &= r - 1) out.tab[ctz64(r)] = (uint8_t)(ctz64(r) + 2);
      for (uint64_t r = mh; r; r &= r - 1) out.tab[64 + ctz64(r)] = (uint8_t)(64 + ctz64(r) + 2);
      out.mask = (uint32_t)slots - 1, out.fill = out.used = n_idle;
      return out;
    }
  }
  SetImg<uint8_t> src = pool_open(key);
  // pool.copy() == set_merge into a fresh set (setA)
  SetImg<uint8_t> cp;
  cp.tab = g_sc.setA;
  for (int i = 0; i < 8; i++) cp.tab[i] = 0;
  cp.mask = 7, cp.fill = 0, cp.used = 0, cp.finger = 0, cp.cap = 0xFFFFFFFFu, cp.big = nullptr, cp.small = nullptr, cp.wide = false;
  if (src.used != 0) {
    if ((cp.fill + src.used) * 5 >= cp.mask * 3) set_resize(cp, (cp.used + src.used) * 2, lds_keys());
    if (cp.mask == src.mask && src.fill == src.used) {
      if (src.wide)  // tables beyond the record's 8 slots have 16 slots or more: 16 bytes at a time
        for (uint32_t w = 0; w < (src.mask + 1) / 16; w++) ((uint4*)cp.tab)[w] = ((const uint4*)src.tab)[w];
      else
        for (uint32_t i = 0; i <= src.mask; i++) cp.tab[i] = src.tab[i];
    } else if (src.wide) {
      for (uint32_t w = 0; w < (src.mask + 1) / 16; w++) {
        const uint4 q = ((const uint4*)src.tab)[w];
        const uint32_t word[4] = {q.x, q.y, q.z, q.w};
        for (int b = 0; b < 16; b++) {
          uint32_t en = (word[b >> 2] >> (8 * (b & 3))) & 0xFFu;
          if (en >= 2) set_insert_clean(cp.tab, cp.mask, en - 2);
        }
      }
    } else {
      for (uint32_t i = 0; i <= src.mask; i++) {
        uint32_t en = src.tab[i];
        if (en >= 2) set_insert_clean(cp.tab, cp.mask, en - 2);
      }
    }
    cp.fill = cp.used = src.used;
  }
  for (uint32_t i = 0; i <= cp.mask; i++) {
    uint32_t en = cp.tab[i];
    if (en >= 2 && !g_hot.ex_executing[en - 2]) set_add(out, en - 2, lds_keys());
  }
  return out;
}

// ENV:745-782 with executor_ids=None: all idle executors of `src`, in set order
SSS_DEV void move_idle_executors_all(uint32_t src) {
  PROF3(10);
  if (src == POOL_NONE) src = H.curr_source;
  CHECK(src != POOL_NONE);
  if (src == POOL_NONE || src == POOL_COMMON) return;
  int j = key_job(src), s = key_stage(src);
  const SssJob* jp0 = jobp(j);
  bool is_sat = (int)jp0->sat_count == (int)jp0->n_stages;
  if (s < 0 && !is_sat) {
    // nothing moves (ENV:766-769) - but the reference has built the idle list by then and asserts that it
    // is not empty ("[_move_idle_executors],2"): the pool's idle members are the executors located in it
    bool any_idle = false;
    for (int e = 0; e < g_c.E; e++) any_idle = any_idle || (g_hot.ex_loc[e] == src && !g_hot.ex_executing[e]);
    CHECK(any_idle);
    return;
  }
  SetImg<uint8_t> idle = get_idle_source_executors(src);
  CHECK(idle.used > 0);  // assert executor_ids, "[_move_idle_executors],2"
  if (H.err) return;
  uint32_t dst = is_sat ? POOL_COMMON : key_job_pool(j);
  for (uint32_t i = 0; i <= idle.mask; i++) {  // list(set): ascending slot order
    uint32_t en = idle.tab[i];
    if (en < 2) continue;
    int e = (int)en - 2;
    trk_move_executor_to_pool(e, dst, false);
    if (dst == POOL_COMMON) job_detach_executor(j, e);
  }
}

SSS_DEV void move_executor_to_stage(int e, int j, int s) {  // ENV:784-819
  PROF3(11);
  JobView v = jobview(j);  // (nothing below hands out cache slots before the view's last use)
  if (v.st[s].remaining == 0) {
    // _try_backup_schedule
    int bj, bs;
    find_backup_stage(e, bj, bs);
    if (bj < 0) {
      move_idle_executor(g_hot.ex_loc[e], e);
      return;
    }
    j = bj, s = bs;  // a schedulable stage has demand > 0, hence remaining > 0: no second detour
    v = jobview(j);
    CHECK(v.st[s].remaining > 0);
    if (H.err) return;
  }
  if (g_hot.ex_job[e] != j) {
    send_executor(e, j, s);
    return;
  }
  if (!(v.job->frontier_mask & bit64(s))) {
    g_hot.ex_task_stage[e] = -1;
    trk_move_executor_to_pool(e, key_job_pool(j), false);
    return;
  }
  trk_move_executor_to_pool(e, key_stage_pool(j, s), false);
  execute_next_task(e, j, s);
}

SSS_DEV void fulfill_commitment(int e, uint32_t dst) {  // ENV:699-712
  // the executor is about to work for (or travel to) the destination's job, whose records then get
  // a cache slot anyway (push_event): taking it now turns the scattered HBM accesses below into LDS ones
  if (dst != POOL_COMMON) cache_acquire(key_job(dst));
  uint32_t src = trk_remove_commitment(e, dst);
  if (H.err) return;
  if (dst == POOL_COMMON) {
    move_idle_executor(src, e);
    return;
  }
  move_executor_to_stage(e, key_job(dst), key_stage(dst));
}

// ENV:730-743, first half. The source's commitments in insertion order (dict copy, TRK:133-134) - all lanes, one
// commitment entry each: an entry's place is the number of the source's entries inserted before it (a v_readlane
// sweep over those entries; lane 0 alone would scan the whole list once per entry) ...
SSS_DEV void fulfil_order_commitments() {
  PROF3(36);
  const int lane = wave_lane();
  const uint32_t src = g_hot.h.curr_source;
#ifdef SSS_WIDE  // up to 128 entries: lane 0 sorts the source's few by insertion
  if (lane == 0) {
    int n = 0;
    for (int i = 0; i < g_hot.h.n_commits; i++) {
      if (g_hot.c_src[i] != src) continue;
      int q = n++;
      for (; q > 0 && g_sc.fc_seq[q - 1] > g_hot.c_seq[i]; q--) g_sc.fc_dst[q] = g_sc.fc_dst[q - 1], g_sc.fc_num[q] = g_sc.fc_num[q - 1], g_sc.fc_seq[q] = g_sc.fc_seq[q - 1];
      g_sc.fc_dst[q] = g_hot.c_dst[i], g_sc.fc_num[q] = g_hot.c_n[i], g_sc.fc_seq[q] = g_hot.c_seq[i];
    }
    g_sc.fc_n = n;
  }
  wave_sync();
  return;
#endif
  const bool mine = lane < g_hot.h.n_commits && g_hot.c_src[lane] == src;
  const uint32_t seq = g_hot.c_seq[lane];
  const uint32_t dst = g_hot.c_dst[lane];
  const int16_t num = g_hot.c_n[lane];
  const uint64_t mm = wave_ballot(mine);
  uint32_t place = 0;
  for (uint64_t m = mm; m; m &= m - 1) place += wave_readlane_u32(seq, ctz64_nz(m)) < seq ? 1u : 0u;
  if (mine) g_sc.fc_dst[place] = dst, g_sc.fc_num[place] = num;
  if (lane == 0) g_sc.fc_n = popc64(mm);
  wave_sync();
}
// ... and (lane 0) the idle executors that will fulfil them, in set.pop() order. What each pop yields does not depend
// on the fulfilments, so the list is complete before the first executor moves.
SSS_DEV void fulfil_build_list() {
  PROF3(12);
  uint32_t src = H.curr_source;
  SetImg<uint8_t> idle = get_idle_source_executors(src);
  const uint32_t* dsts = g_sc.fc_dst;
  const int16_t* nums = g_sc.fc_num;
  const int n = g_sc.fc_n;
  int m = 0, m_par = -1;
  for (int i = 0; i < n; i++) {
    int num = nums[i];
    if (dsts[i] == POOL_COMMON && m_par < 0) m_par = m;  // the common pool is committed to last (ENV:196): a suffix
    while (num && idle.used) {
      g_sc.fi_e[m] = (uint8_t)set_pop(idle), g_sc.fi_k[m] = (uint8_t)i, m++;
      num--;
    }
  }
  g_sc.fi_m = m, g_sc.fi_m_par = m_par < 0 ? m : m_par;
  CHECK(idle.used == 0);
}

// ENV:730-743, second half, one executor at a time (lane 0): items [from, fi_m) of the list
SSS_DEV void fulfil_serial_range(int from, int to) {
  for (int i = from; i < to && !H.err; i++) fulfill_commitment((int)g_sc.fi_e[i], g_sc.fc_dst[g_sc.fi_k[i]]);
}
SSS_DEV void fulfil_serial(int from) { fulfil_serial_range(from, g_sc.fi_m); }

enum { FI_SEND = 1, FI_EXEC = 2, FI_PARK = 3 };

// ------------------------------------------------------------------------------------------
// Lane-parallel fulfilment (all lanes): items [c0, c0 + n) of the list, one lane each, n <= 24.
// An executor committed to a stage is either SENT there (it belongs to another job or to none:
// ENV:617-637, an EXECUTOR_READY event after moving_delay) or it already works for the stage's job:
// then it moves into the stage's pool and STARTS a task if the stage is in the frontier
// (ENV:584-615: a duration draw and a TASK_FINISHED event), else it is PARKED in the job's pool
// (ENV:808-813, no event). What one fulfilment needs from the
// ones before it is little, and computable from ballots because the lanes ARE the order:
//   * the push counter of its event = counter + the number of event-pushing items before it;
//   * the stage's task counters = initial - the tasks started by the items of the same commitment
//     before it (items of one commitment are consecutive);
//   * the job's number of local executors seen by a duration draw = initial - the executors sent away
//     before it (they are detached from the source's job, JOB:86-89);
//   * its position in the random stream = the raw outputs consumed by the draws before it, known
//     without their values (one for random() when the executor-level interval is open, one 32-bit
//     half for the bounded integer: numpy's buffered 32-bit path, parity of the buffered half included).
// Removals from the source pool commute (a removal leaves a dummy, probe chains do not change) - unless
// executors are parked in the source itself (taken out and put back): then its operations run in item order;
// additions to a pool are made in item order by lane 0 on the staged image (pool_stage_in / _out).
// Returns n when the chunk was fulfilled. When it holds anything else (a stage short of tasks -> backup
// scheduling, the source pool as destination, duration lists
// with one or no entry, a draw that needs Lemire's rejection test) nothing is modified and the
// return value is the index (< n) of the first item that cannot go this way; [that item, serial_end)
// - the rest of its commitment - is for the one-at-a-time path, the items before it for a shorter chunk.
// ------------------------------------------------------------------------------------------
SSS_DEV int fulfil_chunk(int c0, int n, int& serial_end) {
  PROF3(35);
  const int lane = wave_lane();
  const bool active = lane < n;
  const int idx = c0 + (active ? lane : 0);
  // ---- reads ----
  const uint32_t src = g_hot.h.curr_source;
  const uint32_t counter0 = g_hot.h.counter, h0 = g_hot.h.rng_has32, u32_0 = g_hot.h.rng_u32;
  const int pos = g_sc.rng_pos;
  const double wall = g_hot.h.wall_time;
  const int e = g_sc.fi_e[idx], k = g_sc.fi_k[idx];
  const uint32_t dst = g_sc.fc_dst[k];
  const int j = key_job(dst), s = key_stage(dst);
  const int exj = g_hot.ex_job[e], exts = g_hot.ex_task_stage[e];
  const int src_job = key_job(src);
  SssStage* sp = stgp(j, s);
  SssJob* jp = jobp(j);
  SssStage st = *sp;
  const uint64_t local = jp->local_mask;
  const int gs = jp->gs_base + s;
  const bool in_frontier = (jp->frontier_mask & bit64(s)) != 0;
  const int slot = lds_slot_of()[j];
  const SssPoolHdr src_hdr = g_c.pool_hdr[pool_index(src)];
  const int type = exj != j ? FI_SEND : (in_frontier ? FI_EXEC : FI_PARK);
  // parked in the pool it is in (the source is its job's pool): the move takes it out and puts it back (TRK:188-222)
  const bool park_here = type == FI_PARK && src == key_job_pool(j);
  bool bad = active && (dst == src || s < 0 || g_hot.ex_executing[e] || g_hot.ex_loc[e] != src || (exj >= 0 && exj != src_job) || (type == FI_SEND && j == src_job));
  const uint64_t below = bit64(lane) - 1;
  const uint64_t m_act = wave_ballot(active);
  const uint64_t m_exec = wave_ballot(active && type == FI_EXEC), m_park = wave_ballot(active && type == FI_PARK);
  const uint64_t m_send = m_act & ~m_exec & ~m_park, m_event = m_act & ~m_park;
  const uint64_t m_send_att = wave_ballot(active && type == FI_SEND && exj >= 0);
  // the items of this lane's commitment (consecutive lanes)
  uint64_t run = 0;
  for (uint64_t rem = m_act; rem;) {
    const int l = ctz64(rem);
    const uint32_t kk = wave_readlane_u32((uint32_t)k, l);
    const uint64_t mk = wave_ballot(active && (uint32_t)k == kk);
    if ((uint32_t)k == kk) run = mk;
    rem &= ~mk;
  }
  const int n_run = popc64(run), n_exec_run = popc64(run & m_exec), n_send_run = popc64(run & m_send), n_park_run = n_run - n_exec_run - n_send_run;
  // a stage without remaining tasks sends the executor to a backup stage (ENV:784-797): one at a time
  bad = bad || (active && ((int)st.remaining < n_exec_run + ((n_send_run || n_park_run) ? 1 : 0) || (int)st.commit_to < n_run));
  // the duration draw of a task start (TPCH:75-106, 216-235)
  int n_local = 0, li = 0, ri = 0;
  int4 da = mk_i4(0, 0, 0, 0), db = da;
  bool open = false;
  if (active && type == FI_EXEC && !bad) {
    n_local = local_count(local) - popc64(m_send_att & below);
    if (n_local <= 0 || n_local > g_c.E)
      bad = true;
    else {
      executor_interval(n_local, li, ri);
      open = li != ri;
      const int mode = exts < 0 ? 0 : (exts == s ? 1 : 2);
      const int32_t* eff = g_c.pk.eff;
      da = *(const int4*)eff_row(eff, gs, li, mode);
      db = open ? *(const int4*)eff_row(eff, gs, ri, mode) : da;
      bad = (da.y & LENW_LEN) <= 1 || (db.y & LENW_LEN) <= 1;
    }
  }
  const uint64_t m_open = wave_ballot(active && type == FI_EXEC && open);
#ifdef SSS_BATCH_STATS
  {
    uint64_t b1 = wave_ballot(active && type == FI_PARK && src == key_job_pool(j)), b2 = wave_ballot(active && dst == src);
    uint64_t b3 = wave_ballot(active && ((int)st.remaining < n_exec_run + (n_send_run ? 1 : 0))), b4 = wave_ballot(active && (int)st.commit_to < n_run);
    uint64_t b5 = wave_ballot(active && type == FI_EXEC && ((da.y & LENW_LEN) <= 1 || (db.y & LENW_LEN) <= 1));
    uint64_t b6 = wave_ballot(active && (g_hot.ex_executing[e] || g_hot.ex_loc[e] != src || (exj >= 0 && exj != src_job) || (type == FI_SEND && j == src_job)));
    STAT(57, b1 != 0), STAT(58, b2 != 0), STAT(59, b3 != 0), STAT(60, b4 != 0), STAT(61, b5 != 0), STAT(62, b6 != 0);
  }
#endif
  {
    const uint64_t m_bad = wave_ballot(bad);
    if (m_bad != 0 || 2 * popc64(m_exec) > 64 - pos) {
      const int fb = m_bad ? ctz64(m_bad) : 0;
      const uint32_t rlo = wave_readlane_u32((uint32_t)run, fb), rhi = wave_readlane_u32((uint32_t)(run >> 32), fb);
      const uint64_t r = ((uint64_t)rhi << 32) | rlo;
      serial_end = c0 + (r ? 64 - __builtin_clzll(r) : fb + 1);
      return fb;
    }
  }
  const uint32_t rank = (uint32_t)popc64(m_exec & below), R = (uint32_t)popc64(m_open & below);
  const uint32_t Fr = h0 ? rank >> 1 : (rank + 1) >> 1;
  const bool fresh = ((h0 + rank) & 1u) == 0;
  const uint32_t P = R + Fr;
  int4 dd = da;
  uint64_t x32 = 0;
  uint32_t u32 = 0;
  const bool is_exec = active && type == FI_EXEC;
  if (is_exec) {
    if (open) {
      const double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
      const int rand_pt = 1 + (int)(u64_to_unit(g_sc.rng_buf[pos + (int)P]) * (right - left));
      if (!((double)rand_pt <= (double)n_local - left)) dd = db;
    }
    if (fresh) {
      x32 = g_sc.rng_buf[pos + (int)P + (open ? 1 : 0)];
      u32 = (uint32_t)x32;
    } else if (rank == 0) {
      u32 = u32_0;
    } else {
      u32 = (uint32_t)(g_sc.rng_buf[pos + (int)R + (int)Fr - 1] >> 32);
    }
  }
  const uint32_t len = (uint32_t)(dd.y & LENW_LEN);
  const uint64_t mm = (uint64_t)u32 * len;
  {
    const uint64_t m_rej = wave_ballot(is_exec && (uint32_t)mm < len);
    if (m_rej != 0) {
      serial_end = c0 + ctz64(m_rej) + 1;
      return ctz64(m_rej);
    }
  }
  // ---- commit ----
  const bool big_src = src_hdr.mask != 7;
  // removals from the source commute - unless executors are put back in between: then the pool's operations
  // run in item order on the staged image
  const bool staged_src = wave_ballot(active && park_here) != 0;
  if (active) {
    double t = wall + g_c.P.moving_delay;
    double dur = 0.0;
    if (is_exec) {
      dur = (double)g_c.pk.durations[dd.x + (int)(mm >> 32)];
      if (dd.y >> 30) dur += g_c.P.warmup_delay;
      t = wall + dur;
    }
    if (type != FI_PARK) {
      SssEvSlot sl;
      sl.t = t, sl.seq = counter0 + (uint32_t)popc64(m_event & below), sl.info = ev_info(is_exec ? EV_TASK_FINISHED : EV_EXECUTOR_READY, j, s, (uint32_t)slot);
      g_hot.ev[e] = sl;
    }
    if (type == FI_PARK) {
      g_hot.ex_loc[e] = key_job_pool(j), g_hot.ex_task_stage[e] = -1;  // ENV:808-813
    } else if (is_exec) {
      g_hot.ex_loc[e] = dst, g_hot.ex_task_stage[e] = (int8_t)s, g_hot.ex_executing[e] = 1;
      if ((run & m_exec & ~(below | bit64(lane))) == 0) *durp(j, s) = (float)dur;  // the commitment's last task start: most recent duration (ENV:604)
      if ((m_exec & ~(below | bit64(lane))) == 0) {  // the chunk's last draw leaves the generator behind
        g_sc.fi_rng_pos = (uint32_t)pos + P + (open ? 1u : 0u) + (fresh ? 1u : 0u);
        g_sc.fi_rng_has32 = fresh ? 1u : 0u, g_sc.fi_rng_u32 = fresh ? (uint32_t)(x32 >> 32) : u32;
      }
    } else {
      g_hot.ex_loc[e] = POOL_NONE;
      if (exj >= 0) {
        g_hot.ex_job[e] = -1, g_hot.ex_task_stage[e] = -1;  // JOB:86-89
#ifndef SSS_WIDE  // (the wide instantiation keeps a count: popc64(m_send_att) below)
        lane_atomic_or_u64(&g_sc.fi_detach, bit64(e));
#endif
      }
    }
    g_sc.fi_type[idx] = (uint8_t)type;
    if ((run & ~(below | bit64(lane))) == 0) {  // last item of its commitment: the stage's counters (TRK:159-176,188-222; STG:53-58)
      st.remaining = st.remaining - n_exec_run, st.executing = (int16_t)(st.executing + n_exec_run);
      st.commit_to = (uint8_t)(st.commit_to - n_run), st.moving_to = (uint8_t)(st.moving_to + n_send_run);
      *sp = st;
      if (n_exec_run && st.remaining == 0) lane_atomic_add_u32((uint32_t*)&jp->supply, 1u << 16);  // sat_count++ (ENV:595-597)
      // executor demand = remaining - (moving_to + commit_to) is what it was unless executors were parked
      // (their commitments are gone, they did not reach the stage): then the saturation bit is re-derived
      if (n_park_run) {
        if ((int)st.remaining - ((int)st.moving_to + (int)st.commit_to) <= 0)
          lane_atomic_or_u64(&jp->sat_mask, bit64(s));
        else
          lane_atomic_and_u64(&jp->sat_mask, ~bit64(s));
      }
    }
    if (big_src && !staged_src) {
      bool was = table_mark_dummy(pool_table_hbm(src), src_hdr.mask, (uint32_t)e);
      CHECK(was);
    }
  }
  wave_sync();
  const uint32_t src_jpool = src_job >= 0 ? key_job_pool(src_job) : POOL_NONE;
  if (staged_src) {
    tabword_t src_words;
    SetImg<uint8_t> sset = pool_stage_in(src, src_words);
    for (int i = c0; i < c0 + n; i++) {  // (every lane: the operations run on the whole wave, staged_add / staged_remove)
      bool was = staged_remove(sset, (uint32_t)g_sc.fi_e[i]);
      CHECK(was);
      if (g_sc.fi_type[i] == FI_PARK && src == src_jpool) staged_add(sset, (uint32_t)g_sc.fi_e[i]);
    }
    sset.aux -= (uint32_t)n;
    wave_sync();
    pool_stage_out(src, sset, src_words);
  }
  if (lane == 0) {
    // the source pool's record: n executors and n commitments fewer
    if (!staged_src) {
      SetImg<uint8_t> sset = pool_open(src);
      if (!big_src) {
        for (int i = c0; i < c0 + n; i++) {
          bool was = set_remove(sset, (uint32_t)g_sc.fi_e[i]);
          CHECK(was);
        }
      } else
        sset.used -= (uint32_t)n;
      sset.aux -= (uint32_t)n;
      pool_close(src, sset);
    }
    // executors sent away leave the source's job (JOB:86-89, TRK:218-221)
    if (src_job >= 0) {
      const int n_send_all = popc64(m_send);
      if (n_send_all) add_supply(src_job, -n_send_all);
#ifdef SSS_WIDE
      if (m_send_att) (*jobp(src_job)).local_mask -= (uint64_t)popc64(m_send_att);
#else
      if (g_sc.fi_detach) (*jobp(src_job)).local_mask &= ~g_sc.fi_detach;
#endif
    }
    g_sc.fi_detach = 0;
  }
  // stage pools (and, for parked executors, job pools) receive their executors in item order, through the staging
  // area; the items of one commitment are consecutive and all of one kind
  for (int i = c0; i < c0 + n;) {
    const int kk = g_sc.fi_k[i];
    const uint32_t d = g_sc.fc_dst[kk];
    int i1 = i, n_ex = 0, n_pk = 0;
    while (i1 < c0 + n && g_sc.fi_k[i1] == kk) n_ex += g_sc.fi_type[i1] == FI_EXEC, n_pk += g_sc.fi_type[i1] == FI_PARK, i1++;
    const uint32_t into = n_ex ? d : key_job_pool(key_job(d));
    if ((n_ex || n_pk) && into != src) {
      tabword_t into_words;
      SetImg<uint8_t> dset = pool_stage_in(into, into_words);
      for (int q = i; q < i1; q++) staged_add(dset, (uint32_t)g_sc.fi_e[q]);
      wave_sync();
      pool_stage_out(into, dset, into_words);
    }
    i = i1;
  }
  if (lane == 0) {
    // commitments are settled; events name their jobs' slots
    for (int i = c0; i < c0 + n;) {
      const int kk = g_sc.fi_k[i];
      const uint32_t d = g_sc.fc_dst[kk];
      int i1 = i, n_pk = 0;
      while (i1 < c0 + n && g_sc.fi_k[i1] == kk) n_pk += g_sc.fi_type[i1] == FI_PARK, i1++;
      int ci;
      for (ci = 0; ci < H.n_commits; ci++)
        if (g_hot.c_src[ci] == src && g_hot.c_dst[ci] == d) break;
      CHECK(ci < H.n_commits);
      if (ci < H.n_commits) {
        g_hot.c_n[ci] = (int16_t)(g_hot.c_n[ci] - (i1 - i));
        if (g_hot.c_n[ci] == 0) {
          int last = H.n_commits - 1;
          g_hot.c_src[ci] = g_hot.c_src[last], g_hot.c_dst[ci] = g_hot.c_dst[last], g_hot.c_n[ci] = g_hot.c_n[last], g_hot.c_seq[ci] = g_hot.c_seq[last];
          H.n_commits = last;
        }
      }
      const int sl = lds_slot_of()[key_job(d)];
      if (sl != SLOT_NONE && !n_pk) lds_slot_ref()[sl] = (uint8_t)(lds_slot_ref()[sl] + (i1 - i));
      i = i1;
    }
    H.counter = counter0 + (uint32_t)popc64(m_event);
    if (m_exec) g_sc.rng_pos = (int32_t)g_sc.fi_rng_pos, H.rng_has32 = g_sc.fi_rng_has32, H.rng_u32 = g_sc.fi_rng_u32;
  }
  wave_sync();
  return n;
}

// The tail of a fulfilment (all lanes): items [from, to) of the list, all of them commitments to the COMMON pool - what
// is left of the source's idle executors when a scheduling round ends (ENV:196, 487-503; the common pool is committed to
// last). Each one settles its commitment (TRK:159-176) and - ENV:702-705 -> 745-782 with a list of one - stays where it is
// (the source is the common pool, or the pool of a job that still has unsaturated stages), or moves from the source to
// its job's pool (the source is a stage's pool), or - the job being saturated - is detached into the common pool. Every
// item has the same source and the same destination, so the whole run is: the commitment entry shrinks by n, the two
// pool images come in with one round trip (pool_pair_*), n removals and n additions in item order with the whole wave,
// and lane 0 rewrites the executors' records. One at a time on lane 0 this was ~8 k ticks per executor - the dependent HBM
// round trips of trk_move_executor_to_pool - and up to 50 executors long: ~100 k ticks of the slowest envs' steps at
// BASELINE config 3. Returns false, with nothing modified, when the run has to go one at a time (64 executors: two
// 512-byte tables do not fit the staging areas).
SSS_DEV bool fulfil_common_wave(int from, int to) {
  UTRACE("fulfil_common_wave");
  PROF3(39);
  const int lane = wave_lane();
  const int n = to - from;
  // ---- reads ----
  const uint32_t src = g_hot.h.curr_source;
  const int n_commits = g_hot.h.n_commits;
  const int j = key_job(src), s = key_stage(src);
  bool moves = false, is_sat = false;
  SssJob* jp = nullptr;
  if (src != POOL_NONE && src != POOL_COMMON) {
    jp = jobp(j);
    is_sat = (int)jp->sat_count == (int)jp->n_stages;  // JOB:53-55
    moves = s >= 0 || is_sat;                          // ENV:766-769: a job's pool keeps its executors while the job has work
  }
  const uint32_t dstp = is_sat ? POOL_COMMON : key_job_pool(j);
  const CommitHit hit = commit_first_wave(src, true, n > 0 && src != POOL_NONE && pair_staging_fits(g_c.E), n_commits);  // (a source has one entry per destination)
  if (hit.ci < 0) return false;
  const int ci = hit.ci;
  const int c_left = hit.num - n;
  if (wave_ballot(c_left < 0) != 0) return false;
  // ---- from here on the items are consumed ----
  LocalGroup moved = local_group();
  if (moves) {
    const PoolPairRegs pr = pool_pair_fetch(src, dstp, true);
    PairImg so, sn;
    pool_pair_stage(pr, true, so, sn);
    pair_remove_many(so, g_sc.fi_e, from, to);  // TRK:188-222, the removals (they commute: every member's own lane)
    for (int i = from; i < to; i++) {  // ... the additions, in item order (wave-uniform: every lane reads the list)
      const uint32_t e = g_sc.fi_e[i];
      local_group_add(moved, (int)e);
      pair_add(sn, e);
    }
    so.s.aux -= (uint32_t)n;  // the source's outgoing commitments (TRK:159-176)
    wave_sync();
    pool_pair_flush_one(src, so);
    pool_pair_flush_one(dstp, sn);
  }
  if (lane == 0) {
    SssHdr& h = g_hot.h;
    if (!moves) {
      SssPoolHdr* hd = g_c.pool_hdr + pool_index(src);
      hd->commit_from = (int16_t)(hd->commit_from - n);
      CHECK(hd->commit_from >= 0);
    }
    if (j >= 0) {  // commitments of a job's executors to the common pool counted as its supply (TRK:146-154, 159-176)
      h.supply_none -= n;
      CHECK(h.supply_none >= 0);
    }
    g_hot.c_n[ci] = (int16_t)c_left;
    if (c_left == 0) {  // dict.pop: swap-remove, the order lives in c_seq
      const int last = h.n_commits - 1;
      g_hot.c_src[ci] = g_hot.c_src[last], g_hot.c_dst[ci] = g_hot.c_dst[last], g_hot.c_n[ci] = g_hot.c_n[last], g_hot.c_seq[ci] = g_hot.c_seq[last];
      h.n_commits = last;
    }
    if (moves) {
      for (int i = from; i < to; i++) {
        const int e = g_sc.fi_e[i];
        g_hot.ex_loc[e] = dstp;
        if (dstp == POOL_COMMON) g_hot.ex_job[e] = -1, g_hot.ex_task_stage[e] = -1;  // JOB:86-89
      }
      if (dstp == POOL_COMMON) local_group_detach(jp, moved);
    }
  }
  wave_sync();
  return true;
}

// ENV:730-743, second half (all lanes): lane-parallel chunks while the list allows, the rest one at a time
SSS_DEV void fulfil_run() {
  PROF3(33);
  const int m = g_sc.fi_m, m_par = g_sc.fi_m_par;
  int done = 0;
#ifndef SSS_NO_BATCH
  while (done < m_par) {
    if (64 - g_sc.rng_pos < 48) rng_refill();
    int n = m_par - done < 24 ? m_par - done : 24;
    // the jobs the chunk's events will name get their cache slots first (as push_event would see to)
    if (wave_lane() == 0)
      for (int i = done; i < done + n; i++) cache_acquire(key_job(g_sc.fc_dst[g_sc.fi_k[i]]));
    wave_sync();
    int serial_end = 0;
    int got = fulfil_chunk(done, n, serial_end);
    if (got < n) {
      STAT(54, 1);
      if (got > 0) {  // the items before the first one that needs the general path
        int dummy = 0;
        int again = fulfil_chunk(done, got, dummy);
        if (again < got) break;  // (cannot happen: the prefix passed every test a moment ago)
        STAT(53, 1), STAT(55, got);
        done += got;
      }
      if (wave_lane() == 0) fulfil_serial_range(done, serial_end);
      wave_sync();
      if (wave_ballot(g_hot.h.err != 0) != 0) break;
      done = serial_end;
      continue;
    }
    STAT(53, 1), STAT(55, n);
    done += n;
  }
  // the commitments to the common pool (a suffix of the list): one source, one destination - with the whole wave
  if (done == m_par && m_par < m && wave_ballot(g_hot.h.err != 0) == 0 && fulfil_common_wave(m_par, m)) {
    STAT(123, 1), STAT(124, m - m_par);
    done = m;
  }
#endif
  STAT(56, m - done);
  (void)m_par;
  if (wave_lane() == 0 && done < m) fulfil_serial(done);
  wave_sync();
}

SSS_DEV void commit_remaining_executors() {  // ENV:487-503
  int n = trk_num_committable();
  if (n > 0) trk_add_commitment(n, POOL_COMMON);
}

// ------------------------------------------------------------------------------------------
// event handlers (lane 0)
// ------------------------------------------------------------------------------------------

// ---- LDS cache of job records (lane 0 flavour) ----
// A slot holds one job's record, stage counters and recent durations. Slots go to the jobs the
// event chain works on: a job gets one when an event is pushed for it (push_event) and keeps it at
// least while events that name it are pending (lds_slot_ref) - so with n_slots >= num_executors every
// pending event finds its job in LDS. Everything else reaches a job through jobp / stgp / durp, which
// fall back to the HBM copy. Slots are written back when their job completes, when they are handed
// to another job, and at the end of the launch.
SSS_DEV void cache_release(int j) {  // LDS -> HBM, slot becomes free
  int k = lds_slot_of()[j];
  if (k == SLOT_NONE) return;
  g_c.jobs[j] = lds_cjobs()[k];
  for (int s = 0; s < g_c.SP; s++) {
    g_c.stages[j * g_c.SP + s] = lds_cstages()[k * g_c.SP + s];
    g_c.durations[j * g_c.SP + s] = lds_cdur()[k * g_c.SP + s];
  }
  lds_slot_of()[j] = SLOT_NONE;
  g_sc.free_slots |= bit64(k);
}
SSS_DEV int cache_acquire(int j) {  // HBM -> LDS if the job has no slot yet; returns its slot or SLOT_NONE
  PROF3(19);
  int k = lds_slot_of()[j];
  if (k != SLOT_NONE) return k;
  if (g_sc.free_slots == 0) {
    // hand over the slot of a job no pending event names (never the job whose event is being handled)
    int victim = -1;
    for (int q = 0; q < g_c.P.n_slots; q++)
      if (lds_slot_ref()[q] == 0 && (int)lds_slot_job()[q] != g_sc.pinned_job) {
        victim = q;
        break;
      }
    if (victim < 0) return SLOT_NONE;
    cache_release((int)lds_slot_job()[victim]);
  }
  k = ctz64(g_sc.free_slots);
  g_sc.free_slots &= g_sc.free_slots - 1;
  lds_cjobs()[k] = g_c.jobs[j];
  {
    // stage counters (SP x 8 bytes) and recent durations (SP x 4 bytes; SP is even) as 64-bit words, eight HBM loads in
    // flight before the first LDS store: written as one load-store loop every word was a round trip of its own (the
    // compiler keeps the loop's loads behind its stores) - ~25 k ticks per miss at 18 stages
    const uint64_t* gs = (const uint64_t*)(g_c.stages + j * g_c.SP);
    const uint64_t* gd = (const uint64_t*)(g_c.durations + j * g_c.SP);
    uint64_t* ls = (uint64_t*)(lds_cstages() + k * g_c.SP);
    uint64_t* ld = (uint64_t*)(lds_cdur() + k * g_c.SP);
    const int nw = g_c.SP + g_c.SP / 2;
    for (int w0 = 0; w0 < nw; w0 += 8) {
      uint64_t v[8];
      SSS_UNROLL8 for (int u = 0; u < 8; u++) {
        const int w = w0 + u;
        v[u] = w < g_c.SP ? gs[w < g_c.SP ? w : 0] : (w < nw ? gd[w - g_c.SP] : 0ull);
      }
      SSS_UNROLL8 for (int u = 0; u < 8; u++) {
        const int w = w0 + u;
        if (w < g_c.SP) ls[w] = v[u];
        else if (w < nw) ld[w - g_c.SP] = v[u];
      }
    }
  }
  lds_slot_of()[j] = (uint8_t)k;
  lds_slot_job()[k] = (uint16_t)j;
  lds_slot_ref()[k] = 0;
  return k;
}

SSS_DEV void handle_job_arrival(int j) {  // ENV:428-438 (pools were created empty at reset)
  lds_active()[H.n_active] = (uint16_t)j;
  H.n_active++;
  g_sc.active_version++, g_sc.active_dirty = 1;
  H.graph_version++;
  if (g_c.pool_hdr[0].used > 0) H.curr_source = POOL_COMMON;
}

SSS_DEV void handle_executor_arrival(int e, int j, int s) {  // ENV:440-450
  PROF3(14);
  const JobView v = jobview(j);
  CHECK(g_hot.ex_task_stage[e] < 0);  // JOB:81-84
  v.job->local_mask = local_with(v.job->local_mask, e);
  g_hot.ex_job[e] = (int16_t)j;
  const int mv = (int)v.st[s].moving_to - 1;  // TRK:185-187
  CHECK(mv >= 0);
  v.st[s].moving_to = (uint8_t)mv;
  update_sat(v, s);
  trk_move_executor_to_pool(e, key_job_pool(j), false);
  move_executor_to_stage(e, j, s);
}

SSS_DEV void process_job_completion(int j) {  // ENV:682-697
  PROF3(15);
  if (pool_size(key_job_pool(j)) > 0) move_idle_executors_all(key_job_pool(j));
  CHECK(pool_size(key_job_pool(j)) == 0);
  int k;
  for (k = 0; k < H.n_active; k++)
    if (lds_active()[k] == j) break;
  CHECK(k < H.n_active);
  if (k >= H.n_active) return;
  for (int i = k; i + 1 < H.n_active; i++) lds_active()[i] = lds_active()[i + 1];
  H.n_active--;
  (*jobp(j)).completion_order = (int16_t)H.n_completed;
  H.n_completed++;
  g_sc.pending_free = j;  // its cache slot is written back once the handler has returned
  g_sc.active_version++, g_sc.active_dirty = 1;
  H.graph_version++;
  g_c.t_completed[j] = H.wall_time;
  double dur = H.wall_time - g_c.t_arrival[j];
  if (H.dur_n < SSS_DUR_RING) {
    g_c.dur_ring[(H.dur_head + H.dur_n) % SSS_DUR_RING] = dur;
    H.dur_n++;
  } else {
    g_c.dur_ring[H.dur_head] = dur;
    H.dur_head = (H.dur_head + 1) % SSS_DUR_RING;
  }
}

SSS_DEV void handle_task_completion(int e, int j, int s) {  // ENV:452-483
  PROF3(16);
  SssStage* stp = stgp(j, s);
  SssStage st = *stp;  // (one 8-byte access; the copy is what the tests below look at)
  CHECK(!stage_completed(st));
  st.executing = (int16_t)(st.executing - 1);  // STG:60-62
  *stp = st;
  g_hot.ex_executing[e] = 0;
  if (st.remaining > 0) {
    execute_next_task(e, j, s);
    return;
  }
#ifdef SSS_BATCH_STATS
  {
    // census: what kind of "no task left in the stage" event is this?
    uint32_t spk = key_stage_pool(j, s);
    uint32_t d0 = trk_peek_commitment(spk);
    bool completes = stage_completed(st);
    int cat = 0;  // 0 no commitment, 1 to common, 2 other job (send), 3 same job not in frontier (park), 4 same job start task
    if (d0 != POOL_NONE) {
      if (d0 == POOL_COMMON) cat = 1;
      else if (key_job(d0) != j) cat = 2;
      else cat = ((*jobp(j)).frontier_mask & bit64(key_stage(d0))) ? 4 : 3;
    }
    sss_batch_stats[24 + cat] += 1;
    if (completes) sss_batch_stats[29] += 1;
    if (cat == 4 && !completes) sss_batch_stats[30] += 1;
  }
#endif
  bool frontier_changed = false;
  if (stage_completed(st)) {
    frontier_changed = job_record_stage_completion(j, s);        // ENV:676-680
    if ((*jobp(j)).active_mask == 0) process_job_completion(j);  // JOB:49-51 (only a stage's completion can empty the job)
  }
  // _handle_released_executor ENV:639-660
  uint32_t sp = key_stage_pool(j, s);
  uint32_t dst = trk_peek_commitment(sp);
  bool had_commitment = dst != POOL_NONE;
  if (had_commitment)
    fulfill_commitment(e, dst);
  else {
    g_hot.ex_task_stage[e] = -1;
    if (frontier_changed) move_idle_executor(sp, e);
  }
  // _update_executor_source ENV:662-674
  if (frontier_changed)
    H.curr_source = key_job_pool(j);
  else if (!had_commitment)
    H.curr_source = sp;
}

// ------------------------------------------------------------------------------------------
// wave-parallel phases
// ------------------------------------------------------------------------------------------

#define POP_EMPTY (-1)
#define POP_ARRIVAL (-2)
// EventQueue.pop (EVQ:44-49) with the whole wave. The "heap" is one slot per executor (an executor
// has at most one pending event; t = +inf when it has none) plus the time-sorted arrival array
// with a cursor. (t, push counter) keys are unique, so the minimum is the heapq order; arrivals
// carry the counters 0..J-1 and therefore win ties against executor events. One lane per
// executor, lexicographic min over (time, push counter) on the DPP network - no LDS round trips
// beyond the one read of the slots. All lanes call it; every lane gets the same result.
SSS_DEV int pop_event_wave(double next_arrival_t, double& t_win, uint32_t& info_win) {
  PROF3(37);
  int lane = wave_lane();
#ifdef SSS_WIDE
  {
    // two executors per lane: the lane's earlier event (by (time, push counter)) enters the wave-wide minimum
    const SssEvSlot a = g_hot.ev[lane], b = g_hot.ev[lane + 64];
    const bool b_first = b.t < a.t || (b.t == a.t && b.seq < a.seq);
    const SssEvSlot sl = b_first ? b : a;
    const int mine = b_first ? lane + 64 : lane;
    const double tmin = wave_min_f64_nonneg(sl.t);
    const bool at_min = sl.t == tmin;
    const uint32_t msq = wave_min_u32(at_min ? sl.seq : 0xFFFFFFFFu);  // equal times: the earlier push wins (EVQ:35)
    const int wl = ctz64(wave_ballot(at_min && sl.seq == msq));
    if (next_arrival_t <= tmin && next_arrival_t < __builtin_inf()) return POP_ARRIVAL;
    if (!(tmin < __builtin_inf())) return POP_EMPTY;
    t_win = tmin;
    info_win = wave_readlane_u32(sl.info, wl);
    return (int)wave_readlane_u32((uint32_t)mine, wl);
  }
#endif
  SssEvSlot sl = g_hot.ev[lane];
  // times are >= +0.0; +inf for empty slots and for the lanes beyond the executors
  double tmin = g_c.E <= 16 ? wave_min_f64_nonneg_row0(sl.t) : wave_min_f64_nonneg(sl.t);
  bool at_min = sl.t == tmin;
  uint64_t cand = wave_ballot(at_min);
  int ex = ctz64(cand);
  if (cand & (cand - 1)) {  // equal times: the earlier push wins (EVQ:35)
    uint32_t msq = wave_min_u32(at_min ? sl.seq : 0xFFFFFFFFu);
    ex = ctz64(wave_ballot(at_min && sl.seq == msq));
  }
  if (next_arrival_t <= tmin && next_arrival_t < __builtin_inf()) return POP_ARRIVAL;
  if (!(tmin < __builtin_inf())) return POP_EMPTY;
  t_win = tmin;                                // the winner's time is the minimum itself
  info_win = wave_readlane_u32(sl.info, ex);   // its event word straight from the winner's register
  return ex;
}

// launch constants the event loop needs, fetched from the LDS context once per loop
struct FastCtx {
  uint8_t* slot_of;
  SssStage* cstages;
  SssJob* cjobs;
  float* cdur;
  SssExDesc* exdesc;
  const int32_t* eff;
  const int32_t* durations;
  int SP, E;
};
SSS_DEV void fastctx_load(FastCtx& f) {
  f.slot_of = lds_slot_of(), f.cstages = lds_cstages(), f.cjobs = lds_cjobs(), f.cdur = lds_cdur(), f.exdesc = lds_exdesc();
  f.eff = g_c.pk.eff, f.durations = g_c.pk.durations, f.SP = g_c.SP, f.E = g_c.E;
}


// the duration lists an executor that stays on pack stage `gs` can draw from next ("same stage"
// mode of TPCH:75-106): one per candidate executor level (li == ri when the interval is closed)
SSS_DEV void exdesc_fetch(const FastCtx& f, SssExDesc& xd, int gs, int li, int ri) {
  const int4 a = *(const int4*)eff_row(f.eff, gs, li, 1);
  int4 b = a;
  if (ri != li) b = *(const int4*)eff_row(f.eff, gs, ri, 1);
  xd.gs = gs, xd.li = (int8_t)li, xd.ri = (int8_t)ri, xd.thr_n = -1;
  xd.off_l = a.x, xd.lenw_l = a.y, xd.thr_lo = 0;
  xd.off_r = b.x, xd.lenw_r = b.y, xd.thr_hi = 0;
}

// The common event (97-99 % of all events are TASK_FINISHED, most of them with tasks left in the
// stage): ENV:452-467 + ENV:584-615 + TPCH:75-106 fused for "same executor continues on the same
// stage". executing-- / executing++ cancel, executor.task.stage_id already equals the stage
// (=> the `rest_wave` mode of task_duration), the event slot keeps its kind/job/stage.
// One event, lane 0 (runs of such events: fast_run below).
// Returns 1 = handled, 0 = the stage has no remaining task (nothing modified: slow path), -1 = failed.
template <bool CACHED>
SSS_DEV int fast_body(const FastCtx& f, int ex, double t_ev, int j, int s, int slot) {
  SssStage* sp;
  SssJob* jp;
  float* dp;
  if (CACHED) {
    sp = f.cstages + slot * f.SP + s, jp = f.cjobs + slot, dp = f.cdur + slot * f.SP + s;
  } else {
    sp = g_c.stages + j * f.SP + s, jp = g_c.jobs + j, dp = g_c.durations + j * f.SP + s;
  }
  SssStage st = *sp;
  uint64_t local = jp->local_mask;
  int gs = jp->gs_base + s;
  SssExDesc xd = f.exdesc[ex];
  if (st.remaining <= 0) return 0;
  g_hot.h.wall_time = t_ev;
  st.remaining = st.remaining - 1;
  int demand = (int)st.remaining - ((int)st.moving_to + (int)st.commit_to);
  if (st.remaining == 0) jp->sat_count = (int16_t)(jp->sat_count + 1);  // stage just became saturated (ENV:595-597)
  if (demand <= 0) lane_atomic_or_u64(&jp->sat_mask, bit64(s));  // fire-and-forget: nothing below waits for the old mask
  *sp = st;
  // task_duration, executor mode 1 ("same stage")
  int n_local = local_count(local);
  int li, ri;
  executor_interval(n_local, li, ri);
  if (!(xd.gs == gs && xd.li == li && xd.ri == ri)) {
    exdesc_fetch(f, xd, gs, li, ri);
    f.exdesc[ex] = xd;
  }
  int lvl = li;
  if (li != ri) {
    double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
    int rand_pt = 1 + (int)(rng_random() * (right - left));
    if (!((double)rand_pt <= (double)n_local - left)) lvl = ri;
  }
  int off = lvl == li ? xd.off_l : xd.off_r, lenw = lvl == li ? xd.lenw_l : xd.lenw_r;
  int len = lenw & LENW_LEN;
#ifdef SSS_CHECK_TRACE
  if (len == 0 || n_local <= 0)
    fprintf(stderr, "[fast_body] CACHED=%d ex=%d j=%d s=%d slot=%d len=%d n_local=%d gs=%d li=%d ri=%d slot_of=%d ex_job=%d\n", (int)CACHED, ex, j, s, slot, len,
            n_local, gs, li, ri, (int)f.slot_of[j], (int)g_hot.ex_job[ex]);
#endif
  if (len == 0 || n_local <= 0) return -1;
  uint32_t i = rng_integers((uint32_t)len);
  double dur = (double)f.durations[off + (int)i];
  *dp = (float)dur;
  g_hot.ev[ex].t = t_ev + dur;
  g_hot.ev[ex].seq = g_hot.h.counter++;
  return 1;
}

SSS_DEV int fast_task_completion(const FastCtx& f, int ex, double t_ev, int j, int s, uint32_t slot) {
  // an event pushed while its job had no slot does not name one; the job may have got one since
  if (slot == INFO_SLOT_NONE && f.slot_of[j] != SLOT_NONE) slot = f.slot_of[j];
  return slot != INFO_SLOT_NONE ? fast_body<true>(f, ex, t_ev, j, s, (int)slot) : fast_body<false>(f, ex, t_ev, j, s, SLOT_NONE);
}

// ------------------------------------------------------------------------------------------
// The fast run (all lanes): consecutive "task finished, its stage has more tasks" events (ENV:452-467 +
// 584-615 + TPCH:75-106), one per iteration, with everything an iteration needs in registers. Such an
// event touches its own executor's slot, its stage's task counter and the shared random stream - and
// changes nothing another such event's handling depends on beyond those: the executor stays on its
// stage, the job keeps its executors, hence the two candidate duration lists stay what they are. So, one
// lane per executor, everything is classified ONCE when the run starts:
//   * t_stop = the earliest pending event of any other kind (and the next job arrival: arrivals win ties,
//     EVQ:35). Only such events earlier than t_stop can be part of this run - the WINDOW;
//   * the events in the window are ranked by (time, push counter), heapq's order (EVQ:35), once.
// After that an iteration is
//   * the head of the queue = the lane with rank 1 (one compare, no reduction);
//   * its draw: EVERY lane has computed, ahead of time and under the generator state the next event will
//     see, the duration its own event would draw (the executor-level choice of TPCH:222-229 is a threshold
//     on the raw output, SssPackDev::lvl_thr; numpy's buffered 32-bit Lemire draw with its spare half; the
//     64 raw outputs the wave produced ahead sit one per lane and are fetched with v_readlane) - so the
//     load from the duration pool has been in flight for a whole iteration when its value is needed;
//   * the commit: the head's lane takes its new time and push counter; the new event is the youngest, so its
//     rank is the number of window events not later than it, and those move up by one; it leaves the
//     window if it lands at or after t_stop. The lanes of the same stage follow its task counter, the
//     generator's position moves on - registers and scalars only.
// LDS sees the result when the run ends (slots, stage counters, most recent durations, saturation, header).
// The run ends when the window is empty or its head needs anything else (a stage out of tasks, a draw that
// needs Lemire's rejection loop): the event then at the head of the queue goes the general way.
// Returns the number of events handled (0: none, nothing modified).
// ------------------------------------------------------------------------------------------
#define FR_OUT 0x40000000u  // rank of a lane whose event is not in the window (never counts down to the head's rank, 1)
SSS_DEV int fast_run(const FastCtx& f) {
  UTRACE("fast_run");
#ifdef SSS_NO_BATCH  // debugging aid: every event goes through the one-at-a-time path
  return 0;
#endif
  PROF3(30);
  PROF3_SEC_BEGIN;
  const int lane = wave_lane();
  // ---- everything that is read from shared state is read before the first collective ----
  // this lane's event (t = +inf beyond the executors and for executors without one) and its executor; wide: the earlier of the
  // lane's two - the other one stops the window like any event of another kind (t_alt)
  const LaneEvent le = lane_event(lane);
  SssEvSlot sl = le.sl;
  const int ex = le.ex;
  const uint32_t counter = g_hot.h.counter;
  uint32_t h0 = g_hot.h.rng_has32, u32_0 = g_hot.h.rng_u32;
  int pos = g_sc.rng_pos;
  const double next_arr_l = g_hot.h.next_arrival < g_hot.h.J ? g_hot.h.next_arrival_t : __builtin_inf();
  uint64_t rngv = g_sc.rng_buf[lane];  // raw output `lane` of the buffer (those from rng_pos on are unconsumed)
  const uint32_t info = sl.info;
  const int s = info_stage(info), j = info_job(info);
  bool elig = ex < f.E && info_kind(info) == EV_TASK_FINISHED;
  // the job's records: its LDS cache slot, else - more jobs with pending events than slots - the HBM copy, read here once per run and
  // written back at its end (an event pushed while its job had no slot does not name one; the job may have got one since)
  uint32_t slot = info_slot(info);
  if (elig && slot == INFO_SLOT_NONE) {
    const uint32_t k = f.slot_of[j];
    slot = k != SLOT_NONE ? k : INFO_SLOT_NONE;
  }
  const bool cached = slot != INFO_SLOT_NONE;
  int rem = 0, mc = 0, off_l = 0, off_r = 0;
  uint32_t len_l = 1, len_r = 1;
  uint64_t thr = 1ull << 53;
  uint32_t open_v = 0;  // all ones: the executor-level interval is open (the draw takes random() first)
  if (elig) {
    SssStage st;
    uint64_t local;
    int gs;
    if (cached) {
      const SssJob* jp = f.cjobs + slot;
      st = f.cstages[slot * f.SP + s], local = jp->local_mask, gs = jp->gs_base + s;
    } else {
      const SssJob* jp = g_c.jobs + j;
      st = g_c.stages[j * f.SP + s], local = jp->local_mask, gs = jp->gs_base + s;
    }
    const int n_local = local_count(local);
    int li, ri;
    executor_interval(n_local, li, ri);
    SssExDesc xd = f.exdesc[ex];
    bool xd_new = false;
    if (!(xd.gs == gs && xd.li == li && xd.ri == ri)) exdesc_fetch(f, xd, gs, li, ri), xd_new = true;
    // the level threshold of an open interval rides with the entry (it is a function of the job's executor count alone): one
    // load from the pack per change of that count instead of one per run
    if (li != ri && (int)xd.thr_n != n_local && n_local > 0 && n_local <= 100) {
      const uint64_t t = g_c.pk.lvl_thr[n_local];
      xd.thr_n = (int16_t)n_local, xd.thr_lo = (uint32_t)t, xd.thr_hi = (uint32_t)(t >> 32), xd_new = true;
    }
    if (xd_new) f.exdesc[ex] = xd;  // an entry is only ever used with its own executor's events
    rem = st.remaining, mc = (int)st.moving_to + (int)st.commit_to;
    // lists with one entry draw nothing, empty ones fail, the idle-executor fallback adds warmup_delay
    // (TPCH:88-106): all of those go one at a time
    elig = rem > 0 && n_local > 0 && n_local <= 100 && (xd.lenw_l & LENW_LEN) > 1 && (xd.lenw_r & LENW_LEN) > 1 && !(xd.lenw_l >> 30) && !(xd.lenw_r >> 30);
    if (elig) {
      off_l = xd.off_l, off_r = xd.off_r, len_l = (uint32_t)(xd.lenw_l & LENW_LEN), len_r = (uint32_t)(xd.lenw_r & LENW_LEN);
      if (li != ri) thr = (uint64_t)xd.thr_lo | ((uint64_t)xd.thr_hi << 32), open_v = 0xFFFFFFFFu;
    }
  }
  const uint32_t tag = ((uint32_t)j << 6) | (uint32_t)s;  // (job, stage): the lanes of one stage, whatever slot their event words name
  PROF3_FSEC(1);
  // wave-uniform values the loop keeps on the scalar unit
  const uint32_t counter0 = wave_lane0_u32(counter);
  h0 = wave_lane0_u32(h0), u32_0 = wave_lane0_u32(u32_0), pos = (int)wave_lane0_u32((uint32_t)pos);
  // ---- the window and the ranks in it ----
  const double t_out = min_f64(elig ? __builtin_inf() : sl.t, le.t_alt);  // what this lane holds that is not part of the run
  double t_stop = f.E <= 16 ? wave_min_f64_nonneg_row0(t_out) : wave_min_f64_nonneg(t_out);
  {
    const double na = wave_lane0_f64(next_arr_l);
    t_stop = na < t_stop ? na : t_stop;
  }
  const uint64_t elig_m = wave_ballot(elig);
  const uint64_t inw_m = wave_ballot(elig && sl.t < t_stop);
  STAT(90, 1), STAT(91, inw_m == 0), STAT(92, popc64(inw_m));
  if (inw_m == 0) return 0;
  PROF3_FSEC(2);
  // ranks among the events of the window. An event pushed to t_stop or beyond keeps a place among them (the run
  // ends before it gets there, see okm); events that start outside never get one.
  uint32_t rank = FR_OUT;
  {
    uint32_t below = 0;
    for (uint64_t m = inw_m; m; m &= m - 1) {
      const int k = ctz64_nz(m);
      const uint64_t tk = wave_readlane_u64(f64_bits(sl.t), k);  // (non-negative doubles order like their bit patterns)
      const uint32_t qk = wave_readlane_u32(sl.seq, k);
      below += (tk < f64_bits(sl.t) || (tk == f64_bits(sl.t) && qk < sl.seq)) ? 1u : 0u;
    }
    if ((inw_m >> lane) & 1ull) rank = below + 1;  // (the head has rank 1)
  }
  PROF3_FSEC(3);
  const uint64_t open_m = wave_ballot(open_v != 0);
  const char* dur_base = (const char*)f.durations;
  const int rem0 = rem;
  const uint32_t seq0 = sl.seq;
  uint32_t seq_next = counter0;
  double wall = 0.0;
  int32_t lastdur = 0;
  // Every lane's draw as if its event were the next one (TPCH:216-235 for "same stage"), under the generator state
  // (pos, h0, u32_0). The load of the duration is issued here and waited for when the head's value is needed -
  // one iteration later. When no executor of the run sits between two executor levels nobody draws random(), every
  // draw is one 32-bit half, and the level choice, the second raw output, the per-lane selects and the bookkeeping
  // of the spare half drop out of the loop (SSS_FAST_DRAW0 below: about one instruction in four).
#define SSS_FAST_DRAW(r0, r1)                                                                                             \
  do {                                                                                                                    \
    r0 = wave_readlane_u64(rngv, pos), r1 = wave_readlane_u64(rngv, pos + 1);                                             \
    const bool sel_l = (r0 >> 11) < thr; /* thr = 2^53 for a closed level interval: always */                             \
    const int off = sel_l ? off_l : off_r;                                                                                \
    const uint32_t len = sel_l ? len_l : len_r;                                                                           \
    /* numpy's spare half, or the low half of a new raw output: the one after random()'s when the interval is open */     \
    const uint32_t ua = h0 ? u32_0 : (uint32_t)r0, ux = h0 ? 0u : (uint32_t)r0 ^ (uint32_t)r1;                            \
    const uint32_t u32 = ua ^ (ux & open_v);                                                                              \
    const uint64_t mm = (uint64_t)u32 * len;                                                                              \
    dur = SSS_EXP_DUR((uint32_t)(off + (int)(mm >> 32))); /* (lanes without such an event read entry 0) */                \
    /* the head goes this way if it comes before everything else that is pending, its stage has a task left and its */    \
    /* draw passes Lemire's test at the first attempt */                                                                  \
    okm = elig_m & wave_ballot(sl.t < t_stop) & wave_ballot(rem > 0) & wave_ballot((uint32_t)mm >= len);                  \
  } while (0)
#ifdef SSS_EXP_NOLOAD  /* timing experiment only (wrong durations): what the load from the duration pool costs */
#define SSS_EXP_DUR(i) (int32_t)(((i) & 1023u) + 100u)
#else
#define SSS_EXP_DUR(i) (*(const int32_t*)(dur_base + (size_t)((i) << 2)))
#endif
  // the head of the queue commits (registers only): lane w takes its new time and push counter
#define SSS_FAST_COMMIT(w)                                                                                                \
  do {                                                                                                                    \
    const double tmin = bits_f64(wave_readlane_u64(f64_bits(sl.t), w));                                                   \
    const int32_t dur_w = (int32_t)wave_readlane_u32((uint32_t)dur, w);                                                   \
    const uint32_t tag_w = wave_readlane_u32(tag, w);                                                                     \
    const double t_new = tmin + (double)dur_w;                                                                            \
    /* the new event is the youngest: it comes after every such event that is not later (EVQ:35); those move up */        \
    const bool le = f64_bits(sl.t) <= f64_bits(t_new); /* (true for w itself: its old time) */                            \
    const uint32_t rank_w = (uint32_t)popc64(wave_ballot(le) & inw_m); /* (ranks count from 1) */                         \
    if (le) rank -= 1; /* (the lanes outside the ranking are far from 0) */                                               \
    if (lane == w) sl.t = t_new, sl.seq = seq_next, rank = rank_w;                                                        \
    if (tag == tag_w) rem -= 1, lastdur = dur_w; /* STG:53-58, ENV:604 (only read back by lanes with such an event) */    \
    seq_next++, wall = tmin;                                                                                              \
  } while (0)
  // The generator's buffer is refilled between passes of an outer loop, so that the loop over the events holds
  // wave-uniform branches only (the compiler then leaves its control flow alone: a scalar compare and branch).
#define SSS_FAST_REFILL(LAST)                                                                                             \
  do {                                                                                                                    \
    if (pos > (LAST)) {                                                                                                   \
      if (lane == 0) g_sc.rng_pos = pos;                                                                                  \
      wave_sync();                                                                                                        \
      rng_refill();                                                                                                       \
      rngv = g_sc.rng_buf[lane], pos = 0;                                                                                 \
    }                                                                                                                     \
  } while (0)
  uint64_t okm;
  int32_t dur;
  if (open_m != 0) {
    // some executor of the run draws random() first: the general form
    for (bool more = true; more;) {
      more = false;
      SSS_FAST_REFILL(62);  // a draw may take two raw outputs
      uint64_t r0, r1;
      SSS_FAST_DRAW(r0, r1);
      for (;;) {
        // the head of the queue, if it is such an event and may go this way (else: the run is over)
        const uint64_t hm = wave_ballot(rank == 1) & okm;
        if (hm == 0) break;
        const int w = ctz64_nz(hm);
        const uint32_t open_w = (uint32_t)(open_m >> w) & 1u;
        if (!h0) u32_0 = (uint32_t)((open_w ? r1 : r0) >> 32), pos += 1;  // a new raw output: its high half is kept
        h0 ^= 1u, pos += (int)open_w;
        SSS_FAST_COMMIT(w);
        if (__builtin_expect(pos > 62, 0)) {
          more = true;
          break;
        }
        SSS_FAST_DRAW(r0, r1);  // for the event after this one
      }
    }
  } else {
    // Every draw is one 32-bit half of the raw stream, in order: low(raw[p]), high(raw[p]), low(raw[p+1]), ... - the
    // loop is written two events per round, so that which half comes next is a matter of where in the loop we are.
#define SSS_FAST_DRAW0(U32)                                                                                               \
  do {                                                                                                                    \
    const uint64_t mm = (uint64_t)(uint32_t)(U32) * len_l;                                                                \
    dur = SSS_EXP_DUR((uint32_t)(off_l + (int)(mm >> 32)));                                                               \
    okm = elig_m & wave_ballot(sl.t < t_stop) & wave_ballot(rem > 0) & wave_ballot((uint32_t)mm >= len_l);                \
  } while (0)
    bool more = true;
    if (h0) {  // numpy's spare half first
      SSS_FAST_DRAW0(u32_0);
      const uint64_t hm = wave_ballot(rank == 1) & okm;
      more = hm != 0;
      if (more) SSS_FAST_COMMIT(ctz64_nz(hm));
    }
    while (more) {
      more = false;
      SSS_FAST_REFILL(63);
      SSS_FAST_DRAW0(wave_readlane_u32((uint32_t)rngv, pos));
      for (;;) {
        // (generator state here: pos, no spare half)
        const uint64_t hm = wave_ballot(rank == 1) & okm;
        if (hm == 0) break;
        SSS_FAST_COMMIT(ctz64_nz(hm));
        u32_0 = wave_readlane_u32((uint32_t)(rngv >> 32), pos);
        pos += 1;
        SSS_FAST_DRAW0(u32_0);
        // (generator state here: pos, the spare half u32_0)
        const uint64_t hm1 = wave_ballot(rank == 1) & okm;
        if (hm1 == 0) break;
        SSS_FAST_COMMIT(ctz64_nz(hm1));
        if (__builtin_expect(pos > 63, 0)) {
          more = true;
          break;
        }
        SSS_FAST_DRAW0(wave_readlane_u32((uint32_t)rngv, pos));
      }
    }
    h0 = (h0 ^ (seq_next - counter0)) & 1u;  // one half per event
#undef SSS_FAST_DRAW0
  }
#undef SSS_FAST_REFILL
#undef SSS_FAST_COMMIT
  const int total = (int)(seq_next - counter0);
  PROF3_FSEC(4);
#ifdef SSS_BATCH_STATS  // why the run ended: the window is used up / the head's stage has no task left / other
  {
    const uint64_t hr = wave_ballot(rank == 1), a = wave_ballot(sl.t < t_stop), b = wave_ballot(rem > 0);
    STAT(93, total), STAT(94, total == 0), STAT(95, (hr & ~a) != 0), STAT(96, (hr & a & ~b) != 0), STAT(97, (hr & a & b) != 0);
  }
#endif
#undef SSS_FAST_DRAW
#undef SSS_EXP_DUR
  if (total > 0) {
    const bool won = sl.seq != seq0, touched = elig && rem != rem0;  // (push counters only grow)
    if (won) g_hot.ev[ex].t = sl.t, g_hot.ev[ex].seq = sl.seq;
    if (touched) {  // (the lanes of one stage hold the same values)
      if (cached) {
        f.cstages[slot * f.SP + s].remaining = rem;
        f.cdur[slot * f.SP + s] = (float)lastdur;
        if (rem - mc <= 0) lane_atomic_or_u64(&f.cjobs[slot].sat_mask, bit64(s));  // executor demand <= 0 (ENV:566-582)
      } else {
        g_c.stages[j * f.SP + s].remaining = rem;
        g_c.durations[j * f.SP + s] = (float)lastdur;
        if (rem - mc <= 0) lane_atomic_or_u64(&g_c.jobs[j].sat_mask, bit64(s));
      }
    }
    // a stage whose last task was started in this run is saturated from now on (ENV:595-597): once per stage
    for (uint64_t zm = wave_ballot(touched && rem == 0); zm;) {
      const int l = ctz64_nz(zm);
      const uint32_t tl = wave_readlane_u32(tag, l);
      if (lane == l) {  // sat_count++ (upper half of the word)
        if (cached) lane_atomic_add_u32((uint32_t*)&f.cjobs[slot].supply, 1u << 16);
        else lane_atomic_add_u32((uint32_t*)&g_c.jobs[j].supply, 1u << 16);
      }
      zm &= ~wave_ballot(touched && tag == tl);
    }
    if (lane == 0) {
      SssHdr& h = g_hot.h;
      h.wall_time = wall;  // the last event's time
      h.counter = counter0 + (uint32_t)total;
      h.n_events += (uint64_t)total, h.n_fast += (uint64_t)total, h.n_batched += (uint64_t)total, h.n_rounds += 1;
      g_sc.events_this_step += (int32_t)total;
      g_sc.rng_pos = pos;
      h.rng_has32 = h0;
      h.rng_u32 = u32_0;
    }
  }
  wave_sync();  // the slots and counters are visible to every lane from here
  PROF3_FSEC(5);
  PROF3_CALLS(30, total - 1);  // (profiling builds: ticks per event of a run)
  return total;
}

// ------------------------------------------------------------------------------------------
// Batches of RELEASED executors (all lanes). The other frequent event while nothing is committable:
// TASK_FINISHED on a stage with no task left to start (ENV:468-483) whose pool holds a commitment
// (the policy lined the executor's next stop up). The executor leaves its stage's pool, the commitment
// is settled (TRK:159-176) and, by destination (ENV:699-712, 784-819, 745-782):
//   START   another stage of its job, in the frontier: it moves into that stage's pool and starts a task
//           (a duration draw, a new TASK_FINISHED event);
//   PARK    another stage of its job, not yet in the frontier: it waits in the job's pool (no event);
//   SEND    a stage of another job: it is detached from its job and travels (EXECUTOR_READY after moving_delay);
//   IDLE    the common pool: it goes to the job's pool, or - the job being saturated - is detached into the
//           common pool (no event).
// The source stays what it is (ENV:662-674), nothing becomes committable, the loop goes on. The
// construction: every pending event that can go this way computes a LOWER BOUND of the time of the event it
// will push (its own time + the minimum of the duration lists it can draw from, or moving_delay); M = min over
// those bounds, the times of all pending events that need the general handlers, and the next arrival. Every
// member event with t < M is popped before anything else can happen, and what they push lands at >= M: that set
// is the batch. Members rank themselves by (time, push counter) - heapq's order, EVQ:35 - in a v_readlane loop;
// the rank gives the push counter and - the number of raw generator outputs a start consumes being known
// beforehand - its position in the env's random stream, which the wave has produced ahead of time (rng_refill);
// counters of stages and pools follow from counts. Left to the one-event path: the event that completes
// its stage (frontier changes), pools without or with exhausted commitments, destination stages short
// of tasks (backup scheduling), jobs without a cache slot, and members whose outcome would depend on
// an earlier member of the same job (a start after a detachment: the job's executor count enters the
// draw; an idle executor after a start: the job's saturation decides where it goes).
// Returns the number of events handled (0: none, nothing modified).
// ------------------------------------------------------------------------------------------
#define RL_NO_COMMITMENT 0xFFu  // rl_idx of a member whose pool holds no commitment
// candidates in the window below which the events go one by one (lean_released / lean_arrival). Measured at BASELINE config 3,
// step launches (profiles/r04_bench.md): 2 / 2 0.342 ms, 3 / 2 0.343, 4 / 3 0.346, 6 / 3 0.351 - a batch of two already beats two
// single events; config 2 does not care (0.171 ms throughout)
#ifndef SSS_MIN_RELEASED_BATCH
#define SSS_MIN_RELEASED_BATCH 2
#endif
#ifndef SSS_MIN_ARRIVAL_BATCH
#define SSS_MIN_ARRIVAL_BATCH 2
#endif
// executor count from which the batches take their pools through the pair staging (pool_pair_*) when all members share them:
// with few executors nearly every pool image has 8 slots and lives in its 16-byte record, where the per-lane register paths
// (pool_leave_many / pool_enter_many / pool_pass_many) are cheaper than staging
#ifndef SSS_PAIR_MIN_E
#define SSS_PAIR_MIN_E 1
#endif
// One lane per pool (batch_released_events): every member of ranks [0, n) that leaves pool `okey` is taken out
// of it - one fetch and one store of the pool's record; removals commute - and the pool's outgoing
// commitments shrink by as many.
// every member's own lane, for pools with more than 8 slots: removals commute and touch one slot each
SSS_DEV void pool_leave_table(uint32_t okey, uint32_t e) {
  const uint32_t mask = g_c.pool_hdr[pool_index(okey)].mask;
  if (mask == 7) return;
  bool was = table_mark_dummy(pool_table_hbm(okey), mask, e);
  CHECK(was);
}
SSS_DEV void pool_leave_many(uint32_t okey, uint32_t n) {
  SssPoolHdr* hd = g_c.pool_hdr + pool_index(okey);
  uint4 rec = *(const uint4*)hd;
  const uint32_t mask = rec.x & 0xFFFFu;
  uint32_t used = rec.y & 0xFFFFu, aux = rec.y >> 16;
  uint64_t t = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
  for (uint32_t q = 0; q < n; q++) {
    if (g_sc.rl_old[q] != okey) continue;
    if (mask == 7) {
      bool was = set8_remove(t, used, (uint32_t)g_sc.fi_e[q]);
      CHECK(was);
    } else
      used--;  // the member's own lane has marked its slot of the table (pool_leave_table)
    if (g_sc.rl_idx[q] != RL_NO_COMMITMENT) aux--;
  }
  *(uint4*)hd = mk_u4(rec.x, (used & 0xFFFFu) | (aux << 16), mask == 7 ? (uint32_t)t : 0u, mask == 7 ? (uint32_t)(t >> 32) : 0u);
}
// ... and every member that enters pool `nkey` is added, in rank order. Returns false, with nothing done, unless
// the image has 8 slots and keeps them (larger tables and growth go through the LDS staging area, pools_staged).
SSS_DEV bool pool_enter_many(uint32_t nkey, uint32_t n) {
  SssPoolHdr* hd = g_c.pool_hdr + pool_index(nkey);
  const uint4 rec = *(const uint4*)hd;
  if ((rec.x & 0xFFFFu) != 7) return false;
  uint32_t fill = rec.x >> 16, used = rec.y & 0xFFFFu;
  uint32_t cnt = 0;
  for (uint32_t q = 0; q < n; q++) cnt += g_sc.fc_dst[q] == nkey ? 1u : 0u;
  if ((fill + cnt) * 5 >= 7 * 3) return false;
  uint64_t t = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
  for (uint32_t q = 0; q < n; q++)
    if (g_sc.fc_dst[q] == nkey) set8_add(t, fill, used, (uint32_t)g_sc.fi_e[q]);
  *(uint4*)hd = mk_u4(7u | (fill << 16), (used & 0xFFFFu) | (rec.y & 0xFFFF0000u), (uint32_t)t, (uint32_t)(t >> 32));
  return true;
}
// All lanes: the pools the lanes of `dm` speak for, one at a time through the LDS staging area. ENTER: the members
// whose fc_dst is the pool are added in rank order. PASS (arriving executors, their job's pool, rl_old): each
// enters and leaves again, or - parked - is taken out and put back by the move to the pool it is already in.
enum { STAGED_ENTER = 0, STAGED_PASS = 1 };
// All lanes: the pools the lanes of `dm` speak for, one at a time. `mykey`: the pool this lane's executor enters
// (ENTER) or passes through (PASS), POOL_NONE for lanes that are not members. ENTER: the members are added in rank
// order. PASS (arriving executors, their job's pool): each enters and leaves again, or - `parks` - is taken out and
// put back by the move to the pool it is already in. Through the LDS staging area, the operations applied in rank order
// (CPython puts a key on the LAST dummy of its probe run: with dummies about, additions do not commute), each with the
// whole wave (staged_add / staged_remove).
template <int MODE>
SSS_DEV void pools_staged(uint64_t dm, uint32_t n, uint32_t mykey, bool parks) {
  while (dm) {
    const int l = ctz64_nz(dm);
    dm &= dm - 1;
    const uint32_t key = wave_readlane_u32(mykey, l);
    tabword_t key_words;
    SetImg<uint8_t> sn = pool_stage_in(key, key_words);
    for (uint32_t q = 0; q < n; q++) {  // (wave-uniform: the lists are read by every lane)
      const uint32_t e = g_sc.fi_e[q];
      if (MODE == STAGED_ENTER) {
        if (g_sc.fc_dst[q] == key) staged_add(sn, e);
      } else if (g_sc.rl_old[q] == key) {
        staged_add(sn, e);
        bool was = staged_remove(sn, e);
        CHECK(was);
        if (g_sc.fi_type[q] == 1 /* AR_PARK */) staged_add(sn, e);
      }
    }
    wave_sync();
    pool_stage_out(key, sn, key_words);
  }
}

enum { RL_START = 0, RL_PARK = 1, RL_SEND = 2, RL_IDLE_JOB = 3, RL_IDLE_COMMON = 4, RL_FREE_JOB = 5, RL_FREE_COMMON = 6 };
// all lanes: is there a schedulable stage whatever the source (ENV:505-555 without the source job's exemption) -
// an active job below the executor cap (ENV:526-531) with a ready, unsaturated, unselected stage?
SSS_DEV bool any_schedulable_without_source() {
  const int lane = wave_lane();
  const int A = g_hot.h.n_active;
  bool any = false;
  for (int a0 = 0; a0 < A; a0 += 64) {
    const int a = a0 + lane;
    if (a < A) {
      const SssJob* job = jobp(lds_active()[a]);
      if ((int)job->supply < g_c.E && ready_mask_of_job(*job, true) != 0) any = true;
    }
  }
  return wave_ballot(any) != 0;
}
SSS_DEV int batch_released_events(const FastCtx& f, int head) {
  UTRACE("batch_released");
#ifdef SSS_NO_BATCH
  return 0;
#endif
  PROF3(0);
  PROF3_SEC_BEGIN;
  const int lane = wave_lane();
  // ---- reads ----
  const LaneEvent le = lane_event(lane);  // (wide: the earlier of the lane's two events; the other one bounds the window, t_alt)
  const SssEvSlot sl = le.sl;
  const int ex = le.ex, hl = head_lane(head);
  const uint32_t counter0 = g_hot.h.counter, h0 = g_hot.h.rng_has32, u32_0 = g_hot.h.rng_u32;
  const int pos = g_sc.rng_pos;
  const int n_commits = g_hot.h.n_commits;
  const double next_arr = g_hot.h.next_arrival < g_hot.h.J ? g_hot.h.next_arrival_t : __builtin_inf();
  const uint32_t info = sl.info;
  const uint32_t slot = info_slot(info);
  const int s = info_stage(info), j = info_job(info);
  const bool tfc = ex < f.E && info_kind(info) == EV_TASK_FINISHED && slot != INFO_SLOT_NONE;
  SssStage st_old = {0, 0, 0, 0};
  if (tfc) st_old = f.cstages[slot * f.SP + s];
  // an executor whose departure does not complete its stage (that one changes the frontier: general path).
  // With a source pool set, an executor entering it would become committable (ENV:331-338, TRK:107-113): such a
  // member goes the general way (below). Leaving the source takes one of its commitments along: no change.
  const uint32_t source = g_hot.h.curr_source;
  bool cand = tfc && st_old.remaining == 0 && st_old.executing >= 2 && g_hot.ex_job[ex] == j;
  {
    const double kq = min_f64(cand ? __builtin_inf() : sl.t, le.t_alt);
    const double t_other = f.E <= 16 ? wave_min_f64_nonneg_row0(kq) : wave_min_f64_nonneg(kq);
    const double t_stop = next_arr < t_other ? next_arr : t_other;
    const uint64_t pre = wave_ballot(cand && sl.t < t_stop);
    // none or too few (a single one goes the wave-uniform single-event way, lean_released), or not the head
    if (popc64(pre) < SSS_MIN_RELEASED_BATCH || !((pre >> hl) & 1ull)) { STAT(64, 1); return 0; }
  }
  PROF3_SEC(1);
  // the commitment its pool would serve first (TRK:178-183: the first-inserted one of that source)
  const uint32_t sp = key_stage_pool(j, s);
  int c_idx = -1, c_cnt = 0;
  uint32_t dst = POOL_NONE, c_best = 0xFFFFFFFFu;
  for (int i = 0; i < n_commits; i++) {
    const uint32_t cs = g_hot.c_src[i], cq = g_hot.c_seq[i];
    if (cand && cs == sp && cq < c_best) c_best = cq, c_idx = i, dst = g_hot.c_dst[i], c_cnt = g_hot.c_n[i];
  }
  // no commitment: the executor has nowhere to go (ENV:655-659). It becomes the source (ENV:662-674), and if nothing
  // is schedulable then - which the members' own jobs (below) and one scan of the others (further below) establish,
  // and which stays so while only such executors and idled ones are processed - it is moved to its job's pool or,
  // the job being saturated, to the common pool, and the source is cleared (ENV:331-341, 745-782)
  const bool freed = cand && c_idx < 0;
  cand = cand && (freed || dst != sp);
  // the head of the queue has to be a member: whenever it turns out not to be one, the round is over
  if (!((wave_ballot(cand) >> hl) & 1ull)) { STAT(65, 1); return 0; }
  PROF3_SEC(2);
  const int j2 = key_job(dst), s2 = key_stage(dst);
  int type = RL_START;
  SssStage st_new = {0, 0, 0, 0};
  SssStage* sp_new = nullptr;
  bool open = false;
  int n_local = 0, li = 0, ri = 0;
  int4 da = mk_i4(0, 0, 0, 0), db = da;
  if (cand) {
    const SssJob* jp = f.cjobs + slot;
    if (freed) {
      type = (int)jp->sat_count == (int)jp->n_stages ? RL_FREE_COMMON : RL_FREE_JOB;
      cand = ready_mask_of_job(*jp, true) == 0;  // its own job passes the filter as the source's job (ENV:526-531)
    } else if (dst == POOL_COMMON) {
      type = (int)jp->sat_count == (int)jp->n_stages ? RL_IDLE_COMMON : RL_IDLE_JOB;  // JOB:53-55
    } else if (s2 < 0) {
      cand = false;  // (commitments name stages or the common pool)
    } else {
      sp_new = j2 == j ? f.cstages + slot * f.SP + s2 : stgp(j2, s2);
      st_new = *sp_new;
      cand = st_new.remaining > 0;  // else: backup scheduling (ENV:784-797)
      type = j2 != j ? RL_SEND : ((jp->frontier_mask & bit64(s2)) ? RL_START : RL_PARK);
      if (cand && type == RL_START) {  // TPCH:75-106: the executor's last task was on another stage of the job
        n_local = local_count(jp->local_mask);
        executor_interval(n_local, li, ri);
        open = li != ri;
        const int gs2 = jp->gs_base + s2;
        da = *(const int4*)eff_row(f.eff, gs2, li, 2);
        db = open ? *(const int4*)eff_row(f.eff, gs2, ri, 2) : da;
        cand = n_local > 0 && (da.y & LENW_LEN) > 1 && (db.y & LENW_LEN) > 1;
      }
    }
  }
  if (!((wave_ballot(cand) >> hl) & 1ull)) { STAT(66, 1); return 0; }
  if (wave_ballot(cand && freed) != 0) {
    // executors idling in a stage's pool would move along with a freed one (ENV:714-728): there are none between events
    bool idle_in_stage = false;
    for (int x = lane; x < f.E; x += 64)
      idle_in_stage = idle_in_stage || (!g_hot.ex_executing[x] && g_hot.ex_loc[x] != POOL_NONE && g_hot.ex_loc[x] != POOL_COMMON && key_stage(g_hot.ex_loc[x]) >= 0);
    if (wave_ballot(idle_in_stage) != 0 || any_schedulable_without_source()) cand = cand && !freed;
  }
  const bool start = type == RL_START, pusher = type == RL_START || type == RL_SEND;
  const bool detach = type == RL_SEND || type == RL_IDLE_COMMON || type == RL_FREE_COMMON;
  const bool idle = type == RL_IDLE_JOB || type == RL_IDLE_COMMON;        // settles a commitment to the common pool
  const bool rests = idle || type == RL_FREE_JOB || type == RL_FREE_COMMON;  // ends up waiting in the job's / the common pool
  // the pool the member enters
  const uint32_t enters = start ? dst : (type == RL_SEND ? POOL_NONE : ((type == RL_IDLE_COMMON || type == RL_FREE_COMMON) ? POOL_COMMON : key_job_pool(j)));
  cand = cand && (freed || source == POOL_NONE || enters != source);
  if (!((wave_ballot(cand) >> hl) & 1ull)) { STAT(67, 1); return 0; }
  PROF3_SEC(3);
  // when the event a member pushes can come at the earliest
  const double push_lb = start ? (double)(da.z < db.z ? da.z : db.z) : (type == RL_SEND ? g_c.P.moving_delay : __builtin_inf());
  const double key = min_f64(cand ? sl.t + push_lb : sl.t, le.t_alt);
  double M = f.E <= 16 ? wave_min_f64_nonneg_row0(key) : wave_min_f64_nonneg(key);
  if (next_arr < M) M = next_arr;
  bool V = cand && sl.t < M;
  uint64_t vm = wave_ballot(V);
  if (vm == 0) { STAT(68, 1); return 0; }
  const uint32_t nmax = (uint32_t)(64 - pos) >> 1;
  const uint32_t tag_old = (slot << 8) | (uint32_t)s;
  const uint32_t tag_new = (rests ? 0x1FFFFu : (((uint32_t)j2 << 6) | (uint32_t)s2)) | (start ? 0x20000u : 0u) | (open ? 0x40000u : 0u) |
                           (detach ? 0x80000u : 0u) | (pusher ? 0x100000u : 0u) | (type == RL_PARK ? 0x200000u : 0u);
  // rank among all members / among the pushers / among the starters; starters before with an open level
  // interval; members before that leave the same stage; starters before on the same new stage; members of
  // the same job before that detach from it / start a task
  // One sweep over the members. Everything a member needs is a count over the members BEFORE it - except ct_take, the
  // starters of its new stage in the whole batch, which is kept as a lane mask. When members have to go (the first one that
  // completes its stage / finds its commitment used up / its new stage dry / depends on an earlier member of its job / runs
  // out of buffered randomness, and everybody after it), the survivors' counts do not change - their predecessors all
  // survive - so there is no second sweep: the mask is intersected with the survivors.
  uint32_t rank = 0, rank_p = 0, rank_x = 0, R = 0, cb_old = 0, cb_take = 0, ct_take, det_job = 0, start_job = 0, stir = 0;
  uint64_t take_m = 0;
  for (uint64_t m = vm; m; m &= m - 1) {
    const int k = ctz64(m);
    const double tk = wave_readlane_f64(sl.t, k);
    const uint32_t qk = wave_readlane_u32(sl.seq, k);
    const uint32_t ok = wave_readlane_u32(tag_old, k), nk = wave_readlane_u32(tag_new, k);
    const bool lt = tk < sl.t || (tk == sl.t && qk < sl.seq);
    const bool xk = (nk & 0x20000u) != 0, same_new = ((nk ^ tag_new) & 0x1FFFFu) == 0, same_old = ok == tag_old, same_job = ((ok ^ tag_old) >> 8) == 0;
    rank += lt ? 1u : 0u;
    rank_p += (lt && (nk & 0x100000u)) ? 1u : 0u;
    rank_x += (lt && xk) ? 1u : 0u;
    R += (lt && (nk & 0x40000u)) ? 1u : 0u;
    cb_old += (lt && same_old) ? 1u : 0u;
    cb_take += (lt && xk && same_new) ? 1u : 0u;
    take_m |= (xk && same_new) ? bit64(k) : 0ull;
    det_job += (lt && same_job && (nk & 0x80000u)) ? 1u : 0u;
    start_job += (lt && same_job && xk) ? 1u : 0u;
    stir += (lt && (nk & 0x300000u)) ? 1u : 0u;  // members before that change a stage's demand or a job's executor count
  }
  {
    // completes its stage / the commitment is used up / the new stage runs dry / depends on an earlier member of its job / randomness
    const bool over = V && ((int)cb_old + 2 > (int)st_old.executing || (!freed && (int)cb_old >= c_cnt) || (!rests && (int)cb_take >= (int)st_new.remaining) ||
                            (start && det_job > 0) || (rests && start_job > 0) || (freed && stir > 0) || rank_x >= nmax);
    if (wave_ballot(over) != 0) {
      const uint32_t rcut = wave_min_u32(over ? rank : 0xFFFFFFFFu);
      V = V && rank < rcut;
      vm = wave_ballot(V);
      if (vm == 0) { STAT(69, 1); return 0; }
    }
    ct_take = (uint32_t)popc64(take_m & vm);
  }
  PROF3_SEC(4);
  const uint32_t n = (uint32_t)popc64(vm);
  const uint32_t n_x = (uint32_t)popc64(wave_ballot(V && start)), n_p = (uint32_t)popc64(wave_ballot(V && pusher));
  const uint32_t n_idle = (uint32_t)popc64(wave_ballot(V && idle));
  const uint64_t freed_m = wave_ballot(V && freed);
  const bool any_freed = freed_m != 0;
  STAT(46, popc64(freed_m));
  // ---- the starters' draws ----
  const uint32_t Fr = h0 ? rank_x >> 1 : (rank_x + 1) >> 1;
  const bool fresh = ((h0 + rank_x) & 1u) == 0;
  const uint32_t P = R + Fr;
  const bool vx = V && start;
  int4 dd = da;
  uint64_t x32 = 0;
  uint32_t u32 = 0;
  if (vx) {
    if (open) {
      const double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
      const int rand_pt = 1 + (int)(u64_to_unit(g_sc.rng_buf[pos + (int)P]) * (right - left));
      if (!((double)rand_pt <= (double)n_local - left)) dd = db;
    }
    if (fresh) {
      x32 = g_sc.rng_buf[pos + (int)P + (open ? 1 : 0)];
      u32 = (uint32_t)x32;
    } else if (rank_x == 0) {
      u32 = u32_0;
    } else {
      u32 = (uint32_t)(g_sc.rng_buf[pos + (int)R + (int)Fr - 1] >> 32);
    }
  }
  const uint32_t len = (uint32_t)(dd.y & LENW_LEN);
  const uint64_t mm = (uint64_t)u32 * len;
  if (wave_ballot(vx && (uint32_t)mm < len) != 0) { STAT(70, 1); return 0; }
  // ---- commit ----
  if (V) {
    SssJob* jp = f.cjobs + slot;
    lane_atomic_add_u32((uint32_t*)(f.cstages + slot * f.SP + s) + 1, 0u - STG_W1_EXECUTING);  // executing-- (STG:60-62)
    g_sc.fi_e[rank] = (uint8_t)ex, g_sc.fi_type[rank] = (uint8_t)type;
    g_sc.rl_old[rank] = sp, g_sc.rl_idx[rank] = freed ? (uint8_t)RL_NO_COMMITMENT : (uint8_t)c_idx, g_sc.rl_seq[rank] = c_best;
    g_sc.fc_dst[rank] = enters;
    if (start) {
      double dur = (double)f.durations[dd.x + (int)(mm >> 32)];
      if (dd.y >> 30) dur += g_c.P.warmup_delay;
      // commitments to the new stage: one fewer (TRK:159-176); remaining--, executing++ (STG:53-58): one 64-bit addition, no field underflows
      lane_atomic_add_u64((uint64_t*)sp_new, ((uint64_t)(STG_W1_EXECUTING - STG_W1_COMMIT_TO) << 32) - 1ull);
      g_hot.ev[ex].t = sl.t + dur;
      g_hot.ev[ex].seq = counter0 + rank_p;
      g_hot.ev[ex].info = ev_info(EV_TASK_FINISHED, j, s2, slot);
      g_hot.ex_task_stage[ex] = (int8_t)s2, g_hot.ex_loc[ex] = dst;
      if (cb_take + 1 == ct_take) {  // the new stage's last starter of the batch
        f.cdur[slot * f.SP + s2] = (float)dur;
        if ((int)st_new.remaining - (int)ct_take == 0) lane_atomic_add_u32((uint32_t*)&jp->supply, 1u << 16);  // sat_count++ (ENV:595-597)
      }
    } else if (type == RL_SEND) {
      lane_atomic_add_u32((uint32_t*)sp_new + 1, STG_W1_MOVING_TO - STG_W1_COMMIT_TO);  // commit_to--, moving_to++
      g_hot.ev[ex].t = sl.t + g_c.P.moving_delay;
      g_hot.ev[ex].seq = counter0 + rank_p;
      g_hot.ev[ex].info = ev_info(EV_EXECUTOR_READY, j2, s2, (uint32_t)lds_slot_of()[j2]);
      g_hot.ex_executing[ex] = 0, g_hot.ex_loc[ex] = POOL_NONE;
      lane_atomic_add_u32((uint32_t*)&jp->supply, 0u - 1u);          // the old job's executor count (TRK:218-221)
    } else {
      if (type == RL_PARK) {
        lane_atomic_add_u32((uint32_t*)sp_new + 1, 0u - STG_W1_COMMIT_TO);
        g_hot.ex_task_stage[ex] = -1;  // ENV:808-813
      }
      g_hot.ev[ex].t = __builtin_inf();
      g_hot.ev[ex].info = EV_NONE;
      if (freed) g_hot.ex_task_stage[ex] = -1;  // executor.task = None (ENV:655-656)
      g_hot.ex_executing[ex] = 0, g_hot.ex_loc[ex] = enters;
    }
    if (detach) {  // JOB:86-89
      local_atomic_detach(jp, ex);
      g_hot.ex_job[ex] = -1, g_hot.ex_task_stage[ex] = -1;
    }
    if (rank == n - 1) {
      SssHdr& h = g_hot.h;
      h.wall_time = sl.t;
      h.counter = counter0 + n_p;
      h.n_events += n, h.n_batched += n, h.n_rounds++;
      h.supply_none -= (int32_t)n_idle;  // TRK:159-176: a commitment to the common pool counted as its supply
      g_sc.events_this_step += (int32_t)n;
      if (any_freed) h.curr_source = POOL_NONE, g_sc.idle_valid = 0;  // ENV:341, after whichever freed executor came last
    }
    if (vx && rank_x == n_x - 1) {
      g_sc.rng_pos = pos + (int)P + (open ? 1 : 0) + (fresh ? 1 : 0);
      g_hot.h.rng_has32 = fresh ? 1u : 0u;
      g_hot.h.rng_u32 = fresh ? (uint32_t)(x32 >> 32) : u32;
    }
  }
  wave_sync();
  PROF3(13);
  PROF3_SEC(5);
  // The usual batch: executors of ONE stage that finish close together - they leave the same pool, take the same commitment
  // (the pool's first-inserted one) and go the same way. One entry of the commitment list shrinks by n, the two pool images
  // come in with one round trip (pool_pair_*: n removals, n additions in rank order), the cache-slot references move in one go.
  const uint32_t sp_h = wave_readlane_u32(sp, hl), en_h = wave_readlane_u32(enters, hl);
  const int type_h = (int)wave_readlane_u32((uint32_t)type, hl);
  const bool uniform = f.E >= SSS_PAIR_MIN_E && pair_staging_fits(f.E) && wave_ballot(V && (sp != sp_h || enters != en_h || type != type_h)) == 0;
  if (uniform) {
    const bool freed_h = wave_readlane_u32(freed ? 1u : 0u, hl) != 0;
    const int ci_h = (int)wave_readlane_u32((uint32_t)c_idx, hl);
    const uint32_t slot_h = wave_readlane_u32(slot, hl);
    const PoolPairRegs pr = pool_pair_fetch(sp_h, en_h, en_h != POOL_NONE);
    if (lane == 0) {
      if (!freed_h) {  // TRK:159-176, n times: dict.pop when the entry is used up (swap-remove, the order lives in c_seq)
        const int left = (int)g_hot.c_n[ci_h] - (int)n;
        CHECK(left >= 0);
        g_hot.c_n[ci_h] = (int16_t)left;
        if (left == 0) {
          const int last = H.n_commits - 1;
          g_hot.c_src[ci_h] = g_hot.c_src[last], g_hot.c_dst[ci_h] = g_hot.c_dst[last], g_hot.c_n[ci_h] = g_hot.c_n[last], g_hot.c_seq[ci_h] = g_hot.c_seq[last];
          H.n_commits = last;
        }
      }
      if (type_h != RL_START) {  // their events are gone, or name another job: that many references to the old job's cache slot fewer
        lds_slot_ref()[slot_h] = (uint8_t)(lds_slot_ref()[slot_h] - n);
        if (type_h == RL_SEND) {
          const uint32_t ns = info_slot(g_hot.ev[g_sc.fi_e[0]].info);
          if (ns != INFO_SLOT_NONE) lds_slot_ref()[ns] = (uint8_t)(lds_slot_ref()[ns] + n);
        }
      }
    }
    PairImg so, sn;
    pool_pair_stage(pr, en_h != POOL_NONE, so, sn);
    pair_remove_many(so, g_sc.fi_e, 0, (int)n);  // (removals commute: every member's own lane)
    if (en_h != POOL_NONE)
      for (uint32_t q = 0; q < n; q++) pair_add(sn, (uint32_t)g_sc.fi_e[q]);  // rank order (wave-uniform: every lane reads the list)
    if (!freed_h) so.s.aux -= n;  // the pool's outgoing commitments
    wave_sync();
    pool_pair_flush_one(sp_h, so);
    if (en_h != POOL_NONE) pool_pair_flush_one(en_h, sn);
    STAT(31, 1), STAT(33, n), STAT(127, 1);
    wave_sync();
  } else {
  if (lane == 0) {
    // commitments (in rank order, so that entries disappear in the order the one-event path removes them) and slot references
    for (uint32_t r = 0; r < n; r++) {
      const uint32_t okey = g_sc.rl_old[r];
      int ci = g_sc.rl_idx[r];
      if (ci == (int)RL_NO_COMMITMENT) {
        const int ks = lds_slot_of()[key_job(okey)];
        if (ks != SLOT_NONE) lds_slot_ref()[ks]--;
        continue;
      }
      if (!(ci < H.n_commits && g_hot.c_src[ci] == okey && g_hot.c_seq[ci] == g_sc.rl_seq[r])) {  // entries have moved (swap-remove)
        ci = -1;
        for (int i = 0; i < H.n_commits; i++)
          if (g_hot.c_src[i] == okey && g_hot.c_seq[i] == g_sc.rl_seq[r]) ci = i;
      }
      CHECK(ci >= 0);
      if (ci >= 0) {
        g_hot.c_n[ci] = (int16_t)(g_hot.c_n[ci] - 1);
        if (g_hot.c_n[ci] == 0) {
          int last = H.n_commits - 1;
          g_hot.c_src[ci] = g_hot.c_src[last], g_hot.c_dst[ci] = g_hot.c_dst[last], g_hot.c_n[ci] = g_hot.c_n[last], g_hot.c_seq[ci] = g_hot.c_seq[last];
          H.n_commits = last;
        }
      }
      if (g_sc.fi_type[r] != RL_START) {  // its event is gone, or names another job: one reference to the old job's cache slot fewer
        const int ks = lds_slot_of()[key_job(okey)];
        if (ks != SLOT_NONE) lds_slot_ref()[ks]--;
        if (g_sc.fi_type[r] == RL_SEND) {
          const uint32_t ns = info_slot(g_hot.ev[g_sc.fi_e[r]].info);
          if (ns != INFO_SLOT_NONE) lds_slot_ref()[ns]++;
        }
      }
    }
  }
  PROF3_SEC(6);
  // pools: one lane per pool, all pools at once. A member speaks for the pool it leaves / enters if no
  // member before it (in rank) shares that pool.
  bool deferred = false;
  if (V) {
    pool_leave_table(sp, (uint32_t)ex);
    if (cb_old == 0) pool_leave_many(sp, n);
    const uint32_t nkey = g_sc.fc_dst[rank];
    bool lead = nkey != POOL_NONE;
    for (uint32_t q = 0; q < rank; q++) lead = lead && g_sc.fc_dst[q] != nkey;
    if (lead) deferred = !pool_enter_many(nkey, n);
  }
  uint64_t dm = wave_ballot(deferred);
  STAT(31, 1), STAT(32, popc64(dm)), STAT(33, n);
  wave_sync();
  PROF3_SEC(7);
  pools_staged<STAGED_ENTER>(dm, n, V ? enters : POOL_NONE, false);  // tables with more than 8 slots, or about to grow
  }
  PROF3_SEC(8);
  if (any_freed) {
    // every scan that found nothing left schedulable_stages empty (ENV:333, 505-540)
    const int A = g_hot.h.n_active;
    for (int a = lane; a < A; a += 64) {
      SssJob* job = jobp(lds_active()[a]);
      if (job->sched_mask) job->sched_mask = 0;
    }
  }
  wave_sync();
  PROF3_SEC(9);
  // saturation bits (ENV:566-582): a parked executor's commitment is gone and it did not reach the stage
  if (V && type == RL_PARK) {
    const SssStage t2 = f.cstages[slot * f.SP + s2];
    SssJob* jp = f.cjobs + slot;
    if ((int)t2.remaining - ((int)t2.moving_to + (int)t2.commit_to) <= 0)
      lane_atomic_or_u64(&jp->sat_mask, bit64(s2));
    else
      lane_atomic_and_u64(&jp->sat_mask, ~bit64(s2));
  }
  wave_sync();
  PROF3_SEC(10);
  if (wave_ballot(V && type == RL_SEND) != 0) {
    // A job with a pending event holds a cache slot if there is one to have (push_event): the jobs executors were
    // sent to get theirs now, so that the arrivals find their job in LDS (and can be batched in their turn). Last
    // thing in the batch: handing a slot on may write another job's records back, and nothing above may point
    // into a slot any more by then.
    if (lane == 0) {
      for (uint32_t r = 0; r < n; r++) {
        if (g_sc.fi_type[r] != RL_SEND) continue;
        const int e = g_sc.fi_e[r];
        const uint32_t inf = g_hot.ev[e].info;
        if (info_slot(inf) != INFO_SLOT_NONE) continue;
        const int k = cache_acquire(info_job(inf));
        if (k == SLOT_NONE) continue;
        g_hot.ev[e].info = info_with_slot(inf, (uint32_t)k);
        lds_slot_ref()[k]++;
      }
    }
    wave_sync();
  }
  PROF3_SEC(11);
  return (int)n;
}

// ------------------------------------------------------------------------------------------
// Batches of ARRIVING executors (all lanes). Executors sent to a job in one fulfilment arrive together
// (same moving_delay, ENV:617-637), and while no source is set their EXECUTOR_READY events (ENV:440-450)
// do not interact beyond the counters of their job and stage: the executor is attached to the job
// (JOB:81-84), passes through the job's pool (TRK:188-222) and
//   START   its stage is in the frontier and has a task left: it enters the stage's pool and starts one
//           (an idle executor's draw, TPCH:75-106 - the job's executor count includes every member that
//           arrived before it; a new TASK_FINISHED event);
//   PARK    its stage is not in the frontier yet: it waits in the job's pool (ENV:808-813, no event).
// A stage that has run out of tasks (backup scheduling, ENV:784-797) ends the batch. Same construction as
// the other batches: window below everything that is not a member and below what members can push, members
// ranked by (time, push counter), draws and counters by rank, one lane per pool for the set images.
// Returns the number of events handled (0: none, nothing modified).
// ------------------------------------------------------------------------------------------
enum { AR_START = 0, AR_PARK = 1 };
// One lane per job: every member of that job enters the job's pool and leaves it again (START) or is taken out
// and put back by the move to the pool it is already in (PARK, TRK:188-222 with old == new), in rank order.
// Returns false, with nothing done, unless the image has 8 slots and stays that way.
SSS_DEV bool pool_pass_many(uint32_t jkey, uint32_t n) {
  SssPoolHdr* hd = g_c.pool_hdr + pool_index(jkey);
  const uint4 rec = *(const uint4*)hd;
  if ((rec.x & 0xFFFFu) != 7) return false;
  uint32_t fill = rec.x >> 16, used = rec.y & 0xFFFFu;
  uint64_t t = (uint64_t)rec.z | ((uint64_t)rec.w << 32);
  for (uint32_t q = 0; q < n; q++) {
    if (g_sc.rl_old[q] != jkey) continue;
    const uint32_t e = g_sc.fi_e[q];
    if (set8_add(t, fill, used, e)) {
      // set_table_resize(used * 4): 8 slots again while the executor is alone in the pool - rebuilt
      // without the dummies, i.e. the one key in its home slot
      if (used >= 2) return false;
      t = (uint64_t)(e + 2) << (8 * (e & 7)), fill = used = 1;
    }
    bool was = set8_remove(t, used, e);
    CHECK(was);
    if (g_sc.fi_type[q] == AR_PARK) set8_add(t, fill, used, e);  // lands on a dummy: no growth
  }
  *(uint4*)hd = mk_u4(7u | (fill << 16), (used & 0xFFFFu) | (rec.y & 0xFFFF0000u), (uint32_t)t, (uint32_t)(t >> 32));
  return true;
}

SSS_DEV int batch_arrival_events(const FastCtx& f, int head) {
  UTRACE("batch_arrival");
#ifdef SSS_NO_BATCH
  return 0;
#endif
  PROF3(20);
  PROF3_SEC_BEGIN;
  const int lane = wave_lane();
  // ---- reads ----
  const LaneEvent le = lane_event(lane);  // (wide: the earlier of the lane's two events; the other one bounds the window, t_alt)
  const SssEvSlot sl = le.sl;
  const int ex = le.ex, hl = head_lane(head);
  const uint32_t counter0 = g_hot.h.counter, h0 = g_hot.h.rng_has32, u32_0 = g_hot.h.rng_u32;
  const int pos = g_sc.rng_pos;
  const double next_arr = g_hot.h.next_arrival < g_hot.h.J ? g_hot.h.next_arrival_t : __builtin_inf();
  const uint32_t info = sl.info;
  const uint32_t slot = info_slot(info);
  const int s = info_stage(info), j = info_job(info);
  const uint32_t source = g_hot.h.curr_source;
  bool cand = ex < f.E && info_kind(info) == EV_EXECUTOR_READY && slot != INFO_SLOT_NONE;
  {
    const double kq = min_f64(cand ? __builtin_inf() : sl.t, le.t_alt);
    const double t_other = f.E <= 16 ? wave_min_f64_nonneg_row0(kq) : wave_min_f64_nonneg(kq);
    const double t_stop = next_arr < t_other ? next_arr : t_other;
    const uint64_t pre = wave_ballot(cand && sl.t < t_stop);
    if (popc64(pre) < SSS_MIN_ARRIVAL_BATCH || !((pre >> hl) & 1ull)) { STAT(80, 1); return 0; }  // none, too few (lean_arrival), or not the head
  }
  PROF3_ASEC(1);
  SssStage st = {0, 0, 0, 0};
  const SssJob* jpc = f.cjobs + (cand ? slot : 0);
  int gs = 0, n_base = 0, type = AR_START;
  double push_lb = __builtin_inf();
  if (cand) {
    st = f.cstages[slot * f.SP + s];
    gs = jpc->gs_base + s;
    n_base = local_count(jpc->local_mask);
    type = (jpc->frontier_mask & bit64(s)) ? AR_START : AR_PARK;
    // with a source pool set, an executor that stays in it would become committable (ENV:331-338): general path
    cand = st.remaining > 0 && st.moving_to > 0 && (source == POOL_NONE || source != (type == AR_START ? key_stage_pool(j, s) : key_job_pool(j)));
    if (type == AR_START) push_lb = (double)f.eff[(((size_t)gs * 8 + 0) * 3 + 0) * 4 + 3];
  }
  const bool start = type == AR_START;
  if (!((wave_ballot(cand) >> hl) & 1ull)) { STAT(81, 1); return 0; }  // the head of the queue has to be a member
  PROF3_ASEC(2);
  const double key = min_f64(cand ? sl.t + push_lb : sl.t, le.t_alt);
  double M = f.E <= 16 ? wave_min_f64_nonneg_row0(key) : wave_min_f64_nonneg(key);
  if (next_arr < M) M = next_arr;
  bool V = cand && sl.t < M;
  uint64_t vm = wave_ballot(V);
  if (vm == 0) { STAT(82, 1); return 0; }
  PROF3_ASEC(3);
  // who comes before this member, who shares its job / its stage
  uint64_t before = 0, same_job = 0, same_stage = 0;
  for (uint64_t m = vm; m; m &= m - 1) {
    const int k = ctz64(m);
    const double tk = wave_readlane_f64(sl.t, k);
    const uint32_t qk = wave_readlane_u32(sl.seq, k);
    const uint32_t ik = wave_readlane_u32(info, k);
    const bool lt = tk < sl.t || (tk == sl.t && qk < sl.seq);
    before |= lt ? bit64(k) : 0ull;
    same_job |= info_job(ik) == j ? bit64(k) : 0ull;
    same_stage |= ((ik ^ info) >> 8) == 0 ? bit64(k) : 0ull;
  }
  PROF3_ASEC(4);
  // the executor count of the job when this member draws (JOB:81-84: every member before it has been attached)
  const int n_local = n_base + popc64(before & same_job) + 1;
  int li = 0, ri = 0;
  executor_interval(n_local, li, ri);
  const bool open = li != ri;
  int4 da = mk_i4(0, 0, 0, 0), db = da;
  bool drawable = true;
  if (V && start) {
    da = *(const int4*)eff_row(f.eff, gs, li, 0);
    db = open ? *(const int4*)eff_row(f.eff, gs, ri, 0) : da;
    drawable = n_local <= f.E && (da.y & LENW_LEN) > 1 && (db.y & LENW_LEN) > 1;
  }
  const uint32_t nmax = (uint32_t)(64 - pos) >> 1;
  const uint64_t startm0 = wave_ballot(V && start);
  {
    // the stage runs dry before this member (backup scheduling) / a list that draws nothing or fails / randomness
    const uint32_t takes_before = (uint32_t)popc64(before & same_stage & startm0);
    const bool over = V && ((start && ((int)takes_before >= (int)st.remaining || !drawable || (uint32_t)popc64(before & startm0) >= nmax)) ||
                            popc64(before & same_stage) >= (int)st.moving_to);
    const uint64_t om = wave_ballot(over);
    if (om) {
      // everything from the first such member on stays for the one-event path
      const uint32_t rcut = wave_min_u32(over ? (uint32_t)popc64(before & vm) : 0xFFFFFFFFu);
      V = V && (uint32_t)popc64(before & vm) < rcut;
      vm = wave_ballot(V);
      if (vm == 0) { STAT(83, 1); return 0; }
    }
  }
  before &= vm;
  const uint64_t startm = wave_ballot(V && start), openm = wave_ballot(V && start && open);
  const uint32_t n = (uint32_t)popc64(vm), n_x = (uint32_t)popc64(startm);
  const uint32_t rank = (uint32_t)popc64(before), rank_x = (uint32_t)popc64(before & startm), R = (uint32_t)popc64(before & openm);
  const uint32_t cb_take = (uint32_t)popc64(before & same_stage & startm), ct_take = (uint32_t)popc64(vm & same_stage & startm);
  const uint32_t cb_stage = (uint32_t)popc64(before & same_stage), ct_stage = (uint32_t)popc64(vm & same_stage);
  // ---- the starters' draws ----
  const uint32_t Fr = h0 ? rank_x >> 1 : (rank_x + 1) >> 1;
  const bool fresh = ((h0 + rank_x) & 1u) == 0;
  const uint32_t P = R + Fr;
  const bool vx = V && start;
  int4 dd = da;
  uint64_t x32 = 0;
  uint32_t u32 = 0;
  if (vx) {
    if (open) {
      const double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
      const int rand_pt = 1 + (int)(u64_to_unit(g_sc.rng_buf[pos + (int)P]) * (right - left));
      if (!((double)rand_pt <= (double)n_local - left)) dd = db;
    }
    if (fresh) {
      x32 = g_sc.rng_buf[pos + (int)P + (open ? 1 : 0)];
      u32 = (uint32_t)x32;
    } else if (rank_x == 0) {
      u32 = u32_0;
    } else {
      u32 = (uint32_t)(g_sc.rng_buf[pos + (int)R + (int)Fr - 1] >> 32);
    }
  }
  const uint32_t len = (uint32_t)(dd.y & LENW_LEN);
  const uint64_t mm = (uint64_t)u32 * len;
  if (wave_ballot(vx && (uint32_t)mm < len) != 0) { STAT(84, 1); return 0; }
  PROF3_ASEC(5);
  // ---- commit ----
  const uint32_t jkey = key_job_pool(j), skey = key_stage_pool(j, s);
  if (V) {
    SssJob* jp = f.cjobs + slot;
    SssStage* stp = f.cstages + slot * f.SP + s;
    local_atomic_attach(jp, ex);  // JOB:81-84
    g_hot.ex_job[ex] = (int16_t)j;
    lane_atomic_add_u32((uint32_t*)stp + 1, 0u - STG_W1_MOVING_TO);  // moving_to-- (TRK:185-187)
    g_sc.fi_e[rank] = (uint8_t)ex, g_sc.fi_type[rank] = (uint8_t)type;
    g_sc.rl_old[rank] = jkey;
    g_sc.fc_dst[rank] = start ? skey : POOL_NONE;  // the pool it enters after the job's
    if (start) {
      double dur = (double)f.durations[dd.x + (int)(mm >> 32)];
      if (dd.y >> 30) dur += g_c.P.warmup_delay;
      lane_atomic_add_u64((uint64_t*)stp, ((uint64_t)STG_W1_EXECUTING << 32) - 1ull);  // remaining--, executing++ (STG:53-58)
      g_hot.ev[ex].t = sl.t + dur;
      g_hot.ev[ex].seq = counter0 + rank_x;
      g_hot.ev[ex].info = ev_info(EV_TASK_FINISHED, j, s, slot);
      g_hot.ex_task_stage[ex] = (int8_t)s, g_hot.ex_executing[ex] = 1, g_hot.ex_loc[ex] = skey;
      if (cb_take + 1 == ct_take) {  // the stage's last starter of the batch
        f.cdur[slot * f.SP + s] = (float)dur;
        if ((int)st.remaining - (int)ct_take == 0) lane_atomic_add_u32((uint32_t*)&jp->supply, 1u << 16);  // sat_count++ (ENV:595-597)
      }
    } else {
      g_hot.ev[ex].t = __builtin_inf();
      g_hot.ev[ex].info = EV_NONE;
      g_hot.ex_task_stage[ex] = -1, g_hot.ex_loc[ex] = jkey;
    }
    if (rank == n - 1) {
      SssHdr& h = g_hot.h;
      h.wall_time = sl.t;
      h.counter = counter0 + n_x;
      h.n_events += n, h.n_batched += n, h.n_rounds++;
      g_sc.events_this_step += (int32_t)n;
    }
    if (vx && rank_x == n_x - 1) {
      g_sc.rng_pos = pos + (int)P + (open ? 1 : 0) + (fresh ? 1 : 0);
      g_hot.h.rng_has32 = fresh ? 1u : 0u;
      g_hot.h.rng_u32 = fresh ? (uint32_t)(x32 >> 32) : u32;
    }
  }
  wave_sync();
  PROF3_ASEC(6);
  // pools: one lane per pool. The first member of a job speaks for the job's pool, the first starter of a
  // stage for the stage's
  bool def_job = false, def_stage = false;
  // every member arrives at the same stage (executors of one fulfilment; a single member): the job's pool and the stage's are
  // the only two images involved - both through the pair staging, one HBM round trip for the batch (fewer than 64 executors)
  const bool one_stage = f.E >= SSS_PAIR_MIN_E && pair_staging_fits(f.E) && wave_ballot(V && same_stage != vm) == 0;
  if (one_stage) {
    const int l0 = ctz64_nz(vm);
    const uint32_t jk = wave_readlane_u32(jkey, l0), sk = wave_readlane_u32(skey, l0);
    const bool starts = n_x != 0;  // (the members of one stage all start, or all park)
    PROF3_ASEC(7);
    const PoolPairRegs pr = pool_pair_fetch(jk, sk, starts);
    PairImg sj, ss;
    pool_pair_stage(pr, starts, sj, ss);
    PROF3_ASEC(10);
    for (uint32_t q = 0; q < n; q++) {  // rank order (wave-uniform: every lane reads the list)
      const uint32_t e = g_sc.fi_e[q];
      pair_add(sj, e);  // ENV:446: into the job's pool ...
      bool was = pair_remove(sj, e);  // ... and out again (the move to the stage's pool), or - parked - out and back in (TRK:188-222 with old == new)
      CHECK(was);
      if (starts) pair_add(ss, e); else pair_add(sj, e);
    }
    wave_sync();
    PROF3_ASEC(11);
    pool_pair_flush_one(jk, sj);
    if (starts) pool_pair_flush_one(sk, ss);
    PROF3_ASEC(12);
  } else if (V) {
    if ((before & same_job) == 0) def_job = !pool_pass_many(jkey, n);
    if (start && cb_take == 0) def_stage = !pool_enter_many(skey, n);
  }
  if (n != n_x && lane == 0) {
    // a parked executor's event is gone: one reference to the job's cache slot fewer (a starter's new event names it again)
    for (uint32_t q = 0; q < n; q++)
      if (g_sc.fi_type[q] == AR_PARK) lds_slot_ref()[lds_slot_of()[key_job(g_sc.rl_old[q])]]--;
  }
  uint64_t dj = wave_ballot(def_job), ds = wave_ballot(def_stage);
  STAT(34, 1), STAT(35, n), STAT(36, popc64(dj)), STAT(37, popc64(ds)), STAT(38, n - n_x);
  wave_sync();
  PROF3_ASEC(7);
  pools_staged<STAGED_PASS>(dj, n, V ? jkey : POOL_NONE, !start);  // tables with more than 8 slots, or about to grow
  PROF3_ASEC(8);
  pools_staged<STAGED_ENTER>(ds, n, (V && start) ? skey : POOL_NONE, false);
  PROF3_ASEC(9);
  // saturation bit of the stage (ENV:566-582), by its last member: arrivals that start a task leave the
  // demand what it was, parked ones raise it
  if (V && cb_stage + 1 == ct_stage) {
    const SssStage t2 = f.cstages[slot * f.SP + s];
    SssJob* jp = f.cjobs + slot;
    if ((int)t2.remaining - ((int)t2.moving_to + (int)t2.commit_to) <= 0)
      lane_atomic_or_u64(&jp->sat_mask, bit64(s));
    else
      lane_atomic_and_u64(&jp->sat_mask, ~bit64(s));
  }
  wave_sync();
  return (int)n;
}

// ------------------------------------------------------------------------------------------
// ONE released executor (all lanes, wave-uniform control flow): TASK_FINISHED on a stage with no task left to start,
// not the stage's last running task, the stage's pool holding a commitment - what batch_released_events does for
// several such events at once, for the single one that heads the queue (most of them are alone: the batch declines,
// and the lane-0 handlers - handle_task_completion -> fulfill_commitment -> move_executor_to_stage ->
// trk_move_executor_to_pool -> execute_next_task - took ~14 k ticks per event in the slowest envs of a config-3 launch,
// nearly half of it the dependent HBM round trips of the two pool images). Here every lane reads the same state and
// takes the same decisions; the commitment is found with one ballot over the list (one entry per lane); both pool
// images come in with one round trip (pool_pair_*), the duration descriptor rides along, the draw is computed from
// the buffered raw outputs before anything is modified (a draw that needs Lemire's rejection loop goes the general way),
// and lane 0 writes the scalars. By destination of the commitment (ENV:639-660, 699-712, 784-819):
//   START  a stage of the same job that is in the frontier: into its pool, a task starts (ENV:584-615, TPCH:75-106);
//   PARK   ... not in the frontier yet: into the job's pool (ENV:808-813);
//   SEND   a stage of another job (whose records are cached): detached, EXECUTOR_READY after moving_delay (ENV:617-637);
//   IDLE   the common pool: into the job's pool, or - the job being saturated - detached into the common pool (ENV:745-782).
// Left to the general handlers: no commitment (the executor becomes the source), the event that completes its stage, a
// destination stage out of tasks (backup scheduling), an executor that would enter the current source (it becomes
// committable), 64 executors (two 512-byte tables do not fit the staging areas), lists that draw nothing.
// Returns 1 = the event is consumed, 0 = nothing was modified.
// ------------------------------------------------------------------------------------------
SSS_DEV int lean_released(const FastCtx& f, int ex, double t_ev, uint32_t info) {
  UTRACE("lean_released");
#ifdef SSS_NO_BATCH
  return 0;
#endif
  PROF3(32);
  const int lane = wave_lane();
  const uint32_t slot = info_slot(info);
  const int s = info_stage(info), j = info_job(info);
  // Every lane reads the same words and takes the same decisions. A "no" sends lane 0 into the general handler, which
  // rewrites the state the lanes look at: so every decision is taken by a ballot - all lanes have evaluated it, on the
  // same state, before any lane acts on it (shared state is read before the collective that guards its use).
  // ---- reads ----
  const SssStage st_old = f.cstages[slot * f.SP + s];
  const uint32_t source = g_hot.h.curr_source;
  const int n_commits = g_hot.h.n_commits;
  const int A = g_hot.h.n_active;
  const uint32_t counter0 = g_hot.h.counter;
  uint32_t h0 = g_hot.h.rng_has32, u32_0 = g_hot.h.rng_u32;
  const uint32_t sp = key_stage_pool(j, s);
  const bool mine = st_old.remaining == 0 && st_old.executing >= 2 && g_hot.ex_job[ex] == j && pair_staging_fits(f.E);  // (64 executors: two 512-byte tables)
  // the commitment its pool serves first (TRK:178-183: the first-inserted entry of that source)
  const CommitHit hit = commit_first_wave(sp, false, mine, n_commits);
  STAT(120, 1), STAT(121, hit.ci < 0);
  if (hit.ci < 0) return 0;
  const int ci = hit.ci;
  const uint32_t dst = hit.dst;
  const int c_left = hit.num - 1;
  SssJob* const jp = f.cjobs + slot;
  const bool job_sat = (int)jp->sat_count == (int)jp->n_stages;  // JOB:53-55
  const int j2 = key_job(dst), s2 = key_stage(dst);  // the stage the commitment names (-1, -1: the common pool)
  bool ok = dst != sp && (dst == POOL_COMMON || s2 >= 0);
  // where the executor goes: the committed stage - or, that stage having no task left, a backup stage (ENV:784-797, 821-845)
  int tj = j2, ts = s2;
  bool backup = false;
  if (ok && dst != POOL_COMMON) {
    const SssStage st_c = j2 == j ? f.cstages[slot * f.SP + s2] : *stgp(j2, s2);
    backup = st_c.remaining <= 0;
  }
  if (backup) {  // (wave-uniform) _find_backup_stage: the executor's own job first, then the others in arrival order
    tj = -1, ts = -1;
    const int srcj = j <= 0 ? trk_source_job_id() : j;  // `if not source_job_id` (ENV:521): job id 0 is falsy
    uint64_t m_own = 0;
    if (j == srcj || (int)jp->supply < f.E) m_own = ready_mask_of_job(*jp, true);
    if (m_own)
      tj = j, ts = ctz64_nz(m_own);
    else {
      const int n_others = A - (jp->active_mask != 0 ? 1 : 0);  // an empty list of others means "all active jobs" (ENV:518-519)
      for (int a0 = 0; a0 < A && tj < 0; a0 += 64) {
        const int a = a0 + lane;
        int jj = -1, ss = -1;
        if (a < A) {
          jj = lds_active()[a];
          if (!(n_others > 0 && jj == j)) {
            const SssJob* q = jobp(jj);
            // (the commitment is settled before the search, TRK:159-176: the committed stage's job counts one executor fewer)
            if (jj == srcj || (int)q->supply - ((jj == j2 && j2 != j) ? 1 : 0) < f.E) {
              const uint64_t m = ready_mask_of_job(*q, true);
              if (m) ss = ctz64_nz(m);
            }
          }
        }
        const uint64_t hm = wave_ballot(ss >= 0);
        if (hm) tj = (int)wave_readlane_u32((uint32_t)jj, ctz64_nz(hm)), ts = (int)wave_readlane_u32((uint32_t)ss, ctz64_nz(hm));
      }
    }
  }
  // what becomes of it
  int type;
  uint32_t tslot = slot;
  SssStage st_t = {0, 0, 0, 0};
  if (tj < 0)  // the common pool was committed to, or no backup stage: ENV:745-782 with a list of one
    type = job_sat ? RL_IDLE_COMMON : RL_IDLE_JOB;
  else {
    if (tj != j) tslot = f.slot_of[tj];  // another job: its records have to be cached
    if (tslot == SLOT_NONE) {
      ok = false, type = RL_SEND;
    } else {
      st_t = f.cstages[tslot * f.SP + ts];
      ok = ok && st_t.remaining > 0;
      type = tj != j ? RL_SEND : ((jp->frontier_mask & bit64(ts)) ? RL_START : RL_PARK);
    }
  }
  const uint32_t enters = type == RL_START ? key_stage_pool(j, ts) : (type == RL_SEND ? POOL_NONE : (type == RL_IDLE_COMMON ? POOL_COMMON : key_job_pool(j)));
  // an executor that enters the source would become committable (ENV:331-338, TRK:107-113): general path
  ok = ok && (source == POOL_NONE || enters != source);
  // everything that comes from HBM is asked for here, in one go: both pool images now (whether or not the event will go this
  // way: loads are harmless), the duration descriptors below - their round trips overlap
  const PoolPairRegs pr = pool_pair_fetch(sp, enters, enters != POOL_NONE);
  const bool start = ok && type == RL_START;
  int n_local = 0, li = 0, ri = 0;
  int4 da = mk_i4(0, 0, 0, 0), db = da;
  if (start) {  // TPCH:75-106: the executor's last task was on another stage of the job ("first_wave" mode)
    n_local = local_count(jp->local_mask);
    ok = n_local > 0 && n_local <= f.E && ts != s;
    if (ok) {
      executor_interval(n_local, li, ri);
      const int gs2 = jp->gs_base + ts;
      da = *(const int4*)eff_row(f.eff, gs2, li, 2);
      db = li != ri ? *(const int4*)eff_row(f.eff, gs2, ri, 2) : da;
      ok = (da.y & LENW_LEN) > 1 && (db.y & LENW_LEN) > 1;  // lists that draw nothing / fail: one at a time
    }
  }
  const bool refill = start && ok && g_sc.rng_pos > 62;  // (a draw takes up to two raw outputs)
  STAT(122, !ok);
  if (wave_ballot(!ok) != 0) return 0;
  if (refill) rng_refill();
  // the draw, from the buffered raw outputs (TPCH:216-235, numpy's buffered 32-bit Lemire path)
  int pos = g_sc.rng_pos;
  double dur = 0.0;
  bool reject = false;
  if (start) {
    int4 dd = da;
    if (li != ri) {
      const double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
      const int rand_pt = 1 + (int)(u64_to_unit(g_sc.rng_buf[pos]) * (right - left));
      pos += 1;
      if (!((double)rand_pt <= (double)n_local - left)) dd = db;
    }
    uint32_t u32;
    if (h0)
      u32 = u32_0, h0 = 0;
    else {
      const uint64_t x = g_sc.rng_buf[pos];
      pos += 1;
      u32 = (uint32_t)x, u32_0 = (uint32_t)(x >> 32), h0 = 1;
    }
    const uint32_t len = (uint32_t)(dd.y & LENW_LEN);
    const uint64_t mm = (uint64_t)u32 * len;
    reject = (uint32_t)mm < len;  // Lemire's rejection test would loop: one at a time
    dur = (double)f.durations[dd.x + (int)(mm >> 32)];
    if (dd.y >> 30) dur += g_c.P.warmup_delay;
  }
  if (wave_ballot(reject) != 0) return 0;
  STAT(111, 1), STAT(112 + type, 1), STAT(117, backup);
  // ---- nothing has been modified up to here; from here on the event is consumed ----
  PairImg so, sn;
  pool_pair_stage(pr, enters != POOL_NONE, so, sn);
  {
    bool was = pair_remove(so, (uint32_t)ex);  // TRK:188-222
    CHECK(was);
    so.s.aux -= 1;  // the pool's outgoing commitments (TRK:159-176)
    if (enters != POOL_NONE) pair_add(sn, (uint32_t)ex);
    wave_sync();
    pool_pair_flush_one(sp, so);
    if (enters != POOL_NONE) pool_pair_flush_one(enters, sn);
  }
  if (lane == 0) {
    SssHdr& h = g_hot.h;
    h.wall_time = t_ev;
    h.n_events += 1, h.n_batched += 1, h.n_rounds += 1;
    g_sc.events_this_step += 1;
    // the stage it leaves (STG:60-62)
    f.cstages[slot * f.SP + s].executing = (int16_t)(st_old.executing - 1);
    // the commitment (TRK:159-176): dict.pop when it is used up - swap-remove, the order lives in c_seq
    g_hot.c_n[ci] = (int16_t)c_left;
    if (c_left == 0) {  // (entry `last` may be this one: the dead entry's bytes are what trk_remove_commitment leaves)
      const int last = h.n_commits - 1;
      g_hot.c_src[ci] = g_hot.c_src[last], g_hot.c_dst[ci] = g_hot.c_dst[last], g_hot.c_n[ci] = g_hot.c_n[last], g_hot.c_seq[ci] = g_hot.c_seq[last];
      h.n_commits = last;
    }
    if (dst == POOL_COMMON) {
      h.supply_none -= 1;  // a commitment to the common pool counted as its supply (TRK:146-154, 159-176)
      CHECK(h.supply_none >= 0);
    } else {  // the committed stage: one commitment fewer; another job's executor count: one fewer (TRK:159-176)
      const JobView v = jobview(j2);
      const int c = (int)v.st[s2].commit_to - 1;
      CHECK(c >= 0);
      v.st[s2].commit_to = (uint8_t)c;
      update_sat(v, s2);
      if (j2 != j) v.job->supply = (int16_t)(v.job->supply - 1);
    }
    uint32_t new_info = EV_NONE;
    double new_t = __builtin_inf();
    if (tj >= 0) {  // the stage it goes to
      SssStage* spt = f.cstages + tslot * f.SP + ts;
      SssJob* jpt = f.cjobs + tslot;
      SssStage t2 = *spt;
      if (type == RL_START) {
        t2.remaining = t2.remaining - 1, t2.executing = (int16_t)(t2.executing + 1);  // STG:53-58
        if (t2.remaining == 0) jpt->sat_count = (int16_t)(jpt->sat_count + 1);  // ENV:595-597
        f.cdur[slot * f.SP + ts] = (float)dur;  // ENV:604
        new_t = t_ev + dur, new_info = ev_info(EV_TASK_FINISHED, j, ts, slot);
        g_sc.rng_pos = pos, h.rng_has32 = h0, h.rng_u32 = u32_0;
      } else if (type == RL_SEND) {
        t2.moving_to = (uint8_t)(t2.moving_to + 1);        // TRK:206-216
        jpt->supply = (int16_t)(jpt->supply + 1);          // the new job's executor count ...
        jp->supply = (int16_t)(jp->supply - 1);            // ... and the old one's (TRK:218-221)
        new_t = t_ev + g_c.P.moving_delay, new_info = ev_info(EV_EXECUTOR_READY, tj, ts, tslot);
      }
      *spt = t2;
      const int demand = (int)t2.remaining - ((int)t2.moving_to + (int)t2.commit_to);  // ENV:566-582
      const uint64_t m = jpt->sat_mask;
      jpt->sat_mask = demand <= 0 ? (m | bit64(ts)) : (m & ~bit64(ts));
    }
    // the executor
    g_hot.ex_executing[ex] = type == RL_START ? 1 : 0;
    g_hot.ex_loc[ex] = enters;
    if (type == RL_START) g_hot.ex_task_stage[ex] = (int8_t)ts;
    if (type == RL_PARK) g_hot.ex_task_stage[ex] = -1;  // ENV:808-813
    if (type == RL_SEND || type == RL_IDLE_COMMON) {      // JOB:86-89
      jp->local_mask = local_without(jp->local_mask, ex);
      g_hot.ex_job[ex] = -1, g_hot.ex_task_stage[ex] = -1;
    }
    // its event slot and the cache-slot references of the events' jobs
    SssEvSlot sl;
    sl.t = new_t, sl.seq = counter0, sl.info = new_info;
    if (new_info == EV_NONE) sl.seq = g_hot.ev[ex].seq;
    g_hot.ev[ex] = sl;
    if (new_info != EV_NONE) h.counter = counter0 + 1;
    if (type != RL_START) {
      lds_slot_ref()[slot]--;
      if (type == RL_SEND) lds_slot_ref()[tslot]++;
    }
  }
  wave_sync();
  return 1;
}

// ------------------------------------------------------------------------------------------
// ONE arriving executor (all lanes, wave-uniform control flow): EXECUTOR_READY (ENV:440-450) for a job whose records
// are cached - the single-member case of batch_arrival_events without the batch machinery (window, ranking, per-pool
// leaders), in the style of lean_released: decisions by ballot on state every lane reads alike, both pool images (the
// job's, which the executor passes through, and the stage's) with one round trip, the idle executor's duration draw
// (TPCH:88-94: fresh durations, else first wave + warmup_delay) from the buffered raw outputs before anything is modified.
//   START  the stage is in the frontier and has a task left: into the stage's pool, a task starts (ENV:584-615);
//   PARK   not in the frontier yet: it waits in the job's pool (ENV:808-813).
// Left to the general handler: a stage out of tasks (backup scheduling, ENV:784-797), an executor that would stay in the
// current source (it becomes committable), 64 executors, lists that draw nothing, a draw that needs Lemire's loop.
// Returns 1 = the event is consumed, 0 = nothing was modified.
// ------------------------------------------------------------------------------------------
SSS_DEV int lean_arrival(const FastCtx& f, int ex, double t_ev, uint32_t info) {
  UTRACE("lean_arrival");
#ifdef SSS_NO_BATCH
  return 0;
#endif
  PROF3(38);
  const int lane = wave_lane();
  const uint32_t slot = info_slot(info);
  const int s = info_stage(info), j = info_job(info);
  // ---- reads (every lane the same words) ----
  SssJob* const jp = f.cjobs + slot;
  SssStage* const stp = f.cstages + slot * f.SP + s;
  const SssStage st = *stp;
  const uint32_t source = g_hot.h.curr_source;
  const uint32_t counter0 = g_hot.h.counter;
  uint32_t h0 = g_hot.h.rng_has32, u32_0 = g_hot.h.rng_u32;
  const uint64_t local = jp->local_mask;
  const bool start = (jp->frontier_mask & bit64(s)) != 0;
  const uint32_t jkey = key_job_pool(j), skey = key_stage_pool(j, s);
  const PoolPairRegs pr = pool_pair_fetch(jkey, skey, start);  // (asked for right away: its round trip overlaps the descriptors')
  // with a source pool set, an executor that stays in it would become committable (ENV:331-338): general path
  bool ok = pair_staging_fits(f.E) && st.remaining > 0 && st.moving_to > 0 && g_hot.ex_task_stage[ex] < 0 && (source == POOL_NONE || source != (start ? skey : jkey));
  const int n_local = local_count(local) + 1;  // JOB:81-84: the executor is attached before it draws
  int li = 0, ri = 0;
  int4 da = mk_i4(0, 0, 0, 0), db = da;
  if (ok && start) {
    ok = n_local <= f.E;
    if (ok) {
      executor_interval(n_local, li, ri);
      const int gs = jp->gs_base + s;
      da = *(const int4*)eff_row(f.eff, gs, li, 0);
      db = li != ri ? *(const int4*)eff_row(f.eff, gs, ri, 0) : da;
      ok = (da.y & LENW_LEN) > 1 && (db.y & LENW_LEN) > 1;
    }
  }
  const bool refill = ok && start && g_sc.rng_pos > 62;
  if (wave_ballot(!ok) != 0) return 0;
  if (refill) rng_refill();
  int pos = g_sc.rng_pos;
  double dur = 0.0;
  bool reject = false;
  if (start) {
    int4 dd = da;
    if (li != ri) {
      const double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
      const int rand_pt = 1 + (int)(u64_to_unit(g_sc.rng_buf[pos]) * (right - left));
      pos += 1;
      if (!((double)rand_pt <= (double)n_local - left)) dd = db;
    }
    uint32_t u32;
    if (h0)
      u32 = u32_0, h0 = 0;
    else {
      const uint64_t x = g_sc.rng_buf[pos];
      pos += 1;
      u32 = (uint32_t)x, u32_0 = (uint32_t)(x >> 32), h0 = 1;
    }
    const uint32_t len = (uint32_t)(dd.y & LENW_LEN);
    const uint64_t mm = (uint64_t)u32 * len;
    reject = (uint32_t)mm < len;
    dur = (double)f.durations[dd.x + (int)(mm >> 32)];
    if (dd.y >> 30) dur += g_c.P.warmup_delay;
  }
  if (wave_ballot(reject) != 0) return 0;
  STAT(125, 1), STAT(126, start);
  // ---- from here on the event is consumed ----
  PairImg sj, ss;
  pool_pair_stage(pr, start, sj, ss);
  {
    pair_add(sj, (uint32_t)ex);  // ENV:446: into the job's pool ...
    bool was = pair_remove(sj, (uint32_t)ex);  // ... and out again, or - parked - out and back in (TRK:188-222 with old == new)
    CHECK(was);
    if (start) pair_add(ss, (uint32_t)ex); else pair_add(sj, (uint32_t)ex);
    wave_sync();
    pool_pair_flush_one(jkey, sj);
    if (start) pool_pair_flush_one(skey, ss);
  }
  if (lane == 0) {
    SssHdr& h = g_hot.h;
    h.wall_time = t_ev;
    h.n_events += 1, h.n_batched += 1, h.n_rounds += 1;
    g_sc.events_this_step += 1;
    jp->local_mask = local_with(local, ex);  // JOB:81-84
    g_hot.ex_job[ex] = (int16_t)j;
    SssStage t2 = st;
    t2.moving_to = (uint8_t)(t2.moving_to - 1);  // TRK:185-187
    SssEvSlot sl = g_hot.ev[ex];
    if (start) {
      t2.remaining = t2.remaining - 1, t2.executing = (int16_t)(t2.executing + 1);  // STG:53-58
      if (t2.remaining == 0) jp->sat_count = (int16_t)(jp->sat_count + 1);                      // ENV:595-597
      f.cdur[slot * f.SP + s] = (float)dur;                                                     // ENV:604
      sl.t = t_ev + dur, sl.seq = counter0, sl.info = ev_info(EV_TASK_FINISHED, j, s, slot);
      h.counter = counter0 + 1;
      g_sc.rng_pos = pos, h.rng_has32 = h0, h.rng_u32 = u32_0;
      g_hot.ex_task_stage[ex] = (int8_t)s, g_hot.ex_executing[ex] = 1, g_hot.ex_loc[ex] = skey;
    } else {
      sl.t = __builtin_inf(), sl.info = EV_NONE;
      g_hot.ex_task_stage[ex] = -1, g_hot.ex_loc[ex] = jkey;  // ENV:808-813
      lds_slot_ref()[slot]--;  // its event is gone (a starter's new event names the job's slot again)
    }
    g_hot.ev[ex] = sl;
    *stp = t2;
    const int demand = (int)t2.remaining - ((int)t2.moving_to + (int)t2.commit_to);  // ENV:566-582
    const uint64_t m = jp->sat_mask;
    jp->sat_mask = demand <= 0 ? (m | bit64(s)) : (m & ~bit64(s));
  }
  wave_sync();
  return 1;
}

// ------------------------------------------------------------------------------------------
// The executors a completing job leaves behind (all lanes), ahead of the event that completes it: when the last running
// task of a job's last active stage finishes, _process_job_completion (ENV:682-697) flushes the idle executors parked in
// the job's pool into the common pool (ENV:745-782: list(set) order of the idle set, TRK:188-222 + JOB:86-89 each) - on lane 0
// that is ~8 k ticks per executor (the dependent HBM round trips of trk_move_executor_to_pool), ~60 k per completed job
// at BASELINE config 3. Nothing between the event's pop and that flush touches the two pools or those executors, so
// the flush is done here, with the whole wave, before the lane-0 handler runs: lane 0 builds the list (the same
// get_idle_source_executors image), both pool images come in with one round trip (pool_pair_*), n removals and n additions
// in list order; the handler then finds the job's pool empty and skips its own loop. Only called for a TASK_FINISHED event
// of a cached job; does nothing unless that event completes the job.
// ------------------------------------------------------------------------------------------
SSS_DEV void preflush_completing_job(const FastCtx& f, uint32_t info) {
#ifdef SSS_NO_BATCH
  return;
#endif
  const int lane = wave_lane();
  const uint32_t slot = info_slot(info);
  const int s = info_stage(info), j = info_job(info);
  const SssStage st = f.cstages[slot * f.SP + s];
  const SssJob* jp = f.cjobs + slot;
  const bool completes = st.remaining == 0 && st.executing == 1 && jp->active_mask == bit64(s) && (int)jp->sat_count == (int)jp->n_stages && pair_staging_fits(f.E);
  if (wave_ballot(completes) == 0) return;
  PROF3(40);
  const uint32_t jkey = key_job_pool(j);
  if (lane == 0) {
    int m = 0;
    if (pool_size(jkey) > 0) {
      SetImg<uint8_t> idle = get_idle_source_executors(jkey);
      for (uint32_t i = 0; i <= idle.mask; i++)  // list(set): ascending slot order
        if (idle.tab[i] >= 2) g_sc.fi_e[m++] = (uint8_t)(idle.tab[i] - 2);
    }
    g_sc.fi_m = m;
  }
  wave_sync();
  const int m = g_sc.fi_m;
  if (m == 0) return;
  const PoolPairRegs pr = pool_pair_fetch(jkey, POOL_COMMON, true);
  PairImg so, sn;
  pool_pair_stage(pr, true, so, sn);
  LocalGroup moved = local_group();
  pair_remove_many(so, g_sc.fi_e, 0, m);
  for (int i = 0; i < m; i++) {
    const uint32_t e = g_sc.fi_e[i];
    local_group_add(moved, (int)e);
    pair_add(sn, e);
  }
  wave_sync();
  pool_pair_flush_one(jkey, so);
  pool_pair_flush_one(POOL_COMMON, sn);
  if (lane == 0) {
    SssJob* jw = f.cjobs + slot;
    local_group_detach(jw, moved);  // JOB:86-89
    for (int i = 0; i < m; i++) {
      const int e = g_sc.fi_e[i];
      g_hot.ex_loc[e] = POOL_COMMON, g_hot.ex_job[e] = -1, g_hot.ex_task_stage[e] = -1;
    }
  }
  STAT(118, 1), STAT(119, m);
  wave_sync();
}

// _find_schedulable_stages() over all active jobs (ENV:505-540): one lane per stage of a job,
// ballot gives the job's ready mask; sat_mask makes the parent test a mask operation.
// Returns len(schedulable_stages); lane 0 stores the per-job masks.
// n_active / source job come from the mailbox lane 0 filled before the preceding wave_sync
// (publish_scan_inputs): lane 0 may already be past this function when another lane reads them.
SSS_DEV int find_schedulable_all() {
  PROF3(21);
  int lane = wave_lane();
  int A = g_sc.m_n_active;
  int src_job = g_sc.m_src_job;
  uint32_t total = 0;
  // one lane per active job; readiness of a stage is a mask test against the job's saturated mask
  for (int a0 = 0; a0 < A; a0 += 64) {
    int a = a0 + lane;
    uint32_t cnt = 0;
    if (a < A) {
      int j = lds_active()[a];
      SssJob* job = jobp(j);
      uint64_t m = 0;
      if (j == src_job || (int)job->supply < g_c.E) m = ready_mask_of_job(*job, false);
      job->sched_mask = m;
      cnt = (uint32_t)popc64(m);
    }
    total += wave_sum_u32(cnt);
  }
  return (int)total;
}

// _observe (ENV:345-406) + utils.subgraph (utils.py:5-22) into the env's padded output rows
SSS_DEV void write_observation(const SssLayout& L, const SssBuffers& B, int env, double reward) {
  PROF3(22);
  int lane = wave_lane();
  uint64_t t_obs0 = wave_clock();
  const SssHdr& h = g_hot.h;
  // the output rows alias nothing that is read here: the loads of later iterations may pass earlier stores
  float* __restrict__ nodes = B.nodes + (size_t)env * L.n_cap * 3;
  int32_t* __restrict__ el = B.edge_links + (size_t)env * L.ed_cap * 2;
  int32_t* __restrict__ dag_ptr = B.dag_ptr + (size_t)env * (L.J_cap + 1);
  int32_t* __restrict__ sup = B.exec_supplies + (size_t)env * L.J_cap;
  int A = h.n_active;
  uint32_t srck = h.curr_source;
  int src_job = (srck == POOL_NONE || srck == POOL_COMMON) ? -1 : key_job(srck);
  int src_idx = A;  // ENV:352
  uint64_t lt = bit64(lane) - 1;
  uint16_t* nbase = lds_keys();  // first node row of each active job (scratch shared with the set code)
  // pass 1 - lanes over jobs: dag_ptr (exclusive scan of active-stage counts), exec_supplies
  uint32_t run = 0;
  for (int a0 = 0; a0 < A; a0 += 64) {
    int a = a0 + lane;
    uint32_t cnt = 0;
    int j = -1, supply = 0;
    if (a < A) {
      j = lds_active()[a];
      const SssJob* job = jobp(j);
      cnt = (uint32_t)popc64(job->active_mask);
      supply = job->supply;
    }
    uint32_t excl = wave_scan_excl_u32(cnt);
    uint32_t tot = wave_sum_u32(cnt);
    uint64_t is_src = wave_ballot(a < A && j == src_job);
    if (is_src) src_idx = a0 + ctz64(is_src);
    if (a < A) {
      nbase[a] = (uint16_t)(run + excl);
      dag_ptr[a] = (int32_t)(run + excl);
      sup[a] = supply;
    }
    run += tot;
  }
  int base_n = (int)run;
  wave_sync();
  // pass 2 - lanes over (job, stage): node rows
  int SPn = g_c.SP;
  // four rows per lane at a time, every load of the four issued before the first store (one round trip to HBM per
  // 256 rows instead of one per 64)
  for (int i0 = lane; i0 < A * SPn; i0 += 64 * 4) {
    int32_t remaining[4];
    float recent[4];
    uint64_t act[4], sched[4];
    int nst[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u;
      remaining[u] = 0, recent[u] = 0.0f, act[u] = 0, sched[u] = 0, nst[u] = 0;
      if (i < A * SPn) {
        const int a = i / SPn, st = i - a * SPn;
        const JobView v = jobview(lds_active()[a]);  // one look-up of the job's slot for the three records
        // the stage's counters and duration are fetched along with the job's record, not after it (their
        // addresses do not depend on it; rows of inactive stages are read and dropped)
        remaining[u] = v.st[st].remaining;
        recent[u] = v.dur[st];
        act[u] = v.job->active_mask, sched[u] = v.job->sched_mask, nst[u] = (int)v.job->n_stages;
      }
    }
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u;
      if (i < A * SPn) {
        const int a = i / SPn, st = i - a * SPn;
        if (st < nst[u] && (act[u] & bit64(st))) {
          const int row = (int)nbase[a] + popc64(act[u] & (bit64(st) - 1));
          // plain stores: non-temporal ones were measured to double the HBM write traffic (partial
          // lines are no longer combined in L2) for no gain in time
          nodes[row * 3 + 0] = (float)remaining[u];
          nodes[row * 3 + 1] = recent[u];
          nodes[row * 3 + 2] = (sched[u] & bit64(st)) ? 1.0f : 0.0f;
        }
      }
    }
  }
  // pass 3 - lanes over (job, template edge): active subgraph, compacted in (job, edge) order. The rows are a
  // function of the active jobs, their order and their active-stage masks alone: when none of that has changed
  // since they were last written to this buffer, they are there already.
  const bool same_graph = h.obs_graph_version == h.graph_version && h.obs_bind_gen == B.gen;
  int ME = g_c.P.max_edges;
  int base_e = same_graph ? h.obs_n_edges : 0;
  // four groups of 64 (job, edge) pairs at a time: the four job records, then the four edges, are fetched together;
  // the compaction below stays in (job, edge) order
  for (int i0 = 0; i0 < (same_graph ? 0 : A * ME); i0 += 64 * 4) {
    uint64_t act[4];
    int eoff[4], nb[4];
    bool has[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u + lane;
      has[u] = false, act[u] = 0, eoff[u] = 0, nb[u] = 0;
      if (i < A * ME) {
        const int a = i / ME, e = i - a * ME;
        const SssJob* job = jobp(lds_active()[a]);
        has[u] = e < (int)job->n_edges;
        act[u] = job->active_mask, eoff[u] = job->edge_off + e, nb[u] = (int)nbase[a];
      }
    }
    int uu[4], vv[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      uu[u] = 0, vv[u] = 0;
      if (has[u]) uu[u] = g_c.pk.edges[2 * eoff[u]], vv[u] = g_c.pk.edges[2 * eoff[u] + 1];
    }
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      if (i0 + 64 * u >= A * ME) break;
      const bool keep = has[u] && (act[u] & bit64(uu[u])) && (act[u] & bit64(vv[u]));
      const uint64_t bal = wave_ballot(keep);
      if (keep) {
        const int pos = base_e + popc64(bal & lt);
        el[2 * pos + 0] = nb[u] + popc64(act[u] & (bit64(uu[u]) - 1));
        el[2 * pos + 1] = nb[u] + popc64(act[u] & (bit64(vv[u]) - 1));
      }
      base_e += popc64(bal);
    }
  }
  if (lane == 0) {
    dag_ptr[A] = base_n;
    int32_t* oi = B.obs_i32 + (size_t)env * SSS_OBS_I32;
    double* of = B.obs_f64 + (size_t)env * SSS_OBS_F64;
    int ncommit = 0;
    if (srck != POOL_NONE) {
      int p = pool_index(srck);
      ncommit = (int)g_c.pool_hdr[p].used - (int)g_c.pool_hdr[p].commit_from;
    }
    oi[OBS_N_NODES] = base_n, oi[OBS_N_EDGES] = base_e, oi[OBS_N_JOBS] = A, oi[OBS_N_SCHED] = h.n_sched;
    oi[OBS_NUM_COMMITTABLE] = ncommit, oi[OBS_SOURCE_JOB_IDX] = src_idx;
    oi[OBS_TERMINATED] = h.terminated, oi[OBS_ERR] = h.err;
    of[OBS_REWARD] = reward, of[OBS_WALL_TIME] = h.wall_time;
    g_hot.h.prof[4] += wave_clock() - t_obs0;
    g_hot.h.obs_n_nodes = base_n;
    g_hot.h.obs_graph_version = h.graph_version, g_hot.h.obs_n_edges = base_e, g_hot.h.obs_bind_gen = B.gen;
    g_hot.h.obs_n_sched = h.n_sched;
    g_hot.h.last_reward = reward;
    // SURVEY 8(d) algorithmic bytes of this step: k*140 + 12N + (12N + 4(A+1) + 4A + 8Ed + 12) + 26
    g_hot.h.model_bytes += (uint64_t)g_sc.events_this_step * 140u + 24u * (uint64_t)base_n + 4u * (uint64_t)(A + 1) +
                            4u * (uint64_t)A + 8u * (uint64_t)base_e + 12u + 26u;
  }
}

// ---- staging at launch boundaries (all lanes) ----
// HBM -> LDS: the hot block verbatim, the active-job list, and the records + stage counters of the
// first n_slots active jobs into the cache. LDS -> HBM at the end of the launch.
SSS_DEV void env_begin(const uint8_t* base) {
  PROF3(23);
  int lane = wave_lane();
  {
    // the header, and of the per-executor arrays the entries of this env's executors (commitments: at most one
    // entry per executor). The rest of the HBM image is never read or written.
    const SssHot* g = (const SssHot*)base;
    if (lane < (int)(sizeof(SssHdr) / 16)) ((uint4*)&g_hot.h)[lane] = ((const uint4*)&g->h)[lane];
    for (int x = lane; x < SSS_MAX_EXEC; x += 64) {  // (one entry per lane; two in the wide instantiation)
      SssEvSlot ev;
      ev.t = __builtin_inf(), ev.seq = 0, ev.info = EV_NONE;  // the queue's reductions run over all slots
      uint32_t loc = POOL_NONE, csrc = 0, cdst = 0, cseq = 0;
      int16_t job = -1, cn = 0;
      int8_t ts = -1;
      uint8_t exe = 0;
      if (x < g_c.E)
        ev = g->ev[x], loc = g->ex_loc[x], job = g->ex_job[x], ts = g->ex_task_stage[x], exe = g->ex_executing[x], csrc = g->c_src[x],
        cdst = g->c_dst[x], cseq = g->c_seq[x], cn = g->c_n[x];
      g_hot.ev[x] = ev, g_hot.ex_loc[x] = loc, g_hot.ex_job[x] = job, g_hot.ex_task_stage[x] = ts, g_hot.ex_executing[x] = exe;
      g_hot.c_src[x] = csrc, g_hot.c_dst[x] = cdst, g_hot.c_seq[x] = cseq, g_hot.c_n[x] = cn;
    }
  }
  for (int i = lane; i < g_c.J_cap; i += 64) lds_slot_of()[i] = SLOT_NONE;
  for (int x = lane; x < g_c.E; x += 64) lds_exdesc()[x].gs = -1;
  lds_slot_ref()[lane] = 0;
  wave_sync();
  int A = g_hot.h.n_active;
  for (int i = lane; i < A; i += 64) lds_active()[i] = g_c.active_g[i];
  {
    // The jobs of the pending events get the cache slots - the jobs with the most pending events first (ties: lowest
    // executor), so that a burst of executors travelling to one job, or many executors working on one job, never
    // finds its job without a slot because single events of other jobs were met first. One lane per executor (two
    // executors per lane in the wide instantiation: x = lane + 64 h): same[job] counts and the "first executor of its
    // job" flag come from a readlane sweep over the executors, the rank of a job among the jobs from a second sweep.
    // The events learn the slot their job got for this launch.
    uint32_t info[SSS_EPL], cnt[SSS_EPL], key[SSS_EPL], rank[SSS_EPL];
    bool has[SSS_EPL], first[SSS_EPL];
    int j[SSS_EPL];
    uint64_t m[SSS_EPL];
    for (int h = 0; h < SSS_EPL; h++) {
      const int x = lane + 64 * h;
      info[h] = g_hot.ev[x].info;  // slots beyond the executors hold EV_NONE
      has[h] = info_kind(info[h]) != EV_NONE;
      j[h] = has[h] ? info_job(info[h]) : -1 - x;
      cnt[h] = 0, first[h] = false, rank[h] = 0;
      m[h] = wave_ballot(has[h]);
    }
    // (one pass per DISTINCT job with an event, not per executor: at 50 executors a third of the iterations)
    for (;;) {
      int h0 = -1, l = 0, jl = 0;  // the lowest executor that is still to be counted: l + 64 h0 (it is the lowest executor of its job)
      for (int h = SSS_EPL - 1; h >= 0; h--)  // (constant indices once unrolled: the arrays stay in registers)
        if (m[h]) h0 = h, l = ctz64_nz(m[h]), jl = (int)wave_readlane_u32((uint32_t)j[h], l);
      if (h0 < 0) break;
      uint64_t same[SSS_EPL];
      uint32_t total = 0;
      for (int h = 0; h < SSS_EPL; h++) same[h] = wave_ballot(has[h] && j[h] == jl), total += (uint32_t)popc64(same[h]);
      for (int h = 0; h < SSS_EPL; h++) {
        if (has[h] && j[h] == jl) cnt[h] = total, first[h] = h == h0 && lane == l;
        m[h] &= ~same[h];
      }
    }
    int nK = 0;
    for (int h = 0; h < SSS_EPL; h++) {
      key[h] = first[h] ? ((cnt[h] << 8) | (uint32_t)(64 * SSS_EPL - 1 - (lane + 64 * h))) : 0u;  // more events first, then the lower executor
      nK += popc64(wave_ballot(first[h]));
    }
    for (int h2 = 0; h2 < SSS_EPL; h2++)
      for (uint64_t fm = wave_ballot(first[h2]); fm; fm &= fm - 1) {
        const uint32_t kq = wave_readlane_u32(key[h2], ctz64_nz(fm));
        for (int h = 0; h < SSS_EPL; h++) rank[h] += kq > key[h] ? 1u : 0u;
      }
    const int n_used = nK < g_c.P.n_slots ? nK : g_c.P.n_slots;
    for (int h = 0; h < SSS_EPL; h++)
      if (first[h] && (int)rank[h] < g_c.P.n_slots) {
        lds_slot_of()[j[h]] = (uint8_t)rank[h];
        lds_slot_job()[rank[h]] = (uint16_t)j[h];
        lds_slot_ref()[rank[h]] = (uint8_t)cnt[h];
      }
    wave_sync();
    for (int h = 0; h < SSS_EPL; h++)
      if (has[h]) g_hot.ev[lane + 64 * h].info = info_with_slot(info[h], (uint32_t)lds_slot_of()[j[h]]);
    if (lane == 0) {
      uint64_t all = g_c.P.n_slots >= 64 ? ~0ull : (bit64(g_c.P.n_slots) - 1);
      uint64_t used = n_used >= 64 ? ~0ull : (bit64(n_used) - 1);
      g_sc.free_slots = all & ~used;
      g_sc.pending_free = -1, g_sc.pinned_job = -1, g_sc.idle_valid = 0, g_sc.fi_detach = 0;
      g_sc.events_this_step = 0;
      g_sc.active_version = 0, g_sc.old_version = 0, g_sc.jobset_valid = 0, g_sc.active_dirty = 0;
      g_sc.rng_pos = 64;  // the HBM image holds the generator's state itself, nothing is buffered yet
    }
  }
  wave_sync();
  // cached records: per slot 8 x u64 of job record, SP x u64 of stage counters, SP/2 x u64 of durations
  uint64_t occ = ~g_sc.free_slots & (g_c.P.n_slots >= 64 ? ~0ull : (bit64(g_c.P.n_slots) - 1));
  int nK = popc64(occ);  // slots 0 .. nK-1
  int per = 8 + g_c.SP + g_c.SP / 2;
  // eight words per lane at a time: all eight HBM loads are issued before the first LDS store (written as one loop
  // the stores - which may alias the slot map for all the compiler knows - would serialise the loads: one round
  // trip per 64 words)
  for (int i0 = lane; i0 < nK * per; i0 += 64 * 8) {
    uint64_t v[8];
    SSS_UNROLL8 for (int u = 0; u < 8; u++) {
      const int i = i0 + 64 * u;
      v[u] = 0;
      if (i < nK * per) {
        const int k = i / per, w = i - k * per;
        const int j = lds_slot_job()[k];
        v[u] = w < 8 ? ((const uint64_t*)(g_c.jobs + j))[w]
             : (w < 8 + g_c.SP ? ((const uint64_t*)(g_c.stages + j * g_c.SP))[w - 8] : ((const uint64_t*)(g_c.durations + j * g_c.SP))[w - 8 - g_c.SP]);
      }
    }
    SSS_UNROLL8 for (int u = 0; u < 8; u++) {
      const int i = i0 + 64 * u;
      if (i < nK * per) {
        const int k = i / per, w = i - k * per;
        if (w < 8)
          ((uint64_t*)(lds_cjobs() + k))[w] = v[u];
        else if (w < 8 + g_c.SP)
          ((uint64_t*)(lds_cstages() + k * g_c.SP))[w - 8] = v[u];
        else
          ((uint64_t*)(lds_cdur() + k * g_c.SP))[w - 8 - g_c.SP] = v[u];
      }
    }
  }
  wave_sync();
}

// What the on-device policies need of an env (read only: nothing is written back): the header, the ordered
// active-job list and "no job is cached" - job records then come straight from HBM, one lane each.
SSS_DEV void env_begin_readonly(const uint8_t* base) {
  int lane = wave_lane();
  const SssHot* g = (const SssHot*)base;
  if (lane < (int)(sizeof(SssHdr) / 16)) ((uint4*)&g_hot.h)[lane] = ((const uint4*)&g->h)[lane];
  for (int i = lane; i < g_c.J_cap; i += 64) lds_slot_of()[i] = SLOT_NONE;
  wave_sync();
  int A = g_hot.h.n_active;
  for (int i = lane; i < A; i += 64) lds_active()[i] = g_c.active_g[i];
  wave_sync();
}

SSS_DEV void env_end(uint8_t* base) {
  PROF3(24);
  int lane = wave_lane();
  wave_sync();
  if (lane == 0) rng_canonicalize();  // the HBM image never depends on what was buffered
  int A = g_hot.h.n_active;
  int per = 8 + g_c.SP + g_c.SP / 2;
  uint64_t occ = ~g_sc.free_slots & (g_c.P.n_slots >= 64 ? ~0ull : (bit64(g_c.P.n_slots) - 1));
  // lanes over (slot, word); free slots are skipped
  for (int i = lane; i < g_c.P.n_slots * per; i += 64) {
    int k = i / per, w = i - k * per;
    if (!(occ & bit64(k))) continue;
    int j = lds_slot_job()[k];
    if (w < 8)
      ((uint64_t*)(g_c.jobs + j))[w] = ((const uint64_t*)(lds_cjobs() + k))[w];
    else if (w < 8 + g_c.SP)
      ((uint64_t*)(g_c.stages + j * g_c.SP))[w - 8] = ((const uint64_t*)(lds_cstages() + k * g_c.SP))[w - 8];
    else
      ((uint64_t*)(g_c.durations + j * g_c.SP))[w - 8 - g_c.SP] = ((const uint64_t*)(lds_cdur() + k * g_c.SP))[w - 8 - g_c.SP];
  }
  if (g_sc.active_dirty)  // (wave-uniform: read behind the ordering point above)
    for (int i = lane; i < A; i += 64) g_c.active_g[i] = lds_active()[i];
  for (int x = lane; x < g_c.E; x += 64) {  // the HBM image of an event does not name an LDS slot
    uint32_t info = g_hot.ev[x].info;
    if (info_kind(info) != EV_NONE) g_hot.ev[x].info = info_with_slot(info, INFO_SLOT_NONE);
  }
  wave_sync();
  {
    SssHot* g = (SssHot*)base;
    if (lane < (int)(sizeof(SssHdr) / 16)) ((uint4*)&g->h)[lane] = ((const uint4*)&g_hot.h)[lane];
    const int n_commits = g_hot.h.n_commits;  // entries behind the live ones are never read again: they stay what they are in HBM
    for (int x = lane; x < g_c.E; x += 64) {
      g->ev[x] = g_hot.ev[x], g->ex_loc[x] = g_hot.ex_loc[x], g->ex_job[x] = g_hot.ex_job[x];
      g->ex_task_stage[x] = g_hot.ex_task_stage[x], g->ex_executing[x] = g_hot.ex_executing[x];
      if (x < n_commits) g->c_src[x] = g_hot.c_src[x], g->c_dst[x] = g_hot.c_dst[x], g->c_seq[x] = g_hot.c_seq[x], g->c_n[x] = g_hot.c_n[x];
    }
  }
}

// ------------------------------------------------------------------------------------------
// step pieces (lane 0)
// ------------------------------------------------------------------------------------------

// stage_selection_map[stage_idx] (ENV:284, 386-392): the k-th set bit over the per-job schedulable masks in active
// order - all lanes, one job each (the records of jobs without a cache slot come from HBM: one round trip for
// all of them instead of one per job on lane 0). Leaves (job, stage) or (-1, -1) in the mailbox.
SSS_DEV void select_stage_wave(int stage_idx) {
  PROF3(34);
  const int lane = wave_lane();
  const int A = g_hot.h.n_active;
  int run = 0, fj = -1, fs = -1;
  for (int a0 = 0; a0 < A && stage_idx >= 0; a0 += 64) {
    const int a = a0 + lane;
    uint64_t m = 0;
    int jj = -1;
    if (a < A) jj = lds_active()[a], m = jobp(jj)->sched_mask;
    const uint32_t n = (uint32_t)popc64(m);
    const int lo = run + (int)wave_scan_excl_u32(n);
    run += (int)wave_sum_u32(n);
    const bool mine = stage_idx >= lo && stage_idx < lo + (int)n;
    const uint64_t hit = wave_ballot(mine);
    if (hit) {
      if (mine) {
        for (int i = 0; i < stage_idx - lo; i++) m &= m - 1;
        g_sc.sel_job = jj, g_sc.sel_stage = ctz64(m);
      }
      fj = 0;
      break;
    }
  }
  if (fj < 0 && lane == 0) g_sc.sel_job = -1, g_sc.sel_stage = -1;
  (void)fs;
  wave_sync();
}

// ENV:275-315. Returns false if the action was rejected (state untouched).
SSS_DEV bool take_action(int stage_idx, int num_exec) {
  PROF3(17);
  // action_space.contains: stage_idx in [-1, n_nodes), num_exec in [1, E] (ENV:85-94, 404)
  if (stage_idx < -1 || stage_idx >= H.obs_n_nodes || num_exec < 1 || num_exec > g_c.E) {
    H.err = SSS_ERR_ACTION_SPACE;
    return false;
  }
  if (stage_idx == -1) {
    commit_remaining_executors();
    return true;
  }
  if (stage_idx >= H.obs_n_sched) {  // KeyError on stage_selection_map (ENV:284)
    H.err = SSS_ERR_STAGE_IDX;
    return false;
  }
  if (num_exec > trk_num_committable()) {
    H.err = SSS_ERR_TOO_MANY;
    return false;
  }
  // stage_selection_map[stage_idx]: found by the whole wave beforehand (select_stage_wave)
  const int j = g_sc.sel_job, s = g_sc.sel_stage;
  CHECK(j >= 0);
  if (j < 0) return false;
  SssStage st = (*stgp(j, s));
  int demand = (int)st.remaining - ((int)st.moving_to + (int)st.commit_to);  // ENV:557-578
  int n = num_exec < demand ? num_exec : demand;
  CHECK(n > 0);
  trk_add_commitment(n, key_stage_pool(j, s));
  SssJob& job = (*jobp(j));
  job.selected_mask |= bit64(s);  // ENV:304
  // ENV:307-315: only this job's slice of schedulable_stages is recomputed
  int old_n = popc64(job.sched_mask);
  uint64_t m = 0;
  if (job_passes_filter(j, trk_source_job_id())) m = ready_mask_of_job(job, false);
  job.sched_mask = m;
  H.n_sched += popc64(m) - old_n;
  return true;
}

// ENV:847-874. The float sum runs in CPython set(list + list) iteration order: lane 0 builds the
// set image (jobtime_build_set), then all lanes evaluate one table slot each and the terms are
// added in slot order (jobtime_sum) - the additions stay sequential, the HBM reads do not.
SSS_DEV void jobtime_build_set() {
  PROF3(18);
  SetImg<uint16_t> all;
  all.tab = lds_jobset();
  for (int i = 0; i < 8; i++) all.tab[i] = 0;
  all.mask = 7, all.fill = 0, all.used = 0, all.finger = 0, all.cap = 0xFFFFFFFFu, all.big = nullptr, all.small = nullptr, all.wide = false;
  for (int k = 0; k < g_sc.n_old_active; k++) set_add(all, (uint32_t)lds_old_active()[k], lds_keys());
  for (int k = 0; k < H.n_active; k++) set_add(all, (uint32_t)lds_active()[k], lds_keys());
  g_sc.jobset_mask = (int32_t)all.mask;
}

// The same image with the whole wave, when it does not depend on the order of the additions: n distinct ids
// grow the table 8 -> 32 (5th) -> 128 (19th) -> 512 (77th) -> 2048 (307th id) slots, every resize re-inserts
// into an empty table, and once the table is larger than the largest id every id sits in its home slot with
// no collision possible - whatever happened in the smaller tables before. Otherwise lane 0 builds it (above).
SSS_DEV void jobtime_build_set_wave() {
  const int lane = wave_lane();
  uint16_t* tab = lds_jobset();
  const int cap = g_c.P.jobset_slots;
  const int n_old = g_sc.n_old_active, n_act = g_hot.h.n_active;
  for (int i = lane * 8; i < cap; i += 64 * 8) *(uint4*)(tab + i) = mk_u4(0u, 0u, 0u, 0u);
  wave_sync();
  uint32_t not_max = 0xFFFFFFFFu;
  for (int k = lane; k < n_old + n_act; k += 64) {
    const uint32_t id = k < n_old ? lds_old_active()[k] : lds_active()[k - n_old];
    tab[id] = (uint16_t)(id + 2);
    not_max = ~id < not_max ? ~id : not_max;
  }
  wave_sync();
  uint32_t cnt = 0;
  for (int i = lane * 8; i < cap; i += 64 * 8) {
    const uint4 q = *(const uint4*)(tab + i);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    for (int b = 0; b < 4; b++) cnt += ((w[b] & 0xFFFFu) != 0 ? 1u : 0u) + ((w[b] >> 16) != 0 ? 1u : 0u);
  }
  const uint32_t n = wave_sum_u32(cnt);
  const uint32_t max_id = n ? ~wave_min_u32(not_max) : 0u;
  const uint32_t mask = n < 5 ? 7u : (n < 19 ? 31u : (n < 77 ? 127u : (n < 307 ? 511u : 2047u)));
  if (max_id <= mask && (int)mask < cap) {
    if (lane == 0) g_sc.jobset_mask = (int32_t)mask;
  } else {
    wave_sync();
    if (lane == 0) jobtime_build_set();
  }
  wave_sync();
}

SSS_DEV double jobtime_sum() {
  PROF3(25);
  int lane = wave_lane();
  double wall_old = g_sc.wall_old, wall = g_hot.h.wall_time;
  int mask = g_sc.jobset_mask;
  double beta = g_c.P.beta;
  const uint16_t* tab = lds_jobset();
  double job_time = 0.0;
  // four groups of 64 slots at a time: the arrival / completion times of all four are on their way from HBM before
  // the first is used (large tables: 512 slots at 200 jobs); the additions stay in slot order
  for (int b0 = 0; b0 <= mask; b0 += 256) {
    double ta[4], tc[4];
    bool live[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int b = b0 + 64 * u;
      const uint32_t en = (b + lane) <= mask ? (uint32_t)tab[b + lane] : 0u;  // tables are >= 8 slots; slots beyond the mask are never live
      live[u] = en >= 2;
      ta[u] = 0.0, tc[u] = 0.0;
      if (live[u]) ta[u] = g_c.t_arrival[(int)en - 2], tc[u] = g_c.t_completed[(int)en - 2];
    }
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      if (b0 + 64 * u > mask) break;
      double term = 0.0;
      if (live[u]) {
        double start = ta[u] > wall_old ? ta[u] : wall_old;
        double end = tc[u] < wall ? tc[u] : wall;
        if (beta == 0.0)
          term = end - start;
        else  // np.exp in the reference: <= 2 ulp agreement only
          term = fd_exp(-beta * 1e-3 * (start - wall_old)) - fd_exp(-beta * 1e-3 * (end - wall_old));
      }
      uint64_t m = wave_ballot(live[u]);
      while (m) {
        int k = ctz64(m);
        m &= m - 1;
        job_time += wave_bcast_f64(term, k);
      }
    }
  }
  if (beta > 0.0) job_time /= beta;
  return job_time;
}

// ------------------------------------------------------------------------------------------
// whole-env procedures (all lanes)
// ------------------------------------------------------------------------------------------

// _resume_simulation (ENV:320-343). Entered and left with LDS in sync.
// lane 0: one popped event. Returns 0 = keep going, 1 = queue empty / failed, 2 = a scan is needed
// (committable executors exist).
SSS_DEV int handle_popped(const FastCtx& f, int ex, double t_win, uint32_t info_win, uint64_t& t_slow) {
  UTRACE("handle_popped");
  PROF3(31);
  if (ex == POP_EMPTY) return 1;
  H.n_events++;
  g_sc.events_this_step++;
  int fast = 0;
  if (ex >= 0 && info_kind(info_win) == EV_TASK_FINISHED)
    fast = fast_task_completion(f, ex, t_win, info_job(info_win), info_stage(info_win), info_slot(info_win));
  if (fast > 0) {
    // the source stays what it was - None right after a scheduling round - so nothing is
    // committable and the loop continues (ENV:331-332)
    H.n_fast++;
    if (H.curr_source == POOL_NONE) return 0;
  } else {
    // everything else goes through the out-of-line handlers
    if (fast < 0) FAIL(SSS_ERR_NO_DURATION);
    uint64_t ts0 = wave_clock();
    if (fast < 0) {
    } else if (ex == POP_ARRIVAL) {
      int job = H.next_arrival;
      H.wall_time = H.next_arrival_t;
      H.next_arrival++;
      H.next_arrival_t = H.next_arrival < H.J ? g_c.t_arrival[H.next_arrival] : __builtin_inf();
      handle_job_arrival(job);
    } else {
      SssHot& hot = g_hot;
      SssEvSlot sl = hot.ev[ex];
      H.wall_time = sl.t;
      hot.ev[ex].t = __builtin_inf();
      hot.ev[ex].info = EV_NONE;
      if (info_slot(sl.info) != INFO_SLOT_NONE) lds_slot_ref()[info_slot(sl.info)]--;
      g_sc.pinned_job = info_job(sl.info);
      if (info_kind(sl.info) == EV_TASK_FINISHED) {
        STAT(43, 1), STAT(44, H.curr_source != POOL_NONE), STAT(45, info_slot(sl.info) == INFO_SLOT_NONE);
        handle_task_completion(ex, info_job(sl.info), info_stage(sl.info));
      }
      else {
        STAT(39, 1), STAT(40, H.curr_source != POOL_NONE), STAT(41, info_slot(sl.info) == INFO_SLOT_NONE);
        STAT(42, (*stgp(info_job(sl.info), info_stage(sl.info))).remaining == 0);
        handle_executor_arrival(ex, info_job(sl.info), info_stage(sl.info));
      }
      g_sc.pinned_job = -1;
    }
    if (g_sc.pending_free >= 0) {
      // a completed job gives its slot back right away - unless an executor is still on its way to
      // it (its EXECUTOR_READY names the slot); then the slot is handed on later like any other
      int k = lds_slot_of()[g_sc.pending_free];
      if (k != SLOT_NONE && lds_slot_ref()[k] == 0) cache_release(g_sc.pending_free);
      g_sc.pending_free = -1;
    }
    t_slow += wave_clock() - ts0;
  }
  if (H.err) return 1;
  if (trk_num_committable() > 0) {
    publish_scan_inputs();
    return 2;
  }
  return 0;
}

// budget > 0: the loop also ends (returns true) at the top of a round once the step has taken that many events - everything is
// in the env's state there, the next launch goes on from it (do_step)
SSS_DEV bool resume_simulation(int budget = 0) {
  PROF3(26);
  int lane = wave_lane();
  FastCtx f;
  fastctx_load(f);
  // raw generator outputs the event loop wants to find buffered at the top of a round: two per
  // event of a batch (batches are cut to what is there, so this only has to be "enough")
  const int rng_need = 2 * (f.E < 20 ? f.E : 20);
  (void)rng_need;
  for (;;) {
    // events run until the wave is needed for a schedulable-stage scan, the queue is empty, or
    // something failed. A round = a run of "task finished, stage has more tasks" events if the head of
    // the queue is one (fast_run), a lane-parallel batch of released or arriving executors if it allows
    // one, else one event popped by a wave reduction and handled on lane 0; the loop decision travels
    // through a lane-0 broadcast (no LDS flags, no barrier per event).
    uint64_t t_slow = 0;
    int status;
    do {
      status = 0;
      // the head of the queue decides what kind of round this is
      double t_win = 0.0;
      uint32_t info_win = 0;
      // A round can be "lane 0 handles one event, then every lane looks at the queue again" with nothing in between (the fast run
      // and the batches end in ordering points, a declined attempt and the one-at-a-time handler do not): this one makes lane 0's
      // LDS writes of the previous round (event slots, arrival cursor, stage counters) something the other lanes' reads below
      // cannot have been scheduled ahead of. Found on the GPU with the wide instantiation (every round is of that kind there; the
      // emulator's lanes run one after the other and cannot show it); at wavefront scope the point costs no instruction.
      wave_sync();
      if (budget > 0 && g_sc.events_this_step - g_sc.events_at_launch >= budget) return true;  // (wave-uniform: an LDS word behind the ordering point)
      double next_arrival_t = g_hot.h.next_arrival < g_hot.h.J ? g_hot.h.next_arrival_t : __builtin_inf();
      int ex = pop_event_wave(next_arrival_t, t_win, info_win);
      const bool head_cached = ex >= 0 && info_slot(info_win) != INFO_SLOT_NONE;
      if (head_cached || (ex >= 0 && info_kind(info_win) == EV_TASK_FINISHED)) {
        int handled = 0;
        // tasks left in its stage: a run of such events (fast_run produces the randomness it needs itself); when it
        // ends, the head of the queue is something else. None left: a batch of released executors. Those and the
        // batches of arriving executors want `rng_need` raw outputs buffered
        const bool tf = info_kind(info_win) == EV_TASK_FINISHED;
        const bool tasks_left = tf && head_cached && f.cstages[info_slot(info_win) * f.SP + info_stage(info_win)].remaining > 0;
        if (!head_cached) {
          // a task completion of a job without a cache slot (more jobs with pending events than slots): the run looks at the job's
          // HBM records itself and declines (0: nothing touched) when the stage has no task left; the batches below want slots
          handled = fast_run(f);
        } else if (tasks_left) {
          handled = fast_run(f);
        } else {
#ifndef SSS_NO_BATCH
          if (64 - g_sc.rng_pos < rng_need) {
            rng_refill();
          }
#endif
          handled = tf ? batch_released_events(f, ex) : batch_arrival_events(f, ex);
          // a released executor on its own (the usual case): the wave-uniform single-event path
#ifndef SSS_NO_LEAN  // (A/B timing builds)
          if (handled == 0) handled = tf ? lean_released(f, ex, t_win, info_win) : lean_arrival(f, ex, t_win, info_win);
#endif
          // the event that completes a job: the executors parked in the job's pool are flushed with the whole wave first
          if (handled == 0 && tf) preflush_completing_job(f, info_win);
        }
        if (handled > 0) continue;
        // nothing was touched: the popped event goes the one-at-a-time way, which is always right
      }
      if (lane == 0) status = handle_popped(f, ex, t_win, info_win, t_slow);
      status = (int)wave_lane0_u32((uint32_t)status);
    } while (status == 0);
    if (lane == 0) {
      g_sc.f_done = status == 1, g_sc.f_scan = status == 2;
      H.prof[0] += t_slow;
    }
    wave_sync();
    if (g_sc.f_done) {
      // queue exhausted (or failed): schedulable_stages = [] (ENV:324,343)
      if (lane == 0) {
        for (int a = 0; a < H.n_active; a++) (*jobp(lds_active()[a])).sched_mask = 0;
        H.n_sched = 0;
      }
      wave_sync();
      return false;
    }
    // f_scan: _find_schedulable_stages() with the whole wave
    int n = find_schedulable_all();
    if (n > 0) {
      if (lane == 0) H.n_sched = n;
      wave_sync();
      return false;
    }
    publish_idle_mask();
    if (lane == 0) {
      move_idle_executors_all(POOL_NONE);  // ENV:340
      H.curr_source = POOL_NONE;               // ENV:341
      g_sc.idle_valid = 0;
    }
    wave_sync();
  }
}

// episode initialisation: ENV:127-186 + TPCH:54-73,176-206 + TRK:32-71
SSS_DEV void do_reset(const SssLayout& L, uint64_t seed, double time_limit) {
  PROF3(27);
  int lane = wave_lane();
  SssHot& hot = g_hot;
  // nothing is cached while the records are (re)built in HBM
  for (int i = lane; i < g_c.J_cap; i += 64) lds_slot_of()[i] = SLOT_NONE;
  lds_slot_ref()[lane] = 0;
  wave_sync();
  if (lane == 0) {
    g_sc.free_slots = g_c.P.n_slots >= 64 ? ~0ull : (bit64(g_c.P.n_slots) - 1);
    g_sc.pending_free = -1, g_sc.pinned_job = -1, g_sc.idle_valid = 0, g_sc.fi_detach = 0;
    // lifetime counters and the duration deque survive resets (ENV:83)
    uint64_t n_steps = H.n_steps, n_events = H.n_events, model_bytes = H.model_bytes;
    int dur_head = H.dur_head, dur_n = H.dur_n, episodes = H.episodes, last_ep_steps = H.last_ep_steps;
    double last_ep_return = H.last_ep_return, last_ep_wall = H.last_ep_wall;
    uint64_t prof[5];
    for (int i = 0; i < 5; i++) prof[i] = H.prof[i];
    uint64_t n_fast_keep = H.n_fast, n_batched_keep = H.n_batched, n_rounds_keep = H.n_rounds, pad0_keep = H.err_line;
    SssHdr z = {};
    H = z;
    for (int i = 0; i < 5; i++) H.prof[i] = prof[i];
    H.n_fast = n_fast_keep, H.n_batched = n_batched_keep, H.n_rounds = n_rounds_keep, H.err_line = pad0_keep;
    H.n_steps = n_steps, H.n_events = n_events, H.model_bytes = model_bytes;
    H.dur_head = dur_head, H.dur_n = dur_n, H.episodes = episodes;
    H.last_ep_steps = last_ep_steps, H.last_ep_return = last_ep_return, H.last_ep_wall = last_ep_wall;
    H.seed = seed, H.time_limit = time_limit;
    H.graph_version = 1;  // (obs_graph_version = 0: the first observation writes its edge rows)
    H.curr_source = POOL_COMMON;
    g_sc.events_this_step = 0;
    g_sc.reset_more = 0;
    if (!(time_limit < __builtin_inf()) && g_c.P.cap_cfg <= 0) {
      H.err = SSS_ERR_NO_LIMIT;  // ENV:137-138
      H.need_reset = 1;
    } else {
      rng_seed(H, seed);
      g_sc.rng_pos = 64;  // nothing buffered: the header holds the generator's state itself
      H.J = 0;
      g_sc.reset_t = 0.0, g_sc.reset_more = 1;
    }
  }
  wave_sync();
  // job_sequence TPCH:54-73: lane 0 draws job after job from raw outputs the wave produces 64 at a time
  // (a job takes two of them unless the exponential leaves the ziggurat's fast path)
  while (g_sc.reset_more) {
    rng_refill();
    if (lane == 0) {
      double t = g_sc.reset_t;
      int J = H.J;
      bool more = true;
      while (g_sc.rng_pos <= 56) {
        if (!(t < time_limit && (g_c.P.cap_cfg <= 0 || J < g_c.P.cap_cfg))) {
          more = false;
          break;
        }
        if (J >= g_c.J_cap) {
          H.err = SSS_ERR_CAPACITY;
          H.need_reset = 1;
          more = false;
          break;
        }
        int q = (int)rng_integers((uint32_t)g_c.pk.n_queries);   // TPCH:177
        int size = (int)rng_integers((uint32_t)g_c.pk.n_sizes);  // TPCH:178
        (*jobp(J)).gs_base = q * g_c.pk.n_sizes + size;  // template id for now; resolved to pack rows below
        g_c.t_arrival[J] = t;
        J++;
        t += g_c.P.mean_interarrival * rng_standard_exponential();  // TPCH:70
      }
      H.J = J, g_sc.reset_t = t, g_sc.reset_more = more ? 1 : 0;
    }
    wave_sync();
  }
  // executors + event slots + commitments
  for (int x = lane; x < SSS_MAX_EXEC; x += 64) {
    hot.ev[x].t = __builtin_inf(), hot.ev[x].seq = 0, hot.ev[x].info = EV_NONE;
    hot.ex_loc[x] = x < g_c.E ? POOL_COMMON : POOL_NONE;
    hot.ex_job[x] = -1;
    hot.ex_task_stage[x] = -1, hot.ex_executing[x] = 0;
    hot.c_src[x] = POOL_NONE, hot.c_dst[x] = POOL_NONE, hot.c_seq[x] = 0, hot.c_n[x] = 0;
  }
  wave_sync();
  int J = hot.h.J;
  // job records: one lane per job
  for (int j = lane; j < J; j += 64) {
    SssJob& job = (*jobp(j));
    int tmpl = job.gs_base;
    int gs = g_c.pk.tmpl_stage_off[tmpl];
    int ns = g_c.pk.tmpl_stage_off[tmpl + 1] - gs;
    uint64_t frontier = 0;
    for (int s = 0; s < ns; s++)
      if (g_c.pk.stage_parent_mask[gs + s] == 0) frontier |= bit64(s);  // JOB:93-111
    job.active_mask = ns >= 64 ? ~0ull : (bit64(ns) - 1);
    job.frontier_mask = frontier;
    job.selected_mask = 0, job.sched_mask = 0, job.sat_mask = 0, job.local_mask = 0;
    job.supply = 0, job.sat_count = 0, job.completion_order = -1;
    job.n_stages = (uint8_t)ns;
    job.n_edges = (uint8_t)(g_c.pk.tmpl_edge_off[tmpl + 1] - g_c.pk.tmpl_edge_off[tmpl]);
    job.edge_off = g_c.pk.tmpl_edge_off[tmpl];
    job.gs_base = gs;
    g_c.t_completed[j] = __builtin_inf();
  }
  wave_sync();
  // stage records: lanes over (job, stage)
  for (int i = lane; i < J * g_c.SP; i += 64) {
    int j = i / g_c.SP, s = i - j * g_c.SP;
    const SssJob& job = (*jobp(j));
    SssStage st = {0, 0, 0, 0};
    float d = 0.0f;
    if (s < (int)job.n_stages) {
      st.remaining = g_c.pk.stage_num_tasks[job.gs_base + s];
      d = (float)g_c.pk.stage_rough[job.gs_base + s];
    }
    g_c.stages[i] = st;
    g_c.durations[i] = d;
  }
  // pools: every job / stage pool starts as an empty 8-slot set (TRK:73-96); the common pool as
  // set(range(E)) (TRK:41), whose image the host has built once (sss_host.h: sss_build_common_pool)
  int n_pools = 1 + g_c.J_cap + J * g_c.SP;
  for (int p = lane; p < n_pools; p += 64) {
    *(uint4*)(g_c.pool_hdr + p) = p == 0 ? ((const uint4*)g_c.pk.common_pool)[0] : mk_u4(7u, 0u, 0u, 0u);  // mask 7, fill 0, used 0, no commitments, empty 8-slot table
  }
  {
    const uint32_t cmask = ((const uint32_t*)g_c.pk.common_pool)[0] & 0xFFFFu;
    if (cmask != 7)
      for (uint32_t w = (uint32_t)lane; w < (cmask + 1) / 16; w += 64) ((uint4*)g_c.pool_tab)[w] = ((const uint4*)g_c.pk.common_pool)[1 + w];
  }
  wave_sync();
  if (lane == 0 && !H.err) {
    // _load_initial_jobs ENV:260-273
    while (H.next_arrival < H.J && g_c.t_arrival[H.next_arrival] <= 0.0) {
      handle_job_arrival(H.next_arrival);
      H.next_arrival++;
    }
    H.next_arrival_t = H.next_arrival < H.J ? g_c.t_arrival[H.next_arrival] : __builtin_inf();
  }
  if (lane == 0) publish_scan_inputs();
  wave_sync();
  int n = find_schedulable_all();
  if (lane == 0) H.n_sched = n;
  wave_sync();
}

SSS_DEV double step_reward(uint64_t t1, uint64_t t2);
// A step whose event loop was cut at its budget (H.mid_step): the scratch the first part of the step left for its end comes
// back from the header and from behind the active list in HBM; the action arguments of this launch are not looked at.
SSS_DEV void step_continue() {
  const int lane = wave_lane();
  for (int a = lane; a < g_hot.h.n_old_active; a += 64) lds_old_active()[a] = g_c.active_g[g_c.J_cap + a];
  if (lane == 0) {
    g_sc.f_round_continues = 0, g_sc.f_fulfil = 0, g_sc.idle_valid = 0;
    g_sc.events_this_step = H.step_events, g_sc.events_at_launch = H.step_events;
    g_sc.wall_old = H.wall_old, g_sc.n_old_active = H.n_old_active;
    g_sc.old_version = g_sc.active_version, g_sc.jobset_valid = 0;
    H.mid_step = 0;
  }
  wave_sync();
}
SSS_DEV void step_yield() {
  const int lane = wave_lane();
  for (int a = lane; a < g_sc.n_old_active; a += 64) g_c.active_g[g_c.J_cap + a] = lds_old_active()[a];
  if (lane == 0) H.mid_step = 1, H.step_events = g_sc.events_this_step, H.wall_old = g_sc.wall_old, H.n_old_active = g_sc.n_old_active;
  wave_sync();
}

// ENV:188-221. `reward` is valid on lane 0 (and uniform). budget > 0 (sss_step_bounded): at most about that many events per
// launch - *yielded is set when the step's event loop has not reached its end (no reward, no observation yet; the next
// launch continues it).
SSS_DEV double do_step(int stage_idx, int num_exec, int budget = 0, bool* yielded = nullptr) {
  PROF3(28);
  int lane = wave_lane();
  uint64_t t0 = wave_clock();
  const bool go_on = wave_ballot(g_hot.h.mid_step != 0) != 0;  // (the ballot: every lane has read the header before lane 0 rewrites it)
  uint64_t t1 = t0;
  if (go_on) {
    step_continue();
  } else {
  publish_idle_mask();  // for fulfil_build_list, should the round end with this action (nothing below moves an executor before it)
  select_stage_wave(stage_idx);
  if (lane == 0) {
    g_sc.f_round_continues = 1, g_sc.f_fulfil = 0;
    g_sc.events_this_step = 0, g_sc.events_at_launch = 0;
    H.last_reward = 0.0;
    if (H.need_reset || H.terminated) {
      H.err = SSS_ERR_NEED_RESET;
    } else {
      H.err = 0;
      bool ok = take_action(stage_idx, num_exec);
      if (ok && !H.err) {
        H.n_steps++;
        H.ep_steps++;
        if (!(trk_num_committable() > 0 && H.n_sched > 0)) {
          // commitment round is over (ENV:195-203)
          commit_remaining_executors();
          g_sc.f_fulfil = 1;
        }
      }
    }
    if (!g_sc.f_fulfil) g_sc.idle_valid = 0;
  }
  wave_sync();
  if (g_sc.f_fulfil) {
    fulfil_order_commitments();
    if (lane == 0) {
      fulfil_build_list();
      g_sc.idle_valid = 0;
    }
    wave_sync();
    fulfil_run();
    if (lane == 0) {
      H.curr_source = POOL_NONE;
      g_sc.wall_old = H.wall_time;
      g_sc.n_old_active = H.n_active;
      g_sc.old_version = g_sc.active_version;
      g_sc.f_round_continues = 0;  // selected_stages.clear() and the old-active snapshot follow, lanes over jobs
    }
  }
  if (lane == 0 && H.err && H.err != SSS_ERR_ACTION_SPACE && H.err != SSS_ERR_STAGE_IDX && H.err != SSS_ERR_TOO_MANY) H.need_reset = 1;
  wave_sync();
  t1 = wave_clock();
  if (lane == 0) H.prof[1] += t1 - t0;
  if (wave_ballot(g_sc.f_round_continues || g_hot.h.err) != 0) return 0.0;  // same round: reward 0 (ENV:191-193)
  for (int a = lane; a < g_hot.h.n_active; a += 64) {  // ENV:203 selected_stages.clear(); active jobs at the round's end
    int j = lds_active()[a];
    (*jobp(j)).selected_mask = 0;
    lds_old_active()[a] = (uint16_t)j;
  }
  wave_sync();
  }
  if (resume_simulation(budget)) {
    step_yield();
    *yielded = true;
    return 0.0;
  }
  return step_reward(t1, wave_clock());
}

// the end of a step: reward = -job_time (ENV:208-209), termination, the stall check
SSS_DEV double step_reward(uint64_t t1, uint64_t t2) {
  const int lane = wave_lane();
  // `duration == 0.0` short-circuits to -0.0 (ENV:850-852)
  if (lane == 0) {
    g_sc.f_need_jobtime = 0;
    if (!H.err && H.wall_time - g_sc.wall_old != 0.0) {
      if (!(g_sc.jobset_valid && g_sc.jobset_old_v == g_sc.old_version && g_sc.jobset_new_v == g_sc.active_version)) {
        g_sc.f_need_jobtime = 2;  // the set image is built first
        g_sc.jobset_valid = 1, g_sc.jobset_old_v = g_sc.old_version, g_sc.jobset_new_v = g_sc.active_version;
      } else
        g_sc.f_need_jobtime = 1;
    }
  }
  wave_sync();
  double job_time = 0.0;
  if (g_sc.f_need_jobtime == 2) jobtime_build_set_wave();
  if (g_sc.f_need_jobtime) job_time = jobtime_sum();
  double reward = 0.0;
  if (lane == 0) {
    if (!H.err) {
      reward = -job_time;
      H.terminated = H.n_completed == H.J;  // ENV:227-229
      if (!H.terminated && !(trk_num_committable() > 0 && H.n_sched > 0)) H.err = SSS_ERR_STALLED;  // ENV:212-215
      H.ep_return += reward;
      if (H.terminated) {
        H.episodes++;
        H.last_ep_return = H.ep_return, H.last_ep_steps = H.ep_steps, H.last_ep_wall = H.wall_time;
      }
    }
    if (H.err) H.need_reset = 1;
    uint64_t t3 = wave_clock();
    H.prof[2] += t2 - t1, H.prof[3] += t3 - t2;
  }
  wave_sync();
  return reward;
}

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
      reward = do_step(si, ne);
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
