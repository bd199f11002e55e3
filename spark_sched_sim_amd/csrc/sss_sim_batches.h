// sss_sim_batches.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// lane-parallel batches of released and of arriving executors.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 8  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
