// sss_sim_lean.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// one released / one arriving executor with wave-uniform control flow; the flush of a completing job.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 9  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
