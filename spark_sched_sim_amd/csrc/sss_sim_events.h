// sss_sim_events.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// the one-at-a-time event handlers (lane 0), the wave-parallel queue pop, the LDS job cache.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 6  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
