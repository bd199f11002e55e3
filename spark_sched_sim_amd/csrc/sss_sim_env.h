// sss_sim_env.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// staging at launch boundaries, the pieces of a step (action, reward), the event loop, episode initialisation, do_step.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 11  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
template <bool FUSED = false>  // (FUSED: the rollout kernel's instantiation - fast_run<true>)
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
          handled = fast_run<FUSED>(f);
        } else if (tasks_left) {
          handled = fast_run<FUSED>(f);
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
template <bool FUSED = false>
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
  if (resume_simulation<FUSED>(budget)) {
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
