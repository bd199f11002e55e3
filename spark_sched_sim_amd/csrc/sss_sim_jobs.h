// sss_sim_jobs.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// jobs / stages, the task-duration sampler, the serial schedulable-stage search, executor movement (lane 0).
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 4  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
