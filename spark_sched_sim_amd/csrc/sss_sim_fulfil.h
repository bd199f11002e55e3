// sss_sim_fulfil.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// lane-parallel fulfilment of commitments.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 5  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
