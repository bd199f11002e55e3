// experiments/batch_fast_events.h - NOT compiled into the product.
//
// Conservative-lookahead batching of independent fast-path events (one lane per event, RNG outputs
// handed out by stream position). It was wired into resume_simulation() as
//
//     if (batch_fast_events(f, r, n_fast) > 0) { status = 0; continue; }   // before pop_event_wave
//
// and is CORRECT: with it the kernel source passed every emulator golden / policy test and, on the
// MI355X, tests/test_gpu_parity.py and tests/test_gpu_fullsize_oracle.py (4096 envs, bit-exact
// episode summaries). It is SLOWER than the one-event-at-a-time loop on this hardware (round 1):
//
//     C2 sizing: 4.77 M vs 5.48 M env-steps/s (step mode), 19.9 M vs 23.4 M (rollout)
//     C3 sizing: 2.00 M vs 2.51 M,                         7.5 M vs 11.4 M
//
// Batch statistics (per env step): C2 16 attempts, 8 succeed, 3.7 events per successful batch
// (80 % of all events); C3 11 attempts, 2.5 succeed, 4.5 events per batch. The per-attempt
// overhead (making lane 0's loop registers uniform, ranking E events with v_readlane, ballots per
// rank, the validity scan, a second walk of the RNG stream) costs about as many issued
// instructions as the parallel handlers save, and a wave issues one instruction at a time no
// matter how many lanes are active. Kept for the next round: cheaper ranking (DPP bitonic),
// attempting only after a fast event, and SALU-only RNG walks are the obvious things to try.
//
// Needs from wave_rt.h: wave_readlane_u32/f64, lane_atomic_add_i32, lane_atomic_or_u64 (all still
// provided by both the gfx950 and the emulator runtime).

// ------------------------------------------------------------------------------------------
// Batched fast path: several independent "task finished, stage has more tasks" events at once.
//
// The next K events by (time, push counter) can be handled together - one lane each - when
//   * they are all fast-path events (TASK_FINISHED, their stage keeps >= 1 remaining task after the
//     batch members before them on the same stage took theirs), none is preceded by a job arrival,
//     and no scheduling round is pending (source == None, so nothing becomes committable);
//   * every rescheduled completion time t_i + d_i is >= the event times of the batch members after
//     it, i.e. the pop order of the reference's heap is exactly the batch order.
// Their only coupling is the RNG stream, consumed in event order: which 64-bit outputs an event
// uses is known before drawing (random() iff the executor level is interpolated, TPCH:221-229; one
// buffered 32-bit half for the bounded integer), so the outputs are generated once, uniformly, and
// handed to the lanes by position. A Lemire rejection (probability len / 2^32) or any doubt ends
// the batch before that event; the single-event path takes over. Results are bit-identical to
// processing the events one by one (tests replay the reference's trajectories through this path).
// All lanes call it; returns the number of events committed (wave-uniform, 0 = none).
// ------------------------------------------------------------------------------------------
#define SSS_BATCH_MAX 8

SSS_DEV int batch_fast_events(const FastCtx& f, EvRegs& r, uint64_t& n_fast) {
  int lane = wave_lane();
  // lane 0 owns the loop registers between batches: make them uniform
  r.rng_state_hi = ((uint64_t)wave_lane0_u32((uint32_t)(r.rng_state_hi >> 32)) << 32) | wave_lane0_u32((uint32_t)r.rng_state_hi);
  r.rng_state_lo = ((uint64_t)wave_lane0_u32((uint32_t)(r.rng_state_lo >> 32)) << 32) | wave_lane0_u32((uint32_t)r.rng_state_lo);
  r.rng_inc_hi = ((uint64_t)wave_lane0_u32((uint32_t)(r.rng_inc_hi >> 32)) << 32) | wave_lane0_u32((uint32_t)r.rng_inc_hi);
  r.rng_inc_lo = ((uint64_t)wave_lane0_u32((uint32_t)(r.rng_inc_lo >> 32)) << 32) | wave_lane0_u32((uint32_t)r.rng_inc_lo);
  r.rng_has32 = wave_lane0_u32(r.rng_has32), r.rng_u32 = wave_lane0_u32(r.rng_u32);
  r.counter = wave_lane0_u32(r.counter);
  r.events_this_step = (int32_t)wave_lane0_u32((uint32_t)r.events_this_step);
  r.n_events = ((uint64_t)wave_lane0_u32((uint32_t)(r.n_events >> 32)) << 32) | wave_lane0_u32((uint32_t)r.n_events);
  r.wall_time = wave_lane0_f64(r.wall_time);
  r.next_arrival_t = wave_lane0_f64(r.next_arrival_t);
  r.curr_source = wave_lane0_u32(r.curr_source);
  if (r.curr_source != POOL_NONE) return 0;

  SssEvSlot sl = g_hot.ev[lane];
  bool live = lane < g_c.E && sl.t < __builtin_inf();
  // rank of this lane's event in pop order, and how many earlier events sit on the same stage
  int rank = 0, same_before = 0;
  for (int e = 0; e < g_c.E; e++) {
    double te = wave_readlane_f64(sl.t, e);
    uint32_t se = wave_readlane_u32(sl.seq, e), ie = wave_readlane_u32(sl.info, e);
    bool before = te < sl.t || (te == sl.t && se < sl.seq);
    rank += before ? 1 : 0;
    same_before += (before && ((ie ^ sl.info) >> 8) == 0) ? 1 : 0;
  }
  bool mine = live && rank < SSS_BATCH_MAX;
  // candidate test + everything the duration draw needs except the random numbers
  int j = info_job(sl.info), st_id = info_stage(sl.info);
  bool ok = mine && info_kind(sl.info) == EV_TASK_FINISHED && sl.t < r.next_arrival_t;
  SssStage* sp = nullptr;
  SssJob* jp = nullptr;
  float* dp = nullptr;
  int gs = 0, n_local = 0, li = 0, ri = 0, new_remaining = 0, demand = 0;
  if (ok) {
    sp = stgp(j, st_id), jp = jobp(j), dp = durp(j, st_id);
    SssStage stv = *sp;
    new_remaining = (int)stv.remaining - same_before - 1;
    demand = new_remaining - ((int)stv.moving_to + (int)stv.commit_to);
    gs = jp->gs_base + st_id;
    n_local = popc64(jp->local_mask);
    ok = new_remaining >= 0 && n_local > 0;
    executor_interval(n_local, li, ri);
  }
  bool need_random = ok && li != ri;
  // the batch is the longest prefix (in rank order) of ok events; lane_of[q] = lane holding rank q
  int lane_of[SSS_BATCH_MAX];
  uint32_t nr_bits = 0;
  int P = SSS_BATCH_MAX;
  _Pragma("unroll") for (int q = 0; q < SSS_BATCH_MAX; q++) {
    uint64_t who = wave_ballot(live && rank == q);
    uint64_t good = wave_ballot(ok && rank == q);
    uint64_t nr = wave_ballot(need_random && rank == q);
    lane_of[q] = who ? ctz64(who) : 0;
    if (good == 0 && q < P) P = q;
    if (nr) nr_bits |= 1u << q;
  }
  if (P == 0) return 0;
  // RNG outputs in event order (uniform): random() first when the level is interpolated, then one
  // 32-bit half for the bounded integer (low half of a fresh output, else the buffered high half)
  uint64_t my_r64 = 0;
  uint32_t my_u32 = 0;
  {
    EvRegs w = r;
    _Pragma("unroll") for (int q = 0; q < SSS_BATCH_MAX; q++) {
      if (q < P) {
        uint64_t o64 = 0;
        if (nr_bits & (1u << q)) o64 = rng_next64(w);
        uint32_t o32 = rng_next32(w);
        if (rank == q) my_r64 = o64, my_u32 = o32;
      }
    }
  }
  // each lane: level, descriptor, index, duration, new completion time
  double t_new = __builtin_inf();
  double dur = 0.0;
  bool fine = ok && rank < P;
  SssExDesc xd;
  bool xd_dirty = false;
  if (fine) {
    int ri_orig = ri;
    if (li != ri) {
      double left = (double)exec_level_value(li), right = (double)exec_level_value(ri);
      double u = (double)(my_r64 >> 11) * (1.0 / 9007199254740992.0);
      int rand_pt = 1 + (int)(u * (right - left));
      if (!((double)rand_pt <= (double)n_local - left)) li = ri;
    }
    xd = f.exdesc[lane];
    int which = li == ri_orig ? 0 : 1;
    int off, lenw;
    if (xd.gs == gs && xd.lvl[which] == li) {
      off = xd.off[which], lenw = xd.lenw[which];
    } else {
      const int2 d = *(const int2*)(f.eff + (((size_t)gs * 8 + li) * 3 + 1) * 2);
      off = d.x, lenw = d.y;
      if (xd.gs != gs) xd.lvl[which ^ 1] = -1;
      xd.gs = gs, xd.off[which] = off, xd.lenw[which] = lenw, xd.lvl[which] = (int8_t)li;
      xd_dirty = true;
    }
    uint32_t len = (uint32_t)(lenw & 0x3FFFFFFF);
    uint64_t m = (uint64_t)my_u32 * len;
    // len == 0 (the reference would raise), len == 1 (choice() draws nothing) and a Lemire
    // rejection candidate are left to the single-event path
    if (len <= 1 || (uint32_t)m < len) {
      fine = false;
    } else {
      dur = (double)f.durations[off + (int)(m >> 32)];
      t_new = sl.t + dur;
    }
  }
  // longest prefix whose members are all fine and whose rescheduled times do not overtake a later member
  {
    double min_new = __builtin_inf();
    int Pv = P;
    _Pragma("unroll") for (int q = 0; q < SSS_BATCH_MAX; q++) {
      if (q < P) {
        int l = lane_of[q];
        double tq = wave_readlane_f64(sl.t, l), nq = wave_readlane_f64(t_new, l);
        uint32_t fq = wave_readlane_u32(fine ? 1u : 0u, l);
        if (q < Pv && (!fq || min_new < tq)) Pv = q;
        min_new = nq < min_new ? nq : min_new;
      }
    }
    P = Pv;
  }
  if (P == 0) return 0;
#ifdef SSS_DEBUG_BATCH
  if (wave_env() == 0 && rank < 8 && live) printf("[batch] P=%d lane=%d rank=%d t=%.3f new=%.3f fine=%d info=%x same_before=%d newrem=%d li=%d u32=%u r64=%llu\n", P, lane, rank, sl.t, t_new, (int)fine, sl.info, same_before, new_remaining, li, my_u32, (unsigned long long)my_r64);
#endif
  // commit the first P events
  bool commit = fine && rank < P;
  if (commit) {
    lane_atomic_add_i32((int32_t*)sp, -1);  // remaining-- (low half of the record's first dword; never borrows)
    if (new_remaining == 0) lane_atomic_add_i32((int32_t*)&jp->supply, 1 << 16);  // sat_count++ (ENV:595-597)
    if (demand <= 0) lane_atomic_or_u64(&jp->sat_mask, bit64(st_id));
    SssEvSlot ns;
    ns.t = t_new, ns.seq = r.counter + (uint32_t)rank, ns.info = sl.info;
    g_hot.ev[lane] = ns;
    if (xd_dirty) f.exdesc[lane] = xd;
  }
  // most_recent_duration of a stage = the duration drawn by its LAST event in the batch: only the
  // highest-ranked committed lane of each stage stores
  bool last_on_stage = commit;
  _Pragma("unroll") for (int q = 0; q < SSS_BATCH_MAX; q++) {
    if (q < P) {
      uint32_t iq = wave_readlane_u32(sl.info, lane_of[q]);
      if (q > rank && ((iq ^ sl.info) >> 8) == 0) last_on_stage = false;
    }
  }
  if (last_on_stage) *dp = (float)dur;
  // uniform bookkeeping: RNG stream position, counters, wall time
  _Pragma("unroll") for (int q = 0; q < SSS_BATCH_MAX; q++) {
    if (q < P) {
      if (nr_bits & (1u << q)) (void)rng_next64(r);
      (void)rng_next32(r);
    }
  }
  r.counter += (uint32_t)P;
  r.n_events += (uint64_t)P;
  r.events_this_step += P;
  r.wall_time = wave_readlane_f64(sl.t, lane_of[P - 1]);
  n_fast += (uint64_t)P;
  return P;
}

