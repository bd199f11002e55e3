// sss_sim_fast_run.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// the fast run: consecutive "task finished, its stage has more tasks" events in registers.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 7  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
// HEAD_ONLY: which lanes read their candidate duration from the pool (see SSS_EXP_DUR below) - the fused rollout kernel's
// instantiation, where every wave is busy all the time and the memory pipeline is what a run waits for
template <bool HEAD_ONLY = false>
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
  // Which lanes read their candidate duration from the pool. Only the head's (rank 1: the ranks are up to date whenever a draw is
  // computed) is ever used; with every lane loading, an event of a 50-executor env touches ~50 lines scattered over the env's
  // duration lists - 2.3 GB of fetches per config-3 launch on the 60 MB "deep" pack (rocprofv3 FETCH_SIZE, profiles/r06_bench.md).
  // Letting the other lanes read entry 0 instead (one compare and one select: two lines per load instruction) puts the rank
  // update of the previous event's commit in front of the load's address - the load no longer issues as early as it can, which
  // a lock-step launch pays for (its slowest env's chain: -5 % at config 2, -11 % on the deep pack) and the fused rollout, where
  // all waves are busy and the memory pipeline is the resource, gains from (+26 % at config 3 sizing on the deep pack, +2-3 %
  // elsewhere): HEAD_ONLY is the rollout kernel's instantiation (profiles/r06_bench.md section 4: also the exec-mask form).
#ifdef SSS_EXP_NOLOAD  /* timing experiment only (wrong durations): what the load from the duration pool costs */
#define SSS_EXP_DUR(i) (int32_t)(((i) & 1023u) + 100u)
#elif defined(SSS_FAST_LOAD_PRED)
#define SSS_EXP_DUR(i) (rank == 1u ? *(const int32_t*)(dur_base + (size_t)((i) << 2)) : 0)
#elif defined(SSS_FAST_LOAD_SEL)
#define SSS_EXP_DUR(i) (*(const int32_t*)(dur_base + (size_t)((rank == 1u ? (i) : 0u) << 2)))
#elif defined(SSS_FAST_LOAD_ALL)
#define SSS_EXP_DUR(i) (*(const int32_t*)(dur_base + (size_t)((i) << 2)))
#else
#define SSS_EXP_DUR(i) (*(const int32_t*)(dur_base + (size_t)(((!HEAD_ONLY || rank == 1u) ? (i) : 0u) << 2)))
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
