// sss_sim_tracker.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// ExecutorTracker restated: commitments, pool records, moving an executor between two pools (pool_pair_*).
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 3  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
