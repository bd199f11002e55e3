// sss_sim_pyset.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// CPython 3.10 set images (executor pools, the job-id set of the reward): add / remove / pop / resize / copy, lane-0 and wave-wide forms.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 2  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
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
