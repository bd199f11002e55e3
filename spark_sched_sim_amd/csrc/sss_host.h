// sss_host.h - host side of the C ABI (include/sss.h): argument checking, layout, constant
// upload, kernel launches. The including translation unit supplies the five be_* primitives
// (spark_sched_sim_amd/csrc/sss_hip.hip: HIP runtime; tests/emu/emu_backend.cpp: the CPU wave
// emulator used by the test-suite); the simulator kernels' launchers come from sss_narrow.h / sss_wide.h.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/sss.h"
#include "sss_layout.h"
#include "sss_train.h"
#include "sss_rows.h"
#include "sss_narrow.h"
#include "sss_wide.h"

// envs with more than 64 executors run on the wide instantiation of the kernels (sss_wide.h)
static bool sss_is_wide(int num_executors) { return num_executors > 64; }
static int sss_hot_bytes(int num_executors) { return sss_is_wide(num_executors) ? sss_wide_hot_bytes() : sss_narrow_hot_bytes(); }

static thread_local std::string g_sss_err;
static int sss_fail(int code, const std::string& msg) {
  g_sss_err = msg;
  return code;
}

struct SssPackHost {
  int T, L, s_max, total_stages, total_edges, total_durations;
  int n_sizes, n_queries;  // len(QUERY_SIZES), NUM_QUERIES of the trace set (tpch.py:14-15); T = n_queries * n_sizes
  int64_t sec_off[12], sec_len[12];
  int max_edges_per_job;
};

static int sss_pack_parse(const uint8_t* p, size_t n, SssPackHost* v) {
  if (!p || n < 8 + 64 + 12 * 16 || memcmp(p, "SSSPACK2", 8) != 0) return -1;
  int64_t h[8];
  memcpy(h, p + 8, sizeof(h));
  v->T = (int)h[0], v->L = (int)h[1], v->s_max = (int)h[2], v->total_stages = (int)h[3];
  v->total_edges = (int)h[4], v->total_durations = (int)h[5];
  if (h[6] != 12) return -1;
  for (int i = 0; i < 12; i++) {
    int64_t e[2];
    memcpy(e, p + 8 + 64 + 16 * i, 16);
    if (e[0] < 0 || e[1] < 0 || (size_t)(e[0] + e[1]) > n || (e[0] & 7)) return -1;
    v->sec_off[i] = e[0], v->sec_len[i] = e[1];
  }
  // header word 7: input sizes per query; 0 = the reference's seven (tpch.py:14), which keeps packs written before the word
  // had a meaning - the frozen default among them, whose digest the fixtures record - byte for byte what they were
  v->n_sizes = h[7] > 0 ? (int)h[7] : 7;
  if (v->T < 1 || v->T % v->n_sizes || v->L < 1 || v->L > SSS_MAX_LEVELS || v->s_max < 1 || v->s_max > SSS_MAX_STAGES) return -1;
  v->n_queries = v->T / v->n_sizes;
  if ((size_t)v->sec_len[1] < sizeof(int32_t) * ((size_t)v->T + 1) || (size_t)v->sec_len[2] < sizeof(int32_t) * ((size_t)v->T + 1)) return -1;
  const int32_t* eo = (const int32_t*)(p + v->sec_off[2]);
  int me = 0;
  for (int t = 0; t < v->T; t++) me = eo[t + 1] - eo[t] > me ? eo[t + 1] - eo[t] : me;
  v->max_edges_per_job = me;
  if (me > 255) return -1;
  // Fields the pack carries narrower than the reference's Python objects (workload.py refuses what does not fit when it
  // builds a pack; a pack from anywhere else is looked at again here): a task count or duration that wrapped shows up
  // negative, list descriptors must stay inside the duration pool.
  if ((size_t)v->sec_len[3] < sizeof(int32_t) * (size_t)v->total_stages || (size_t)v->sec_len[11] < sizeof(int32_t) * (size_t)v->total_durations ||
      (size_t)v->sec_len[10] < sizeof(int32_t) * (size_t)v->total_stages * 3 * (size_t)v->L * 2)
    return -1;
  const int32_t* nt = (const int32_t*)(p + v->sec_off[3]);
  for (int i = 0; i < v->total_stages; i++)
    if (nt[i] < 0) return -2;
  const int32_t* du = (const int32_t*)(p + v->sec_off[11]);
  for (int i = 0; i < v->total_durations; i++)
    if (du[i] < 0) return -3;
  const int32_t* de = (const int32_t*)(p + v->sec_off[10]);
  for (int64_t i = 0; i < (int64_t)v->total_stages * 3 * v->L; i++) {
    const int64_t off = de[2 * i], len = de[2 * i + 1];
    if (len > 0 && (off < 0 || len >= (1 << 30) || off + len > v->total_durations)) return -4;
  }
  return 0;
}

// the longest path (in edges) of any template's stage DAG: what bounds the DAG layers (topological generations beyond the first,
// decima/utils.py:246-267) of any observation of this workload (workload.py: pack_max_depth)
static int sss_pack_max_depth(const uint8_t* p, const SssPackHost& v) {
  const int32_t* so = (const int32_t*)(p + v.sec_off[1]);
  const int32_t* eo = (const int32_t*)(p + v.sec_off[2]);
  const int32_t* ed = (const int32_t*)(p + v.sec_off[9]);
  int best = 0;
  for (int t = 0; t < v.T; t++) {
    const int n = so[t + 1] - so[t];
    if (n < 1 || n > SSS_MAX_STAGES) continue;
    int dist[SSS_MAX_STAGES] = {0};
    for (int it = 0; it < n; it++) {  // longest-path relaxation; a DAG of n stages settles within n sweeps
      bool moved = false;
      for (int e = eo[t]; e < eo[t + 1]; e++) {
        const int u = ed[2 * e], w = ed[2 * e + 1];
        if (u >= 0 && u < n && w >= 0 && w < n && dist[w] < dist[u] + 1) dist[w] = dist[u] + 1, moved = true;
      }
      if (!moved) break;
    }
    for (int i = 0; i < n; i++) best = dist[i] > best ? dist[i] : best;
  }
  return best;
}

struct sss_handle {
  sss_cfg cfg;
  SssPackHost ph;
  int max_dag_depth;
  SssLayout L;
  SssBuffers B;
  bool bound;
  int device;
  void* pack_dev;
  void* zig_dev;
  void* eff_dev;
  void* jump_dev;
  void* common_dev;
  SssParams P;
  SssPackDev pk;
};

static int sss_validate(const sss_cfg* cfg, const void* pack, size_t pack_bytes, int num_envs, SssPackHost* ph, int* J_cap) {
  if (!cfg) return sss_fail(-1, "cfg is NULL");
  if (int prc = sss_pack_parse((const uint8_t*)pack, pack_bytes, ph))
    return sss_fail(-2, prc == -2   ? "workload pack: a stage has a negative task count (a count beyond int32 wrapped when the pack was written)"
                        : prc == -3 ? "workload pack: negative task duration (a duration beyond int32 milliseconds wrapped when the pack was written, or the trace set holds one)"
                        : prc == -4 ? "workload pack: a duration list descriptor points outside the duration pool or holds 2^30 or more entries"
                                    : "workload pack is malformed (expected SSSPACK2, n_queries x n_sizes templates, <= 64 stages/job, <= 255 edges/job, <= 16 executor levels)");
  if (num_envs < 1) return sss_fail(-3, "num_envs must be >= 1");
  if (cfg->num_executors < 1 || cfg->num_executors > SSS_MAX_EXEC_ANY) return sss_fail(-4, "num_executors must be in [1, 128]");
  int cap = cfg->job_arrival_cap > 0 ? cfg->job_arrival_cap : 0;
  int jc = cfg->max_jobs > 0 ? cfg->max_jobs : cap;
  if (jc <= 0) return sss_fail(-5, "max_jobs is required when job_arrival_cap is None");
  if (cap > jc) return sss_fail(-5, "max_jobs is smaller than job_arrival_cap");
  if (jc > SSS_MAX_JOBS) return sss_fail(-5, "job capacity above 1024 is not supported by this build");
  if (!(cfg->job_arrival_rate > 0)) return sss_fail(-6, "job_arrival_rate must be > 0");
  if (!(cfg->beta >= 0)) return sss_fail(-6, "beta must be >= 0");
  *J_cap = jc;
  return 0;
}

// Resolves TPCHDataSampler.task_duration's key substitution and fallback chain (reference
// data_samplers/tpch.py:88-106, 231-233) per (stage, executor level, executor mode):
//   mode 0  executor idle:        fresh_durations, else first_wave + warmup_delay
//   mode 1  same stage as before: rest_wave, else first_wave, else fresh_durations
//   mode 2  new to the stage:     first_wave, else fresh_durations
// A wave "fails" when the level key is missing (KeyError) or its list is empty (ValueError from
// Generator.choice before any draw); len 0 in the result = the exception would escape.
// Each entry also carries a lower bound of any duration the list can yield (its minimum, plus the
// whole milliseconds of warmup_delay where that is added): the batched event paths (sss_sim.h,
// batch_released_events / batch_arrival_events) use it to bound the time of events that do not exist yet.
static std::vector<int32_t> sss_build_eff(const uint8_t* pack, const SssPackHost& ph, const int8_t lvl_of[8], double warmup_delay) {
  const uint32_t* keymask = (const uint32_t*)(pack + ph.sec_off[7]);
  const int32_t* maxlvl = (const int32_t*)(pack + ph.sec_off[8]);
  const int32_t* desc = (const int32_t*)(pack + ph.sec_off[10]);
  const int32_t* durations = (const int32_t*)(pack + ph.sec_off[11]);
  int warm_floor = warmup_delay > 0 ? (warmup_delay < 1e9 ? (int)floor(warmup_delay) : 1000000000) : 0;
  // rows 0 .. 8 * total_stages - 1: the eight executor levels; the tail (SssPackDev::eff0): "level" 8 = a key that is in no
  // first_wave dict - num_local_executors == exec_cap > 100 yields key 0 (tpch.py:244, 258-260) - hence always the largest level
  std::vector<int32_t> eff((size_t)ph.total_stages * 9 * 3 * 4, 0);
  for (int gs = 0; gs < ph.total_stages; gs++)
    for (int i = 0; i < 9; i++) {
      int lvl = i < 8 ? lvl_of[i] : -1;
      if (lvl < 0 || !((keymask[gs] >> lvl) & 1)) lvl = maxlvl[gs];
      auto d = [&](int wave, int k) { return desc[((gs * 3 + wave) * ph.L + lvl) * 2 + k]; };
      static const int chain[3][3] = {{0, 1, -1}, {2, 1, 0}, {1, 0, -1}};
      for (int mode = 0; mode < 3; mode++) {
        int32_t* out = i < 8 ? &eff[(((size_t)gs * 8 + i) * 3 + mode) * 4] : &eff[(size_t)ph.total_stages * 8 * 3 * 4 + ((size_t)gs * 3 + mode) * 4];
        for (int k = 0; k < 3; k++) {
          int wave = chain[mode][k];
          if (wave < 0) break;
          if (d(wave, 1) > 0) {
            bool warm = mode == 0 && k == 1;
            out[0] = d(wave, 0);
            out[1] = d(wave, 1) | (warm ? (1 << 30) : 0);
            int mn = durations[d(wave, 0)];
            for (int q = 1; q < d(wave, 1); q++) mn = durations[d(wave, 0) + q] < mn ? durations[d(wave, 0) + q] : mn;
            out[2] = (mn < 0 ? 0 : mn) + (warm ? warm_floor : 0);
            break;
          }
        }
      }
    }
  // fourth word of the idle-executor entries: the bound over ALL executor levels of the stage - for an executor
  // whose job's executor count is not known yet when the bound is needed (batch_arrival_events)
  for (int gs = 0; gs < ph.total_stages; gs++) {
    int32_t lb = 0x7FFFFFFF;
    for (int i = 0; i < 8; i++) {
      const int32_t* e = &eff[(((size_t)gs * 8 + i) * 3 + 0) * 4];
      if ((e[1] & 0x3FFFFFFF) > 0 && e[2] < lb) lb = e[2];
    }
    if (lb == 0x7FFFFFFF) lb = 0;
    for (int i = 0; i < 8; i++) eff[(((size_t)gs * 8 + i) * 3 + 0) * 4 + 3] = lb;
  }
  return eff;
}

static void sss_fill_dims(const SssLayout& L, sss_dims* d) {
  memset(d, 0, sizeof(*d));
  d->num_envs = L.num_envs, d->num_executors = L.E, d->job_cap = L.J_cap, d->stage_stride = L.SP;
  d->node_cap = L.n_cap, d->edge_cap = L.ed_cap, d->obs_i32 = SSS_OBS_I32, d->obs_f64 = SSS_OBS_F64;
  d->state_bytes = L.state_bytes, d->env_stride = L.env_stride;
  d->off_t_arrival = L.off_t_arrival, d->off_t_completed = L.off_t_completed, d->off_jobs = L.off_jobs;
  d->off_active = L.off_active, d->off_dur_ring = L.off_dur_ring;
  d->job_rec_bytes = (int32_t)sizeof(SssJob), d->hdr_bytes = (int32_t)sizeof(SssHdr);
}

extern "C" int sss_query_dims(const sss_cfg* cfg, const void* pack, size_t pack_bytes, int num_envs, sss_dims* out) {
  SssPackHost ph;
  int J_cap;
  if (int rc = sss_validate(cfg, pack, pack_bytes, num_envs, &ph, &J_cap)) return rc;
  if (!out) return sss_fail(-1, "out is NULL");
  SssLayout L;
  sss_compute_layout(&L, num_envs, cfg->num_executors, J_cap, ph.s_max, ph.L, ph.max_edges_per_job, sss_hot_bytes(cfg->num_executors));
  sss_fill_dims(L, out);
  return 0;
}

// PCG64 jump-ahead table (numpy's pcg64: 128-bit LCG, multiplier 0x2360ED051FC65DA44385DF649FCCF645):
// state_{n+k} = A_k * state_n + C_k * inc (mod 2^128) for k = -64..64, rows (A_hi, A_lo, C_hi, C_lo).
// With it the 64 lanes of a wave produce the next 64 raw outputs of an env's stream at once
// (sss_sim.h, rng_refill) - the stream itself is exactly numpy's.
static std::vector<uint64_t> sss_build_pcg_jump() {
  typedef unsigned __int128 u128;
  const u128 a = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
  u128 ainv = a;  // a * a == 1 (mod 8); each Newton step doubles the number of correct low bits
  for (int i = 0; i < 7; i++) ainv *= 2 - a * ainv;
  std::vector<uint64_t> t(129 * 4);
  auto put = [&](int k, u128 A, u128 C) {
    uint64_t* r = &t[(size_t)(k + 64) * 4];
    r[0] = (uint64_t)(A >> 64), r[1] = (uint64_t)A, r[2] = (uint64_t)(C >> 64), r[3] = (uint64_t)C;
  };
  u128 A = 1, C = 0;
  put(0, A, C);
  for (int k = 1; k <= 64; k++) A = a * A, C = a * C + 1, put(k, A, C);
  A = 1, C = 0;
  for (int k = 1; k <= 64; k++) A = ainv * A, C = ainv * (C - 1), put(-k, A, C);  // state_{n-1} = ainv * (state_n - inc)
  return t;
}

// The executor-level draw of TPCHDataSampler (tpch.py:222-229) as a threshold on the raw generator output.
// With n local executors strictly between two levels (left, right) the reference computes
//   rand_pt = 1 + int(rng.random() * (right - left));  level = left if rand_pt <= n - left else right
// where rng.random() = (x >> 11) * 2^-53 for the raw 64-bit output x. The outcome is monotone in m = x >> 11
// (a product of non-negative doubles, then truncation), so there is one threshold T per n with "right" chosen
// exactly when m >= T. It is found by bisection over the SAME double arithmetic (compiled with
// -ffp-contract=off like the device code), not derived: entry n holds T, or 2^53 (never) when the interval is closed.
static std::vector<uint64_t> sss_build_lvl_thr() {
  static const int lv[8] = {5, 10, 20, 40, 50, 60, 80, 100};
  std::vector<uint64_t> t(101, 1ull << 53);
  for (int n = 1; n <= 100; n++) {
    int ri = (n > 5) + (n > 10) + (n > 20) + (n > 40) + (n > 50) + (n > 60) + (n > 80);
    int li = (n <= 5 || n == lv[ri]) ? ri : ri - 1;
    if (li == ri) continue;
    const double left = (double)lv[li], right = (double)lv[ri];
    auto picks_right = [&](uint64_t m) {
      volatile double u = (double)m * (1.0 / 9007199254740992.0);
      volatile double prod = u * (right - left);
      int rand_pt = 1 + (int)prod;
      return !((double)rand_pt <= (double)n - left);
    };
    uint64_t lo = 0, hi = 1ull << 53;  // picks_right(lo) is false (rand_pt = 1 <= n - left), "hi" stands for true
    while (hi - lo > 1) {
      uint64_t mid = lo + (hi - lo) / 2;
      if (picks_right(mid)) hi = mid; else lo = mid;
    }
    t[n] = hi;
  }
  return t;
}

// The common executor pool right after reset is set(range(E)) (executor_tracker.py:41): the image CPython 3.10
// builds by adding 0, 1, .., E-1 to an empty set (set_add_entry / set_table_resize for keys with hash(k) == k:
// LINEAR_PROBES 9, PERTURB_SHIFT 5, resize at fill * 5 >= mask * 3 to the first power of two > 4 * used).
// No removals are involved, so there are no dummies. Layout: the 16-byte pool record (mask | fill << 16,
// used, 8 inline slots), then the table when it has more than 8 slots. Slot bytes: 0 = empty, key + 2.
static std::vector<uint8_t> sss_build_common_pool(int E) {
  std::vector<uint8_t> tab(8, 0);
  uint32_t mask = 7, fill = 0;
  auto insert_clean = [](std::vector<uint8_t>& t, uint32_t m, uint32_t key) {
    uint32_t perturb = key, i = key & m;
    for (;;) {
      uint32_t probes = (i + 9 <= m) ? 9 : 0;
      for (uint32_t p = 0; p <= probes; p++)
        if (t[i + p] == 0) {
          t[i + p] = (uint8_t)(key + 2);
          return;
        }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & m;
    }
  };
  for (uint32_t key = 0; key < (uint32_t)E; key++) {
    insert_clean(tab, mask, key);  // keys are distinct and nothing was removed: an add is a clean insert
    fill++;
    if (fill * 5 >= mask * 3) {
      uint32_t newsize = 8;
      while (newsize <= fill * 4) newsize <<= 1;
      std::vector<uint8_t> nt(newsize, 0);
      for (uint32_t i = 0; i <= mask; i++)
        if (tab[i] >= 2) insert_clean(nt, newsize - 1, tab[i] - 2u);
      tab.swap(nt), mask = newsize - 1;
    }
  }
  std::vector<uint8_t> out(16 + (mask == 7 ? 0 : mask + 1), 0);
  uint32_t w0 = mask | (fill << 16), w1 = fill;  // used == fill; no outgoing commitments
  memcpy(&out[0], &w0, 4), memcpy(&out[4], &w1, 4);
  if (mask == 7)
    memcpy(&out[8], tab.data(), 8);
  else
    memcpy(&out[16], tab.data(), mask + 1);
  return out;
}

extern "C" const char* sss_last_error(void) { return g_sss_err.c_str(); }

extern "C" int sss_abi_sizeof(const char* name) {
  if (!name) return -1;
#define SSS_ABI_SIZE(T) \
  if (!strcmp(name, #T)) return (int)sizeof(T);
  SSS_ABI_SIZE(sss_cfg) SSS_ABI_SIZE(sss_dims) SSS_ABI_SIZE(sss_buffers) SSS_ABI_SIZE(sss_decima_graph) SSS_ABI_SIZE(sss_decima_lists)
  SSS_ABI_SIZE(sss_bit_list_args) SSS_ABI_SIZE(sss_gnn_args) SSS_ABI_SIZE(sss_decima_policy_args) SSS_ABI_SIZE(sss_decima_sample_args)
  SSS_ABI_SIZE(sss_gnn_encode_args) SSS_ABI_SIZE(sss_collect_args) SSS_ABI_SIZE(sss_mlp_args) SSS_ABI_SIZE(sss_arena_array)
  SSS_ABI_SIZE(sss_arena_args) SSS_ABI_SIZE(sss_returns_args) SSS_ABI_SIZE(sss_baseline_args) SSS_ABI_SIZE(sss_rows_args) SSS_ABI_SIZE(sss_concat_part) SSS_ABI_SIZE(sss_concat_args) SSS_ABI_SIZE(sss_segcat_args)
#undef SSS_ABI_SIZE
  return -1;
}

extern "C" int sss_create(const sss_cfg* cfg, const void* pack, size_t pack_bytes, int num_envs, int device, sss_handle** out) {
  SssPackHost ph;
  int J_cap;
  if (int rc = sss_validate(cfg, pack, pack_bytes, num_envs, &ph, &J_cap)) return rc;
  if (!out) return sss_fail(-1, "out is NULL");
  BeDeviceGuard guard(device);  // allocations and uploads on the env's device; the caller's current device is restored
  if (int rc = be_set_device(device)) return sss_fail(-10, "cannot select device " + std::to_string(device) + ": " + be_error(rc));
  sss_handle* h = new sss_handle();
  h->cfg = *cfg, h->ph = ph, h->device = device, h->bound = false;
  h->max_dag_depth = sss_pack_max_depth((const uint8_t*)pack, ph);
  sss_compute_layout(&h->L, num_envs, cfg->num_executors, J_cap, ph.s_max, ph.L, ph.max_edges_per_job, sss_hot_bytes(cfg->num_executors));
  memset(&h->B, 0, sizeof(h->B));

  // pack + ziggurat tables -> device
  h->pack_dev = be_alloc(pack_bytes);
  std::vector<uint8_t> zig(256 * 8 * 3);
  memcpy(zig.data(), ZIG_KE, 2048), memcpy(zig.data() + 2048, ZIG_WE, 2048), memcpy(zig.data() + 4096, ZIG_FE, 2048);
  h->zig_dev = be_alloc(zig.size());
  if (!h->pack_dev || !h->zig_dev) {
    sss_destroy(h);
    return sss_fail(-11, "device allocation failed");
  }
  if (be_h2d(h->pack_dev, pack, pack_bytes) || be_h2d(h->zig_dev, zig.data(), zig.size())) {
    sss_destroy(h);
    return sss_fail(-13, "copying the workload pack to the device failed");
  }

  SssPackDev pk;
  memset(&pk, 0, sizeof(pk));
  pk.n_queries = ph.n_queries, pk.n_sizes = ph.n_sizes;
  pk.T = ph.T, pk.L = ph.L, pk.s_max = ph.s_max, pk.total_stages = ph.total_stages, pk.total_edges = ph.total_edges;
  pk.total_durations = ph.total_durations;
  const uint8_t* b = (const uint8_t*)h->pack_dev;
  pk.levels = (const int32_t*)(b + ph.sec_off[0]);
  pk.tmpl_stage_off = (const int32_t*)(b + ph.sec_off[1]);
  pk.tmpl_edge_off = (const int32_t*)(b + ph.sec_off[2]);
  pk.stage_num_tasks = (const int32_t*)(b + ph.sec_off[3]);
  pk.stage_rough = (const double*)(b + ph.sec_off[4]);
  pk.stage_parent_mask = (const uint64_t*)(b + ph.sec_off[5]);
  pk.stage_child_mask = (const uint64_t*)(b + ph.sec_off[6]);
  pk.stage_first_keymask = (const uint32_t*)(b + ph.sec_off[7]);
  pk.stage_max_first_lvl = (const int32_t*)(b + ph.sec_off[8]);
  pk.edges = (const int32_t*)(b + ph.sec_off[9]);
  pk.desc = (const int32_t*)(b + ph.sec_off[10]);
  pk.durations = (const int32_t*)(b + ph.sec_off[11]);
  pk.zig_ke = (const uint64_t*)h->zig_dev;
  pk.zig_we = (const double*)((const uint8_t*)h->zig_dev + 2048);
  pk.zig_fe = (const double*)((const uint8_t*)h->zig_dev + 4096);

  SssParams& P = h->P;
  memset(&P, 0, sizeof(P));
  P.cap_cfg = cfg->job_arrival_cap > 0 ? cfg->job_arrival_cap : 0;
  P.mean_interarrival = 1 / cfg->job_arrival_rate;  // tpch.py:42
  P.moving_delay = cfg->moving_delay, P.warmup_delay = cfg->warmup_delay, P.beta = cfg->beta;
  static const int exec_levels[8] = {5, 10, 20, 40, 50, 60, 80, 100};  // tpch.py:238
  const int32_t* levels = (const int32_t*)((const uint8_t*)pack + ph.sec_off[0]);
  for (int i = 0; i < 8; i++) {
    P.lvl_of[i] = -1;
    for (int l = 0; l < ph.L; l++)
      if (levels[l] == exec_levels[i]) P.lvl_of[i] = (int8_t)l;
  }
  P.max_edges = ph.max_edges_per_job;
  if (sss_compute_lds_pool(&P, h->L.J_cap, h->L.SP, h->L.E, sss_is_wide(h->L.E) ? sss_wide_static_lds_bytes() : sss_narrow_static_lds_bytes())) {
    sss_destroy(h);
    return sss_fail(-12, "LDS working set does not fit");
  }
  std::vector<int32_t> eff = sss_build_eff((const uint8_t*)pack, ph, P.lvl_of, cfg->warmup_delay);
  h->eff_dev = be_alloc(eff.size() * sizeof(int32_t));
  if (!h->eff_dev) {
    sss_destroy(h);
    return sss_fail(-11, "device allocation failed");
  }
  std::vector<uint64_t> jump = sss_build_pcg_jump();
  const size_t jump_words = jump.size();
  {
    std::vector<uint64_t> thr = sss_build_lvl_thr();  // rides in the same allocation
    jump.insert(jump.end(), thr.begin(), thr.end());
  }
  h->jump_dev = be_alloc(jump.size() * sizeof(uint64_t));
  if (!h->jump_dev) {
    sss_destroy(h);
    return sss_fail(-11, "device allocation failed");
  }
  if (be_h2d(h->eff_dev, eff.data(), eff.size() * sizeof(int32_t)) || be_h2d(h->jump_dev, jump.data(), jump.size() * sizeof(uint64_t))) {
    sss_destroy(h);
    return sss_fail(-13, "copying the duration descriptors to the device failed");
  }
  std::vector<uint8_t> common = sss_build_common_pool(cfg->num_executors);
  h->common_dev = be_alloc(common.size());
  if (!h->common_dev || be_h2d(h->common_dev, common.data(), common.size())) {
    sss_destroy(h);
    return sss_fail(-11, "device allocation failed");
  }
  pk.eff = (const int32_t*)h->eff_dev;
  pk.eff0 = pk.eff + (size_t)ph.total_stages * 8 * 3 * 4;
  pk.pcg_jump = (const uint64_t*)h->jump_dev;
  pk.lvl_thr = (const uint64_t*)h->jump_dev + jump_words;
  pk.common_pool = (const uint8_t*)h->common_dev;
  h->pk = pk;
  *out = h;
  return 0;
}

extern "C" int sss_bind_buffers(sss_handle* h, const sss_buffers* b) {
  if (!h || !b) return sss_fail(-1, "NULL argument");
  if (!b->state_dev || !b->nodes_dev || !b->edge_links_dev || !b->dag_ptr_dev || !b->exec_supplies_dev || !b->obs_i32_dev || !b->obs_f64_dev)
    return sss_fail(-20, "every buffer in sss_buffers must be provided");
  if (((uintptr_t)b->state_dev) & 255) return sss_fail(-21, "state_dev must be 256-byte aligned");
  h->B.state = b->state_dev, h->B.nodes = b->nodes_dev, h->B.edge_links = b->edge_links_dev, h->B.dag_ptr = b->dag_ptr_dev;
  h->B.exec_supplies = b->exec_supplies_dev, h->B.obs_i32 = b->obs_i32_dev, h->B.obs_f64 = b->obs_f64_dev;
  h->B.gen = h->bound ? h->B.gen + 1 : 1, h->B.gen_pad = 0;
  h->bound = true;
  return 0;
}

static SssKernelArgs sss_args(const sss_handle* h) {
  SssKernelArgs a;
  a.L = h->L, a.B = h->B, a.P = h->P, a.pk = h->pk;
  return a;
}

extern "C" int sss_reset(sss_handle* h, const uint64_t* seeds_dev, const double* time_limits_dev, const uint8_t* mask_dev, void* stream) {
  if (!h || !seeds_dev) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  BeDeviceGuard guard(h->device);
  const SssKernelArgs ka = sss_args(h);
  if (int rc = sss_is_wide(h->L.E) ? sss_wide_launch_reset(ka, h->L.num_envs, seeds_dev, time_limits_dev, mask_dev, stream)
                                   : be_launch_reset(ka, h->L.num_envs, seeds_dev, time_limits_dev, mask_dev, stream)) return sss_fail(-30, std::string("reset launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_step(sss_handle* h, const int32_t* stage_idx_dev, const int32_t* num_exec_dev, int auto_reset, uint64_t seed_stride, void* stream) {
  if (!h || !stage_idx_dev || !num_exec_dev) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  BeDeviceGuard guard(h->device);
  const SssKernelArgs ka = sss_args(h);
  if (int rc = sss_is_wide(h->L.E) ? sss_wide_launch_step(ka, h->L.num_envs, stage_idx_dev, num_exec_dev, auto_reset, seed_stride, stream)
                                   : be_launch_step(ka, h->L.num_envs, stage_idx_dev, num_exec_dev, auto_reset, seed_stride, stream)) return sss_fail(-30, std::string("step launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_step_bounded(sss_handle* h, const int32_t* stage_idx_dev, const int32_t* num_exec_dev, int auto_reset, uint64_t seed_stride, int max_events,
                                uint8_t* ready_dev, void* stream) {
  if (!h || !stage_idx_dev || !num_exec_dev || !ready_dev) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  if (max_events < 1) return sss_fail(-35, "sss_step_bounded: max_events must be at least 1");
  BeDeviceGuard guard(h->device);
  const SssKernelArgs ka = sss_args(h);
  if (int rc = sss_is_wide(h->L.E) ? sss_wide_launch_step_bounded(ka, h->L.num_envs, stage_idx_dev, num_exec_dev, auto_reset, seed_stride, max_events, ready_dev, stream)
                                   : be_launch_step_bounded(ka, h->L.num_envs, stage_idx_dev, num_exec_dev, auto_reset, seed_stride, max_events, ready_dev, stream))
    return sss_fail(-30, std::string("step launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_policy(sss_handle* h, int policy, int param, int32_t* stage_idx_dev, int32_t* num_exec_dev, void* stream) {
  if (!h || !stage_idx_dev || !num_exec_dev) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  BeDeviceGuard guard(h->device);
  if (policy < 0 || policy > 2) return sss_fail(-23, "unknown policy");
  const SssKernelArgs ka = sss_args(h);
  if (int rc = sss_is_wide(h->L.E) ? sss_wide_launch_policy(ka, h->L.num_envs, policy, param, stage_idx_dev, num_exec_dev, stream)
                                   : be_launch_policy(ka, h->L.num_envs, policy, param, stage_idx_dev, num_exec_dev, stream)) return sss_fail(-30, std::string("policy launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_rollout(sss_handle* h, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void* stream) {
  if (!h) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  BeDeviceGuard guard(h->device);
  if (policy < 0 || policy > 2) return sss_fail(-23, "unknown policy");
  if (n_steps < 0) return sss_fail(-24, "n_steps must be >= 0");
  const SssKernelArgs ka = sss_args(h);
  if (int rc = sss_is_wide(h->L.E) ? sss_wide_launch_rollout(ka, h->L.num_envs, policy, param, n_steps, auto_reset, seed_stride, stream)
                                   : be_launch_rollout(ka, h->L.num_envs, policy, param, n_steps, auto_reset, seed_stride, stream)) return sss_fail(-30, std::string("rollout launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_decima_graph_build(sss_handle* h, const sss_decima_graph* g, void* stream) {
  if (!h || !g) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  BeDeviceGuard guard(h->device);
  if (!g->node_off_dev || !g->job_off_dev || !g->edge_off_dev || !g->x_dev || !g->node_obs_dev || !g->node_loc_dev || !g->node_job_dev ||
      !g->sched_rank_dev || !g->gen_dev || !g->node_recv_dev || !g->stage_mask_dev || !g->src_dev || !g->dst_dev || !g->edge_obs_dev ||
      !g->edge_layers_dev || !g->job_obs_dev || !g->job_cap_dev || !g->job_first_dev || !g->obs_depth_dev ||
      !g->job_nodes_dev || !g->out_start_dev || !g->out_deg_dev || !g->layer_cnt_dev)
    return sss_fail(-1, "NULL argument");
  if ((int64_t)8 * h->L.n_cap + (int64_t)8 * (h->L.J_cap + 1) > 65536 || h->L.n_cap > 65535 || h->L.ed_cap > 65535 || h->max_dag_depth > 24)
    return sss_fail(-25, "node / edge capacity too large for the Decima graph kernel's LDS working set, or a job template deeper than 24 DAG layers");
  SssDecimaArgs d;
  d.active = g->active_dev, d.node_off = g->node_off_dev, d.job_off = g->job_off_dev, d.edge_off = g->edge_off_dev;
  d.num_tasks_scale = g->num_tasks_scale, d.work_scale = g->work_scale;
  d.x = g->x_dev, d.node_obs = g->node_obs_dev, d.node_loc = g->node_loc_dev, d.node_job = g->node_job_dev, d.sched_rank = g->sched_rank_dev;
  d.gen = g->gen_dev, d.node_recv = g->node_recv_dev, d.stage_mask = g->stage_mask_dev;
  d.src = g->src_dev, d.dst = g->dst_dev, d.edge_obs = g->edge_obs_dev, d.edge_layers = g->edge_layers_dev;
  d.job_obs = g->job_obs_dev, d.job_cap = g->job_cap_dev, d.job_first = g->job_first_dev, d.obs_depth = g->obs_depth_dev;
  d.job_nodes = g->job_nodes_dev, d.out_start = g->out_start_dev, d.out_deg = g->out_deg_dev, d.layer_cnt = g->layer_cnt_dev;
  d.sched_off = g->sched_off_dev, d.sched_list = g->sched_off_dev ? g->sched_list_dev : nullptr;
  d.layer_totals = g->layer_totals_dev, d.recv_lists = g->layer_totals_dev ? g->recv_lists_dev : nullptr, d.recv_stride = g->recv_stride;
  d.layer_totals_clear = g->layer_totals_clear_dev;
  if (g->layer_totals_clear_dev && g->layer_totals_clear_dev == g->layer_totals_dev) return sss_fail(-1, "layer_totals_clear_dev must be another set of counters than layer_totals_dev");
  if ((g->layer_totals_dev || g->layer_totals_clear_dev) && g->layer_totals_len != 33 * SSS_LIST_SETS)
    return sss_fail(-1, "layer_totals_len must be 33 * 32: the list counters are i64[33][32] (a binding built against the i64[32] layout?)");
  if (g->layer_totals_dev && (!g->recv_lists_dev || g->recv_stride < 1)) return sss_fail(-1, "NULL argument");
  if (g->sched_off_dev && !g->sched_list_dev) return sss_fail(-1, "NULL argument");
  if (int rc = be_launch_decima(h->L, h->B, h->cfg.num_executors, d, stream)) return sss_fail(-30, std::string("decima graph launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_decima_layer_lists(int num_envs, const sss_decima_lists* g, void* stream) {
  if (!g || !g->node_off_dev || !g->obs_nodes_dev || !g->node_recv_dev || !g->env_off_dev || !g->recv_dev) return sss_fail(-1, "NULL argument");
  if (num_envs < 1 || g->n_layers < 0 || g->n_layers > 32) return sss_fail(-27, "bad layer count");
  SssDecimaListArgs d;
  d.node_off = g->node_off_dev, d.obs_nodes = g->obs_nodes_dev, d.node_recv = g->node_recv_dev, d.env_off = g->env_off_dev;
  for (int l = 0; l < 32; l++) d.layer_base[l] = g->layer_base[l];
  d.recv = g->recv_dev, d.n_layers = g->n_layers, d.totals = nullptr;
  if (int rc = be_launch_decima_lists(num_envs, d, stream)) return sss_fail(-30, std::string("decima lists launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_decima_policy(sss_handle* h, const sss_decima_policy_args* g, void* stream) {
  if (!h || !g) return sss_fail(-1, "NULL argument");
  if (!h->bound) return sss_fail(-22, "sss_bind_buffers has not been called");
  BeDeviceGuard guard(h->device);
  if (!g->w_prep_dev || !g->w_msg_dev || !g->w_upd_dev || !g->w_dag_dev || !g->w_glob_dev || !g->w_stage_dev || !g->w_exec_dev ||
      !g->node_scratch_dev || !g->job_scratch_dev || !g->stage_idx_dev || !g->num_exec_dev || !g->stage_sel_dev || !g->job_idx_dev ||
      !g->exec_sel_dev || !g->lgprob_dev)
    return sss_fail(-1, "NULL argument");
  if ((int64_t)20 * h->L.n_cap + 64 > 65536 || h->L.n_cap > 65535) return sss_fail(-25, "node capacity too large for the Decima policy kernel's LDS working set");
  if (h->cfg.num_executors > 128) return sss_fail(-28, "num_executors must be <= 128");
  SssDecimaPolicyArgs d;
  d.active = g->active_dev, d.num_tasks_scale = g->num_tasks_scale, d.work_scale = g->work_scale, d.slope = g->slope;
  d.w_prep = g->w_prep_dev, d.w_msg = g->w_msg_dev, d.w_upd = g->w_upd_dev, d.w_dag = g->w_dag_dev, d.w_glob = g->w_glob_dev;
  d.w_stage = g->w_stage_dev, d.w_exec = g->w_exec_dev, d.node_scratch = g->node_scratch_dev, d.job_scratch = g->job_scratch_dev;
  d.rng_seed = g->rng_seed, d.rng_counter = g->rng_counter, d.stage_idx = g->stage_idx_dev, d.num_exec = g->num_exec_dev;
  d.stage_sel = g->stage_sel_dev, d.job_idx = g->job_idx_dev, d.exec_sel = g->exec_sel_dev, d.lgprob = g->lgprob_dev;
  d.stage_scores = g->stage_scores_dev, d.exec_scores = g->exec_scores_dev, d.prof = g->prof_dev;
  if (int rc = be_launch_decima_policy(h->L, h->B, h->cfg.num_executors, d, stream)) return sss_fail(-30, std::string("decima policy launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_decima_sample(int n_obs, int which, const sss_decima_sample_args* g, void* stream) {
  if (!g || n_obs < 1 || (which != 0 && which != 1)) return sss_fail(-1, "bad argument");
  if (g->num_executors < 1) return sss_fail(-28, "num_executors must be >= 1");
  SssDecimaSampleArgs d;
  d.n_pad = g->n_pad, d.E = g->num_executors, d.rng_seed = g->rng_seed, d.rng_counter = g->rng_counter;
  d.stage_scores = g->stage_scores_dev, d.exec_scores = g->exec_scores_dev, d.obs_nodes = g->obs_nodes_dev, d.obs_node_off = g->obs_node_off_dev;
  d.obs_job_off = g->obs_job_off_dev, d.sched_rank = g->sched_rank_dev, d.node_job = g->node_job_dev, d.job_gid = g->job_gid_dev;
  d.stage_idx = g->stage_idx_dev, d.num_exec = g->num_exec_dev, d.stage_sel = g->stage_sel_dev, d.job_idx = g->job_idx_dev;
  d.exec_sel = g->exec_sel_dev, d.lgprob = g->lgprob_dev, d.any_stage = g->any_stage_dev;
  if (int rc = be_launch_decima_sample(n_obs, which, d, stream)) return sss_fail(-30, std::string("decima sample launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_prefix_rows(const int32_t* src_dev, int64_t src_row_stride, int64_t src_col_stride, const uint8_t* mask_dev, int n_rows, int n_cols,
                               int64_t* off_dev, int64_t* cnt_dev, int64_t* totals_dev, void* stream) {
  if (!src_dev || !off_dev || !totals_dev) return sss_fail(-1, "NULL argument");
  if (n_rows < 1 || n_cols < 1) return sss_fail(-1, "empty scan");
  SssPrefixArgs a;
  a.src = src_dev, a.row_stride = src_row_stride, a.col_stride = src_col_stride, a.mask = mask_dev, a.n_rows = n_rows, a.n_cols = n_cols;
  a.off = off_dev, a.cnt = cnt_dev, a.totals = totals_dev;
  if (int rc = be_launch_prefix_rows(a, stream)) return sss_fail(-30, std::string("prefix launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_gnn_launch(int kind, const sss_gnn_args* g, void* stream) {
  if (!g) return sss_fail(-1, "NULL argument");
  if (kind < 0 || kind >= GNN_KINDS) return sss_fail(-26, "unknown GNN stage");
  if (g->n_rows < 0 || (!g->w_dev && kind != GNN_COMMIT && kind != GNN_MERGE)) return sss_fail(-1, "NULL argument");
  SssGnnArgs a;
  a.n_rows = g->n_rows, a.w = g->w_dev, a.w2 = g->w2_dev, a.slope = g->slope, a.E = g->num_executors, a.layer = g->layer, a.n_pad = g->n_pad;
  a.x = g->x_dev, a.h_init = g->h_init_dev, a.h = g->h_dev, a.tmp = g->tmp_dev, a.h_dag = g->h_dag_dev, a.h_glob = g->h_glob_dev, a.out = g->out_dev;
  a.out_deg = g->out_deg_dev, a.obs_depth = g->obs_depth_dev, a.idx0 = g->idx0_dev, a.dst = g->dst_dev, a.out_start = g->out_start_dev;
  a.edge_layers = g->edge_layers_dev, a.node_job = g->node_job_dev, a.node_obs = g->node_obs_dev, a.node_loc = g->node_loc_dev;
  a.job_obs = g->job_obs_dev, a.job_first = g->job_first_dev, a.job_cap = g->job_cap_dev, a.job_nodes = g->job_nodes_dev;
  a.obs_job_off = g->obs_job_off_dev, a.obs_jobs = g->obs_jobs_dev;
  a.w16 = g->w16_dev, a.w2_16 = g->w2_16_dev, a.node_recv = g->node_recv_dev, a.layer_totals = nullptr, a.idx0_stride = 0;
  a.seg_off = nullptr, a.n_seg = 0, a.list_q = 0;
  a.n_rows_dev = kind == GNN_LAYER ? nullptr : g->n_rows_dev;
  if (a.n_rows_dev && a.n_rows < 1) a.n_rows = 1;  // (the count is on the device: n_rows is a grid-size guess)
  if (kind == GNN_MERGE && !g->node_recv_dev) return sss_fail(-1, "NULL argument");
  if (kind == GNN_LAYER && !g->w2_dev) return sss_fail(-1, "NULL argument");
  if (int rc = be_launch_gnn(kind, a, stream)) return sss_fail(-30, std::string("gnn launch failed: ") + be_error(rc));
  return 0;
}

// The encoder of a Decima step in one call: list sizes stay on the device (the layer launches are sized by an upper bound
// and read their row count themselves), so nothing here waits for the device.
extern "C" int sss_gnn_encode(const sss_gnn_encode_args* g, void* stream) {
  if (!g) return sss_fail(-1, "NULL argument");
  if (g->n_nodes < 0 || g->n_jobs < 0 || g->n_obs < 1) return sss_fail(-33, "sss_gnn_encode: bad sizes");
  if (g->max_depth < 0 || g->max_depth > 32) return sss_fail(-27, "bad layer count");
  if (!g->w_prep_dev || !g->w_update_dev || !g->w_msg_dev || !g->w_dag_dev || !g->w_glob_dev || !g->x_dev || !g->out_deg_dev || !g->obs_depth_dev || !g->node_obs_dev ||
      !g->out_start_dev || !g->node_recv_dev || !g->job_first_dev || !g->job_nodes_dev || !g->obs_job_off_dev || !g->obs_jobs_dev ||
      !g->obs_node_off_dev || !g->obs_nodes_dev || !g->layer_cnt_dev || !g->h_init_dev || !g->h_dev || !g->tmp_dev || !g->h_dag_dev || !g->h_glob_dev ||
      (!g->env_off_dev && g->recv_stride == 0) || !g->layer_totals_dev || !g->recv_dev)
    return sss_fail(-1, "NULL argument");
  // (dst_dev / edge_layers_dev may be NULL: a batch without edges)
  if (g->recv_stride ? (g->recv_stride < g->n_nodes || g->recv_cap < g->recv_stride * (int64_t)g->max_depth) : g->recv_cap < g->n_nodes * (int64_t)g->max_depth)
    return sss_fail(-33, "sss_gnn_encode: recv_dev must hold n_nodes * max_depth entries");
  if ((g->n_nodes_dev == nullptr) != (g->n_jobs_dev == nullptr)) return sss_fail(-33, "sss_gnn_encode: n_nodes_dev and n_jobs_dev go together");
  if (g->n_nodes == 0 || g->n_jobs == 0) return 0;
  const bool on_dev = g->n_nodes_dev != nullptr;
  // grid sizes: the real totals when the host has them, else a guess with some room (the kernels stride over all rows)
  auto guess = [](int64_t hint, int64_t cap) { int64_t v = hint > 0 ? hint + hint / 4 + 64 : cap; return v > cap ? cap : (v < 1 ? 1 : v); };
  const int64_t rows_nodes = on_dev ? guess(g->n_nodes_hint, g->n_nodes) : g->n_nodes, rows_jobs = on_dev ? guess(g->n_jobs_hint, g->n_jobs) : g->n_jobs;
  auto fail = [](const char* what, int rc) { return sss_fail(-30, std::string(what) + " launch failed: " + be_error(rc)); };
  // the DAG layers: a launch per layer over the layer's receiving nodes of all observations, or ONE launch with a wave per
  // observation (layers_mode; 0: by the size of the observations - a wave walks its observation's tiles alone, which beats nine
  // launches at the launch floor only while observations are small: sss_gnn_mfma.h)
  if (g->layers_mode < 0 || g->layers_mode > 2) return sss_fail(-33, "sss_gnn_encode: layers_mode is 0 (the library chooses), 1 (a launch per layer) or 2 (one launch)");
  // (the one launch ends with its LARGEST observation; measured against nine launches per pass, profiles/r05_layers_per_observation.txt:
  // at 1024 observations it wins up to ~300 nodes in the largest one (a wave per SIMD), at 4096 - two waves per SIMD, twice the
  // matrix-core work per SIMD - only while every observation is small)
  const int64_t one_launch_max = g->n_obs <= 2048 ? 192 : 64;
  bool one_launch = g->max_depth > 0 && (g->layers_mode == 2 || (g->layers_mode == 0 && g->n_obs >= 64 && g->max_obs_nodes_hint > 0 && g->max_obs_nodes_hint <= one_launch_max));
  // lengths of the layers' lists of receiving nodes, their per-env offsets, the lists (only the launches per layer need them)
  // (recv_stride != 0: sss_decima_graph_build has written the lists and their lengths already - recv_lists_dev / layer_totals_dev)
  auto build_lists = [&]() -> int {
    if (g->recv_stride != 0) return 0;
    SssPrefixArgs p;
    p.src = g->layer_cnt_dev, p.row_stride = g->n_obs, p.col_stride = 1, p.mask = nullptr, p.n_rows = 32, p.n_cols = g->n_obs;
    p.off = g->env_off_dev, p.cnt = nullptr, p.totals = g->layer_totals_dev;
    if (int rc = be_launch_prefix_rows(p, stream)) return fail("prefix", rc);
    if (g->max_depth > 0) {
      SssDecimaListArgs d;
      d.node_off = g->obs_node_off_dev, d.obs_nodes = g->obs_nodes_dev, d.node_recv = (const uint32_t*)g->node_recv_dev, d.env_off = g->env_off_dev;
      for (int l = 0; l < 32; l++) d.layer_base[l] = 0;
      d.recv = g->recv_dev, d.n_layers = g->max_depth, d.totals = g->layer_totals_dev;
      if (int rc = be_launch_decima_lists(g->n_obs, d, stream)) return fail("decima lists", rc);
    }
    return 0;
  };
  SssGnnArgs a;
  memset(&a, 0, sizeof a);
  a.slope = g->slope, a.x = g->x_dev, a.h_init = g->h_init_dev, a.h = g->h_dev, a.tmp = g->tmp_dev, a.h_dag = g->h_dag_dev, a.h_glob = g->h_glob_dev;
  a.out_deg = g->out_deg_dev, a.obs_depth = g->obs_depth_dev, a.node_obs = g->node_obs_dev, a.dst = g->dst_dev, a.out_start = g->out_start_dev;
  a.edge_layers = g->edge_layers_dev, a.job_first = g->job_first_dev, a.job_nodes = g->job_nodes_dev, a.obs_job_off = g->obs_job_off_dev, a.obs_jobs = g->obs_jobs_dev;
  auto run = [&](int kind, int64_t rows, const float* w, const int64_t* rows_dev = nullptr) {
    a.n_rows = rows, a.w = w, a.n_rows_dev = rows_dev;
    return be_launch_gnn(kind, a, stream);
  };
  a.out = g->h_init_dev, a.w2 = g->w_update_dev;  // PREP and SINK in one pass over the nodes
  if (int rc = run(GNN_PREP, rows_nodes, g->w_prep_dev, g->n_nodes_dev)) return fail("gnn", rc);
  a.out = nullptr;
  // the layers, deepest first (scheduler.py:209-211): embeddings alternate between h and tmp per update (sss_gnn.h)
  a.node_recv = g->node_recv_dev, a.w2 = g->w_update_dev;
  if (one_launch) {
    a.w = g->w_msg_dev;
    const int rc = be_launch_gnn_layers_obs(a, g->obs_node_off_dev, g->obs_nodes_dev, g->layer_cnt_dev, g->n_obs, g->max_depth, stream);
    if (rc == BE_UNAVAILABLE) one_launch = false;  // (a build without the kernel: the launches per layer below)
    else if (rc) return fail("gnn", rc);
  }
  if (!one_launch) {
    if (int rc = build_lists()) return rc;
    a.idx0 = g->recv_dev, a.layer_totals = g->layer_totals_dev, a.idx0_stride = g->recv_stride;
    if (g->recv_stride)  // the graph kernel's lists: a dense piece per block of observations (sss_decima.h SSS_LIST_SETS)
      a.seg_off = g->obs_node_off_dev, a.n_seg = g->n_obs, a.list_q = (g->n_obs + SSS_LIST_SETS - 1) / SSS_LIST_SETS;
    a.w16 = g->w_msg16_dev, a.w2_16 = g->w_update16_dev;
    for (int lvl = g->max_depth - 1; lvl >= 0; lvl--) {
      a.layer = lvl;
      // (n_rows only sizes the grid here: the kernel reads the list's length itself and strides over all of it)
      int64_t rows = g->layer_rows_hint[lvl] >= 0 ? g->layer_rows_hint[lvl] + g->layer_rows_hint[lvl] / 4 + 64 : rows_nodes;
      if (rows > g->n_nodes) rows = g->n_nodes;
      if (int rc = run(GNN_LAYER, rows, g->w_msg_dev)) return fail("gnn", rc);
    }
  }
  a.idx0 = nullptr, a.layer_totals = nullptr, a.idx0_stride = 0, a.w2 = nullptr, a.w16 = nullptr, a.w2_16 = nullptr, a.layer = 0;
  a.seg_off = nullptr, a.n_seg = 0, a.list_q = 0;
  if (int rc = run(GNN_DAGHID, rows_nodes, g->w_dag_dev, g->n_nodes_dev)) return fail("gnn", rc);  // (brings the embeddings left in tmp home: MERGE)
  a.node_recv = nullptr;
  if (int rc = run(GNN_DAGSUM, rows_jobs, g->w_dag_dev, g->n_jobs_dev)) return fail("gnn", rc);
  if (int rc = run(GNN_GLOBHID, rows_jobs, g->w_glob_dev, g->n_jobs_dev)) return fail("gnn", rc);
  if (int rc = run(GNN_GLOBSUM, g->n_obs, g->w_glob_dev)) return fail("gnn", rc);
  return 0;
}

// the grid of sss_linear_wgrad: at most this many workgroups (4 waves each, 4 per CU) - each writes one partial
#define SSS_WGRAD_WGS 1024
extern "C" int64_t sss_linear_wgrad_scratch(int M, int N) {
  if (M < 1 || N < 1 || M > 64 || N > 64) return 0;
  return (int64_t)SSS_WGRAD_WGS * (int64_t)(N * M + N);
}
extern "C" int sss_linear_wgrad(const float* x_dev, int64_t ldx, const float* dy_dev, int64_t ldy, int64_t K, int M, int N, float* gw_dev, float* gb_dev,
                                float* scratch_dev, void* stream) {
  if (!x_dev || !dy_dev || !gw_dev || !scratch_dev) return sss_fail(-1, "NULL argument");
  if (M < 1 || N < 1 || M > 64 || N > 64) return sss_fail(-29, "sss_linear_wgrad: M and N must be in 1..64");
  if (K < 0 || ldx < M || ldy < N) return sss_fail(-29, "sss_linear_wgrad: bad row count or leading dimension");
  SssWgradArgs a;
  a.x = x_dev, a.dy = dy_dev, a.K = K, a.ldx = ldx, a.ldy = ldy, a.M = M, a.N = N, a.partial = scratch_dev, a.gw = gw_dev, a.gb = gb_dev;
  // no more workgroups than there are 64-row slabs (at least one: K = 0 still has to write zeros)
  int64_t slabs = (K + 63) / 64;
  a.n_partials = (int)(slabs < SSS_WGRAD_WGS ? (slabs > 0 ? slabs : 1) : SSS_WGRAD_WGS);
  if (int rc = be_launch_wgrad(a, stream)) return sss_fail(-30, std::string("wgrad launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_collect_step(const sss_collect_args* c, int phase, void* stream) {
  if (!c) return sss_fail(-1, "NULL argument");
  if (phase != 0 && phase != 1) return sss_fail(-32, "sss_collect_step: phase must be 0 (actions) or 1 (record)");
  if (c->num_envs < 0 || c->t < 0) return sss_fail(-32, "sss_collect_step: negative size");
  if (!c->active_dev || !c->stage_sel_dev || !c->exec_sel_dev) return sss_fail(-1, "NULL argument");
  if (phase == 0 && (!c->stage_idx_dev || !c->num_exec_dev)) return sss_fail(-1, "NULL argument");
  if (phase == 1 && (!c->obs_f64_dev || !c->obs_i32_dev || !c->time_limit_dev || !c->wall_dev || !c->elapsed_dev || !c->step_counts_dev || !c->pending_reset_dev ||
                     !c->job_idx_dev || !c->lgprob_dev || !c->rec_active_dev || !c->rec_t_before_dev || !c->rec_t_after_dev || !c->rec_rewards_dev ||
                     !c->rec_stage_sel_dev || !c->rec_job_idx_dev || !c->rec_exec_sel_dev || !c->rec_lgprobs_dev || !c->rec_resets_dev || !c->flags_dev))
    return sss_fail(-1, "NULL argument");
  if (phase == 1 && c->obs_i32_stride < 8) return sss_fail(-32, "sss_collect_step: obs_i32 rows have 8 entries");
  SssCollectArgs a;
  a.num_envs = c->num_envs, a.asynchronous = c->asynchronous, a.t = c->t, a.duration = c->duration;
  a.obs_f64 = c->obs_f64_dev, a.obs_i32 = c->obs_i32_dev, a.obs_i32_stride = c->obs_i32_stride, a.time_limit = c->time_limit_dev;
  a.active = c->active_dev, a.wall = c->wall_dev, a.elapsed = c->elapsed_dev, a.step_counts = c->step_counts_dev, a.pending_reset = c->pending_reset_dev;
  a.stage_sel = c->stage_sel_dev, a.job_idx = c->job_idx_dev, a.exec_sel = c->exec_sel_dev, a.lgprob = c->lgprob_dev;
  a.stage_idx = c->stage_idx_dev, a.num_exec = c->num_exec_dev;
  a.rec_active = c->rec_active_dev, a.rec_t_before = c->rec_t_before_dev, a.rec_t_after = c->rec_t_after_dev, a.rec_rewards = c->rec_rewards_dev;
  a.rec_stage_sel = c->rec_stage_sel_dev, a.rec_job_idx = c->rec_job_idx_dev, a.rec_exec_sel = c->rec_exec_sel_dev, a.rec_lgprobs = c->rec_lgprobs_dev;
  a.rec_resets = c->rec_resets_dev, a.flags = c->flags_dev, a.in_group = c->in_group_dev;
  if (a.num_envs == 0) return 0;
  if (int rc = be_launch_collect(a, phase, stream)) return sss_fail(-30, std::string("collect launch failed: ") + be_error(rc));
  return 0;
}

static int be_mlp_recompute_supported(int in_dim);
extern "C" int sss_mlp_recompute_supported(int in_dim) { return be_mlp_recompute_supported(in_dim); }
static int sss_mlp_check(const sss_mlp_args* a, bool backward) {
  if (!a || !a->w_dev) return sss_fail(-1, "NULL argument");
  // the forward pass of a GNN-shaped MLP may leave the hidden activations out (both NULL): sss_mlp_backward_wgrad then recomputes them
  const bool no_hidden = !a->a1_dev && !a->a2_dev && !backward && a->h1 == 32 && a->h2 == 16 && a->out_dim == 16 && a->act == 0 && be_mlp_recompute_supported(a->in_dim);
  if (!no_hidden && (!a->a1_dev || !a->a2_dev)) return sss_fail(-1, "NULL argument");
  if (a->rows < 0) return sss_fail(-31, "sss_mlp: negative row count");
  if (backward ? (!a->dy_dev || !a->g1_dev || !a->g2_dev) : (!a->x_dev || !a->y_dev)) return sss_fail(-1, "NULL argument");
  return 0;
}
static SssMlpArgs sss_mlp_args_of(const sss_mlp_args* a) {
  SssMlpArgs m;
  m.rows = a->rows, m.in_dim = a->in_dim, m.h1 = a->h1, m.h2 = a->h2, m.out_dim = a->out_dim, m.act = a->act, m.slope = a->slope;
  m.w = a->w_dev, m.x = a->x_dev, m.a1 = a->a1_dev, m.a2 = a->a2_dev, m.y = a->y_dev, m.dy = a->dy_dev, m.g1 = a->g1_dev, m.g2 = a->g2_dev, m.dx = a->dx_dev;
  m.x2 = a->x2_dev, m.dx2 = a->dx2_dev;
  return m;
}
static int be_mlp_split_supported(int in_dim);
extern "C" int sss_mlp_split_supported(int in_dim) { return be_mlp_split_supported(in_dim) && be_mlp_recompute_supported(in_dim); }
// x2_dev / dx2_dev: only where sss_mlp_split_supported says so, only without stored hidden activations, dx2 only with x2 and without dx
static int sss_mlp_check_split(const sss_mlp_args* a, bool forward) {
  if (!a->x2_dev && !a->dx2_dev) return 0;
  if (!sss_mlp_split_supported(a->in_dim) || a->a1_dev || a->a2_dev || !a->x2_dev || (forward && a->dx2_dev) || (a->dx2_dev && a->dx_dev))
    return sss_fail(-31, "sss_mlp: an input in two pieces (x2_dev / dx2_dev) is for sss_mlp_forward / sss_mlp_backward_wgrad of the 21 -> 32 -> 16 -> 16 MLP without stored "
                         "activations (sss_mlp_split_supported); dx2_dev excludes dx_dev");
  return 0;
}
extern "C" int sss_mlp_supported(int in_dim, int h1, int h2, int out_dim, int act) {
  const bool gnn = h1 == 32 && h2 == 16 && out_dim == 16 && act == 0, head = h1 == 64 && h2 == 64 && out_dim == 1 && act == 1;
  return (gnn && (in_dim == GNN_NF || in_dim == 16 || in_dim == GNN_NF + 16)) || (head && (in_dim == GNN_NF + 48 || in_dim == GNN_DF + 33));
}
extern "C" int sss_mlp_forward(const sss_mlp_args* a, void* stream) {
  if (int rc = sss_mlp_check(a, false)) return rc;
  if (int rc = sss_mlp_check_split(a, true)) return rc;
  if (!sss_mlp_supported(a->in_dim, a->h1, a->h2, a->out_dim, a->act)) return sss_fail(-31, "sss_mlp: not one of the architecture's MLP shapes");
  if (int rc = be_launch_mlp(sss_mlp_args_of(a), 0, stream)) return sss_fail(-30, std::string("mlp forward launch failed: ") + be_error(rc));
  return 0;
}
extern "C" int sss_mlp_backward(const sss_mlp_args* a, void* stream) {
  if (int rc = sss_mlp_check(a, true)) return rc;
  if (a->x2_dev || a->dx2_dev) return sss_fail(-31, "sss_mlp_backward: x2_dev / dx2_dev are for sss_mlp_backward_wgrad");
  if (!sss_mlp_supported(a->in_dim, a->h1, a->h2, a->out_dim, a->act)) return sss_fail(-31, "sss_mlp: not one of the architecture's MLP shapes");
  if (int rc = be_launch_mlp(sss_mlp_args_of(a), 1, stream)) return sss_fail(-30, std::string("mlp backward launch failed: ") + be_error(rc));
  return 0;
}

#include "sss_arena.h"
static int be_launch_arena(const SssArenaArgs& a, int64_t rows_hint, void* stream);
extern "C" int sss_arena_append(const sss_arena_args* a, void* stream) {
  if (!a || !a->totals_dev || !a->cursor_dev) return sss_fail(-1, "NULL argument");
  if (a->n_arrays < 0 || a->n_arrays > SSS_ARENA_MAX_ARRAYS || a->n_obs < 0) return sss_fail(-34, "sss_arena_append: bad array or observation count");
  SssArenaArgs r;
  r.n_arrays = a->n_arrays, r.n_obs = a->n_obs, r.totals = a->totals_dev, r.cursor = a->cursor_dev;
  for (int k = 0; k < 4; k++) r.capacity[k] = a->capacity[k];
  for (int i = 0; i < a->n_arrays; i++) {
    const sss_arena_array& s = a->arrays[i];
    if (!s.src_dev || !s.dst_dev) return sss_fail(-1, "NULL argument");
    if ((s.elem_bytes != 1 && s.elem_bytes != 4 && s.elem_bytes != 8) || s.per_row < 1 || s.kind < 0 || s.kind > 3 || s.shift < 0 || s.shift > 3 ||
        (s.shift && s.elem_bytes != 8))
      return sss_fail(-34, "sss_arena_append: bad array description");
    r.arrays[i].src = s.src_dev, r.arrays[i].dst = s.dst_dev, r.arrays[i].elem_bytes = s.elem_bytes, r.arrays[i].per_row = s.per_row, r.arrays[i].kind = s.kind,
    r.arrays[i].shift = s.shift;
  }
  if (int rc = be_launch_arena(r, a->rows_hint, stream)) return sss_fail(-30, std::string("arena launch failed: ") + be_error(rc));
  return 0;
}

#include "sss_returns.h"
static int be_launch_returns(const SssReturnsArgs& a, void* stream);
static int be_launch_baselines(const SssBaselineArgs& a, void* stream);
extern "C" int sss_discounted_returns(const sss_returns_args* a, void* stream) {
  if (!a || !a->active_dev || !a->t_before_dev || !a->t_after_dev || !a->rewards_dev || !a->out_dev) return sss_fail(-1, "NULL argument");
  if (a->T < 0 || a->B < 0) return sss_fail(-36, "sss_discounted_returns: negative size");
  SssReturnsArgs r;
  r.T = a->T, r.B = a->B, r.active = a->active_dev, r.t_before = a->t_before_dev, r.t_after = a->t_after_dev, r.rewards = a->rewards_dev, r.beta = a->beta, r.out = a->out_dev;
  if (r.T == 0 || r.B == 0) return 0;
  if (int rc = be_launch_returns(r, stream)) return sss_fail(-30, std::string("returns launch failed: ") + be_error(rc));
  return 0;
}
extern "C" int sss_sequence_baselines(const sss_baseline_args* a, void* stream) {
  if (!a || !a->active_dev || !a->times_dev || !a->values_dev || !a->n_dev || !a->out_dev) return sss_fail(-1, "NULL argument");
  if (a->T < 0 || a->B < 0 || a->R < 1 || a->B % a->R != 0) return sss_fail(-36, "sss_sequence_baselines: B must be a multiple of R >= 1");
  SssBaselineArgs r;
  r.T = a->T, r.B = a->B, r.R = a->R, r.skip_empty = a->skip_empty, r.active = a->active_dev, r.times = a->times_dev, r.values = a->values_dev, r.n = a->n_dev, r.out = a->out_dev;
  if (r.T == 0 || r.B == 0) return 0;
  if (int rc = be_launch_baselines(r, stream)) return sss_fail(-30, std::string("baselines launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_bit_lists(const sss_bit_list_args* a, void* stream) {
  if (!a || !a->bits_dev) return sss_fail(-1, "NULL argument");
  if (a->n < 0 || a->n_layers < 1 || a->n_layers > 32 || a->chunk < 64 || a->chunk % 64 != 0 || a->n_chunks != (int)((a->n + a->chunk - 1) / a->chunk) ||
      (a->phase != 0 && a->phase != 1))
    return sss_fail(-37, "sss_bit_lists: 1..32 layers, chunk a multiple of 64, n_chunks = ceil(n / chunk), pass 0 or 1");
  if (a->phase == 0 ? !a->cnt_dev : (!a->off_dev || !a->out_dev)) return sss_fail(-1, "NULL argument");
  if (a->n_chunks == 0) return 0;
  SssBitListArgs r;
  r.bits = a->bits_dev, r.n = a->n, r.n_layers = a->n_layers, r.chunk = a->chunk, r.n_chunks = a->n_chunks, r.pass = a->phase, r.cnt = a->cnt_dev, r.off = a->off_dev, r.out = a->out_dev;
  for (int l = 0; l < 32; l++) r.base[l] = a->base[l];
  if (int rc = be_launch_bit_lists(r, stream)) return sss_fail(-30, std::string("bit lists launch failed: ") + be_error(rc));
  return 0;
}

extern "C" int sss_rows_op(const sss_rows_args* a, void* stream) {
  if (!a || !a->a_dev || !a->b_dev || (a->n > 0 && !a->idx_dev)) return sss_fail(-1, "NULL argument");
  if (a->op < 0 || a->op > 5) return sss_fail(-33, "sss_rows_op: unknown operation");
  if (a->width < 1 || a->width > 64 || a->ld_a < a->width || a->n < 0) return sss_fail(-33, "sss_rows_op: width must be in 1..64, ld_a >= width, n >= 0");
  if ((a->op == 2 || a->op == 3) && !a->c_dev) return sss_fail(-1, "NULL argument");
  SssRowsArgs r;
  r.n = a->n, r.ld_a = a->ld_a, r.width = a->width, r.op = a->op, r.idx = a->idx_dev, r.a = a->a_dev, r.b = a->b_dev, r.c = a->c_dev;
  if (r.n == 0) return 0;
  if (int rc = be_launch_rows(r, stream)) return sss_fail(-30, std::string("rows launch failed: ") + be_error(rc));
  return 0;
}

#include "sss_segcat.h"
static int be_launch_segcat(const SssSegcatArgs& a, int backward, void* stream);
extern "C" int sss_segment_categorical(const sss_segcat_args* a, int backward, void* stream) {
  if (!a || !a->scores_dev || !a->ptr_dev || !a->chosen_dev) return sss_fail(-1, "NULL argument");
  if (a->n_seg < 0) return sss_fail(-38, "sss_segment_categorical: negative segment count");
  if (backward ? (!a->g_lg_dev || !a->g_ent_dev || !a->g_scores_dev) : (!a->lg_dev || !a->ent_dev)) return sss_fail(-1, "NULL argument");
  if (a->n_seg == 0) return 0;
  SssSegcatArgs r;
  r.n_seg = a->n_seg, r.scores = a->scores_dev, r.ptr = a->ptr_dev, r.chosen = a->chosen_dev, r.den_eps = a->den_eps, r.lg = a->lg_dev, r.ent = a->ent_dev;
  r.g_lg = a->g_lg_dev, r.g_ent = a->g_ent_dev, r.g_scores = a->g_scores_dev;
  if (int rc = be_launch_segcat(r, backward, stream)) return sss_fail(-30, std::string("segment categorical launch failed: ") + be_error(rc));
  return 0;
}

static int be_launch_concat(const SssConcatArgs& r, void* stream);
extern "C" int sss_rows_concat(const sss_concat_args* a, void* stream) {
  if (!a || !a->out_dev) return sss_fail(-1, "NULL argument");
  if (a->op != 0 && a->op != 1) return sss_fail(-33, "sss_rows_concat: unknown operation");
  if (a->n < 0 || a->n_parts < 1 || a->n_parts > SSS_CONCAT_MAX_PARTS) return sss_fail(-33, "sss_rows_concat: n >= 0 and 1..4 parts");
  SssConcatArgs r;
  r.n = a->n, r.n_parts = a->n_parts, r.op = a->op, r.out = a->out_dev;
  int w = 0;
  for (int k = 0; k < SSS_CONCAT_MAX_PARTS; k++) {
    const bool used = k < a->n_parts;
    if (used && (a->parts[k].width < 1 || a->parts[k].width > 64)) return sss_fail(-33, "sss_rows_concat: a part has 1..64 floats per row");
    if (used && !a->parts[k].table_dev && a->op == 0) return sss_fail(-1, "NULL argument");
    r.table[k] = used ? a->parts[k].table_dev : nullptr, r.idx[k] = used ? a->parts[k].idx_dev : nullptr, r.pw[k] = used ? a->parts[k].width : 0;
    w += r.pw[k], r.end[k] = w;
  }
  if (w > 64) return sss_fail(-33, "sss_rows_concat: at most 64 floats per row");
  r.width = w, r.inv = ((1 << 20) + w - 1) / w;
  for (uint32_t t = 0; t < 64u * (uint32_t)w; t++)  // (the kernel's division: exact for every width up to 64 - kept as a check of the constant)
    if (((t * (uint32_t)r.inv) >> 20) != t / (uint32_t)w) return sss_fail(-33, "sss_rows_concat: internal (division constant)");
  if (r.n == 0) return 0;
  if (int rc = be_launch_concat(r, stream)) return sss_fail(-30, std::string("concat launch failed: ") + be_error(rc));
  return 0;
}

#ifndef SSS_MLPW_SLOTS
#define SSS_MLPW_SLOTS 2048
#endif
#ifndef SSS_MLPW_HEAD_SLOTS
#define SSS_MLPW_HEAD_SLOTS 512
#endif
static int be_mlp_head_bwdw_supported();
// the MLP shapes sss_mlp_backward_wgrad takes, by input width (the architecture's five MLPs have five different ones)
struct SssMlpwShape {
  int h1, h2, out, act, slots;
};
static bool sss_mlpw_shape(int in_dim, SssMlpwShape* s) {
  if (in_dim == GNN_NF || in_dim == 16 || in_dim == GNN_NF + 16) return *s = SssMlpwShape{32, 16, 16, 0, SSS_MLPW_SLOTS}, true;
  if ((in_dim == GNN_NF + 48 || in_dim == GNN_DF + 33) && be_mlp_head_bwdw_supported()) return *s = SssMlpwShape{64, 64, 1, 1, SSS_MLPW_HEAD_SLOTS}, true;
  return false;
}
extern "C" int64_t sss_mlp_wgrad_scratch(int in_dim) {
  SssMlpwShape s;
  return sss_mlpw_shape(in_dim, &s) ? (int64_t)s.slots * ((s.out * s.h2 + s.out) + (s.h2 * s.h1 + s.h2) + (s.h1 * in_dim + s.h1)) : 0;
}
extern "C" int sss_mlp_backward_wgrad(const sss_mlp_args* a, float* acc_dev, void* stream) {
  if (!a || !a->w_dev || !a->dy_dev || !a->x_dev || !acc_dev) return sss_fail(-1, "NULL argument");
  if (!a->a1_dev != !a->a2_dev) return sss_fail(-1, "sss_mlp_backward_wgrad: a1_dev and a2_dev are given together or not at all");
  if (a->rows < 0) return sss_fail(-31, "sss_mlp: negative row count");
  SssMlpwShape s;
  if (!sss_mlpw_shape(a->in_dim, &s) || a->h1 != s.h1 || a->h2 != s.h2 || a->out_dim != s.out || a->act != s.act)
    return sss_fail(-31, "sss_mlp_backward_wgrad: (5 | 16 | 21) -> 32 -> 16 -> 16 LeakyReLU and (53 | 36) -> 64 -> 64 -> 1 Tanh MLPs only");
  if (int rc = sss_mlp_check_split(a, false)) return rc;
  if (!a->a1_dev && (s.act != 0 || !be_mlp_recompute_supported(a->in_dim)))
    return sss_fail(-31, "sss_mlp_backward_wgrad: this build cannot recompute this MLP's hidden activations (a1_dev / a2_dev required)");
  if (int rc = be_launch_mlp_bwdw(sss_mlp_args_of(a), acc_dev, stream)) return sss_fail(-30, std::string("mlp backward launch failed: ") + be_error(rc));
  return 0;
}
extern "C" int sss_mlp_wgrad_finish(int in_dim, const float* acc_dev, float* gw1_dev, float* gb1_dev, float* gw2_dev, float* gb2_dev, float* gw3_dev, float* gb3_dev,
                                    void* stream) {
  if (!acc_dev || !gw1_dev || !gb1_dev || !gw2_dev || !gb2_dev || !gw3_dev || !gb3_dev) return sss_fail(-1, "NULL argument");
  SssMlpwShape s;
  if (!sss_mlpw_shape(in_dim, &s)) return sss_fail(-31, "sss_mlp_wgrad_finish: not one of the MLPs sss_mlp_backward_wgrad takes");
  const float* l3 = acc_dev;
  const float* l2 = l3 + (size_t)s.slots * (s.out * s.h2 + s.out);
  const float* l1 = l2 + (size_t)s.slots * (s.h2 * s.h1 + s.h2);
  const struct { const float* part; int N, M; float* gw; float* gb; } job[3] = {{l3, s.out, s.h2, gw3_dev, gb3_dev}, {l2, s.h2, s.h1, gw2_dev, gb2_dev}, {l1, s.h1, in_dim, gw1_dev, gb1_dev}};
  for (int k = 0; k < 3; k++) {
    SssWgradArgs r;
    r.x = nullptr, r.dy = nullptr, r.K = 0, r.ldx = 0, r.ldy = 0, r.M = job[k].M, r.N = job[k].N, r.partial = const_cast<float*>(job[k].part), r.n_partials = s.slots;
    r.gw = job[k].gw, r.gb = job[k].gb;
    if (int rc = be_launch_wgrad_reduce(r, stream)) return sss_fail(-30, std::string("wgrad reduce launch failed: ") + be_error(rc));
  }
  return 0;
}

extern "C" void sss_destroy(sss_handle* h) {
  if (!h) return;
  BeDeviceGuard guard(h->device);
  be_free(h->pack_dev), be_free(h->zig_dev), be_free(h->eff_dev), be_free(h->jump_dev), be_free(h->common_dev);
  delete h;
}
