// sss_decima_policy.h - Decima's whole decision for one env in ONE wavefront:
// observation transform (env_wrapper.py:69-143, utils.py:238-267), GNN encoder and both score
// networks (scheduler.py:142-385) and the two softmax draws of DecimaScheduler.schedule
// (scheduler.py:71-99), reading only the env's observation rows and the packed MLP parameters.
// No host round trip, no device->host sync, no intermediate graph: with sss_step this makes the
// Decima-in-the-loop step two launches.
//
// Work split: lanes stride over the rows of each phase (nodes, a layer's receiving nodes, jobs,
// schedulable stages, executor counts); phases are separated by wave_sync. Per-node vectors
// (features, h_init, h, scratch) live in a per-env slab of global memory that stays in L2; the DAG
// analysis (generations, layer membership bits, out-edge ranges, node->job) lives in LDS,
// 18 bytes per node slot. The MLPs are the row functions of sss_gnn.h. The gfx950 build packs up to
// four envs (wavefronts) into one workgroup that first stages all seven MLPs' parameters (83 KB)
// in LDS - one workgroup per CU, every weight read a broadcast LDS read; phases of one env are
// separated by wavefront-local fences, never by workgroup barriers (envs differ in depth).
//
// Sampling: Gumbel-max with a counter-based uniform stream keyed by (seed, counter, env, candidate)
// - the reference samples with Python's unseeded `random.choices` (utils.py:19-23), i.e. any exact
// softmax sampler is faithful; this one needs no prefix sums and is reproducible.
#pragma once

enum { DP_W_PREP = 0, DP_W_MSG = 992, DP_W_UPD = 2336, DP_W_DAG = 3680, DP_W_GLOB = 5184, DP_W_STAGE = 6528, DP_W_EXEC = 14212,
       DP_W_TOTAL = 20808 };  // float offsets of the packed MLPs inside the staged block (sizes: gnn_mlp_params, 16-byte aligned starts)
enum { DP_X = 0, DP_HINIT = 5, DP_H = 21, DP_TMP = 37, DP_NODE_FLOATS = 53, DP_JOB_FLOATS = 32 };

struct SssDecimaPolicyArgs {
  const uint8_t* active;  // u8[B] or null
  float num_tasks_scale, work_scale, slope;
  const float *w_prep, *w_msg, *w_upd, *w_dag, *w_glob, *w_stage, *w_exec;
  float* node_scratch;  // f32[B][n_cap][DP_NODE_FLOATS]
  float* job_scratch;   // f32[B][J_cap][DP_JOB_FLOATS]
  uint64_t rng_seed, rng_counter;
  int32_t *stage_idx, *num_exec;             // the env's action format; stage_idx -1 when nothing is schedulable
  int32_t *stage_sel, *job_idx, *exec_sel;   // Decima's action tuple (scheduler.py:93)
  float* lgprob;
  float* stage_scores;  // nullable: f32[B][n_cap], -inf where not schedulable
  float* exec_scores;   // nullable: f32[B][E], -inf where not allowed
  uint64_t* prof;       // nullable: u64[B][8] shader cycles per phase (analysis, prep, layers, summaries, stage, exec)
};

SSS_SHARED_DYN(g_dp_lds);

SSS_DEV uint32_t dp_ordered(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
// lane-uniform (max value, index of the max; smallest index on ties); idx < 2^31
SSS_DEV void dp_wave_argmax(float v, uint32_t idx, float& vmax, uint32_t& imax) {
  uint64_t key = ((uint64_t)dp_ordered(v) << 32) | (uint64_t)(0xFFFFFFFFu - idx);
  uint64_t best = ~wave_min_u64(~key);
  imax = 0xFFFFFFFFu - (uint32_t)best;
  uint32_t o = (uint32_t)(best >> 32);
  o ^= (o >> 31) ? 0x80000000u : 0xFFFFFFFFu;
  memcpy(&vmax, &o, 4);
}
SSS_DEV float dp_gumbel(uint64_t seed, uint64_t counter, int env, uint32_t idx, uint32_t draw) {
  uint64_t z = seed ^ (counter * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)env << 32) ^ ((uint64_t)draw << 28) ^ idx;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  float u = ((float)(uint32_t)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
  return -logf(-logf(u));
}

SSS_DEV void decima_policy_wave(const SssLayout& L, const SssBuffers& B, int E, const SssDecimaPolicyArgs& d, int env, uint8_t* lds) {
  constexpr int F = GNN_EMB;
  int lane = wave_lane() & 63;  // several envs (wavefronts) may share a workgroup
  const int32_t* oi = B.obs_i32 + (size_t)env * SSS_OBS_I32;
  bool on = d.active == nullptr || d.active[env] != 0;
  int n = on ? oi[OBS_N_NODES] : 0, ne = on ? oi[OBS_N_EDGES] : 0, A = on ? oi[OBS_N_JOBS] : 0;
  if (n == 0 || A == 0) {  // wave-uniform
    if (lane == 0) {
      d.stage_idx[env] = -1, d.num_exec[env] = 1, d.stage_sel[env] = 0, d.job_idx[env] = 0, d.exec_sel[env] = 0, d.lgprob[env] = 0.0f;
    }
    return;
  }
  int ncommit = oi[OBS_NUM_COMMITTABLE], src_idx = oi[OBS_SOURCE_JOB_IDX];
  const float* nodes = B.nodes + (size_t)env * L.n_cap * 3;
  const int32_t* el = B.edge_links + (size_t)env * L.ed_cap * 2;
  const int32_t* dag_ptr = B.dag_ptr + (size_t)env * (L.J_cap + 1);
  const int32_t* sup = B.exec_supplies + (size_t)env * L.J_cap;
  float* NS = d.node_scratch + (size_t)env * L.n_cap * DP_NODE_FLOATS;
  float* JS = d.job_scratch + (size_t)env * L.J_cap * DP_JOB_FLOATS;
  int32_t* gen = (int32_t*)lds;
  uint32_t* memb = (uint32_t*)(lds + (size_t)4 * L.n_cap);
  uint32_t* recv = (uint32_t*)(lds + (size_t)8 * L.n_cap);
  uint16_t* ostart = (uint16_t*)(lds + (size_t)12 * L.n_cap);
  uint16_t* oend = (uint16_t*)(lds + (size_t)14 * L.n_cap);
  uint16_t* njob = (uint16_t*)(lds + (size_t)16 * L.n_cap);
  uint16_t* list = (uint16_t*)(lds + (size_t)18 * L.n_cap);  // compacted row ids of the current phase
  float* hglob = (float*)(lds + (size_t)20 * L.n_cap);    // 16 floats (L.n_cap is even)

  uint64_t pc0 = wave_clock();
  // ---- DAG analysis (as in sss_decima_graph_kernel) -------------------------------------------
  for (int i = lane; i < n; i += 64) gen[i] = 0, recv[i] = 0, ostart[i] = 0, oend[i] = 0;
  wave_sync_local();
  for (int it = 0; it <= n; it++) {
    bool moved = false;
    for (int e = lane; e < ne; e += 64) {
      int u = el[2 * e], v = el[2 * e + 1];
      int gu = gen[u] + 1;
      if (gen[v] < gu) lane_atomic_max_i32(&gen[v], gu), moved = true;
    }
    wave_sync_local();
    if (!wave_ballot(moved)) break;
  }
  uint32_t depth = 0;
  for (int i = lane; i < n; i += 64) {
    memb[i] = 1u << gen[i];
    if ((uint32_t)gen[i] > depth) depth = (uint32_t)gen[i];
    int lo = 0, hi = A;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (dag_ptr[mid] <= i) lo = mid; else hi = mid;
    }
    njob[i] = (uint16_t)lo;
  }
  depth = ~wave_min_u32(~depth);
  wave_sync_local();
  for (int e = lane; e < ne; e += 64) lane_atomic_or_u32(&memb[el[2 * e + 1]], 1u << gen[el[2 * e]]);
  wave_sync_local();
  for (int e = lane; e < ne; e += 64) {
    int u = el[2 * e], v = el[2 * e + 1];
    lane_atomic_or_u32(&recv[u], memb[u] & memb[v]);
    if (e == 0 || el[2 * (e - 1)] != u) ostart[u] = (uint16_t)e;
    if (e == ne - 1 || el[2 * (e + 1)] != u) oend[u] = (uint16_t)(e + 1);
  }
  wave_sync_local();

  uint64_t pc1 = wave_clock();
  // ---- node features, h_init, starting h (scheduler.py:200-209) ----------------------------------
  for (int i = lane; i < n; i += 64) {
    int a = njob[i];
    int supply = sup[a];
    int gap = E - supply;
    if (gap < 0) gap = 0;
    int cap = gap < ncommit ? gap : ncommit;
    if (a == src_idx) cap = ncommit;
    float rem = nodes[3 * i], dur = nodes[3 * i + 1];
    float x[GNN_NF], h2[16];
    x[0] = (float)((double)cap / (double)E);
    x[1] = a == src_idx ? 1.0f : -1.0f;
    x[2] = (float)((double)supply / (double)E);
    x[3] = rem / d.num_tasks_scale;
    x[4] = rem * dur / d.work_scale;
    float* row = NS + (size_t)i * DP_NODE_FLOATS;
    GNN_UNROLL for (int k = 0; k < GNN_NF; k++) row[DP_X + k] = x[k];
    gnn_hidden<GNN_NF, 32, 16, 0>(d.w_prep, x, h2, d.slope);
    gnn_out<GNN_NF, 32, 16, F>(d.w_prep, h2, 1.0f, [&](int o, float v) { row[DP_HINIT + o] = v; });
    bool par = oend[i] != 0;
    if (depth == 0 || par) {
      for (int o = 0; o < F; o++) row[DP_H + o] = depth == 0 ? row[DP_HINIT + o] : 0.0f;
    } else {
      float hi_[F];
      gnn_load<F>(row + DP_HINIT, hi_);
      gnn_hidden<F, 32, 16, 0>(d.w_upd, hi_, h2, d.slope);
      gnn_out<F, 32, 16, F>(d.w_upd, h2, 1.0f, [&](int o, float v) { row[DP_H + o] = v; });
    }
  }
  wave_sync_local();

  uint64_t pc2 = wave_clock();
  // ---- message passing, deepest DAG layer first (scheduler.py:211-236) ---------------------------
  uint64_t lt = bit64(lane) - 1;
  for (int l = (int)depth - 1; l >= 0; l--) {
    // the layer's receiving nodes, compacted: the heavy body then runs on full lanes instead of once
    // per 64-node round that happens to contain a receiver
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      int i = i0 + lane;
      bool on = i < n && ((recv[i] >> l) & 1u);
      uint64_t bal = wave_ballot(on);
      if (on) list[cnt + popc64(bal & lt)] = (uint16_t)i;
      cnt += popc64(bal);
    }
    wave_sync_local();
    for (int k = lane; k < cnt; k += 64) {
      int i = list[k];
      float acc[16], x[F], h2[16];
      GNN_UNROLL for (int k = 0; k < 16; k++) acc[k] = 0.0f;
      int used = 0;
      for (int e = ostart[i]; e < (int)oend[i]; e++) {
        int v = el[2 * e + 1];
        if (!(((memb[i] & memb[v]) >> l) & 1u)) continue;
        gnn_load<F>(NS + (size_t)v * DP_NODE_FLOATS + DP_H, x);
        gnn_hidden<F, 32, 16, 0>(d.w_msg, x, h2, d.slope);
        GNN_UNROLL for (int k = 0; k < 16; k++) acc[k] += h2[k];
        used++;
      }
      float agg[F];
      {
        GNN_FP_CONTRACT
        const float* W3 = d.w_msg + 32 * F + 32 + 16 * 32 + 16;
        const float* b3 = W3 + F * 16;
        GNN_UNROLL for (int o = 0; o < F; o++) {
          float v = b3[o] * (float)used;
          GNN_UNROLL for (int k = 0; k < 16; k++) v += W3[o * 16 + k] * acc[k];
          agg[o] = v;
        }
      }
      gnn_hidden<F, 32, 16, 0>(d.w_upd, agg, h2, d.slope);
      float* row = NS + (size_t)i * DP_NODE_FLOATS;
      gnn_out<F, 32, 16, F>(d.w_upd, h2, 1.0f, [&](int o, float v) { row[DP_TMP + o] = row[DP_HINIT + o] + v; });
    }
    wave_sync_local();
    for (int k = lane; k < cnt; k += 64) {
      float* row = NS + (size_t)list[k] * DP_NODE_FLOATS;
      for (int o = 0; o < F; o++) row[DP_H + o] = row[DP_TMP + o];
    }
    wave_sync_local();
  }

  uint64_t pc3 = wave_clock();
  // ---- job summaries and the global summary (scheduler.py:246-283) -------------------------------
  for (int i = lane; i < n; i += 64) {
    float* row = NS + (size_t)i * DP_NODE_FLOATS;
    float x[GNN_NF + F], h2[16];
    gnn_load<GNN_NF>(row + DP_X, x);
    gnn_load<F>(row + DP_H, x + GNN_NF);
    gnn_hidden<GNN_NF + F, 32, 16, 0>(d.w_dag, x, h2, d.slope);
    GNN_UNROLL for (int k = 0; k < 16; k++) row[DP_TMP + k] = h2[k];
  }
  wave_sync_local();
  for (int a = lane; a < A; a += 64) {
    float acc[16], x[F], h2[16];
    GNN_UNROLL for (int k = 0; k < 16; k++) acc[k] = 0.0f;
    int i0 = dag_ptr[a], i1 = dag_ptr[a + 1];
    for (int i = i0; i < i1; i++) {
      const float* t = NS + (size_t)i * DP_NODE_FLOATS + DP_TMP;
      GNN_UNROLL for (int k = 0; k < 16; k++) acc[k] += t[k];
    }
    float* jr = JS + (size_t)a * DP_JOB_FLOATS;
    gnn_out<GNN_NF + F, 32, 16, F>(d.w_dag, acc, (float)(i1 - i0), [&](int o, float v) { jr[o] = v; });
    gnn_load<F>(jr, x);
    gnn_hidden<F, 32, 16, 0>(d.w_glob, x, h2, d.slope);
    GNN_UNROLL for (int k = 0; k < 16; k++) jr[16 + k] = h2[k];
  }
  wave_sync_local();
  {
    float acc[16];
    GNN_UNROLL for (int k = 0; k < 16; k++) acc[k] = 0.0f;
    for (int a = 0; a < A; a++) {
      const float* t = JS + (size_t)a * DP_JOB_FLOATS + 16;
      GNN_UNROLL for (int k = 0; k < 16; k++) acc[k] += t[k];
    }
    gnn_out<F, 32, 16, F>(d.w_glob, acc, (float)A, [&](int o, float v) { if (lane == 0) hglob[o] = v; });
  }
  wave_sync_local();

  uint64_t pc4 = wave_clock();
  // ---- stage scores + first draw (scheduler.py:80-84, 296-318) -----------------------------------
  float best_key = -__builtin_inff(), best_score = 0.0f, m_run = -__builtin_inff(), s_run = 0.0f;
  uint32_t best_i = 0x7FFFFFFFu;
  int n_sched = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    int i = i0 + lane;
    bool on = i < n && nodes[3 * i + 2] != 0.0f;
    uint64_t bal = wave_ballot(on);
    if (on) list[n_sched + popc64(bal & (bit64(lane) - 1))] = (uint16_t)i;
    n_sched += popc64(bal);
    if (d.stage_scores && i < n && !on) d.stage_scores[(size_t)env * L.n_cap + i] = -__builtin_inff();
  }
  wave_sync_local();
  for (int k = lane; k < n_sched; k += 64) {
    int i = list[k];
    float sc;
    {
      const float* row = NS + (size_t)i * DP_NODE_FLOATS;
      float x[GNN_NF + 3 * F], h2[64];
      gnn_load<GNN_NF>(row + DP_X, x);
      gnn_load<F>(row + DP_H, x + GNN_NF);
      gnn_load<F>(JS + (size_t)njob[i] * DP_JOB_FLOATS, x + GNN_NF + F);
      GNN_UNROLL for (int q = 0; q < F; q++) x[GNN_NF + 2 * F + q] = hglob[q];
      gnn_hidden<GNN_NF + 3 * F, 64, 64, 1>(d.w_stage, x, h2, 0.0f);
      gnn_out<GNN_NF + 3 * F, 64, 64, 1>(d.w_stage, h2, 1.0f, [&](int, float v) { sc = v; });
      float key = sc + dp_gumbel(d.rng_seed, d.rng_counter, env, (uint32_t)i, 0);
      if (key > best_key) best_key = key, best_i = (uint32_t)i, best_score = sc;
      float m_new = sc > m_run ? sc : m_run;
      s_run = s_run * expf(m_run - m_new) + expf(sc - m_new);
      m_run = m_new;
    }
    if (d.stage_scores) d.stage_scores[(size_t)env * L.n_cap + i] = sc;
  }
  float kmax;
  uint32_t sel;
  dp_wave_argmax(best_key, best_i, kmax, sel);
  bool any_stage = wave_ballot(best_i != 0x7FFFFFFFu) != 0;
  if (!any_stage) {  // wave-uniform
    if (lane == 0) {
      d.stage_idx[env] = -1, d.num_exec[env] = 1, d.stage_sel[env] = 0, d.job_idx[env] = 0, d.exec_sel[env] = 0, d.lgprob[env] = 0.0f;
    }
    return;
  }
  float dummy, M;
  uint32_t dummy_i;
  dp_wave_argmax(m_run, 0, M, dummy_i);
  float S = wave_sum_f32(m_run == -__builtin_inff() ? 0.0f : s_run * expf(m_run - M));
  dp_wave_argmax(best_i == sel ? best_score : -__builtin_inff(), 0, dummy, dummy_i);
  float lg_stage = dummy - M - logf(S);
  uint32_t rank = 0;
  for (int i = lane; i < (int)sel; i += 64) rank += nodes[3 * i + 2] != 0.0f;
  rank = wave_sum_u32(rank);

  uint64_t pc5 = wave_clock();
  // ---- executor-count scores of the chosen stage's job + second draw (scheduler.py:90-91, 337-385)
  int a_sel = njob[sel];
  int cap;
  {
    int gap = E - sup[a_sel];
    if (gap < 0) gap = 0;
    cap = gap < ncommit ? gap : ncommit;
    if (a_sel == src_idx) cap = ncommit;
  }
  // executor count c lives on lane c & 63, slot c >> 6 (up to 128 executors: two counts per lane)
  float esc[2] = {-__builtin_inff(), -__builtin_inff()};
  float ekey = -__builtin_inff(), ebest = -__builtin_inff(), emax = -__builtin_inff();
  uint32_t ebest_c = (uint32_t)lane;
  for (int q = 0; q < 2; q++) {
    int c = lane + 64 * q;
    if (c >= E) break;
    float x[GNN_DF + 2 * F + 1], h2[64];
    gnn_load<GNN_DF>(NS + (size_t)dag_ptr[a_sel] * DP_NODE_FLOATS + DP_X, x);
    gnn_load<F>(JS + (size_t)a_sel * DP_JOB_FLOATS, x + GNN_DF);
    GNN_UNROLL for (int k = 0; k < F; k++) x[GNN_DF + F + k] = hglob[k];
    x[GNN_DF + 2 * F] = (float)c / (float)E;
    gnn_hidden<GNN_DF + 2 * F + 1, 64, 64, 1>(d.w_exec, x, h2, 0.0f);
    float v0 = 0.0f;
    gnn_out<GNN_DF + 2 * F + 1, 64, 64, 1>(d.w_exec, h2, 1.0f, [&](int, float v) { v0 = v; });
    if (c < cap) esc[q] = v0;
    if (d.exec_scores) d.exec_scores[(size_t)env * E + c] = esc[q];
    if (esc[q] != -__builtin_inff()) {
      float key = esc[q] + dp_gumbel(d.rng_seed, d.rng_counter, env, (uint32_t)c, 1);
      if (key > ekey) ekey = key, ebest = esc[q], ebest_c = (uint32_t)c;
      if (esc[q] > emax) emax = esc[q];
    }
  }
  bool ok = emax != -__builtin_inff();
  uint32_t csel;
  dp_wave_argmax(ekey, ebest_c, kmax, csel);
  bool any_exec = wave_ballot(ok) != 0;
  float EM, ES = 0.0f, esel = 0.0f;
  dp_wave_argmax(emax, 0, EM, dummy_i);
  if (any_exec) {
    float part = 0.0f;
    for (int q = 0; q < 2; q++) part += esc[q] != -__builtin_inff() ? expf(esc[q] - EM) : 0.0f;
    ES = wave_sum_f32(part);
    dp_wave_argmax(ok && ebest_c == csel ? ebest : -__builtin_inff(), 0, esel, dummy_i);
  } else {
    csel = 0;
  }
  if (lane == 0) {
    d.stage_idx[env] = (int32_t)rank, d.num_exec[env] = (int32_t)csel + 1;
    d.stage_sel[env] = (int32_t)rank, d.job_idx[env] = a_sel, d.exec_sel[env] = (int32_t)csel;
    d.lgprob[env] = lg_stage + (any_exec ? esel - EM - logf(ES) : 0.0f);
    if (d.prof) {
      uint64_t pc6 = wave_clock();
      uint64_t* pr = d.prof + (size_t)env * 8;
      pr[0] = pc1 - pc0, pr[1] = pc2 - pc1, pr[2] = pc3 - pc2, pr[3] = pc4 - pc3, pr[4] = pc5 - pc4, pr[5] = pc6 - pc5, pr[6] = depth, pr[7] = (uint64_t)n;
    }
  }
}

SSS_KERNEL void sss_decima_policy_kernel(SssLayout L, SssBuffers B, int E, SssDecimaPolicyArgs d) {
  decima_policy_wave(L, B, E, d, wave_env(), g_dp_lds);
}

// ---- the two softmax draws of DecimaScheduler.schedule for the row-parallel pipeline ------------
// (scheduler.py:80-99): one wavefront per observation; `first` = the stage draw from the padded
// stage scores (sss_gnn_launch STAGE), `second` = the executor-count draw from the scores of the
// chosen stage's job (sss_gnn_launch EXEC). Same Gumbel-max stream as decima_policy_wave.
struct SssDecimaSampleArgs {
  int64_t n_pad;
  int E;
  uint64_t rng_seed, rng_counter;
  const float* stage_scores;   // f32[B][n_pad]
  const float* exec_scores;    // f32[B][E] (second draw)
  const int64_t *obs_nodes, *obs_node_off, *obs_job_off;
  const int64_t *sched_rank, *node_job;  // flat [M]
  int64_t* job_gid;            // i64[B]: flat job id of the chosen stage (0 when nothing is schedulable)
  int32_t *stage_idx, *num_exec;
  int64_t *stage_sel, *job_idx, *exec_sel;
  float* lgprob;
  uint8_t* any_stage;
};

SSS_KERNEL void sss_decima_sample_stage_kernel(SssDecimaSampleArgs d) {
  int env = wave_env(), lane = wave_lane();
  int n = (int)d.obs_nodes[env];
  const float* row = d.stage_scores + (size_t)env * d.n_pad;
  float best_key = -__builtin_inff(), best_score = 0.0f, m_run = -__builtin_inff(), s_run = 0.0f;
  uint32_t best_i = 0x7FFFFFFFu;
  const int64_t* rank = d.sched_rank + d.obs_node_off[env];
  for (int i = lane; i < n; i += 64) {
    float sc = row[i];
    // (a slot that is not a schedulable stage holds -inf - or, in a score matrix that is refilled without being cleared, whatever an
    // earlier pass left there: the stage's rank decides as well)
    if (rank[i] < 0 || sc == -__builtin_inff()) continue;
    float key = sc + dp_gumbel(d.rng_seed, d.rng_counter, env, (uint32_t)i, 0);
    if (key > best_key) best_key = key, best_i = (uint32_t)i, best_score = sc;
    float m_new = sc > m_run ? sc : m_run;
    s_run = s_run * expf(m_run - m_new) + expf(sc - m_new);
    m_run = m_new;
  }
  float kmax, M, sel_score;
  uint32_t sel, dummy_i;
  dp_wave_argmax(best_key, best_i, kmax, sel);
  bool any_stage = wave_ballot(best_i != 0x7FFFFFFFu) != 0;
  if (!any_stage) {  // wave-uniform
    if (lane == 0) {
      d.any_stage[env] = 0, d.job_gid[env] = 0, d.stage_idx[env] = -1, d.stage_sel[env] = 0, d.job_idx[env] = 0, d.lgprob[env] = 0.0f;
    }
    return;
  }
  dp_wave_argmax(m_run, 0, M, dummy_i);
  float S = wave_sum_f32(m_run == -__builtin_inff() ? 0.0f : s_run * expf(m_run - M));
  dp_wave_argmax(best_i == sel ? best_score : -__builtin_inff(), 0, sel_score, dummy_i);
  if (lane == 0) {
    int64_t flat = d.obs_node_off[env] + sel;
    int64_t jg = d.node_job[flat];
    d.any_stage[env] = 1, d.job_gid[env] = jg;
    d.stage_idx[env] = (int32_t)d.sched_rank[flat], d.stage_sel[env] = d.sched_rank[flat], d.job_idx[env] = jg - d.obs_job_off[env];
    d.lgprob[env] = sel_score - M - logf(S);
  }
}

SSS_KERNEL void sss_decima_sample_exec_kernel(SssDecimaSampleArgs d) {
  int env = wave_env(), lane = wave_lane();
  bool live = d.any_stage[env] != 0;
  const float* row = d.exec_scores + (size_t)env * d.E;
  // lanes stride over the executor counts (any E); with E <= 64 every lane holds one count as before
  float ekey = -__builtin_inff(), ebest = -__builtin_inff(), emax = -__builtin_inff();
  uint32_t ebest_c = (uint32_t)lane;
  for (int c = lane; live && c < d.E; c += 64) {
    float esc = row[c];
    if (esc == -__builtin_inff()) continue;
    float key = esc + dp_gumbel(d.rng_seed, d.rng_counter, env, (uint32_t)c, 1);
    if (key > ekey) ekey = key, ebest = esc, ebest_c = (uint32_t)c;
    if (esc > emax) emax = esc;
  }
  bool ok = emax != -__builtin_inff();
  float kmax, EM, esel;
  uint32_t csel, dummy_i;
  dp_wave_argmax(ekey, ebest_c, kmax, csel);
  bool any_exec = wave_ballot(ok) != 0;
  dp_wave_argmax(emax, 0, EM, dummy_i);
  float part = 0.0f;
  for (int c = lane; ok && c < d.E; c += 64) {
    float esc = row[c];
    part += esc != -__builtin_inff() ? expf(esc - EM) : 0.0f;
  }
  float ES = wave_sum_f32(part);
  dp_wave_argmax(ok && ebest_c == csel ? ebest : -__builtin_inff(), 0, esel, dummy_i);
  if (lane == 0) {
    if (!any_exec) csel = 0;
    d.exec_sel[env] = csel, d.num_exec[env] = (int32_t)csel + 1;
    if (any_exec) d.lgprob[env] += esel - EM - logf(ES);
  }
}
