// sss_sim_observe.h - part of the simulator device code (csrc/sss_sim.h includes the parts in order; not a stand-alone header):
// the wave-parallel schedulable-stage scan and the observation writer.
// Reference citations as in sss_sim.h (ENV / TRK / JOB / STG / TPCH / EVQ : line).
#undef SSS_SRC_ID
#define SSS_SRC_ID 10  // SssHdr::err_line = SSS_SRC_ID * 100000 + line of the check that failed
// _find_schedulable_stages() over all active jobs (ENV:505-540): one lane per stage of a job,
// ballot gives the job's ready mask; sat_mask makes the parent test a mask operation.
// Returns len(schedulable_stages); lane 0 stores the per-job masks.
// n_active / source job come from the mailbox lane 0 filled before the preceding wave_sync
// (publish_scan_inputs): lane 0 may already be past this function when another lane reads them.
SSS_DEV int find_schedulable_all() {
  PROF3(21);
  int lane = wave_lane();
  int A = g_sc.m_n_active;
  int src_job = g_sc.m_src_job;
  uint32_t total = 0;
  // one lane per active job; readiness of a stage is a mask test against the job's saturated mask
  for (int a0 = 0; a0 < A; a0 += 64) {
    int a = a0 + lane;
    uint32_t cnt = 0;
    if (a < A) {
      int j = lds_active()[a];
      SssJob* job = jobp(j);
      uint64_t m = 0;
      if (j == src_job || (int)job->supply < g_c.E) m = ready_mask_of_job(*job, false);
      job->sched_mask = m;
      cnt = (uint32_t)popc64(m);
    }
    total += wave_sum_u32(cnt);
  }
  return (int)total;
}

// _observe (ENV:345-406) + utils.subgraph (utils.py:5-22) into the env's padded output rows
SSS_DEV void write_observation(const SssLayout& L, const SssBuffers& B, int env, double reward) {
  PROF3(22);
  int lane = wave_lane();
  uint64_t t_obs0 = wave_clock();
  const SssHdr& h = g_hot.h;
  // the output rows alias nothing that is read here: the loads of later iterations may pass earlier stores
  float* __restrict__ nodes = B.nodes + (size_t)env * L.n_cap * 3;
  int32_t* __restrict__ el = B.edge_links + (size_t)env * L.ed_cap * 2;
  int32_t* __restrict__ dag_ptr = B.dag_ptr + (size_t)env * (L.J_cap + 1);
  int32_t* __restrict__ sup = B.exec_supplies + (size_t)env * L.J_cap;
  int A = h.n_active;
  uint32_t srck = h.curr_source;
  int src_job = (srck == POOL_NONE || srck == POOL_COMMON) ? -1 : key_job(srck);
  int src_idx = A;  // ENV:352
  uint64_t lt = bit64(lane) - 1;
  uint16_t* nbase = lds_keys();  // first node row of each active job (scratch shared with the set code)
  // pass 1 - lanes over jobs: dag_ptr (exclusive scan of active-stage counts), exec_supplies
  uint32_t run = 0;
  for (int a0 = 0; a0 < A; a0 += 64) {
    int a = a0 + lane;
    uint32_t cnt = 0;
    int j = -1, supply = 0;
    if (a < A) {
      j = lds_active()[a];
      const SssJob* job = jobp(j);
      cnt = (uint32_t)popc64(job->active_mask);
      supply = job->supply;
    }
    uint32_t excl = wave_scan_excl_u32(cnt);
    uint32_t tot = wave_sum_u32(cnt);
    uint64_t is_src = wave_ballot(a < A && j == src_job);
    if (is_src) src_idx = a0 + ctz64(is_src);
    if (a < A) {
      nbase[a] = (uint16_t)(run + excl);
      dag_ptr[a] = (int32_t)(run + excl);
      sup[a] = supply;
    }
    run += tot;
  }
  int base_n = (int)run;
  wave_sync();
  // pass 2 - lanes over (job, stage): node rows
  int SPn = g_c.SP;
  // four rows per lane at a time, every load of the four issued before the first store (one round trip to HBM per
  // 256 rows instead of one per 64)
  for (int i0 = lane; i0 < A * SPn; i0 += 64 * 4) {
    int32_t remaining[4];
    float recent[4];
    uint64_t act[4], sched[4];
    int nst[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u;
      remaining[u] = 0, recent[u] = 0.0f, act[u] = 0, sched[u] = 0, nst[u] = 0;
      if (i < A * SPn) {
        const int a = i / SPn, st = i - a * SPn;
        const JobView v = jobview(lds_active()[a]);  // one look-up of the job's slot for the three records
        // the stage's counters and duration are fetched along with the job's record, not after it (their
        // addresses do not depend on it; rows of inactive stages are read and dropped)
        remaining[u] = v.st[st].remaining;
        recent[u] = v.dur[st];
        act[u] = v.job->active_mask, sched[u] = v.job->sched_mask, nst[u] = (int)v.job->n_stages;
      }
    }
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u;
      if (i < A * SPn) {
        const int a = i / SPn, st = i - a * SPn;
        if (st < nst[u] && (act[u] & bit64(st))) {
          const int row = (int)nbase[a] + popc64(act[u] & (bit64(st) - 1));
          // plain stores: non-temporal ones were measured to double the HBM write traffic (partial
          // lines are no longer combined in L2) for no gain in time
          nodes[row * 3 + 0] = (float)remaining[u];
          nodes[row * 3 + 1] = recent[u];
          nodes[row * 3 + 2] = (sched[u] & bit64(st)) ? 1.0f : 0.0f;
        }
      }
    }
  }
  // pass 3 - lanes over (job, template edge): active subgraph, compacted in (job, edge) order. The rows are a
  // function of the active jobs, their order and their active-stage masks alone: when none of that has changed
  // since they were last written to this buffer, they are there already.
  const bool same_graph = h.obs_graph_version == h.graph_version && h.obs_bind_gen == B.gen;
  int ME = g_c.P.max_edges;
  int base_e = same_graph ? h.obs_n_edges : 0;
  // four groups of 64 (job, edge) pairs at a time: the four job records, then the four edges, are fetched together;
  // the compaction below stays in (job, edge) order
  for (int i0 = 0; i0 < (same_graph ? 0 : A * ME); i0 += 64 * 4) {
    uint64_t act[4];
    int eoff[4], nb[4];
    bool has[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u + lane;
      has[u] = false, act[u] = 0, eoff[u] = 0, nb[u] = 0;
      if (i < A * ME) {
        const int a = i / ME, e = i - a * ME;
        const SssJob* job = jobp(lds_active()[a]);
        has[u] = e < (int)job->n_edges;
        act[u] = job->active_mask, eoff[u] = job->edge_off + e, nb[u] = (int)nbase[a];
      }
    }
    int uu[4], vv[4];
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      uu[u] = 0, vv[u] = 0;
      if (has[u]) uu[u] = g_c.pk.edges[2 * eoff[u]], vv[u] = g_c.pk.edges[2 * eoff[u] + 1];
    }
    SSS_UNROLL4 for (int u = 0; u < 4; u++) {
      if (i0 + 64 * u >= A * ME) break;
      const bool keep = has[u] && (act[u] & bit64(uu[u])) && (act[u] & bit64(vv[u]));
      const uint64_t bal = wave_ballot(keep);
      if (keep) {
        const int pos = base_e + popc64(bal & lt);
        el[2 * pos + 0] = nb[u] + popc64(act[u] & (bit64(uu[u]) - 1));
        el[2 * pos + 1] = nb[u] + popc64(act[u] & (bit64(vv[u]) - 1));
      }
      base_e += popc64(bal);
    }
  }
  if (lane == 0) {
    dag_ptr[A] = base_n;
    int32_t* oi = B.obs_i32 + (size_t)env * SSS_OBS_I32;
    double* of = B.obs_f64 + (size_t)env * SSS_OBS_F64;
    int ncommit = 0;
    if (srck != POOL_NONE) {
      int p = pool_index(srck);
      ncommit = (int)g_c.pool_hdr[p].used - (int)g_c.pool_hdr[p].commit_from;
    }
    oi[OBS_N_NODES] = base_n, oi[OBS_N_EDGES] = base_e, oi[OBS_N_JOBS] = A, oi[OBS_N_SCHED] = h.n_sched;
    oi[OBS_NUM_COMMITTABLE] = ncommit, oi[OBS_SOURCE_JOB_IDX] = src_idx;
    oi[OBS_TERMINATED] = h.terminated, oi[OBS_ERR] = h.err;
    of[OBS_REWARD] = reward, of[OBS_WALL_TIME] = h.wall_time;
    g_hot.h.prof[4] += wave_clock() - t_obs0;
    g_hot.h.obs_n_nodes = base_n;
    g_hot.h.obs_graph_version = h.graph_version, g_hot.h.obs_n_edges = base_e, g_hot.h.obs_bind_gen = B.gen;
    g_hot.h.obs_n_sched = h.n_sched;
    g_hot.h.last_reward = reward;
    // SURVEY 8(d) algorithmic bytes of this step: k*140 + 12N + (12N + 4(A+1) + 4A + 8Ed + 12) + 26
    g_hot.h.model_bytes += (uint64_t)g_sc.events_this_step * 140u + 24u * (uint64_t)base_n + 4u * (uint64_t)(A + 1) +
                            4u * (uint64_t)A + 8u * (uint64_t)base_e + 12u + 26u;
  }
}
