// sss_decima.h - the Decima observation transform as ONE kernel over the env's observation buffers
// (SURVEY 8(f) next-1): DecimaObsWrapper.observation (reference schedulers/decima/env_wrapper.py:69-143),
// the DAG-layer edge masks of schedulers/decima/utils.py:238-267 and the batch collation of
// utils.py:117-204, written straight into the flat ("compact graph") arrays the GNN consumes.
//
// One wavefront per env. Inputs are the env's own observation rows (nodes, edge_links, dag_ptr,
// exec_supplies, the scalar block) plus the env's offsets into the flat outputs (exclusive prefix
// sums of the per-env counts, computed by the caller); nothing of the simulator state is touched,
// so the kernel is a pure function of the observation. LDS: 8 bytes per node slot
// (generation + receiver bits; layer-membership bits, then the out-edge range) + 8 per job slot (first node, supply).
//
// Included by sss_hip.hip (gfx950) and tests/emu/emu_backend.cpp (CPU wave emulator) after sss_sim.h.
#pragma once

#define SSS_LIST_SETS 32  // blocks of envs with their own list counters (sss_decima_graph_kernel)
struct SssDecimaArgs {
  const uint8_t* active;  // u8[B] or null
  const int64_t *node_off, *job_off, *edge_off;
  float num_tasks_scale, work_scale;
  float* x;
  int64_t *node_obs, *node_loc, *node_job, *sched_rank;
  int32_t* gen;
  uint32_t* node_recv;
  uint8_t* stage_mask;
  int64_t *src, *dst, *edge_obs;
  uint32_t* edge_layers;
  int64_t *job_obs, *job_cap, *job_first, *job_nodes;
  int64_t* out_start;  // i64[M] flat id of the node's first out-edge (edges are ordered by source node)
  int32_t* out_deg;    // i32[M] number of out-edges
  int32_t* obs_depth;
  int32_t* layer_cnt;   // i32[32][B]: number of receiving nodes of DAG layer l in env b
  const int64_t* sched_off;  // nullable, i64[B]: exclusive prefix of the envs' schedulable-stage counts ...
  int64_t* sched_list;       // ... and where the flat ids of the schedulable nodes go, env by env in node order
  // nullable: the layers' lists of receiving nodes written by this kernel itself. An env reserves its share of a list with one
  // fetch-add - on a counter it shares with its BLOCK of envs only (SSS_LIST_SETS blocks of consecutive envs): 4096 envs adding to
  // the same 9 addresses were 35 us of a 78 us launch (same-address atomics retire one at a time, ~6 ns each;
  // profiles/r05_graph_kernel.txt). Layer l's list is therefore up to SSS_LIST_SETS dense pieces: block s's piece starts at
  // recv_lists[l * recv_stride + node_off[first env of s]] (a block's receivers are among its own nodes: the pieces cannot
  // overlap) and has layer_totals[l * SSS_LIST_SETS + s] entries (i64[33][SSS_LIST_SETS], ZERO on entry; row 32: the largest node
  // count of an env of the block - what decides between a launch per layer and one launch, sss_host.h); the order of the envs
  // inside a piece varies from launch to launch (the layer launches do not care)
  int64_t* layer_totals;
  int64_t* recv_lists;
  int64_t* layer_totals_clear;  // nullable: another set of counters, zeroed by this launch (the caller's next launch reserves on it)
  int64_t recv_stride;
};

struct SssDecimaListArgs {
  const int64_t* node_off;   // i64[B]
  const int64_t* obs_nodes;  // i64[B] nodes the env contributed (0 for inactive envs)
  const uint32_t* node_recv; // u32[M]
  const int64_t* env_off;    // i64[32][B] exclusive prefix sums of layer_cnt along the env axis
  int64_t layer_base[32];    // start of layer l's list inside recv
  int64_t* recv;             // i64[sum of all counts]
  int n_layers;
  const int64_t* totals;     // nullable: i64[32] on the device, the lists' lengths - layer_base is then their running sum
};

SSS_SHARED_DYN(g_dec_lds);

SSS_KERNEL void sss_decima_graph_kernel(SssLayout L, SssBuffers B, int E, SssDecimaArgs d) {
  int env = wave_env(), lane = wave_lane();
  const int32_t* oi = B.obs_i32 + (size_t)env * SSS_OBS_I32;
  bool on = d.active == nullptr || d.active[env] != 0;
  int n = on ? oi[OBS_N_NODES] : 0, ne = on ? oi[OBS_N_EDGES] : 0, A = on ? oi[OBS_N_JOBS] : 0;
  if (d.layer_totals_clear)
    for (int i = env * 64 + lane; i < 33 * SSS_LIST_SETS; i += L.num_envs * 64) d.layer_totals_clear[i] = 0;
  if (n == 0) {  // wave-uniform
    if (lane == 0) d.obs_depth[env] = 0;
    if (lane < 32) d.layer_cnt[(size_t)lane * L.num_envs + env] = 0;
    return;
  }
  int ncommit = oi[OBS_NUM_COMMITTABLE], src_idx = oi[OBS_SOURCE_JOB_IDX];
  const float* nodes = B.nodes + (size_t)env * L.n_cap * 3;
  const int32_t* el = B.edge_links + (size_t)env * L.ed_cap * 2;
  const int32_t* dag_ptr = B.dag_ptr + (size_t)env * (L.J_cap + 1);
  const int32_t* sup = B.exec_supplies + (size_t)env * L.J_cap;
  int64_t n0 = d.node_off[env], j0 = d.job_off[env], e0 = d.edge_off[env];
  // two words per node slot (LDS decides how many envs a CU holds at once - at 200 jobs x 18 stage slots four words were 57.6 KB,
  // two envs per CU):  gr = generation in bits 0..7 (alone while the relaxation runs: its atomic max sees a plain integer), then the
  // receiver bits from bit 8 up (at most 24 layers: the host checks the longest path of the workload's job templates);  mb = the layer-membership bits, then
  // the node's out-edge range as two 16-bit halves (first edge | end << 16: edge slots are below 65536)
  uint32_t* gr = (uint32_t*)g_dec_lds;
  uint32_t* mb = gr + L.n_cap;
  // the env's job table next to them: every node looks its job up (a binary search over dag_ptr - in global memory that was four
  // dependent round trips per 64 nodes)
  int32_t* jp = (int32_t*)(g_dec_lds + (size_t)8 * L.n_cap);
  int32_t* js = jp + (L.J_cap + 1);
  for (int i = lane; i < n; i += 64) gr[i] = 0, mb[i] = 0;
  for (int a = lane; a <= A; a += 64) jp[a] = dag_ptr[a], js[a] = a < A ? sup[a] : 0;
  // the first 256 edges stay in registers (end points packed 16 + 16 bits: node slots are below 65536), the rest is re-read from the
  // observation where a phase needs it: the relaxation below walks the edge list once per DAG level, and re-reading it from
  // global memory every time was a third of the kernel (~45 dependent round trips per env at ~250 edges)
  uint32_t ev[4];
  for (int k = 0; k < 4; k++) {
    const int e = 64 * k + lane;
    ev[k] = e < ne ? ((uint32_t)el[2 * e] | ((uint32_t)el[2 * e + 1] << 16)) : 0u;
  }
  auto for_edges = [&](auto f) {
    for (int k = 0; k < 4; k++) {
      const int e = 64 * k + lane;
      if (e < ne) f(e, (int)(ev[k] & 0xFFFFu), (int)(ev[k] >> 16));
    }
    for (int e = 256 + lane; e < ne; e += 64) f(e, el[2 * e], el[2 * e + 1]);
  };
  wave_sync();
  // topological generations of the active subgraph (nx.topological_generations, utils.py:246-247):
  // longest-path relaxation over the edge list until nothing moves
  for (int it = 0; it <= n; it++) {
    bool moved = false;
    for_edges([&](int, int u, int v) {
      int gu = (int)gr[u] + 1;
      if ((int)gr[v] < gu) lane_atomic_max_i32((int32_t*)&gr[v], gu), moved = true;
    });
    wave_sync();
    if (!wave_ballot(moved)) break;
  }
  // membership bits: bit l of memb[i] <=> node i is in (generation l) U succ(generation l)
  for (int i = lane; i < n; i += 64) mb[i] = 1u << gr[i];
  wave_sync();
  for_edges([&](int, int u, int v) { lane_atomic_or_u32(&mb[v], 1u << gr[u]); });
  wave_sync();
  // edges: global endpoints, the layers whose mask holds the edge (both ends in the layer's node set)
  for_edges([&](int e, int u, int v) {
    uint32_t lay = mb[u] & mb[v];
    d.src[e0 + e] = n0 + u, d.dst[e0 + e] = n0 + v, d.edge_obs[e0 + e] = env;
    d.edge_layers[e0 + e] = lay;
    lane_atomic_or_u32(&gr[u], lay << 8);  // (nobody reads a generation in this phase; from here on it is gr & 0xFF)
  });
  wave_sync();
  // out-edge range of every node: edge_links is ordered by (job, source, destination)
  // (spark_sched_sim.py:249-258 + utils.subgraph keep the template's row-major edge order), so a
  // node's out-edges are contiguous; the membership bits are done and their words become the ranges (two 16-bit halves)
  uint16_t* rng = (uint16_t*)mb;  // rng[2 u] = first out-edge of u, rng[2 u + 1] = one past its last (0: none)
  for (int i = lane; i < n; i += 64) mb[i] = 0;
  wave_sync();
  {
    // (the neighbours of a register-resident edge come from the neighbouring lanes; wave-uniform: every lane takes part)
    uint32_t prev_last = 0;
    for (int k = 0; k < 4 && 64 * k < ne; k++) {
      const int e = 64 * k + lane;
      const uint32_t left = wave_bcast_u32(ev[k], (lane + 63) & 63), right = wave_bcast_u32(ev[k], (lane + 1) & 63);
      const uint32_t next_first = k < 3 ? wave_bcast_u32(ev[k + 1], 0) : (256 < ne ? (uint32_t)el[2 * 256] : 0u);
      if (e < ne) {
        const int u = (int)(ev[k] & 0xFFFFu);
        const int pu = lane > 0 ? (int)(left & 0xFFFFu) : (int)(prev_last & 0xFFFFu), nu = lane < 63 ? (int)(right & 0xFFFFu) : (int)(next_first & 0xFFFFu);
        if (e == 0 || pu != u) rng[2 * u] = (uint16_t)e;
        if (e == ne - 1 || nu != u) rng[2 * u + 1] = (uint16_t)(e + 1);
      }
      prev_last = wave_bcast_u32(ev[k], 63);
    }
    for (int e = 256 + lane; e < ne; e += 64) {
      int u = el[2 * e];
      if (el[2 * (e - 1)] != u) rng[2 * u] = (uint16_t)e;
      if (e == ne - 1 || el[2 * (e + 1)] != u) rng[2 * u + 1] = (uint16_t)(e + 1);
    }
  }
  wave_sync();
  // jobs: executor cap (env_wrapper.py:72-82), first node
  for (int a = lane; a < A; a += 64) {
    int gap = E - js[a];
    if (gap < 0) gap = 0;
    int cap = gap < ncommit ? gap : ncommit;
    if (a == src_idx) cap = ncommit;
    d.job_obs[j0 + a] = env, d.job_cap[j0 + a] = cap, d.job_first[j0 + a] = n0 + jp[a];
    d.job_nodes[j0 + a] = jp[a + 1] - jp[a];
  }
  // nodes: features (env_wrapper.py:110-143), job, schedulable rank, generation
  uint32_t run = 0, depth = 0;
  uint64_t lt = bit64(lane) - 1;
  for (int i0 = 0; i0 < n; i0 += 64) {
    int i = i0 + lane;
    bool sched = false;
    if (i < n) {
      // job slot of node i: last a with dag_ptr[a] <= i
      int lo = 0, hi = A;  // invariant: jp[lo] <= i < jp[hi]  (jp: the env's dag_ptr in LDS)
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (jp[mid] <= i) lo = mid; else hi = mid;
      }
      int a = lo;
      int supply = js[a];
      int gap = E - supply;
      if (gap < 0) gap = 0;
      int cap = gap < ncommit ? gap : ncommit;
      if (a == src_idx) cap = ncommit;
      float rem = nodes[3 * i], dur = nodes[3 * i + 1];
      sched = nodes[3 * i + 2] != 0.0f;
      float* x = d.x + (size_t)(n0 + i) * 5;
      x[0] = (float)((double)cap / (double)E);
      x[1] = a == src_idx ? 1.0f : -1.0f;
      x[2] = (float)((double)supply / (double)E);
      x[3] = rem / d.num_tasks_scale;
      x[4] = rem * dur / d.work_scale;
      d.node_obs[n0 + i] = env, d.node_loc[n0 + i] = i, d.node_job[n0 + i] = j0 + a;
      const uint32_t g_i = gr[i] & 0xFFu, first = mb[i] & 0xFFFFu, end = mb[i] >> 16;
      d.gen[n0 + i] = (int32_t)g_i, d.node_recv[n0 + i] = gr[i] >> 8, d.stage_mask[n0 + i] = sched;
      int deg = end > 0 ? (int)(end - first) : 0;
      d.out_start[n0 + i] = e0 + first, d.out_deg[n0 + i] = deg;
      if (g_i > depth) depth = g_i;
    }
    uint64_t bal = wave_ballot(sched);
    if (i < n) d.sched_rank[n0 + i] = sched ? (int64_t)(run + popc64(bal & lt)) : -1;
    if (sched && d.sched_list) d.sched_list[d.sched_off[env] + run + popc64(bal & lt)] = n0 + i;
    run += popc64(bal);
  }
  depth = ~wave_min_u32(~depth);
  if (lane == 0) d.obs_depth[env] = (int32_t)depth;
  // receiving nodes per layer (lane l counts layer l): sizes the per-layer lists / launches
  uint32_t cnt = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    uint32_t rv = i0 + lane < n ? gr[i0 + lane] >> 8 : 0u;
    for (uint32_t l = 0; l < depth; l++) {
      uint64_t bal = wave_ballot((rv >> l) & 1u);
      if ((uint32_t)lane == l) cnt += (uint32_t)popc64(bal);
    }
  }
  if (lane < 32) d.layer_cnt[(size_t)lane * L.num_envs + env] = (int32_t)cnt;
  if (d.recv_lists) {
    const int q = (L.num_envs + SSS_LIST_SETS - 1) / SSS_LIST_SETS, set = env / q;  // this env's block of envs
    const int64_t set_n0 = d.node_off[set * q];
    if (lane == 0) global_atomic_max_i64(d.layer_totals + 32 * SSS_LIST_SETS + set, (int64_t)n);  // row 32: the block's largest observation
    int64_t base = 0;
    if (lane < 32 && cnt) base = global_fetch_add_i64(d.layer_totals + lane * SSS_LIST_SETS + set, (int64_t)cnt);
    for (uint32_t l = 0; l < depth; l++) {
      int64_t pos = (int64_t)l * d.recv_stride + set_n0 + (int64_t)wave_readlane_u64((uint64_t)base, (int)l);
      for (int i0 = 0; i0 < n; i0 += 64) {
        const bool on = i0 + lane < n && ((gr[i0 + lane] >> (8 + l)) & 1u);
        const uint64_t bal = wave_ballot(on);
        if (on) d.recv_lists[pos + popc64(bal & lt)] = n0 + i0 + lane;
        pos += popc64(bal);
      }
    }
  }
}

// the receiving nodes of every DAG layer as index lists (layer l = recv[layer_base[l] ...), env by
// env in node order - what torch.nonzero over (node_recv >> l) & 1 would return
SSS_KERNEL void sss_decima_lists_kernel(int num_envs, SssDecimaListArgs d) {
  int env = wave_env(), lane = wave_lane();
  int n = (int)d.obs_nodes[env];
  if (n == 0) return;
  int64_t n0 = d.node_off[env];
  uint64_t lt = bit64(lane) - 1;
  int64_t base = 0;
  for (int l = 0; l < d.n_layers; l++) {
    int64_t pos = (d.totals ? base : d.layer_base[l]) + d.env_off[(size_t)l * num_envs + env];
    if (d.totals) base += d.totals[l];
    for (int i0 = 0; i0 < n; i0 += 64) {
      bool on = i0 + lane < n && ((d.node_recv[n0 + i0 + lane] >> l) & 1u);
      uint64_t bal = wave_ballot(on);
      if (on) d.recv[pos + popc64(bal & lt)] = n0 + i0 + lane;
      pos += popc64(bal);
    }
  }
}

// sss_bit_lists (include/sss.h): for every bit l < n_layers the ascending list of the positions e with bit l of bits[e] set - the
// DAG layers' edge lists and receiver lists of a batch graph from its per-edge / per-node layer masks (what torch.nonzero over
// (mask >> l) & 1 returns, layer after layer: decima/utils.py:249-267's edge_masks as index lists). One wave per chunk of
// `chunk` consecutive positions; pass 0 counts (cnt[chunk][l]), the caller scans the counts (sss_prefix_rows), pass 1 writes
// out[base[l] + off[l][chunk] + rank inside the chunk].
struct SssBitListArgs {
  const int32_t* bits;
  int64_t n;
  int32_t n_layers, chunk;  // chunk: a multiple of 64
  int32_t n_chunks, pass;
  int32_t* cnt;             // [n_chunks][n_layers]
  const int64_t* off;       // [n_layers][n_chunks]
  int64_t base[32];
  int64_t* out;
};
SSS_KERNEL void sss_bit_lists_kernel(SssBitListArgs a) {
  const int c = wave_env(), lane = wave_lane();
  const uint64_t lt = bit64(lane) - 1;
  const int64_t e0 = (int64_t)c * a.chunk;
  uint32_t run = 0;  // lane l: positions with bit l seen so far in this chunk
  for (int i0 = 0; i0 < a.chunk; i0 += 64) {
    const int64_t e = e0 + i0 + lane;
    const uint32_t v = e < a.n ? (uint32_t)a.bits[e] : 0u;
    if (wave_ballot(v != 0) == 0) continue;
    for (int l = 0; l < a.n_layers; l++) {
      const bool on = (v >> l) & 1u;
      const uint64_t bal = wave_ballot(on);
      if (a.pass == 1 && bal) {
        const uint32_t before = wave_readlane_u32(run, l);
        if (on) a.out[a.base[l] + a.off[(size_t)l * a.n_chunks + c] + before + popc64(bal & lt)] = e;
      }
      if (lane == l) run += (uint32_t)popc64(bal);
    }
  }
  if (a.pass == 0 && lane < a.n_layers) a.cnt[(size_t)c * a.n_layers + lane] = (int32_t)run;
}

// sss_prefix_rows (include/sss.h): one row's exclusive prefix sums, `tid` of `nt` cooperating threads; `part` is
// scratch for nt partial sums shared by them; `sync` orders the phases.
struct SssPrefixArgs {
  const int32_t* src;
  int64_t row_stride, col_stride;
  const uint8_t* mask;
  int n_rows, n_cols;
  int64_t *off, *cnt, *totals;
};
template <typename Sync>
SSS_DEV void prefix_row(const SssPrefixArgs& a, int row, int tid, int nt, int64_t* part, Sync sync) {
  const int per = (a.n_cols + nt - 1) / nt;  // a contiguous run of columns per thread
  const int c0 = tid * per, c1 = c0 + per < a.n_cols ? c0 + per : a.n_cols;
  int64_t sum = 0;
  for (int c = c0; c < c1; c++) sum += (a.mask && !a.mask[c]) ? 0 : (int64_t)a.src[row * a.row_stride + c * a.col_stride];
  part[tid] = sum;
  sync();
  int64_t base = 0;
  for (int t = 0; t < tid; t++) base += part[t];
  if (tid == nt - 1) a.totals[row] = base + sum;
  for (int c = c0; c < c1; c++) {
    const int64_t v = (a.mask && !a.mask[c]) ? 0 : (int64_t)a.src[row * a.row_stride + c * a.col_stride];
    a.off[(int64_t)row * a.n_cols + c] = base;
    if (a.cnt) a.cnt[(int64_t)row * a.n_cols + c] = v;
    base += v;
  }
}
