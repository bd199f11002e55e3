// sss_gnn_mfma.h - the GNN's 16 -> 32 -> 16 -> 16 MLPs on the matrix cores (gfx950 only; the emulator keeps sss_gnn.h).
//
// The DAG-layer launches are the largest share of a Decima step (nine launches, 219 of ~745 us of kernel time at 4096 envs,
// profiles/r03_bench.md), and in the 16-lanes-per-row form of sss_gnn16.h a row's Linear costs one DPP move, one LDS read and
// one FMA per weight. `v_mfma_f32_16x16x4_f32` (exact fp32 products, fp32 accumulation - the same arithmetic class, at twice
// the FMA rate of the vector unit and with no data movement between the layers) does a 16-row tile's Linear in a few
// instructions. One wave owns a tile of 16 rows; it computes Y^T = W X^T:
//   A operand (lane l: A[i = l & 15][k = l >> 4])   a weight W[16 t' + i][n(step, k)]        - loaded once per wave, in registers
//   B operand (lane l: B[k = l >> 4][j = l & 15])   the input X[row j][n(step, k)]
//   D (four registers r: D[4 (l >> 4) + r][j])      output neurons 16 t' + 4 q + r of row j, q = l >> 4
// A K-step carries any four input features as long as A and B agree; taking step (t, r) = features {16 t + 4 q + r, q = 0..3}
// makes register r of output tile t of one layer exactly the B operand of step (t, r) of the next: the layers chain with no
// shuffle, no LDS, no barrier. Rows enter the same way: lane (q, j) loads the float4 X[row j][4 q .. 4 q + 3] (16 features).
// Same packed parameters as everywhere ([W1, b1, W2^T, b2, W3, b3], sss_gnn.h); only the summation order inside a dot product
// differs from the other formulations - within the 2e-5 agreement the fixtures are held to.
#pragma once

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

SSS_DEV mfma_f4 mfma16(float a, float b, mfma_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
SSS_DEV mfma_f4 leaky4(mfma_f4 v, float slope) {
  mfma_f4 o;
  o.x = v.x > 0.0f ? v.x : v.x * slope, o.y = v.y > 0.0f ? v.y : v.y * slope, o.z = v.z > 0.0f ? v.z : v.z * slope, o.w = v.w > 0.0f ? v.w : v.w * slope;
  return o;
}

// the A operands and bias registers of one 16 -> 32 -> 16 -> 16 MLP for lane l (i = l & 15, q = l >> 4)
struct MfmaGnnMlp {
  float a1[2][4];  // [t'][r]  W1[16 t' + i][4 q + r]
  float a2[2][4];  // [t][r]   W2[i][16 t + 4 q + r]
  float a3[4];     // [r]      W3[i][4 q + r]
  mfma_f4 b1[2];   // [t'] . r b1[16 t' + 4 q + r]
  mfma_f4 b2, b3;  //          b2[4 q + r], b3[4 q + r]
  // `in_dim`: width of the MLP's first Linear; the 16-feature segment these A operands multiply is columns col0 .. col0 + n_col - 1
  // of it (features beyond n_col: zero weights)
  SSS_DEV void load(const float* __restrict__ w, int lane, int in_dim = 16, int col0 = 0, int n_col = 16) {
    const int i = lane & 15, q = lane >> 4;
    const float* W1 = w;
    const float* B1 = W1 + 32 * in_dim;
    const float* W2T = B1 + 32;  // [j][m] = W2[m][j]
    const float* B2 = W2T + 32 * 16;
    const float* W3 = B2 + 16;
    const float* B3 = W3 + 16 * 16;
    for (int t = 0; t < 2; t++)
      for (int r = 0; r < 4; r++) {
        a1[t][r] = 4 * q + r < n_col ? W1[(16 * t + i) * in_dim + col0 + 4 * q + r] : 0.0f;
        a2[t][r] = W2T[(16 * t + 4 * q + r) * 16 + i];
      }
    for (int r = 0; r < 4; r++) a3[r] = W3[i * 16 + 4 * q + r];
    for (int t = 0; t < 2; t++) b1[t] = mfma_f4{B1[16 * t + 4 * q], B1[16 * t + 4 * q + 1], B1[16 * t + 4 * q + 2], B1[16 * t + 4 * q + 3]};
    b2 = mfma_f4{B2[4 * q], B2[4 * q + 1], B2[4 * q + 2], B2[4 * q + 3]};
    b3 = mfma_f4{B3[4 * q], B3[4 * q + 1], B3[4 * q + 2], B3[4 * q + 3]};
  }
  // hidden-2 activations of the tile's rows (register r: neuron 4 q + r) for inputs x (register r: feature 4 q + r)
  SSS_DEV mfma_f4 hidden(mfma_f4 x, float slope) const {
    mfma_f4 d0 = b1[0], d1 = b1[1];
    d0 = mfma16(a1[0][0], x.x, d0), d1 = mfma16(a1[1][0], x.x, d1);
    d0 = mfma16(a1[0][1], x.y, d0), d1 = mfma16(a1[1][1], x.y, d1);
    d0 = mfma16(a1[0][2], x.z, d0), d1 = mfma16(a1[1][2], x.z, d1);
    d0 = mfma16(a1[0][3], x.w, d0), d1 = mfma16(a1[1][3], x.w, d1);
    d0 = leaky4(d0, slope), d1 = leaky4(d1, slope);
    // two accumulators (the dependent-issue latency of the instruction is longer than its issue interval), added at the end
    mfma_f4 e0 = b2, e1 = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
    e0 = mfma16(a2[0][0], d0.x, e0), e1 = mfma16(a2[1][0], d1.x, e1);
    e0 = mfma16(a2[0][1], d0.y, e0), e1 = mfma16(a2[1][1], d1.y, e1);
    e0 = mfma16(a2[0][2], d0.z, e0), e1 = mfma16(a2[1][2], d1.z, e1);
    e0 = mfma16(a2[0][3], d0.w, e0), e1 = mfma16(a2[1][3], d1.w, e1);
    return leaky4(e0 + e1, slope);
  }
  // the same with a second input segment (its A operands `s1`: another column range of the same W1)
  SSS_DEV mfma_f4 hidden2(mfma_f4 x, mfma_f4 xs, const float (&s1)[2][4], float slope) const {
    mfma_f4 d0 = b1[0], d1 = b1[1];
    d0 = mfma16(a1[0][0], x.x, d0), d1 = mfma16(a1[1][0], x.x, d1);
    d0 = mfma16(a1[0][1], x.y, d0), d1 = mfma16(a1[1][1], x.y, d1);
    d0 = mfma16(a1[0][2], x.z, d0), d1 = mfma16(a1[1][2], x.z, d1);
    d0 = mfma16(a1[0][3], x.w, d0), d1 = mfma16(a1[1][3], x.w, d1);
    d0 = mfma16(s1[0][0], xs.x, d0), d1 = mfma16(s1[1][0], xs.x, d1);
    d0 = mfma16(s1[0][1], xs.y, d0), d1 = mfma16(s1[1][1], xs.y, d1);
    d0 = mfma16(s1[0][2], xs.z, d0), d1 = mfma16(s1[1][2], xs.z, d1);
    d0 = mfma16(s1[0][3], xs.w, d0), d1 = mfma16(s1[1][3], xs.w, d1);
    d0 = leaky4(d0, slope), d1 = leaky4(d1, slope);
    mfma_f4 e0 = b2, e1 = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
    e0 = mfma16(a2[0][0], d0.x, e0), e1 = mfma16(a2[1][0], d1.x, e1);
    e0 = mfma16(a2[0][1], d0.y, e0), e1 = mfma16(a2[1][1], d1.y, e1);
    e0 = mfma16(a2[0][2], d0.z, e0), e1 = mfma16(a2[1][2], d1.z, e1);
    e0 = mfma16(a2[0][3], d0.w, e0), e1 = mfma16(a2[1][3], d1.w, e1);
    return leaky4(e0 + e1, slope);
  }
  // the last Linear applied to (a sum of `count` per row) hidden-2 vector(s): W3 . v + count * b3
  SSS_DEV mfma_f4 out(mfma_f4 v, float count) const {
    mfma_f4 f = b3 * count;
    f = mfma16(a3[0], v.x, f), f = mfma16(a3[1], v.y, f), f = mfma16(a3[2], v.z, f), f = mfma16(a3[3], v.w, f);
    return f;
  }
};

// LAYER (sss_gnn.h): tmp[r] = h_init[r] + update( sum over r's out-edges e in DAG layer l of msg(h[dst_e]) ), 16 receiving
// nodes per wave and pass. Out-edge slot k of all 16 rows goes through the message MLP together (rows without such an edge
// contribute zero), so a tile costs 16 MFMAs per slot + 24 for the aggregate and the update MLP.
// One tile of DAG layer `layer`: lane (q, j) names row j's node `n` (-1: no row).
SSS_DEV void gnn_layer_mfma_rows(const SssGnnArgs& a, const MfmaGnnMlp& msg, const MfmaGnnMlp& upd, int64_t n, int layer, int lane) {
  const int q = lane >> 4;
  const uint32_t above = layer >= 31 ? 0u : ~((2u << layer) - 1u);
  const bool valid = n >= 0;
  const int64_t e0 = valid ? a.out_start[n] : 0;
  const int deg = valid ? a.out_deg[n] : 0;
  mfma_f4 acc = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  float used = 0.0f;
  for (int k = 0; __builtin_amdgcn_ballot_w64(k < deg) != 0; k++) {
    const bool use = k < deg && ((a.edge_layers[e0 + k] >> layer) & 1u);
    mfma_f4 x = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (use) {
      const int64_t c = a.dst[e0 + k];
      const float* cur = (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[c] & above) & 1)) ? a.tmp : a.h;
      x = *(const mfma_f4*)(cur + c * 16 + 4 * q);
    }
    const mfma_f4 h2 = msg.hidden(x, a.slope);
    if (use) acc += h2, used += 1.0f;
  }
  const mfma_f4 agg = msg.out(acc, used);
  const mfma_f4 y = upd.out(upd.hidden(agg, a.slope), 1.0f);
  if (valid) {
    float* nxt = (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[n] & above) & 1)) ? a.h : a.tmp;
    const mfma_f4 hi = *(const mfma_f4*)(a.h_init + n * 16 + 4 * q);
    *(mfma_f4*)(nxt + n * 16 + 4 * q) = hi + y;
  }
}
// The same tile with its loads batched - what a wave of a layer launch waits for is the chain of dependent loads, so three
// round trips instead of three per out-edge slot: (a) everything that depends on the row's node alone, (b) the layer bits and
// end points of up to four out-edge slots at once, (c) the children's receiver bits together with BOTH copies of their
// embeddings (the right one is picked when the bits are there). Needs node_recv; same arithmetic in the same order -
// bit-identical embeddings. 18.7 -> 14.7 us per launch at 4096 envs (profiles/r05_layers_per_observation.txt, which also has
// the measured dead end: all layers in one launch with a wave per observation).
SSS_DEV void gnn_layer_mfma_rows_batched(const SssGnnArgs& a, const MfmaGnnMlp& msg, const MfmaGnnMlp& upd, int64_t n, int layer, int lane) {
  const int q = lane >> 4;
  const uint32_t above = layer >= 31 ? 0u : ~((2u << layer) - 1u);
  const bool valid = n >= 0;
  const int64_t nn = valid ? n : 0;
  const int64_t e0 = a.out_start[nn];
  const int deg = valid ? a.out_deg[nn] : 0;
  const uint32_t rv = (uint32_t)a.node_recv[nn];
  const mfma_f4 hi = *(const mfma_f4*)(a.h_init + nn * 16 + 4 * q);
  mfma_f4 acc = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  float used = 0.0f;
  for (int k0 = 0; __builtin_amdgcn_ballot_w64(k0 < deg) != 0; k0 += 4) {
    bool use[4];
    int64_t c[4];
    uint32_t el[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool in = k0 + u < deg;
      const int64_t e = in ? e0 + k0 + u : 0;
      el[u] = in ? a.edge_layers[e] : 0u;
      c[u] = in ? a.dst[e] : nn;
    }
    uint32_t rc[4];
    mfma_f4 xh[4], xt[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      use[u] = (el[u] >> layer) & 1u;
      if (!use[u]) c[u] = nn;  // (a row of this observation: both copies readable, nothing of it used)
      rc[u] = (uint32_t)a.node_recv[c[u]];
      xh[u] = *(const mfma_f4*)(a.h + c[u] * 16 + 4 * q);
      xt[u] = *(const mfma_f4*)(a.tmp + c[u] * 16 + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (__builtin_amdgcn_ballot_w64(use[u]) == 0) continue;  // (no row of the tile has such an edge: nothing would be added)
      const mfma_f4 zero = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
      const mfma_f4 x = use[u] ? ((__builtin_popcount(rc[u] & above) & 1) ? xt[u] : xh[u]) : zero;
      const mfma_f4 h2 = msg.hidden(x, a.slope);
      if (use[u]) acc += h2, used += 1.0f;
    }
  }
  const mfma_f4 agg = msg.out(acc, used);
  const mfma_f4 y = upd.out(upd.hidden(agg, a.slope), 1.0f);
  if (valid) {
    float* nxt = (__builtin_popcount(rv & above) & 1) ? a.h : a.tmp;
    *(mfma_f4*)(nxt + n * 16 + 4 * q) = hi + y;
  }
}

// One tile: rows idx0[first + j], j < count (count <= 16), of DAG layer `layer`.
SSS_DEV void gnn_layer_mfma_tile(const SssGnnArgs& a, const MfmaGnnMlp& msg, const MfmaGnnMlp& upd, const int64_t* idx0, int64_t first, int count, int layer,
                                  int lane) {
  const int j = lane & 15;
  const int64_t n = j < count ? idx0[first + j] : -1;
  if (a.node_recv) gnn_layer_mfma_rows_batched(a, msg, upd, n, layer, lane);  // (the same arithmetic with the loads of a tile batched)
  else gnn_layer_mfma_rows(a, msg, upd, n, layer, lane);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void sss_gnn_layer_mfma_kernel(SssGnnArgs a) {
  if (a.list_q) {  // the graph kernel's lists: a dense piece per block of observations
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    GnnListPieces pc;
    pc.load(a, lane);
    if ((int)blockIdx.x * 4 + wave >= pc.total) return;  // (before the parameters are fetched)
    MfmaGnnMlp msg, upd;
    msg.load(a.w, lane), upd.load(a.w2, lane);
    // (the next tile's row ids are asked for before this tile is worked on: one of a tile's four dependent round trips less; also
    // fetching ahead what depends on the next rows' nodes alone was slower again - 0.5495 -> 0.5558 ms per step)
    const int j = lane & 15, step = (int)gridDim.x * 4;
    int64_t first;
    int count;
    int t = (int)blockIdx.x * 4 + wave;
    pc.tile_uniform(t, lane, first, count);
    int64_t n = j < count ? a.idx0[first + j] : -1;
    while (true) {
      const int t_next = t + step;
      int64_t n_next = -1;
      if (t_next < pc.total) {
        pc.tile_uniform(t_next, lane, first, count);
        if (j < count) n_next = a.idx0[first + j];
      }
      gnn_layer_mfma_rows_batched(a, msg, upd, n, a.layer, lane);
      if (t_next >= pc.total) break;
      t = t_next, n = n_next;
    }
    return;
  }
  if (a.layer_totals) {  // list length and position from the device (sss_gnn_encode)
    int64_t off = (int64_t)a.layer * a.idx0_stride;
    if (a.idx0_stride == 0)
      for (int l = 0; l < a.layer; l++) off += a.layer_totals[l];
    a.n_rows = a.layer_totals[a.layer], a.idx0 += off;
    a.n_rows_dev = nullptr;
    if ((int64_t)blockIdx.x * 64 >= a.n_rows) return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  MfmaGnnMlp msg, upd;
  msg.load(a.w, lane), upd.load(a.w2, lane);
  if (a.n_rows_dev) a.n_rows = *a.n_rows_dev;  // (the row count of a step without a host round trip)
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t left = a.n_rows - tile * 16;
    gnn_layer_mfma_tile(a, msg, upd, a.idx0, tile * 16, left < 16 ? (int)left : 16, a.layer, lane);
  }
}

// how many workgroups of a kernel the device holds at once (occupancy x CUs) - the grid of a launch whose waves stride over tiles;
// `fallback` if the runtime will not say. Cached per launch site AND per device (a process may drive several GPUs: the answer of the
// device that happened to be current at the first launch is not the answer for the others), and a failed query is not cached (the
// optimum is sharp - 768 workgroups for the layers, profiles/r05_hints.txt - so a remembered fallback would be a silent slow-down for
// the life of the process).
struct GnnGridCap {
  int64_t per_device[16] = {0};  // 0: not known yet
};
static int64_t gnn_resident_workgroups(GnnGridCap& cache, const void* kernel, int threads, size_t lds, int64_t fallback = 1024) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fallback;
  const bool slot = dev >= 0 && dev < 16;
  if (slot) {
    const int64_t known = __atomic_load_n(&cache.per_device[dev], __ATOMIC_RELAXED);
    if (known > 0) return known;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) return fallback;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return fallback;
  const int64_t n = (int64_t)per_cu * cus;
  if (slot) __atomic_store_n(&cache.per_device[dev], n, __ATOMIC_RELAXED);
  return n;
}
static int gnn_layer_mfma_launch(const SssGnnArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  const int64_t wgs = (a.n_rows + 63) / 64;  // four tiles of 16 rows per workgroup and pass
  // at most ONE resident set of workgroups, each wave striding over its tiles with the parameters loaded once: measured at 4096 envs
  // (profiles/r05_hints.txt) 768 workgroups - exactly what fits at this kernel's register count - beat 512 and 1024 as well as 2048
  static GnnGridCap cache;
  const int64_t cap = gnn_resident_workgroups(cache, (const void*)sss_gnn_layer_mfma_kernel, 256, 0);
  const unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
  hipLaunchKernelGGL(sss_gnn_layer_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// ALL the DAG layers of a pass in ONE launch, one wave per OBSERVATION - for batches of SMALL observations. Message passing never
// leaves an observation (edges join nodes of one job), so the only ordering the layers need is inside an observation, which a wave
// has for free: no launch boundary between layers. A row's arithmetic does not depend on which rows share its tile, so the embeddings
// are bit-identical to the launch-per-layer path; they stay in global memory (`h` / `tmp` alternate per update exactly as above),
// a wave reads back what it wrote itself, ordered by a workgroup-scope fence between layers. The receiving nodes of each layer are
// compacted over the whole observation first (one pass over its receiver bits, 16-bit positions in LDS, lengths from the graph
// kernel's per-observation counts). The wave walks its tiles alone, ~5 us each, and the launch ends with its largest observation:
// at ~40 nodes per observation (a PPO collection at config 5: 1024 envs, nine launches of ~8 us each, all at the launch floor)
// that is a third of the nine launches; at ~200 nodes per observation (config 2 in steady state, 17 tiles per wave) it is slower
// than they are (profiles/r05_layers_per_observation.txt) - sss_gnn_encode chooses by the nodes per observation.
#ifndef GNN_OBS_LIST_CAP  // (a test build sets it low to reach the chunk-by-chunk path: tests/gpu_variant.py obscap64)
#define GNN_OBS_LIST_CAP 1024  // list entries (node, layer) per observation kept in LDS; a larger observation compacts chunk by chunk
#endif
__global__ __launch_bounds__(256) void sss_gnn_layers_obs_kernel(SssGnnArgs a, const int64_t* __restrict__ obs_node_off, const int64_t* __restrict__ obs_nodes,
                                                                 const int32_t* __restrict__ layer_cnt, int n_obs, int max_depth) {
  __shared__ uint16_t lists[4][GNN_OBS_LIST_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int obs = (int)blockIdx.x * 4 + wave;
  if (obs >= n_obs) return;
  const int cnt = (int)obs_nodes[obs];
  if (cnt == 0) return;
  const int64_t n0 = obs_node_off[obs];
  // lane l: length and start of layer l's list
  const int len = lane < max_depth ? layer_cnt[(size_t)lane * n_obs + obs] : 0;
  int off = len;
  for (int s = 1; s < 32; s <<= 1) {
    const int t = __shfl_up(off, s);
    if (lane >= s) off += t;
  }
  const int total = __shfl(off, 31);
  off -= len;
  if (total == 0) return;
  MfmaGnnMlp msg, upd;
  msg.load(a.w, lane), upd.load(a.w2, lane);
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t depth_mask = max_depth < 32 ? (1u << max_depth) - 1u : ~0u;  // (the per-layer launches stop at max_depth as well)
  if (total <= GNN_OBS_LIST_CAP && cnt <= 65536) {
    uint16_t* list = lists[wave];
    int run = off;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const uint32_t rv = c0 + lane < cnt ? (uint32_t)a.node_recv[n0 + c0 + lane] & depth_mask : 0u;
      if (__builtin_amdgcn_ballot_w64(rv != 0) == 0) continue;
      for (int l = 0; l < max_depth; l++) {
        const bool on = (rv >> l) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(on);
        if (m == 0) continue;
        const int base = __builtin_amdgcn_readlane(run, l);
        if (on) list[base + __builtin_popcountll(m & lt)] = (uint16_t)(c0 + lane);
        if (lane == l) run += __builtin_popcountll(m);
      }
    }
    __threadfence_block();
    for (int l = max_depth - 1; l >= 0; l--) {
      const int c = __builtin_amdgcn_readlane(len, l);
      if (c == 0) continue;
      const int o = __builtin_amdgcn_readlane(off, l);
      for (int t = 0; 16 * t < c; t++) {
        const int r = 16 * t + (lane & 15);
        gnn_layer_mfma_rows_batched(a, msg, upd, r < c ? n0 + list[o + r] : -1, l, lane);
      }
      __threadfence_block();  // the next layer reads what this one wrote (same wave, other lanes)
    }
    return;
  }
  // an observation whose lists do not fit: the receiving nodes of a layer compacted chunk by chunk (one ballot + one ds_permute)
  uint32_t layers = 0;
  for (int l = 0; l < max_depth; l++)
    if (__builtin_amdgcn_readlane(len, l)) layers |= 1u << l;
  while (layers) {
    const int l = 31 - __builtin_clz(layers);
    layers &= ~(1u << l);
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const bool on = c0 + lane < cnt && (((uint32_t)a.node_recv[n0 + c0 + lane] >> l) & 1u);
      const uint64_t m = __builtin_amdgcn_ballot_w64(on);
      if (m == 0) continue;
      const int n_on = __builtin_popcountll(m);
      // lane r < n_on receives the chunk position of the chunk's r-th receiving node (a permutation of the lanes: the others go behind)
      const int to = on ? __builtin_popcountll(m & lt) : n_on + __builtin_popcountll(~m & lt);
      const int sel = __builtin_amdgcn_ds_permute(to << 2, lane);
      for (int t = 0; 16 * t < n_on; t++) {
        const int r = 16 * t + (lane & 15);
        const int pos = __shfl(sel, r);
        gnn_layer_mfma_rows_batched(a, msg, upd, r < n_on ? n0 + c0 + pos : -1, l, lane);
      }
    }
    __threadfence_block();
  }
}
static int gnn_layers_obs_launch(const SssGnnArgs& a, const int64_t* obs_node_off, const int64_t* obs_nodes, const int32_t* layer_cnt, int n_obs, int max_depth,
                                 void* stream) {
  hipLaunchKernelGGL(sss_gnn_layers_obs_kernel, dim3((unsigned)((n_obs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, obs_node_off, obs_nodes, layer_cnt, n_obs,
                     max_depth);
  return (int)hipGetLastError();
}

// tanh(v) = 1 - 2 / (1 + e^(2v)) on the transcendental unit (v_exp_f32, v_rcp_f32): absolute error ~1e-7, saturates
// correctly at both ends; the library tanhf costs ~40 instructions, and a tile of a policy head needs 32 per lane
SSS_DEV float fast_tanh(float v) { return 1.0f - __fdividef(2.0f, 1.0f + __expf(2.0f * v)); }

// ---- the two policy heads (IN -> 64 -> 64 -> 1, Tanh) -------------------------------------------------------------------
// Same chaining; the 2 x 4096 weights of the first two Linears do not fit a wave's registers, so their A-operand images sit
// in LDS in [step][lane] order (one conflict-free 4-byte read per MFMA). The input row is U segments of 16 features, each a
// float4 per lane from its own source row (node embedding, job summary, observation summary, raw features): `col(u, f)` names
// the column of W1 that feature f of segment u multiplies (-1: padding). The last Linear has one output: 16 FMAs per lane on
// the vector unit and a sum over the four lanes of a row.
template <int U>
struct MfmaHead {
  static constexpr int NS = 4 * U;                 // K-steps of the first Linear
  static constexpr int A1 = 0, A2 = A1 + 4 * NS * 64, B1 = A2 + 4 * 16 * 64, B2 = B1 + 64, TOTAL = B2 + 64;
  template <typename Col>
  SSS_DEV static void stage(float* lds, const float* __restrict__ w, int in_dim, int tid, int nthreads, Col col) {
    const float* W1 = w;
    const float* gB1 = W1 + 64 * in_dim;
    const float* W2T = gB1 + 64;  // [n][m] = W2[m][n]
    const float* gB2 = W2T + 64 * 64;
    for (int t = tid; t < 4 * NS * 64; t += nthreads) {
      const int lane = t & 63, s = (t >> 6) % NS, tp = t / (64 * NS);
      const int c = col(s >> 2, 4 * (lane >> 4) + (s & 3));
      lds[A1 + t] = c >= 0 ? W1[(16 * tp + (lane & 15)) * in_dim + c] : 0.0f;
    }
    for (int t = tid; t < 4 * 16 * 64; t += nthreads) {
      const int lane = t & 63, s = (t >> 6) & 15, tp = t >> 10;
      lds[A2 + t] = W2T[(16 * (s >> 2) + 4 * (lane >> 4) + (s & 3)) * 64 + 16 * tp + (lane & 15)];
    }
    for (int t = tid; t < 64; t += nthreads) lds[B1 + t] = gB1[t], lds[B2 + t] = gB2[t];
  }
  // the two hidden layers' activations of the tile's rows (d[t] / e[t], register r: neuron 16 t + 4 q + r) from the input
  // segments x[u] (register r: feature 4 q + r of segment u)
  SSS_DEV static void hidden(const float* lds, const mfma_f4 (&x)[U], int lane, mfma_f4 (&d)[4], mfma_f4 (&e)[4]) {
    const int q = lane >> 4;
#pragma unroll
    for (int tp = 0; tp < 4; tp++) {
      d[tp] = *(const mfma_f4*)(lds + B1 + 16 * tp + 4 * q);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const float* ap = lds + A1 + ((tp * NS + 4 * u) * 64) + lane;
        d[tp] = mfma16(ap[0], x[u].x, d[tp]), d[tp] = mfma16(ap[64], x[u].y, d[tp]);
        d[tp] = mfma16(ap[128], x[u].z, d[tp]), d[tp] = mfma16(ap[192], x[u].w, d[tp]);
      }
    }
#pragma unroll
    for (int tp = 0; tp < 4; tp++) d[tp] = mfma_f4{fast_tanh(d[tp].x), fast_tanh(d[tp].y), fast_tanh(d[tp].z), fast_tanh(d[tp].w)};
#pragma unroll
    for (int tp = 0; tp < 4; tp++) {
      e[tp] = *(const mfma_f4*)(lds + B2 + 16 * tp + 4 * q);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const float* ap = lds + A2 + ((tp * 16 + 4 * t) * 64) + lane;
        e[tp] = mfma16(ap[0], d[t].x, e[tp]), e[tp] = mfma16(ap[64], d[t].y, e[tp]);
        e[tp] = mfma16(ap[128], d[t].z, e[tp]), e[tp] = mfma16(ap[192], d[t].w, e[tp]);
      }
    }
#pragma unroll
    for (int tp = 0; tp < 4; tp++) e[tp] = mfma_f4{fast_tanh(e[tp].x), fast_tanh(e[tp].y), fast_tanh(e[tp].z), fast_tanh(e[tp].w)};
  }
  // the row's score (on every lane of the row's four): the last Linear has one output - 16 FMAs per lane and a sum over the row's lanes
  SSS_DEV static float score(const float* lds, const mfma_f4 (&x)[U], const float (&w3)[16], float b3, int lane) {
    mfma_f4 d[4], e[4];
    hidden(lds, x, lane, d, e);
    return reduce(e, w3, b3);
  }
  SSS_DEV static float reduce(const mfma_f4 (&e)[4], const float (&w3)[16], float b3) {
    float p = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) p += w3[4 * t] * e[t].x, p += w3[4 * t + 1] * e[t].y, p += w3[4 * t + 2] * e[t].z, p += w3[4 * t + 3] * e[t].w;
    p += __shfl_xor(p, 16), p += __shfl_xor(p, 32);
    return p + b3;
  }
};

// STAGE: score[obs(n), loc(n)] = stage([x, h, h_dag[job], h_glob[obs]] of schedulable node n) - segments h, h_dag, h_glob, x
// EXEC:  score[b, c] = exec([x[first(j), :3], h_dag[j], h_glob[obs(j)], c / E]) - segments h_dag, h_glob, (x0, x1, x2, c / E)
template <int KIND>
__global__ __launch_bounds__(256) void sss_gnn_head_mfma_kernel(SssGnnArgs a) {
  constexpr int U = KIND == GNN_STAGE ? 4 : 3;
  constexpr int IN = KIND == GNN_STAGE ? GNN_NF + 48 : GNN_DF + 33;
  using H = MfmaHead<U>;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  if (a.w2_16) {  // the host has the images ready (decima.py `_mfma_head_image`, the layout of MfmaHead::stage): a straight copy
    static_assert(H::TOTAL % 4 == 0, "images are copied 16 bytes at a time");
    // (all of a thread's loads first, then its LDS writes: written as one loop the compiler waits for every load before the next
    // is issued - eight to nine dependent round trips at the head of every workgroup)
    constexpr int N4 = H::TOTAL / 4, IT = (N4 + 255) / 256;
    float4 img[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int t = (int)threadIdx.x + 256 * i;
      img[i] = ((const float4*)a.w2_16)[t < N4 ? t : N4 - 1];
    }
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int t = (int)threadIdx.x + 256 * i;
      if (t < N4) ((float4*)w_lds)[t] = img[i];
    }
  } else
    H::stage(w_lds, a.w, IN, threadIdx.x, 256, [](int u, int f) {
      if (KIND == GNN_STAGE) return u == 0 ? GNN_NF + f : u == 1 ? GNN_NF + 16 + f : u == 2 ? GNN_NF + 32 + f : (f < GNN_NF ? f : -1);
      return u == 0 ? GNN_DF + f : u == 1 ? GNN_DF + 16 + f : (f < GNN_DF ? f : (f == GNN_DF ? GNN_DF + 32 : -1));
    });
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float* W3 = a.w + 64 * IN + 64 + 64 * 64 + 64;
  float w3[16];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) w3[4 * t + r] = W3[16 * t + 4 * q + r];
  const float b3 = W3[64];
  if (a.n_rows_dev) a.n_rows = *a.n_rows_dev;  // (the row count of a step without a host round trip)
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t r = tile * 16 + j;
    const mfma_f4 zero = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
    mfma_f4 x[U];
    if (KIND == GNN_STAGE) {
      const int64_t n = r < a.n_rows ? a.idx0[r] : -1;
      const bool valid = n >= 0;
      if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;  // (a padded list: nothing but -1 from some row on)
      x[0] = valid ? *(const mfma_f4*)(a.h + n * 16 + 4 * q) : zero;
      x[1] = valid ? *(const mfma_f4*)(a.h_dag + a.node_job[n] * 16 + 4 * q) : zero;
      x[2] = valid ? *(const mfma_f4*)(a.h_glob + a.node_obs[n] * 16 + 4 * q) : zero;
      x[3] = zero;
      if (valid && q == 0) x[3] = mfma_f4{a.x[n * GNN_NF], a.x[n * GNN_NF + 1], a.x[n * GNN_NF + 2], a.x[n * GNN_NF + 3]};
      if (valid && q == 1) x[3].x = a.x[n * GNN_NF + 4];
      const float v = H::score(w_lds, x, w3, b3, lane);
      if (valid && q == 0) a.out[a.node_obs[n] * a.n_pad + a.node_loc[n]] = v;
    } else {
      const bool valid = r < a.n_rows;
      const int64_t b = valid ? r / a.E : 0;
      const int c = (int)(r - b * a.E);
      const int64_t jj = a.idx0[b];
      // a tile whose rows are all executor counts beyond their job's cap (a job allows ~7 of 50 counts at config 5): -inf, no network
      if (__builtin_amdgcn_ballot_w64(valid && c < a.job_cap[jj]) == 0) {
        if (valid && q == 0) a.out[r] = -__builtin_inff();
        continue;
      }
      x[0] = valid ? *(const mfma_f4*)(a.h_dag + jj * 16 + 4 * q) : zero;
      x[1] = valid ? *(const mfma_f4*)(a.h_glob + a.job_obs[jj] * 16 + 4 * q) : zero;
      x[2] = zero;
      if (valid && q == 0) {
        const float* xr = a.x + a.job_first[jj] * GNN_NF;
        x[2] = mfma_f4{xr[0], xr[1], xr[2], (float)c / (float)a.E};
      }
      const float v = H::score(w_lds, x, w3, b3, lane);
      if (valid && q == 0) a.out[r] = c < a.job_cap[jj] ? v : -__builtin_inff();
    }
  }
}

template <int KIND>
static int gnn_head_mfma_launch(const SssGnnArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  constexpr int U = KIND == GNN_STAGE ? 4 : 3;
  const int64_t wgs = (a.n_rows + 63) / 64;
  // (33 KB of LDS images per workgroup, staged once and reused over its tiles: one resident set of workgroups)
  static GnnGridCap cache;
  const int64_t hcap = gnn_resident_workgroups(cache, (const void*)sss_gnn_head_mfma_kernel<KIND>, 256, (size_t)MfmaHead<U>::TOTAL * sizeof(float));
  const unsigned grid = (unsigned)(wgs < hcap ? wgs : hcap);
  hipLaunchKernelGGL(sss_gnn_head_mfma_kernel<KIND>, dim3(grid), dim3(256), (size_t)MfmaHead<U>::TOTAL * sizeof(float), (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// ---- node / job rows ----------------------------------------------------------------------------------------------------
// the row's raw features x[n][0..4] as a 16-feature segment: lane (q, j) holds features 4 q .. 4 q + 3
SSS_DEV mfma_f4 mfma_x_segment(const float* __restrict__ x, int64_t n, int q, bool valid) {
  mfma_f4 v = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
  if (valid && q == 0) v = mfma_f4{x[n * GNN_NF], x[n * GNN_NF + 1], x[n * GNN_NF + 2], x[n * GNN_NF + 3]};
  if (valid && q == 1) v.x = x[n * GNN_NF + 4];
  return v;
}

// PREP + SINK (sss_gnn.h): h_init[n] = prep(x[n]); h[n] = is_parent[n] ? 0 : update(h_init[n]) (h_init[n] for observations with a
// single DAG layer) - the update MLP takes h_init straight from the registers the prep MLP left it in.
// DAGHID: tmp[n] = hidden part of dag([x[n], h[n]]), with the MERGE of embeddings that ended up in tmp.
// GLOBHID: tmp[j] = hidden part of glob(h_dag[j]).
template <int KIND>
__global__ __launch_bounds__(256) void sss_gnn_rows_mfma_kernel(SssGnnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  MfmaGnnMlp m0, m1;
  float s1[2][4];
  if (KIND == GNN_PREP) m0.load(a.w, lane, GNN_NF, 0, GNN_NF), m1.load(a.w2, lane);
  if (KIND == GNN_DAGHID) {
    m0.load(a.w, lane, GNN_NF + 16, GNN_NF, 16);  // the embedding's columns of the first Linear ...
    m1.load(a.w, lane, GNN_NF + 16, 0, GNN_NF);   // ... and the raw features' (only their first-Linear operands are used)
    for (int t = 0; t < 2; t++)
      for (int r = 0; r < 4; r++) s1[t][r] = m1.a1[t][r];
  }
  if (KIND == GNN_GLOBHID) m0.load(a.w, lane);
  if (a.n_rows_dev) a.n_rows = *a.n_rows_dev;  // (the row count of a step without a host round trip)
  const int64_t n_tiles = (a.n_rows + 15) / 16;
  // (asking for the NEXT tile's inputs before this tile's matrix-core work was tried: 0.5495 -> 0.5533 ms per step, more registers)
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t n = tile * 16 + j;
    const bool valid = n < a.n_rows;
    if (KIND == GNN_PREP) {
      const mfma_f4 hi = m0.out(m0.hidden(mfma_x_segment(a.x, n, q, valid), a.slope), 1.0f);
      mfma_f4 h = m1.out(m1.hidden(hi, a.slope), 1.0f);
      if (valid) {
        *(mfma_f4*)(a.out + n * 16 + 4 * q) = hi;
        const bool par = a.out_deg[n] != 0;
        const bool skip = a.obs_depth != nullptr && a.obs_depth[a.node_obs[n]] == 0;
        if (skip) h = hi;
        else if (par) h = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
        *(mfma_f4*)(a.h + n * 16 + 4 * q) = h;
      }
    } else if (KIND == GNN_DAGHID) {
      mfma_f4 hv = mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
      if (valid) {
        if (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[n]) & 1)) {
          hv = *(const mfma_f4*)(a.tmp + n * 16 + 4 * q);  // MERGE: an odd number of updates left the embedding in tmp
          *(mfma_f4*)(a.h + n * 16 + 4 * q) = hv;
        } else
          hv = *(const mfma_f4*)(a.h + n * 16 + 4 * q);
      }
      const mfma_f4 h2 = m0.hidden2(hv, mfma_x_segment(a.x, n, q, valid), s1, a.slope);
      if (valid) *(mfma_f4*)(a.tmp + n * 16 + 4 * q) = h2;
    } else {
      const mfma_f4 xv = valid ? *(const mfma_f4*)(a.h_dag + n * 16 + 4 * q) : mfma_f4{0.0f, 0.0f, 0.0f, 0.0f};
      const mfma_f4 h2 = m0.hidden(xv, a.slope);
      if (valid) *(mfma_f4*)(a.tmp + n * 16 + 4 * q) = h2;
    }
  }
}

template <int KIND>
static int gnn_rows_mfma_launch(const SssGnnArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  const int64_t wgs = (a.n_rows + 63) / 64;
  static GnnGridCap cache;
  const int64_t cap = gnn_resident_workgroups(cache, (const void*)sss_gnn_rows_mfma_kernel<KIND>, 256, 0);  // (one resident set, as for the layers)
  const unsigned grid = (unsigned)(wgs < cap ? wgs : cap);
  hipLaunchKernelGGL(sss_gnn_rows_mfma_kernel<KIND>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
