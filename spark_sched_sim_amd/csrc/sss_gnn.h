// sss_gnn.h - Decima's GNN forward pass (inference) as a handful of row-parallel kernels
// (reference schedulers/decima/scheduler.py:142-385). Every MLP of the
// published architecture (config/decima_tpch.yaml:66-78: embed 16, GNN MLPs [32,16] LeakyReLU,
// policy MLPs [64,64] Tanh) is evaluated whole - Linear/act/Linear/act/Linear - by ONE thread per
// row, with the gather that builds the row's input and the scatter / segment-sum that consumes its
// output fused in:
//
//   PREP    h_init[n] = prep(x[n])                                            (scheduler.py:200)
//           (given the update MLP as `w2` and `h`, the same launch also does SINK for the row)
//   SINK    h[n]      = is_parent[n] ? 0 : update(h_init[n])                  (scheduler.py:206-209)
//                       (h_init[n] instead where the node's observation has a single DAG layer, :196-198)
//   LAYER   tmp[r]    = h_init[r] + update( sum over r's out-edges e in DAG layer l of msg(h[dst_e]) )
//                       for the layer's receiving nodes r                       (scheduler.py:214-232)
//   COMMIT  h[r]      = tmp[r]   (a layer reads the previous h everywhere before any node moves)
//   With `node_recv` given (bit l of node_recv[n]: n receives in layer l) LAYER needs no COMMIT: every node keeps
//   its embedding alternately in `h` and `tmp` - after v updates in buffer v & 1 -, a layer reads each child's
//   current buffer and writes the receiver's OTHER one, so nothing a layer reads is written by it. Layers run from
//   the highest index down (scheduler.py:209-211), so v = popcount of the bits above the layer's.
//   MERGE   h[n]      = tmp[n] where n was updated an odd number of times (once, after the last layer; DAGHID does
//                       the same on the fly when it is given node_recv, which saves the launch)
//   DAGHID  tmp[n]    = hidden part of dag([x[n], h[n]])
//   DAGSUM  h_dag[j]  = sum over the job's nodes n of dag([x[n], h[n]])        (scheduler.py:256-262)
//   GLOBHID tmp[j]    = hidden part of glob(h_dag[j])
//   GLOBSUM h_glob[o] = sum over the observation's jobs j of glob(h_dag[j])    (scheduler.py:271-283)
//   STAGE   score[obs(n), loc(n)] = stage([x, h, h_dag[job], h_glob[obs]] of schedulable node n)
//                                                                                (scheduler.py:296-318)
//   EXEC    score[b,c] = exec([x[first(j), :3], h_dag[j], h_glob[obs(j)], c/E]), j = job_sel[b];
//                        -inf where c >= cap[j]                                  (scheduler.py:337-385)
//
// No atomics: a node's out-edges, a job's nodes and an observation's jobs are contiguous ranges of
// the compact graph (sss_decima.h), so every sum is a short loop inside one thread, in a fixed order.
// The parameters of the launch's MLP(s) are staged in LDS once per workgroup and read from there
// as broadcasts (every thread of a wave multiplies its own row by the same weight; leaving them to
// scalar loads makes the compiler hoist thousands of loads and spill SGPRs through v_writelane). This is fp32 vector work: the GEMMs are [rows x <=53] x [<=53 x <=64] - far too
// thin for MFMA tiles to pay, and bf16/fp8 MFMA would not hold the 2e-5 agreement with the reference.
// Parameters of one MLP are packed [W1 (H1 x IN), b1, W2^T (H1 x H2), b2, W3 (OUT x H2), b3]: W1 and
// W3 row-major exactly as torch.nn.Linear.weight, the middle layer transposed.
//
// Included by sss_hip.hip (gfx950) and tests/emu/emu_backend.cpp (rows run in a plain loop there).
#pragma once

#if defined(__HIP_DEVICE_COMPILE__) || defined(__clang__)
#define GNN_UNROLL _Pragma("unroll")
#define GNN_NO_UNROLL _Pragma("clang loop unroll(disable)")
#define GNN_FP_CONTRACT _Pragma("clang fp contract(fast)")
// scheduling fence after every group of output neurons (~64 weights): without it the machine
// scheduler hoists the weight reads of ALL neurons to the top and spills hundreds of registers
#define GNN_FENCE(o, in) if ((((o) + 1) * (in)) % 64 < (in)) __builtin_amdgcn_sched_barrier(0)
#else
#define GNN_FENCE(o, in)
#define GNN_NO_UNROLL
#define GNN_UNROLL
#define GNN_FP_CONTRACT
#endif

enum { GNN_PREP = 0, GNN_SINK, GNN_LAYER, GNN_COMMIT, GNN_DAGSUM, GNN_GLOBSUM, GNN_STAGE, GNN_EXEC, GNN_DAGHID, GNN_GLOBHID, GNN_MERGE, GNN_KINDS };
enum { GNN_EMB = 16, GNN_NF = 5, GNN_DF = 3 };

struct SssGnnArgs {
  int64_t n_rows;
  const int64_t* n_rows_dev;  // nullable: the row count lives on the device (a Decima step without a host round trip); n_rows then
                              // only sizes the grid - the kernels stride over all rows - and may be any positive guess
  const float* w;       // packed parameters of the MLP this launch evaluates
  const float* w2;      // LAYER: the update MLP (w = the message MLP)
  const float *w16, *w2_16;  // LAYER, nullable: the two MLPs in the 16-lanes-per-row image (sss_gnn16.h)
  const int32_t* node_recv;  // LAYER (nullable: then COMMIT launches follow) / MERGE: per node, the layers it receives in
  const int64_t* layer_totals;  // LAYER, nullable: i64[32] on the device - the row count is layer_totals[layer] and idx0 starts
                                // after the lists of the layers below it (no host round trip for the list sizes); n_rows is then
                                // an upper bound that sizes the grid
  int64_t idx0_stride;          // ... with layer_totals: > 0 = layer l's list starts at idx0[l * idx0_stride] (the graph kernel's lists)
  // LAYER with the graph kernel's lists (sss_decima.h SssDecimaArgs::recv_lists): the layer's list is one dense piece per block of
  // `list_q` consecutive observations - block s's piece starts at idx0[l * idx0_stride + seg_off[s * list_q]] and has
  // layer_totals[l * SSS_LIST_SETS + s] rows (layer_totals is then i64[32][SSS_LIST_SETS]); list_q = 0: one dense list
  const int64_t* seg_off;       // i64[n_seg] first node of every observation
  int32_t n_seg, list_q;
  float slope;          // LeakyReLU negative slope (GNN MLPs)
  int E;                // EXEC: number of executors
  int layer;            // LAYER
  int64_t n_pad;        // STAGE: row stride of the padded score matrix
  const float* x;       // f32[M,5]
  const float* h_init;  // f32[M,16]
  float* h;             // f32[M,16]
  float* tmp;           // f32[M,16]
  float* h_dag;         // f32[J,16]
  float* h_glob;        // f32[n_obs,16]
  float* out;           // PREP: h_init; STAGE: f32[n_obs, n_pad]; EXEC: f32[B,E]
  const int32_t* out_deg;                            // SINK: a node with out-edges is a parent
  const int32_t* obs_depth;                          // SINK (nullable): observations with depth 0 keep h_init
  const int64_t* idx0;                               // LAYER/COMMIT: receiving nodes (-1 = padding); STAGE: nodes (-1 = padding); EXEC: job_sel
  const int64_t *dst, *out_start;                    // LAYER
  const uint32_t* edge_layers;                       // LAYER
  const int64_t *node_job, *node_obs, *node_loc, *job_obs, *job_first, *job_cap, *job_nodes, *obs_job_off, *obs_jobs;
};

template <int ACT>
SSS_DEV float gnn_act(float v, float slope) {
  if (ACT == 0) return v > 0.0f ? v : v * slope;
  return tanhf(v);
}

// bias + w . x with four independent accumulators (the serial FMA chain is what a lone wave waits on)
template <int N>
SSS_DEV float gnn_dot(const float* __restrict__ w, const float (&x)[N], float bias) {
  GNN_FP_CONTRACT
  float a0 = bias, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  GNN_UNROLL for (int i = 0; i + 3 < N; i += 4) {
    a0 += w[i] * x[i], a1 += w[i + 1] * x[i + 1], a2 += w[i + 2] * x[i + 2], a3 += w[i + 3] * x[i + 3];
  }
  GNN_UNROLL for (int i = N & ~3; i < N; i++) a0 += w[i] * x[i];
  return (a0 + a1) + (a2 + a3);
}

// One row through Linear-act-Linear-act (`gnn_hidden`) and the last Linear (`gnn_out`). Packed
// parameters: W1 (H1 x IN, row-major), b1, W2T (H1 x H2: the SECOND layer transposed), b2, W3
// (OUT x H2), b3. Layer 1 is streamed into layer 2: each hidden-1 neuron is computed and immediately
// scattered into the H2 accumulators, in a real (not unrolled) loop - both reads are contiguous, and
// the loop keeps the compiler from hoisting every weight read of the MLP to the top and spilling
// hundreds of registers. The last layer is linear, so sums of MLP outputs (messages over edges,
// node summaries over a job, job summaries over an observation) are taken over the hidden-2
// vectors and pushed through W3 once: sum_k (W3 h2_k + b3) = W3 (sum_k h2_k) + count * b3.
template <int IN, int H1, int H2, int ACT>
SSS_DEV void gnn_hidden(const float* __restrict__ w, const float (&x)[IN], float (&h2)[H2], float slope) {
  GNN_FP_CONTRACT
  const float* W1 = w;
  const float* b1 = W1 + H1 * IN;
  const float* W2T = b1 + H1;
  const float* b2 = W2T + H2 * H1;
  GNN_UNROLL for (int o = 0; o < H2; o++) h2[o] = b2[o];
  GNN_NO_UNROLL for (int j = 0; j < H1; j++) {
    const float* wr = W1 + j * IN;
    float t = gnn_act<ACT>(gnn_dot<IN>(wr, x, b1[j]), slope);
    const float* wc = W2T + j * H2;
    GNN_UNROLL for (int o = 0; o < H2; o++) h2[o] += wc[o] * t;
  }
  GNN_UNROLL for (int o = 0; o < H2; o++) h2[o] = gnn_act<ACT>(h2[o], slope);
}

// emit(o, b3[o] * bias_count + W3[o,:] . h2) for every output o
template <int IN, int H1, int H2, int OUT, typename Emit>
SSS_DEV void gnn_out(const float* __restrict__ w, const float (&h2)[H2], float bias_count, Emit emit) {
  GNN_FP_CONTRACT
  const float* W3 = w + H1 * IN + H1 + H2 * H1 + H2;
  const float* b3 = W3 + OUT * H2;
  GNN_NO_UNROLL for (int o = 0; o < OUT; o++) {
    emit(o, gnn_dot<H2>(W3 + o * H2, h2, b3[o] * bias_count));
  }
}

template <int N>
SSS_DEV void gnn_load(const float* p, float* dst) {
  GNN_UNROLL for (int i = 0; i < N; i++) dst[i] = p[i];
}

// number of packed parameters of the MLP(s) a stage evaluates (LAYER: message MLP then update MLP)
constexpr int gnn_mlp_params(int in, int h1, int h2, int out) { return h1 * in + h1 + h2 * h1 + h2 + out * h2 + out; }
constexpr int GNN_W_GNN16 = gnn_mlp_params(16, 32, 16, 16);
template <int KIND>
constexpr int gnn_weight_count() {
  return KIND == GNN_PREP ? gnn_mlp_params(GNN_NF, 32, 16, 16)
       : (KIND == GNN_DAGSUM || KIND == GNN_DAGHID) ? gnn_mlp_params(GNN_NF + 16, 32, 16, 16)
       : KIND == GNN_STAGE ? gnn_mlp_params(GNN_NF + 48, 64, 64, 1)
       : KIND == GNN_EXEC ? gnn_mlp_params(GNN_DF + 33, 64, 64, 1)
       : (KIND == GNN_COMMIT || KIND == GNN_MERGE) ? 0 : GNN_W_GNN16;
}

// `w` / `w2`: where this thread reads the parameters from (the gfx950 build stages them in LDS and
// every lane reads the same address - a broadcast; the emulator reads the argument buffers)
template <int KIND>
SSS_DEV void gnn_row(const SssGnnArgs& a, int64_t r, const float* w, const float* w2) {
  constexpr int F = GNN_EMB;
  if (KIND == GNN_PREP) {
    float x[GNN_NF], h2[16], hi[F];
    gnn_load<GNN_NF>(a.x + r * GNN_NF, x);
    gnn_hidden<GNN_NF, 32, 16, 0>(w, x, h2, a.slope);
    gnn_out<GNN_NF, 32, 16, F>(w, h2, 1.0f, [&](int o, float v) { hi[o] = v, a.out[r * F + o] = v; });
    if (a.w2 && a.h) {  // SINK in the same pass (w2 = the update MLP): the row's h_init is still in registers
      bool par = a.out_deg[r] != 0;
      bool skip = a.obs_depth != nullptr && a.obs_depth[a.node_obs[r]] == 0;
      if (skip || par) {
        GNN_UNROLL for (int i = 0; i < F; i++) a.h[r * F + i] = skip ? hi[i] : 0.0f;
        return;
      }
      gnn_hidden<F, 32, 16, 0>(w2, hi, h2, a.slope);
      gnn_out<F, 32, 16, F>(w2, h2, 1.0f, [&](int o, float v) { a.h[r * F + o] = v; });
    }
  } else if (KIND == GNN_SINK) {
    float x[F], h2[16];
    gnn_load<F>(a.h_init + r * F, x);
    bool par = a.out_deg[r] != 0;
    bool skip = a.obs_depth != nullptr && a.obs_depth[a.node_obs[r]] == 0;  // single-layer observation: mlp_prep only
    if (skip || par) {
      GNN_UNROLL for (int i = 0; i < F; i++) a.h[r * F + i] = skip ? x[i] : 0.0f;
      return;
    }
    gnn_hidden<F, 32, 16, 0>(w, x, h2, a.slope);
    gnn_out<F, 32, 16, F>(w, h2, 1.0f, [&](int o, float v) { a.h[r * F + o] = v; });
  } else if (KIND == GNN_LAYER) {
    int64_t n = a.idx0[r];
    if (n < 0) return;
    float acc[16], x[F], h2[16];
    GNN_UNROLL for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    int64_t e0 = a.out_start[n];
    int deg = a.out_deg[n], used = 0;
    const uint32_t above = a.layer >= 31 ? 0u : ~((2u << a.layer) - 1u);  // the layers that ran before this one
    for (int k = 0; k < deg; k++) {
      if (!((a.edge_layers[e0 + k] >> a.layer) & 1u)) continue;
      const int64_t c = a.dst[e0 + k];
      const float* cur = (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[c] & above) & 1)) ? a.tmp : a.h;
      gnn_load<F>(cur + c * F, x);
      gnn_hidden<F, 32, 16, 0>(w, x, h2, a.slope);
      GNN_UNROLL for (int i = 0; i < 16; i++) acc[i] += h2[i];
      used++;
    }
    float agg[F];
    GNN_UNROLL for (int i = 0; i < F; i++) agg[i] = 0.0f;
    // the aggregated message = W3 . (sum of hidden vectors) + used * b3; it feeds the update MLP,
    // whose input must sit in registers: F is small, so this last layer is unrolled
    {
      GNN_FP_CONTRACT
      const float* W3 = w + 32 * F + 32 + 16 * 32 + 16;
      const float* b3 = W3 + F * 16;
      GNN_UNROLL for (int o = 0; o < F; o++) {
        float v = b3[o] * (float)used;
        GNN_UNROLL for (int i = 0; i < 16; i++) v += W3[o * 16 + i] * acc[i];
        agg[o] = v;
        GNN_FENCE(o, 16);
      }
    }
    gnn_hidden<F, 32, 16, 0>(w2, agg, h2, a.slope);
    float* nxt = (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[n] & above) & 1)) ? a.h : a.tmp;  // the buffer its current value is NOT in
    gnn_out<F, 32, 16, F>(w2, h2, 1.0f, [&](int o, float v) { nxt[n * F + o] = a.h_init[n * F + o] + v; });
  } else if (KIND == GNN_COMMIT) {
    int64_t n = a.idx0[r];
    if (n < 0) return;
    GNN_UNROLL for (int i = 0; i < F; i++) a.h[n * F + i] = a.tmp[n * F + i];
  } else if (KIND == GNN_MERGE) {
    if (__builtin_popcount((uint32_t)a.node_recv[r]) & 1) {
      GNN_UNROLL for (int i = 0; i < F; i++) a.h[r * F + i] = a.tmp[r * F + i];
    }
  } else if (KIND == GNN_DAGHID) {
    float x[GNN_NF + F], h2[16];
    gnn_load<GNN_NF>(a.x + r * GNN_NF, x);
    if (a.node_recv && (__builtin_popcount((uint32_t)a.node_recv[r]) & 1)) {
      // the MERGE of a node whose embedding ended up in `tmp` (odd number of updates), done here where the row
      // is read anyway: this thread owns row r of both buffers, and tmp[r] is overwritten only further down
      gnn_load<F>(a.tmp + r * F, x + GNN_NF);
      GNN_UNROLL for (int i = 0; i < F; i++) a.h[r * F + i] = x[GNN_NF + i];
    } else
      gnn_load<F>(a.h + r * F, x + GNN_NF);
    gnn_hidden<GNN_NF + F, 32, 16, 0>(w, x, h2, a.slope);
    GNN_UNROLL for (int i = 0; i < 16; i++) a.tmp[r * 16 + i] = h2[i];
  } else if (KIND == GNN_DAGSUM) {
    float acc[16];
    GNN_UNROLL for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    int64_t n0 = a.job_first[r], cnt = a.job_nodes[r];
    for (int64_t n = n0; n < n0 + cnt; n++) {
      GNN_UNROLL for (int i = 0; i < 16; i++) acc[i] += a.tmp[n * 16 + i];
    }
    gnn_out<GNN_NF + F, 32, 16, F>(w, acc, (float)cnt, [&](int o, float v) { a.h_dag[r * F + o] = v; });
  } else if (KIND == GNN_GLOBHID) {
    float x[F], h2[16];
    gnn_load<F>(a.h_dag + r * F, x);
    gnn_hidden<F, 32, 16, 0>(w, x, h2, a.slope);
    GNN_UNROLL for (int i = 0; i < 16; i++) a.tmp[r * 16 + i] = h2[i];
  } else if (KIND == GNN_GLOBSUM) {
    float acc[16];
    GNN_UNROLL for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    int64_t j0 = a.obs_job_off[r], cnt = a.obs_jobs[r];
    for (int64_t j = j0; j < j0 + cnt; j++) {
      GNN_UNROLL for (int i = 0; i < 16; i++) acc[i] += a.tmp[j * 16 + i];
    }
    gnn_out<F, 32, 16, F>(w, acc, (float)cnt, [&](int o, float v) { a.h_glob[r * F + o] = v; });
  } else if (KIND == GNN_STAGE) {
    int64_t n = a.idx0[r];
    if (n < 0) return;
    float x[GNN_NF + 3 * F], h2[64];
    gnn_load<GNN_NF>(a.x + n * GNN_NF, x);
    gnn_load<F>(a.h + n * F, x + GNN_NF);
    gnn_load<F>(a.h_dag + a.node_job[n] * F, x + GNN_NF + F);
    gnn_load<F>(a.h_glob + a.node_obs[n] * F, x + GNN_NF + 2 * F);
    gnn_hidden<GNN_NF + 3 * F, 64, 64, 1>(w, x, h2, 0.0f);
    gnn_out<GNN_NF + 3 * F, 64, 64, 1>(w, h2, 1.0f, [&](int, float v) { a.out[a.node_obs[n] * a.n_pad + a.node_loc[n]] = v; });
  } else if (KIND == GNN_EXEC) {
    int64_t b = r / a.E;
    int c = (int)(r - b * a.E);
    int64_t j = a.idx0[b];
    float x[GNN_DF + 2 * F + 1], h2[64];
    gnn_load<GNN_DF>(a.x + a.job_first[j] * GNN_NF, x);
    gnn_load<F>(a.h_dag + j * F, x + GNN_DF);
    gnn_load<F>(a.h_glob + a.job_obs[j] * F, x + GNN_DF + F);
    x[GNN_DF + 2 * F] = (float)c / (float)a.E;
    gnn_hidden<GNN_DF + 2 * F + 1, 64, 64, 1>(w, x, h2, 0.0f);
    bool ok = c < a.job_cap[j];
    gnn_out<GNN_DF + 2 * F + 1, 64, 64, 1>(w, h2, 1.0f, [&](int, float v) { a.out[r] = ok ? v : -__builtin_inff(); });
  }
}
