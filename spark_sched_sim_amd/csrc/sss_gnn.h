// sss_gnn.h - Decima's GNN forward pass (inference) as a handful of row-parallel kernels
// (reference schedulers/decima/scheduler.py:142-385). Every MLP of the
// published architecture (config/decima_tpch.yaml:66-78: embed 16, GNN MLPs [32,16] LeakyReLU,
// policy MLPs [64,64] Tanh) is evaluated whole - Linear/act/Linear/act/Linear - by ONE thread per
// row, with the gather that builds the row's input and the scatter / segment-sum that consumes its
// output fused in:
//
//   PREP    h_init[n] = prep(x[n])                                            (scheduler.py:200)
//   SINK    h[n]      = is_parent[n] ? 0 : update(h_init[n])                  (scheduler.py:206-209)
//                       (h_init[n] instead where the node's observation has a single DAG layer, :196-198)
//   LAYER   tmp[r]    = h_init[r] + update( sum over r's out-edges e in DAG layer l of msg(h[dst_e]) )
//                       for the layer's receiving nodes r                       (scheduler.py:214-232)
//   COMMIT  h[r]      = tmp[r]   (a layer reads the previous h everywhere before any node moves)
//   DAGSUM  h_dag[j]  = sum over the job's nodes n of dag([x[n], h[n]])        (scheduler.py:256-262)
//   GLOBSUM h_glob[o] = sum over the observation's jobs j of glob(h_dag[j])    (scheduler.py:271-283)
//   STAGE   score[obs(n), loc(n)] = stage([x, h, h_dag[job], h_glob[obs]] of schedulable node n)
//                                                                                (scheduler.py:296-318)
//   EXEC    score[b,c] = exec([x[first(j), :3], h_dag[j], h_glob[obs(j)], c/E]), j = job_sel[b];
//                        -inf where c >= cap[j]                                  (scheduler.py:337-385)
//
// No atomics: a node's out-edges, a job's nodes and an observation's jobs are contiguous ranges of
// the compact graph (sss_decima.h), so every sum is a short loop inside one thread, in a fixed order.
// Weights are read through uniform (scalar) loads: every thread of a wave multiplies its own row by
// the same weight. This is fp32 vector work: the GEMMs are [rows x <=53] x [<=53 x <=64] - far too
// thin for MFMA tiles to pay, and bf16/fp8 MFMA would not hold the 2e-5 agreement with the reference.
// Parameters of one MLP are packed [W1 (H1 x IN), b1, W2 (H2 x H1), b2, W3 (OUT x H2), b3], each W
// row-major exactly as torch.nn.Linear.weight.
//
// Included by sss_hip.hip (gfx950) and tests/emu/emu_backend.cpp (rows run in a plain loop there).
#pragma once

#if defined(__HIP_DEVICE_COMPILE__) || defined(__clang__)
#define GNN_UNROLL _Pragma("unroll")
#define GNN_FP_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define GNN_UNROLL
#define GNN_FP_CONTRACT
#endif

enum { GNN_PREP = 0, GNN_SINK, GNN_LAYER, GNN_COMMIT, GNN_DAGSUM, GNN_GLOBSUM, GNN_STAGE, GNN_EXEC, GNN_KINDS };
enum { GNN_EMB = 16, GNN_NF = 5, GNN_DF = 3 };

struct SssGnnArgs {
  int64_t n_rows;
  const float* w;       // packed parameters of the MLP this launch evaluates
  const float* w2;      // LAYER: the update MLP (w = the message MLP)
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

template <int IN, int H1, int H2, int OUT, int ACT>
SSS_DEV void gnn_mlp(const float* __restrict__ w, const float (&x)[IN], float (&y)[OUT], float slope) {
GNN_FP_CONTRACT
  const float* W1 = w;
  const float* b1 = W1 + H1 * IN;
  const float* W2 = b1 + H1;
  const float* b2 = W2 + H2 * H1;
  const float* W3 = b2 + H2;
  const float* b3 = W3 + OUT * H2;
  float h1[H1], h2[H2];
  GNN_UNROLL for (int o = 0; o < H1; o++) {
    float acc = b1[o];
    GNN_UNROLL for (int i = 0; i < IN; i++) acc += W1[o * IN + i] * x[i];
    h1[o] = gnn_act<ACT>(acc, slope);
  }
  GNN_UNROLL for (int o = 0; o < H2; o++) {
    float acc = b2[o];
    GNN_UNROLL for (int i = 0; i < H1; i++) acc += W2[o * H1 + i] * h1[i];
    h2[o] = gnn_act<ACT>(acc, slope);
  }
  GNN_UNROLL for (int o = 0; o < OUT; o++) {
    float acc = b3[o];
    GNN_UNROLL for (int i = 0; i < H2; i++) acc += W3[o * H2 + i] * h2[i];
    y[o] = acc;
  }
}

template <int N>
SSS_DEV void gnn_load(const float* p, float* dst) {
  GNN_UNROLL for (int i = 0; i < N; i++) dst[i] = p[i];
}

template <int KIND>
SSS_DEV void gnn_row(const SssGnnArgs& a, int64_t r) {
  constexpr int F = GNN_EMB;
  if (KIND == GNN_PREP) {
    float x[GNN_NF], y[F];
    gnn_load<GNN_NF>(a.x + r * GNN_NF, x);
    gnn_mlp<GNN_NF, 32, 16, F, 0>(a.w, x, y, a.slope);
    GNN_UNROLL for (int i = 0; i < F; i++) a.out[r * F + i] = y[i];
  } else if (KIND == GNN_SINK) {
    float x[F], y[F];
    gnn_load<F>(a.h_init + r * F, x);
    gnn_mlp<F, 32, 16, F, 0>(a.w, x, y, a.slope);
    bool par = a.out_deg[r] != 0;
    bool skip = a.obs_depth != nullptr && a.obs_depth[a.node_obs[r]] == 0;  // single-layer observation: mlp_prep only
    GNN_UNROLL for (int i = 0; i < F; i++) a.h[r * F + i] = skip ? x[i] : (par ? 0.0f : y[i]);
  } else if (KIND == GNN_LAYER) {
    int64_t n = a.idx0[r];
    if (n < 0) return;
    float acc[F], x[F], y[F];
    GNN_UNROLL for (int i = 0; i < F; i++) acc[i] = 0.0f;
    int64_t e0 = a.out_start[n];
    int deg = a.out_deg[n];
    for (int k = 0; k < deg; k++) {
      if (!((a.edge_layers[e0 + k] >> a.layer) & 1u)) continue;
      gnn_load<F>(a.h + a.dst[e0 + k] * F, x);
      gnn_mlp<F, 32, 16, F, 0>(a.w, x, y, a.slope);
      GNN_UNROLL for (int i = 0; i < F; i++) acc[i] += y[i];
    }
    gnn_mlp<F, 32, 16, F, 0>(a.w2, acc, y, a.slope);
    GNN_UNROLL for (int i = 0; i < F; i++) a.tmp[n * F + i] = a.h_init[n * F + i] + y[i];
  } else if (KIND == GNN_COMMIT) {
    int64_t n = a.idx0[r];
    if (n < 0) return;
    GNN_UNROLL for (int i = 0; i < F; i++) a.h[n * F + i] = a.tmp[n * F + i];
  } else if (KIND == GNN_DAGSUM) {
    float acc[F], x[GNN_NF + F], y[F];
    GNN_UNROLL for (int i = 0; i < F; i++) acc[i] = 0.0f;
    int64_t n0 = a.job_first[r], cnt = a.job_nodes[r];
    for (int64_t n = n0; n < n0 + cnt; n++) {
      gnn_load<GNN_NF>(a.x + n * GNN_NF, x);
      gnn_load<F>(a.h + n * F, x + GNN_NF);
      gnn_mlp<GNN_NF + F, 32, 16, F, 0>(a.w, x, y, a.slope);
      GNN_UNROLL for (int i = 0; i < F; i++) acc[i] += y[i];
    }
    GNN_UNROLL for (int i = 0; i < F; i++) a.h_dag[r * F + i] = acc[i];
  } else if (KIND == GNN_GLOBSUM) {
    float acc[F], x[F], y[F];
    GNN_UNROLL for (int i = 0; i < F; i++) acc[i] = 0.0f;
    int64_t j0 = a.obs_job_off[r], cnt = a.obs_jobs[r];
    for (int64_t j = j0; j < j0 + cnt; j++) {
      gnn_load<F>(a.h_dag + j * F, x);
      gnn_mlp<F, 32, 16, F, 0>(a.w, x, y, a.slope);
      GNN_UNROLL for (int i = 0; i < F; i++) acc[i] += y[i];
    }
    GNN_UNROLL for (int i = 0; i < F; i++) a.h_glob[r * F + i] = acc[i];
  } else if (KIND == GNN_STAGE) {
    int64_t n = a.idx0[r];
    if (n < 0) return;
    float x[GNN_NF + 3 * F], y[1];
    gnn_load<GNN_NF>(a.x + n * GNN_NF, x);
    gnn_load<F>(a.h + n * F, x + GNN_NF);
    gnn_load<F>(a.h_dag + a.node_job[n] * F, x + GNN_NF + F);
    gnn_load<F>(a.h_glob + a.node_obs[n] * F, x + GNN_NF + 2 * F);
    gnn_mlp<GNN_NF + 3 * F, 64, 64, 1, 1>(a.w, x, y, 0.0f);
    a.out[a.node_obs[n] * a.n_pad + a.node_loc[n]] = y[0];
  } else if (KIND == GNN_EXEC) {
    int64_t b = r / a.E;
    int c = (int)(r - b * a.E);
    int64_t j = a.idx0[b];
    float x[GNN_DF + 2 * F + 1], y[1];
    gnn_load<GNN_DF>(a.x + a.job_first[j] * GNN_NF, x);
    gnn_load<F>(a.h_dag + j * F, x + GNN_DF);
    gnn_load<F>(a.h_glob + a.job_obs[j] * F, x + GNN_DF + F);
    x[GNN_DF + 2 * F] = (float)c / (float)a.E;
    gnn_mlp<GNN_DF + 2 * F + 1, 64, 64, 1, 1>(a.w, x, y, 0.0f);
    a.out[r] = c < a.job_cap[j] ? y[0] : -__builtin_inff();
  }
}
