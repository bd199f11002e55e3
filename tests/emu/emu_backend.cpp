// tests/emu/emu_backend.cpp - the C ABI of include/sss.h on the CPU wave emulator
// (TEST INFRASTRUCTURE: lets tests/ run the kernel source of spark_sched_sim_amd/csrc/sss_sim.h
// without a GPU, also under ASan/UBSan). "Device" memory is host memory; launches are synchronous.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#include "wave_rt.h"
namespace emu {
void launch(int grid, const std::function<void()>& body);
}

#ifdef SSS_BATCH_STATS
extern "C" { long long sss_batch_stats[128]; }
#endif
#include "sss_sim.h"
#include "sss_decima.h"
#include <math.h>
#include "sss_gnn.h"
#include "sss_decima_policy.h"
#include "zig_tables.inc"

#include "sss_narrow.h"
int sss_narrow_hot_bytes() { return (int)sizeof(SssHot); }
int sss_narrow_static_lds_bytes() { return SSS_STATIC_LDS_BYTES; }

static int be_set_device(int) { return 0; }
struct BeDeviceGuard {
  explicit BeDeviceGuard(int) {}
};
static const char* be_error(int) { return "emulator"; }
static void* be_alloc(size_t n) { return calloc(1, n ? n : 1); }
static void be_free(void* p) { free(p); }
static int be_h2d(void* dst, const void* src, size_t n) {
  memcpy(dst, src, n);
  return 0;
}
static int be_launch_reset(const SssKernelArgs& a, int num_envs, const uint64_t* seeds, const double* tl, const uint8_t* mask, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_reset_kernel(a, seeds, tl, mask); });
  return 0;
}
static int be_launch_step_bounded(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                  int budget, uint8_t* ready, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_step_bounded_kernel(a, stage_idx, num_exec, auto_reset, seed_stride, budget, ready); });
  return 0;
}
static int be_launch_step(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_step_kernel(a, stage_idx, num_exec, auto_reset, seed_stride); });
  return 0;
}

static int be_launch_policy(const SssKernelArgs& a, int num_envs, int policy, int param, int32_t* stage_idx, int32_t* num_exec, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_policy_kernel(a, policy, param, stage_idx, num_exec); });
  return 0;
}
static int be_launch_rollout(const SssKernelArgs& a, int num_envs, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void*) {
  emu::g_kernargs = &a;
  emu::launch(num_envs, [&]() { sss_rollout_kernel(a, policy, param, n_steps, auto_reset, seed_stride); });
  return 0;
}

static int be_launch_decima(const SssLayout& L, const SssBuffers& B, int E, const SssDecimaArgs& d, void*) {
  emu::launch(L.num_envs, [&]() { sss_decima_graph_kernel(L, B, E, d); });
  return 0;
}

#include "sss_train.h"
#include "sss_collect.h"
static int be_launch_collect(const SssCollectArgs& a, int phase, void*) {
  for (int b = 0; b < a.num_envs; b++) collect_env(a, phase, b, [&](int i, int v) { a.flags[i] |= v; }, [&](int i, int v) { a.flags[i] = v > a.flags[i] ? v : a.flags[i]; });
  return 0;
}
// sss_rows_op on the host: the element statement of sss_rows.h in plain loops
#include "sss_rows.h"
static int be_launch_rows(const SssRowsArgs& r, void*) {
  for (int64_t i = 0; i < r.n; i++)
    for (int j = 0; j < r.width; j++) sss_rows_element(r, i, j, [](float* p, float v) { *p += v; });
  return 0;
}
static int be_launch_concat(const SssConcatArgs& r, void*) {
  for (int64_t i = 0; i < r.n; i++)
    for (int c = 0; c < r.width; c++) sss_concat_element(r, i, c, [](float* p, float v) { *p += v; });
  return 0;
}
// sss_segment_categorical on the host: the per-segment function of sss_segcat.h in a plain loop
#include "sss_segcat.h"
static int be_launch_segcat(const SssSegcatArgs& a, int backward, void*) {
  for (int64_t s = 0; s < a.n_seg; s++) segcat_segment(a, s, backward != 0);
  return 0;
}
// sss_discounted_returns / sss_sequence_baselines on the host: the per-env / per-query functions of sss_returns.h in plain loops
#include "sss_returns.h"
static int be_launch_returns(const SssReturnsArgs& a, void*) {
  for (int64_t b = 0; b < a.B; b++) returns_env(a, b);
  return 0;
}
static int be_launch_baselines(const SssBaselineArgs& a, void*) {
  for (int64_t t = 0; t < a.T; t++)
    for (int64_t b = 0; b < a.B; b++) baseline_query(a, t, b);
  return 0;
}
// sss_arena_append on the host
#include "sss_arena.h"
static int be_launch_arena(const SssArenaArgs& a, int64_t, void*) {
  if (arena_fits(a))
    for (int k = 0; k < a.n_arrays; k++) {
      const int64_t n = arena_rows(a, a.arrays[k].kind) * a.arrays[k].per_row;
      for (int64_t e = 0; e < n; e++) arena_copy_element(a, a.arrays[k], e);
    }
  arena_advance(a);
  return 0;
}
// sss_linear_wgrad on the host (the MFMA kernel is gfx950-only): plain loops, same results up to summation order
static int be_launch_wgrad(const SssWgradArgs& a, void*) {
  for (int n = 0; n < a.N; n++) {
    for (int m = 0; m < a.M; m++) {
      double s = 0;
      for (int64_t k = 0; k < a.K; k++) s += (double)a.dy[k * a.ldy + n] * (double)a.x[k * a.ldx + m];
      a.gw[n * a.M + m] = (float)s;
    }
    if (a.gb) {
      double s = 0;
      for (int64_t k = 0; k < a.K; k++) s += (double)a.dy[k * a.ldy + n];
      a.gb[n] = (float)s;
    }
  }
  return 0;
}

// sss_mlp_forward / sss_mlp_backward on the host (the 16-lanes-per-row kernels are gfx950-only): the arithmetic of
// sss_train16.h with plain loops
static float emu_act(int act, float v, float slope) { return act == 0 ? (v > 0.0f ? v : v * slope) : tanhf(v); }
static float emu_act_grad(int act, float a, float slope) { return act == 0 ? (a > 0.0f ? 1.0f : slope) : 1.0f - a * a; }
static int be_mlp_recompute_supported(int in_dim) { return in_dim == GNN_NF || in_dim == 16 || in_dim == GNN_NF + 16; }
static int be_mlp_split_supported(int in_dim) { return in_dim == GNN_NF + 16; }
// (an input in two pieces, SssMlpArgs.x2: the rows put together on the host, then the loops below)
static std::vector<float> emu_mlp_joined_x(const SssMlpArgs& a) {
  const int IN = a.in_dim, P = IN - 16;
  std::vector<float> x((size_t)std::max<int64_t>(a.rows, 1) * IN);
  for (int64_t r = 0; r < a.rows; r++)
    for (int c = 0; c < IN; c++) x[r * IN + c] = c < P ? a.x[r * P + c] : a.x2[r * 16 + c - P];
  return x;
}
static int be_launch_mlp(const SssMlpArgs& a0, int backward, void*) {
  // (forward without a1 / a2: the hidden activations are not kept - the backward pass recomputes them, sss_mlp_recompute_supported)
  SssMlpArgs a = a0;
  std::vector<float> joined;
  if (a.x2) joined = emu_mlp_joined_x(a0), a.x = joined.data(), a.x2 = nullptr;
  std::vector<float> tmp1, tmp2;
  if (!backward && !a.a1) tmp1.resize((size_t)std::max<int64_t>(a.rows, 1) * a.h1), tmp2.resize((size_t)std::max<int64_t>(a.rows, 1) * a.h2), a.a1 = tmp1.data(), a.a2 = tmp2.data();
  const int IN = a.in_dim, H1 = a.h1, H2 = a.h2, OUT = a.out_dim;
  const float* W1 = a.w;
  const float* b1 = W1 + H1 * IN;
  const float* W2T = b1 + H1;
  const float* b2 = W2T + H1 * H2;
  const float* W3 = b2 + H2;
  const float* b3 = W3 + OUT * H2;
  for (int64_t r = 0; r < a.rows; r++) {
    if (!backward) {
      for (int j = 0; j < H1; j++) {
        float v = b1[j];
        for (int i = 0; i < IN; i++) v += W1[j * IN + i] * a.x[r * IN + i];
        a.a1[r * H1 + j] = emu_act(a.act, v, a.slope);
      }
      for (int m = 0; m < H2; m++) {
        float v = b2[m];
        for (int j = 0; j < H1; j++) v += W2T[j * H2 + m] * a.a1[r * H1 + j];
        a.a2[r * H2 + m] = emu_act(a.act, v, a.slope);
      }
      for (int o = 0; o < OUT; o++) {
        float v = b3[o];
        for (int m = 0; m < H2; m++) v += W3[o * H2 + m] * a.a2[r * H2 + m];
        a.y[r * OUT + o] = v;
      }
    } else {
      for (int m = 0; m < H2; m++) {
        float v = 0.0f;
        for (int o = 0; o < OUT; o++) v += W3[o * H2 + m] * a.dy[r * OUT + o];
        a.g2[r * H2 + m] = v * emu_act_grad(a.act, a.a2[r * H2 + m], a.slope);
      }
      for (int j = 0; j < H1; j++) {
        float v = 0.0f;
        for (int m = 0; m < H2; m++) v += W2T[j * H2 + m] * a.g2[r * H2 + m];
        a.g1[r * H1 + j] = v * emu_act_grad(a.act, a.a1[r * H1 + j], a.slope);
      }
      if (a.dx)
        for (int i = 0; i < IN; i++) {
          float v = 0.0f;
          for (int j = 0; j < H1; j++) v += W1[j * IN + i] * a.g1[r * H1 + j];
          a.dx[r * IN + i] = v;
        }
    }
  }
  return 0;
}

// sss_mlp_backward_wgrad / sss_mlp_wgrad_finish on the host: be_launch_mlp's backward arithmetic, the parameter gradients added to slot 0
static int be_launch_mlp_bwdw(const SssMlpArgs& a0, float* acc, void*) {
  const int IN = a0.in_dim, H1 = a0.h1, H2 = a0.h2, OUT = a0.out_dim;
  std::vector<float> g1((size_t)std::max<int64_t>(a0.rows, 1) * H1), g2((size_t)std::max<int64_t>(a0.rows, 1) * H2);
  SssMlpArgs a = a0;
  a.g1 = g1.data(), a.g2 = g2.data();
  std::vector<float> joined, dx_full;
  if (a.x2) {  // an input in two pieces: joined rows in, the second piece's columns of dx out (below)
    joined = emu_mlp_joined_x(a0), a.x = joined.data(), a.x2 = nullptr;
    if (a.dx2) dx_full.resize(joined.size()), a.dx = dx_full.data(), a.dx2 = nullptr;
  }
  std::vector<float> r1, r2, ry;
  if (!a.a1) {  // the hidden activations were not stored: the forward pass again (same loops, same order)
    r1.resize(g1.size()), r2.resize(g2.size()), ry.resize((size_t)std::max<int64_t>(a0.rows, 1) * OUT);
    SssMlpArgs f = a;
    f.a1 = r1.data(), f.a2 = r2.data(), f.y = ry.data();
    if (int rc = be_launch_mlp(f, 0, nullptr)) return rc;
    a.a1 = r1.data(), a.a2 = r2.data();
  }
  if (int rc = be_launch_mlp(a, 1, nullptr)) return rc;
  if (a0.dx2)
    for (int64_t r = 0; r < a.rows; r++)
      for (int c = 0; c < 16; c++) a0.dx2[r * 16 + c] = dx_full[r * IN + IN - 16 + c];
  const size_t slots = H1 == 64 ? 512 : 2048;  // (csrc/sss_host.h SSS_MLPW_HEAD_SLOTS / SSS_MLPW_SLOTS)
  float* l3 = acc;
  float* l2 = l3 + slots * (OUT * H2 + OUT);
  float* l1 = l2 + slots * (H2 * H1 + H2);
  for (int64_t r = 0; r < a.rows; r++) {
    for (int o = 0; o < OUT; o++) {
      for (int m = 0; m < H2; m++) l3[o * H2 + m] += a.dy[r * OUT + o] * a.a2[r * H2 + m];
      l3[OUT * H2 + o] += a.dy[r * OUT + o];
    }
    for (int m = 0; m < H2; m++) {
      for (int j = 0; j < H1; j++) l2[m * H1 + j] += g2[r * H2 + m] * a.a1[r * H1 + j];
      l2[H2 * H1 + m] += g2[r * H2 + m];
    }
    for (int j = 0; j < H1; j++) {
      for (int c = 0; c < IN; c++) l1[j * IN + c] += g1[r * H1 + j] * a.x[r * IN + c];
      l1[H1 * IN + j] += g1[r * H1 + j];
    }
  }
  return 0;
}
static int be_mlp_head_bwdw_supported() { return 1; }
static int be_launch_wgrad_reduce(const SssWgradArgs& a, void*) {
  const int n_out = a.N * a.M + a.N;
  for (int i = 0; i < n_out; i++) {
    float s = 0.0f;
    for (int p = 0; p < a.n_partials; p++) s += a.partial[(size_t)p * n_out + i];
    if (i < a.N * a.M) a.gw[i] = s;
    else if (a.gb) a.gb[i - a.N * a.M] = s;
  }
  return 0;
}

static int be_launch_prefix_rows(const SssPrefixArgs& a, void*) {
  int64_t part[1];
  for (int r = 0; r < a.n_rows; r++) prefix_row(a, r, 0, 1, part, [] {});
  return 0;
}
static int be_launch_bit_lists(const SssBitListArgs& a, void*) {
  emu::launch(a.n_chunks, [&]() { sss_bit_lists_kernel(a); });
  return 0;
}
static int be_launch_decima_lists(int num_envs, const SssDecimaListArgs& d, void*) {
  emu::launch(num_envs, [&]() { sss_decima_lists_kernel(num_envs, d); });
  return 0;
}

static int be_launch_decima_policy(const SssLayout& L, const SssBuffers& B, int E, const SssDecimaPolicyArgs& d, void*) {
  emu::launch(L.num_envs, [&]() { sss_decima_policy_kernel(L, B, E, d); });
  return 0;
}

static int be_launch_decima_sample(int n_obs, int which, const SssDecimaSampleArgs& d, void*) {
  if (which == 0) emu::launch(n_obs, [&]() { sss_decima_sample_stage_kernel(d); });
  else emu::launch(n_obs, [&]() { sss_decima_sample_exec_kernel(d); });
  return 0;
}

template <int KIND>
static int gnn_run_kind(const SssGnnArgs& a) {
  const int64_t rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  for (int64_t r = 0; r < rows; r++) gnn_row<KIND>(a, r, a.w, a.w2);
  return 0;
}
#define BE_UNAVAILABLE (-1000)
static int be_launch_gnn_layers_obs(const SssGnnArgs&, const int64_t*, const int64_t*, const int32_t*, int, int, void*) { return BE_UNAVAILABLE; }  // (matrix-core kernel: gfx950 only)
static int be_launch_gnn(int kind, const SssGnnArgs& a, void*) {
  switch (kind) {
    case GNN_PREP: return gnn_run_kind<GNN_PREP>(a);
    case GNN_SINK: return gnn_run_kind<GNN_SINK>(a);
    case GNN_LAYER: {
      SssGnnArgs b = a;
      if (b.list_q) {  // the graph kernel's lists (sss_gnn.h list_q): a dense piece per block of observations, piece after piece
        const int n_sets = (b.n_seg + b.list_q - 1) / b.list_q;
        for (int s = 0; s < n_sets && s < SSS_LIST_SETS; s++) {
          SssGnnArgs c = b;
          c.list_q = 0, c.layer_totals = nullptr, c.n_rows_dev = nullptr;
          c.n_rows = b.layer_totals[(int64_t)b.layer * SSS_LIST_SETS + s], c.idx0 = b.idx0 + (int64_t)b.layer * b.idx0_stride + b.seg_off[(int64_t)s * b.list_q];
          if (int rc = gnn_run_kind<GNN_LAYER>(c)) return rc;
        }
        return 0;
      }
      if (b.layer_totals) {  // (the gfx950 kernel reads the list's length and position from the device, sss_gnn16.h)
        int64_t off = (int64_t)b.layer * b.idx0_stride;
        if (b.idx0_stride == 0)
          for (int l = 0; l < b.layer; l++) off += b.layer_totals[l];
        b.n_rows = b.layer_totals[b.layer], b.idx0 += off;
      }
      return gnn_run_kind<GNN_LAYER>(b);
    }
    case GNN_COMMIT: return gnn_run_kind<GNN_COMMIT>(a);
    case GNN_MERGE: return gnn_run_kind<GNN_MERGE>(a);
    case GNN_DAGSUM: return gnn_run_kind<GNN_DAGSUM>(a);
    case GNN_GLOBSUM: return gnn_run_kind<GNN_GLOBSUM>(a);
    case GNN_STAGE: return gnn_run_kind<GNN_STAGE>(a);
    case GNN_EXEC: return gnn_run_kind<GNN_EXEC>(a);
    case GNN_DAGHID: return gnn_run_kind<GNN_DAGHID>(a);
    case GNN_GLOBHID: return gnn_run_kind<GNN_GLOBHID>(a);
  }
  return -1;
}

#include "sss_host.h"

// test-only export: the PCG64 jump-ahead table the host uploads (tests/test_emu_event_batches.py)
extern "C" void sss_test_pcg_jump_table(uint64_t* out) {
  std::vector<uint64_t> t = sss_build_pcg_jump();
  memcpy(out, t.data(), t.size() * sizeof(uint64_t));
}
// ... and the executor-level thresholds (101 entries)
extern "C" void sss_test_lvl_thr_table(uint64_t* out) {
  std::vector<uint64_t> t = sss_build_lvl_thr();
  memcpy(out, t.data(), t.size() * sizeof(uint64_t));
}
