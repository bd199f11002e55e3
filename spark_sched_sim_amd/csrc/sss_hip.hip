// sss_hip.hip - gfx950 build of the C ABI (sss_host.h) on the HIP runtime and of every kernel but the simulator's own (those:
// sss_hip_sim.hip / sss_hip_wide.hip): the Decima graph / policy / GNN kernels, the record and training kernels. Built by __graft_entry__.build() / spark_sched_sim_amd/build.py with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: the f64 arithmetic must round exactly like the reference's; no FMA fusion).
#include <hip/hip_runtime.h>

#include "../../include/sss.h"
#include "sss_layout.h"
#include "sss_narrow.h"
#include "sss_wide.h"
#include <wave_rt.h>
#include "sss_decima.h"
#include "sss_gnn.h"
#include "sss_decima_policy.h"
#include "sss_train.h"
#include "sss_rows.h"
#include "sss_collect.h"

#include <stdint.h>
#include <stdlib.h>
#include "zig_tables.inc"

#define BE_UNAVAILABLE (-1000)
static int be_set_device(int device) { return (int)hipSetDevice(device); }
// launches of a handle go to the handle's device whatever the caller's current device is, and leave
// the caller's current device as it was
struct BeDeviceGuard {
  int prev = -1, dev;
  explicit BeDeviceGuard(int device) : dev(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~BeDeviceGuard() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};
static int be_current_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess ? d : 0;
}
static const char* be_error(int rc) { return hipGetErrorString((hipError_t)rc); }
static void* be_alloc(size_t n) {
  void* p = nullptr;
  return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
}
static void be_free(void* p) {
  if (p) (void)hipFree(p);
}
static int be_h2d(void* dst, const void* src, size_t n) { return (int)hipMemcpy(dst, src, n, hipMemcpyHostToDevice); }

// the simulator kernels themselves are two translation units of their own (sss_hip_sim.hip: up to 64 executors; sss_hip_wide.hip: 65..128)
static int be_launch_reset(const SssKernelArgs& a, int num_envs, const uint64_t* seeds, const double* tl, const uint8_t* mask, void* stream) {
  return sss_narrow_launch_reset(a, num_envs, seeds, tl, mask, stream);
}
static int be_launch_step_bounded(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                                  int budget, uint8_t* ready, void* stream) {
  return sss_narrow_launch_step_bounded(a, num_envs, stage_idx, num_exec, auto_reset, seed_stride, budget, ready, stream);
}
static int be_launch_step(const SssKernelArgs& a, int num_envs, const int32_t* stage_idx, const int32_t* num_exec, int auto_reset, uint64_t seed_stride,
                          void* stream) {
  return sss_narrow_launch_step(a, num_envs, stage_idx, num_exec, auto_reset, seed_stride, stream);
}
static int be_launch_policy(const SssKernelArgs& a, int num_envs, int policy, int param, int32_t* stage_idx, int32_t* num_exec, void* stream) {
  return sss_narrow_launch_policy(a, num_envs, policy, param, stage_idx, num_exec, stream);
}
static int be_launch_rollout(const SssKernelArgs& a, int num_envs, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void* stream) {
  return sss_narrow_launch_rollout(a, num_envs, policy, param, n_steps, auto_reset, seed_stride, stream);
}

static int be_launch_decima(const SssLayout& L, const SssBuffers& B, int E, const SssDecimaArgs& d, void* stream) {
  hipLaunchKernelGGL(sss_decima_graph_kernel, dim3(L.num_envs), dim3(64), (size_t)8 * L.n_cap + (size_t)8 * (L.J_cap + 1), (hipStream_t)stream, L, B, E, d);
  return (int)hipGetLastError();
}

static int be_launch_bit_lists(const SssBitListArgs& a, void* stream) {
  hipLaunchKernelGGL(sss_bit_lists_kernel, dim3((unsigned)a.n_chunks), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
static int be_launch_decima_lists(int num_envs, const SssDecimaListArgs& d, void* stream) {
  hipLaunchKernelGGL(sss_decima_lists_kernel, dim3(num_envs), dim3(64), 0, (hipStream_t)stream, num_envs, d);
  return (int)hipGetLastError();
}

// up to four envs per workgroup behind one staged copy of the parameters (sss_decima_policy.h)
extern "C" __global__ __launch_bounds__(256) void sss_decima_policy_mw_kernel(SssLayout L, SssBuffers B, int E, SssDecimaPolicyArgs d,
                                                                              int envs_per_wg, int lds_per_env) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dp_lds[];
  float* wl = (float*)dp_lds;
  const float* src[7] = {d.w_prep, d.w_msg, d.w_upd, d.w_dag, d.w_glob, d.w_stage, d.w_exec};
  const int off[7] = {DP_W_PREP, DP_W_MSG, DP_W_UPD, DP_W_DAG, DP_W_GLOB, DP_W_STAGE, DP_W_EXEC};
  const int len[7] = {992, 1344, 1344, 1504, 1344, 7681, 6593};  // gnn_mlp_params of the seven MLPs
  for (int m = 0; m < 7; m++)
    for (int i = threadIdx.x; i < len[m]; i += blockDim.x) wl[off[m] + i] = src[m][i];
  __syncthreads();
  int wave = threadIdx.x >> 6;
  int env = blockIdx.x * envs_per_wg + wave;
  if (env >= L.num_envs) return;
  d.w_prep = wl + DP_W_PREP, d.w_msg = wl + DP_W_MSG, d.w_upd = wl + DP_W_UPD, d.w_dag = wl + DP_W_DAG, d.w_glob = wl + DP_W_GLOB;
  d.w_stage = wl + DP_W_STAGE, d.w_exec = wl + DP_W_EXEC;
  decima_policy_wave(L, B, E, d, env, dp_lds + ((DP_W_TOTAL * 4 + 63) & ~63) + (size_t)wave * lds_per_env);
}

static int be_launch_decima_policy(const SssLayout& L, const SssBuffers& B, int E, const SssDecimaPolicyArgs& d, void* stream) {
  const int w_bytes = (DP_W_TOTAL * 4 + 63) & ~63;
  const int per_env = (20 * L.n_cap + 64 + 63) & ~63;
  int k = (160 * 1024 - w_bytes) / per_env;
  if (k < 1) return -1;
  if (k > 4) k = 4;
  size_t lds = (size_t)w_bytes + (size_t)k * per_env;
  static size_t granted[64] = {0};  // per device: the attribute belongs to the device's copy of the kernel
  size_t& g = granted[be_current_device() & 63];
  if (lds > g) {  // more than the default 64 KB of dynamic LDS has to be asked for
    if (hipFuncSetAttribute((const void*)sss_decima_policy_mw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    g = lds;
  }
  hipLaunchKernelGGL(sss_decima_policy_mw_kernel, dim3((L.num_envs + k - 1) / k), dim3(64 * k), lds, (hipStream_t)stream, L, B, E, d, k, per_env);
  return (int)hipGetLastError();
}

static int be_launch_decima_sample(int n_obs, int which, const SssDecimaSampleArgs& d, void* stream) {
  if (which == 0) hipLaunchKernelGGL(sss_decima_sample_stage_kernel, dim3(n_obs), dim3(64), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(sss_decima_sample_exec_kernel, dim3(n_obs), dim3(64), 0, (hipStream_t)stream, d);
  return (int)hipGetLastError();
}

#include "sss_gnn16.h"
#include "sss_gnn_mfma.h"
#include "sss_train16.h"

// one workgroup per row (sss_decima.h prefix_row is the emulator's form of the same thing): a contiguous run of columns per
// thread, the threads' sums scanned inside each wave by shuffles and across the waves through LDS
#define SSS_PREFIX_THREADS 1024  // (a row is one workgroup: the more threads, the fewer dependent rounds of loads per thread)
__global__ __launch_bounds__(SSS_PREFIX_THREADS) void sss_prefix_rows_kernel(SssPrefixArgs a) {
  __shared__ int64_t wave_tot[SSS_PREFIX_THREADS / 64];
  const int row = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (a.n_cols + SSS_PREFIX_THREADS - 1) / SSS_PREFIX_THREADS;
  const int c0 = tid * per, c1 = c0 + per < a.n_cols ? c0 + per : a.n_cols;
  // (the loads unconditional and eight in flight: a masked-out column is read and dropped - a load behind the mask test waits for it)
  auto at = [&](int c) -> int64_t {
    const int32_t v = a.src[row * a.row_stride + c * a.col_stride];
    const uint8_t m = a.mask ? a.mask[c] : (uint8_t)1;
    return m ? (int64_t)v : 0;
  };
  int64_t sum = 0;
#pragma unroll 8
  for (int c = c0; c < c1; c++) sum += at(c);
  int64_t incl = sum;
  for (int s = 1; s < 64; s <<= 1) {
    const int64_t t = __shfl_up(incl, s);
    if (lane >= s) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int64_t base = incl - sum;
  for (int w = 0; w < wave; w++) base += wave_tot[w];
  if (tid == SSS_PREFIX_THREADS - 1) a.totals[row] = base + sum;
#pragma unroll 8
  for (int c = c0; c < c1; c++) {
    const int64_t v = at(c);
    a.off[(int64_t)row * a.n_cols + c] = base;
    if (a.cnt) a.cnt[(int64_t)row * a.n_cols + c] = v;
    base += v;
  }
}
template <int NT>
static void be_launch_wgrad_nt(const SssWgradArgs& a, int mt, dim3 grid, hipStream_t st) {
  switch (mt) {
    case 1: hipLaunchKernelGGL((sss_wgrad_partial_kernel<NT, 1>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((sss_wgrad_partial_kernel<NT, 2>), grid, dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL((sss_wgrad_partial_kernel<NT, 3>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((sss_wgrad_partial_kernel<NT, 4>), grid, dim3(256), 0, st, a); break;
  }
}
static int be_launch_wgrad(const SssWgradArgs& a, void* stream) {
  const int nt = (a.N + 15) / 16, mt = (a.M + 15) / 16;
  const dim3 grid((unsigned)a.n_partials);
  hipStream_t st = (hipStream_t)stream;
  switch (nt) {
    case 1: be_launch_wgrad_nt<1>(a, mt, grid, st); break;
    case 2: be_launch_wgrad_nt<2>(a, mt, grid, st); break;
    case 3: be_launch_wgrad_nt<3>(a, mt, grid, st); break;
    default: be_launch_wgrad_nt<4>(a, mt, grid, st); break;
  }
  if (int rc = (int)hipGetLastError()) return rc;
  hipLaunchKernelGGL(sss_wgrad_reduce_kernel, dim3((unsigned)((a.N * a.M + a.N + 15) / 16)), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void sss_collect_kernel(SssCollectArgs a, int phase) {
  const int b = (int)(blockIdx.x * 256 + threadIdx.x);
  if (b < a.num_envs) collect_env(a, phase, b, [&](int i, int v) { atomicOr(a.flags + i, v); }, [&](int i, int v) { atomicMax(a.flags + i, v); });
}
static int be_launch_rows(const SssRowsArgs& r, void* stream) { return sss_rows_launch(r, stream); }
static int be_launch_concat(const SssConcatArgs& r, void* stream) { return sss_concat_launch(r, stream); }
#include "sss_segcat.h"
static int be_launch_segcat(const SssSegcatArgs& a, int backward, void* stream) { return sss_segcat_launch(a, backward, stream); }
#include "sss_returns.h"
static int be_launch_returns(const SssReturnsArgs& a, void* stream) {
  hipLaunchKernelGGL(sss_returns_kernel, dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
static int be_launch_baselines(const SssBaselineArgs& a, void* stream) {
  hipLaunchKernelGGL(sss_baseline_kernel, dim3((unsigned)((a.T * a.B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
#include "sss_arena.h"
static int be_launch_arena(const SssArenaArgs& a, int64_t rows_hint, void* stream) {
  int64_t bx = rows_hint > 0 ? (rows_hint + 1023) / 1024 : 256;
  bx = bx < 1 ? 1 : bx > 2048 ? 2048 : bx;
  if (a.n_arrays > 0) hipLaunchKernelGGL(sss_arena_copy_kernel, dim3((unsigned)bx, (unsigned)a.n_arrays), dim3(256), 0, (hipStream_t)stream, a);
  if (int rc = (int)hipGetLastError()) return rc;
  hipLaunchKernelGGL(sss_arena_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
static int be_launch_collect(const SssCollectArgs& a, int phase, void* stream) {
  hipLaunchKernelGGL(sss_collect_kernel, dim3((unsigned)((a.num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, phase);
  return (int)hipGetLastError();
}

static int be_launch_prefix_rows(const SssPrefixArgs& a, void* stream) {
  hipLaunchKernelGGL(sss_prefix_rows_kernel, dim3((unsigned)a.n_rows), dim3(SSS_PREFIX_THREADS), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

template <int KIND>
__global__ __launch_bounds__(256) void sss_gnn_kernel(SssGnnArgs a) {
  constexpr int NW = gnn_weight_count<KIND>();
  constexpr int NW2 = (KIND == GNN_LAYER || KIND == GNN_PREP) ? GNN_W_GNN16 : 0;  // (PREP: the update MLP of a fused SINK)
  __shared__ __attribute__((aligned(16))) float w_lds[NW + NW2 + 4];
  for (int i = threadIdx.x; i < NW; i += 256) w_lds[i] = a.w[i];
  if (NW2 && a.w2) for (int i = threadIdx.x; i < NW2; i += 256) w_lds[NW + i] = a.w2[i];
  __syncthreads();
  const int64_t rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) gnn_row<KIND>(a, r, w_lds, w_lds + NW);
}
template <int KIND>
static int gnn_launch_kind(const SssGnnArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  hipLaunchKernelGGL(sss_gnn_kernel<KIND>, dim3((unsigned)((a.n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
// Which formulation a GNN launch takes is fixed when the library is compiled. The product build runs every MLP of the pass on
// the matrix cores (sss_gnn_mfma.h); a TEST build with -DSSS_TEST_VECTOR_FORMS (tests/gpu_variant.py -> tests/_build/) takes the
// vector-unit kernels of sss_gnn.h / sss_gnn16.h / sss_train16.h instead, so that the two can be compared on the same inputs.
#ifdef SSS_TEST_VECTOR_FORMS
static constexpr bool kVectorForms = true;
#else
static constexpr bool kVectorForms = false;
#endif
static int be_launch_gnn(int kind, const SssGnnArgs& a, void* stream) {
  switch (kind) {
    // a DAG layer's receiving nodes: few rows per launch and the longest chain (message MLP per edge + update MLP),
    // nine launches back to back per step: 16 rows per wave on the matrix cores, the three Linears of an MLP chained in
    // registers (sss_gnn_mfma.h; round 2's 16-lanes-per-row form, sss_gnn16.h: 41 -> 25 us per launch at 4096 envs)
    case GNN_LAYER: return kVectorForms ? gnn16_launch<GNN_LAYER>(a, stream) : gnn_layer_mfma_launch(a, stream);
    // The two policy heads (53-64-64-1 / 36-64-64-1, 128 tanh per row). Vector forms: with few rows the chain of one row is
    // the bound and 16 lanes per row win (1024 envs: 45 -> 22 us / 41 -> 29 us); with many rows the launch is
    // throughput-bound and one thread per row, where 64 rows share every weight read, wins (4096 envs: 52 / 42 us
    // against 59 / 69 us). STAGE's n_rows counts all nodes (its list is padded; ~1 in 22 is schedulable).
    // (a.layer != 0: the list holds exactly the schedulable nodes, sss_decima_graph_build's sched_list)
    case GNN_STAGE:
      if (!kVectorForms) return gnn_head_mfma_launch<GNN_STAGE>(a, stream);
      return (a.w16 && a.n_rows <= (a.layer ? 24000 : 22 * 24000)) ? gnn16_launch<GNN_STAGE>(a, stream) : gnn_launch_kind<GNN_STAGE>(a, stream);
    case GNN_EXEC:
      if (!kVectorForms) return gnn_head_mfma_launch<GNN_EXEC>(a, stream);
      return (a.w16 && a.n_rows <= 24000) ? gnn16_launch<GNN_EXEC>(a, stream) : gnn_launch_kind<GNN_EXEC>(a, stream);
    // node rows (a million per launch at 4096 envs): on the matrix cores as well. A PREP launch without the fused SINK keeps
    // the one-thread-per-row form of sss_gnn.h.
    case GNN_PREP: return (!kVectorForms && a.w2 && a.h) ? gnn_rows_mfma_launch<GNN_PREP>(a, stream) : gnn_launch_kind<GNN_PREP>(a, stream);
    case GNN_SINK: return gnn_launch_kind<GNN_SINK>(a, stream);
    case GNN_DAGHID: return kVectorForms ? gnn_launch_kind<GNN_DAGHID>(a, stream) : gnn_rows_mfma_launch<GNN_DAGHID>(a, stream);
    case GNN_GLOBHID: return kVectorForms ? gnn_launch_kind<GNN_GLOBHID>(a, stream) : gnn_rows_mfma_launch<GNN_GLOBHID>(a, stream);
    // copies: one thread per row (sss_gnn.h)
    case GNN_COMMIT: return gnn_launch_kind<GNN_COMMIT>(a, stream);
    case GNN_MERGE: return gnn_launch_kind<GNN_MERGE>(a, stream);
    // range sums + the last Linear: 16 lanes per row (sss_gnn16.h), the same additions in the same order
    case GNN_DAGSUM: return kVectorForms ? gnn_launch_kind<GNN_DAGSUM>(a, stream) : gnn_sum16_launch<GNN_DAGSUM>(a, stream);
    case GNN_GLOBSUM: return kVectorForms ? gnn_launch_kind<GNN_GLOBSUM>(a, stream) : gnn_sum16_launch<GNN_GLOBSUM>(a, stream);
  }
  return -1;
}

// all DAG layers of a pass in one launch, a wave per observation (sss_gnn_mfma.h); BE_UNAVAILABLE: this build has no such kernel
static int be_launch_gnn_layers_obs(const SssGnnArgs& a, const int64_t* obs_node_off, const int64_t* obs_nodes, const int32_t* layer_cnt, int n_obs, int max_depth,
                                    void* stream) {
  if (kVectorForms) return BE_UNAVAILABLE;
  return gnn_layers_obs_launch(a, obs_node_off, obs_nodes, layer_cnt, n_obs, max_depth, stream);
}

#include "sss_host.h"
