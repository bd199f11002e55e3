// sss_rows.h - row gathers / scatters of the PPO update. What the reference runs here are the indexing
// operations of PyG's message passing and of `torch.cat([x[idx], h[idx], ...])` under autograd (schedulers/decima/scheduler.py:
// 209-232 message passing, :289-318 / :337-385 the score networks' inputs, :246-283 the per-job / per-observation sums) and
// their backward passes (index_select <-> index_add_). In a PPO update at BASELINE config 5 they were a third of the device
// time as library calls (profiles/r04_ppo.md: 0.6 TB/s for rows of 64 bytes).
//
// One kernel family, a "list side" array `a` (row i of the list, leading dimension ld_a floats: it may be a column slice of a
// wider matrix) and "table side" arrays `b`, `c` (row idx[i], contiguous rows of `width` floats):
//   GATHER       a[i] = b[idx[i]]
//   SCATTER_ADD  b[idx[i]] += a[i]                         (float atomics: the order of the additions into a row is not fixed)
//   UPDATE       b[idx[i]] = a[i] + c[idx[i]]              (idx without repeats: the receivers of a DAG layer, forward)
//   TAKE         a[i] = b[idx[i]], b[idx[i]] = 0, c[idx[i]] += a[i]    (idx without repeats: the same, backward)
//   SCATTER      b[idx[i]] = a[i]                          (idx without repeats)
//   SEGMENT_SUM  b[s] = sum of a[i] for idx[s] <= i < idx[s + 1]     (idx: n + 1 row offsets of n segments - the sums over the rows
//                of a job, the jobs of an observation, the edges of a receiving node: no atomics, a fixed order)
// A row is handled by 2^k adjacent lanes, 16 bytes per lane when width, ld_a and the pointers allow it and 4 bytes otherwise;
// a thread has four rows in flight. Bound: HBM (8 bytes of index + 2..3 x 4 x width bytes per row).
#pragma once
#include <stdint.h>

enum { ROWS_GATHER = 0, ROWS_SCATTER_ADD = 1, ROWS_UPDATE = 2, ROWS_TAKE = 3, ROWS_SCATTER = 4, ROWS_SEGMENT_SUM = 5 };

struct SssRowsArgs {
  int64_t n;        // rows of the list
  int64_t ld_a;     // leading dimension of a, in floats (>= width)
  int32_t width;    // floats per row, 1..64
  int32_t op;
  const int64_t* idx;
  float* a;
  float* b;
  float* c;
};

// one element (row i, column j) of an operation: the host backend's loop body and the statement the kernel vectorises
template <class AddFn>
static inline void sss_rows_element(const SssRowsArgs& r, int64_t i, int j, AddFn&& atomic_add) {
  if (r.op == ROWS_SEGMENT_SUM) {
    float v = 0.0f;
    for (int64_t k = r.idx[i]; k < r.idx[i + 1]; k++) v += r.a[k * r.ld_a + j];
    r.b[i * (int64_t)r.width + j] = v;
    return;
  }
  const int64_t t = r.idx[i] * (int64_t)r.width + j, l = i * r.ld_a + j;
  switch (r.op) {
    case ROWS_GATHER: r.a[l] = r.b[t]; break;
    case ROWS_SCATTER_ADD: atomic_add(&r.b[t], r.a[l]); break;
    case ROWS_UPDATE: r.b[t] = r.a[l] + r.c[t]; break;
    case ROWS_SCATTER: r.b[t] = r.a[l]; break;
    default: {
      const float v = r.b[t];
      r.a[l] = v, r.b[t] = 0.0f, r.c[t] += v;
    }
  }
}

// ---- several tables side by side (round 6): the score networks' input rows `cat([t_0[idx_0], t_1[idx_1], ...], -1)` --------------
// Built part by part with GATHER into column slices, a 53-float row is written in four launches of 20 / 64-byte pieces that no
// memory transaction is aligned with (1.6 ms for 5 M rows, as much as the head's forward kernel; the backward pass read the
// gradient rows three times, 64 bytes of every 212 each). Here ONE launch walks the output as a flat array - a wave's instruction
// stores (loads) 64 consecutive floats - and every element finds its part by its column:
//   CONCAT_GATHER       out[i][off_k + j] = table_k[idx_k[i]][j]        (idx_k NULL: row i itself)
//   CONCAT_SCATTER_ADD  table_k[idx_k[i]][j] += out[i][off_k + j]      (float atomics; parts with table_k NULL are skipped)
#define SSS_CONCAT_MAX_PARTS 4
enum { CONCAT_GATHER = 0, CONCAT_SCATTER_ADD = 1 };
struct SssConcatArgs {
  int64_t n;       // rows
  int32_t n_parts;
  int32_t width;   // floats per row of `out` = the sum of the parts' widths, 1..64
  int32_t op;
  int32_t inv;     // ceil(2^20 / width): (t * inv) >> 20 == t / width for every t < 64 * width (checked by the host)
  float* out;      // [n][width], contiguous
  float* table[SSS_CONCAT_MAX_PARTS];          // [rows_k][pw[k]], contiguous
  const int64_t* idx[SSS_CONCAT_MAX_PARTS];    // i64[n] or NULL
  int32_t pw[SSS_CONCAT_MAX_PARTS];            // floats per row of the part
  int32_t end[SSS_CONCAT_MAX_PARTS];           // first column behind the part
};
template <class AddFn>
static inline void sss_concat_element(const SssConcatArgs& r, int64_t i, int c, AddFn&& atomic_add) {
  int k = 0;
  while (k + 1 < r.n_parts && c >= r.end[k]) k++;
  if (!r.table[k]) return;
  const int j = c - (r.end[k] - r.pw[k]);
  float* t = r.table[k] + (r.idx[k] ? r.idx[k][i] : i) * (int64_t)r.pw[k] + j;
  if (r.op == CONCAT_GATHER) r.out[i * (int64_t)r.width + c] = *t;
  else atomic_add(t, r.out[i * (int64_t)r.width + c]);
}

#if defined(__HIPCC__)
#ifndef SSS_ROWS_VEC_ATOMICS
#define SSS_ROWS_VEC_ATOMICS 0
#endif
typedef float sss_rows_v4 __attribute__((ext_vector_type(4)));

// global_atomic_add_f32, no return value (the generic atomicAdd(float*) is a compare-and-swap loop)
static __device__ __forceinline__ void sss_rows_fadd(float* p, float v) { (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v); }

template <int VEC>
struct SssRowsVec;
template <>
struct SssRowsVec<4> {
  typedef sss_rows_v4 T;
  static __device__ T zero() { return (T){0.0f, 0.0f, 0.0f, 0.0f}; }
  static __device__ void atomic_add(float* p, T v) { sss_rows_fadd(p, v.x), sss_rows_fadd(p + 1, v.y), sss_rows_fadd(p + 2, v.z), sss_rows_fadd(p + 3, v.w); }
};
typedef float sss_rows_v2 __attribute__((ext_vector_type(2)));
template <>
struct SssRowsVec<2> {  // 8 bytes per lane: rows of 64-bit values moved as two floats each (decima._take of index arrays)
  typedef sss_rows_v2 T;
  static __device__ T zero() { return (T){0.0f, 0.0f}; }
  static __device__ void atomic_add(float* p, T v) { sss_rows_fadd(p, v.x), sss_rows_fadd(p + 1, v.y); }
};
template <>
struct SssRowsVec<1> {
  typedef float T;
  static __device__ T zero() { return 0.0f; }
  static __device__ void atomic_add(float* p, T v) { sss_rows_fadd(p, v); }
};

// lanes_log: a row has 2^lanes_log lanes, of which ceil(width / VEC) work
template <int VEC, int OP>
__global__ __launch_bounds__(256) void sss_rows_kernel(SssRowsArgs r, int lanes_log) {
  typedef typename SssRowsVec<VEC>::T T;
  constexpr int U = 4;
  const int64_t total = r.n << lanes_log, stride = (int64_t)gridDim.x * 256;
  const int lane_mask = (1 << lanes_log) - 1;
  for (int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += U * stride) {
    int64_t row[U], tab[U];
    bool ok[U];
    int col[U];
    T v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t t = t0 + u * stride;
      col[u] = (int)(t & lane_mask) * VEC;
      row[u] = t >> lanes_log;
      ok[u] = t < total && col[u] < r.width;
      tab[u] = ok[u] ? r.idx[row[u]] * (int64_t)r.width + col[u] : 0;
    }
    if (OP == ROWS_SEGMENT_SUM) {  // (tab: the segment's first row; a lane walks its column of the segment's rows)
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (!ok[u]) continue;
        T acc = SssRowsVec<VEC>::zero();
        const int64_t k1 = r.idx[row[u] + 1];
        for (int64_t k = r.idx[row[u]]; k < k1; k++) acc += *(const T*)(r.a + k * r.ld_a + col[u]);
        *(T*)(r.b + row[u] * (int64_t)r.width + col[u]) = acc;
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      v[u] = w[u] = SssRowsVec<VEC>::zero();
      if (!ok[u]) continue;
      if (OP == ROWS_GATHER || OP == ROWS_TAKE) v[u] = *(const T*)(r.b + tab[u]);
      else v[u] = *(const T*)(r.a + row[u] * r.ld_a + col[u]);
      if (OP == ROWS_UPDATE || OP == ROWS_TAKE) w[u] = *(const T*)(r.c + tab[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!ok[u]) continue;
      if (OP == ROWS_GATHER) *(T*)(r.a + row[u] * r.ld_a + col[u]) = v[u];
      else if (OP == ROWS_SCATTER_ADD) SssRowsVec<VEC>::atomic_add(r.b + tab[u], v[u]);
      else if (OP == ROWS_UPDATE) *(T*)(r.b + tab[u]) = v[u] + w[u];
      else if (OP == ROWS_SCATTER) *(T*)(r.b + tab[u]) = v[u];
      else *(T*)(r.a + row[u] * r.ld_a + col[u]) = v[u], *(T*)(r.b + tab[u]) = SssRowsVec<VEC>::zero(), *(T*)(r.c + tab[u]) = w[u] + v[u];
    }
  }
}

// a wave takes 64 rows at a time: 64 * width consecutive floats of `out`, 64 per instruction. The rows' indices come in first, one
// coalesced load per part into the wave's corner of LDS (an element then finds its source row with an LDS read instead of a
// second dependent trip to memory); eight elements per lane are in flight.
template <int OP>
__global__ __launch_bounds__(256) void sss_concat_kernel(SssConcatArgs r) {
  __shared__ int64_t sidx[4][SSS_CONCAT_MAX_PARTS][64];
  constexpr int F = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_blocks = (r.n + 63) >> 6;
  const uint32_t W = (uint32_t)r.width, inv = (uint32_t)r.inv;
  for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blocks; blk += (int64_t)gridDim.x * 4) {
    const int64_t row0 = blk << 6;
    const int64_t left = r.n - row0;
    const uint32_t elems = (uint32_t)(left < 64 ? left : 64) * W;
    float* o = r.out + row0 * (int64_t)W;
    {
      const int64_t me = row0 + lane < r.n ? row0 + lane : r.n - 1;
#pragma unroll
      for (int k = 0; k < SSS_CONCAT_MAX_PARTS; k++)
        if (k < r.n_parts) sidx[wave][k][lane] = r.idx[k] ? r.idx[k][me] : me;
    }
    for (uint32_t t0 = 0; t0 < elems; t0 += 64 * F) {
      float v[F];
      float* tp[F];
      bool ok[F];
#pragma unroll
      for (int u = 0; u < F; u++) {
        const uint32_t t = t0 + 64 * u + lane;
        const bool in = t < elems;
        const uint32_t tc = in ? t : 0;
        const uint32_t i = (tc * inv) >> 20, c = tc - i * W;
        int k = 0;
#pragma unroll
        for (int p = 0; p + 1 < SSS_CONCAT_MAX_PARTS; p++) k += (p + 1 < r.n_parts && (int)c >= r.end[p]) ? 1 : 0;
        const int j = (int)c - (r.end[k] - r.pw[k]);
        ok[u] = in && r.table[k] != nullptr;
        tp[u] = r.table[k] + sidx[wave][k][i] * (int64_t)r.pw[k] + j;
        v[u] = 0.0f;
        if (ok[u]) v[u] = OP == CONCAT_GATHER ? *tp[u] : o[t];
      }
#pragma unroll
      for (int u = 0; u < F; u++) {
        if (!ok[u]) continue;
        if (OP == CONCAT_GATHER) o[t0 + 64 * u + lane] = v[u];
        else sss_rows_fadd(tp[u], v[u]);
      }
    }
  }
}
static int sss_concat_launch(const SssConcatArgs& r, void* stream) {
  const int64_t waves = (r.n + 63) >> 6;
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 65536) blocks = 65536;  // (the loop strides)
  if (blocks < 1) return 0;
  if (r.op == CONCAT_GATHER) hipLaunchKernelGGL(sss_concat_kernel<CONCAT_GATHER>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r);
  else hipLaunchKernelGGL(sss_concat_kernel<CONCAT_SCATTER_ADD>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r);
  return (int)hipGetLastError();
}

template <int VEC>
static void sss_rows_launch_op(const SssRowsArgs& r, int lanes_log, dim3 grid, hipStream_t st) {
  switch (r.op) {
    case ROWS_GATHER: hipLaunchKernelGGL((sss_rows_kernel<VEC, ROWS_GATHER>), grid, dim3(256), 0, st, r, lanes_log); break;
    case ROWS_SCATTER_ADD: hipLaunchKernelGGL((sss_rows_kernel<VEC, ROWS_SCATTER_ADD>), grid, dim3(256), 0, st, r, lanes_log); break;
    case ROWS_UPDATE: hipLaunchKernelGGL((sss_rows_kernel<VEC, ROWS_UPDATE>), grid, dim3(256), 0, st, r, lanes_log); break;
    case ROWS_SCATTER: hipLaunchKernelGGL((sss_rows_kernel<VEC, ROWS_SCATTER>), grid, dim3(256), 0, st, r, lanes_log); break;
    case ROWS_SEGMENT_SUM: hipLaunchKernelGGL((sss_rows_kernel<VEC, ROWS_SEGMENT_SUM>), grid, dim3(256), 0, st, r, lanes_log); break;
    default: hipLaunchKernelGGL((sss_rows_kernel<VEC, ROWS_TAKE>), grid, dim3(256), 0, st, r, lanes_log); break;
  }
}
static int sss_rows_launch(const SssRowsArgs& r, void* stream) {
  auto aligned = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  // (atomics: one float per lane - a wave's instruction then covers whole rows, 64 bytes per request; with 16 bytes per lane it
  // takes four instructions that each touch a quarter of four times as many rows: measured 2x slower, profiles/r04_ppo.md)
  const bool vec = r.width % 4 == 0 && r.ld_a % 4 == 0 && aligned(r.a) && aligned(r.b) && (!r.c || aligned(r.c)) && (r.op != ROWS_SCATTER_ADD || SSS_ROWS_VEC_ATOMICS);
  auto aligned8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
  // (the data-moving operations only: 8 bytes per lane where 16 do not divide the row - the gathers of 64-bit index arrays)
  const bool vec2 = !vec && (r.op == ROWS_GATHER || r.op == ROWS_SCATTER) && r.width % 2 == 0 && r.ld_a % 2 == 0 && aligned8(r.a) && aligned8(r.b);
  const int per_row = vec ? r.width / 4 : vec2 ? r.width / 2 : r.width;
  int lanes_log = 0;
  while ((1 << lanes_log) < per_row) lanes_log++;
  const int64_t total = r.n << lanes_log, per_block = 256 * 4;
  int64_t blocks = (total + per_block - 1) / per_block;
  if (blocks > 65536 * 16) blocks = 65536 * 16;  // (the loop strides)
  if (blocks < 1) return 0;
  if (vec) sss_rows_launch_op<4>(r, lanes_log, dim3((unsigned)blocks), (hipStream_t)stream);
  else if (vec2 && r.op == ROWS_GATHER) hipLaunchKernelGGL((sss_rows_kernel<2, ROWS_GATHER>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r, lanes_log);
  else if (vec2) hipLaunchKernelGGL((sss_rows_kernel<2, ROWS_SCATTER>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r, lanes_log);
  else sss_rows_launch_op<1>(r, lanes_log, dim3((unsigned)blocks), (hipStream_t)stream);
  return (int)hipGetLastError();
}
#endif
