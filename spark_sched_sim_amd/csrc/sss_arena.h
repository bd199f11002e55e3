// sss_arena.h - the rollout workers' record of observations (trainers/rollout_worker.py:133-159 appends every
// observation to `rollout_buffer`, trainers/trainer.py:208-233 + schedulers/decima/utils.py:117-204 collate them into one batch
// graph for the update). Here the batch graph of ALL steps is built while collecting: the compact graph of a step (written by
// sss_decima_graph_build into the env's capacity buffers, its totals on the device) is appended to an arena at cursors that stay
// on the device, with the ids shifted to arena ids - no device->host read of the step's sizes, which the collection loop
// waited for twice per step before (profiles/r04_ppo.md).
//
// Two launches: copy (one grid row per array; an array is copied as 1 / 4 / 8-byte elements, 8-byte id arrays get the cursor of
// the thing they name added) and advance (one thread: cursors += totals, or the overflow flag when an array would not fit - the
// host keeps enough headroom that this never happens, the kernel never writes outside the arena).
#pragma once
#include <stdint.h>

#ifndef SSS_ARENA_MAX_ARRAYS
#define SSS_ARENA_MAX_ARRAYS 24  // (include/sss.h)
#endif
enum { ARENA_NODES = 0, ARENA_EDGES = 1, ARENA_JOBS = 2, ARENA_OBS = 3 };               // what an array has one row per
enum { ARENA_SHIFT_NONE = 0, ARENA_SHIFT_NODE = 1, ARENA_SHIFT_JOB = 2, ARENA_SHIFT_OBS = 3 };  // what an i64 id array names

struct SssArenaArray {
  const void* src;
  void* dst;
  int32_t elem_bytes;  // 1, 4 or 8
  int32_t per_row;     // elements per row (x: 5)
  int32_t kind;        // ARENA_NODES ..
  int32_t shift;       // ARENA_SHIFT_* (8-byte elements only)
};
struct SssArenaArgs {
  int32_t n_arrays, n_obs;
  const int64_t* totals;  // i64[4]: nodes, edges, jobs of this step (sss_prefix_rows' totals; [3] is not used)
  int64_t* cursor;        // i64[8]: rows appended so far [nodes, edges, jobs, observations], [4] steps appended, [5] overflow flag,
                          // [6], [7]: free
  int64_t capacity[4];    // rows the arena's node / edge / job / observation arrays hold
  SssArenaArray arrays[SSS_ARENA_MAX_ARRAYS];
};

SSS_DEV int64_t arena_rows(const SssArenaArgs& a, int kind) { return kind == ARENA_OBS ? (int64_t)a.n_obs : a.totals[kind]; }
SSS_DEV bool arena_fits(const SssArenaArgs& a) {
  for (int k = 0; k < 4; k++)
    if (a.cursor[k] + arena_rows(a, k) > a.capacity[k]) return false;
  return true;
}
// element e of array k (the host backend's loop body and the kernel's)
SSS_DEV void arena_copy_element(const SssArenaArgs& a, const SssArenaArray& r, int64_t e) {
  const int64_t at = a.cursor[r.kind] * r.per_row + e;
  if (r.elem_bytes == 8) {
    const int64_t v = ((const int64_t*)r.src)[e];
    ((int64_t*)r.dst)[at] = r.shift == ARENA_SHIFT_NONE ? v : v + a.cursor[r.shift == ARENA_SHIFT_NODE ? ARENA_NODES : r.shift == ARENA_SHIFT_JOB ? ARENA_JOBS : ARENA_OBS];
  } else if (r.elem_bytes == 4) {
    ((int32_t*)r.dst)[at] = ((const int32_t*)r.src)[e];
  } else {
    ((uint8_t*)r.dst)[at] = ((const uint8_t*)r.src)[e];
  }
}
SSS_DEV void arena_advance(const SssArenaArgs& a) {
  if (!arena_fits(a)) {
    a.cursor[5] = 1;
    return;
  }
  for (int k = 0; k < 4; k++) a.cursor[k] += arena_rows(a, k);
  a.cursor[4] += 1;
}

#if defined(__HIPCC__)
__global__ __launch_bounds__(256) void sss_arena_copy_kernel(SssArenaArgs a) {
  if (!arena_fits(a)) return;
  const SssArenaArray& r = a.arrays[blockIdx.y];
  const int64_t n = arena_rows(a, r.kind) * r.per_row, stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) arena_copy_element(a, r, e);
}
__global__ void sss_arena_advance_kernel(SssArenaArgs a) { arena_advance(a); }
#endif
