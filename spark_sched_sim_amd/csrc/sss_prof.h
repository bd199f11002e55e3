// sss_prof.h - the scoped profiler of the timing builds (tools/debug/evprof3.py: -DSSS_EVPROF3). In the product build
// every macro below expands to nothing; sss_sim.h only carries the one-line PROF3(id) scope markers.
#pragma once
// -DSSS_EVPROF3 (tools/debug/evprof3.py): inclusive shader-clock ticks and call counts of the lane-0
// procedures, in a device-global table read back through sss_debug_prof (timing builds only)
#ifdef SSS_EVPROF3
__device__ unsigned long long g_prof3[96];  // 48 scopes x (ticks, calls)
__device__ unsigned long long g_prof3_min;  // only launches whose do_step took at least this long are recorded (tail census)
SSS_SHARED unsigned long long g_prof3_lds[98];  // per-wave totals, added to the table once per launch (prof3_flush); [96..97]: discarded
struct Prof3Scope {
  int id;
  uint64_t t0;
  __device__ Prof3Scope(int i) : id(i), t0(wave_clock()) {}
  __device__ ~Prof3Scope() {
    if (wave_lane() == 0) g_prof3_lds[2 * id] += wave_clock() - t0, g_prof3_lds[2 * id + 1] += 1;
  }
};
#if defined(SSS_EVPROF3B) || defined(SSS_EVPROF3C) || defined(SSS_EVPROF3D)  // experiments: ids 1..12 time the sections of batch_released_events (3B) / fast_run (3C) / batch_arrival_events (3D) instead of the lane-0 procedures
#define PROF3(id) Prof3Scope prof3_scope_##id((id) >= 1 && (id) <= 12 ? 48 : (id))
#define PROF3_SEC_BEGIN uint64_t prof3_sec_t = wave_clock()
#define PROF3_SEC_(id) do { uint64_t now_ = wave_clock(); if (wave_lane() == 0) g_prof3_lds[2 * (id)] += now_ - prof3_sec_t, g_prof3_lds[2 * (id) + 1] += 1; prof3_sec_t = now_; } while (0)
#if defined(SSS_EVPROF3B)
#define PROF3_SEC(id) PROF3_SEC_(id)
#define PROF3_FSEC(id) ((void)0)
#define PROF3_ASEC(id) ((void)0)
#elif defined(SSS_EVPROF3C)
#define PROF3_SEC(id) ((void)0)
#define PROF3_FSEC(id) PROF3_SEC_(id)
#define PROF3_ASEC(id) ((void)0)
#else
#define PROF3_SEC(id) ((void)0)
#define PROF3_FSEC(id) ((void)0)
#define PROF3_ASEC(id) PROF3_SEC_(id)
#endif
#else
#define PROF3(id) Prof3Scope prof3_scope_##id(id)
#define PROF3_SEC_BEGIN ((void)0)
#define PROF3_SEC(id) ((void)0)
#define PROF3_FSEC(id) ((void)0)
#define PROF3_ASEC(id) ((void)0)
#endif
#define PROF3_CALLS(id, n) ((void)(wave_lane() == 0 ? (g_prof3_lds[2 * (id) + 1] += (n)) : 0))  // count units of work instead of calls
SSS_DEV void prof3_clear() { g_prof3_lds[wave_lane()] = 0; if (wave_lane() < 34) g_prof3_lds[64 + wave_lane()] = 0; }
SSS_DEV void prof3_flush() {
  wave_sync();
  if (g_prof3_lds[2 * 28] < g_prof3_min) return;
  if (g_prof3_lds[wave_lane()]) atomicAdd(&g_prof3[wave_lane()], g_prof3_lds[wave_lane()]);
  if (wave_lane() < 32 && g_prof3_lds[64 + wave_lane()]) atomicAdd(&g_prof3[64 + wave_lane()], g_prof3_lds[64 + wave_lane()]);
}
#else
SSS_DEV void prof3_clear() {}
SSS_DEV void prof3_flush() {}
#define PROF3(id) ((void)0)
#define PROF3_CALLS(id, n) ((void)0)
#define PROF3_SEC_BEGIN ((void)0)
#define PROF3_SEC(id) ((void)0)
#define PROF3_FSEC(id) ((void)0)
#define PROF3_ASEC(id) ((void)0)
#endif

