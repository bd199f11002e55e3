/* oracle/sss_oracle.c - scalar CPU restatement of the reference env's reset()/step() path.
 *
 * TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product (spark_sched_sim_amd/) never does. See oracle/README.md.
 *
 * One env, one thread, plain C, deliberately close to the reference's own structure (its
 * functions are restated one by one, each citing the reference file:line it follows), with the
 * Python containers the trajectory depends on modelled explicitly: `set` (pyset.h), insertion
 * ordered `dict` (small vectors), `heapq` (binary heap on the unique (t, counter) key) and the
 * numpy Generator stream (np_random.h). Pinned against trajectories recorded from the reference
 * itself: tests/golden/<set>.npz via tests/test_oracle_golden.py.
 *
 * Citations "ENV:n" = reference spark_sched_sim/spark_sched_sim.py line n, "TRK:n" =
 * components/executor_tracker.py, "JOB:n" = components/job.py, "STG:n" = components/stage.py,
 * "TPCH:n" = data_samplers/tpch.py, "EVQ:n" = components/event.py.
 */
#include <math.h>
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "digest.h"
#include "np_random.h"
#include "pyset.h"
#include "sss_oracle.h"

/* ---------------------------------------------------------------- pack view */

typedef struct {
  int T, L, s_max, total_stages, total_edges, total_durations;
  int n_sizes, n_queries; /* len(QUERY_SIZES), NUM_QUERIES of the trace set (TPCH:14-15); T = n_queries * n_sizes */
  const int32_t *levels, *tmpl_stage_off, *tmpl_edge_off, *stage_num_tasks;
  const double *stage_rough;
  const uint64_t *stage_parent_mask, *stage_child_mask;
  const uint32_t *stage_first_keymask;
  const int32_t *stage_max_first_lvl, *edges, *desc, *durations;
} pack_view;

static int pack_parse(const uint8_t *p, size_t n, pack_view *v) {
  if (n < 8 + 64 || memcmp(p, "SSSPACK2", 8) != 0) return -1;
  const int64_t *h = (const int64_t *)(p + 8);
  v->T = (int)h[0], v->L = (int)h[1], v->s_max = (int)h[2];
  v->total_stages = (int)h[3], v->total_edges = (int)h[4], v->total_durations = (int)h[5];
  if (h[6] != 12) return -1;
  v->n_sizes = h[7] > 0 ? (int)h[7] : 7; /* header word 7: sizes per query; 0 = the reference's seven (TPCH:14) */
  if (v->T < 1 || v->T % v->n_sizes) return -1;
  v->n_queries = v->T / v->n_sizes;
  const int64_t *toc = h + 8;
  const void *sec[12];
  for (int i = 0; i < 12; i++) {
    if ((size_t)(toc[2 * i] + toc[2 * i + 1]) > n) return -1;
    sec[i] = p + toc[2 * i];
  }
  v->levels = sec[0], v->tmpl_stage_off = sec[1], v->tmpl_edge_off = sec[2];
  v->stage_num_tasks = sec[3], v->stage_rough = sec[4], v->stage_parent_mask = sec[5];
  v->stage_child_mask = sec[6], v->stage_first_keymask = sec[7], v->stage_max_first_lvl = sec[8];
  v->edges = sec[9], v->desc = sec[10], v->durations = sec[11];
  return 0;
}

/* ---------------------------------------------------------------- types */

enum { EV_JOB_ARRIVAL = 1, EV_TASK_FINISHED = 2, EV_EXECUTOR_READY = 3 }; /* EVQ:9-12 */

typedef struct {
  double t;
  int64_t cnt;
  int type;
  int job, stage, executor;
} event;

typedef struct { /* STG:4-62 */
  int num_tasks, num_remaining, num_executing, num_completed;
  double most_recent_duration;
  int gs; /* stage row in the pack */
} stage_t;

typedef struct { /* JOB:9-128 */
  int tmpl, n_stages, stage_base, edge_off, n_edges;
  double t_arrival, t_completed;
  uint64_t active_mask;   /* JOB:22  active_stages: ascending stage ids, order-preserving removal */
  uint64_t frontier_mask; /* JOB:25  membership only */
  int n_active;
  int n_local; /* JOB:38 len(local_executors) */
  uint8_t *local;
  int saturated_stage_count;
  int completion_order;
} job_t;

typedef struct { /* components/executor.py:4-44 */
  int task_valid, task_stage; /* executor.task: only .stage_id is ever read back (TPCH:95) */
  int job_id;                 /* -1 = None */
  int is_executing;
} executor_t;

typedef struct {
  int dst, n;
} commit_t;
typedef struct {
  commit_t *v;
  int n, cap;
} commit_vec; /* insertion-ordered dict dst -> n  (TRK:48-51) */

#define POOL_NONE (-1)
#define POOL_COMMON 0

struct sss_oracle {
  pack_view pk;
  uint8_t *pack_copy;
  /* cfg */
  int E, cap_cfg;
  double moving_delay, warmup_delay, mean_interarrival, beta;
  double (*intervals)[2];
  /* episode */
  sso_rng rng;
  double wall_time;
  event *heap;
  int heap_n, heap_cap;
  int64_t counter;
  job_t *jobs;
  int J;
  stage_t *stages;
  int n_stages_total;
  int *stage_job; /* global stage index -> job id */
  executor_t *ex;
  /* tracker (TRK:32-71) */
  int n_pools;
  int *exec_loc;
  pyset *pools;
  uint8_t *pool_exists;
  commit_vec *commits;
  int *n_commit_from, *n_commit_to, *n_moving_to;
  int *supply; /* _total_executor_count[job] */
  int supply_none;
  int curr_source;
  /* env lists */
  int *active_jobs;
  int n_active_jobs;
  int n_completed;
  uint8_t *selected;
  int *selected_list;
  int n_selected;
  int *sched; /* schedulable_stages: global stage index (stage_base + s) */
  int n_sched;
  int *sched_tmp;
  int obs_n_sched; /* len(stage_selection_map) at the last _observe */
  int obs_n_nodes;
  double *dur_buff; /* deque(maxlen=200) ENV:83 */
  int dur_n, dur_head;
  /* status */
  int err;
  int terminated;
  int need_reset;
  jmp_buf jb;
  /* counters for the roofline model */
  int64_t n_events, n_steps;
  /* scratch */
  int32_t *tmp_ids;
  uint8_t *is_sched_flag;
};

#define FAIL(o, code)       \
  do {                      \
    (o)->err = (code);      \
    longjmp((o)->jb, 1);    \
  } while (0)
#define CHECK(o, cond)                      \
  do {                                      \
    if (!(cond)) FAIL(o, SSO_ERR_INVARIANT); \
  } while (0)

/* pool ids: 0 common, 1 + j job pool, 1 + J + stage_base[j] + s stage pool */
static inline int job_pool(const sss_oracle *o, int j) {
  (void)o;
  return 1 + j;
}
static inline int stage_pool(const sss_oracle *o, int j, int s) { return 1 + o->J + o->jobs[j].stage_base + s; }
static inline int pool_job(const sss_oracle *o, int pid) { /* pool_key[0]; -1 = None */
  if (pid <= 0) return -1;
  if (pid <= o->J) return pid - 1;
  return o->stage_job[pid - 1 - o->J];
}
static inline int pool_stage(const sss_oracle *o, int pid) { /* pool_key[1]; -1 = None */
  if (pid <= o->J) return -1;
  int j = pool_job(o, pid);
  return pid - 1 - o->J - o->jobs[j].stage_base;
}
static inline stage_t *STG(sss_oracle *o, int j, int s) { return &o->stages[o->jobs[j].stage_base + s]; }

/* ---------------------------------------------------------------- event queue (EVQ:19-49) */

static inline int ev_less(const event *a, const event *b) { return a->t < b->t || (a->t == b->t && a->cnt < b->cnt); }

static void evq_push(sss_oracle *o, double t, int type, int job, int stage, int executor) {
  if (o->heap_n == o->heap_cap) {
    o->heap_cap = o->heap_cap ? o->heap_cap * 2 : 64;
    o->heap = realloc(o->heap, sizeof(event) * (size_t)o->heap_cap);
  }
  event e = {t, o->counter++, type, job, stage, executor};
  int i = o->heap_n++;
  while (i > 0) {
    int p = (i - 1) / 2;
    if (!ev_less(&e, &o->heap[p])) break;
    o->heap[i] = o->heap[p];
    i = p;
  }
  o->heap[i] = e;
}

static int evq_pop(sss_oracle *o, event *out) {
  if (o->heap_n == 0) return 0;
  *out = o->heap[0];
  event last = o->heap[--o->heap_n];
  int i = 0, n = o->heap_n;
  for (;;) {
    int c = 2 * i + 1;
    if (c >= n) break;
    if (c + 1 < n && ev_less(&o->heap[c + 1], &o->heap[c])) c++;
    if (!ev_less(&o->heap[c], &last)) break;
    o->heap[i] = o->heap[c];
    i = c;
  }
  if (n) o->heap[i] = last;
  return 1;
}

/* ---------------------------------------------------------------- tracker (TRK) */

static void commit_vec_clear(commit_vec *c) { c->n = 0; }

static void trk_add_pool(sss_oracle *o, int pid) { /* TRK:73-96 */
  CHECK(o, !o->pool_exists[pid]);
  pyset_init(&o->pools[pid]);
  o->pool_exists[pid] = 1;
  commit_vec_clear(&o->commits[pid]);
  o->n_commit_from[pid] = 0;
  o->n_commit_to[pid] = 0;
  o->n_moving_to[pid] = 0;
}

static int trk_source_job_id(const sss_oracle *o) { /* TRK:101-105 */
  if (o->curr_source == POOL_NONE || o->curr_source == POOL_COMMON) return -1;
  return pool_job(o, o->curr_source);
}

static int trk_pool_size(const sss_oracle *o, int pid) { return pid == POOL_NONE ? 0 : (int)o->pools[pid].used; }

static int trk_num_committable(sss_oracle *o) { /* TRK:107-113 */
  if (o->curr_source == POOL_NONE) return 0;
  int n = trk_pool_size(o, o->curr_source) - o->n_commit_from[o->curr_source];
  CHECK(o, n >= 0);
  return n;
}

static void trk_increment_commitments(sss_oracle *o, int dst, int n) { /* TRK:226-238 */
  int src = o->curr_source;
  commit_vec *c = &o->commits[src];
  int i;
  for (i = 0; i < c->n; i++)
    if (c->v[i].dst == dst) break;
  if (i < c->n)
    c->v[i].n += n;
  else {
    if (c->n == c->cap) {
      c->cap = c->cap ? c->cap * 2 : 4;
      c->v = realloc(c->v, sizeof(commit_t) * (size_t)c->cap);
    }
    c->v[c->n].dst = dst;
    c->v[c->n].n = n;
    c->n++;
  }
  o->n_commit_from[src] += n;
  o->n_commit_to[dst] += n;
  CHECK(o, trk_pool_size(o, src) >= o->n_commit_from[src]);
}

static void trk_add_commitment(sss_oracle *o, int n, int dst) { /* TRK:148-157 */
  CHECK(o, o->curr_source != POOL_NONE);
  int src_job = pool_job(o, o->curr_source), dst_job = pool_job(o, dst);
  trk_increment_commitments(o, dst, n);
  if (dst_job != src_job) {
    if (dst_job < 0)
      o->supply_none += n;
    else
      o->supply[dst_job] += n;
  }
}

static int trk_remove_commitment(sss_oracle *o, int executor, int dst) { /* TRK:159-176, 240-251 */
  int src = o->exec_loc[executor];
  CHECK(o, src != POOL_NONE);
  commit_vec *c = &o->commits[src];
  int i;
  for (i = 0; i < c->n; i++)
    if (c->v[i].dst == dst) break;
  CHECK(o, i < c->n); /* ValueError("no commitments from ...") */
  c->v[i].n -= 1;
  o->n_commit_from[src] -= 1;
  o->n_commit_to[dst] -= 1;
  CHECK(o, o->n_commit_from[src] >= 0 && o->n_commit_to[dst] >= 0);
  if (c->v[i].n == 0) { /* dict.pop: later keys keep their order */
    memmove(&c->v[i], &c->v[i + 1], sizeof(commit_t) * (size_t)(c->n - i - 1));
    c->n--;
  }
  int src_job = pool_job(o, src), dst_job = pool_job(o, dst);
  if (dst_job != src_job) {
    if (dst_job < 0) {
      o->supply_none -= 1;
      CHECK(o, o->supply_none >= 0);
    } else {
      o->supply[dst_job] -= 1;
      CHECK(o, o->supply[dst_job] >= 0);
    }
  }
  return src;
}

static int trk_peek_commitment(const sss_oracle *o, int pid) { /* TRK:178-183: first key or None */
  if (pid == POOL_NONE || !o->pool_exists[pid] || o->commits[pid].n == 0) return POOL_NONE;
  return o->commits[pid].v[0].dst;
}

static void trk_move_executor_to_pool(sss_oracle *o, int executor, int new_pool, int send) { /* TRK:188-222 */
  if (send) CHECK(o, new_pool > o->J); /* "can only send executors to stages" */
  int old = o->exec_loc[executor];
  if (old != POOL_NONE) {
    int was = pyset_remove(&o->pools[old], executor);
    CHECK(o, was);
    o->exec_loc[executor] = POOL_NONE;
  }
  if (!send) {
    o->exec_loc[executor] = new_pool;
    if (new_pool != POOL_NONE) pyset_add(&o->pools[new_pool], executor);
    return;
  }
  o->n_moving_to[new_pool] += 1;
  int old_job = old != POOL_NONE ? pool_job(o, old) : -1;
  int new_job = pool_job(o, new_pool);
  CHECK(o, old_job != new_job);
  o->supply[new_job] += 1;
  if (old_job >= 0) {
    o->supply[old_job] -= 1;
    CHECK(o, o->supply[old_job] >= 0);
  }
}

/* ---------------------------------------------------------------- job helpers (JOB) */

static void job_attach_executor(sss_oracle *o, job_t *job, int jid, int e) { /* JOB:81-84 */
  CHECK(o, !o->ex[e].task_valid);
  if (!job->local[e]) {
    job->local[e] = 1;
    job->n_local++;
  }
  o->ex[e].job_id = jid;
}

static void job_detach_executor(sss_oracle *o, job_t *job, int e) { /* JOB:86-89 */
  CHECK(o, job->local[e]); /* set.remove raises KeyError otherwise */
  job->local[e] = 0;
  job->n_local--;
  o->ex[e].job_id = -1;
  o->ex[e].task_valid = 0;
}

static inline int stage_completed(const stage_t *s) { return s->num_completed == s->num_tasks; } /* STG:41-43 */
static inline int job_saturated(const job_t *j) { return j->saturated_stage_count == j->n_stages; } /* JOB:53-55 */

/* JOB:65-73, 100-128: returns whether new stages entered the frontier */
static int job_record_stage_completion(sss_oracle *o, int jid, int s) {
  job_t *job = &o->jobs[jid];
  CHECK(o, (job->active_mask >> s) & 1);
  CHECK(o, (job->frontier_mask >> s) & 1); /* set.remove would raise KeyError */
  job->active_mask &= ~(1ull << s);
  job->n_active--;
  job->frontier_mask &= ~(1ull << s);
  if (!stage_completed(STG(o, jid, s))) return 0;
  uint64_t newm = 0;
  uint64_t children = o->pk.stage_child_mask[STG(o, jid, s)->gs];
  for (int c = 0; c < job->n_stages; c++) {
    if (!((children >> c) & 1)) continue;
    if (stage_completed(STG(o, jid, c))) continue;
    uint64_t parents = o->pk.stage_parent_mask[STG(o, jid, c)->gs];
    int ok = 1;
    for (int p = 0; p < job->n_stages; p++)
      if (((parents >> p) & 1) && !stage_completed(STG(o, jid, p))) {
        ok = 0;
        break;
      }
    if (ok) newm |= 1ull << c;
  }
  job->frontier_mask |= newm;
  return newm != 0;
}

/* ---------------------------------------------------------------- data sampler (TPCH) */

static void init_executor_intervals(sss_oracle *o) { /* TPCH:237-262 */
  static const int lv[8] = {5, 10, 20, 40, 50, 60, 80, 100};
  int cap = o->E;
  o->intervals = calloc((size_t)cap + 1, sizeof(double[2]));
#define ROWS(lo, hi, a, b)                                   \
  for (int r_ = (lo); r_ < (hi) && r_ <= cap; r_++) {        \
    if (r_ < 0) continue;                                    \
    o->intervals[r_][0] = (a), o->intervals[r_][1] = (b);    \
  }
  ROWS(0, lv[0] + 1, lv[0], lv[0]);
  for (int i = 0; i < 7; i++) {
    ROWS(lv[i] + 1, lv[i + 1], lv[i], lv[i + 1]);
    if (lv[i + 1] > cap) break;
    ROWS(lv[i + 1], lv[i + 1] + 1, lv[i + 1], lv[i + 1]);
  }
  if (cap > lv[7]) {
    for (int r = lv[7] + 1; r < cap; r++) o->intervals[r][0] = o->intervals[r][1] = lv[7];
  }
#undef ROWS
}

static int level_index(const sss_oracle *o, double key) {
  for (int i = 0; i < o->pk.L; i++)
    if ((double)o->pk.levels[i] == key) return i;
  return -1;
}

/* TPCH:208-214; returns 0 and leaves the stream untouched when the key is missing (KeyError)
 * or the list is empty (ValueError raised by Generator.choice before any draw) */
static int sample_task_duration(sss_oracle *o, int gs, int wave, int lvl, int warmup, double *out) {
  const int32_t *d = &o->pk.desc[((gs * 3 + wave) * o->pk.L + lvl) * 2];
  int off = d[0], len = d[1];
  if (len <= 0) return 0;
  uint32_t i = sso_integers(&o->rng, (uint32_t)len);
  double v = (double)o->pk.durations[off + (int)i];
  if (warmup) v += o->warmup_delay;
  *out = v;
  return 1;
}

static double task_duration(sss_oracle *o, int jid, int s, int e) { /* TPCH:75-106, 216-235 */
  job_t *job = &o->jobs[jid];
  int gs = STG(o, jid, s)->gs;
  int n_local = job->n_local;
  CHECK(o, n_local > 0 && n_local <= o->E);
  double left = o->intervals[n_local][0], right = o->intervals[n_local][1];
  double key;
  if (left == right)
    key = left;
  else {
    int rand_pt = 1 + (int)(sso_random(&o->rng) * (right - left));
    key = ((double)rand_pt <= (double)n_local - left) ? left : right;
  }
  int lvl = level_index(o, key);
  if (lvl < 0 || !((o->pk.stage_first_keymask[gs] >> lvl) & 1)) lvl = o->pk.stage_max_first_lvl[gs];
  double d;
  executor_t *x = &o->ex[e];
  if (!x->task_valid) { /* executor.is_idle */
    if (sample_task_duration(o, gs, 0, lvl, 0, &d)) return d;
    if (sample_task_duration(o, gs, 1, lvl, 1, &d)) return d;
    FAIL(o, SSO_ERR_NO_DURATION);
  }
  if (x->task_stage == s) { /* same stage id (job is not compared, TPCH:95) */
    if (sample_task_duration(o, gs, 2, lvl, 0, &d)) return d;
  }
  if (sample_task_duration(o, gs, 1, lvl, 0, &d)) return d;
  if (sample_task_duration(o, gs, 0, lvl, 0, &d)) return d;
  FAIL(o, SSO_ERR_NO_DURATION);
  return 0;
}

/* ---------------------------------------------------------------- env helpers (ENV) */

static int get_executor_demand(sss_oracle *o, int jid, int s) { /* ENV:566-578 */
  int pid = stage_pool(o, jid, s);
  return STG(o, jid, s)->num_remaining - (o->n_moving_to[pid] + o->n_commit_to[pid]);
}
static int is_stage_saturated(sss_oracle *o, int jid, int s) { return get_executor_demand(o, jid, s) <= 0; } /* ENV:580-582 */

static int is_stage_ready(sss_oracle *o, int jid, int s) { /* ENV:542-555 */
  if (is_stage_saturated(o, jid, s)) return 0;
  uint64_t parents = o->pk.stage_parent_mask[STG(o, jid, s)->gs];
  for (int p = 0; p < o->jobs[jid].n_stages; p++)
    if (((parents >> p) & 1) && !is_stage_saturated(o, jid, p)) return 0;
  return 1;
}

/* ENV:505-540. job_ids == NULL/n == 0 means "falsy" -> all active jobs; source_job_id -1 = None.
 * Both `if not job_ids` (:518) and `if not source_job_id` (:521, which also swallows job id 0)
 * are reproduced. Appends global stage indices to out, returns count. */
static int find_schedulable_stages(sss_oracle *o, const int *job_ids, int n_ids, int source_job_id, int *out) {
  if (n_ids == 0) {
    job_ids = o->active_jobs;
    n_ids = o->n_active_jobs;
  }
  if (source_job_id <= 0) source_job_id = trk_source_job_id(o);
  int n = 0;
  for (int k = 0; k < n_ids; k++) {
    int jid = job_ids[k];
    if (!(jid == source_job_id || o->supply[jid] < o->E)) continue;
    job_t *job = &o->jobs[jid];
    for (int s = 0; s < job->n_stages; s++) {
      if (!((job->active_mask >> s) & 1)) continue;
      int g = job->stage_base + s;
      if (o->selected[g]) continue;
      if (is_stage_ready(o, jid, s)) out[n++] = g;
    }
  }
  return n;
}

static void stage_of_global(const sss_oracle *o, int g, int *jid, int *s) {
  *jid = o->stage_job[g];
  *s = g - o->jobs[*jid].stage_base;
}

static void move_executor_to_stage(sss_oracle *o, int e, int jid, int s);
static void move_idle_executors(sss_oracle *o, int src_pool, const int *ids, int n_ids);

static void execute_next_task(sss_oracle *o, int e, int jid, int s) { /* ENV:584-615 */
  stage_t *st = STG(o, jid, s);
  job_t *job = &o->jobs[jid];
  CHECK(o, st->num_remaining > 0);
  CHECK(o, o->ex[e].job_id == jid);
  CHECK(o, !o->ex[e].is_executing);
  /* stage.launch_next_task STG:53-58 */
  CHECK(o, st->num_executing + st->num_completed < st->num_tasks);
  st->num_remaining -= 1;
  st->num_executing += 1;
  if (st->num_remaining == 0) job->saturated_stage_count += 1;
  double d = task_duration(o, jid, s, e);
  o->ex[e].task_valid = 1;
  o->ex[e].task_stage = s;
  o->ex[e].is_executing = 1;
  st->most_recent_duration = d;
  evq_push(o, o->wall_time + d, EV_TASK_FINISHED, jid, s, e);
}

static void send_executor(sss_oracle *o, int e, int jid, int s) { /* ENV:617-637 */
  CHECK(o, !o->ex[e].is_executing);
  CHECK(o, o->ex[e].job_id != jid);
  trk_move_executor_to_pool(o, e, stage_pool(o, jid, s), 1);
  if (o->ex[e].job_id >= 0) job_detach_executor(o, &o->jobs[o->ex[e].job_id], e);
  evq_push(o, o->wall_time + o->moving_delay, EV_EXECUTOR_READY, jid, s, e);
}

static int find_backup_stage(sss_oracle *o, int e) { /* ENV:821-845; returns global stage or -1 */
  int ejob = o->ex[e].job_id;
  CHECK(o, ejob >= 0);
  int n = find_schedulable_stages(o, &ejob, 1, ejob, o->sched_tmp);
  if (n) return o->sched_tmp[0];
  int *others = o->tmp_ids;
  int n_others = 0;
  for (int k = 0; k < o->n_active_jobs; k++)
    if (o->active_jobs[k] != ejob) others[n_others++] = o->active_jobs[k];
  n = find_schedulable_stages(o, others, n_others, ejob, o->sched_tmp);
  if (n) return o->sched_tmp[0];
  return -1;
}

static void try_backup_schedule(sss_oracle *o, int e) { /* ENV:784-797 */
  int g = find_backup_stage(o, e);
  if (g >= 0) {
    int jid, s;
    stage_of_global(o, g, &jid, &s);
    move_executor_to_stage(o, e, jid, s);
    return;
  }
  int loc = o->exec_loc[e];
  move_idle_executors(o, loc, &e, 1);
}

static void move_executor_to_stage(sss_oracle *o, int e, int jid, int s) { /* ENV:799-819 */
  stage_t *st = STG(o, jid, s);
  if (st->num_remaining == 0) {
    try_backup_schedule(o, e);
    return;
  }
  if (o->ex[e].job_id != jid) {
    send_executor(o, e, jid, s);
    return;
  }
  job_t *job = &o->jobs[jid];
  if (!((job->frontier_mask >> s) & 1)) {
    o->ex[e].task_valid = 0;
    trk_move_executor_to_pool(o, e, job_pool(o, jid), 0);
    return;
  }
  trk_move_executor_to_pool(o, e, stage_pool(o, jid, s), 0);
  execute_next_task(o, e, jid, s);
}

/* ENV:714-728: set(id for id in pool.copy() if not executing) */
static void get_idle_source_executors(sss_oracle *o, int pid, pyset *out) {
  pyset_init(out);
  if (pid == POOL_NONE) return;
  pyset cp;
  pyset_copy(&cp, &o->pools[pid]);
  for (int64_t i = 0; i <= cp.mask; i++) {
    int32_t id = cp.table[i];
    if (id >= 0 && !o->ex[id].is_executing) pyset_add(out, id);
  }
  pyset_free(&cp);
}

/* ENV:745-782. ids == NULL -> list(idle executors of the pool) in set order */
static void move_idle_executors(sss_oracle *o, int src_pool, const int *ids, int n_ids) {
  if (src_pool == POOL_NONE) src_pool = o->curr_source;
  CHECK(o, src_pool != POOL_NONE);
  if (src_pool == POOL_COMMON) return;
  int32_t *own = NULL;
  if (ids == NULL) {
    pyset idle;
    get_idle_source_executors(o, src_pool, &idle);
    own = malloc(sizeof(int32_t) * (size_t)(idle.used + 1));
    n_ids = (int)pyset_list(&idle, own);
    pyset_free(&idle);
    ids = own;
  }
  if (n_ids <= 0) {
    free(own);
    FAIL(o, SSO_ERR_INVARIANT); /* assert executor_ids, "[_move_idle_executors],2" */
  }
  int jid = pool_job(o, src_pool), sid = pool_stage(o, src_pool);
  CHECK(o, jid >= 0);
  int is_sat = job_saturated(&o->jobs[jid]);
  if (sid < 0 && !is_sat) {
    free(own);
    return;
  }
  int dst = is_sat ? POOL_COMMON : job_pool(o, jid);
  for (int k = 0; k < n_ids; k++) {
    int e = ids[k];
    trk_move_executor_to_pool(o, e, dst, 0);
    if (dst == POOL_COMMON) job_detach_executor(o, &o->jobs[jid], e);
  }
  free(own);
}

static void fulfill_commitment(sss_oracle *o, int e, int dst) { /* ENV:699-712 */
  int src = trk_remove_commitment(o, e, dst);
  if (dst == POOL_COMMON) {
    move_idle_executors(o, src, &e, 1);
    return;
  }
  int jid = pool_job(o, dst), s = pool_stage(o, dst);
  CHECK(o, jid >= 0 && s >= 0);
  move_executor_to_stage(o, e, jid, s);
}

static void fulfill_commitments_from_source(sss_oracle *o) { /* ENV:730-743 */
  pyset idle;
  get_idle_source_executors(o, o->curr_source, &idle);
  /* get_source_commitments(): a copy of the dict (TRK:133-134) */
  int n = 0;
  commit_t *cp = NULL;
  if (o->curr_source != POOL_NONE) {
    n = o->commits[o->curr_source].n;
    cp = malloc(sizeof(commit_t) * (size_t)(n + 1));
    memcpy(cp, o->commits[o->curr_source].v, sizeof(commit_t) * (size_t)n);
  }
  for (int i = 0; i < n; i++) {
    int num = cp[i].n;
    while (num && idle.used) {
      int e = pyset_pop(&idle);
      fulfill_commitment(o, e, cp[i].dst);
      num--;
    }
  }
  int left = (int)idle.used;
  free(cp);
  pyset_free(&idle);
  CHECK(o, left == 0);
}

static void commit_remaining_executors(sss_oracle *o) { /* ENV:487-503 */
  int n = trk_num_committable(o);
  if (n > 0) trk_add_commitment(o, n, POOL_COMMON);
}

/* ---------------------------------------------------------------- event handlers */

static void handle_job_arrival(sss_oracle *o, int jid) { /* ENV:428-438 */
  o->active_jobs[o->n_active_jobs++] = jid;
  trk_add_pool(o, job_pool(o, jid));
  o->supply[jid] = 0;
  for (int s = 0; s < o->jobs[jid].n_stages; s++) trk_add_pool(o, stage_pool(o, jid, s));
  if (o->pools[POOL_COMMON].used > 0) o->curr_source = POOL_COMMON;
}

static void handle_executor_arrival(sss_oracle *o, int e, int jid, int s) { /* ENV:440-450 */
  job_t *job = &o->jobs[jid];
  job_attach_executor(o, job, jid, e);
  int pid = stage_pool(o, jid, s);
  o->n_moving_to[pid] -= 1; /* TRK:185-187 */
  CHECK(o, o->n_moving_to[pid] >= 0);
  trk_move_executor_to_pool(o, e, job_pool(o, jid), 0);
  move_executor_to_stage(o, e, jid, s);
}

static void process_job_completion(sss_oracle *o, int jid) { /* ENV:682-697 */
  job_t *job = &o->jobs[jid];
  if (trk_pool_size(o, job_pool(o, jid)) > 0) move_idle_executors(o, job_pool(o, jid), NULL, 0);
  CHECK(o, trk_pool_size(o, job_pool(o, jid)) == 0);
  int k;
  for (k = 0; k < o->n_active_jobs; k++)
    if (o->active_jobs[k] == jid) break;
  CHECK(o, k < o->n_active_jobs);
  memmove(&o->active_jobs[k], &o->active_jobs[k + 1], sizeof(int) * (size_t)(o->n_active_jobs - k - 1));
  o->n_active_jobs--;
  job->completion_order = o->n_completed++;
  job->t_completed = o->wall_time;
  double dur = job->t_completed - job->t_arrival;
  if (o->dur_n < 200)
    o->dur_buff[(o->dur_head + o->dur_n++) % 200] = dur;
  else {
    o->dur_buff[o->dur_head] = dur;
    o->dur_head = (o->dur_head + 1) % 200;
  }
}

static int handle_released_executor(sss_oracle *o, int e, int jid, int s, int frontier_changed) { /* ENV:639-660 */
  int pid = stage_pool(o, jid, s);
  int dst = trk_peek_commitment(o, pid);
  if (dst != POOL_NONE) {
    fulfill_commitment(o, e, dst);
    return 1;
  }
  o->ex[e].task_valid = 0;
  if (frontier_changed) move_idle_executors(o, pid, &e, 1);
  return 0;
}

static void handle_task_completion(sss_oracle *o, int jid, int s, int e) { /* ENV:452-483 */
  stage_t *st = STG(o, jid, s);
  job_t *job = &o->jobs[jid];
  CHECK(o, !stage_completed(st));
  st->num_executing -= 1; /* STG:60-62 */
  st->num_completed += 1;
  o->ex[e].is_executing = 0;
  if (st->num_remaining > 0) {
    execute_next_task(o, e, jid, s);
    return;
  }
  int frontier_changed = 0;
  if (stage_completed(st)) frontier_changed = job_record_stage_completion(o, jid, s); /* ENV:676-680 */
  if (job->n_active == 0) process_job_completion(o, jid);                             /* JOB:49-51 */
  int had_commitment = handle_released_executor(o, e, jid, s, frontier_changed);
  /* _update_executor_source ENV:662-674 */
  if (frontier_changed)
    o->curr_source = job_pool(o, jid);
  else if (!had_commitment)
    o->curr_source = stage_pool(o, jid, s);
}

static void handle_event(sss_oracle *o, const event *ev) { /* ENV:317-318 */
  o->n_events++;
  switch (ev->type) {
    case EV_JOB_ARRIVAL: handle_job_arrival(o, ev->job); break;
    case EV_EXECUTOR_READY: handle_executor_arrival(o, ev->executor, ev->job, ev->stage); break;
    case EV_TASK_FINISHED: handle_task_completion(o, ev->job, ev->stage, ev->executor); break;
  }
}

/* ---------------------------------------------------------------- main loop pieces */

static void resume_simulation(sss_oracle *o) { /* ENV:320-343 */
  int n = 0;
  event ev;
  while (evq_pop(o, &ev)) {
    o->wall_time = ev.t;
    handle_event(o, &ev);
    if (!trk_num_committable(o)) continue;
    n = find_schedulable_stages(o, NULL, 0, -1, o->sched_tmp);
    if (n) break;
    move_idle_executors(o, POOL_NONE, NULL, 0);
    o->curr_source = POOL_NONE;
  }
  /* an exhausted queue leaves the last (empty) list in place (ENV:324,343) */
  memcpy(o->sched, o->sched_tmp, sizeof(int) * (size_t)n);
  o->n_sched = n;
}

static double compute_jobtime(sss_oracle *o, double wall_old, const int *old_active, int n_old) { /* ENV:847-874 */
  double duration = o->wall_time - wall_old;
  if (duration == 0.0) return 0.0;
  pyset all;
  pyset_init(&all);
  for (int k = 0; k < n_old; k++) pyset_add(&all, old_active[k]);
  for (int k = 0; k < o->n_active_jobs; k++) pyset_add(&all, o->active_jobs[k]);
  double job_time = 0.0;
  for (int64_t i = 0; i <= all.mask; i++) {
    int32_t jid = all.table[i];
    if (jid < 0) continue;
    job_t *job = &o->jobs[jid];
    double start = job->t_arrival > wall_old ? job->t_arrival : wall_old;
    double end = job->t_completed < o->wall_time ? job->t_completed : o->wall_time;
    if (o->beta == 0.0)
      job_time += end - start;
    else /* np.exp in the reference: agreement to <= 2 ulp only */
      job_time += sso_exp(-o->beta * 1e-3 * (start - wall_old)) - sso_exp(-o->beta * 1e-3 * (end - wall_old));
  }
  pyset_free(&all);
  if (o->beta > 0.0) job_time /= o->beta;
  return job_time;
}

static int all_jobs_complete(const sss_oracle *o) { return o->n_completed == o->J; } /* ENV:227-229 */

/* ENV:275-315 */
static void take_action(sss_oracle *o, int stage_idx, int num_exec) {
  /* action_space.contains: stage_idx in [-1, n_nodes), num_exec in [1, E] (ENV:85-94,404) */
  if (stage_idx < -1 || stage_idx >= o->obs_n_nodes || num_exec < 1 || num_exec > o->E) FAIL(o, SSO_ERR_ACTION_SPACE);
  if (stage_idx == -1) {
    commit_remaining_executors(o);
    return;
  }
  if (stage_idx >= o->obs_n_sched) FAIL(o, SSO_ERR_STAGE_IDX); /* KeyError on stage_selection_map, ENV:284 */
  int g = o->sched[stage_idx];
  if (num_exec > trk_num_committable(o)) FAIL(o, SSO_ERR_TOO_MANY);
  int jid, s;
  stage_of_global(o, g, &jid, &s);
  int demand = get_executor_demand(o, jid, s); /* ENV:557-564 */
  int n = num_exec < demand ? num_exec : demand;
  CHECK(o, n > 0);
  trk_add_commitment(o, n, stage_pool(o, jid, s));
  o->selected[g] = 1;
  o->selected_list[o->n_selected++] = g;
  /* splice ENV:307-315: bisect over the job ids of the current list */
  int i = 0;
  while (i < o->n_sched) { /* bisect_left */
    int jj, ss;
    stage_of_global(o, o->sched[i], &jj, &ss);
    if (jj >= jid) break;
    i++;
  }
  int hi = i + o->jobs[jid].n_active;
  if (hi > o->n_sched) hi = o->n_sched;
  int j = i;
  while (j < hi) { /* bisect_right within [i, hi) */
    int jj, ss;
    stage_of_global(o, o->sched[j], &jj, &ss);
    if (jj > jid) break;
    j++;
  }
  int n_mid = find_schedulable_stages(o, &jid, 1, -1, o->sched_tmp);
  int n_tail = o->n_sched - j;
  int *tail = malloc(sizeof(int) * (size_t)(n_tail + 1));
  memcpy(tail, &o->sched[j], sizeof(int) * (size_t)n_tail);
  memcpy(&o->sched[i], o->sched_tmp, sizeof(int) * (size_t)n_mid);
  memcpy(&o->sched[i + n_mid], tail, sizeof(int) * (size_t)n_tail);
  free(tail);
  o->n_sched = i + n_mid + n_tail;
}

/* ---------------------------------------------------------------- reset (ENV:127-186) */

static void free_episode(sss_oracle *o) {
  if (o->pools) {
    for (int p = 0; p < o->n_pools; p++)
      if (o->pool_exists[p]) pyset_free(&o->pools[p]);
  }
  if (o->commits)
    for (int p = 0; p < o->n_pools; p++) free(o->commits[p].v);
  if (o->jobs)
    for (int j = 0; j < o->J; j++) free(o->jobs[j].local);
  free(o->jobs), free(o->stages), free(o->pools), free(o->pool_exists), free(o->commits);
  free(o->n_commit_from), free(o->n_commit_to), free(o->n_moving_to), free(o->supply);
  free(o->active_jobs), free(o->selected), free(o->selected_list), free(o->sched), free(o->sched_tmp);
  free(o->tmp_ids), free(o->stage_job), free(o->is_sched_flag);
  o->stage_job = NULL, o->is_sched_flag = NULL;
  o->jobs = NULL, o->stages = NULL, o->pools = NULL, o->pool_exists = NULL, o->commits = NULL;
  o->n_commit_from = o->n_commit_to = o->n_moving_to = o->supply = NULL;
  o->active_jobs = NULL, o->selected = NULL, o->selected_list = o->sched = o->sched_tmp = NULL;
  o->tmp_ids = NULL;
}

int sso_reset(sss_oracle *o, uint64_t seed, double time_limit) {
  o->err = 0;
  if (!(time_limit < INFINITY) && o->cap_cfg <= 0) return o->err = SSO_ERR_NO_LIMIT; /* ENV:137-138 */
  free_episode(o);
  sso_rng_seed(&o->rng, seed); /* ENV:130 */
  o->wall_time = 0;
  o->heap_n = 0;
  o->counter = 0;

  /* job_sequence TPCH:54-73 */
  int cap = 0, J = 0;
  job_t *jobs = NULL;
  double t = 0;
  int total_stages = 0;
  while (t < time_limit && (o->cap_cfg <= 0 || J < o->cap_cfg)) {
    if (J == cap) {
      cap = cap ? cap * 2 : 64;
      jobs = realloc(jobs, sizeof(job_t) * (size_t)cap);
    }
    int q = (int)sso_integers(&o->rng, (uint32_t)o->pk.n_queries);  /* TPCH:177 (query_num - 1) */
    int size = (int)sso_integers(&o->rng, (uint32_t)o->pk.n_sizes); /* TPCH:178 */
    job_t *job = &jobs[J];
    memset(job, 0, sizeof(*job));
    job->tmpl = q * o->pk.n_sizes + size;
    job->n_stages = o->pk.tmpl_stage_off[job->tmpl + 1] - o->pk.tmpl_stage_off[job->tmpl];
    job->edge_off = o->pk.tmpl_edge_off[job->tmpl];
    job->n_edges = o->pk.tmpl_edge_off[job->tmpl + 1] - job->edge_off;
    job->stage_base = total_stages;
    total_stages += job->n_stages;
    job->t_arrival = t;
    job->t_completed = INFINITY;
    job->completion_order = -1;
    J++;
    t += sso_exponential(&o->rng, o->mean_interarrival); /* TPCH:70 */
  }
  o->jobs = jobs;
  o->J = J;
  o->n_stages_total = total_stages;
  o->stages = calloc((size_t)total_stages + 1, sizeof(stage_t));
  o->stage_job = calloc((size_t)total_stages + 1, sizeof(int));
  o->is_sched_flag = calloc((size_t)total_stages + 1, 1);
  for (int j = 0; j < J; j++) {
    job_t *job = &o->jobs[j];
    job->local = calloc((size_t)o->E, 1);
    job->active_mask = job->n_stages == 64 ? ~0ull : ((1ull << job->n_stages) - 1);
    job->n_active = job->n_stages;
    for (int s = 0; s < job->n_stages; s++) {
      stage_t *st = &o->stages[job->stage_base + s];
      st->gs = o->pk.tmpl_stage_off[job->tmpl] + s;
      o->stage_job[job->stage_base + s] = j;
      st->num_tasks = st->num_remaining = o->pk.stage_num_tasks[st->gs];
      st->most_recent_duration = o->pk.stage_rough[st->gs];
      if (o->pk.stage_parent_mask[st->gs] == 0) job->frontier_mask |= 1ull << s; /* JOB:93-111 */
    }
    evq_push(o, job->t_arrival, EV_JOB_ARRIVAL, j, -1, -1); /* ENV:152-154 */
  }

  /* executors + tracker reset (ENV:161-162, TRK:32-71) */
  for (int e = 0; e < o->E; e++) {
    o->ex[e].task_valid = 0, o->ex[e].task_stage = -1, o->ex[e].job_id = -1, o->ex[e].is_executing = 0;
    o->exec_loc[e] = POOL_COMMON;
  }
  o->n_pools = 1 + J + total_stages;
  o->pools = calloc((size_t)o->n_pools, sizeof(pyset));
  o->pool_exists = calloc((size_t)o->n_pools, 1);
  o->commits = calloc((size_t)o->n_pools, sizeof(commit_vec));
  o->n_commit_from = calloc((size_t)o->n_pools, sizeof(int));
  o->n_commit_to = calloc((size_t)o->n_pools, sizeof(int));
  o->n_moving_to = calloc((size_t)o->n_pools, sizeof(int));
  o->supply = calloc((size_t)J + 1, sizeof(int));
  o->supply_none = 0;
  pyset_init(&o->pools[POOL_COMMON]);
  o->pool_exists[POOL_COMMON] = 1;
  for (int e = 0; e < o->E; e++) pyset_add(&o->pools[POOL_COMMON], e); /* set(range(E)) */
  o->curr_source = POOL_COMMON;

  o->active_jobs = calloc((size_t)J + 1, sizeof(int));
  o->n_active_jobs = 0;
  o->n_completed = 0;
  o->selected = calloc((size_t)total_stages + 1, 1);
  o->selected_list = calloc((size_t)total_stages + 1, sizeof(int));
  o->n_selected = 0;
  o->sched = calloc((size_t)total_stages + 1, sizeof(int));
  o->sched_tmp = calloc((size_t)total_stages + 1, sizeof(int));
  o->tmp_ids = calloc((size_t)J + 1, sizeof(int32_t));
  o->n_sched = 0;
  /* job_duration_buff is created in __init__ and survives resets (ENV:83): not cleared here */
  o->terminated = 0;
  o->need_reset = 0;

  if (setjmp(o->jb)) {
    o->need_reset = 1;
    return o->err;
  }
  /* _load_initial_jobs ENV:260-273 */
  while (o->heap_n && o->heap[0].t <= 0) {
    event ev;
    evq_pop(o, &ev);
    handle_job_arrival(o, ev.job);
  }
  o->n_sched = find_schedulable_stages(o, NULL, 0, -1, o->sched);
  /* _observe bookkeeping (ENV:354-356, 403-404) */
  o->obs_n_sched = o->n_sched;
  o->obs_n_nodes = 0;
  for (int k = 0; k < o->n_active_jobs; k++) o->obs_n_nodes += o->jobs[o->active_jobs[k]].n_active;
  return 0;
}

/* ---------------------------------------------------------------- step (ENV:188-221) */

int sso_step(sss_oracle *o, int stage_idx, int num_exec, double *reward, int *terminated) {
  *reward = 0.0;
  *terminated = o->terminated;
  if (o->need_reset || o->terminated) return o->err = SSO_ERR_NEED_RESET;
  o->err = 0;
  int *old_active = NULL;
  if (setjmp(o->jb)) {
    free(old_active);
    /* invalid actions are rejected before any mutation (ENV:276-295): the env stays usable */
    if (o->err != SSO_ERR_ACTION_SPACE && o->err != SSO_ERR_STAGE_IDX && o->err != SSO_ERR_TOO_MANY) o->need_reset = 1;
    return o->err;
  }
  take_action(o, stage_idx, num_exec);
  o->n_steps++;

  if (trk_num_committable(o) && o->n_sched) {
    /* same scheduling round continues: reward 0 (ENV:191-193) */
  } else {
    commit_remaining_executors(o);
    fulfill_commitments_from_source(o);
    o->curr_source = POOL_NONE;
    for (int k = 0; k < o->n_selected; k++) o->selected[o->selected_list[k]] = 0;
    o->n_selected = 0;

    double wall_old = o->wall_time;
    int n_old = o->n_active_jobs;
    old_active = malloc(sizeof(int) * (size_t)(n_old + 1));
    memcpy(old_active, o->active_jobs, sizeof(int) * (size_t)n_old);

    resume_simulation(o);

    double job_time = compute_jobtime(o, wall_old, old_active, n_old);
    free(old_active);
    old_active = NULL;
    *reward = -job_time;
    o->terminated = all_jobs_complete(o);
    *terminated = o->terminated;
    if (!o->terminated && !(trk_num_committable(o) && o->n_sched)) FAIL(o, SSO_ERR_STALLED); /* assert "[step]" ENV:212-215 */
  }
  o->obs_n_sched = o->n_sched;
  o->obs_n_nodes = 0;
  for (int k = 0; k < o->n_active_jobs; k++) o->obs_n_nodes += o->jobs[o->active_jobs[k]].n_active;
  return 0;
}

/* ---------------------------------------------------------------- observation (ENV:345-406) */

void sso_obs_sizes(const sss_oracle *o, sso_obs_info *info) {
  int n_nodes = 0, n_edges = 0;
  for (int k = 0; k < o->n_active_jobs; k++) {
    const job_t *job = &o->jobs[o->active_jobs[k]];
    n_nodes += job->n_active;
    for (int i = 0; i < job->n_edges; i++) {
      int u = o->pk.edges[2 * (job->edge_off + i)], v = o->pk.edges[2 * (job->edge_off + i) + 1];
      if (((job->active_mask >> u) & 1) && ((job->active_mask >> v) & 1)) n_edges++;
    }
  }
  info->n_nodes = n_nodes;
  info->n_edges = n_edges;
  info->n_jobs = o->n_active_jobs;
  info->n_schedulable = o->n_sched;
  info->num_committable_execs = o->curr_source == POOL_NONE ? 0 : trk_pool_size(o, o->curr_source) - o->n_commit_from[o->curr_source];
  int src_job = trk_source_job_id(o);
  info->source_job_idx = o->n_active_jobs; /* ENV:352 */
  for (int k = 0; k < o->n_active_jobs; k++)
    if (o->active_jobs[k] == src_job) info->source_job_idx = k;
  info->wall_time = o->wall_time;
  info->terminated = o->terminated;
  info->num_jobs = o->J;
  info->num_completed = o->n_completed;
}

void sso_obs_fill(const sss_oracle *o, float *nodes, int32_t *edge_links, int32_t *dag_ptr, int32_t *exec_supplies) {
  int n = 0, ne = 0;
  dag_ptr[0] = 0;
  /* is_schedulable flags come from the schedulable list (ENV:354-356) */
  for (int i = 0; i < o->n_sched; i++) o->is_sched_flag[o->sched[i]] = 1;
  for (int k = 0; k < o->n_active_jobs; k++) {
    int jid = o->active_jobs[k];
    const job_t *job = &o->jobs[jid];
    exec_supplies[k] = o->supply[jid];
    int base = n;
    for (int s = 0; s < job->n_stages; s++) {
      if (!((job->active_mask >> s) & 1)) continue;
      const stage_t *st = &o->stages[job->stage_base + s];
      int is_sched = o->is_sched_flag[job->stage_base + s];
      nodes[3 * n + 0] = (float)st->num_remaining;
      nodes[3 * n + 1] = (float)st->most_recent_duration;
      nodes[3 * n + 2] = (float)is_sched;
      n++;
    }
    dag_ptr[k + 1] = n;
    /* utils.subgraph (utils.py:5-22): keep edges with both ends active, relabel by active rank */
    for (int i = 0; i < job->n_edges; i++) {
      int u = o->pk.edges[2 * (job->edge_off + i)], v = o->pk.edges[2 * (job->edge_off + i) + 1];
      if (!(((job->active_mask >> u) & 1) && ((job->active_mask >> v) & 1))) continue;
      edge_links[2 * ne + 0] = base + __builtin_popcountll(job->active_mask & ((1ull << u) - 1));
      edge_links[2 * ne + 1] = base + __builtin_popcountll(job->active_mask & ((1ull << v) - 1));
      ne++;
    }
  }
  for (int i = 0; i < o->n_sched; i++) o->is_sched_flag[o->sched[i]] = 0; /* ENV:374 */
}

void sso_obs_digests(const sss_oracle *o, uint64_t out[4]) {
  sso_obs_info info;
  sso_obs_sizes(o, &info);
  float *nodes = malloc(sizeof(float) * 3 * (size_t)(info.n_nodes + 1));
  int32_t *el = malloc(sizeof(int32_t) * 2 * (size_t)(info.n_edges + 1));
  int32_t *ptr = malloc(sizeof(int32_t) * (size_t)(info.n_jobs + 2));
  int32_t *sup = malloc(sizeof(int32_t) * (size_t)(info.n_jobs + 1));
  sso_obs_fill(o, nodes, el, ptr, sup);
  out[0] = sss_digest_words((const uint32_t *)nodes, 3 * (size_t)info.n_nodes);
  out[1] = sss_digest_words((const uint32_t *)el, 2 * (size_t)info.n_edges);
  out[2] = sss_digest_words((const uint32_t *)ptr, (size_t)info.n_jobs + 1);
  out[3] = sss_digest_words((const uint32_t *)sup, (size_t)info.n_jobs);
  free(nodes), free(el), free(ptr), free(sup);
}

/* ---------------------------------------------------------------- misc accessors */

int sso_last_error(const sss_oracle *o) { return o->err; }
int sso_num_jobs(const sss_oracle *o) { return o->J; }
int64_t sso_event_count(const sss_oracle *o) { return o->n_events; }
int64_t sso_step_count(const sss_oracle *o) { return o->n_steps; }

void sso_job_times(const sss_oracle *o, double *t_arrival, double *t_completed, int32_t *tmpl, int32_t *completion_order) {
  for (int j = 0; j < o->J; j++) {
    t_arrival[j] = o->jobs[j].t_arrival;
    t_completed[j] = o->jobs[j].t_completed;
    tmpl[j] = o->jobs[j].tmpl;
    completion_order[j] = o->jobs[j].completion_order;
  }
}

int sso_active_jobs(const sss_oracle *o, int32_t *out) {
  for (int k = 0; k < o->n_active_jobs; k++) out[k] = o->active_jobs[k];
  return o->n_active_jobs;
}

int sso_duration_buffer(const sss_oracle *o, double *out) { /* deque order, oldest first */
  for (int i = 0; i < o->dur_n; i++) out[i] = o->dur_buff[(o->dur_head + i) % 200];
  return o->dur_n;
}

/* ---------------------------------------------------------------- create / destroy */

sss_oracle *sso_create(const void *pack, size_t pack_bytes, const sso_cfg *cfg) {
  sss_oracle *o = calloc(1, sizeof(*o));
  o->pack_copy = malloc(pack_bytes);
  memcpy(o->pack_copy, pack, pack_bytes);
  if (pack_parse(o->pack_copy, pack_bytes, &o->pk) || cfg->num_executors < 1) {
    free(o->pack_copy);
    free(o);
    return NULL;
  }
  o->E = cfg->num_executors;
  o->cap_cfg = cfg->job_arrival_cap;
  o->moving_delay = cfg->moving_delay;
  o->warmup_delay = cfg->warmup_delay;
  o->mean_interarrival = 1 / cfg->job_arrival_rate; /* TPCH:42 */
  o->beta = cfg->beta;
  init_executor_intervals(o);
  o->ex = calloc((size_t)o->E, sizeof(executor_t));
  o->exec_loc = calloc((size_t)o->E, sizeof(int));
  o->dur_buff = calloc(200, sizeof(double));
  o->need_reset = 1;
  return o;
}

void sso_destroy(sss_oracle *o) {
  if (!o) return;
  free_episode(o);
  free(o->heap), free(o->ex), free(o->exec_loc), free(o->dur_buff), free(o->intervals), free(o->pack_copy);
  free(o);
}

/* whole-episode driver used by bench.py's cpu_baseline leg: runs `policy` (0 = fair / 1 = hash)
 * in C so the timing is the oracle's, not Python's. Returns the number of steps taken. */
static int fair_policy(const sss_oracle *o, int E, int *num_exec);
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int64_t sso_run_episode(sss_oracle *o, uint64_t seed, int policy, int64_t max_steps, double *sum_reward) {
  return sso_run_episode_tl(o, seed, INFINITY, policy, max_steps, sum_reward);
}

/* the same with reset(options={"time_limit": ...}) (ENV:127-186: the limit bounds the arrival sequence) */
int64_t sso_run_episode_tl(sss_oracle *o, uint64_t seed, double time_limit, int policy, int64_t max_steps, double *sum_reward) {
  if (sso_reset(o, seed, time_limit)) return -1;
  int64_t steps = 0;
  double acc = 0;
  /* the observation is materialised every step, as the reference does (ENV:193,221) */
  size_t ne_cap = 0;
  for (int j = 0; j < o->J; j++) ne_cap += (size_t)o->jobs[j].n_edges;
  float *nodes = malloc(sizeof(float) * 3 * ((size_t)o->n_stages_total + 1));
  int32_t *el = malloc(sizeof(int32_t) * 2 * (ne_cap + 1));
  int32_t *ptr = malloc(sizeof(int32_t) * ((size_t)o->J + 2));
  int32_t *sup = malloc(sizeof(int32_t) * ((size_t)o->J + 1));
  while (!o->terminated && steps < max_steps) {
    sso_obs_fill(o, nodes, el, ptr, sup);
    int stage_idx, num_exec;
    if (policy == 0)
      stage_idx = fair_policy(o, o->E, &num_exec);
    else {
      sso_obs_info info;
      sso_obs_sizes(o, &info);
      uint64_t h = splitmix64((seed << 32) ^ (uint64_t)steps), h2 = splitmix64(h);
      stage_idx = info.n_schedulable ? (int)(h % (uint64_t)info.n_schedulable) : -1;
      num_exec = 1 + (int)(h2 % (uint64_t)(info.num_committable_execs > 0 ? info.num_committable_execs : 1));
    }
    double r;
    int term;
    if (sso_step(o, stage_idx, num_exec, &r, &term)) {
      steps = -(int64_t)o->err - 100;
      break;
    }
    acc += r;
    steps++;
  }
  free(nodes), free(el), free(ptr), free(sup);
  if (sum_reward) *sum_reward = acc;
  return steps;
}

/* The reference's fair heuristic, evaluated on the oracle's state instead of the observation
 * (schedulers/heuristics/round_robin.py:14-49, utils.py:5-37). Used only to drive the timed
 * baseline; parity tests drive the oracle from recorded actions or from the Python policy. */
static int find_stage_in_job(const sss_oracle *o, int k, int sched_base_of_job[], int *first_sched_idx) {
  (void)sched_base_of_job;
  int jid = o->active_jobs[k];
  const job_t *job = &o->jobs[jid];
  int selected = -1;
  for (int i = 0; i < o->n_sched; i++) {
    int g = o->sched[i];
    if (g < job->stage_base || g >= job->stage_base + job->n_stages) continue;
    int s = g - job->stage_base;
    /* "frontier" in the obs = no incoming edge from an active stage */
    uint64_t parents = o->pk.stage_parent_mask[o->stages[g].gs];
    if ((parents & job->active_mask) == 0) return i;
    if (selected == -1) selected = i;
    (void)s;
  }
  (void)first_sched_idx;
  return selected;
}

static int fair_policy(const sss_oracle *o, int E, int *num_exec) {
  sso_obs_info info;
  sso_obs_sizes(o, &info);
  int A = info.n_jobs;
  int cap = (E + (A > 1 ? A : 1) - 1) / (A > 1 ? A : 1);
  if (info.source_job_idx < A) {
    int i = find_stage_in_job(o, info.source_job_idx, NULL, NULL);
    if (i != -1) {
      *num_exec = info.num_committable_execs;
      return i;
    }
  }
  for (int k = 0; k < A; k++) {
    int sup = o->supply[o->active_jobs[k]];
    if (sup >= cap || k == info.source_job_idx) continue;
    int i = find_stage_in_job(o, k, NULL, NULL);
    if (i == -1) continue;
    int n = cap - sup;
    *num_exec = info.num_committable_execs < n ? info.num_committable_execs : n;
    return i;
  }
  *num_exec = info.num_committable_execs;
  return -1;
}
