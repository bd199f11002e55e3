/* oracle/sss_oracle.h - C API of the CPU restatement (TEST INFRASTRUCTURE, see oracle/README.md) */
#ifndef SSS_ORACLE_H
#define SSS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sss_oracle sss_oracle;

typedef struct {
  int32_t num_executors;   /* env_cfg["num_executors"]            ENV:37 */
  int32_t job_arrival_cap; /* env_cfg.get("job_arrival_cap"); <= 0 = None  ENV:48 */
  double job_arrival_rate; /* TPCH:21,42 */
  double moving_delay;     /* ENV:40 */
  double warmup_delay;     /* TPCH:24,43 */
  double beta;             /* ENV:44 */
} sso_cfg;

typedef struct {
  int32_t n_nodes, n_edges, n_jobs, n_schedulable;
  int32_t num_committable_execs, source_job_idx;
  int32_t terminated, num_jobs, num_completed, pad_;
  double wall_time;
} sso_obs_info;

enum {
  SSO_OK = 0,
  SSO_ERR_ACTION_SPACE = 1, /* ValueError: does not belong to the action space (ENV:276-277) */
  SSO_ERR_STAGE_IDX = 2,    /* KeyError on stage_selection_map (ENV:284), SURVEY quirk 4 */
  SSO_ERR_TOO_MANY = 4,     /* ValueError: too many executors requested (ENV:294-295) */
  SSO_ERR_STALLED = 5,      /* AssertionError "[step]" (ENV:212-215) */
  SSO_ERR_NO_DURATION = 6,  /* KeyError/ValueError escaping task_duration (TPCH:88-106) */
  SSO_ERR_INVARIANT = 7,    /* any other reference assert / container error */
  SSO_ERR_NEED_RESET = 8,   /* stepping a finished or failed episode */
  SSO_ERR_NO_LIMIT = 9      /* ValueError: must either have a limit on job arrivals or time (ENV:137-138) */
};

sss_oracle *sso_create(const void *pack, size_t pack_bytes, const sso_cfg *cfg);
void sso_destroy(sss_oracle *o);
int sso_reset(sss_oracle *o, uint64_t seed, double time_limit);
int sso_step(sss_oracle *o, int stage_idx, int num_exec, double *reward, int *terminated);
void sso_obs_sizes(const sss_oracle *o, sso_obs_info *info);
void sso_obs_fill(const sss_oracle *o, float *nodes, int32_t *edge_links, int32_t *dag_ptr, int32_t *exec_supplies);
void sso_obs_digests(const sss_oracle *o, uint64_t out[4]);
int sso_last_error(const sss_oracle *o);
int sso_num_jobs(const sss_oracle *o);
int64_t sso_event_count(const sss_oracle *o);
int64_t sso_step_count(const sss_oracle *o);
void sso_job_times(const sss_oracle *o, double *t_arrival, double *t_completed, int32_t *tmpl, int32_t *completion_order);
int sso_active_jobs(const sss_oracle *o, int32_t *out);
int sso_duration_buffer(const sss_oracle *o, double *out);
int64_t sso_run_episode(sss_oracle *o, uint64_t seed, int policy, int64_t max_steps, double *sum_reward);
int64_t sso_run_episode_tl(sss_oracle *o, uint64_t seed, double time_limit, int policy, int64_t max_steps, double *sum_reward);

#ifdef __cplusplus
}
#endif
#endif
