/* include/sss.h - C ABI of the MI355X-native batched Spark-scheduling simulator.
 *
 * This is the drop-in boundary for the reference's hot path: everything below
 * `SparkSchedSimEnv.reset()/step()` (reference spark_sched_sim/spark_sched_sim.py:127-221),
 * batched over `num_envs` independent environments, one wavefront per environment.
 * The reference is pure Python with no FFI of its own; these entry points are what a binding
 * for this path binds (INTEGRATION.md shows the ctypes stub the reference would add, and
 * spark_sched_sim_amd/binding.py is that stub in this repo).
 *
 * Conventions: plain C types only; every pointer named *_dev is a device pointer owned by the
 * caller (torch tensors in the Python host); calls are asynchronous on the HIP stream passed
 * as `void* stream` (NULL = default stream); return 0 on success, otherwise a negative code
 * and sss_last_error() describes it. A handle is not re-entrant.
 */
#ifndef SSS_H
#define SSS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sss_handle sss_handle;

/* env_cfg of the reference (spark_sched_sim.py:34-52, data_samplers/tpch.py:19-26) */
typedef struct sss_cfg {
  int32_t num_executors;   /* env_cfg["num_executors"], 1..128 (65..128: the wide instantiation of the kernels) */
  int32_t job_arrival_cap; /* env_cfg.get("job_arrival_cap"); <= 0 means None */
  int32_t max_jobs;        /* arena capacity in jobs; 0 = job_arrival_cap (required when cap is None) */
  int32_t reserved;
  double job_arrival_rate; /* tpch.py:21,42 (jobs per ms) */
  double moving_delay;     /* spark_sched_sim.py:40 */
  double warmup_delay;     /* tpch.py:24,43 */
  double beta;             /* spark_sched_sim.py:44 */
} sss_cfg;

/* sizes the caller needs to allocate the buffers below (all per env unless noted) */
typedef struct sss_dims {
  int32_t num_envs, num_executors, job_cap, stage_stride;
  int32_t node_cap;    /* rows of nodes[env]            */
  int32_t edge_cap;    /* rows of edge_links[env]       */
  int32_t obs_i32;     /* ints per env in obs_i32       */
  int32_t obs_f64;     /* doubles per env in obs_f64    */
  int64_t state_bytes; /* whole arena, all envs; must be zero-initialised */
  int64_t env_stride;  /* bytes per env inside the arena */
  /* byte offsets inside one env's arena block of arrays the host may view zero-copy
   * (replaces the attribute reads of reference metrics.py:4-23, rollout_worker.py:122-129) */
  int64_t off_t_arrival, off_t_completed, off_jobs, off_active, off_dur_ring;
  int32_t job_rec_bytes, hdr_bytes;
} sss_dims;

/* raw device pointers of caller-allocated buffers (observation = reference spark_sched_sim.py:345-406) */
typedef struct sss_buffers {
  void* state_dev;            /* uint8 [state_bytes]                                   */
  float* nodes_dev;           /* f32  [num_envs][node_cap][3]  dag_batch.nodes          */
  int32_t* edge_links_dev;    /* i32  [num_envs][edge_cap][2]  dag_batch.edge_links     */
  int32_t* dag_ptr_dev;       /* i32  [num_envs][job_cap + 1]  dag_ptr                  */
  int32_t* exec_supplies_dev; /* i32  [num_envs][job_cap]      exec_supplies            */
  int32_t* obs_i32_dev;       /* i32  [num_envs][8]: n_nodes, n_edges, n_active_jobs, n_schedulable,
                                 num_committable_execs, source_job_idx, terminated, err */
  double* obs_f64_dev;        /* f64  [num_envs][2]: reward, wall_time                  */
} sss_buffers;

/* per-env error codes reported in obs_i32[..][7] (reference exceptions they stand for) */
enum {
  SSS_E_OK = 0,
  SSS_E_ACTION_SPACE = 1, /* ValueError spark_sched_sim.py:276-277 */
  SSS_E_STAGE_IDX = 2,    /* KeyError   spark_sched_sim.py:284     */
  SSS_E_TOO_MANY = 4,     /* ValueError spark_sched_sim.py:294-295 */
  SSS_E_STALLED = 5,      /* AssertionError "[step]" spark_sched_sim.py:212-215 */
  SSS_E_NO_DURATION = 6,  /* KeyError/ValueError out of tpch.py:88-106 */
  SSS_E_INVARIANT = 7,    /* any other reference assert */
  SSS_E_NEED_RESET = 8,   /* step() on a finished / failed episode */
  SSS_E_NO_LIMIT = 9,     /* ValueError spark_sched_sim.py:137-138 */
  SSS_E_CAPACITY = 10     /* more jobs than max_jobs (time-limit mode) */
};

/* Host-only: validates cfg + workload pack (spark_sched_sim_amd/workload.py format) and reports
 * buffer sizes. Replaces nothing in the reference (Python allocates as it goes). */
int sss_query_dims(const sss_cfg* cfg, const void* pack, size_t pack_bytes, int num_envs, sss_dims* out);

/* Replaces SparkSchedSimEnv.__init__ (spark_sched_sim.py:34-125) + TPCHDataSampler.__init__
 * (tpch.py:19-52): copies the pack and the derived constants to device `device`. */
int sss_create(const sss_cfg* cfg, const void* pack, size_t pack_bytes, int num_envs, int device, sss_handle** out);

int sss_bind_buffers(sss_handle* h, const sss_buffers* buffers);

/* Replaces SparkSchedSimEnv.reset(seed, options) (spark_sched_sim.py:127-186) for every env whose
 * mask byte is non-zero (mask_dev NULL = all). seeds_dev: u64[num_envs]; time_limits_dev:
 * f64[num_envs] or NULL (= inf, options["time_limit"]). Writes the first observation. */
int sss_reset(sss_handle* h, const uint64_t* seeds_dev, const double* time_limits_dev, const uint8_t* mask_dev, void* stream);

/* Replaces SparkSchedSimEnv.step(action) (spark_sched_sim.py:188-221) for all envs:
 * action = {"stage_idx": stage_idx_dev[i], "num_exec": num_exec_dev[i]}. Writes reward, wall_time,
 * terminated, err and the next observation. auto_reset != 0: an env found terminated at entry
 * starts its next episode instead (seed = previous seed + seed_stride), reward 0.
 * stage_idx_dev[i] == SSS_SKIP_ENV leaves env i completely untouched (state, outputs, counters):
 * how a rollout collector freezes envs whose episode was truncated by a wrapper
 * (wrappers/stochastic_time_limit.py:26-31) while the rest of the batch keeps stepping. */
#define SSS_SKIP_ENV (-2147483647 - 1)
int sss_step(sss_handle* h, const int32_t* stage_idx_dev, const int32_t* num_exec_dev, int auto_reset, uint64_t seed_stride, void* stream);

/* sss_step with an event budget per launch - the reference's envs run in worker processes of their own, one step() never waits
 * for another env's (trainers/rollout_worker.py:133-206); a launch of sss_step ends with its slowest env. Here an env whose step
 * has taken max_events events (>= 1; batches of events are not cut, so it may overshoot by one batch) stops between two events:
 * ready_dev[i] = 0, its outputs (observation, reward, terminated, err) are NOT written, and the next sss_step_bounded / sss_step
 * call continues that step - its action arguments for env i are then not looked at. ready_dev[i] = 1: the step is complete, the
 * outputs are the ones sss_step writes. SSS_SKIP_ENV envs: ready_dev[i] is left alone. Every env goes through the states and
 * outputs it goes through under sss_step, bit for bit; only the launch in which its k-th step completes differs. */
int sss_step_bounded(sss_handle* h, const int32_t* stage_idx_dev, const int32_t* num_exec_dev, int auto_reset, uint64_t seed_stride, int max_events,
                     uint8_t* ready_dev, void* stream);

/* On-device counterparts of the reference's heuristic Scheduler plugins; they fill one action per
 * env for the next sss_step. policy: 0 = fair (RoundRobinScheduler(dynamic_partition=True),
 * schedulers/heuristics/round_robin.py:7-49), 1 = FIFO (dynamic_partition=False), 2 = the build's
 * counter-based uniform-random policy (param = per-mille probability of stage_idx = -1). */
int sss_policy(sss_handle* h, int policy, int param, int32_t* stage_idx_dev, int32_t* num_exec_dev, void* stream);

/* n_steps x (policy -> step -> observe) per env in ONE launch: the episode loop of reference
 * examples.py:84-102 / trainers/rollout_worker.py:135-157 with an on-device policy. Per-step
 * semantics are exactly those of sss_policy + sss_step; only the last observation survives. */
int sss_rollout(sss_handle* h, int policy, int param, int n_steps, int auto_reset, uint64_t seed_stride, void* stream);

/* Decima's view of the current observations, for all envs in one launch: replaces
 * DecimaObsWrapper.observation (schedulers/decima/env_wrapper.py:69-143), the DAG-layer edge masks
 * (schedulers/decima/utils.py:238-267) and the PyG batch collation (utils.py:117-204). Reads the
 * bound observation buffers only. The envs' nodes / jobs / edges are written back to back into flat
 * arrays at the offsets the caller supplies (exclusive prefix sums of n_nodes / n_jobs / n_edges over
 * the envs with active != 0); M, J, Ed below are the totals.
 *   x f32[M,5]             node features [commit cap / E, +-1 source-job flag, supply / E,
 *                          remaining / num_tasks_scale, remaining * duration / work_scale]
 *   node_obs/loc/job i64[M] env, row inside the env's observation, flat job id
 *   sched_rank i64[M]      index among the env's schedulable stages (what stage_idx means), or -1
 *   gen i32[M]             topological generation inside the active subgraph
 *   node_recv u32[M]       bit l: the node is the source end of an edge of DAG layer l
 *   stage_mask u8[M]
 *   src/dst/edge_obs i64[Ed]  flat node ids of the edge's ends, env
 *   edge_layers u32[Ed]    bit l: the edge is in the reference's edge_masks[l]
 *   job_obs/cap/first i64[J]  env, number of allowed executor counts, flat id of the job's first node
 *   obs_depth i32[num_envs]   number of DAG layers (rows of the reference's edge_masks) per env */
typedef struct sss_decima_graph {
  const uint8_t* active_dev; /* u8[num_envs] or NULL (= all) */
  const int64_t* node_off_dev;
  const int64_t* job_off_dev;
  const int64_t* edge_off_dev;
  float num_tasks_scale; /* 200 in the reference (env_wrapper.py:48) */
  float work_scale;      /* 1e5 (env_wrapper.py:49) */
  float* x_dev;
  int64_t* node_obs_dev;
  int64_t* node_loc_dev;
  int64_t* node_job_dev;
  int64_t* sched_rank_dev;
  int32_t* gen_dev;
  uint32_t* node_recv_dev;
  uint8_t* stage_mask_dev;
  int64_t* src_dev;
  int64_t* dst_dev;
  int64_t* edge_obs_dev;
  uint32_t* edge_layers_dev;
  int64_t* job_obs_dev;
  int64_t* job_cap_dev;
  int64_t* job_first_dev;
  int32_t* obs_depth_dev;
  int64_t* job_nodes_dev; /* i64[J] number of nodes of the job (they follow job_first back to back) */
  int64_t* out_start_dev; /* i64[M] flat id of the node's first out-edge; a node's out-edges are contiguous */
  int32_t* out_deg_dev;   /* i32[M] number of out-edges */
  int32_t* layer_cnt_dev; /* i32[32][num_envs]: [l][b] = nodes of env b that are sources of layer-l edges */
  /* optional (both or neither): sched_off_dev i64[num_envs] = exclusive prefix sums of the envs' schedulable-stage counts
   * (obs_i32[OBS_N_SCHED], 0 for inactive envs); sched_list_dev then receives the flat ids of the schedulable nodes, env by
   * env in node order - the row list of the stage-score launch (sss_gnn_launch STAGE), with no padding */
  const int64_t* sched_off_dev;
  int64_t* sched_list_dev;
  /* optional (all or none): the layers' lists of receiving nodes written by this launch itself. An env reserves its share of
   * a list with a fetch-add on a counter it shares with its BLOCK of envs only - SSS_LIST_SETS = 32 blocks of
   * q = ceil(num_envs / 32) consecutive envs (4096 envs adding to the same addresses: 35 us of a 78 us launch,
   * profiles/r05_graph_kernel.txt) - so layer l's list is up to 32 dense pieces: block s's piece starts at
   * recv_lists_dev[l * recv_stride + node_off_dev[s * q]] and has layer_totals_dev[l * 32 + s] entries (i64[33][32], ZERO on
   * entry; recv_stride >= the total node count; a block's receivers are among its own nodes, so pieces never overlap; row 32
   * receives the largest node count of an env of block s - sss_gnn_encode_args.max_obs_nodes_hint is made of it). The
   * order of the envs inside a piece is not fixed (the layer launches treat rows independently). sss_gnn_encode takes the
   * pieces as they are (recv_stride there): no scan, no list kernel */
  int64_t* layer_totals_dev;
  int64_t* recv_lists_dev;
  int64_t recv_stride;
  /* optional: ANOTHER set of list counters (i64[33][32]) that this launch sets to zero - a caller that builds a graph per step
   * keeps two sets and passes them in turn (this step's zeroed by the previous launch, the previous step's - its consumers are
   * behind this launch in the stream - cleared now), which saves the clearing launch in between */
  int64_t* layer_totals_clear_dev;
  /* number of i64 entries behind layer_totals_dev (and behind layer_totals_clear_dev): must be 33 * 32 = 1056 whenever either is
   * passed. The counters grew from i64[32] to i64[33][32] in round 5; a caller built against the older layout now gets -1 and a
   * message instead of fetch-adds past the end of its 32-entry buffer. */
  int64_t layer_totals_len;
} sss_decima_graph;
int sss_decima_graph_build(sss_handle* h, const sss_decima_graph* g, void* stream);

/* sizeof() of an argument structure of this header as the LIBRARY was compiled with ("sss_cfg", "sss_decima_graph", ...; -1: unknown
 * name): a binding checks its mirrors against it once (spark_sched_sim_amd/binding.py: check_abi; tests/test_abi.py), so that a
 * stale binding fails loudly instead of handing the library a structure of another layout. No reference counterpart (the reference
 * has no native boundary). */
int sss_abi_sizeof(const char* struct_name);

/* The nodes each DAG layer updates, as index lists (what nonzero((node_recv >> l) & 1) returns), for
 * all layers in one launch: layer l's list starts at recv_dev[layer_base[l]]; env_off_dev = exclusive
 * prefix sums of layer_cnt along the env axis. */
typedef struct sss_decima_lists {
  const int64_t* node_off_dev;
  const int64_t* obs_nodes_dev;
  const uint32_t* node_recv_dev;
  const int64_t* env_off_dev; /* i64[32][num_envs] */
  int64_t layer_base[32];
  int64_t* recv_dev;
  int n_layers;
} sss_decima_lists;
int sss_decima_layer_lists(int num_envs, const sss_decima_lists* a, void* stream);

/* The DAG layers' index lists of a batch graph from its per-edge / per-node layer masks (sss_decima_graph_build's edge_layers /
 * node_recv; the reference's `edge_masks[l]` as index lists, decima/utils.py:249-267): for every bit l < n_layers the ascending
 * positions e with bit l of bits_dev[e] set - what torch.nonzero over (bits >> l) & 1 returns, for all layers in two launches.
 *   pass 0: cnt_dev[c * n_layers + l] (i32[n_chunks][n_layers]) = such positions in chunk c = [c * chunk, (c + 1) * chunk);
 *           chunk a multiple of 64, n_chunks = ceil(n / chunk). The caller scans the counts along the chunks (sss_prefix_rows with
 *           src_row_stride 1, src_col_stride n_layers) into off_dev i64[n_layers][n_chunks] and reads the totals;
 *   pass 1: out_dev[base[l] + off_dev[l][c] + rank] = e  (base[l]: where layer l's list starts in out_dev; host values). */
typedef struct sss_bit_list_args {
  const int32_t* bits_dev;
  int64_t n;
  int32_t n_layers, chunk;
  int32_t n_chunks, phase; /* phase: 0 = count, 1 = write ("pass" above) */
  int32_t* cnt_dev;
  const int64_t* off_dev;
  int64_t base[32];
  int64_t* out_dev;
} sss_bit_list_args;
int sss_bit_lists(const sss_bit_list_args* a, void* stream);

/* Exclusive prefix sums and totals of n_rows rows of n_cols non-negative counts, one launch (the offsets the two
 * kernels above are fed: per-env node / edge / job offsets of the compact graph, per-env offsets into each DAG
 * layer's receiver list; what utils.collate_obsns does with torch.cumsum on the host, decima/utils.py:117-204).
 * src_dev[row * src_row_stride + col * src_col_stride] (i32; strides in elements, so that columns of the env's
 * obs_i32 rows can be scanned in place), mask_dev (u8[n_cols], nullable): masked-out columns count as 0.
 * off_dev: i64[n_rows][n_cols] exclusive prefix along the columns, cnt_dev (nullable): i64[n_rows][n_cols] the
 * (masked) counts themselves, totals_dev: i64[n_rows]. */
int sss_prefix_rows(const int32_t* src_dev, int64_t src_row_stride, int64_t src_col_stride, const uint8_t* mask_dev, int n_rows, int n_cols,
                    int64_t* off_dev, int64_t* cnt_dev, int64_t* totals_dev, void* stream);

/* Decima's GNN forward pass for inference (schedulers/decima/scheduler.py:142-385) on a compact graph
 * written by sss_decima_graph_build: one launch per stage of the pass, each evaluating one whole MLP
 * per row with its gather / scatter fused in. Supports the published architecture
 * (config/decima_tpch.yaml:66-78: embed_dim 16, GNN MLPs [32,16] + LeakyReLU, policy MLPs [64,64] +
 * Tanh). `w_dev` = that stage's MLP parameters packed [W1, b1, W2^T, b2, W3, b3] (W1, W3 in torch.nn.Linear's
 * [out,in] layout, the middle layer transposed to [in,out]).
 * kind: 0 PREP (rows = nodes: out = h_init[M,16] from x), 1 SINK (h = h_init where the node's
 * observation has depth 0 [obs_depth given], else 0 for nodes with out-edges, else update(h_init)),
 * 2 LAYER (rows = idx0, -1 = skip: tmp[n] = h_init[n] + update(sum over n's out-edges in DAG layer
 * `layer` of msg(h[dst])); w = msg, w2 = update), 3 COMMIT (h[n] = tmp[n] for the same rows),
 * 4 DAGSUM (rows = jobs: h_dag[j] = sum over its nodes of dag([x,h]), from the hidden vectors
 * 8 DAGHID left in tmp[M,16]), 5 GLOBSUM (rows = observations: h_glob[o] = sum over its jobs of
 * glob(h_dag), from the hidden vectors 9 GLOBHID left in tmp[J,16]), 6 STAGE (rows = idx0, -1 = skip:
 * out[node_obs*n_pad + node_loc] = score), 7 EXEC (rows = (b, c), c < E: out[b*E + c] = score of c+1
 * executors for job idx0[b], -inf where c >= job_cap). No atomics, fixed summation order. */
typedef struct sss_gnn_args {
  int64_t n_rows;
  const float* w_dev;
  const float* w2_dev;
  float slope;
  int num_executors;
  int layer;
  int64_t n_pad;
  const float* x_dev;
  const float* h_init_dev;
  float* h_dev;
  float* tmp_dev;
  float* h_dag_dev;
  float* h_glob_dev;
  float* out_dev;
  const int32_t* out_deg_dev;
  const int32_t* obs_depth_dev; /* SINK, nullable: batch semantics (no per-observation skip) when NULL */
  const int64_t* idx0_dev;
  const int64_t* dst_dev;
  const int64_t* out_start_dev;
  const uint32_t* edge_layers_dev;
  const int64_t* node_job_dev;
  const int64_t* node_obs_dev;
  const int64_t* node_loc_dev;
  const int64_t* job_obs_dev;
  const int64_t* job_first_dev;
  const int64_t* job_cap_dev;
  const int64_t* job_nodes_dev;
  const int64_t* obs_job_off_dev;
  const int64_t* obs_jobs_dev;
  /* LAYER, optional: the message / update MLPs once more, laid out for the 16-lanes-per-row kernel (csrc/sss_gnn16.h:
   * w1[i][g][q] = W1[g+16q][i], b1[g][q], w2[jj][q][g][r] = W2T[jj+16q][g+16r], b2[g][r], w3[k][g] = W3[g][k], b3[g]).
   * NULL: the kernel transposes w_dev / w2_dev while staging them (slower: strided reads in every workgroup). */
  const float* w16_dev;
  const float* w2_16_dev;
  /* LAYER, optional / MERGE (kind 10): i32 per node, bit l set = the node receives in DAG layer l (sss_decima_graph_build's
   * node_recv). With it a LAYER launch needs no COMMIT: each node's embedding alternates between h_dev and tmp_dev
   * (after v updates it is in buffer v & 1; layers run from the highest index down), a layer reads every child's
   * current buffer and writes the receiver's other one. MERGE (rows = nodes), once after the last layer: h = tmp
   * where the number of updates is odd; DAGHID (kind 8) does that on the fly when it is given node_recv_dev. */
  const int32_t* node_recv_dev;
  /* optional: the row count on the device (i64[1]) - for a Decima step without a device->host round trip (the graph's totals
   * stay where sss_prefix_rows left them). n_rows then only sizes the launch grid (any positive guess, e.g. last step's total);
   * every kernel strides over all *n_rows_dev rows. Not for LAYER (its list lengths come from layer_totals in sss_gnn_encode). */
  const int64_t* n_rows_dev;
} sss_gnn_args;
int sss_gnn_launch(int kind, const sss_gnn_args* args, void* stream);

/* Decima's whole decision for every env in ONE launch (one wavefront per env): observation transform,
 * GNN encoder, stage and executor-count scores and the two softmax draws of
 * DecimaScheduler.schedule (schedulers/decima/scheduler.py:71-99) - sss_decima_graph_build +
 * the sss_gnn_launch sequence + sampling, without the intermediate graph or any host round trip.
 * w_*_dev: the seven MLPs packed as for sss_gnn_launch. node_scratch_dev: f32[num_envs][node_cap][53],
 * job_scratch_dev: f32[num_envs][job_cap][32] (work space). Draws are Gumbel-max over a counter-based
 * uniform stream keyed by (rng_seed, rng_counter, env, candidate): pass a new rng_counter per call.
 * Outputs per env: stage_idx / num_exec (for sss_step; stage_idx -1 when nothing is schedulable),
 * Decima's action tuple (stage_sel, job_idx, exec_sel) and its log-probability; optional dense
 * scores (stage_scores_dev f32[num_envs][node_cap], exec_scores_dev f32[num_envs][E]; -inf = masked).
 * Any executor count the simulator takes (1..128: an executor count sits on lane c & 63, two per lane beyond 64). */
typedef struct sss_decima_policy_args {
  const uint8_t* active_dev; /* u8[num_envs] or NULL */
  float num_tasks_scale, work_scale, slope;
  const float* w_prep_dev;
  const float* w_msg_dev;
  const float* w_upd_dev;
  const float* w_dag_dev;
  const float* w_glob_dev;
  const float* w_stage_dev;
  const float* w_exec_dev;
  float* node_scratch_dev;
  float* job_scratch_dev;
  uint64_t rng_seed, rng_counter;
  int32_t* stage_idx_dev;
  int32_t* num_exec_dev;
  int32_t* stage_sel_dev;
  int32_t* job_idx_dev;
  int32_t* exec_sel_dev;
  float* lgprob_dev;
  float* stage_scores_dev; /* nullable */
  float* exec_scores_dev;  /* nullable */
  uint64_t* prof_dev;      /* nullable: u64[num_envs][8] shader cycles per phase + depth + node count */
} sss_decima_policy_args;
int sss_decima_policy(sss_handle* h, const sss_decima_policy_args* a, void* stream);

/* The two softmax draws of DecimaScheduler.schedule (schedulers/decima/scheduler.py:80-99) for the
 * sss_gnn_launch pipeline, one wavefront per observation, same Gumbel-max stream as sss_decima_policy.
 * which = 0: stage draw from stage_scores_dev (f32[n_obs][n_pad], -inf = masked) -> job_gid_dev (the
 * idx0 of the EXEC launch), stage_idx / stage_sel / job_idx, lgprob, any_stage. which = 1: executor
 * count draw from exec_scores_dev (f32[n_obs][E], any E >= 1: lanes stride over the counts) -> num_exec / exec_sel, lgprob += . */
typedef struct sss_decima_sample_args {
  int64_t n_pad;
  int num_executors;
  uint64_t rng_seed, rng_counter;
  const float* stage_scores_dev;
  const float* exec_scores_dev;
  const int64_t* obs_nodes_dev;
  const int64_t* obs_node_off_dev;
  const int64_t* obs_job_off_dev;
  const int64_t* sched_rank_dev;
  const int64_t* node_job_dev;
  int64_t* job_gid_dev;
  int32_t* stage_idx_dev;
  int32_t* num_exec_dev;
  int64_t* stage_sel_dev;
  int64_t* job_idx_dev;
  int64_t* exec_sel_dev;
  float* lgprob_dev;
  uint8_t* any_stage_dev;
} sss_decima_sample_args;
int sss_decima_sample(int n_obs, int which, const sss_decima_sample_args* a, void* stream);

/* Decima's encoder for one batch of observations in ONE call (scheduler.py:142-283: node embeddings by message passing over
 * the DAG layers, per-job and per-observation summaries): the launches sss_gnn_launch offers one by one - PREP, SINK, one LAYER
 * per DAG layer from max_depth - 1 down to 0, DAGHID (+ MERGE), DAGSUM, GLOBHID, GLOBSUM - preceded by the scan and the kernel
 * that build the layers' lists of receiving nodes. The lists' lengths never leave the device: a LAYER launch is sized by n_nodes
 * and reads its row count and list position from layer_totals_dev, so the call enqueues everything and returns (no device->host
 * round trip; with sss_gnn_launch the caller has to read the list sizes back between SINK and the first LAYER).
 * max_depth: an upper bound of the observations' DAG depth (e.g. the deepest template of the workload); layers beyond an
 * observation's own depth are empty. Graph arrays as written by sss_decima_graph_build (dst_dev / edge_layers_dev may be NULL
 * when the batch has no edge); layer_cnt_dev = its i32[32][n_obs] output. Parameters packed as for sss_gnn_launch (w_msg16_dev / w_update16_dev: the optional 16-lane images). Outputs:
 * h_dev f32[n_nodes,16] node embeddings, h_dag_dev f32[n_jobs,16], h_glob_dev f32[n_obs,16]; h_init_dev f32[n_nodes,16] and
 * tmp_dev f32[max(n_nodes, n_jobs),16] are work space, as are env_off_dev i64[32 * n_obs], layer_totals_dev i64[32] and
 * recv_dev i64[recv_cap] with recv_cap >= n_nodes * max_depth. Launches on the CURRENT device's stream (no handle). */
typedef struct sss_gnn_encode_args {
  int64_t n_nodes, n_jobs;
  int32_t n_obs, max_depth;
  float slope;
  int32_t layers_mode; /* the DAG layers: 1 = a launch per layer over the layer's receiving nodes of all observations; 2 = ONE launch,
                          a wave per observation walking its own layers (message passing never leaves an observation; the same
                          arithmetic per node - bit-identical embeddings; needs none of the lists); 0 = the library chooses: 2 for
                          batches of small observations (64 observations or more, max_obs_nodes_hint small: see there), where nine
                          launches at the launch floor cost more than a wave walking a handful of tiles alone, else 1 */
  const float* w_prep_dev;
  const float* w_update_dev;
  const float* w_msg_dev;
  const float* w_dag_dev;
  const float* w_glob_dev;
  const float* w_msg16_dev;     /* nullable */
  const float* w_update16_dev;  /* nullable */
  const float* x_dev;
  const int32_t* out_deg_dev;
  const int32_t* obs_depth_dev;
  const int64_t* node_obs_dev;
  const int64_t* dst_dev;
  const int64_t* out_start_dev;
  const uint32_t* edge_layers_dev;
  const int32_t* node_recv_dev;
  const int64_t* job_first_dev;
  const int64_t* job_nodes_dev;
  const int64_t* obs_job_off_dev;
  const int64_t* obs_jobs_dev;
  const int64_t* obs_node_off_dev;
  const int64_t* obs_nodes_dev;
  const int32_t* layer_cnt_dev;
  float* h_init_dev;
  float* h_dev;
  float* tmp_dev;
  float* h_dag_dev;
  float* h_glob_dev;
  int64_t* env_off_dev;
  int64_t* layer_totals_dev;
  int64_t* recv_dev;
  int64_t recv_cap;
  int64_t recv_stride;         /* 0: this call builds the lists (scan + list kernel into env_off_dev / layer_totals_dev / recv_dev);
                                  > 0: sss_decima_graph_build wrote them - layer l's pieces at recv_dev[l * recv_stride ..], their
                                  lengths in layer_totals_dev (then i64[33][32], see sss_decima_graph; env_off_dev unused, may be
                                  NULL) */
  int64_t layer_rows_hint[32]; /* host values: roughly how many nodes layer l updates (e.g. layer_totals of the previous step,
                                  read back lazily); only sizes the launch grids - every row is processed whatever it says.
                                  -1: no idea (the grid is sized by n_nodes) */
  /* optional, both or neither: the node / job totals on the device (i64[1] each, e.g. sss_prefix_rows' totals_dev) - n_nodes /
   * n_jobs above are then CAPACITIES (what the buffers hold; they also size the grids unless the hints say less) and nothing
   * on the host needs the real totals: the launches read them (sss_gnn_args::n_rows_dev) */
  const int64_t* n_nodes_dev;
  const int64_t* n_jobs_dev;
  int64_t n_nodes_hint, n_jobs_hint; /* with the pointers: roughly the real totals (grid sizes); <= 0: use the capacities */
  int64_t max_obs_nodes_hint;        /* roughly the node count of the LARGEST observation (e.g. row 32 of an earlier pass's
                                        sss_decima_graph counters, read back lazily); <= 0: unknown. layers_mode 0 takes the one
                                        launch while it is at most 192 (up to 2048 observations) / 64 (more): that launch ends
                                        with its largest observation */
} sss_gnn_encode_args;
int sss_gnn_encode(const sss_gnn_encode_args* a, void* stream);

/* The weight and bias gradient of a Linear layer over a minibatch (what autograd's AddmmBackward
 * computes inside the reference's loss.backward(), trainers/ppo.py:129-131 / schedulers/scheduler.py:44-54):
 *   gw[n][m] = sum_k dy[k][n] * x[k][m],  gb[n] = sum_k dy[k][n]   for x f32[K][ldx] (M columns used), dy f32[K][ldy]
 * (N columns used), M, N in 1..64. scratch_dev: f32 workspace of sss_linear_wgrad_scratch(M, N) floats. gb_dev may be
 * NULL. Launches on the CURRENT device's stream `stream` (no handle). fp32 throughout; the sum runs in a fixed order. */
int64_t sss_linear_wgrad_scratch(int M, int N);
int sss_linear_wgrad(const float* x_dev, int64_t ldx, const float* dy_dev, int64_t ldy, int64_t K, int M, int N, float* gw_dev, float* gb_dev,
                     float* scratch_dev, void* stream);

/* The per-step bookkeeping of the rollout workers for all envs at once (trainers/rollout_worker.py:133-159 sync, :162-206
 * async, with spark_sched_sim/wrappers StochasticTimeLimit's truncation rule), two launches around sss_step:
 *   phase 0: stage_idx[b] = active[b] ? stage_sel[b] : SSS_SKIP_ENV, num_exec[b] = max(1, 1 + exec_sel[b])
 *   phase 1: from the step's outputs (obs_f64 = [reward, wall_time], obs_i32[6] terminated, [7] error code) and time_limit:
 *            an env with an error code leaves the collection (pending_reset set, its step not recorded); row t of the
 *            [T][num_envs] record arrays (active, reward - 0 for frozen envs -, the sample, times, reset flag):
 *              sync:  t_before = the env's clock before the step, t_after = after it; the env stops when its episode ends
 *              async: t_before / t_after = simulated time collected before / after; resets = the episode ended (the caller
 *                     resets those envs; their clock restarts at 0); the env stops when `duration` is reached
 *            wall / elapsed / step_counts / active are updated in place;
 *            flags (i32[8], zero on entry, OR-ed): [0] an env failed, [1] an episode ended, [2] an env goes on,
 *            [3] 1 + index of a failed env, [4] an env was recorded.
 * Launches on the CURRENT device's stream `stream` (no handle). */
typedef struct sss_collect_args {
  int32_t num_envs;
  int32_t asynchronous;
  int64_t t;
  double duration;
  const double* obs_f64_dev;
  const int32_t* obs_i32_dev;
  int64_t obs_i32_stride;
  const double* time_limit_dev;
  uint8_t* active_dev;
  double* wall_dev;
  double* elapsed_dev;
  int64_t* step_counts_dev;
  uint8_t* pending_reset_dev;
  const int64_t* stage_sel_dev;
  const int64_t* job_idx_dev;
  const int64_t* exec_sel_dev;
  const float* lgprob_dev;
  int32_t* stage_idx_dev;
  int32_t* num_exec_dev;
  uint8_t* rec_active_dev;
  double* rec_t_before_dev;
  double* rec_t_after_dev;
  double* rec_rewards_dev;
  int64_t* rec_stage_sel_dev;
  int64_t* rec_job_idx_dev;
  int64_t* rec_exec_sel_dev;
  float* rec_lgprobs_dev;
  uint8_t* rec_resets_dev;
  int32_t* flags_dev;
  const uint8_t* in_group_dev; /* nullable: only envs with in_group[b] != 0 take part in this call (phase 0 gives the others
                                  SSS_SKIP_ENV, phase 1 leaves their state and their entries of row t alone): the caller can
                                  keep two groups of envs one step apart on two streams, sharing the record */
} sss_collect_args;
int sss_collect_step(const sss_collect_args* a, int phase, void* stream);

/* One MLP of Decima's networks (schedulers/decima/utils.py:44-64: Linear - act - Linear - act - Linear) over a minibatch of
 * rows, forward and backward, for the PPO update (trainers/ppo.py:104-138 -> scheduler.py:101-139 -> the nn.Sequential
 * forward / autograd backward of every MLP). All tensors f32, row-major, contiguous.
 *   forward:  a1 = act(W1 x + b1) [rows,h1], a2 = act(W2 a1 + b2) [rows,h2], y = W3 a2 + b3 [rows,out_dim]   (a1, a2, y written)
 *   backward: g2 = (W3^T dy) * act'(a2), g1 = (W2^T g2) * act'(a1), dx = W1^T g1 (dx_dev NULL: not computed) - g1 / g2 are the
 *             gradients w.r.t. the two hidden layers' pre-activations, so that the parameter gradients are
 *             sss_linear_wgrad(x = a2, dy = dy), (a1, g2), (x, g1).
 * w_dev: the MLP's parameters packed [W1 (h1 x in), b1, W2^T (h1 x h2), b2, W3 (out x h2), b3] as for sss_gnn_launch.
 * act: 0 LeakyReLU(slope), 1 Tanh. Shapes: the seven MLPs of config/decima_tpch.yaml:66-78 - (in, 32, 16, 16, LeakyReLU)
 * with in = 5 / 16 / 21 and (in, 64, 64, 1, Tanh) with in = 53 / 36; sss_mlp_supported says whether a shape is one of them
 * (anything else: error). Launches on the CURRENT device's stream `stream` (no handle). fp32 FMAs in a fixed order. */
typedef struct sss_mlp_args {
  int64_t rows;
  int32_t in_dim, h1, h2, out_dim;
  int32_t act;
  float slope;
  const float* w_dev;
  const float* x_dev;  /* forward */
  float* a1_dev;
  float* a2_dev;
  float* y_dev;        /* forward */
  const float* dy_dev; /* backward */
  float* g1_dev;       /* backward */
  float* g2_dev;       /* backward */
  float* dx_dev;       /* backward, nullable */
  /* the input rows in two pieces (sss_mlp_split_supported(in_dim); NULL otherwise): columns 0 .. in_dim - 17 from x_dev (rows of
   * in_dim - 16 floats), the last 16 columns from x2_dev (rows of 16 floats) - the DAG encoder's torch.cat([x, h_node], -1)
   * (schedulers/decima/scheduler.py:246-262) is never built; the first Linear then adds the x2 features first (the MLP on the
   * concatenation up to the order of fp32 additions). sss_mlp_forward and sss_mlp_backward_wgrad without stored activations
   * only; dx2_dev (sss_mlp_backward_wgrad, instead of dx_dev): the gradient w.r.t. the x2 piece, f32[rows][16] */
  const float* x2_dev;
  float* dx2_dev;
} sss_mlp_args;
int sss_mlp_supported(int in_dim, int h1, int h2, int out_dim, int act);
/* 1: for the (in_dim) -> 32 -> 16 -> 16 LeakyReLU MLP sss_mlp_forward may be called with a1_dev == a2_dev == NULL (the hidden
 * activations are not stored) and sss_mlp_backward_wgrad with the same two NULL: it then computes them again from x_dev - same
 * instructions, same bits as the stored ones. Stored activations are two thirds of these kernels' memory traffic and 48 floats
 * per row of the update's memory (what autograd keeps alive for nn.Sequential in the reference, ppo.py:104-138). 0: this build keeps them. */
int sss_mlp_recompute_supported(int in_dim);
/* 1: sss_mlp_forward / sss_mlp_backward_wgrad of the (in_dim) -> 32 -> 16 -> 16 MLP take x2_dev / dx2_dev (see sss_mlp_args) */
int sss_mlp_split_supported(int in_dim);
int sss_mlp_forward(const sss_mlp_args* a, void* stream);
int sss_mlp_backward(const sss_mlp_args* a, void* stream);
/* sss_mlp_backward with the six parameter gradients in the same pass, for the (5 | 16 | 21) -> 32 -> 16 -> 16 LeakyReLU MLPs and
 * (sss_mlp_wgrad_scratch(in_dim) > 0 says whether this build does) the two policy heads (53 | 36) -> 64 -> 64 -> 1 Tanh:
 * x_dev, a1_dev, a2_dev (both NULL: recomputed from x_dev, see sss_mlp_recompute_supported - the GNN-shaped MLPs only), dy_dev in,
 * dx_dev out (nullable); g1 / g2 are not written. The weight / bias gradients are ADDED to
 * per-workgroup slots of acc_dev (f32[sss_mlp_wgrad_scratch(in_dim)], 0 for any other input width; zeroed by the caller before the
 * first call of a group of calls whose gradients belong together - the layers of the message passing); sss_mlp_wgrad_finish adds
 * the slots in a fixed order into gw1 [h1][in_dim], gb1 [h1], gw2 [h2][h1], gb2 [h2], gw3 [out][h2], gb3 [out]. Same inputs, same
 * bits. What the reference runs here: autograd's AddmmBackward / TanhBackward chain inside loss.backward() (trainers/ppo.py:129-131). */
int64_t sss_mlp_wgrad_scratch(int in_dim);
int sss_mlp_backward_wgrad(const sss_mlp_args* a, float* acc_dev, void* stream);
int sss_mlp_wgrad_finish(int in_dim, const float* acc_dev, float* gw1_dev, float* gb1_dev, float* gw2_dev, float* gb2_dev, float* gw3_dev, float* gb3_dev, void* stream);

/* The rollout workers' record of observations, built while collecting (trainers/rollout_worker.py:133-159 appends every
 * observation to its buffer; trainers/trainer.py:208-233 with schedulers/decima/utils.py:117-204 collates them into one batch graph
 * for the update): the compact graph of one step - the arrays sss_decima_graph_build wrote, `totals_dev` = sss_prefix_rows'
 * totals [nodes, edges, jobs, ..] - is appended to the arena's arrays at the cursors in `cursor_dev` (i64[8], on the device:
 * [0..3] rows appended so far of the node / edge / job / observation arrays, [4] steps appended, [5] set to 1 instead when
 * something would not fit `capacity` - nothing is written then). An array is copied as `per_row` elements of `elem_bytes`
 * (1, 4, 8) per row of its `kind` (0 nodes, 1 edges, 2 jobs, 3 observations = n_obs rows per step); 8-byte id arrays get the
 * cursor of what they name added (`shift`: 0 none, 1 node ids, 2 job ids, 3 observation ids). No device->host traffic.
 * `rows_hint`: roughly the largest number of elements an array has this step (grid size only; <= 0: a default). */
#define SSS_ARENA_MAX_ARRAYS 24
typedef struct sss_arena_array {
  const void* src_dev;
  void* dst_dev;
  int32_t elem_bytes, per_row, kind, shift;
} sss_arena_array;
typedef struct sss_arena_args {
  int32_t n_arrays, n_obs;
  const int64_t* totals_dev;
  int64_t* cursor_dev;
  int64_t capacity[4];
  int64_t rows_hint;
  sss_arena_array arrays[SSS_ARENA_MAX_ARRAYS];
} sss_arena_args;
int sss_arena_append(const sss_arena_args* a, void* stream);

/* What the trainer computes from the collected rollouts before the PPO epochs, over [T][B] records (row = step, column = env;
 * active_dev u8[T][B] marks the rows an env recorded - a prefix of its column), all f64, row-major, on the CURRENT device:
 *   sss_discounted_returns   R_k = r_k + exp(-beta * 1e-3 * (t_after_k - t_before_k)) * R_{k+1} per env from its last row
 *                            (trainers/utils/returns_calculator.py:67-76); out = 0 on rows that are not active
 *   sss_sequence_baselines   envs g*R .. g*R+R-1 are the rollouts of one job sequence: out[t][b] = mean over the sequence's
 *                            rollouts j of numpy.interp(times[t][b], times[:n_j][j], values[:n_j][j]) (trainers/utils/baselines.py:
 *                            12-37, trainer.py:206-207); n_dev i64[B] = recorded rows per env; skip_empty != 0: rollouts with
 *                            n = 0 are left out of the mean. The arithmetic is numpy's, operation by operation - the mean's sum in
 *                            numpy's pairwise order (sequential below 8 rollouts, eight partial sums from 8 on, halved above 128). */
typedef struct sss_returns_args {
  int64_t T, B;
  const uint8_t* active_dev;
  const double* t_before_dev;
  const double* t_after_dev;
  const double* rewards_dev;
  double beta;
  double* out_dev;
} sss_returns_args;
typedef struct sss_baseline_args {
  int64_t T, B;
  int32_t R, skip_empty;
  const uint8_t* active_dev;
  const double* times_dev;
  const double* values_dev;
  const int64_t* n_dev;
  double* out_dev;
} sss_baseline_args;
int sss_discounted_returns(const sss_returns_args* a, void* stream);
int sss_sequence_baselines(const sss_baseline_args* a, void* stream);

/* Row gathers / scatters of the PPO update (what PyG's message passing, the score networks' `torch.cat([x[idx], h[idx], ..])`
 * inputs and the per-job / per-observation sums run as index_select / index_add_ under autograd in the reference:
 * schedulers/decima/scheduler.py:209-232, :246-283, :289-318, :337-385). `a` is the list side (row i, leading dimension ld_a
 * floats - it may be a column slice of a wider matrix), `b` / `c` the table side (row idx[i], contiguous rows of `width` floats):
 *   SSS_ROWS_GATHER       a[i] = b[idx[i]]
 *   SSS_ROWS_SCATTER_ADD  b[idx[i]] += a[i]                  (float atomics: the order of the additions into a row is not fixed)
 *   SSS_ROWS_UPDATE       b[idx[i]] = a[i] + c[idx[i]]       (idx without repeats)
 *   SSS_ROWS_TAKE         a[i] = b[idx[i]], b[idx[i]] = 0, c[idx[i]] += a[i]     (idx without repeats)
 *   SSS_ROWS_SCATTER      b[idx[i]] = a[i]                   (idx without repeats)
 *   SSS_ROWS_SEGMENT_SUM  b[s] = sum of a[i], idx[s] <= i < idx[s + 1]  (idx: n + 1 non-decreasing row offsets of the n segments;
 *                         no atomics, the additions run in row order)
 * idx entries must be valid rows of the tables (not checked). Launches on the CURRENT device's stream `stream` (no handle). */
#define SSS_ROWS_GATHER 0
#define SSS_ROWS_SCATTER_ADD 1
#define SSS_ROWS_UPDATE 2
#define SSS_ROWS_TAKE 3
#define SSS_ROWS_SCATTER 4
#define SSS_ROWS_SEGMENT_SUM 5
typedef struct sss_rows_args {
  int64_t n;              /* rows of the list */
  int64_t ld_a;           /* floats between rows of a (>= width) */
  int32_t width;          /* floats per row, 1..64 */
  int32_t op;             /* SSS_ROWS_* */
  const int64_t* idx_dev; /* i64[n] (SEGMENT_SUM: i64[n + 1]) */
  float* a_dev;
  float* b_dev;
  float* c_dev;           /* UPDATE / TAKE only */
} sss_rows_args;
int sss_rows_op(const sss_rows_args* a, void* stream);

/* Several tables side by side - the score networks' input rows torch.cat([t_0[idx_0], t_1[idx_1], ...], -1) of
 * schedulers/decima/scheduler.py:289-318 / :337-385 - in ONE launch that walks `out` as a flat array (csrc/sss_rows.h):
 *   op 0  out[i][off_k + j] = table_k[idx_k[i]][j]       for every part k (idx_k NULL: row i itself)
 *   op 1  table_k[idx_k[i]][j] += out[i][off_k + j]      (the backward pass; float atomics; parts with table_dev NULL are skipped)
 * off_k = the widths of the parts before k; out_dev f32[n][sum of widths <= 64] contiguous, table_k f32[rows_k][width_k] contiguous. */
typedef struct sss_concat_part {
  float* table_dev;
  const int64_t* idx_dev; /* i64[n], nullable */
  int32_t width;          /* floats per row of the table, 1..64 */
  int32_t pad_;
} sss_concat_part;
typedef struct sss_concat_args {
  int64_t n;       /* rows */
  int32_t n_parts; /* 1..4 */
  int32_t op;
  float* out_dev;
  sss_concat_part parts[4];
} sss_concat_args;
int sss_rows_concat(const sss_concat_args* a, void* stream);

/* Log-probability of the recorded action and entropy of a categorical distribution per SEGMENT of a flat score array - the PPO
 * update's evaluate_actions (schedulers/decima/utils.py:26-41 `evaluate`: segment softmax, torch.distributions' clamp of the
 * probabilities to [eps, 1 - eps], log, the chosen entry, -sum p log p; scheduler.py:101-139 the same over the executor counts
 * a job allows), forward (backward = 0: lg_dev, ent_dev written) and backward (backward = 1: g_scores_dev written from g_lg_dev,
 * g_ent_dev). Segment s = rows ptr[s] .. ptr[s + 1] - 1 of scores_dev; chosen[s] counts inside the segment; an empty segment
 * gives lg = ent = 0. den_eps is added to the sum of exponentials (1e-16 in decima/utils.py:35, 0 for torch.softmax). */
typedef struct sss_segcat_args {
  int64_t n_seg;
  const float* scores_dev;   /* f32[rows] */
  const int64_t* ptr_dev;    /* i64[n_seg + 1], non-decreasing */
  const int64_t* chosen_dev; /* i64[n_seg] */
  float den_eps;
  int32_t pad_;
  float* lg_dev;             /* forward out f32[n_seg] */
  float* ent_dev;            /* forward out f32[n_seg] */
  const float* g_lg_dev;     /* backward in f32[n_seg] */
  const float* g_ent_dev;    /* backward in f32[n_seg] */
  float* g_scores_dev;       /* backward out f32[rows] */
} sss_segcat_args;
int sss_segment_categorical(const sss_segcat_args* a, int backward, void* stream);

const char* sss_last_error(void);
void sss_destroy(sss_handle* h);

#ifdef __cplusplus
}
#endif
#endif
