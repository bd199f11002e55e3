"""-m gpu: the workload boundary and the second trace regime on the HIP path.

* a stage with 40 000 tasks (beyond a 16-bit counter: round 5's silent wrap) step by step against the C oracle;
* BASELINE-size batches (4096 envs) on the "deep" trace regime (workload.PROFILES["deep"]: <= 40 stages, in-degree <= 6 over all
  predecessors, <= 3000 tasks per stage, a 60 MB pack) run to the end of their episodes, every env's episode summary against the oracle
  (the reference-recorded `deep_*` goldens are replayed in tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from boundary_util import SMALL_QUERIES, SMALL_SIZES, lockstep_vs_oracle
from golden_util import bits
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from test_emu_boundary import CFG, _big_stage_raw
from test_gpu_fullsize_oracle import oracle_episodes

pytestmark = pytest.mark.gpu


def test_a_stage_with_40000_tasks_is_exact_on_the_gpu():
    big = workload.build_pack(_big_stage_raw(), query_sizes=SMALL_SIZES, num_queries=SMALL_QUERIES)
    env = VecSparkSchedSimEnv(CFG, 64, device="cuda:0", pack=big)
    env.reset(seed=500)
    nodes = env.nodes.cpu().numpy()
    assert nodes[..., 0].max() == 40000.0 and nodes[..., 0].min() >= 0.0 and int(env.obs_i32[:, 7].abs().sum()) == 0
    env.close()
    bad = lockstep_vs_oracle(big, CFG, list(range(500, 532)), 400, device="cuda:0")
    assert not bad, "\n".join(bad[:8])


def test_a_2_to_the_20_task_stage_whole_episodes_match_the_oracle():
    """task counts far beyond 16 bits, whole episodes: 1 048 576 tasks in every job's first stage (32 envs, fair policy)"""
    big = workload.build_pack(_big_stage_raw(1 << 20), query_sizes=SMALL_SIZES, num_queries=SMALL_QUERIES)
    cfg = dict(CFG, job_arrival_cap=3)
    B, base = 32, 900
    env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=big)
    env.reset(seed=base)
    for _ in range(200):
        env.rollout("fair", 200)
        if bool((env.header_field("terminated") != 0).all()):
            break
    torch.cuda.synchronize()
    assert bool((env.header_field("terminated") != 0).all()) and int((env.obs_i32[:, 7] != 0).sum()) == 0
    got = [env.header_field(k).cpu().numpy() for k in ("last_ep_steps", "last_ep_return", "last_ep_wall", "J")]
    exp = oracle_episodes(big, cfg, 0, [base + i for i in range(B)])
    bad = [i for i in range(B) if (int(got[0][i]), bits(got[1][i]), bits(got[2][i]), int(got[3][i])) != (exp[i][0], bits(exp[i][1]), bits(exp[i][2]), exp[i][3])]
    assert not bad, (bad[:4], exp[bad[0]])
    env.close()


DEEP_C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
DEEP_C3 = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)


@pytest.mark.parametrize("cfg,policy,policy_id,B,max_steps", [(DEEP_C2, "hash", 1, 4096, 6000), (DEEP_C3, "fair", 0, 1024, 30000)], ids=["c2_hash_4096", "c3_fair_1024"])
def test_deep_regime_full_batch_episode_summaries_match_oracle(cfg, policy, policy_id, B, max_steps):
    pack = workload.profile_pack("deep")
    env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=pack)
    base = 47000
    env.reset(seed=base)
    done = 0
    while done < max_steps:
        env.rollout(policy, 200)
        done += 200
        if bool(((env.header_field("terminated") != 0) | (env.obs_i32[:, 7] == 5)).all()):
            break
    torch.cuda.synchronize()
    err = env.obs_i32[:, 7].cpu().numpy()
    term = env.header_field("terminated").cpu().numpy()
    assert ((term != 0) | (err == 5)).all(), ("episodes did not finish", np.unique(err))   # 5 = the reference's own [step] stall (random policy)
    steps = env.header_field("ep_steps").cpu().numpy()
    ret = env.header_field("ep_return").cpu().numpy()
    wall = env.header_field("wall_time").cpu().numpy()
    J = env.header_field("J").cpu().numpy()
    exp = oracle_episodes(pack, cfg, policy_id, [base + i for i in range(B)])
    bad = []
    for i in range(B):
        if err[i] == 5:
            if exp[i][0] != -105:
                bad.append(i)
        elif (int(steps[i]), bits(ret[i]), bits(wall[i]), int(J[i])) != (exp[i][0], bits(exp[i][1]), bits(exp[i][2]), exp[i][3]):
            bad.append(i)
    assert not bad, f"{len(bad)} of {B} envs differ from the oracle, first: env {bad[0]} got {(int(steps[bad[0]]), ret[bad[0]], wall[bad[0]])} expected {exp[bad[0]]}"
    env.close()
