"""The N > 1 path on CPU: two gloo ranks, each with its own shard of (emulated) envs, one
all-gather of episode summaries; the gathered table must equal a single-process run of the union
(placement invariance), row for row."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(num_executors=5, job_arrival_cap=8, job_arrival_rate=1.0e-4, moving_delay=1500.0, warmup_delay=500.0)
ENVS_PER_RANK, BASE_SEED, STEPS = 3, 500, 220


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_shard(rank: int, world: int):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from emu_util import load_emu
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.distributed import all_gather_episode_summaries, shard_seeds

    env = VecSparkSchedSimEnv(CFG, ENVS_PER_RANK, device="cpu", _lib=load_emu(), auto_reset=False)
    env.reset(seed=shard_seeds(BASE_SEED, ENVS_PER_RANK, rank))
    env.rollout("fair", STEPS)
    table = all_gather_episode_summaries(env)
    env.close()
    return table


def _worker(rank: int, world: int, port: int, out_path: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = _run_shard(rank, world)
    if rank == 0:
        torch.save(table, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_all_gather(tmp_path):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from emu_util import load_emu

    load_emu()  # build once, before forking workers
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    gathered = torch.load(out)
    assert gathered.shape == (2 * ENVS_PER_RANK, 4)

    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.distributed import episode_summaries

    env = VecSparkSchedSimEnv(CFG, 2 * ENVS_PER_RANK, device="cpu", _lib=load_emu())
    env.reset(seed=BASE_SEED)
    env.rollout("fair", STEPS)
    single = episode_summaries(env)
    env.close()
    assert torch.equal(gathered, single)
    assert (single[:, 3] == 1).all(), "every env should have finished its episode"


# ---- PPO across ranks: every rank trains on its own job sequences, gradients are averaged ---------

TRAIN = dict(trainer_cls="PPO", num_iterations=2, num_sequences=1, num_rollouts=2, seed=7, checkpointing_freq=50,
             num_epochs=2, num_batches=2, clip_range=0.2, target_kl=None, entropy_coeff=0.04, beta_discount=5.0e-3,
             opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5)
ENV = dict(num_executors=5, job_arrival_cap=5, job_arrival_rate=1.0e-4, moving_delay=1500.0, warmup_delay=500.0,
           mean_time_limit=2.0e5)


def _ppo_worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from decima_util import AGENT
    from emu_util import load_emu
    from spark_sched_sim_amd.training import Trainer

    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), ENV, dict(TRAIN, artifacts_dir=os.path.join(out_dir, f"a{rank}")),
                 device="cpu", _lib=load_emu())
    seeds = tr.collector.base_seeds.tolist()
    tr.train(verbose=False)
    torch.save({"sd": tr.policy.state_dict(), "seeds": seeds, "seed_step": tr.collector.seed_step,
                "samples": [h["samples"] for h in tr.history]}, os.path.join(out_dir, f"r{rank}.pt"))
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ppo_keeps_parameters_in_sync(tmp_path):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from emu_util import load_emu

    load_emu()
    mp.spawn(_ppo_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(str(tmp_path / "r0.pt")), torch.load(str(tmp_path / "r1.pt"))
    # rank r owns global job sequence r: disjoint base seeds, common seed step = total sequences
    assert r0["seeds"] == [7, 7] and r1["seeds"] == [8, 8] and r0["seed_step"] == r1["seed_step"] == 2
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k


def _ppo_epochs_worker(rank: int, world: int, port: int, out_dir: str, num_epochs: int):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from decima_util import AGENT
    from emu_util import load_emu
    from spark_sched_sim_amd import training

    perm_sizes = []
    real_randperm = torch.randperm

    def spy(n, *a, **kw):
        perm_sizes.append(int(n))
        return real_randperm(n, *a, **kw)

    training.torch.randperm = spy
    try:
        tr = training.Trainer(dict(AGENT, agent_cls="DecimaScheduler"), ENV,
                              dict(TRAIN, num_iterations=1, num_epochs=num_epochs, artifacts_dir=os.path.join(out_dir, f"e{num_epochs}a{rank}")),
                              device="cpu", _lib=load_emu())
        tr.train(verbose=False)
    finally:
        training.torch.randperm = real_randperm
    torch.save({"sd": tr.policy.state_dict(), "perm_sizes": perm_sizes, "samples": tr.history[0]["samples"]},
               os.path.join(out_dir, f"e{num_epochs}r{rank}.pt"))
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ppo_every_epoch_permutes_all_samples(tmp_path):
    """regression: the gradient-bucket loop of the multi-rank path once overwrote the sample count,
    so every epoch after the first permuted a single sample (and only the first epoch trained)"""
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from emu_util import load_emu

    load_emu()
    for epochs in (1, 3):
        mp.spawn(_ppo_epochs_worker, args=(2, _free_port(), str(tmp_path), epochs), nprocs=2, join=True)
    for rank in (0, 1):
        r3 = torch.load(str(tmp_path / f"e3r{rank}.pt"))
        assert len(r3["perm_sizes"]) == 3 and r3["samples"] > 2
        assert all(n == r3["samples"] for n in r3["perm_sizes"]), r3["perm_sizes"]
    one, three = torch.load(str(tmp_path / "e1r0.pt")), torch.load(str(tmp_path / "e3r0.pt"))
    assert any(not torch.equal(one["sd"][k], three["sd"][k]) for k in one["sd"]), "epochs 2 and 3 must move the parameters"
    other = torch.load(str(tmp_path / "e3r1.pt"))
    for k in three["sd"]:
        assert torch.equal(three["sd"][k], other["sd"][k]), k


def _record_grads(policy, store: list):
    """wraps `policy.update_parameters` so that the gradients every optimiser step sees are kept (after the ranks' all-reduce)"""
    real = policy.update_parameters

    def spy(loss=None):
        store.append({k: p.grad.detach().clone() for k, p in policy.named_parameters() if p.grad is not None})
        return real(loss)
    policy.update_parameters = spy


def _ppo_halves_worker(rank: int, world: int, port: int, out_dir: str):
    """rank r trains on the rollouts of job sequence r of a record made by ONE process (columns 2r, 2r + 1 of it)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from decima_util import AGENT
    from spark_sched_sim_amd.decima import DecimaPolicy
    from spark_sched_sim_amd.training import PPO, Rollouts

    blob = torch.load(os.path.join(out_dir, "record.pt"))
    full = Rollouts(**blob["ro"])
    T, B = full.active.shape
    cols = torch.arange(2 * rank, 2 * rank + 2)
    ids = (torch.arange(T)[:, None] * B + torch.arange(B)[None, :])[:, cols]
    part = Rollouts(graph=full.graph, obs_index=ids, **{k: getattr(full, k)[:, cols] for k in
                    ("active", "t_before", "t_after", "rewards", "stage_sel", "job_idx", "exec_sel", "lgprobs", "resets")})
    pol = DecimaPolicy(num_executors=ENV["num_executors"], **AGENT, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5)
    pol.load_state_dict(blob["sd"])
    grads = []
    _record_grads(pol, grads)
    pol.train()
    PPO(pol, dict(blob["train"], num_sequences=1), generator=torch.Generator().manual_seed(rank)).train_on_rollouts(part)
    torch.save({"grads": grads, "sd": pol.state_dict()}, os.path.join(out_dir, f"half{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ppo_update_equals_the_one_rank_update_on_the_union(tmp_path):
    """The reference has ONE learner: the workers' rollouts are gathered into it (trainers/trainer.py:113-121) and every minibatch's
    advantages are normalised over the whole minibatch (trainers/ppo.py:113-116). With one learner per rank each holds a part of
    every minibatch: the ranks exchange the minibatch's advantage sums and weigh their losses by their share of it, so that the
    averaged gradient is the gradient the single learner computes. Checked on a record made by one process (2 job sequences x 2
    rollouts): the gradients of the optimiser steps of (a) one process training on all of it and (b) two gloo ranks training on one
    sequence's rollouts each agree to float32 round-off (rtol 2e-4 of the largest entry: different summation order; the ranks'
    chunks of a minibatch hold the same samples as the single learner's minibatch when there is one minibatch per epoch), and
    so do the parameters after the steps whose gradients are not ~0 in any entry (Adam's first steps are sign-like)."""
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from decima_util import AGENT
    from emu_util import load_emu
    from spark_sched_sim_amd.training import PPO, Trainer

    train = dict(TRAIN, num_iterations=1, num_sequences=2, num_rollouts=2, num_epochs=2, num_batches=1)
    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), ENV, dict(train, artifacts_dir=str(tmp_path / "a")), device="cpu", _lib=load_emu())
    tr.policy.eval()
    ro = tr.collector.collect_sync(with_stats=False)
    assert int(ro.active.sum()) > 20 and ro.active.shape[1] == 4
    sd = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    fields = {k: getattr(ro, k) for k in ("graph", "active", "t_before", "t_after", "rewards", "stage_sel", "job_idx", "exec_sel", "lgprobs", "resets")}
    fields["graph"] = {k: v for k, v in ro.graph.items() if torch.is_tensor(v) or isinstance(v, (int, float))}
    torch.save({"ro": fields, "sd": sd, "train": train}, str(tmp_path / "record.pt"))
    one = []
    _record_grads(tr.policy, one)
    tr.policy.train()
    PPO(tr.policy, train, generator=torch.Generator().manual_seed(5)).train_on_rollouts(ro)
    sd_one = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    tr.close()
    mp.spawn(_ppo_halves_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    h0, h1 = torch.load(str(tmp_path / "half0.pt")), torch.load(str(tmp_path / "half1.pt"))
    assert len(one) == len(h0["grads"]) == len(h1["grads"]) == 2  # (two epochs x one minibatch)
    g1, g2 = one[0], h0["grads"][0]
    scale = max(float(v.abs().max()) for v in g1.values())  # (entries whose true gradient is 0 - the last bias of a softmax's scores - are noise)
    for k in g1:  # first optimiser step: same parameters on both sides
        assert torch.allclose(g1[k], g2[k], rtol=0.0, atol=2e-4 * scale), (k, float((g1[k] - g2[k]).abs().max()), scale)
        assert torch.equal(h0["grads"][0][k], h1["grads"][0][k]), k
    for k in sd_one:  # after both steps (the second one's gradients follow from the first one's parameters)
        assert torch.equal(h0["sd"][k], h1["sd"][k]), k
        assert torch.allclose(sd_one[k], h0["sd"][k], rtol=0.0, atol=2.5 * 2 * 3.0e-4), k  # (bounded by two sign-like Adam steps)
    close = sum(int(torch.isclose(sd_one[k], h0["sd"][k], rtol=0.0, atol=1e-5).sum()) for k in sd_one)
    total = sum(v.numel() for v in sd_one.values())
    assert close >= 0.97 * total, (close, total)


# ---- self-starting ranks (bench.py --gpus N with no outer launcher) ----------------------------------

_RANK_SCRIPT = """
import json, os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(os.environ["RANK"]) + 10.0 * float(os.environ["LOCAL_RANK"])])
out = [torch.zeros(1) for _ in range(dist.get_world_size())]
dist.all_gather(out, t)
if dist.get_rank() == 0:
    print(json.dumps({"world": dist.get_world_size(), "gathered": [float(x) for x in out], "master": os.environ["MASTER_ADDR"]}), flush=True)
else:
    print("not the result line", flush=True)
dist.destroy_process_group()
"""


def test_launch_ranks_starts_one_fresh_process_per_rank(tmp_path, capfd):
    """spark_sched_sim_amd.distributed.launch_ranks: what bench.py / tools/bench_*.py use when they are
    asked for N > 1 GPUs outside torch.distributed.run (reference: the trainer starts its own workers,
    trainers/trainer.py:264-293)"""
    import json

    sys.path[:0] = [os.path.dirname(HERE)]
    from spark_sched_sim_amd.distributed import launch_ranks

    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    assert launch_ranks(2, [str(script)]) == 0
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]  # gloo's own chatter aside
    assert len(lines) == 1, lines  # only rank 0 owns stdout
    rec = json.loads(lines[0])
    assert rec == {"world": 2, "gathered": [0.0, 11.0], "master": "127.0.0.1"}
    assert "not the result line" in err


def test_bench_starts_its_own_ranks(capfd):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment spawns two ranks; here (no GPU)
    each of them stops at the "needs an AMD GPU" check - after the rank environment was set up"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    if r.returncode == 0:
        pytest.skip("a GPU is present: the two ranks ran the bench")
    assert r.stderr.count("bench.py needs an AMD GPU") == 2, r.stderr[-2000:]


_FAILING_RANK_SCRIPT = """
import os, sys, time
if os.environ["RANK"] == "1":
    sys.exit(7)            # dies at start-up (bad device, import error, ...)
time.sleep(600)            # its sibling would wait in init_process_group / a collective
"""

_HANGING_RANK_SCRIPT = """
import time
time.sleep(600)
"""


def test_launch_ranks_stops_the_other_ranks_when_one_fails(tmp_path):
    """a rank that exits non-zero takes the launch down with its own exit code; the siblings are terminated
    instead of waiting for the backend's collective timeout"""
    import time

    sys.path[:0] = [os.path.dirname(HERE)]
    from spark_sched_sim_amd.distributed import launch_ranks

    script = tmp_path / "rank_fail.py"
    script.write_text(_FAILING_RANK_SCRIPT)
    t0 = time.monotonic()
    assert launch_ranks(2, [str(script)], grace=2.0) == 7
    assert time.monotonic() - t0 < 60


def test_launch_ranks_timeout(tmp_path):
    import time

    sys.path[:0] = [os.path.dirname(HERE)]
    from spark_sched_sim_amd.distributed import launch_ranks

    script = tmp_path / "rank_hang.py"
    script.write_text(_HANGING_RANK_SCRIPT)
    t0 = time.monotonic()
    assert launch_ranks(2, [str(script)], timeout=1.0, grace=2.0) == 124
    assert time.monotonic() - t0 < 60
