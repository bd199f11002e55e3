"""debug: device memory over several PPO iterations at the BASELINE config-5 share (arena reuse, allocator rounding)"""
import sys, json, time, torch
import os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, osp.join(ROOT, "tools"))
from bench_ppo import AGENT
from spark_sched_sim_amd.training import Trainer
train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=256, num_rollouts=4, seed=42, checkpointing_freq=10 ** 9, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01,
             entropy_coeff=0.04, beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir="/tmp/sss_ppo")
env = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
tr = Trainer(AGENT, env, train, device="cuda:0")
for it in range(7):
    t0 = time.perf_counter()
    tr.policy.eval(); ro = tr.collector.collect_sync(with_stats=False); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.policy.train(); tr.ppo.train_on_rollouts(ro); torch.cuda.synchronize(); t2 = time.perf_counter()
    del ro
    print(json.dumps({"it": it, "collect_s": round(t1 - t0, 2), "train_s": round(t2 - t1, 2), "allocated_GB": round(torch.cuda.memory_allocated() / 2**30, 1),
                      "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 1), "max_allocated_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
