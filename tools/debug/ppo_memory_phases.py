import sys, json, time, torch
import os.path as osp
ROOT = "/root/repo" if osp.exists("/root/repo/tools") else "."
sys.path.insert(0, ROOT); sys.path.insert(0, osp.join(ROOT, "tools"))
from bench_ppo import AGENT
from spark_sched_sim_amd.training import Trainer
train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=256, num_rollouts=4, seed=42, checkpointing_freq=10 ** 9, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01,
             entropy_coeff=0.04, beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir="/tmp/sss_ppo")
env = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
tr = Trainer(AGENT, env, train, device="cuda:0")
G = 1e9
for it in range(3):
    torch.cuda.reset_peak_memory_stats()
    tr.policy.eval(); ro = tr.collector.collect_sync(with_stats=False); torch.cuda.synchronize()
    a1, p1 = torch.cuda.memory_allocated() / G, torch.cuda.max_memory_allocated() / G
    rec_bytes = sum(v.numel() * v.element_size() for v in ro.graph.values() if isinstance(v, torch.Tensor)) / G
    torch.cuda.reset_peak_memory_stats()
    tr.policy.train(); tr.ppo.train_on_rollouts(ro); torch.cuda.synchronize()
    a2, p2 = torch.cuda.memory_allocated() / G, torch.cuda.max_memory_allocated() / G
    del ro
    print(json.dumps({"it": it, "after_collect_GB": round(a1, 1), "peak_during_collect_GB": round(p1, 1), "record_graph_GB": round(rec_bytes, 1), "after_train_GB": round(a2, 1),
                      "peak_during_train_GB": round(p2, 1), "reserved_GB": round(torch.cuda.memory_reserved() / G, 1)}), flush=True)
