"""debug: who is the slowest env of a step launch? per-env header deltas across single launches (step mode)"""
import sys, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np, torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.vec_env import HDR_OFF, HDR_PROF
CFG = {"c2": (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash"),
       "c3": (dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair")}
name = sys.argv[1]; cfg, pol = CFG[name]
env = VecSparkSchedSimEnv(cfg, 4096, device="cuda:0", pack=workload.default_pack(), auto_reset=True)
env.reset(seed=0); env.rollout(pol, 700); torch.cuda.synchronize()
def snap():
    h = env._env_view[:, :env.dims.hdr_bytes].cpu().numpy()
    f = lambda off, dt=np.uint64: np.ascontiguousarray(h[:, off:off + 8]).view(dt).ravel().astype(np.int64)
    prof = np.ascontiguousarray(h[:, HDR_PROF:HDR_PROF + 40]).view(np.uint64).astype(np.int64)
    ep = np.ascontiguousarray(h[:, HDR_OFF["episodes"]:HDR_OFF["episodes"] + 4]).view(np.int32).ravel()
    return dict(pad=f(272), pad1=f(280), ev=f(HDR_OFF["n_events"]), fast=f(HDR_OFF["n_fast"]), bat=f(HDR_OFF["n_batched"]), rounds=f(HDR_OFF["n_rounds"]), prof=prof, ep=ep.astype(np.int64))
rows = []
for it in range(60):
    a = snap()
    act = env.policy_actions(pol)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step_async(act["stage_idx"], act["num_exec"]); e1.record(); torch.cuda.synchronize()
    b = snap()
    tot = (b["prof"] - a["prof"])[:, 1:].sum(1)  # action + events + reward + observe ticks (slow is inside events)
    k = int(np.argmax(tot))
    d = {x: int(b[x][k] - a[x][k]) for x in ("ev", "fast", "bat", "rounds", "ep")}
    pd, p1 = int(b["pad"][k] - a["pad"][k]), int(b["pad1"][k] - a["pad1"][k])  # -DSSS_TAILSTAT builds (tailstat.sh): one-at-a-time events by kind
    d["serial"] = dict(arrival=pd & 0xFFFF, exec_ready=(pd >> 16) & 0xFFFF, completes_stage=(pd >> 32) & 0xFFFF, other_task=(pd >> 48) & 0xFFFF)
    d["er_why"] = dict(no_slot=p1 & 0xFFFF, no_tasks=(p1 >> 16) & 0xFFFF, source=(p1 >> 32) & 0xFFFF, other=(p1 >> 48) & 0xFFFF)
    p = (b["prof"] - a["prof"])[k]
    rows.append((e0.elapsed_time(e1), int(tot[k]), d, p.tolist(), float(np.mean(tot)), int(np.sort(tot)[-40])))
for ms, t, d, p, mean, p99 in rows[:25]:
    print(f"launch {ms:.3f} ms | slowest env: {t:8d} ticks (mean {mean:8.0f}, p99 {p99:8d}) events {d['ev']:4d} fast {d['fast']:4d} batched {d['bat']:4d} rounds {d['rounds']:3d} slow {d['ev']-d['fast']:3d} reset {d['ep']} serial {d['serial']} er_why {d['er_why']} | slow_ev/action/events/reward/observe {p}")
ms = np.array([r[0] for r in rows]); tk = np.array([r[1] for r in rows])
print("mean launch ms", ms.mean(), "mean slowest-env ticks", tk.mean(), "=> ticks/ms", tk.mean() / ms.mean())
