"""Where the sampling phase of an iteration goes besides the rollout kernel (jvrc_walk @ 4096 by default): wall time of each part of
PPO.sample_parallel_with_workers with a device synchronisation after it.  usage: sample_phase_timing.py [env] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
from learninghumanoidwalking_amd import ppo as ppo_mod
from learninghumanoidwalking_amd.ppo import PPO
name = sys.argv[1] if len(sys.argv) > 1 else "jvrc_walk"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=32768, epochs=3, max_traj_len=400, num_procs=N, num_envs=N,
                       max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=None, learn_std=False, std_dev=0.223, no_mirror=False,
                       continued=None, logdir="/tmp/lhw_spt", device_index=0)
algo = PPO(ENVIRONMENTS[name], args, seed=0)
for i in range(3):
    algo.iterate(i)
acc = {}
def timed(label, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        return r
    return w
ro = algo.rollout
ro._collect_resident = timed("rollout launch (lhw_env_rollout)", ro._collect_resident)
ro._batched_values = timed("critic values of the stored states / truncated terminal observations", ro._batched_values)
ro.collect = timed("Rollout.collect (all of the above + final value)", ro.collect)
algo.kernels.gae = timed("gae", algo.kernels.gae)
ro.pop_episode_stats = timed("episode statistics read-back", ro.pop_episode_stats)
algo._traj_idx = timed("traj_idx", algo._traj_idx)
K = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    algo.sample_parallel_with_workers()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / K
print(f"{name} @ {N}: sample_parallel_with_workers {tot * 1e3:.2f} ms per call (with the synchronisations of this script)")
for k, v in acc.items():
    print(f"  {k:75s} {v / K * 1e3:8.2f} ms")
