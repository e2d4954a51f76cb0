"""Diagnostic: resident rollout vs launch-per-step rollout on the GPU, first differences per buffer.  usage: resident_diff.py ENV [N] [T] [ROLLOUTS]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
from learninghumanoidwalking_amd.ppo import PPO

env_name = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 97
T = int(sys.argv[3]) if len(sys.argv) > 3 else 12
R = int(sys.argv[4]) if len(sys.argv) > 4 else 3
NAMES = ("obs", "act", "logp", "tob", "rew", "done")


def run(mode):
    os.environ["LHW_ROLLOUT_MODE"] = mode
    args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=1,
                           max_traj_len=T, num_procs=N, num_envs=N, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                           recurrent=False, imitate=None, learn_std=False, std_dev=0.4, no_mirror=True, continued=None,
                           logdir="/tmp/lhw_diag", device_index=0)
    algo = PPO(ENVIRONMENTS[env_name], args, seed=9)
    out = []
    for _ in range(R):
        algo.sample_parallel_with_workers()
        ro = algo.rollout
        out.append([x.cpu().numpy().copy() for x in (ro.obs, ro.act, ro.logp, ro.tob_all, ro.rew, ro.done)])
    return out


a, b = run("steps"), run("resident")
for r in range(R):
    done = a[r][5]
    for n, x, y in zip(NAMES, a[r], b[r]):
        d = x != y
        if d.any():
            idx = np.argwhere(d)
            t0, e0 = idx[0][0], idx[0][1]
            prior_done = bool((done[:t0, e0] != 0).any()) or r > 0
            print(f"rollout {r} {n}: {int(d.sum())} differ, first at t={t0} env={e0} ({x[tuple(idx[0])]} vs {y[tuple(idx[0])]}), envs {sorted(set(idx[:, 1].tolist()))[:12]}, "
                  f"an episode of that env ended before: {prior_done}; max |diff| {np.nanmax(np.abs(x.astype(np.float64) - y.astype(np.float64)))}")
            if n == "obs":
                e = idx[0][1]
                cols = sorted(set(idx[(idx[:, 0] == t0) & (idx[:, 1] == e)][:, 2].tolist()))
                print("   columns", cols)
                print("   steps   ", x[t0, e, cols])
                print("   resident", y[t0, e, cols])
                print("   full row steps   ", np.array2string(x[t0, e], precision=4, max_line_width=250))
        else:
            print(f"rollout {r} {n}: equal")
