"""Spread of the wavefronts' total times over a resident rollout (lhw_env_rollout): a wave that finishes early leaves its slot idle
until the launch ends.  usage: rollout_wave_spread.py ENV [N] [ITERS]   (PPO iterations first, so that the policy is not the initial one)"""
import os, sys
os.environ.setdefault("LHW_ROLLOUT_CHUNK", "0")   # one wave per env group for the whole rollout: the job queue of the stepping task would hide the spread it exists for
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np, torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
from learninghumanoidwalking_amd.ppo import PPO
name = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=32768, epochs=3, max_traj_len=400, num_procs=N, num_envs=N,
                       max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=None, learn_std=False, std_dev=0.223, no_mirror=False,
                       continued=None, logdir="/tmp/lhw_spread", device_index=0)
algo = PPO(ENVIRONMENTS[name], args, seed=0)
for i in range(iters):
    algo.iterate(i)
env = algo.env
env.wave_cycles()      # arms the recording
algo.sample_parallel_with_workers()
torch.cuda.synchronize()
c = env.wave_cycles().astype(float)
per_wave = 1 if name == "jvrc_step" else 2
w = c.reshape(-1, per_wave).max(1)
print(f"{name} @ {N}, rollout mode {algo.rollout.last_mode}: waves {len(w)}  mean {w.mean():.4e} ticks  min/mean {w.min() / w.mean():.3f}  p10 {np.percentile(w, 10) / w.mean():.3f}  "
      f"p50 {np.percentile(w, 50) / w.mean():.3f}  p90 {np.percentile(w, 90) / w.mean():.3f}  p99 {np.percentile(w, 99) / w.mean():.3f}  max {w.max() / w.mean():.3f}")
print(f"   wave slots occupied on average {w.mean() / w.max():.3f} of the launch (mean / max)")
