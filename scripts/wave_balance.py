"""Load balance of the control-step kernel across wavefronts in the training regime: runs a few PPO iterations (untrained
policy) and prints the distribution of per-group launch durations of the last control step of each iteration."""
import os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from learninghumanoidwalking_amd import envs as lenvs
from learninghumanoidwalking_amd.ppo import PPO
args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=32768, epochs=3, max_traj_len=400,
                       num_procs=4096, num_envs=4096, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=None,
                       imitate_coeff=0.3, learn_std=False, std_dev=0.223, no_mirror=False, infer_fp16=False, continued=None,
                       logdir="/tmp/lhw_wb", device_index=0)
algo = PPO(lenvs.ENVIRONMENTS["jvrc_walk"], args, seed=0)
algo.env.wave_cycles()          # arm
for i in range(3):
    algo.iterate(i)
    c = algo.env.wave_cycles().astype(float)
    w = c.reshape(-1, 2).max(1)          # a wave = two envs
    print(f"iter {i}: per-wave cycles of the last control step: mean {w.mean():.3e}  p50 {np.percentile(w,50):.3e}  p90 {np.percentile(w,90):.3e}  "
          f"p99 {np.percentile(w,99):.3e}  max {w.max():.3e}   max/mean {w.max()/w.mean():.2f}   (= {w.max()/2.4e6:.2f} ms at 2.4 GHz)")
