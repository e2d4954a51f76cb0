"""Train a jvrc_walk actor for a few dozen PPO iterations on the GPU and store its weights as a test fixture
(tests/golden/jvrc_walk_actor_trained.npz): tests/test_freerun_gpu.py drives the HIP stepper and the CPU oracle with this
policy's mean action for 1000 free-running control steps.  Run on the GPU box: `python scripts/make_policy_fixture.py [iters]`."""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from learninghumanoidwalking_amd import envs as lenvs
from learninghumanoidwalking_amd.ppo import PPO

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "jvrc_walk_actor_trained.npz")
args = SimpleNamespace(
    gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=32768, epochs=3, max_traj_len=400,
    num_procs=4096, num_envs=4096, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=None,
    imitate_coeff=0.3, learn_std=False, std_dev=0.223, no_mirror=False, infer_fp16=False, continued=None,
    logdir=os.path.join("/tmp", f"lhw_fixture_{os.getpid()}"), device_index=0)
algo = PPO(lenvs.ENVIRONMENTS["jvrc_walk"], args, seed=0)
t0 = time.time()
for i in range(iters):
    algo.iterate(i)
    r, l, c = algo._ep_stats
    if i % 5 == 0 or i == iters - 1:
        print(f"iter {i:3d}  mean episode return {r / max(c, 1):8.2f}  mean length {l / max(c, 1):6.1f}  ({time.time() - t0:.0f} s)", flush=True)
t = algo.kernels.get_tensors()
np.savez_compressed(out, obs_mean=algo.kernels.obs_mean.cpu().numpy(), obs_std=algo.kernels.obs_std.cpu().numpy(),
                    **{k: v.numpy() for k, v in t.items() if k.startswith("a_")})
print("wrote", out, os.path.getsize(out), "bytes")
