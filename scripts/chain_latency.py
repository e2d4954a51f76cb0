"""Latency of one group's policy-inference chain (normalize + weight transposes + strip forward + sample) while the OTHER group's
control-step kernel occupies half of the GPU, against the same chain on an idle GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from learninghumanoidwalking_amd import envs as lenvs
from learninghumanoidwalking_amd.ppo import PPO
args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=32768, epochs=3, max_traj_len=400,
                       num_procs=4096, num_envs=4096, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=None,
                       imitate_coeff=0.3, learn_std=False, std_dev=0.223, no_mirror=False, infer_fp16=False, continued=None,
                       logdir="/tmp/lhw_cl", device_index=0)
algo = PPO(lenvs.ENVIRONMENTS["jvrc_walk"], args, seed=0)
env, k = algo.env, algo.kernels
N = 4096
obs = env.reset().clone()
act = torch.zeros(N, 12, device="cuda"); logp = torch.zeros(N, device="cuda")
obs2 = torch.zeros_like(obs); tob = torch.zeros_like(obs); rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def chain(a, b):
    k.forward(obs[a:b], seed=0, env_id_base=a, counter=0, deterministic=False, want_value=False, want_mu=False, ws_row=a, act=act[a:b], logp=logp[a:b])
def timed(busy):
    ts = []
    for rep in range(30):
        torch.cuda.synchronize()
        if busy:
            with torch.cuda.stream(sb):
                env.step_range(2048, 2048, act, obs2, tob, rew, done)
            time.sleep(0.0003)      # the other group's kernel is in flight
        with torch.cuda.stream(sa):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); chain(0, 2048); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts), np.min(ts), np.max(ts)
for busy in (0, 1, 0, 1):
    m, lo, hi = timed(busy)
    print(f"inference chain of 2048 envs, other group's step kernel {'RUNNING' if busy else 'idle   '}: median {m:7.1f} us  (min {lo:.1f}, max {hi:.1f})")
# the env step of one group alone vs with the other group's running
def step_timed(busy):
    ts = []
    for rep in range(20):
        torch.cuda.synchronize()
        if busy:
            with torch.cuda.stream(sb):
                env.step_range(2048, 2048, act, obs2, tob, rew, done)
        with torch.cuda.stream(sa):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_range(0, 2048, act, obs2, tob, rew, done); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts)
print(f"control step of 2048 envs alone: {step_timed(0):.0f} us;  beside the other group's: {step_timed(1):.0f} us")

# breakdown of the whole-batch chain on an idle GPU: repeated launches of each piece alone (back to back, so launch gaps included)
import ctypes
from learninghumanoidwalking_amd import _lib
def rep(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rows in (2048, 4096):
    print(f"rows {rows}: whole chain {rep(lambda: chain(0, rows)):.1f} us per call (back-to-back calls)")
