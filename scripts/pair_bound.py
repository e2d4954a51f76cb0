"""Upper bound of what pairing similar envs in a wave could buy (two envs per wave: every data-dependent loop runs for the slower
of the two).  A: control-step time with independent envs; B: the same with env 2i+1 made a copy of env 2i (same state, same
actions: the two envs of every wave do the same work).  python scripts/pair_bound.py [N] [env]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "jvrc_walk"
env = ENVIRONMENTS[name]().make_batched(N, seed=1, device=0, max_traj_len=100000)
env.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
A = env.act_dim
def draw(paired):
    a = torch.randn(N, A, device="cuda", generator=gen) * 0.223
    if paired: a[1::2] = a[0::2]
    return a.contiguous()
def timed(paired, steps=40):
    acts = [draw(paired) for _ in range(steps)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for a in acts: env.step(a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
for _ in range(40): env.step(draw(False))
for rep in range(3):
    tA = timed(False)
    q, v = env.get_state()
    q[1::2] = q[0::2]; v[1::2] = v[0::2]
    env.set_state(q, v)
    for _ in range(3): env.step(draw(True))      # (warm starts and action filters of the copies settle)
    tB = timed(True)
    q2, v2 = env.get_state()
    same = float(np.mean(np.all(np.abs(q2[1::2] - q2[0::2]) < 1e-9, axis=1)))
    print(f"{name} N={N}: independent {tA:.4f} ms/step, waves of two identical envs {tB:.4f} ms/step ({100 * (tB / tA - 1):+.1f} %); pairs still identical after the run: {100 * same:.0f} %")
    for _ in range(20): env.step(draw(False))    # diverge again
