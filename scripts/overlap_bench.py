"""Experiment: does splitting the 4096-env batch into independent groups on separate HIP streams (no global barrier per
control step) recover the throughput that 16384-env batches show?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
spec = JvrcWalkSpec()
N, T = 4096, 60


def run(groups):
    n = N // groups
    envs = [spec.make_batched(n, seed=1, device=0, max_traj_len=400, env_id_base=g * n) for g in range(groups)]
    streams = [torch.cuda.Stream() for _ in range(groups)]
    acts = [torch.randn(n, 12, device="cuda") * 0.2 for _ in range(groups)]
    for e in envs:
        e.reset()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.time()
        for t in range(T):
            for e, s, a in zip(envs, streams, acts):
                with torch.cuda.stream(s):
                    e.step(a)
                    a.mul_(1.0)          # stand-in for the dependent policy forward of this group
        torch.cuda.synchronize()
        dt = time.time() - t0
    for e in envs:
        e.close()
    return dt / T * 1e3


for g in (1, 2, 4, 8):
    print(f"groups {g}: {run(g):.3f} ms per control step of {N} envs")
