"""Worker of tests/test_distributed_gpu.py: PPO iterations of jvrc_walk, either as one rank of a data-parallel job
(`--mode ranks`, N envs per rank, launched with torch.distributed.run) or as ONE process holding the union of the ranks' envs
(`--mode union`, world x N envs) with the ranks' minibatches merged -- the single-process semantics the data-parallel run has
to reproduce (reference rl/algos/ppo.py:393-394 gradient clipping on the whole minibatch, :484-485 advantage statistics of the
whole batch).

Every process writes `<out>.rank<r>.npz` (the union: rank 0) with, per iteration i: the rollout (`obs_i act_i logp_i rew_i done_i`,
time-major), `ret_i` and the normalised `adv_i`, the first minibatch's gradient before and after the all-reduce (`g0_i`, `g1_i`,
already multiplied by 1 / world) and the weights after the iteration (`theta_i`) -- so that a mismatch can be localised to the
stage where it first appears instead of being read off the final weights."""
import argparse
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["ranks", "union"], required=True)
ap.add_argument("--world", type=int, default=2)
ap.add_argument("--envs", type=int, default=64)
ap.add_argument("--traj", type=int, default=32)
ap.add_argument("--mb", type=int, default=512)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--backend", default="gloo")
ap.add_argument("--out", required=True)
a = ap.parse_args()

torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", 1))
if a.mode == "ranks":
    assert world == a.world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend)      # both ranks share GPU 0 on the 1-GPU test box (LHW_SHARE_GPU semantics)
from learninghumanoidwalking_amd import dist_utils
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
from learninghumanoidwalking_amd.ppo import PPO

union = a.mode == "union"
if union:
    # the union's gradient is dumped where the ranks' is: at the (here no-op) all-reduce between lhw_ppo_grad and lhw_ppo_apply.  The
    # one-launch optimiser step of a single process (lhw_ppo_step) has no such seam; it is bitwise this path (tests/test_iteration_gpu.py)
    os.environ["LHW_PPO_GRAPH"] = "0"
N = a.envs * (a.world if union else 1)
args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=a.mb * (a.world if union else 1),
                       epochs=2, max_traj_len=a.traj, num_procs=N, num_envs=N, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                       recurrent=False, imitate=None, imitate_coeff=0.3, learn_std=False, std_dev=0.223, no_mirror=False, infer_fp16=False,
                       continued=None, logdir=os.path.join("/tmp", f"lhw_dp_{os.getpid()}"), device_index=0)
algo = PPO(ENVIRONMENTS["jvrc_walk"], args, seed=5)
if union:
    # minibatch k of the union = minibatch k of every rank, expressed in the union's [T][world * n] sample numbering
    n, T, W, mb = a.envs, a.traj, a.world, a.mb
    local = PPO._minibatch_perm

    def union_perm(self, itr, epoch, n_samples, rank=None):
        parts = []
        for r in range(W):
            p = local(self, itr, epoch, n * T, rank=r).long()
            t, e = p // n, p % n
            parts.append((t * (W * n) + r * n + e).reshape(-1, 1)[: (n * T // mb) * mb].reshape(-1, mb))
        return torch.cat(parts, dim=1).reshape(-1).to(torch.int32)

    PPO._minibatch_perm = union_perm

rank = dist.get_rank() if dist.is_initialized() else 0
dump = {}
state = {"itr": 0, "first": True}
orig_allreduce = dist_utils.allreduce_grad_


def traced_allreduce(flat_grad):
    if state["first"]:
        dump[f"g0_{state['itr']}"] = flat_grad.detach().cpu().numpy().copy()
    scale = orig_allreduce(flat_grad)
    if state["first"]:
        dump[f"g1_{state['itr']}"] = flat_grad.detach().cpu().numpy() * np.float32(scale)
        state["first"] = False
    return scale


dist_utils.allreduce_grad_ = traced_allreduce
orig_optimize = algo.optimize


def traced_optimize(itr):
    ro = algo.rollout
    T = ro.T
    torch.cuda.synchronize()
    for name, t in (("obs", ro.obs[:T]), ("act", ro.act), ("logp", ro.logp), ("rew", ro.rew), ("done", ro.done), ("ret", algo._ret)):
        dump[f"{name}_{itr}"] = t.detach().cpu().numpy().copy()
    state["itr"], state["first"] = itr, True
    out = orig_optimize(itr)
    torch.cuda.synchronize()
    dump[f"adv_{itr}"] = algo._adv.detach().cpu().numpy().copy()      # normalised in place by optimize()
    dump[f"theta_{itr}"] = algo.kernels.theta.detach().cpu().numpy().copy()
    return out


algo.optimize = traced_optimize
for i in range(a.iters):
    algo.iterate(i)
faults = algo.env.pop_fault_stats() if hasattr(algo.env, "pop_fault_stats") else (0, 0)
dump["faults"] = np.asarray(faults, np.int64)          # (contact overflows, diverged envs): both must be 0
np.savez(f"{a.out}.rank{rank}.npz", **dump)
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
