"""Cross-rank reductions of the data-parallel trainer (pure torch: run on RCCL in production and on
gloo in the CPU tests).  Environments are sharded across GPUs with no data-path collective; only
these scalars and the flat gradient cross ranks."""
from __future__ import annotations

import math
import os

import torch

# LHW_FORCE_DIST=1 (tests): issue the collectives even with a single rank, so that the RCCL calls execute on a 1-GPU box
_FORCE = os.environ.get("LHW_FORCE_DIST") == "1"


def _active(d) -> bool:
    return bool(d) and (d.get_world_size() > 1 or _FORCE)


def dist():
    import torch.distributed as d
    return d if d.is_available() and d.is_initialized() else None


def global_mean_std(sum_x: torch.Tensor, sum_x2: torch.Tensor, n: float):
    """Mean and UNBIASED std of the union of all ranks' samples from per-rank (sum, sum of squares, count).
    Matches torch's ``x.mean()`` / ``x.std()`` on the concatenated batch (reference rl/algos/ppo.py:485)."""
    pack = torch.stack([sum_x.double().reshape(()), sum_x2.double().reshape(()),
                        torch.tensor(float(n), dtype=torch.float64, device=sum_x.device)])
    d = dist()
    if _active(d):
        d.all_reduce(pack)
    s, s2, cnt = (float(v) for v in pack)
    mean = s / cnt
    var = max(0.0, (s2 - cnt * mean * mean) / max(1.0, cnt - 1.0))
    return mean, math.sqrt(var), cnt


def global_moments_pack(mom2: torch.Tensor, n: float) -> torch.Tensor:
    """Device-resident {sum, sum of squares, count} of the union of all ranks' samples: this rank's two moments (float64, on the
    device) with the count appended, sum-all-reduced -- the input of lhw_standardize.  No value visits the host."""
    pack = torch.empty(3, dtype=torch.float64, device=mom2.device)
    pack[:2] = mom2.reshape(2)
    pack[2] = float(n)
    d = dist()
    if _active(d):
        d.all_reduce(pack)
    return pack


def global_batch_moments(x: torch.Tensor):
    """Per-feature mean / biased variance / count of the union of all ranks' rows (RunningMeanStd.update
    on the concatenated batch, reference rl/envs/normalize.py:16-33)."""
    x = x.double()
    n = torch.tensor([float(x.shape[0])], dtype=torch.float64, device=x.device)
    pack = torch.cat([x.sum(0), (x * x).sum(0), n])
    d = dist()
    if _active(d):
        d.all_reduce(pack)
    D = x.shape[1]
    cnt = pack[-1]
    mean = pack[:D] / cnt
    var = (pack[D:2 * D] / cnt - mean * mean).clamp_min(0.0)
    return mean, var, float(cnt)


# bench.py: when a list, every gradient all-reduce is bracketed by two events on the current stream and appended here
allreduce_events = None


def allreduce_grad_(flat_grad: torch.Tensor) -> float:
    """Sum-all-reduce the flat gradient in place; returns the scale (1/world) that turns the sum of per-rank
    mean-gradients into the mean over the global minibatch."""
    d = dist()
    if _active(d):
        if allreduce_events is not None and flat_grad.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            d.all_reduce(flat_grad)
            e1.record()
            allreduce_events.append((e0, e1))
        else:
            d.all_reduce(flat_grad)
        return 1.0 / d.get_world_size()
    return 1.0


def relaunch_under_torchrun(n_procs: int, script: str, argv: list) -> int:
    """Fan-out is the entry point's job (the reference's PPO starts its own Ray workers, rl/algos/ppo.py:215-297): a plain
    `python <script> --gpus N` with N > 1 and no launcher environment re-executes itself as N ranks on this node under
    torch.distributed.run (one process per GPU) and returns the launcher's exit code.  `--standalone`: the launcher itself opens the
    rendezvous store on a free port of 127.0.0.1 and hands the ranks MASTER_ADDR / MASTER_PORT (no bind-then-close port guess here)."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={int(n_procs)}", script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def launched() -> bool:
    """True inside a torch.distributed.run / torchrun worker (RANK and WORLD_SIZE are set by the launcher)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def shard_env_ids(n_envs_per_rank: int, rank: int) -> int:
    """Global index of this rank's first environment (RNG keys use global env ids)."""
    return rank * n_envs_per_rank


def global_episode_stats(ret_sum: float, len_sum: float, count: float, device=None):
    """Finished-episode statistics summed over all ranks: the reference averages the episodes of ALL workers for the
    `Mean Eprew / Mean Eplen` table and for the evaluation reward that drives save-if-best (rl/algos/ppo.py:408-426,
    :540-548), so every rank logs / checkpoints on the same numbers."""
    d = dist()
    if not _active(d):
        return ret_sum, len_sum, count
    pack = torch.tensor([float(ret_sum), float(len_sum), float(count)], dtype=torch.float64, device=device)
    d.all_reduce(pack)
    r, l, c = (float(v) for v in pack)
    return r, l, c
