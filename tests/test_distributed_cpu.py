"""world_size-2 gloo tests of the cross-rank host logic (the N>1 path of bench.py / PPO.train).
The kernels themselves need a GPU; what crosses ranks is plain torch and is checked here against the
single-process result on the concatenated data."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from learninghumanoidwalking_amd import dist_utils
    rs = np.random.default_rng(100 + rank)
    x = torch.tensor(rs.normal(size=1000 + 37 * rank) * (1 + rank) + rank)
    mean, std, cnt = dist_utils.global_mean_std(x.sum(), (x * x).sum(), x.numel())
    pack = dist_utils.global_moments_pack(torch.stack([x.sum(), (x * x).sum()]), x.numel()).numpy()   # device-resident form (lhw_standardize)
    obs = torch.tensor(rs.normal(size=(50 + 10 * rank, 5)) + rank)
    m, v, n = dist_utils.global_batch_moments(obs)
    g = torch.full((7,), float(rank + 1))
    scale = dist_utils.allreduce_grad_(g)
    eps = dist_utils.global_episode_stats(10.0 * (rank + 1), 100.0 + rank, 3 + rank)     # every rank reports all ranks' episodes
    q.put((rank, mean, std, cnt, m.numpy(), v.numpy(), n, (g * scale).numpy(), dist_utils.shard_env_ids(4096, rank), eps, pack))
    dist.destroy_process_group()


def test_two_rank_reductions_match_concatenated_batch():
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xs = [torch.tensor(np.random.default_rng(100 + r).normal(size=1000 + 37 * r) * (1 + r) + r) for r in range(world)]
    obs = []
    for r in range(world):
        rs = np.random.default_rng(100 + r)
        rs.normal(size=1000 + 37 * r)
        obs.append(torch.tensor(rs.normal(size=(50 + 10 * r, 5)) + r))
    allx, allo = torch.cat(xs), torch.cat(obs)
    for rank, mean, std, cnt, m, v, n, g, base, eps, pack in res:
        # the kernel's formula (standardize_kernel) on the all-reduced pack
        pm = pack[0] / pack[2]
        ps = np.sqrt(max(0.0, (pack[1] - pack[2] * pm * pm) / max(1.0, pack[2] - 1.0)))
        assert pack[2] == allx.numel() and abs(pm - float(allx.mean())) < 1e-12 and abs(ps - float(allx.std())) < 1e-12
        assert eps == (30.0, 201.0, 7.0)               # rl/algos/ppo.py:408-426: mean over ALL workers' episodes
        assert abs(mean - float(allx.mean())) < 1e-12
        assert abs(std - float(allx.std())) < 1e-12          # unbiased, like ppo.py:485
        assert cnt == allx.numel()
        np.testing.assert_allclose(m, allo.mean(0).numpy(), atol=1e-12)
        np.testing.assert_allclose(v, allo.var(0, unbiased=False).numpy(), atol=1e-12)
        assert n == allo.shape[0]
        np.testing.assert_allclose(g, np.full(7, 1.5))       # mean of the per-rank gradients
        assert base == rank * 4096


def test_running_mean_std_matches_reference_formula():
    from learninghumanoidwalking_amd.ppo import RunningMeanStd
    rs = np.random.default_rng(0)
    rms = RunningMeanStd(shape=(3,))
    xs = [rs.normal(size=(n, 3)) * 2 + 1 for n in (10, 200, 35)]
    for x in xs:
        rms.update(x)
    allx = np.concatenate(xs)
    # Chan's algorithm with the 1e-4 pseudo-count prior (reference rl/envs/normalize.py:10-14)
    assert np.allclose(rms.mean, allx.mean(0), atol=1e-4)
    assert np.allclose(rms.var, allx.var(0), rtol=1e-3)
    assert abs(rms.count - (len(allx) + 1e-4)) < 1e-9
