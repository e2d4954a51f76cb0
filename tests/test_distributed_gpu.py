"""Data-parallel readiness without an 8-GPU box: two ranks sharing the one GPU (gloo) must end two PPO iterations with the
weights of a single process that holds the union of their envs and takes the merged minibatches -- i.e. env-id sharding,
global advantage statistics and gradient averaging BEFORE the norm clips are the single-process semantics of the reference
(rl/algos/ppo.py:393-394, 484-485)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_equiv_worker.py")


def test_two_ranks_reproduce_the_single_process_union(tmp_path):
    f2, f1 = str(tmp_path / "ranks.npy"), str(tmp_path / "union.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for port in ("29571", "29577"):     # (one retry on another port if the launcher itself fails, e.g. a rendezvous hiccup on a cold box;
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", port, WORKER, "--mode", "ranks", "--out", f2], capture_output=True, text=True, timeout=900, env=env)
        if r.returncode == 0:           # a numerical mismatch below is never retried)
            break
        print(r.stdout[-2000:] + r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, WORKER, "--mode", "union", "--out", f1], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a, b = np.load(f2), np.load(f1)
    assert a.shape == b.shape and np.isfinite(a).all()
    moved = np.abs(a - a.mean()).max()
    assert np.abs(a - b).max() <= 1e-6, f"max |theta_ranks - theta_union| = {np.abs(a - b).max():.3e} (weights span {moved:.2f})"


def test_rccl_branch_of_bench_runs_with_one_rank():
    """bench.py's RCCL path (init_process_group("nccl"), barrier, MAX all-reduce of the timing, gradient all-reduce inside
    PPO) is otherwise reached only on a multi-GPU node: run it with world size 1 on this box."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LHW_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--num-envs", "128",
                        "--traj-len", "8", "--minibatch-size", "512", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0
