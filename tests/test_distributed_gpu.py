"""Data-parallel readiness without an 8-GPU box: two ranks sharing the one GPU (gloo) must end two PPO iterations with the
weights of a single process that holds the union of their envs and takes the merged minibatches -- i.e. env-id sharding,
global advantage statistics and gradient averaging BEFORE the norm clips are the single-process semantics of the reference
(rl/algos/ppo.py:393-394, 484-485).  The comparison is staged (tests/dp_equiv_worker.py dumps every stage), so a failure names the
first stage that differs: rollout -> returns -> normalised advantages -> first gradient -> weights."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_equiv_worker.py")


def _run(cmd, env, tag):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, f"{tag} failed (rc {r.returncode}):\n" + r.stdout[-3000:] + r.stderr[-3000:]


def _launch_ranks(out, env, port, extra=()):
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
          "--master-port", port, WORKER, "--mode", "ranks", "--out", out, *extra], env, "2-rank launch")


def _compare_staged(out2, out1, n=64, iters=2):
    r0, r1, u = (dict(np.load(f)) for f in (out2 + ".rank0.npz", out2 + ".rank1.npz", out1 + ".rank0.npz"))
    for d in (r0, r1, u):
        assert not d["faults"].any(), f"contact overflows / diverged envs: {d['faults']}"
    # (1) the two ranks hold the same weights and applied the same averaged gradient, bit for bit (an all-reduce that races with
    # the gradient kernels or with Adam shows up here first)
    for i in range(iters):
        np.testing.assert_array_equal(r0[f"g1_{i}"], r1[f"g1_{i}"], err_msg=f"all-reduced gradient differs between the ranks, iteration {i}")
        np.testing.assert_array_equal(r0[f"theta_{i}"], r1[f"theta_{i}"], err_msg=f"theta differs between the ranks after iteration {i}")
    # (2) iteration 0 starts from identical weights: each rank's rollout is bitwise the union's columns of its envs
    for r, d in enumerate((r0, r1)):
        cols = slice(r * n, (r + 1) * n)
        for name in ("obs", "act", "logp", "rew", "done", "ret"):
            np.testing.assert_array_equal(d[f"{name}_0"], u[f"{name}_0"][:, cols], err_msg=f"rank {r} rollout `{name}` != union columns, iteration 0")
        # (3) advantages normalised with the GLOBAL mean / unbiased std (two partial sums merged vs one sum: rounding only)
        np.testing.assert_allclose(d["adv_0"], u["adv_0"][:, cols], rtol=0, atol=2e-5, err_msg=f"rank {r} normalised advantages, iteration 0")
    # (4) first minibatch: mean of the ranks' gradients == gradient of the merged minibatch (summation order differs)
    g, gu = r0["g1_0"], u["g1_0"]
    assert np.abs(g - gu).max() <= 1e-5 * max(1.0, np.abs(gu).max()), f"first averaged gradient: max diff {np.abs(g - gu).max():.3e} (|g| max {np.abs(gu).max():.3e})"
    # (5) weights after every iteration
    for i in range(iters):
        a, b = r0[f"theta_{i}"], u[f"theta_{i}"]
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 1e-6, f"max |theta_ranks - theta_union| = {np.abs(a - b).max():.3e} after iteration {i} (weights span {np.abs(a - a.mean()).max():.2f})"


def test_rccl_branch_of_bench_runs_with_one_rank():
    """bench.py's RCCL path (init_process_group("nccl"), barrier, MAX all-reduce of the timing, gradient all-reduce inside
    PPO) is otherwise reached only on a multi-GPU node: run it with world size 1 on this box."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LHW_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--num-envs", "128",
                        "--traj-len", "8", "--minibatch-size", "512", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0


def test_two_ranks_reproduce_the_single_process_union(tmp_path):
    f2, f1 = str(tmp_path / "ranks"), str(tmp_path / "union")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    _launch_ranks(f2, env, "29571")
    _run([sys.executable, WORKER, "--mode", "union", "--out", f1], env, "union run")
    _compare_staged(f2, f1)
