"""Size-independent properties at the BASELINE batch size (4096 envs per GPU; the oracle only covers small batches):
* batch-size independence: env i of a 4096-env batch produces bit-identical observations / rewards / flags to env i of a
  32-env batch with the same seed (an env owns a wavefront or half of one; nothing may leak between envs, and which
  kernel -- two envs per wave or the one-env-per-wave re-run -- advanced it must not matter);
* run-to-run determinism at full size (bitwise);
* every output finite, no env flagged as diverged, no contact dropped."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,N", [("jvrc_walk", 4096), ("jvrc_step", 4096), ("h1", 4096), ("h1", 8192), ("h1_walk", 4096), ("cartpole", 4096)])
def test_full_batch_equals_small_batch_and_is_deterministic(name, N):
    """(h1 @ 8192: the batch of BASELINE config 5)"""
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    spec = ENVIRONMENTS[name]()
    n, T = 32, 30
    g = torch.Generator(device="cpu").manual_seed(11)
    act = (torch.randn(T, N, spec.act_dim, generator=g) * 0.3).cuda()

    def run(n_envs):
        env = spec.make_batched(n_envs, seed=123, device=0, max_traj_len=12)     # short episodes: resets inside the run
        out = [env.reset().clone()]
        rews, flags = [], []
        for t in range(T):
            obs, rew, done, tob = env.step(act[t, :n_envs].contiguous())
            out.append(obs.clone())
            out.append(tob.clone())
            rews.append(rew.clone())
            flags.append(done.clone())
        faults = env.pop_fault_stats() if name != "cartpole" else (0, 0)
        stats = env.pop_episode_stats()
        env.close()
        return torch.stack(out), torch.stack(rews), torch.stack(flags), faults, stats

    big = run(N)
    big2 = run(N)
    small = run(n)
    assert torch.isfinite(big[0]).all() and torch.isfinite(big[1]).all()
    assert torch.equal(big[0], big2[0]) and torch.equal(big[1], big2[1]) and torch.equal(big[2], big2[2])       # determinism
    assert torch.equal(big[0][:, :n], small[0]) and torch.equal(big[1][:, :n], small[1]) and torch.equal(big[2][:, :n], small[2])
    assert big[3][1] == 0, f"{big[3][1]} env-steps diverged"
    assert big[3][0] == 0, f"{big[3][0]} env-steps dropped contacts beyond the 16-contact layout"
    flags = big[2].cpu().numpy()
    assert (flags & 2).any(), "truncations expected with max_traj_len=12"
    ret, length, count = big[4]
    assert count >= N * (T // 12) * 0.9 and 1 <= length / count <= 12


@pytest.mark.parametrize("name,N", [("jvrc_walk", 4096), ("h1", 8192), ("jvrc_step", 2048), ("jvrc_step", 4096)])
def test_full_size_resident_rollout_equals_small_batch_and_is_deterministic(name, N, monkeypatch):
    """The resident rollout (lhw_env_rollout: one launch, policy step inside the stepper's wavefronts) at the BASELINE batch: env i
    of the full batch stores bit-identical observations / actions / log-densities / rewards / flags to env i of a 32-env batch
    (env ids, hence every random draw, are global; a wavefront owns its envs for the whole rollout), twice the same bits, all
    finite, nothing diverged, no contact dropped.  (jvrc_step @ 4096: more envs than wave slots -- the resident waves drain the job
    queue, three chunks per env; the 32-env batch runs one wave per env.)"""
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    monkeypatch.setenv("LHW_ROLLOUT_MODE", "resident")
    T = 24

    def run(n_envs):
        args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=4096, epochs=1,
                               max_traj_len=T, num_procs=n_envs, num_envs=n_envs, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                               recurrent=False, imitate=None, learn_std=False, std_dev=0.3, no_mirror=True, continued=None,
                               logdir="/tmp/lhw_test_fullsize", device_index=0)
        algo = PPO(ENVIRONMENTS[name], args, seed=21)
        for _ in range(2):
            algo.sample_parallel_with_workers()
        ro = algo.rollout
        assert ro.last_mode == "resident"
        out = [x.clone() for x in (ro.obs, ro.act, ro.logp, ro.rew, ro.done)]
        faults = algo.env.pop_fault_stats()
        algo.env.close()
        return out, faults

    (big, fb), (big2, _), (small, _) = run(N), run(N), run(32)
    for x, y, z in zip(big, big2, small):
        assert torch.isfinite(x.float()).all()
        assert torch.equal(x, y)
        assert torch.equal(x[:, :32], z)
    assert fb == (0, 0), fb
    assert (big[4] != 0).any()
