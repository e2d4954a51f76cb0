"""The persistent rollout (lhw_env_rollout: one launch for T control steps, actor evaluated inside the stepper) against the
launch-per-step path on the GPU: the stepper's part must agree BIT FOR BIT when fed the same actions (observations, terminal
observations, rewards, flags, final state, episode statistics), and the in-kernel actor must reproduce the GEMM path's means /
samples / log-densities on the same observations (same float32 operation order; the tolerance below only allows for the
MFMA's internal accumulation differing from an fmaf chain in the last bit).  Covers auto-resets in mid-rollout, an odd batch
(half-filled last wave), and envs that start in poses with more than 8 contacts (handed to the one-env-per-wave layout inside
the kernel).  (tests/test_emu_rollout.py runs the same kernel source on the CPU emulator against a numpy actor.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(spec, N, seed, max_traj_len):
    return spec.make_batched(N, seed=seed, device=0, max_traj_len=max_traj_len)


def _kernels(spec, N, seed):
    import torch
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
    k = PpoKernels(spec.obs_dim, spec.act_dim, hidden=256, max_rows=max(N, 64), device=0)
    k.set_tensors(reference_init(spec.obs_dim, spec.act_dim, hidden=256, generator_seed=seed))
    if spec.obs_mean is not None:
        k.set_obs_norm(spec.obs_mean, spec.obs_std)
    return k


@pytest.mark.parametrize("task,N,T,mtl", [("jvrc_walk", 67, 24, 9), ("h1", 33, 12, 5), ("h1_walk", 18, 12, 0)])
def test_persistent_rollout_equals_stepwise(task, N, T, mtl, monkeypatch):
    import torch
    monkeypatch.setenv("LHW_ROLLOUT_PERSISTENT", "1")     # (the one-launch rollout is opt-in)
    from learninghumanoidwalking_amd.envs.h1 import H1Spec
    from learninghumanoidwalking_amd.envs.h1_walk import H1WalkSpec
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from learninghumanoidwalking_amd.ppo import Rollout
    spec = dict(jvrc_walk=JvrcWalkSpec, h1=H1Spec, h1_walk=H1WalkSpec)[task]()
    envA, envB = _make(spec, N, 5, mtl), _make(spec, N, 5, mtl)
    assert envA.supports_rollout
    k = _kernels(spec, N, 3)
    ro = Rollout(envA, k, T, seed=17)
    assert ro.persistent
    # some envs start lying on the floor (many contacts): those control steps leave the two-envs-per-wave layout
    obsA, obsB = envA.reset(), envB.reset()
    q, v = envA.get_state()
    rs = np.random.default_rng(1)
    for i in range(0, N, 7):
        q[i, 2] = 0.25
        quat = rs.normal(size=4); q[i, 3:7] = quat / np.linalg.norm(quat)
        q[i, 7:] += rs.normal(size=q.shape[1] - 7) * 0.4
    envA.set_state(q, v); envB.set_state(q, v)
    # one ordinary control step on both sides yields the observation the rollout starts from (Rollout.collect would reset)
    zero = torch.zeros(N, spec.act_dim, device=envA.device)
    o1, _, _, _ = envA.step(zero)
    o2, _, _, _ = envB.step(zero)
    assert torch.equal(o1, o2)
    ro.obs[T].copy_(o1)
    ro.started = True
    # two batches: episodes carry over from one launch to the next
    reruns = 0
    for batch in range(2):
        ro.collect()
        torch.cuda.synchronize()
        reruns += envA.pop_rerun_count()
        for t in range(T):
            # the stepper, fed the persistent rollout's actions
            obs, rew, done, tob = envB.step(ro.act[t].contiguous())
            assert torch.equal(obs, ro.obs[t + 1]), f"obs batch {batch} t {t}: max |diff| {(obs - ro.obs[t + 1]).abs().max().item():.3e} in envs {torch.nonzero((obs != ro.obs[t + 1]).any(1)).flatten().tolist()[:20]}"
            assert torch.equal(rew, ro.rew[t]), f"reward batch {batch} t {t}"
            assert torch.equal(done, ro.done[t]), f"flags batch {batch} t {t}"
            assert torch.equal(tob, ro.tob_all[t]), f"terminal obs batch {batch} t {t}"
            # the actor: GEMM path on the same observation, same RNG keys
            mu, act, logp, _ = k.forward(ro.obs[t], seed=17, env_id_base=0, counter=batch * T + t, want_value=False)
            np.testing.assert_allclose(ro.act[t].cpu().numpy(), act.cpu().numpy(), rtol=0, atol=2e-6, err_msg=f"actions batch {batch} t {t}")
            np.testing.assert_allclose(ro.logp[t].cpu().numpy(), logp.cpu().numpy(), rtol=0, atol=2e-4, err_msg=f"log-density batch {batch} t {t}")
        assert torch.equal(envB.rew_terms, envA.rew_terms)
    assert ro.done.any(), "episode ends expected inside the rollouts"
    qa, va = envA.get_state(); qb, vb = envB.get_state()
    np.testing.assert_array_equal(qa, qb); np.testing.assert_array_equal(va, vb)
    (ra, la, ca), (rb, lb, cb) = envA.pop_episode_stats(), envB.pop_episode_stats()
    assert (la, ca) == (lb, cb) and abs(ra - rb) <= 1e-9 * max(1.0, abs(rb))   # (the return sum is an atomic float accumulation)
    assert envA.pop_fault_stats() == (0, 0)
    print(f"{task}: {reruns} control steps re-run with the one-env-per-wave layout inside the rollout launches")
    if task == "jvrc_walk":
        assert reruns > 0


def test_persistent_and_stepwise_collect_agree():
    """Rollout.collect with and without the persistent launch: same buffers up to the actor's last-bit differences for as long as
    the trajectories have not separated -- checked on the first control steps (bitwise equality is reported, not required)."""
    import os
    import torch
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from learninghumanoidwalking_amd.ppo import Rollout
    spec = JvrcWalkSpec()
    N, T = 64, 6
    out = []
    for flag in ("1", "0"):
        os.environ["LHW_ROLLOUT_PERSISTENT"] = flag
        try:
            env = _make(spec, N, 2, 0)
            ro = Rollout(env, _kernels(spec, N, 4), T, seed=9)
            assert ro.persistent == (flag == "1")
            ro.collect()
            torch.cuda.synchronize()
            out.append(ro)
        finally:
            os.environ.pop("LHW_ROLLOUT_PERSISTENT", None)
    a, b = out
    same = all(torch.equal(getattr(a, n), getattr(b, n)) for n in ("obs", "act", "logp", "rew", "done", "val", "vterm", "vfinal"))
    print("persistent vs launch-per-step rollout buffers bitwise identical:", same)
    np.testing.assert_allclose(a.act[0].cpu().numpy(), b.act[0].cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(a.obs[:3].cpu().numpy(), b.obs[:3].cpu().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(a.val[:2].cpu().numpy(), b.val[:2].cpu().numpy(), rtol=0, atol=1e-4)
