"""The persistent rollout kernel (lhw_env_rollout: T control steps + in-kernel actor per launch) on the host-side SIMT
emulator, against (a) a numpy restatement of the float32 actor / Gaussian head evaluated on the rollout's own observations and
(b) the launch-per-step path (lhw_env_step) fed with the rollout's actions, which must reproduce every stored observation,
terminal observation, reward and flag BIT FOR BIT -- including auto-resets in mid-rollout, a batch with an odd number of envs
(half-filled last wave) and an env that starts with 13 contacts (handed to the one-env-per-wave layout inside the kernel)."""
import numpy as np

from oracle import rng as orng
from tests import emu


def _fma32(a, b, c):
    """fmaf on float32 operands via float64 (the product is exact in float64)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _actor(w, obs_mean, obs_std, obs):
    x = ((obs - obs_mean) / obs_std).astype(np.float32)

    def layer(W, b, h, relu):
        acc = np.zeros(W.shape[0], np.float32)
        for k in range(W.shape[1]):
            acc = _fma32(np.full(W.shape[0], h[k], np.float32), W[:, k], acc)
        y = acc + b
        return np.maximum(y, 0) if relu else y

    return layer(w["a_w3"], w["a_b3"], layer(w["a_w2"], w["a_b2"], layer(w["a_w1"], w["a_b1"], x, True), True), False)


def test_rollout_kernel_matches_stepwise_path_and_numpy_actor():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    spec = JvrcWalkSpec()
    N, T, H, seed = 3, 7, 256, 11
    rs = np.random.default_rng(0)
    D, A = spec.obs_dim, spec.act_dim
    w = dict(a_w1=(rs.normal(size=(H, D)) * 0.15).astype(np.float32), a_b1=(rs.normal(size=H) * 0.05).astype(np.float32),
             a_w2=(rs.normal(size=(H, H)) * 0.06).astype(np.float32), a_b2=(rs.normal(size=H) * 0.05).astype(np.float32),
             a_w3=(rs.normal(size=(A, H)) * 0.02).astype(np.float32), a_b3=np.zeros(A, np.float32), stds=np.full(A, 0.223, np.float32))
    om, osd = spec.obs_mean.astype(np.float32), spec.obs_std.astype(np.float32)
    lying = np.array([0.0, 0.0, 0.142, -0.7055, -0.3568, 0.5444, 0.2804, 0.2372, -0.0299, -0.0106, 1.5965, -0.2912, -0.7387, -0.6059,
                      -0.272, 0.0915, 0.445, -0.3003, 0.7158])
    lying[3:7] /= np.linalg.norm(lying[3:7])
    envs = []
    for _ in range(2):
        e = emu.make_emulated(spec, N, seed=3, max_traj_len=4)
        obs0 = e.reset().copy()
        q, v = e.get_state()
        q[1] = lying; v[1] = 0          # env 1 lies on the floor with 13 contacts: beyond the two-envs-per-wave layout
        e.set_state(q, v)
        envs.append(e)
    A_env, B_env = envs
    ro = A_env.rollout(T, w, om, osd, obs0, seed=seed, env_id_base=0, counter0=5)
    assert A_env.pop_rerun_count() > 0 and A_env.pop_fault_stats() == (0, 0)
    assert (ro["done"] & 2).any(), "truncation (max_traj_len = 4) expected inside the rollout"
    for t in range(T):
        for n in range(N):     # (a) actor + Gaussian head
            mu = _actor(w, om, osd, ro["obs"][t, n])
            lp = np.float32(0)
            for a in range(A):
                u1, u2 = orng.u01(seed, n, orng.STREAM_POLICY, 5 + t, 2 * a), orng.u01(seed, n, orng.STREAM_POLICY, 5 + t, 2 * a + 1)
                z = np.float32(np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(6.283185307179586 * u2))
                x = _fma32(np.float32(0.223), z, mu[a])
                np.testing.assert_allclose(ro["act"][t, n, a], x, rtol=2e-6, atol=2e-7, err_msg=f"action t={t} env={n} a={a}")
                d = (ro["act"][t, n, a] - mu[a]) / np.float32(0.223)
                lp += np.float32(-0.5) * d * d - np.log(np.float32(0.223)) - np.float32(0.9189385332046727)
            np.testing.assert_allclose(ro["logp"][t, n], lp, rtol=1e-4, atol=1e-4)
        # (b) the launch-per-step path on the rollout's actions
        obs, rew, done, tob = B_env.step(ro["act"][t])
        np.testing.assert_array_equal(obs, ro["obs"][t + 1], err_msg=f"obs t={t}")
        np.testing.assert_array_equal(tob, ro["tob"][t], err_msg=f"terminal obs t={t}")
        np.testing.assert_array_equal(rew, ro["rew"][t], err_msg=f"reward t={t}")
        np.testing.assert_array_equal(done, ro["done"][t], err_msg=f"flags t={t}")
    np.testing.assert_array_equal(B_env.rew_terms, ro["rew_terms"])
    for a, b in zip(A_env.get_state(), B_env.get_state()):
        np.testing.assert_array_equal(a, b)
    assert A_env.pop_episode_stats() == B_env.pop_episode_stats()
