"""Pins the CPU oracle of the learner math to fixtures generated from the REFERENCE's own classes
(tests/golden/gen_golden.py).  CPU-only."""
import os

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MIR_OBS = [-0.1, 1, -2, 3, -4, 11, -12, -13, 14, -15, 16, 5, -6, -7, 8, -9, 10, 23, -24, -25, 26, -27, 28, 17, -18, -19,
           20, -21, 22] + list(range(29, 37))
MIR_ACT = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]


def test_gae_known_answer_and_random():
    g = np.load(os.path.join(G, "gae.npz"))
    toy = po.gae_returns(np.ones(5, np.float32), np.full(5, 0.5, np.float32), 0.25, 0.99, 0.95)
    np.testing.assert_allclose(toy, [4.723518165117246, 3.9327678523309375, 3.091991336875, 2.19802375, 1.2475], rtol=0, atol=1e-14)
    np.testing.assert_allclose(toy, g["toy_returns"], rtol=0, atol=1e-14)
    start, out = 0, []
    for e, last in zip(g["ends"], g["lasts"]):
        out.append(po.gae_returns(g["rew"][start:e + 1], g["val"][start:e + 1], last, 0.99, 0.95))
        start = e + 1
    np.testing.assert_allclose(np.concatenate(out), g["returns"], rtol=0, atol=1e-13)


def test_mirror_tables_match_reference_matrices():
    g = np.load(os.path.join(G, "misc.npz"))
    np.testing.assert_array_equal(po.symmetry_matrix(MIR_OBS), g["mir_obs"])
    np.testing.assert_array_equal(po.symmetry_matrix(MIR_ACT), g["mir_act"])
    src, sign = po.mirror_tables(MIR_OBS)
    x = np.random.default_rng(0).normal(size=(5, 37))
    np.testing.assert_allclose(x[:, src] * sign, x @ g["mir_obs"], rtol=0, atol=0)
    # involution
    np.testing.assert_array_equal(g["mir_obs"] @ g["mir_obs"], np.eye(37))
    np.testing.assert_array_equal(g["mir_act"] @ g["mir_act"], np.eye(12))


@pytest.mark.parametrize("tag", ["h64_mirror", "h64_learnstd"])
def test_update_actor_critic_matches_reference(tag):
    g = np.load(os.path.join(G, f"ppo_{tag}.npz"))
    mirror, learn_std = bool(g["mirror"]), bool(g["learn_std"])
    mo = po.mirror_tables(MIR_OBS, [29, 30]) if mirror else None
    ma = po.mirror_tables(MIR_ACT) if mirror else None
    orc = po.OraclePPO([g[f"a0_{k}"] for k in range(6)], [g[f"c0_{k}"] for k in range(6)], g["stds0"], g["obs_mean"], g["obs_std"],
                       entropy_coeff=0.01 if learn_std else 0.0, learn_std=learn_std, mirror_obs=mo, mirror_act=ma)
    for u in range(len(g["scalars"])):
        t = lambda k: torch.tensor(g[f"{k}_{u}"])
        res = orc.update(t("obs"), t("act"), t("ret"), t("adv"), t("old_logp"))
        np.testing.assert_allclose(res, g["scalars"][u], rtol=2e-5, atol=2e-6)
    for k in range(6):
        np.testing.assert_allclose(orc.actor[k].detach().numpy(), g[f"a1_{k}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(orc.critic[k].detach().numpy(), g[f"c1_{k}"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(orc.stds.detach().numpy(), g["stds1"], rtol=0, atol=2e-6)


def test_update_with_imitation_matches_reference():
    """Reference PPO.update_actor_critic with an imitation projector + frozen expert (gen_golden.py gen_ppo imitate=True)."""
    g = np.load(os.path.join(G, "ppo_h64_imitate.npz"))
    mo, ma = po.mirror_tables(MIR_OBS, [29, 30]), po.mirror_tables(MIR_ACT)
    orc = po.OraclePPO([g[f"a0_{k}"] for k in range(6)], [g[f"c0_{k}"] for k in range(6)], g["stds0"], g["obs_mean"], g["obs_std"],
                       mirror_obs=mo, mirror_act=ma)
    expert = [torch.tensor(g[f"e_{k}"]) for k in range(6)]
    for u in range(len(g["scalars"])):
        t = lambda k: torch.tensor(g[f"{k}_{u}"])
        obs = t("obs")
        smask = obs[:, 0] > 0
        target = po.mlp(obs[smask][:, :20], *expert)          # expert normalisation is the identity in the fixture
        res = orc.update(obs, t("act"), t("ret"), t("adv"), t("old_logp"), imit=(0.3, smask, torch.tensor([0, 2, 5]), target))
        np.testing.assert_allclose(res, g["scalars"][u], rtol=2e-5, atol=2e-6)
        assert res[5] > 0
    for k in range(6):
        np.testing.assert_allclose(orc.actor[k].detach().numpy(), g[f"a1_{k}"], rtol=0, atol=2e-6)


def _rppo_case(g, u):
    """Columns-with-resets view of update u of a recurrent fixture (trajectories stored back to back)."""
    lengths = [int(x) for x in g["lengths"]]
    T = max(lengths)
    cols = po.trajectories_to_columns(lengths, T)
    B = len(cols)

    def pack(a):
        out = np.zeros((T, B) + a.shape[1:], np.float32)
        for b, segs in enumerate(cols):
            t = 0
            for off, n in segs:
                out[t:t + n, b] = a[off:off + n]
                t += n
        return torch.tensor(out)
    reset = np.zeros((T, B), bool)
    for b, segs in enumerate(cols):
        t = 0
        for off, n in segs:
            reset[t, b] = True
            t += n
    return (pack(g[f"obs_{u}"]), torch.tensor(reset), pack(g[f"act_{u}"]), pack(g[f"ret_{u}"]), pack(g[f"adv_{u}"]),
            pack(g[f"old_logp_{u}"]))


@pytest.mark.parametrize("tag", ["h32_padded", "h32_mirror"])
def test_recurrent_update_matches_reference(tag):
    """Reference Gaussian_LSTM_Actor / LSTM_V + recurrent update_actor_critic on a PADDED trajectory list vs the oracle's
    padding-free columns with in-column resets: same losses, same weights after two Adam steps."""
    g = np.load(os.path.join(G, f"rppo_{tag}.npz"))
    mirror = bool(g["mirror"])
    mo = po.mirror_tables(MIR_OBS, [29, 30]) if mirror else None
    ma = po.mirror_tables(MIR_ACT) if mirror else None
    orc = po.OracleRecurrentPPO([g[f"a0_{k}"] for k in range(10)], [g[f"c0_{k}"] for k in range(10)], g["stds0"], g["obs_mean"],
                                g["obs_std"], mirror_obs=mo, mirror_act=ma)
    for u in range(len(g["scalars"])):
        obs, reset, act, ret, adv, old_logp = _rppo_case(g, u)
        a_loss, c_loss, m_loss = orc.update(obs, reset, act, ret, adv, old_logp)
        ref = g["scalars"][u]      # actor, entropy, critic, kl, mirror, imitation, clip fraction
        np.testing.assert_allclose([a_loss, c_loss, m_loss], [ref[0], ref[2], ref[4]], rtol=3e-5, atol=2e-6)
    for k in range(10):
        np.testing.assert_allclose(orc.actor[k].detach().numpy(), g[f"a1_{k}"], rtol=0, atol=2e-6, err_msg=f"actor {k}")
        np.testing.assert_allclose(orc.critic[k].detach().numpy(), g[f"c1_{k}"], rtol=0, atol=2e-6, err_msg=f"critic {k}")
