"""Recurrent (LSTM) PPO kernels vs fixtures produced by the reference's Gaussian_LSTM_Actor / LSTM_V + recurrent
update_actor_critic (tests/golden/rppo_*.npz) and vs the CPU oracle, through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MIR_OBS = [-0.1, 1, -2, 3, -4, 11, -12, -13, 14, -15, 16, 5, -6, -7, 8, -9, 10, 23, -24, -25, 26, -27, 28, 17, -18, -19,
           20, -21, 22] + list(range(29, 37))
MIR_ACT = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]
NET = ["wih1", "whh1", "bih1", "bhh1", "wih2", "whh2", "bih2", "bhh2", "wout", "bout"]


def _kernels(g, mirror, T, cols, rows=64):
    from learninghumanoidwalking_amd.rnn_kernels import RnnKernels
    from oracle import ppo_oracle as po
    mo = po.mirror_tables(MIR_OBS, [29, 30]) if mirror else None
    ma = po.mirror_tables(MIR_ACT) if mirror else None
    k = RnnKernels(37, 12, hidden=int(g["hidden"]), seq_len=T, seq_cols=cols, rollout_rows=rows, mirror_obs=mo, mirror_act=ma)
    k.set_obs_norm(g["obs_mean"], g["obs_std"])
    k.set_tensors({f"a_{n}": g[f"a0_{i}"] for i, n in enumerate(NET)})
    k.set_tensors({f"c_{n}": g[f"c0_{i}"] for i, n in enumerate(NET)})
    k.set_tensors({"stds": g["stds0"]})
    return k


def _columns(g, u):
    from tests.test_oracle_ppo import _rppo_case
    obs, reset, act, ret, adv, old_logp = _rppo_case(g, u)
    T, B = reset.shape
    done = torch.zeros(T, B, dtype=torch.uint8)
    done[:-1][reset[1:]] = 1          # an episode ends at t where the next step starts a new one
    return obs, reset, act, ret, adv, old_logp, done


@pytest.mark.parametrize("tag", ["h32_padded", "h32_mirror"])
def test_bptt_update_matches_reference_fixture(tag):
    g = np.load(os.path.join(G, f"rppo_{tag}.npz"))
    mirror = bool(g["mirror"])
    obs0 = _columns(g, 0)[0]
    T, B = obs0.shape[:2]
    k = _kernels(g, mirror, T, B)
    for u in range(len(g["scalars"])):
        obs, reset, act, ret, adv, old_logp, done = _columns(g, u)
        c = lambda x: x.cuda().contiguous()
        xn, xm = k.normalize(c(obs.reshape(T * B, -1)))
        k.stats.zero_()
        cols = torch.arange(B, dtype=torch.int32, device="cuda")
        k.grad_columns(T, B, xn, xm, c(act.reshape(T * B, -1)), c(old_logp.reshape(-1)), c(adv.reshape(-1)), c(ret.reshape(-1)), c(done), cols)
        k.apply()
        s = k.stats.cpu().numpy()
        ref = g["scalars"][u]
        np.testing.assert_allclose([s[0], s[1], s[2]], [ref[0], ref[2], ref[4]], rtol=2e-4, atol=2e-6, err_msg=f"losses update {u}")
    t = k.get_tensors()
    for i, n in enumerate(NET):
        np.testing.assert_allclose(t[f"a_{n}"].numpy(), g[f"a1_{i}"], rtol=0, atol=3e-6, err_msg=f"actor {n}")
        np.testing.assert_allclose(t[f"c_{n}"].numpy(), g[f"c1_{i}"], rtol=0, atol=3e-6, err_msg=f"critic {n}")


def test_rollout_steps_equal_sequence_forward_and_oracle():
    """lhw_rnn_forward stepped over T control steps with episode-start resets == the oracle's sequence forward; commit=0
    leaves the state untouched; a column subset in a different order gives the same gradient contribution."""
    from oracle import ppo_oracle as po
    g = np.load(os.path.join(G, "rppo_h32_padded.npz"))
    obs, reset, act, ret, adv, old_logp, done = _columns(g, 0)
    T, B = reset.shape
    k = _kernels(g, False, T, B, rows=B)
    orc = po.OracleRecurrentPPO([g[f"a0_{i}"] for i in range(10)], [g[f"c0_{i}"] for i in range(10)], g["stds0"], g["obs_mean"], g["obs_std"])
    with torch.no_grad():
        mu_ref, v_ref = orc.mu(obs, reset).numpy(), orc.value(obs, reset).numpy()[..., 0]
    mus, vals = [], []
    for t in range(T):
        r = reset[t].to(torch.uint8).cuda()
        o = obs[t].cuda().contiguous()
        # an evaluate-only call in between must not disturb the trajectory
        k.forward(torch.randn_like(o), commit=False, want_actor=False)
        mu, a, lp, v = k.forward(o, reset=r, deterministic=True, commit=True)
        mus.append(mu.cpu().numpy())
        vals.append(v.cpu().numpy())
        np.testing.assert_allclose(a.cpu().numpy(), mu.cpu().numpy())
    np.testing.assert_allclose(np.array(mus), mu_ref, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(np.array(vals), v_ref, rtol=1e-4, atol=2e-6)
    # sampled actions: logp consistent with the oracle's distribution
    mu, a, lp, _ = k.forward(obs[0].cuda().contiguous(), reset=torch.ones(B, dtype=torch.uint8, device="cuda"), seed=3, counter=1, commit=True)
    ref_lp = torch.distributions.Normal(mu.cpu(), torch.tensor(g["stds0"])).log_prob(a.cpu()).sum(-1)
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), rtol=1e-4, atol=1e-4)
    assert float((a - mu).abs().max()) > 1e-3


def test_larger_network_and_longer_sequences_against_oracle():
    """2 x 64 LSTM, T = 24, 10 columns with random episode boundaries, mirror loss on: losses and post-Adam weights vs the
    oracle's autograd."""
    from learninghumanoidwalking_amd.rnn_kernels import RnnKernels
    from oracle import ppo_oracle as po
    rs = np.random.default_rng(5)
    D, A, H, T, N = 37, 12, 64, 24, 10
    mo, ma = po.mirror_tables(MIR_OBS, [29, 30]), po.mirror_tables(MIR_ACT)
    k = RnnKernels(D, A, hidden=H, seq_len=T, seq_cols=6, rollout_rows=N, mirror_obs=mo, mirror_act=ma)
    torch.manual_seed(0)
    shapes = [(4 * H, D), (4 * H, H), (4 * H,), (4 * H,), (4 * H, H), (4 * H, H), (4 * H,), (4 * H,)]
    aw = [torch.randn(*s) * 0.1 for s in shapes] + [torch.randn(A, H) * 0.05, torch.randn(A) * 0.01]
    cw = [torch.randn(*s) * 0.1 for s in shapes] + [torch.randn(1, H) * 0.1, torch.randn(1) * 0.01]
    k.set_tensors({f"a_{n}": w for n, w in zip(NET, aw)})
    k.set_tensors({f"c_{n}": w for n, w in zip(NET, cw)})
    stds = torch.full((A,), 0.223)
    k.set_tensors({"stds": stds})
    om, osd = rs.normal(size=D).astype(np.float32) * 0.1, (0.5 + rs.uniform(size=D)).astype(np.float32)
    k.set_obs_norm(om, osd)
    orc = po.OracleRecurrentPPO([w.numpy() for w in aw], [w.numpy() for w in cw], stds.numpy(), om, osd, mirror_obs=mo, mirror_act=ma)
    obs = torch.tensor(rs.normal(size=(T, N, D)).astype(np.float32))
    done = torch.tensor((rs.uniform(size=(T, N)) < 0.12).astype(np.uint8) * rs.integers(1, 3, size=(T, N)).astype(np.uint8))
    reset = torch.zeros(T, N, dtype=torch.bool)
    reset[0] = True
    reset[1:] = done[:-1] != 0
    with torch.no_grad():
        mu = orc.mu(obs, reset)
    act = mu + 0.223 * torch.tensor(rs.normal(size=mu.shape).astype(np.float32))
    with torch.no_grad():
        old_logp = orc.log_prob(obs, reset, act) + torch.tensor(rs.normal(size=(T, N, 1)).astype(np.float32)) * 0.05
    ret = torch.tensor(rs.normal(size=(T, N, 1)).astype(np.float32))
    adv = torch.tensor(rs.normal(size=(T, N, 1)).astype(np.float32))
    cols = torch.tensor([7, 2, 9, 0, 4], dtype=torch.int32)
    c = lambda x: x.cuda().contiguous()
    xn, xm = k.normalize(c(obs.reshape(T * N, D)))
    k.stats.zero_()
    k.grad_columns(T, N, xn, xm, c(act.reshape(T * N, A)), c(old_logp.reshape(-1)), c(adv.reshape(-1)), c(ret.reshape(-1)), c(done), cols.cuda())
    k.apply()
    ci = cols.long()
    a_loss, c_loss, m_loss = orc.update(obs[:, ci], reset[:, ci], act[:, ci], ret[:, ci], adv[:, ci], old_logp[:, ci])
    s = k.stats.cpu().numpy()
    np.testing.assert_allclose([s[0], s[1], s[2]], [a_loss, c_loss, m_loss], rtol=3e-4, atol=2e-6)
    t = k.get_tensors()
    for i, n in enumerate(NET):
        np.testing.assert_allclose(t[f"a_{n}"].numpy(), orc.actor[i].detach().numpy(), rtol=0, atol=5e-6, err_msg=f"actor {n}")
        np.testing.assert_allclose(t[f"c_{n}"].numpy(), orc.critic[i].detach().numpy(), rtol=0, atol=5e-6, err_msg=f"critic {n}")


def _args(tmp, **kw):
    from types import SimpleNamespace
    d = dict(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=16, epochs=2, max_traj_len=12,
             num_procs=32, num_envs=32, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=True, imitate=None,
             learn_std=False, std_dev=0.223, no_mirror=False, continued=None, logdir=str(tmp), device_index=0, lstm_hidden=64)
    d.update(kw)
    return SimpleNamespace(**d)


def test_recurrent_ppo_rollout_is_consistent_with_sequence_forward(tmp_path):
    """jvrc_walk with LSTM policies: the log-probs and values stored step by step during the rollout (hidden state carried,
    reset at episode ends, terminal values evaluated on the side) equal the oracle's sequence forward over the stored
    observations with resets derived from the done flags; then one optimisation phase runs and checkpoints round-trip."""
    from learninghumanoidwalking_amd.checkpoint import load_recurrent_checkpoint
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    from oracle import ppo_oracle as po
    # large exploration noise so that robots fall (episodes end) inside the 48-step batch
    algo = PPO(ENVIRONMENTS["jvrc_walk"], _args(tmp_path, max_traj_len=48, std_dev=0.9), seed=3)
    k = algo.kernels
    algo.sample_parallel_with_workers()
    ro = algo.rollout
    T, N = ro.T, ro.N
    t = k.get_tensors()
    names = ["wih1", "whh1", "bih1", "bhh1", "wih2", "whh2", "bih2", "bhh2", "wout", "bout"]
    orc = po.OracleRecurrentPPO([t[f"a_{n}"].numpy() for n in names], [t[f"c_{n}"].numpy() for n in names], t["stds"].numpy(),
                                k.obs_mean.cpu().numpy(), k.obs_std.cpu().numpy())
    obs, done = ro.obs[:T].cpu(), ro.done.cpu()
    reset = torch.zeros(T, N, dtype=torch.bool)
    reset[0] = True
    reset[1:] = done[:-1] != 0
    assert int(reset[1:].sum()) > 0, "no episode ended inside the batch: resets not exercised"
    with torch.no_grad():
        logp = orc.log_prob(obs, reset, ro.act.cpu())[..., 0]
        val = orc.value(obs, reset)[..., 0]
    np.testing.assert_allclose(ro.logp.cpu().numpy(), logp.numpy(), rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(ro.val.cpu().numpy(), val.numpy(), rtol=1e-3, atol=3e-4)
    w0 = k.theta.clone()
    losses = algo.optimize(0)
    assert losses["n_updates"] == 2 * 2 and np.isfinite([losses["actor"], losses["critic"], losses["mirror"]]).all()
    assert float((k.theta - w0).abs().max()) > 0
    algo.save(0, metric=1.0)
    t2, om, osd, hidden = load_recurrent_checkpoint(tmp_path / "actor_0.pt", tmp_path / "critic_0.pt")
    assert hidden == 64
    cur = k.get_tensors()
    for n in cur:
        np.testing.assert_array_equal(t2[n].numpy(), cur[n].numpy(), err_msg=n)
    # --continued picks the files up again
    algo2 = PPO(ENVIRONMENTS["jvrc_walk"], _args(tmp_path, max_traj_len=48, continued=str(tmp_path / "actor_0.pt")), seed=3)
    np.testing.assert_array_equal(algo2.kernels.get_tensors()["a_whh2"].numpy(), cur["a_whh2"].numpy())


def test_recurrent_training_same_seed_is_identical(tmp_path):
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    def run():
        algo = PPO(ENVIRONMENTS["cartpole"], _args(tmp_path, max_traj_len=16, num_procs=64, num_envs=64, minibatch_size=32, no_mirror=True), seed=7)
        for itr in range(2):
            algo.iterate(itr)
        return algo.kernels.theta.clone()
    a, b = run(), run()
    assert torch.equal(a, b)
