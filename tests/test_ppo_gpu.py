"""On-device PPO kernels vs (a) fixtures produced by the reference's own PPO.update_actor_critic and
(b) the CPU oracle, through the C ABI.  float32 network math: tolerances are stated per check."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MIR_OBS = [-0.1, 1, -2, 3, -4, 11, -12, -13, 14, -15, 16, 5, -6, -7, 8, -9, 10, 23, -24, -25, 26, -27, 28, 17, -18, -19,
           20, -21, 22] + list(range(29, 37))
MIR_ACT = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]
NAMES = ["w1", "b1", "w2", "b2", "w3", "b3"]


def _kernels(g, hidden, mirror, learn_std, max_rows=256):
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels
    from oracle import ppo_oracle as po
    mo = po.mirror_tables(MIR_OBS, [29, 30]) if mirror else None
    ma = po.mirror_tables(MIR_ACT) if mirror else None
    k = PpoKernels(37, 12, hidden=hidden, max_rows=max_rows, learn_std=learn_std, entropy_coeff=0.01 if learn_std else 0.0,
                   mirror_obs=mo, mirror_act=ma)
    k.set_obs_norm(g["obs_mean"], g["obs_std"])
    return k


def _run_updates(k, g):
    scal = []
    for u in range(len(g["scalars"])):
        c = lambda n: torch.tensor(g[f"{n}_{u}"]).cuda().contiguous()
        obs = c("obs")
        B = obs.shape[0]
        xn, xm = k.normalize(obs)
        k.stats.zero_()
        idx = torch.arange(B, dtype=torch.int32, device="cuda")
        k.grad_minibatch(xn, xm, c("act"), c("old_logp").view(-1), c("adv").view(-1), c("ret").view(-1), idx)
        k.apply()
        s = k.stats.cpu().numpy()
        scal.append(s[:5].copy())
    return np.array(scal)


@pytest.mark.parametrize("tag", ["h64_mirror", "h64_learnstd"])
def test_update_matches_reference_fixture(tag):
    g = np.load(os.path.join(G, f"ppo_{tag}.npz"))
    mirror, learn_std = bool(g["mirror"]), bool(g["learn_std"])
    k = _kernels(g, 64, mirror, learn_std)
    k.set_tensors({f"a_{n}": g[f"a0_{i}"] for i, n in enumerate(NAMES)})
    k.set_tensors({f"c_{n}": g[f"c0_{i}"] for i, n in enumerate(NAMES)})
    k.set_tensors({"stds": g["stds0"]})
    scal = _run_updates(k, g)
    ref = g["scalars"]  # actor_loss, entropy_penalty, critic_loss, approx_kl, mirror_loss, imitation, clip_fraction
    np.testing.assert_allclose(scal[:, 0], ref[:, 0], rtol=1e-4, atol=2e-6)   # actor loss
    np.testing.assert_allclose(scal[:, 1], ref[:, 2], rtol=1e-4, atol=2e-6)   # critic loss
    np.testing.assert_allclose(scal[:, 2], ref[:, 4], rtol=1e-4, atol=2e-7)   # mirror loss
    np.testing.assert_allclose(scal[:, 3], ref[:, 3], rtol=1e-3, atol=2e-6)   # approx KL
    np.testing.assert_allclose(scal[:, 4], ref[:, 6], rtol=0, atol=1e-6)      # clip fraction
    t = k.get_tensors()
    for i, n in enumerate(NAMES):
        # two Adam steps of lr 3e-4: a wrong gradient SIGN moves a weight by 1.2e-3, so 3e-6 is a tight bar
        np.testing.assert_allclose(t[f"a_{n}"].numpy(), g[f"a1_{i}"], rtol=0, atol=3e-6, err_msg=f"actor {n}")
        np.testing.assert_allclose(t[f"c_{n}"].numpy(), g[f"c1_{i}"], rtol=0, atol=3e-6, err_msg=f"critic {n}")
    np.testing.assert_allclose(t["stds"].numpy(), g["stds1"], rtol=0, atol=3e-6)


def test_update_h256_matches_reference_fixture():
    """Real network size (2x256): weights regenerate from the torch seed through reference_init."""
    from learninghumanoidwalking_amd.ppo_kernels import reference_init
    g = np.load(os.path.join(G, "ppo_h256_mirror.npz"))
    k = _kernels(g, 256, True, False)
    w0 = reference_init(37, 12, 256, 0.223, generator_seed=int(g["torch_seed"]))
    k.set_tensors(w0)
    scal = _run_updates(k, g)
    ref = g["scalars"]
    np.testing.assert_allclose(scal[:, 0], ref[:, 0], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(scal[:, 1], ref[:, 2], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(scal[:, 2], ref[:, 4], rtol=1e-4, atol=2e-7)
    t = k.get_tensors()
    for i, n in enumerate(NAMES):
        for pre in ("a", "c"):
            w1 = t[f"{pre}_{n}"].numpy()
            np.testing.assert_allclose(w1.reshape(-1)[:64], g[f"{pre}1_head_{i}"], rtol=0, atol=3e-6)
            dn = np.linalg.norm((w1 - w0[f"{pre}_{n}"].numpy()).astype(np.float64))
            np.testing.assert_allclose(dn, float(g[f"{pre}1_delta_norm_{i}"]), rtol=2e-3, atol=1e-7)


def test_forward_matches_oracle():
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
    from oracle import ppo_oracle as po
    k = PpoKernels(37, 12, hidden=256, max_rows=1000)
    w = reference_init(37, 12, 256, 0.223, generator_seed=5)
    w["a_b1"] += 0.1; w["c_b3"] += 0.3  # exercise the biases
    k.set_tensors(w)
    rs = np.random.default_rng(0)
    mean, std = rs.normal(size=37).astype(np.float32), (0.5 + rs.uniform(size=37)).astype(np.float32)
    k.set_obs_norm(mean, std)
    obs = torch.tensor(rs.normal(size=(777, 37)).astype(np.float32))
    mu, act, logp, val = k.forward(obs.cuda(), deterministic=True)
    xn = (obs - torch.tensor(mean)) / torch.tensor(std)
    mu_ref = po.mlp(xn, *[w[f"a_{n}"] for n in NAMES])
    v_ref = po.mlp(xn, *[w[f"c_{n}"] for n in NAMES])
    np.testing.assert_allclose(mu.cpu().numpy(), mu_ref.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(val.cpu().numpy(), v_ref.numpy()[:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(act.cpu().numpy(), mu.cpu().numpy())
    # stochastic: log-prob of the sampled action is consistent, noise is ~N(0, std)
    mu2, act2, logp2, _ = k.forward(obs.cuda(), seed=3, counter=7, deterministic=False)
    z = (act2 - mu2).cpu().numpy() / 0.223
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    lp_ref = torch.distributions.Normal(mu2.cpu(), 0.223 * torch.ones(12)).log_prob(act2.cpu()).sum(-1)
    np.testing.assert_allclose(logp2.cpu().numpy(), lp_ref.numpy(), rtol=1e-4, atol=1e-4)
    # counter-based: same key -> same noise; different counter -> different noise
    _, act3, _, _ = k.forward(obs.cuda(), seed=3, counter=7, deterministic=False)
    _, act4, _, _ = k.forward(obs.cuda(), seed=3, counter=8, deterministic=False)
    assert torch.equal(act2, act3) and not torch.equal(act2, act4)


def test_gae_matches_oracle_and_toy():
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels
    from oracle import ppo_oracle as po
    k = PpoKernels(5, 1, hidden=64, max_rows=64)
    rs = np.random.default_rng(3)
    T, N = 50, 33
    rew = rs.normal(size=(T, N)).astype(np.float32)
    val = rs.normal(size=(T, N)).astype(np.float32)
    done = (rs.uniform(size=(T, N)) < 0.08).astype(np.uint8) * rs.integers(1, 4, size=(T, N)).astype(np.uint8)
    vterm = rs.normal(size=(T, N)).astype(np.float32)
    vfinal = rs.normal(size=N).astype(np.float32)
    c = lambda a: torch.tensor(a).cuda()
    ret, adv = k.gae(c(rew), c(val), c(done), c(vterm), c(vfinal), 0.99, 0.95)
    ref = po.gae_batch(rew, val, done, vterm, vfinal, 0.99, 0.95)
    np.testing.assert_allclose(ret.cpu().numpy(), ref, rtol=0, atol=1e-6)   # float64 scan, float32 store
    np.testing.assert_allclose(adv.cpu().numpy(), ref - val, rtol=0, atol=1e-6)
    # SURVEY.md 8c known answer
    one = lambda v, n: torch.full((n, 1), v, dtype=torch.float32, device="cuda")
    r2, _ = k.gae(one(1.0, 5), one(0.5, 5), torch.zeros(5, 1, dtype=torch.uint8, device="cuda"), one(0.0, 5), one(0.25, 1).view(1), 0.99, 0.95)
    np.testing.assert_allclose(r2.cpu().numpy()[:, 0], [4.723518165117246, 3.9327678523309375, 3.091991336875, 2.19802375, 1.2475], rtol=0, atol=5e-7)


def test_large_minibatch_against_oracle():
    """B = 5000 rows (several 64-row tiles, split-K chunks, ragged edges) vs the CPU oracle."""
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
    from oracle import ppo_oracle as po
    B, D, A, H = 5000, 37, 12, 256
    mo, ma = po.mirror_tables(MIR_OBS, [29, 30]), po.mirror_tables(MIR_ACT)
    k = PpoKernels(D, A, hidden=H, max_rows=B, mirror_obs=mo, mirror_act=ma)
    w = reference_init(D, A, H, 0.223, generator_seed=21)
    k.set_tensors(w)
    rs = np.random.default_rng(4)
    mean, std = rs.normal(size=D).astype(np.float32) * 0.1, (0.5 + rs.uniform(size=D)).astype(np.float32)
    k.set_obs_norm(mean, std)
    orc = po.OraclePPO([w[f"a_{n}"] for n in NAMES], [w[f"c_{n}"] for n in NAMES], w["stds"], mean, std, mirror_obs=mo, mirror_act=ma)
    R = 6000
    obs = torch.tensor(rs.normal(size=(R, D)).astype(np.float32))
    with torch.no_grad():
        mu = orc.mu(obs)
    act = mu + 0.3 * torch.tensor(rs.normal(size=(R, A)).astype(np.float32))
    with torch.no_grad():
        old_logp = orc.log_prob(obs, act) + 0.05 * torch.tensor(rs.normal(size=(R, 1)).astype(np.float32))
    ret = torch.tensor(rs.normal(size=(R, 1)).astype(np.float32))
    adv = torch.tensor(rs.normal(size=(R, 1)).astype(np.float32))
    idx = torch.tensor(rs.permutation(R)[:B].astype(np.int32))
    li = idx.long()
    res = orc.update(obs[li], act[li], ret[li], adv[li], old_logp[li])
    xn, xm = k.normalize(obs.cuda())
    k.stats.zero_()
    k.grad_minibatch(xn, xm, act.cuda(), old_logp.view(-1).cuda(), adv.view(-1).cuda(), ret.view(-1).cuda(), idx.cuda())
    k.apply()
    s = k.stats.cpu().numpy()
    np.testing.assert_allclose([s[0], s[1], s[2], s[3], s[4]], [res[0], res[2], res[4], res[3], res[6]], rtol=2e-4, atol=2e-6)
    t = k.get_tensors()
    for i, n in enumerate(NAMES):
        np.testing.assert_allclose(t[f"a_{n}"].numpy(), orc.actor[i].detach().numpy(), rtol=0, atol=2e-6, err_msg=f"actor {n}")
        np.testing.assert_allclose(t[f"c_{n}"].numpy(), orc.critic[i].detach().numpy(), rtol=0, atol=2e-6, err_msg=f"critic {n}")


def test_update_is_bitwise_deterministic():
    """Same inputs -> bitwise identical parameters after several optimiser steps (no floating-point atomics anywhere in
    the update: split-K slices, bias column sums and loss scalars are reduced in a fixed order).  This is the property the
    reference's tests/test_determinism.py:79-146 asserts with torch.equal on CPU."""
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
    from oracle import ppo_oracle as po
    B, D, A, H = 3000, 37, 12, 256
    mo, ma = po.mirror_tables(MIR_OBS, [29, 30]), po.mirror_tables(MIR_ACT)
    rs = np.random.default_rng(9)
    obs = torch.tensor(rs.normal(size=(4000, D)).astype(np.float32)).cuda()
    act = torch.tensor(rs.normal(size=(4000, A)).astype(np.float32) * 0.3).cuda()
    logp = torch.tensor(rs.normal(size=4000).astype(np.float32) - 8).cuda()
    adv = torch.tensor(rs.normal(size=4000).astype(np.float32)).cuda()
    ret = torch.tensor(rs.normal(size=4000).astype(np.float32)).cuda()
    idx = torch.tensor(rs.permutation(4000)[:B].astype(np.int32)).cuda()
    outs = []
    for rep in range(3):
        k = PpoKernels(D, A, hidden=H, max_rows=B, learn_std=True, entropy_coeff=0.01, mirror_obs=mo, mirror_act=ma)
        k.set_tensors(reference_init(D, A, H, 0.223, generator_seed=21))
        xn, xm = k.normalize(obs)
        for _ in range(4):
            k.grad_minibatch(xn, xm, act, logp, adv, ret, idx)
            k.apply()
        outs.append((k.theta.clone(), k.stats.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])
    assert torch.equal(outs[0][1][:5], outs[1][1][:5])


def _imit_inputs(g, u, coeff=0.3):
    """Dense imitation target / mask of update u of the imitation fixture, built with the oracle MLP as the expert."""
    from oracle import ppo_oracle as po
    from learninghumanoidwalking_amd.imitation import ImitationQuery, dense_imitation_target
    obs = torch.tensor(g[f"obs_{u}"])
    smask = obs[:, 0] > 0
    expert = [torch.tensor(g[f"e_{k}"]) for k in range(6)]
    target = po.mlp(obs[smask][:, :20], *expert)
    q = ImitationQuery(expert_obs=obs[smask][:, :20], sample_mask=smask, action_indices=torch.tensor([0, 2, 5]))
    dense, mask, count = dense_imitation_target(q, target.cuda(), obs.shape[0], 12)
    return (coeff, dense, mask, count), (smask, target)


def test_update_with_imitation_matches_reference_fixture():
    g = np.load(os.path.join(G, "ppo_h64_imitate.npz"))
    k = _kernels(g, 64, True, False)
    k.set_tensors({f"a_{n}": g[f"a0_{i}"] for i, n in enumerate(NAMES)})
    k.set_tensors({f"c_{n}": g[f"c0_{i}"] for i, n in enumerate(NAMES)})
    k.set_tensors({"stds": g["stds0"]})
    for u in range(len(g["scalars"])):
        c = lambda n: torch.tensor(g[f"{n}_{u}"]).cuda().contiguous()
        obs = c("obs")
        xn, xm = k.normalize(obs)
        k.stats.zero_()
        idx = torch.arange(obs.shape[0], dtype=torch.int32, device="cuda")
        imit, _ = _imit_inputs(g, u)
        k.grad_minibatch(xn, xm, c("act"), c("old_logp").view(-1), c("adv").view(-1), c("ret").view(-1), idx, imitation=imit)
        k.apply()
        s = k.stats.cpu().numpy()
        ref = g["scalars"][u]
        np.testing.assert_allclose(s[5], ref[5], rtol=1e-4, atol=1e-9)            # imitation loss
        np.testing.assert_allclose([s[0], s[1], s[2]], [ref[0], ref[2], ref[4]], rtol=1e-4, atol=2e-6)
    t = k.get_tensors()
    for i, n in enumerate(NAMES):
        np.testing.assert_allclose(t[f"a_{n}"].numpy(), g[f"a1_{i}"], rtol=0, atol=3e-6, err_msg=f"actor {n}")


def test_imitation_gradient_against_oracle_with_a_dominant_coefficient():
    """With coefficient 200 the imitation term dominates the actor gradient: post-Adam actor weights must follow the
    oracle's autograd (a missing / mis-scaled term would show at the 1e-4 level, the bar is 5e-6)."""
    from oracle import ppo_oracle as po
    g = np.load(os.path.join(G, "ppo_h64_imitate.npz"))
    k = _kernels(g, 64, True, False)
    k.set_tensors({f"a_{n}": g[f"a0_{i}"] for i, n in enumerate(NAMES)})
    k.set_tensors({f"c_{n}": g[f"c0_{i}"] for i, n in enumerate(NAMES)})
    k.set_tensors({"stds": g["stds0"]})
    mo, ma = po.mirror_tables(MIR_OBS, [29, 30]), po.mirror_tables(MIR_ACT)
    orc = po.OraclePPO([g[f"a0_{i}"] for i in range(6)], [g[f"c0_{i}"] for i in range(6)], g["stds0"], g["obs_mean"], g["obs_std"],
                       mirror_obs=mo, mirror_act=ma)
    for u in range(2):
        c = lambda n: torch.tensor(g[f"{n}_{u}"]).cuda().contiguous()
        obs = c("obs")
        xn, xm = k.normalize(obs)
        k.stats.zero_()
        idx = torch.arange(obs.shape[0], dtype=torch.int32, device="cuda")
        imit, (smask, target) = _imit_inputs(g, u, coeff=200.0)
        k.grad_minibatch(xn, xm, c("act"), c("old_logp").view(-1), c("adv").view(-1), c("ret").view(-1), idx, imitation=imit)
        k.apply()
        tt = lambda n: torch.tensor(g[f"{n}_{u}"])
        res = orc.update(tt("obs"), tt("act"), tt("ret"), tt("adv"), tt("old_logp"), imit=(200.0, smask, torch.tensor([0, 2, 5]), target))
        np.testing.assert_allclose(k.stats.cpu().numpy()[5], res[5], rtol=1e-4)
    t = k.get_tensors()
    for i, n in enumerate(NAMES):
        np.testing.assert_allclose(t[f"a_{n}"].numpy(), orc.actor[i].detach().numpy(), rtol=0, atol=5e-6, err_msg=f"actor {n}")
    # the term did matter: weights differ from the no-imitation fixture by much more than the tolerance
    g0 = np.load(os.path.join(G, "ppo_h64_imitate.npz"))
    assert np.abs(t["a_w3"].numpy() - g0["a1_4"]).max() > 1e-4


def test_fp16_inference_matches_half_rounded_reference():
    """lhw_ppo_set_inference_dtype: rollout forward with fp16 operands on the fp16 MFMA == a torch computation whose weights
    and layer inputs are rounded to fp16 (float32 accumulation); the float32 path is unchanged and close."""
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
    D, A, H, N = 35, 10, 256, 1000
    k = PpoKernels(D, A, hidden=H, max_rows=1024)
    t = reference_init(D, A, H, 0.223, generator_seed=2)
    for n in ("a_b1", "a_b2", "a_b3", "c_b1", "c_b2", "c_b3"):
        t[n] = t[n] + torch.randn_like(t[n]) * 0.05
    k.set_tensors(t)
    om, osd = torch.randn(D) * 0.1, torch.rand(D) + 0.5
    k.set_obs_norm(om.numpy(), osd.numpy())
    obs = torch.randn(N, D)
    mu32, _, _, v32 = k.forward(obs.cuda(), deterministic=True)
    mu32, v32 = mu32.cpu().clone(), v32.cpu().clone()
    k.set_inference_fp16(True)
    mu16, _, _, v16 = k.forward(obs.cuda(), deterministic=True)
    mu16, v16 = mu16.cpu(), v16.cpu()
    h = lambda x: x.half().float()

    def net(p):
        x = (obs - om) / osd
        x = torch.relu(h(x) @ h(t[f"{p}_w1"]).t() + t[f"{p}_b1"])
        x = torch.relu(h(x) @ h(t[f"{p}_w2"]).t() + t[f"{p}_b2"])
        return h(x) @ h(t[f"{p}_w3"]).t() + t[f"{p}_b3"]
    # float32 accumulation order differs from torch's, so a few hidden activations land on the neighbouring fp16 value
    # (2^-11 relative) when they are rounded for the next layer: tolerance of a few fp16 ulps, far below the fp16-vs-fp32 gap
    np.testing.assert_allclose(mu16.numpy(), net("a").numpy(), rtol=3e-3, atol=1e-4)
    np.testing.assert_allclose(v16.numpy(), net("c").numpy()[:, 0], rtol=3e-3, atol=1e-3)
    assert float(np.abs(mu16.numpy() - net("a").numpy()).mean()) < 2e-5
    assert 1e-7 < float((mu16 - mu32).abs().max()) < 5e-3           # really a different precision, and still close
    k.set_inference_fp16(False)
    mu_again = k.forward(obs.cuda(), deterministic=True)[0].cpu()
    assert torch.equal(mu_again, mu32)


def test_fp16_update_matches_half_rounded_reference():
    """lhw_ppo_set_update_dtype (BASELINE config 5): every GEMM of the update rounds both operands to fp16 and accumulates in
    float32 -- forward, activation gradients, weight gradients (bias gradients are sums of the rounded output gradients);
    loss, master weights and Adam are float32.  Pinned to a torch computation with exactly that rounding (a custom autograd
    linear layer under the oracle's update); the float32 path is a different, nearby result."""
    from oracle import ppo_oracle as po
    g = np.load(os.path.join(G, "ppo_h64_mirror.npz"))
    h = lambda x: x.half().float()

    class HalfLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, W, b):
            ctx.save_for_backward(x, W)
            return h(x) @ h(W).t() + b

        @staticmethod
        def backward(ctx, gy):
            x, W = ctx.saved_tensors
            gh = h(gy)
            return gh @ h(W), gh.t() @ h(x), gh.sum(0)

    def half_mlp(x, W1, b1, W2, b2, W3, b3):
        x = torch.relu(HalfLinear.apply(x, W1, b1))
        x = torch.relu(HalfLinear.apply(x, W2, b2))
        return HalfLinear.apply(x, W3, b3)

    def run(fp16):
        k = _kernels(g, 64, True, False)
        k.set_tensors({f"a_{n}": g[f"a0_{i}"] for i, n in enumerate(NAMES)})
        k.set_tensors({f"c_{n}": g[f"c0_{i}"] for i, n in enumerate(NAMES)})
        k.set_tensors({"stds": g["stds0"]})
        k.set_update_fp16(fp16)
        scal = _run_updates(k, g)
        return scal, k.get_tensors()

    s16, t16 = run(True)
    s32, t32 = run(False)
    orc = po.OraclePPO([g[f"a0_{i}"] for i in range(6)], [g[f"c0_{i}"] for i in range(6)], g["stds0"], g["obs_mean"], g["obs_std"],
                       mirror_obs=po.mirror_tables(MIR_OBS, [29, 30]), mirror_act=po.mirror_tables(MIR_ACT))
    saved = po.mlp
    po.mlp = half_mlp
    try:
        ref = []
        for u in range(len(g["scalars"])):
            c = lambda n: torch.tensor(g[f"{n}_{u}"])
            ref.append(orc.update(c("obs"), c("act"), c("ret"), c("adv"), c("old_logp")))
    finally:
        po.mlp = saved
    ref = np.array(ref)
    np.testing.assert_allclose(s16[:, 0], ref[:, 0], rtol=2e-3, atol=2e-5)   # actor loss
    np.testing.assert_allclose(s16[:, 1], ref[:, 2], rtol=2e-3, atol=2e-5)   # critic loss
    np.testing.assert_allclose(s16[:, 2], ref[:, 4], rtol=5e-3, atol=2e-6)   # mirror loss
    worst16 = worst32 = 0.0
    for i, n in enumerate(NAMES):
        for net, params in (("a", orc.actor), ("c", orc.critic)):
            want = params[i].detach().numpy()
            worst16 = max(worst16, float(np.abs(t16[f"{net}_{n}"].numpy() - want).max()))
            worst32 = max(worst32, float(np.abs(t32[f"{net}_{n}"].numpy() - want).max()))
    # two Adam steps of lr 3e-4: a wrong gradient sign moves a weight by 1.2e-3.  Summation order inside a GEMM flips a few
    # fp16 roundings of the next layer's inputs, hence looser than the float32 fixture test (3e-6), far tighter than a wrong term
    assert worst16 < 6e-5, worst16
    assert worst32 > 1e-7, "the float32 update should differ from the half-rounded reference"
    print(f"fp16 update vs half-rounded torch reference: worst |dw| {worst16:.2e} (float32 update vs the same reference {worst32:.2e})")


def test_fp16_update_gradients_survive_large_minibatches():
    """d loss / d output carries 1 / B, so at B = 32768 the back-propagated gradients of the first layers sit in the
    fp16 subnormal range once the fp16 update rounds them per GEMM.  With the power-of-two loss scaling of lhw_ppo_grad the
    fp16 update's gradient stays within fp16 rounding of the float32 one at the benchmark's minibatch size."""
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
    D, A, H, B = 37, 12, 256, 32768
    gen = torch.Generator().manual_seed(3)
    obs = torch.randn(B, D, generator=gen).cuda()
    adv = torch.randn(B, generator=gen).cuda()
    ret = torch.randn(B, generator=gen).cuda()
    idx = torch.arange(B, dtype=torch.int32, device="cuda")
    grads = {}
    for half in (False, True):
        k = PpoKernels(D, A, hidden=H, max_rows=B)
        k.set_tensors(reference_init(D, A, H, 0.223, generator_seed=0))     # actor read-out x 0.01: the early-training regime
        k.set_update_fp16(half)
        mu, act, logp, val = k.forward(obs, seed=1)
        xn, _ = k.normalize(obs, want_mirror=False)
        k.grad_minibatch(xn, None, act.clone(), logp.clone(), adv, ret, idx)
        torch.cuda.synchronize()
        grads[half] = {n: k._view(k.grad, n).clone().cpu() for n in ("a_w1", "a_w2", "a_w3", "c_w1", "c_w2", "c_w3")}
    for n, g32 in grads[False].items():
        g16 = grads[True][n]
        rel = float((g16 - g32).norm() / g32.norm())
        assert rel < 2e-2, f"{n}: fp16-update gradient off by {rel:.3f} (relative) at B = {B}"
