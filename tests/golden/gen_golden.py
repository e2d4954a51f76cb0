"""Generates the golden fixtures in this directory by importing the REFERENCE's own classes from
/root/reference (only possible in the build container; the fixtures are committed so the tests
run anywhere).  Run: python tests/golden/gen_golden.py

What can be imported: rl.storage.rollout_storage (GAE), rl.policies.*,
rl.envs.wrappers, rl.algos.ppo.PPO once `ray` and `torch.utils.tensorboard` are stubbed,
tasks/rewards.py by file path.  The physics (MuJoCo) cannot be imported: no physics golden exists.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    ray = types.ModuleType("ray")
    ray.remote = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda c: c))
    ray.put = lambda x: x
    ray.get = lambda x: x
    ray.is_initialized = lambda: True
    sys.modules["ray"] = ray
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return lambda *a, **k: None

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    for name in ("imageio", "mujoco", "mujoco.viewer", "transforms3d"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


def gen_gae():
    from rl.storage.rollout_storage import PPOBuffer
    rs = np.random.default_rng(0)
    cases = {}
    # known-answer from SURVEY.md 8c
    buf = PPOBuffer(1, 1, gamma=0.99, lam=0.95, size=5)
    for _ in range(5):
        buf.store(torch.zeros(1), torch.zeros(1), torch.tensor(1.0), torch.tensor(0.5), False)
    buf.finish_path(last_val=torch.tensor(0.25))
    cases["toy_returns"] = buf.returns[:, 0].numpy().copy()
    # random multi-trajectory buffer
    T = 64
    rew = rs.normal(size=T).astype(np.float32)
    val = rs.normal(size=T).astype(np.float32)
    ends = [9, 30, 31, 63]
    lasts = rs.normal(size=len(ends)).astype(np.float32)
    buf = PPOBuffer(1, 1, gamma=0.99, lam=0.95, size=T)
    k = 0
    for t in range(T):
        buf.store(torch.zeros(1), torch.zeros(1), torch.tensor(rew[t]), torch.tensor(val[t]), t in ends)
        if t in ends:
            buf.finish_path(last_val=torch.tensor(lasts[k]))
            k += 1
    cases.update(rew=rew, val=val, ends=np.array(ends), lasts=lasts, returns=buf.returns[:, 0].numpy().copy())
    np.savez(os.path.join(OUT, "gae.npz"), **cases)


def gen_clock():
    spec = importlib.util.spec_from_file_location("ref_rewards", os.path.join(REF, "tasks", "rewards.py"))
    rw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rw)
    out = {}
    for tag, (sw, st, freq, period) in {"jvrc": (0.75, 0.35, 40, 88), "h1": (0.4, 0.1, 40, 40)}.items():
        right, left = rw.create_phase_reward(sw, st, 0.1, "grounded", freq)
        ph = np.arange(period)
        out[f"{tag}_lut"] = np.stack([right[0](ph), right[1](ph), left[0](ph), left[1](ph)])
    # scalar reward terms on fixed inputs
    rs = np.random.default_rng(1)
    qvel, qacc = rs.normal(size=18), rs.normal(size=18)
    tq, ptq = rs.normal(size=12) * 20, rs.normal(size=12) * 20
    a, pa = rs.normal(size=12), rs.normal(size=12)
    out["inputs"] = np.concatenate([qvel, qacc, tq, ptq, a, pa])
    out["terms"] = np.array([
        rw.calc_fwd_vel_reward(np.array([0.3, -0.1]), np.array([0.2, 0.0])),
        rw.calc_yaw_vel_reward(0.37, 0.1),
        rw.calc_action_reward(a, pa),
        rw.calc_torque_reward(tq, ptq),
        rw.calc_height_reward(0.77, 0.8, 0.2, 0.005),
        rw.calc_height_reward(0.795, 0.8, 0.0, 0.0),
        rw.calc_root_accel_reward(qvel, qacc),
        rw.calc_foot_frc_clock_reward(120.0, 500.0, 10, lambda p: -1.0, lambda p: 0.5, 62.0),
        rw.calc_foot_vel_clock_reward(np.array([0.1, 0.0, 0.05]), np.array([0.3, 0.1, 0.0]), 3, lambda p: 1.0, lambda p: -0.25),
    ])
    np.savez(os.path.join(OUT, "rewards.npz"), **out)


JVRC_MIR_OBS = [-0.1, 1, -2, 3, -4, 11, -12, -13, 14, -15, 16, 5, -6, -7, 8, -9, 10, 23, -24, -25, 26, -27, 28, 17, -18, -19,
                20, -21, 22] + list(range(29, 37))   # reference envs/jvrc/jvrc_base.py:73-110, 8 external obs
JVRC_MIR_ACT = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]
JVRC_CLOCK = [29, 30]


def gen_ppo(tag, hidden, B, mirror, learn_std, n_updates, seed):
    from argparse import Namespace
    from rl.algos.ppo import PPO
    from rl.envs.wrappers import SymmetricEnv
    from rl.policies.actor import Gaussian_FF_Actor
    from rl.policies.critic import FF_V
    torch.manual_seed(seed)
    D, A = 37, 12
    policy = Gaussian_FF_Actor(D, A, layers=(hidden, hidden), init_std=0.223, learn_std=learn_std, bounded=False)
    critic = FF_V(D, layers=(hidden, hidden))
    rs = np.random.default_rng(seed)
    obs_mean = rs.normal(size=D).astype(np.float32) * 0.1
    obs_std = (0.5 + rs.uniform(size=D)).astype(np.float32)
    policy.obs_mean = torch.tensor(obs_mean); policy.obs_std = torch.tensor(obs_std)
    critic.obs_mean = policy.obs_mean; critic.obs_std = policy.obs_std
    ppo = PPO.__new__(PPO)
    ppo.policy, ppo.critic = policy, critic
    import copy
    ppo.old_policy = copy.deepcopy(policy)
    ppo.clip, ppo.ent_coeff, ppo.mirror_coeff, ppo.imitate_coeff, ppo.grad_clip = 0.2, 0.01 if learn_std else 0.0, 0.4, 0.3, 0.5
    ppo.recurrent, ppo.imitation_projector, ppo.base_policy = False, None, None
    ppo.actor_optimizer = torch.optim.Adam(policy.parameters(), lr=3e-4, eps=1e-5)
    ppo.critic_optimizer = torch.optim.Adam(critic.parameters(), lr=3e-4, eps=1e-5)
    sym = SymmetricEnv.__new__(SymmetricEnv)
    sym.act_mirror_matrix = torch.tensor(__import__("rl.envs.wrappers", fromlist=["x"])._get_symmetry_matrix(JVRC_MIR_ACT), dtype=torch.float32)
    sym.obs_mirror_matrix = torch.tensor(__import__("rl.envs.wrappers", fromlist=["x"])._get_symmetry_matrix(JVRC_MIR_OBS), dtype=torch.float32)
    sym.clock_inds = JVRC_CLOCK
    sym.env = types.SimpleNamespace(base_obs_len=D)

    def params(net):
        return [p.detach().numpy().copy() for p in net.parameters() if p.dim() > 0 and p.shape != (A,) or p.dim() == 2] 

    def weights(actor, critic):
        a = [actor.actor_layers[0].weight, actor.actor_layers[0].bias, actor.actor_layers[1].weight, actor.actor_layers[1].bias,
             actor.means.weight, actor.means.bias]
        c = [critic.critic_layers[0].weight, critic.critic_layers[0].bias, critic.critic_layers[1].weight,
             critic.critic_layers[1].bias, critic.network_out.weight, critic.network_out.bias]
        return [x.detach().numpy().copy() for x in a], [x.detach().numpy().copy() for x in c]

    out = dict(obs_mean=obs_mean, obs_std=obs_std, hidden=hidden, learn_std=int(learn_std), mirror=int(mirror))
    a0, c0 = weights(policy, critic)
    for k, w in enumerate(a0):
        out[f"a0_{k}"] = w
    for k, w in enumerate(c0):
        out[f"c0_{k}"] = w
    out["stds0"] = policy.stds.detach().numpy().copy()
    scal = []
    for u in range(n_updates):
        obs = torch.tensor(rs.normal(size=(B, D)).astype(np.float32))
        obs[:, 29:31] = torch.tensor(np.stack([np.sin(rs.uniform(0, 6.28, B)), np.cos(rs.uniform(0, 6.28, B))], 1).astype(np.float32) * 0.99)
        with torch.no_grad():
            mu = ppo.old_policy(obs)
        act = mu + 0.223 * torch.tensor(rs.normal(size=(B, A)).astype(np.float32)) * (1.5 if u else 1.0)
        ret = torch.tensor(rs.normal(size=(B, 1)).astype(np.float32))
        adv = torch.tensor(rs.normal(size=(B, 1)).astype(np.float32))
        with torch.no_grad():
            old_logp = ppo.old_policy.distribution(obs).log_prob(act).sum(-1, keepdim=True)
        res = ppo.update_actor_critic(obs, act, ret, adv, 1,
                                      mirror_observation=sym.mirror_clock_observation if mirror else None,
                                      mirror_action=sym.mirror_action if mirror else None)
        scal.append([float(x) for x in res])
        out[f"obs_{u}"], out[f"act_{u}"], out[f"ret_{u}"], out[f"adv_{u}"] = obs.numpy(), act.numpy(), ret.numpy(), adv.numpy()
        out[f"old_logp_{u}"] = old_logp.numpy()
    a1, c1 = weights(policy, critic)
    if hidden <= 64:
        for k, w in enumerate(a1):
            out[f"a1_{k}"] = w
        for k, w in enumerate(c1):
            out[f"c1_{k}"] = w
    else:  # keep the fixture small: per-tensor checks instead of full tensors
        for k, (w0, w1) in enumerate(zip(a0, a1)):
            out[f"a1_delta_norm_{k}"] = np.float64(np.linalg.norm((w1 - w0).astype(np.float64)))
            out[f"a1_head_{k}"] = w1.reshape(-1)[:64].copy()
        for k, (w0, w1) in enumerate(zip(c0, c1)):
            out[f"c1_delta_norm_{k}"] = np.float64(np.linalg.norm((w1 - w0).astype(np.float64)))
            out[f"c1_head_{k}"] = w1.reshape(-1)[:64].copy()
        out["torch_seed"] = seed  # weights regenerate from the seed through the reference-init path
        for k in range(6):
            del out[f"a0_{k}"], out[f"c0_{k}"]
    out["stds1"] = policy.stds.detach().numpy().copy()
    out["scalars"] = np.array(scal)
    np.savez_compressed(os.path.join(OUT, f"ppo_{tag}.npz"), **out)


def gen_misc():
    from rl.utils.seeding import get_worker_seed
    from rl.envs.wrappers import _get_symmetry_matrix
    np.savez(os.path.join(OUT, "misc.npz"),
             worker_seeds=np.array([get_worker_seed(0, 0), get_worker_seed(7, 3), get_worker_seed(123456, 11, 1)], dtype=np.int64),
             mir_obs=_get_symmetry_matrix(JVRC_MIR_OBS), mir_act=_get_symmetry_matrix(JVRC_MIR_ACT))


if __name__ == "__main__":
    sys.path.insert(0, REF)
    _stub_modules()
    gen_gae()
    gen_clock()
    gen_misc()
    gen_ppo("h64_mirror", 64, 96, True, False, 2, 11)
    gen_ppo("h64_learnstd", 64, 70, False, True, 2, 12)
    gen_ppo("h256_mirror", 256, 128, True, False, 1, 13)
    print("golden fixtures written to", OUT)
