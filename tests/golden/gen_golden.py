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


def gen_ppo(tag, hidden, B, mirror, learn_std, n_updates, seed, imitate=False):
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
    base = None
    if imitate:   # rl/algos/ppo.py:360-368 with an env-style projector (rl/algos/imitation.py contract)
        from rl.algos.imitation import ImitationQuery
        base = Gaussian_FF_Actor(20, 3, layers=(32, 32), init_std=0.1, learn_std=False, bounded=False)
        base.obs_mean, base.obs_std = torch.zeros(20), torch.ones(20)
        base.eval()

        class Proj:
            def __call__(self, obs_batch):
                mask = obs_batch[:, 0] > 0
                return ImitationQuery(expert_obs=obs_batch[mask][:, :20], sample_mask=mask, action_indices=torch.tensor([0, 2, 5]))

        ppo.imitation_projector, ppo.base_policy = Proj(), base
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

    out = dict(obs_mean=obs_mean, obs_std=obs_std, hidden=hidden, learn_std=int(learn_std), mirror=int(mirror), imitate=int(imitate))
    if imitate:
        eb = [base.actor_layers[0].weight, base.actor_layers[0].bias, base.actor_layers[1].weight, base.actor_layers[1].bias,
              base.means.weight, base.means.bias]
        for kk, w in enumerate(eb):
            out[f"e_{kk}"] = w.detach().numpy().copy()
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


# reference envs/h1/h1_walk.py:68-113 (35 robot-state entries + 8 external), actions left(5) / right(5)
H1W_MIR_OBS = [-0.1, 1, -2, 3, -4, -10, -11, 12, 13, 14, -5, -6, 7, 8, 9, -20, -21, 22, 23, 24, -15, -16, 17, 18, 19,
               -30, -31, 32, 33, 34, -25, -26, 27, 28, 29] + list(range(35, 43))
H1W_MIR_ACT = [-5, -6, 7, 8, 9, -0.1, -1, 2, 3, 4]


def gen_misc():
    from rl.utils.seeding import get_worker_seed
    from rl.envs.wrappers import _get_symmetry_matrix
    np.savez(os.path.join(OUT, "misc.npz"),
             worker_seeds=np.array([get_worker_seed(0, 0), get_worker_seed(7, 3), get_worker_seed(123456, 11, 1)], dtype=np.int64),
             mir_obs=_get_symmetry_matrix(JVRC_MIR_OBS), mir_act=_get_symmetry_matrix(JVRC_MIR_ACT),
             mir_obs_h1walk=_get_symmetry_matrix(H1W_MIR_OBS), mir_act_h1walk=_get_symmetry_matrix(H1W_MIR_ACT),
             mir_obs_step=_get_symmetry_matrix(JVRC_MIR_OBS[:29] + list(range(29, 39))))


def gen_rppo(tag, mirror, lengths, seed, n_updates=2, H=32):
    """Reference recurrent branch: Gaussian_LSTM_Actor / LSTM_V, PPO.update_actor_critic on a padded list of trajectories
    with the mask of rl/algos/ppo.py:512-533.  With mirror=True all trajectories have equal length (the reference's mirror
    term averages over padded positions too -- a quirk the padding-free device formulation has no counterpart for)."""
    import copy
    from torch.nn.utils.rnn import pad_sequence
    from rl.algos.ppo import PPO
    from rl.envs.wrappers import SymmetricEnv, _get_symmetry_matrix
    from rl.policies.actor import Gaussian_LSTM_Actor
    from rl.policies.critic import LSTM_V
    torch.manual_seed(seed)
    D, A = 37, 12
    policy = Gaussian_LSTM_Actor(D, A, layers=(H, H), init_std=0.223, learn_std=False)
    critic = LSTM_V(D, layers=(H, H))
    rs = np.random.default_rng(seed)
    obs_mean = rs.normal(size=D).astype(np.float32) * 0.1
    obs_std = (0.5 + rs.uniform(size=D)).astype(np.float32)
    policy.obs_mean = torch.tensor(obs_mean); policy.obs_std = torch.tensor(obs_std)
    critic.obs_mean = policy.obs_mean; critic.obs_std = policy.obs_std
    ppo = PPO.__new__(PPO)
    ppo.policy, ppo.critic = policy, critic
    ppo.old_policy = copy.deepcopy(policy)
    ppo.clip, ppo.ent_coeff, ppo.mirror_coeff, ppo.imitate_coeff, ppo.grad_clip = 0.2, 0.0, 0.4, 0.3, 0.5
    ppo.recurrent, ppo.imitation_projector, ppo.base_policy = True, None, None
    ppo.actor_optimizer = torch.optim.Adam(policy.parameters(), lr=3e-4, eps=1e-5)
    ppo.critic_optimizer = torch.optim.Adam(critic.parameters(), lr=3e-4, eps=1e-5)
    sym = SymmetricEnv.__new__(SymmetricEnv)
    sym.act_mirror_matrix = torch.tensor(_get_symmetry_matrix(JVRC_MIR_ACT), dtype=torch.float32)
    sym.obs_mirror_matrix = torch.tensor(_get_symmetry_matrix(JVRC_MIR_OBS), dtype=torch.float32)
    sym.clock_inds = JVRC_CLOCK
    sym.env = types.SimpleNamespace(base_obs_len=D)

    def weights(net, layers, outl):
        w = []
        for cell in layers:
            w += [cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh]
        w += [outl.weight, outl.bias]
        return [x.detach().numpy().copy() for x in w]

    out = dict(obs_mean=obs_mean, obs_std=obs_std, hidden=H, mirror=int(mirror), lengths=np.array(lengths))
    for k, w in enumerate(weights(policy, policy.actor_layers, policy.network_out)):
        out[f"a0_{k}"] = w
    for k, w in enumerate(weights(critic, critic.critic_layers, critic.network_out)):
        out[f"c0_{k}"] = w
    out["stds0"] = policy.stds.detach().numpy().copy()
    scal = []
    for u in range(n_updates):
        obs_l, act_l, ret_l, adv_l = [], [], [], []
        for Lk in lengths:
            o = torch.tensor(rs.normal(size=(Lk, D)).astype(np.float32))
            ph = rs.uniform(0, 6.28, Lk)
            o[:, 29] = torch.tensor(np.sin(ph).astype(np.float32) * 0.99)
            o[:, 30] = torch.tensor(np.cos(ph).astype(np.float32) * 0.99)
            obs_l.append(o)
            ret_l.append(torch.tensor(rs.normal(size=(Lk, 1)).astype(np.float32)))
            adv_l.append(torch.tensor(rs.normal(size=(Lk, 1)).astype(np.float32)))
        with torch.no_grad():     # actions sampled around the old policy's means, trajectory by trajectory
            for o in obs_l:
                mu = ppo.old_policy(o.unsqueeze(1)).squeeze(1)
                act_l.append(mu + 0.223 * torch.tensor(rs.normal(size=mu.shape).astype(np.float32)) * (1.5 if u else 1.0))
        mask = [torch.ones_like(r) for r in ret_l]
        ob, ab, rb, vb, mb = (pad_sequence(x, batch_first=False) for x in (obs_l, act_l, ret_l, adv_l, mask))
        with torch.no_grad():     # what update_actor_critic recomputes internally with old_policy (ppo.py:305-306)
            olp = ppo.old_policy.distribution(ob).log_prob(ab).sum(-1, keepdim=True)
        out[f"old_logp_{u}"] = torch.cat([olp[:Lk, k] for k, Lk in enumerate(lengths)]).numpy()
        res = ppo.update_actor_critic(ob, ab, rb, vb, mb, mirror_observation=sym.mirror_clock_observation if mirror else None,
                                      mirror_action=sym.mirror_action if mirror else None)
        scal.append([float(x) for x in res])
        out[f"obs_{u}"], out[f"act_{u}"] = torch.cat(obs_l).numpy(), torch.cat(act_l).numpy()     # trajectories back to back
        out[f"ret_{u}"], out[f"adv_{u}"] = torch.cat(ret_l).numpy(), torch.cat(adv_l).numpy()
    for k, w in enumerate(weights(policy, policy.actor_layers, policy.network_out)):
        out[f"a1_{k}"] = w
    for k, w in enumerate(weights(critic, critic.critic_layers, critic.network_out)):
        out[f"c1_{k}"] = w
    out["scalars"] = np.array(scal)
    np.savez_compressed(os.path.join(OUT, f"rppo_{tag}.npz"), **out)


def gen_stepping():
    """Executes the reference's SteppingTask (tasks/stepping_task.py) against a scripted fake RobotInterface: reset for
    every walk mode / initial phase / curriculum iteration with the np.random draws forced to listed values, then a run
    of step() / calc_reward() / done() calls on scripted kinematic states.  Stored: inputs, forced draws and every output
    the oracle env (oracle/env_jvrc_step.py) must reproduce."""
    import random as pyrandom
    sys.path.insert(0, OUT)
    import _tf3_min
    _tf3_min.install(sys.modules)
    from tasks import stepping_task as st
    cwd = os.getcwd()

    class Obj:
        def __init__(self, **k):
            self.__dict__.update(k)

    class FakeModel:
        def __init__(self):
            self.bodies = {f"box{i + 1:02d}": Obj(pos=np.zeros(3), quat=np.zeros(4)) for i in range(20)}
            self.bodies["floor"] = Obj(pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]))
            self.geoms = {f"box{i + 1:02d}": Obj(size=np.array([1.0, 1.0, 0.1]), rgba=np.zeros(4)) for i in range(20)}

        def body(self, n):
            return self.bodies[n]

        def geom(self, n):
            return self.geoms[n]

    class FakeClient:
        def __init__(self):
            self.model = FakeModel()
            self.k = {}

        def get_robot_mass(self):
            return 16062.0

        def get_object_xpos_by_name(self, name, typ):
            return self.k["xpos"][name]

        def get_object_xquat_by_name(self, name, typ):
            return self.k["xquat"][name]

        def get_lfoot_body_pos(self):
            return self.k["xpos"]["lfoot"].copy()

        def get_rfoot_body_pos(self):
            return self.k["xpos"]["rfoot"].copy()

        def get_lfoot_body_vel(self, frame=0):
            return [self.k["lvel"], np.zeros(3)]

        def get_rfoot_body_vel(self, frame=0):
            return [self.k["rvel"], np.zeros(3)]

        def get_lfoot_grf(self):
            return self.k["lgrf"]

        def get_rfoot_grf(self):
            return self.k["rgrf"]

        def check_rfoot_floor_collision(self):
            return len(self.k["rcon"]) > 0

        def check_lfoot_floor_collision(self):
            return len(self.k["lcon"]) > 0

        def get_rfoot_floor_contacts(self):
            return [(i, Obj(pos=np.array(p))) for i, p in enumerate(self.k["rcon"])]

        def get_lfoot_floor_contacts(self):
            return [(i, Obj(pos=np.array(p))) for i, p in enumerate(self.k["lcon"])]

        def check_self_collisions(self):
            return self.k["selfcol"]

    rs = np.random.default_rng(42)
    cases = []
    real = dict(choice=np.random.choice, uniform=np.random.uniform, randint=np.random.randint, pychoice=pyrandom.choice)
    MODES = [st.WalkModes.CURVED, st.WalkModes.STANDING, st.WalkModes.BACKWARD, st.WalkModes.LATERAL, st.WalkModes.FORWARD]
    out = {}
    n = 0
    for mode_idx in range(5):
        for phase_half in (False, True):
            for itr in (0, 7000, 20000):
                if mode_idx != 4 and itr != 0:
                    continue
                client = FakeClient()
                os.chdir(REF)          # the constructor opens "utils/footstep_plans.txt" relative to the cwd
                try:
                    task = st.SteppingTask(client=client, dt=0.025, neutral_foot_orient=np.array([1, 0, 0, 0]), root_body="root",
                                           lfoot_body="lfoot", rfoot_body="rfoot", head_body="head")
                finally:
                    os.chdir(cwd)
                task._goal_height_ref, task._total_duration, task._swing_duration, task._stance_duration = 0.80, 1.1, 0.75, 0.35
                yaw = rs.uniform(-1, 1)
                rootq = _tf3_min.euler2quat(rs.uniform(-0.1, 0.1), rs.uniform(-0.1, 0.1), yaw)
                client.k = dict(xpos=dict(root=np.array([rs.uniform(-1, 1), rs.uniform(-1, 1), 0.8]), head=np.zeros(3),
                                          lfoot=np.array([rs.uniform(-1, 1), rs.uniform(-1, 1), 0.1]),
                                          rfoot=np.array([rs.uniform(-1, 1), rs.uniform(-1, 1), 0.1])),
                                xquat=dict(root=rootq))
                choice, first_u, cflat, plan_idx = int(rs.integers(0, 2)), float(rs.uniform(0.095, 0.105)), int(rs.integers(2, 4)), int(rs.integers(0, len(task.plans)))
                queue = []

                def fake_choice(a, p=None, _q=queue):
                    _q.append("c")
                    k = len([x for x in _q if x == "c"])
                    if k == 1:                       # initial phase: np.random.choice([0, period / 2])
                        return a[1] if phase_half else a[0]
                    if k == 2:                       # walk mode
                        return MODES[mode_idx]
                    return a[choice]                 # LATERAL side / FORWARD stair direction
                np.random.choice = fake_choice
                np.random.uniform = lambda lo, hi, *_a: first_u
                np.random.randint = lambda lo, hi=None, *_a: cflat
                pyrandom.choice = lambda seq: seq[plan_idx]
                try:
                    task.reset(iter_count=itr)
                finally:
                    np.random.choice, np.random.uniform, np.random.randint, pyrandom.choice = real["choice"], real["uniform"], real["randint"], real["pychoice"]
                pre = f"r{n}_"
                out[pre + "in"] = np.concatenate([[mode_idx, int(phase_half), itr, choice, first_u, cflat, plan_idx],
                                                  client.k["xpos"]["root"], client.k["xpos"]["lfoot"], client.k["xpos"]["rfoot"], rootq])
                out[pre + "plan"] = np.array(task.plans[plan_idx])
                out[pre + "sequence"] = np.array(task.sequence)
                out[pre + "state"] = np.array([task._phase, task._period, task.t1, task.t2, task.delay_frames, task.target_radius])
                out[pre + "boxpos"] = np.array([client.model.body(f"box{i + 1:02d}").pos for i in range(20)])
                out[pre + "boxquat"] = np.array([client.model.body(f"box{i + 1:02d}").quat for i in range(20)])
                out[pre + "floor"] = client.model.body("floor").pos.copy()
                # ---- a run of control steps on scripted kinematics: feet hop onto successive targets
                T = 110
                log_goal, log_rew, log_done, log_state, log_kin = [], [], [], [], []
                for t in range(T):
                    tgt = np.array(task.sequence[task.t1][0:3])
                    near = (t // 36) % 3 != 2         # dwell on the target long enough to advance it (30 frames), then wander off
                    lpos = tgt + rs.normal(size=3) * (0.05 if near else 0.5)
                    rpos = tgt + rs.normal(size=3) * 0.4
                    rq = _tf3_min.euler2quat(rs.uniform(-0.2, 0.2), rs.uniform(-0.2, 0.2), rs.uniform(-3, 3))
                    rootp = np.array([tgt[0] + rs.normal() * 0.3, tgt[1] + rs.normal() * 0.3, 0.8 + rs.normal() * 0.1])
                    if t % 17 == 16:
                        rootp[2] = min(lpos[2], rpos[2]) + 0.55          # triggers the height termination
                    rcon = [[0, 0, rs.uniform(-0.02, 0.1)]] if t % 3 else []
                    lcon = [[0, 0, rs.uniform(-0.02, 0.1)], [0, 0, rs.uniform(-0.02, 0.1)]] if t % 4 else []
                    client.k = dict(xpos=dict(root=rootp, head=rootp + np.array([rs.normal() * 0.05, rs.normal() * 0.05, 0.5]), lfoot=lpos, rfoot=rpos,
                                              lf_force=lpos, rf_force=rpos),
                                    xquat=dict(root=rq, lf_force=rq, rf_force=rq), lvel=rs.normal(size=3) * 0.2, rvel=rs.normal(size=3) * 0.2,
                                    lgrf=float(rs.uniform(0, 600)), rgrf=float(rs.uniform(0, 600)), rcon=rcon, lcon=lcon, selfcol=bool(t % 29 == 28))
                    task.step()
                    rew = task.calc_reward(None, None, None)
                    log_goal.append(np.concatenate([task._goal_steps_x, task._goal_steps_y, task._goal_steps_z, task._goal_steps_theta]))
                    log_rew.append([rew[k] for k in ("foot_frc_score", "foot_vel_score", "orient_cost", "height_error", "step_reward", "upper_body_reward")])
                    log_done.append(int(task.done()))
                    log_state.append([task._phase, task.t1, task.t2, int(task.target_reached), task.target_reached_frames])
                    log_kin.append(np.concatenate([rootp, client.k["xpos"]["head"], lpos, rpos, rq, client.k["lvel"], client.k["rvel"],
                                                   [client.k["lgrf"], client.k["rgrf"], float(client.k["selfcol"]),
                                                    min([c[2] for c in rcon + lcon]) if (rcon or lcon) else 0.0, float(bool(rcon or lcon))]]))
                out[pre + "goal"], out[pre + "rew"], out[pre + "done"] = np.array(log_goal), np.array(log_rew), np.array(log_done)
                out[pre + "tstate"], out[pre + "kin"] = np.array(log_state), np.array(log_kin)
                n += 1
    out["n"] = n
    np.savez_compressed(os.path.join(OUT, "stepping.npz"), **out)
    print("stepping golden:", n, "cases")


if __name__ == "__main__":
    sys.path.insert(0, REF)
    _stub_modules()
    if len(sys.argv) > 1 and sys.argv[1] == "stepping":     # regenerate only the stepping-task fixture
        gen_stepping()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rppo":
        gen_rppo("h32_padded", False, [6, 4, 2, 6, 3, 3], 21)
        gen_rppo("h32_mirror", True, [5, 5, 5], 22)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "imitate":
        gen_ppo("h64_imitate", 64, 80, True, False, 2, 14, imitate=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "misc":
        gen_misc()
        sys.exit(0)
    gen_gae()
    gen_clock()
    gen_misc()
    gen_ppo("h64_mirror", 64, 96, True, False, 2, 11)
    gen_ppo("h64_learnstd", 64, 70, False, True, 2, 12)
    gen_ppo("h256_mirror", 256, 128, True, False, 1, 13)
    gen_ppo("h64_imitate", 64, 80, True, False, 2, 14, imitate=True)
    gen_rppo("h32_padded", False, [6, 4, 2, 6, 3, 3], 21)
    gen_rppo("h32_mirror", True, [5, 5, 5], 22)
    gen_stepping()
    print("golden fixtures written to", OUT)
