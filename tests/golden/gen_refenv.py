"""Fixture generator: EXECUTES the reference's own environment classes (envs/jvrc/jvrc_walk.py, envs/h1/h1_env.py,
envs/h1/h1_walk.py -> envs/common/base_humanoid_env.py, envs/common/robot_interface.py, robots/robot_base.py,
tasks/{walking_task,standing_task,rewards,observations}.py, envs/common/domain_randomization.py) in the build container and
records what they compute, so that tests/test_refenv_pin.py can hold the oracle envs (oracle/env_*.py) -- the checker of the
HIP kernels -- to the reference's Python layer by EXECUTION rather than by reading.

What makes this possible without the MuJoCo wheel: tests/golden/_fake_mujoco.py stands in for the `mujoco` module and serves
mj_step / mj_forward / contacts / mj_contactForce / mj_objectVelocity from the float64 CPU oracle on the stand-in robot
models (the reference's `_build_xml` finds the stand-in MJCF at its export path and does not call dm_control).  The physics
under the reference code is therefore the oracle's, identical on both sides of the comparison; everything ABOVE mj_step is the
reference's code, unmodified.  Every np.random call the reference makes is logged (kind, parameters, result) in call order:
the test replays that tape into the oracle env's draws and checks kind and parameters of each one, which pins the draw ORDER.

Run (build container only):  python tests/golden/gen_refenv.py
"""
import os
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)
sys.path.insert(0, OUT)
sys.path.insert(0, REF)


def _install():
    import _fake_mujoco
    import _tf3_min
    _fake_mujoco.install(sys.modules)
    _tf3_min.install(sys.modules)
    dm = types.ModuleType("dm_control")
    dm.mjcf = types.ModuleType("dm_control.mjcf")
    sys.modules["dm_control"], sys.modules["dm_control.mjcf"] = dm, dm.mjcf
    for name in ("imageio",):
        sys.modules.setdefault(name, types.ModuleType(name))


class Tape:
    """Logs the np.random calls of the reference code (global stream, seeded), flattened to scalars."""
    KINDS = {"uniform": 0, "randint": 1, "choice": 2, "randn": 3}

    def __init__(self):
        self.rows = []            # (kind, p0, p1, value)
        self._real = {}

    def __enter__(self):
        r = self._real = dict(uniform=np.random.uniform, randint=np.random.randint, choice=np.random.choice, randn=np.random.randn)

        def uniform(lo=0.0, hi=1.0, size=None):
            v = r["uniform"](lo, hi, size)
            lo_b, hi_b, v_b = np.broadcast_arrays(np.asarray(lo, float), np.asarray(hi, float), np.asarray(v, float))
            for a, b, c in zip(lo_b.reshape(-1), hi_b.reshape(-1), v_b.reshape(-1)):
                self.rows.append((0, a, b, c))
            return v

        def randint(lo, hi=None, size=None):
            assert size is None
            v = r["randint"](lo, hi)
            self.rows.append((1, 0 if hi is None else lo, lo if hi is None else hi, v))
            return v

        def choice(a, size=None, replace=True, p=None):
            assert size is None
            n = len(a)
            k = r["choice"](n, p=p)
            cdf = np.cumsum(p) if p is not None else np.arange(1, n + 1) / n
            self.rows.append((2, 0.0 if k == 0 else cdf[k - 1], cdf[k], k))      # the bin of the uniform draw behind it
            return a[k]

        def randn(*shape):
            v = r["randn"](*shape)
            for c in np.asarray(v, float).reshape(-1):
                self.rows.append((3, 0.0, 1.0, c))
            return v

        np.random.uniform, np.random.randint, np.random.choice, np.random.randn = uniform, randint, choice, randn
        return self

    def __exit__(self, *a):
        for k, f in self._real.items():
            setattr(np.random, k, f)

    def mark(self):
        return len(self.rows)


def _export(env_dir, name, src):
    d = os.path.join("/tmp/mjcf-export", env_dir)
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(src, os.path.join(d, name))


def run_env(tag, make_env, seed, T, act_dim, act_std, out, extra=None):
    """Roll the reference env for T control steps from a reset with a fixed action tape; episodes that end are reset, as the
    reference's rollout worker does (rl/workers/rollout_worker.py:165-176)."""
    np.random.seed(seed)
    tape = Tape()
    rs = np.random.default_rng(1000 + seed)
    acts = (rs.normal(size=(T, act_dim)) * act_std).astype(np.float32)
    log = dict(obs=[], rew=[], done=[], terms=[], qpos=[], qvel=[], mark=[], reset_obs=[], reset_at=[], reset_qpos=[], reset_qvel=[],
               act_pos=[], act_vel=[], act_tau=[], ctrl=[])
    with tape:
        env = make_env()
        names = None
        obs0 = env.reset()
        log["reset_obs"].append(obs0.copy()); log["reset_at"].append(-1)
        log["reset_qpos"].append(env.data.qpos.copy()); log["reset_qvel"].append(env.data.qvel.copy())
        log["mark"].append(tape.mark())
        for t in range(T):
            obs, r, done, info = env.step(acts[t].astype(np.float64))
            if names is None:
                names = list(info.keys())
            assert list(info.keys()) == names
            log["obs"].append(obs.copy()); log["rew"].append(r); log["done"].append(int(done))
            log["terms"].append([info[k] for k in names])
            log["qpos"].append(env.data.qpos.copy()); log["qvel"].append(env.data.qvel.copy())
            log["act_pos"].append(np.array(env.interface.get_act_joint_positions())); log["act_vel"].append(np.array(env.interface.get_act_joint_velocities()))
            log["act_tau"].append(np.array(env.interface.get_act_joint_torques())); log["ctrl"].append(env.data.ctrl.copy())
            log["mark"].append(tape.mark())
            if done:
                o = env.reset()
                log["reset_obs"].append(o.copy()); log["reset_at"].append(t)
                log["reset_qpos"].append(env.data.qpos.copy()); log["reset_qvel"].append(env.data.qvel.copy())
                log["mark"][-1] = tape.mark()
    pre = tag + "_"
    out[pre + "acts"] = acts
    out[pre + "tape"] = np.array(tape.rows, dtype=np.float64).reshape(-1, 4)
    out[pre + "term_names"] = np.array(names)
    for k, v in log.items():
        out[pre + k] = np.array(v)
    if extra:
        extra(env, out, pre)
    print(f"{tag}: {T} steps, {int(np.sum(log['done']))} episode ends, {len(tape.rows)} random draws, "
          f"mean reward {np.mean(log['rew']):.4f}")


def gen_rollout_worker(out_path):
    """EXECUTES the reference's RolloutWorker.sample (rl/workers/rollout_worker.py:97-199) + PPOBuffer on a scripted env with
    the reference's own Gaussian_FF_Actor / FF_V: three consecutive sample() calls whose episodes end by termination, by
    truncation at max_traj_len, in the middle of a buffer (carry-over) and exactly at a buffer end."""
    import torch
    ray = types.ModuleType("ray")
    ray.remote = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda c: c))
    sys.modules["ray"] = ray
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = tb
    from rl.policies.actor import Gaussian_FF_Actor
    from rl.policies.critic import FF_V
    from rl.workers.rollout_worker import RolloutWorker
    torch.manual_seed(5)
    D, A, MAXLEN, STEPS = 6, 3, 7, 20
    rs = np.random.default_rng(9)
    table = rs.normal(size=(400, D))
    rewards = rs.uniform(0, 1, size=400)
    term_at = {4, 30, 33, 39}            # env steps (global count) that terminate; 39 is the last step of the 2nd buffer

    class Env:
        def __init__(self):
            self.k = 0                     # global step counter
            self.resets = 0
            self.robot = types.SimpleNamespace(iteration_count=0)
            self.log = []

        def reset(self):
            self.resets += 1
            self.log.append(("reset", self.k))
            return table[(self.k * 7 + 3 * self.resets) % 400].copy()

        def step(self, a):
            k = self.k
            self.k += 1
            return table[(k * 7 + 1) % 400] + 0.01 * float(np.sum(a)), rewards[k], k in term_at, {}

    policy = Gaussian_FF_Actor(D, A, layers=(16, 16), init_std=0.2, learn_std=False, bounded=False)
    critic = FF_V(D, layers=(16, 16))
    policy.obs_mean = critic.obs_mean = torch.zeros(D)
    policy.obs_std = critic.obs_std = torch.ones(D)
    w = RolloutWorker(Env, policy, critic, seed=None, worker_id=0)
    out = {}
    for call in range(3):
        b = w.sample(0.99, 0.95, STEPS, MAXLEN, deterministic=True)
        pre = f"c{call}_"
        for f in ("states", "actions", "rewards", "values", "returns", "dones", "traj_idx", "ep_lens", "ep_rewards"):
            out[pre + f] = getattr(b, f).numpy().copy()
        out[pre + "carried"] = np.array([w.current_state is not None])
        out[pre + "resets"] = np.array([w.env.resets])
        out[pre + "final_state"] = (w.current_state.numpy().copy() if w.current_state is not None else np.zeros(D, np.float32))
    with torch.no_grad():
        for k, v in list(policy.state_dict().items()) + [("c_" + k, v) for k, v in critic.state_dict().items()]:
            out["w_" + k] = v.numpy().copy()
    out["table"], out["rew_table"], out["term_at"] = table, rewards, np.array(sorted(term_at))
    out["cfg"] = np.array([D, A, MAXLEN, STEPS])
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "resets:", w.env.resets, "env log:", w.env.log)


def main():
    _install()
    from learninghumanoidwalking_amd.envs.h1 import H1_STANDIN_XML
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML
    _export("jvrc_walk", "jvrc.xml", JVRC_STANDIN_XML)
    _export("h1", "h1.xml", H1_STANDIN_XML)
    _export("h1_walk", "h1.xml", H1_STANDIN_XML)
    from envs.h1.h1_env import H1Env
    from envs.h1.h1_walk import H1WalkEnv
    from envs.jvrc.jvrc_walk import JvrcWalkEnv
    out = {}

    def jvrc_extra(env, out, pre):
        out[pre + "obs_mean"], out[pre + "obs_std"] = env.obs_mean, env.obs_std
        out[pre + "mirrored_obs"], out[pre + "mirrored_acts"] = np.array(env.robot.mirrored_obs), np.array(env.robot.mirrored_acts)
        out[pre + "nominal_pose"] = np.array(env.nominal_pose)

    # random exploration-level actions: the stand-in robot falls after ~70 steps, so terminations, resets and the
    # prev_action / prev_torque carry-over across episodes (robots/robot_base.py:82-85) are all in the tape
    run_env("jvrc_walk_a", JvrcWalkEnv, seed=3, T=260, act_dim=12, act_std=0.223, out=out, extra=jvrc_extra)
    run_env("jvrc_walk_b", JvrcWalkEnv, seed=11, T=200, act_dim=12, act_std=0.05, out=out)
    run_env("h1_a", H1Env, seed=5, T=260, act_dim=10, act_std=0.223, out=out, extra=jvrc_extra if False else None)
    run_env("h1_b", H1Env, seed=17, T=200, act_dim=10, act_std=0.05, out=out)
    run_env("h1_walk_a", H1WalkEnv, seed=7, T=220, act_dim=10, act_std=0.15, out=out)
    np.savez_compressed(os.path.join(OUT, "refenv.npz"), **out)
    print("wrote", os.path.join(OUT, "refenv.npz"))
    gen_rollout_worker(os.path.join(OUT, "rollout_worker.npz"))


def gen_history(out_path):
    """obs_history_len = 3 (base_humanoid_env.py:177-197: deque of base observations, newest first, zero-filled after a reset):
    the reference's JvrcWalkEnv on a copy of its own YAML with only that key changed."""
    _install()
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML
    _export("jvrc_walk", "jvrc.xml", JVRC_STANDIN_XML)
    from envs.jvrc.jvrc_walk import JvrcWalkEnv
    src = open(os.path.join(REF, "envs", "jvrc", "configs", "base.yaml")).read()
    assert "obs_history_len: 1" in src
    yml = "/tmp/jvrc_hist3.yaml"
    open(yml, "w").write(src.replace("obs_history_len: 1", "obs_history_len: 3"))
    out = {}

    def extra(env, out, pre):
        out[pre + "obs_mean"], out[pre + "obs_std"] = env.obs_mean, env.obs_std
        out[pre + "base_obs_len"], out[pre + "history_len"] = env.base_obs_len, env.history_len
    run_env("jvrc_walk_h3", lambda: JvrcWalkEnv(path_to_yaml=yml), seed=3, T=160, act_dim=12, act_std=0.223, out=out, extra=extra)
    keep = {k: v for k, v in out.items() if k.split("jvrc_walk_h3_")[1] in ("obs", "done", "reset_obs", "reset_at", "obs_mean", "obs_std", "base_obs_len", "history_len", "acts")}
    np.savez_compressed(out_path, **keep)
    print("wrote", out_path)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "history":
        gen_history(os.path.join(OUT, "refenv_history.npz"))
    elif len(sys.argv) > 1 and sys.argv[1] == "rollout":
        _install()
        gen_rollout_worker(os.path.join(OUT, "rollout_worker.npz"))
    else:
        main()
