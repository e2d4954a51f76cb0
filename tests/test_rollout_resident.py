"""The resident rollout (lhw_env_rollout: policy step + control step for T control steps inside the stepper's wavefronts,
csrc/lhw_humanoid_rollout.hip) against the launch-per-step pipeline it replaces (T x { fused strip policy launch ; control-step
launch }), on the SIMT emulator: every buffer of the rollout must be BITWISE the same -- observations, actions, log-densities,
terminal observations, rewards, done flags, and the env state afterwards.  Reference semantics: the body of
RolloutWorker.sample's loop, /root/reference/rl/workers/rollout_worker.py:142-181.  CPU twin of tests/test_rollout_resident_gpu.py."""
import ctypes

import numpy as np
import pytest

from tests import emu


class NumpyActor:
    """A random float32 actor obs -> 256 -> 256 -> act as the LhwRolloutPolicy view (host arrays: the emulated library reads them)."""

    def __init__(self, obs_dim, act_dim, seed, scale=1.0, deterministic=False, counter=7):
        from learninghumanoidwalking_amd import _lib as product
        rs = np.random.default_rng(seed)
        Dp, Op, H = (obs_dim + 3) // 4 * 4, (act_dim + 3) // 4 * 4, 256
        w1 = np.zeros((H, Dp), np.float32)
        w1[:, :obs_dim] = rs.normal(size=(H, obs_dim)) * scale / np.sqrt(obs_dim)
        w2 = (rs.normal(size=(H, H)) * scale / np.sqrt(H)).astype(np.float32)
        w3 = np.zeros((Op, H), np.float32)
        w3[:act_dim] = rs.normal(size=(act_dim, H)) * 0.3 * scale / np.sqrt(H)
        self.a = dict(w1t=np.ascontiguousarray(w1.T), b1=(rs.normal(size=H) * 0.1).astype(np.float32), w2t=np.ascontiguousarray(w2.T),
                      b2=(rs.normal(size=H) * 0.1).astype(np.float32), w3t=np.ascontiguousarray(w3.T),
                      b3=np.zeros(Op, np.float32), stdv=np.full(act_dim, 0.223, np.float32),
                      obs_mean=(rs.normal(size=obs_dim) * 0.1).astype(np.float32), obs_std=(0.5 + rs.uniform(size=obs_dim)).astype(np.float32))
        self.a["b3"][:act_dim] = rs.normal(size=act_dim) * 0.05
        q = product.LhwRolloutPolicy()
        for k, v in self.a.items():
            setattr(q, k, v.ctypes.data)
        q.obs_dim, q.obs_pad, q.act_dim, q.act_pad, q.hidden = obs_dim, Dp, act_dim, Op, H
        q.deterministic, q.seed, q.counter = int(deterministic), 1234567, counter
        self.view = q


def _buffers(T, N, D, A):
    return dict(obs=np.zeros((T + 1, N, D), np.float32), act=np.zeros((T, N, A), np.float32), logp=np.zeros((T, N), np.float32),
                tob=np.zeros((T, N, D), np.float32), rew=np.zeros((T, N), np.float32), done=np.zeros((T, N), np.uint8))


def _per_step(env, pol, T, obs0, env_id_base=0):
    """the launch-per-step pipeline: fused policy launch, then the control-step launch(es)"""
    L = emu.lib()
    N, D, A = env.n_envs, env.obs_dim, env.act_dim
    b = _buffers(T, N, D, A)
    b["obs"][0] = obs0
    y = np.zeros((N, pol.view.act_pad), np.float32)
    for t in range(T):
        rc = L.lhw_debug_policy_step(ctypes.byref(pol.view), b["obs"][t].ctypes.data, N, env_id_base, pol.view.counter + t, y.ctypes.data,
                                     b["act"][t].ctypes.data, b["logp"][t].ctypes.data, None)
        assert rc == 0, L.lhw_last_error()
        obs, rew, done, tob = env.step(b["act"][t])
        b["obs"][t + 1], b["rew"][t], b["done"][t], b["tob"][t] = obs, rew, done, tob
    return b


def _resident(env, pol, T, obs0, first=0, count=None):
    N, D, A = env.n_envs, env.obs_dim, env.act_dim
    b = _buffers(T, N, D, A)
    b["obs"][0] = obs0
    env.rollout(pol.view, T, b["obs"], b["act"], b["logp"], b["tob"], b["rew"], b["done"], first=first, count=count)
    return b


def _same(a, b, rows=slice(None)):
    for k in a:
        np.testing.assert_array_equal(a[k][:, rows], b[k][:, rows], err_msg=k)


def _fallen_states(spec, N, seed):
    """poses found offline (tests/test_jvrc_gpu.py) in which the robot lies on the floor with 10 .. 13 simultaneous contacts: beyond
    the 8 of the two-envs-per-wave layout, within the 16 of the one-env-per-wave layout.  Envs 0 and 3 take them (one env of each
    wavefront overflows), the others stand."""
    poses = ([0.0, 0.0, 0.2571, -0.7309, -0.1371, 0.232, 0.627, -0.9859, -0.3243, -0.4729, 0.119, 0.609, 0.1118, -1.4146, 0.1458, 0.4932, 2.1903, 0.42, -0.5225],
             [0.0, 0.0, 0.1223, -0.4502, 0.0287, 0.3794, 0.8078, -1.2999, -0.2333, 0.4393, 0.4916, 0.2839, -0.8672, -1.533, 0.0192, -0.4235, 2.2823, -0.1645, -1.051])
    rs = np.random.default_rng(seed)
    q = np.tile(spec.nominal_pose, (N, 1))
    v = rs.normal(size=(N, spec.model().nv)) * 0.1
    for i, pose in zip((0, 3), poses):
        q[i] = pose
        q[i, 3:7] /= np.linalg.norm(q[i, 3:7])
        v[i] = 0
    return q, v


def test_resident_rollout_is_bitwise_the_launch_per_step_rollout_jvrc_walk():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    spec = JvrcWalkSpec()
    N, T = 5, 7                                   # odd: the last wavefront holds one env
    envs = [emu.make_emulated(spec, N, seed=3, max_traj_len=4) for _ in range(2)]     # truncation + auto-reset inside the rollout
    pol = NumpyActor(37, 12, seed=5, scale=2.0)
    obs0 = [e.reset().copy() for e in envs]
    np.testing.assert_array_equal(obs0[0], obs0[1])
    a = _per_step(envs[0], pol, T, obs0[0])
    b = _resident(envs[1], pol, T, obs0[1])
    _same(a, b)
    assert (a["done"] & 2).any(), "no truncation / auto-reset inside the rollout"
    for x, y in zip(envs[0].get_state(), envs[1].get_state()):
        np.testing.assert_array_equal(x, y)
    assert envs[0].pop_episode_stats() == envs[1].pop_episode_stats()
    # a second rollout continues from the first one's last observation and counters
    pol.view.counter += T
    a2 = _per_step(envs[0], pol, 3, a["obs"][T])
    b2 = _resident(envs[1], pol, 3, b["obs"][T])
    _same(a2, b2)


def test_resident_rollout_repeats_overflowing_envs_inside_the_wave():
    """an env with more than 8 contacts: the two-envs-per-wave step hands it to the one-env-per-wave layout -- a second launch in the
    launch-per-step path, the same wavefront (LDS re-interpreted) in the resident rollout"""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    spec = JvrcWalkSpec()
    N, T = 4, 4
    envs = [emu.make_emulated(spec, N, seed=11, max_traj_len=50) for _ in range(2)]
    pol = NumpyActor(37, 12, seed=8)
    q, v = _fallen_states(spec, N, seed=21)
    for e in envs:
        e.reset()
        e.set_state(q, v)
    obs0 = envs[0].obs.copy()      # (the observation still describes the reset pose: the same stale input for both paths)
    a = _per_step(envs[0], pol, T, obs0)
    b = _resident(envs[1], pol, T, obs0)
    _same(a, b)
    ra, rb = envs[0].pop_rerun_count(), envs[1].pop_rerun_count()
    assert ra > 0 and ra == rb, (ra, rb)
    assert envs[0].pop_fault_stats() == envs[1].pop_fault_stats() == (0, 0)
    for x, y in zip(envs[0].get_state(), envs[1].get_state()):
        np.testing.assert_array_equal(x, y)


def test_resident_rollout_of_a_sub_range_leaves_the_other_envs_alone():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    spec = JvrcWalkSpec()
    N, T = 5, 3
    envs = [emu.make_emulated(spec, N, seed=4, max_traj_len=0) for _ in range(2)]
    pol = NumpyActor(37, 12, seed=6, deterministic=True)
    obs0 = [e.reset().copy() for e in envs]
    a = _per_step(envs[0], pol, T, obs0[0])
    b = _resident(envs[1], pol, T, obs0[1], first=1, count=3)      # envs 1..3: an odd range that starts inside a wavefront pair
    _same(a, b, rows=slice(1, 4))
    assert not b["act"][:, [0, 4]].any() and not b["obs"][1:, [0, 4]].any()
    qa, qb = envs[0].get_state()[0], envs[1].get_state()[0]
    np.testing.assert_array_equal(qa[1:4], qb[1:4])
    assert not np.array_equal(qa[0], qb[0])


@pytest.mark.parametrize("name", ["h1", "h1_walk", "jvrc_step"])
def test_resident_rollout_other_tasks(name):
    if name == "h1":
        from learninghumanoidwalking_amd.envs.h1 import H1Spec as S
    elif name == "h1_walk":
        from learninghumanoidwalking_amd.envs.h1_walk import H1WalkSpec as S
    else:
        from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec as S
    spec = S()
    N, T = 3, 4
    envs = [emu.make_emulated(spec, N, seed=2, max_traj_len=3) for _ in range(2)]
    pol = NumpyActor(spec.obs_dim, spec.act_dim, seed=9)
    obs0 = [e.reset().copy() for e in envs]
    a = _per_step(envs[0], pol, T, obs0[0])
    b = _resident(envs[1], pol, T, obs0[1])
    _same(a, b)
    for x, y in zip(envs[0].get_state(), envs[1].get_state()):
        np.testing.assert_array_equal(x, y)


def test_resident_rollout_through_the_job_queue_is_bitwise_the_same(monkeypatch):
    """Stepping task with more env groups than wave slots (jvrc_step @ 4096 on the chip; forced here by LHW_ROLLOUT_SLOTS): the
    resident waves pop (group, chunk of control steps) jobs from a queue, so a group's chunks run on whichever wave is free and a
    wave advances one group after the other.  Nothing of an env lives in a wave between control steps: every buffer and the state
    afterwards must be bitwise those of the one-wave-per-group launch, including across resets, for a chunk that does not divide T,
    and for a sub-range whose queue words sit elsewhere."""
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    spec = JvrcStepSpec()
    N, T = 4, 7
    envs = [emu.make_emulated(spec, N, seed=3, max_traj_len=4) for _ in range(3)]
    pol = NumpyActor(spec.obs_dim, spec.act_dim, seed=5, scale=2.0)
    obs0 = [e.reset().copy() for e in envs]
    monkeypatch.setenv("LHW_ROLLOUT_CHUNK", "0")
    a = _resident(envs[0], pol, T, obs0[0])
    monkeypatch.setenv("LHW_ROLLOUT_CHUNK", "3")
    monkeypatch.setenv("LHW_ROLLOUT_SLOTS", "1")
    b = _resident(envs[1], pol, T, obs0[1])
    _same(a, b)
    assert (a["done"] & 2).any()
    for x, y in zip(envs[0].get_state(), envs[1].get_state()):
        np.testing.assert_array_equal(x, y)
    assert envs[0].pop_episode_stats() == envs[1].pop_episode_stats()
    c = _resident(envs[2], pol, T, obs0[2], first=1, count=N - 1)
    _same(a, c, rows=slice(1, N))
    assert not c["act"][:, 0].any()


def test_fp16_operand_policy_step_rounds_every_operand_and_accumulates_in_float32():
    """BASELINE config 5 in the resident rollout: with `fp16_operands` the in-wave policy step rounds weights and activations to fp16
    before each product and sums in float32 over ascending k (read-out: eight 32-k partials, then the bias).  The product of two
    fp16 values is exact in float32, so a numpy restatement reproduces the means bit for bit."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    spec = JvrcWalkSpec()
    N = 3
    env = emu.make_emulated(spec, N, seed=6, max_traj_len=0)
    pol = NumpyActor(37, 12, seed=12, scale=1.5, deterministic=True)
    pol.view.fp16_operands = 1
    obs0 = env.reset().copy()
    b = _resident(env, pol, 1, obs0)
    a = pol.a
    h = lambda v: v.astype(np.float16).astype(np.float32)

    def layer(x, wt, bias):            # x [K] float32 (already fp16 values), wt [K][256]
        acc = np.zeros(wt.shape[1], np.float32)
        for k in range(wt.shape[0]):
            acc = (acc + h(wt[k]) * x[k]).astype(np.float32)     # exact product, one float32 rounding per step = fmaf
        return h(np.maximum(acc + bias, np.float32(0)))

    for i in range(N):
        x = np.zeros(40, np.float32)
        x[:37] = h(((obs0[i] - a["obs_mean"]) / a["obs_std"]).astype(np.float32))
        h1 = layer(x, a["w1t"], a["b1"])
        h2 = layer(h1, a["w2t"], a["b2"])
        parts = []
        for q in range(8):
            acc = np.zeros(12, np.float32)
            for k in range(32 * q, 32 * q + 32):
                acc = (acc + h2[k] * h(a["w3t"][k, :12])).astype(np.float32)
            parts.append(acc)
        mu = parts[0]
        for q in range(1, 8):
            mu = (mu + parts[q]).astype(np.float32)
        mu = (mu + a["b3"][:12]).astype(np.float32)
        np.testing.assert_array_equal(b["act"][0, i], mu)
    # and it is not the float32 policy
    pol.view.fp16_operands = 0
    env2 = emu.make_emulated(spec, N, seed=6, max_traj_len=0)
    c = _resident(env2, pol, 1, env2.reset().copy())
    assert not np.array_equal(b["act"], c["act"]) and np.abs(b["act"] - c["act"]).max() < 2e-2


def test_resident_rollout_exports_the_task_inputs_of_every_control_step():
    """lhw_env_rollout_task_inputs: the [T][N][160] record a reward-only task plug-in evaluates after the launch equals, step by step,
    what lhw_env_get_task_inputs returns behind each control step of the launch-per-step pipeline (robots/robot_base.py:88-96: what
    the robot hands its task once per control step) -- with a truncation / auto-reset and a two-envs-per-wave overflow re-run inside."""
    from learninghumanoidwalking_amd import _lib as product
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    spec = JvrcWalkSpec()
    N, T = 5, 6
    envs = [emu.make_emulated(spec, N, seed=3, max_traj_len=4) for _ in range(2)]
    pol = NumpyActor(37, 12, seed=5, scale=2.0)
    q, v = _fallen_states(spec, N, seed=21)
    for e in envs:
        e.reset()
        e.set_state(q, v)
    obs0 = np.zeros((N, 37), np.float32)
    L = emu.lib()
    # launch per step, the record read back after every control step
    envs[0].enable_task_inputs(True)
    a = _buffers(T, N, 37, 12)
    a["obs"][0] = obs0
    want = np.zeros((T, N, product.TASK_INPUT_DIM))
    y = np.zeros((N, pol.view.act_pad), np.float32)
    for t in range(T):
        assert L.lhw_debug_policy_step(ctypes.byref(pol.view), a["obs"][t].ctypes.data, N, 0, pol.view.counter + t, y.ctypes.data,
                                       a["act"][t].ctypes.data, a["logp"][t].ctypes.data, None) == 0
        obs, rew, done, tob = envs[0].step(a["act"][t])
        a["obs"][t + 1], a["rew"][t], a["done"][t], a["tob"][t] = obs, rew, done, tob
        rec = np.zeros((N, product.TASK_INPUT_DIM))
        envs[0]._check(L.lhw_env_get_task_inputs(envs[0]._h, rec.ctypes.data))
        want[t] = rec
    # one launch, every record
    b = _buffers(T, N, 37, 12)
    b["obs"][0] = obs0
    got = np.full((T, N, product.TASK_INPUT_DIM), np.nan)
    envs[1].rollout(pol.view, T, b["obs"], b["act"], b["logp"], b["tob"], b["rew"], b["done"], task_inputs=got)
    _same(a, b)
    used = np.zeros(product.TASK_INPUT_DIM, bool)
    for name, (o, n) in product.TASK_INPUT_FIELDS.items():
        n = dict(qpos=19, qvel=18, qacc=18).get(name, n)
        used[o:o + n] = True
    np.testing.assert_array_equal(got[:, :, used], want[:, :, used])
    assert (a["done"] & 2).any() and envs[0].pop_rerun_count() == envs[1].pop_rerun_count() > 0
