"""The HIP stepper SOURCES (csrc/lhw_humanoid.hip, lhw_cartpole.hip, lhw_api.hip), compiled for the host against the SIMT
emulator in tests/emu and driven through the C ABI, versus the float64 CPU oracle.  This is the CPU (`-m "not gpu"`)
twin of tests/test_{jvrc,h1,h1_walk,jvrc_step,cartpole}_gpu.py: it checks lane mappings, cross-lane reductions, LDS
hand-offs and the sub-wave grouping of the kernels without a GPU.  The emulated library is test infrastructure only."""
import numpy as np
import pytest

from tests import emu


def _states(orc):
    return np.array([o.sim.qpos.copy() for o in orc]), np.array([o.sim.qvel.copy() for o in orc])


def _mk(spec_cls, orc_cls, n, seed, max_traj_len=0):
    spec = spec_cls()
    env = emu.make_emulated(spec, n, seed=seed, max_traj_len=max_traj_len)
    orc = [orc_cls(spec, seed=seed, env_id=i, max_traj_len=max_traj_len) for i in range(n)]
    return spec, env, orc


def _run_tape(env, orc, tape, qtol=1e-12, vtol=1e-10, resync=3, otol=2e-6):
    N = len(orc)
    for t in range(tape.shape[0]):
        obs, rew, done, _ = env.step(tape[t])
        res = [o.step(tape[t, i]) for i, o in enumerate(orc)]
        q, v = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_allclose(q, oq, rtol=0, atol=qtol, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(v, ov, rtol=0, atol=vtol, err_msg=f"qvel t={t}")
        np.testing.assert_allclose(obs, np.array([r[0] for r in res]), rtol=1e-5, atol=otol, err_msg=f"obs t={t}")
        np.testing.assert_allclose(rew, np.array([r[1] for r in res]), rtol=0, atol=2e-6, err_msg=f"rew t={t}")
        terms = np.array([[r[3][k] for k in o.TERMS] for r, o in zip(res, orc)])
        np.testing.assert_allclose(env.rew_terms, terms, rtol=0, atol=2e-6, err_msg=f"terms t={t}")
        np.testing.assert_array_equal(done & 1, np.array([int(r[2]) for r in res], dtype=np.uint8), err_msg=f"done t={t}")
        if t % resync == resync - 1:
            env.set_state(oq, ov)
            for o in orc:
                o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    over, div = env.pop_fault_stats()
    assert over == 0 and div == 0


def test_emulated_jvrc_walk_reset_and_tape():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    spec, env, orc = _mk(JvrcWalkSpec, OracleJvrcWalkEnv, 3, seed=9)
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    q, v = env.get_state()
    oq, ov = _states(orc)
    np.testing.assert_allclose(q, oq, rtol=0, atol=1e-13)
    np.testing.assert_allclose(v, ov, rtol=0, atol=1e-11)
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=1e-6)
    tape = (np.random.default_rng(1234).normal(size=(6, 3, 12)) * 0.223).astype(np.float32)
    _run_tape(env, orc, tape)


def test_emulated_jvrc_walk_contact_variety():
    """fallen / tangled / limit-violating poses: every primitive narrow phase and both limit sides (as the GPU test)."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    N = 6
    spec, env, orc = _mk(JvrcWalkSpec, OracleJvrcWalkEnv, N, seed=2)
    env.reset()
    for o in orc:
        o.reset()
    m = spec.model()
    rs = np.random.default_rng(11)
    q = np.tile(spec.nominal_pose, (N, 1))
    v = rs.normal(size=(N, 18)) * 0.3
    lo, hi = m.jnt_range[1:, 0], m.jnt_range[1:, 1]
    q[0, 2] = 0.3
    quat = rs.normal(size=4)
    q[0, 3:7] = quat / np.linalg.norm(quat)
    q[0, 7:] = rs.uniform(lo, hi)
    q[1] = [0.0, 0.0, 0.0707, -0.003, 0.4379, -0.8595, 0.2636, -0.793, 0.1178, -0.2801, 0.1269, -0.1164, -0.9956, -1.8286,
            0.3116, -0.2094, 1.6397, -0.3666, 0.8188]
    q[1, 3:7] /= np.linalg.norm(q[1, 3:7])
    v[1] = 0
    q[2, 2] = 1.3
    q[2, 8], q[2, 14], q[2, 9], q[2, 15] = 0.3, -0.3, 0.2, -0.2      # legs pushed through each other
    q[3, 2] = 1.2
    q[3, 7:] = np.where(rs.uniform(size=12) < 0.5, lo - rs.uniform(0.02, 0.2, 12), hi + rs.uniform(0.02, 0.2, 12))
    env.set_state(q, v)
    for i, o in enumerate(orc):
        o.set_state(q[i], v[i])
    act = (rs.normal(size=(2, N, 12)) * 0.2).astype(np.float32)
    kinds = set()
    for t in range(2):
        obs, rew, done, _ = env.step(act[t])
        res = [o.step(act[t, i]) for i, o in enumerate(orc)]
        for o in orc:
            for k in range(o.sim.ncon):
                c = o.sim.contact(k)
                kinds.add((int(m.geom_type[c["geom1"]]), int(m.geom_type[c["geom2"]])))
            if o.sim.nefc > 4 * o.sim.ncon:
                kinds.add("limit")
        gq, gv = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_allclose(gq, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(gv, ov, rtol=0, atol=1e-8, err_msg=f"qvel t={t}")
        np.testing.assert_array_equal(done & 1, np.array([int(r[2]) for r in res], dtype=np.uint8))
        env.set_state(oq, ov)
        for o in orc:
            o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    assert {(0, 3), (0, 6), "limit"} <= kinds, kinds
    assert env.pop_fault_stats() == (0, 0)


def test_emulated_plane_box_corner_lanes_partial_foot_contact():
    """The floor-foot pairs run on corner lanes (one lane per box corner, ranks by ballot): tilted robots whose feet touch with one, two,
    three or four corners, next to feet in the air (pairs the broad phase removes) -- same contacts, in the same order, as the oracle's
    serial corner loop."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    N = 10
    spec, env, orc = _mk(JvrcWalkSpec, OracleJvrcWalkEnv, N, seed=5)
    env.reset()
    for o in orc:
        o.reset()
    m = spec.model()
    rs = np.random.default_rng(3)
    q = np.tile(orc[0].sim.qpos.copy(), (N, 1))          # the settled standing pose
    v = np.zeros((N, 18))
    for i in range(N):
        roll, pitch = rs.uniform(-0.12, 0.12, 2) if i < N - 2 else (0.0, 0.0)
        cr, sr, cp, sp = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2)
        q[i, 3:7] = [cr * cp, sr * cp, cr * sp, -sr * sp]
        q[i, 2] += rs.uniform(-0.004, 0.02)
    q[N - 1, 2] += 0.3                                    # both feet in the air
    env.set_state(q, v)
    for i, o in enumerate(orc):
        o.set_state(q[i], v[i])
    floor = {g for g in range(len(m.geom_type)) if m.geom_type[g] == 0}
    counts = set()
    zero = np.zeros((N, 12), np.float32)
    for t in range(2):
        env.step(zero)
        for i, o in enumerate(orc):
            o.step(zero[i])
            per_pair = {}
            for k in range(o.sim.ncon):
                c = o.sim.contact(k)
                if c["geom1"] in floor and m.geom_type[c["geom2"]] == 6:
                    per_pair[c["geom2"]] = per_pair.get(c["geom2"], 0) + 1
            counts |= set(per_pair.values())
            if i == N - 1 and t == 0:
                assert not per_pair
        gq, gv = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_allclose(gq, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(gv, ov, rtol=0, atol=1e-8, err_msg=f"qvel t={t}")
        env.set_state(oq, ov)
        for o in orc:
            o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    assert 4 in counts and counts & {1, 2, 3}, counts
    assert env.pop_fault_stats() == (0, 0)


def test_emulated_jvrc_walk_auto_reset():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    N, T, L = 3, 14, 6
    spec, env, orc = _mk(JvrcWalkSpec, OracleJvrcWalkEnv, N, seed=21, max_traj_len=L)
    env.reset()
    for o in orc:
        o.reset()
    tape = (np.random.default_rng(7).normal(size=(T, N, 12)) * 0.4).astype(np.float32)
    seen = 0
    for t in range(T):
        obs, rew, done, tob = env.step(tape[t])
        res = [o.step_auto(tape[t, i]) for i, o in enumerate(orc)]
        flags = np.array([r[2] for r in res], dtype=np.uint8)
        np.testing.assert_array_equal(done, flags, err_msg=f"flags t={t}")
        np.testing.assert_allclose(obs, np.array([r[0] for r in res]), rtol=1e-4, atol=1e-4, err_msg=f"obs t={t}")
        np.testing.assert_allclose(tob, np.array([r[3] for r in res]), rtol=1e-4, atol=1e-4, err_msg=f"term obs t={t}")
        seen |= int(np.bitwise_or.reduce(flags))
        oq, ov = _states(orc)
        q, v = env.get_state()
        np.testing.assert_allclose(q, oq, rtol=0, atol=1e-9, err_msg=f"qpos t={t}")
    assert seen & 2
    assert env.pop_episode_stats()[2] > 0


def test_emulated_h1_and_h1_walk():
    from learninghumanoidwalking_amd.envs.h1 import H1Spec
    from learninghumanoidwalking_amd.envs.h1_walk import H1WalkSpec
    from oracle.env_h1 import OracleH1Env
    from oracle.env_h1_walk import OracleH1WalkEnv
    for spec_cls, orc_cls, seed in ((H1Spec, OracleH1Env, 12), (H1WalkSpec, OracleH1WalkEnv, 4)):
        spec, env, orc = _mk(spec_cls, orc_cls, 2, seed=seed)
        obs = env.reset().copy()
        ref = np.array([o.reset() for o in orc])
        q, v = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_allclose(q, oq, rtol=0, atol=1e-13)
        np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
        assert all(o.sim.nefc >= 10 for o in orc)            # frictionloss rows active
        tape = (np.random.default_rng(5).normal(size=(4, 2, 10)) * 0.05).astype(np.float32)
        _run_tape(env, orc, tape, otol=2e-5)


def test_emulated_jvrc_step_forward_mode_on_boxes():
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    from oracle.env_jvrc_step import OracleJvrcStepEnv
    N = 4
    spec, env, orc = _mk(JvrcStepSpec, OracleJvrcStepEnv, N, seed=12)
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=1e-6)
    assert any(o.mode == 4 for o in orc), [o.mode for o in orc]   # at least one env stands on the boxes (FORWARD mode)
    seq, fz, ist = env.debug_step_record()
    for i, o in enumerate(orc):
        np.testing.assert_allclose(seq[i, :len(o.sequence), :4], np.asarray(o.sequence)[:, :4], rtol=0, atol=1e-12)
    tape = (np.random.default_rng(3).normal(size=(3, N, 12)) * 0.1).astype(np.float32)
    _run_tape(env, orc, tape)


def test_emulated_jvrc_step_every_walk_mode_on_the_reference_terrain():
    """Outside FORWARD mode the reference leaves the terrain boxes coplanar with the floor (tasks/stepping_task.py:320-334): a foot
    rests on the floor and on every box under it -- 16 (STANDING) to ~110 (LATERAL) contacts per env, far beyond the 16 whose
    rows fit one lane each.  Those sub-steps take the many-contact path (contacts / Jacobian rows / row state in the HBM
    workspace, rows walked in strides of the wave: newton_big) and must match the oracle, which collides every box, like any other."""
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    from oracle.env_jvrc_step import OracleJvrcStepEnv
    N = 7
    spec, env, orc = _mk(JvrcStepSpec, OracleJvrcStepEnv, N, seed=7)
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=1e-6)
    assert {o.mode for o in orc} == {0, 1, 2, 3, 4}, [o.mode for o in orc]     # CURVED, STANDING, BACKWARD, LATERAL, FORWARD
    tape = (np.random.default_rng(3).normal(size=(3, N, 12)) * 0.1).astype(np.float32)
    env.step(tape[0]); [o.step(tape[0, i]) for i, o in enumerate(orc)]
    ncon = {o.mode: o.sim.ncon for o in orc}
    assert ncon[4] == 8 and ncon[3] > 64 and ncon[2] > 16 and ncon[0] > 16, ncon
    _run_tape(env, orc, tape)


def test_emulated_cartpole():
    from learninghumanoidwalking_amd.envs import CartpoleSpec
    from oracle.env_cartpole import OracleCartpoleEnv
    spec = CartpoleSpec()
    N, T = 5, 40
    env = emu.make_emulated(spec, N, seed=1)
    orc = [OracleCartpoleEnv(spec.model(), seed=1, env_id=i) for i in range(N)]
    env.reset()
    for o in orc:
        o.reset()
    tape = np.random.default_rng(0).uniform(-1, 1, size=(T, N, 1)).astype(np.float32)
    for t in range(T):
        env.step(tape[t])
        for i, o in enumerate(orc):
            o.step(tape[t, i, 0])
    q, v = env.get_state()
    oq, ov = _states(orc)
    np.testing.assert_allclose(q, oq, rtol=0, atol=1e-11)
    np.testing.assert_allclose(v, ov, rtol=0, atol=1e-10)


def test_emulated_h1_gaussian_observation_noise(tmp_path):
    """observation_noise.type: gaussian (base_humanoid_env.py:326-327): kernel == oracle draw for draw, and the noise has the
    configured standard deviation (the YAML is the reference's H1 config with the type switched)."""
    import yaml
    from learninghumanoidwalking_amd.envs.h1 import H1_BASE_YAML as H1_YAML, H1Spec
    from oracle.env_h1 import OracleH1Env
    cfg = yaml.safe_load(open(H1_YAML))
    cfg["observation_noise"]["type"] = "gaussian"
    path = tmp_path / "h1_gauss.yaml"
    path.write_text(yaml.safe_dump(cfg))
    spec = H1Spec(yaml_path=str(path))
    assert spec.obs_noise_type == "gaussian" and (spec.task_params()[4:] <= 0).all()
    n = 4
    env = emu.make_emulated(spec, n, seed=21)
    orc = [OracleH1Env(spec, seed=21, env_id=i) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-5)
    tape = (np.random.default_rng(1).normal(size=(3, n, 10)) * 0.05).astype(np.float32)
    _run_tape(env, orc, tape, otol=2e-4)
    # statistics of the draw itself: 35 entries x many counters, unit variance after scaling
    from oracle import rng
    u = lambda c, s: rng.u01(5, 0, 4, c, s)
    z = np.array([np.sqrt(-2.0 * np.log(1.0 - u(c, k))) * np.cos(6.283185307179586 * u(c, 64 + k)) for c in range(300) for k in range(35)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    with pytest.raises(ValueError):
        cfg["observation_noise"]["type"] = "laplace"
        path.write_text(yaml.safe_dump(cfg))
        H1Spec(yaml_path=str(path))


def test_emulated_frictionless_contacts(tmp_path):
    """condim = 1 (one row per contact, no friction pyramid): a constraint type the stand-in models do not use by default."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML, JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    xml = open(JVRC_STANDIN_XML).read()
    assert xml.count('<geom condim="3"') == 1
    path = tmp_path / "jvrc_condim1.xml"
    path.write_text(xml.replace('<geom condim="3"', '<geom condim="1"'))
    spec = JvrcWalkSpec(xml_path=str(path))
    n = 2
    env = emu.make_emulated(spec, n, seed=6)
    orc = [OracleJvrcWalkEnv(spec, seed=6, env_id=i) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    assert all(o.sim.ncon >= 4 and o.sim.nefc == o.sim.ncon for o in orc)      # one row per contact
    tape = (np.random.default_rng(2).normal(size=(4, n, 12)) * 0.2).astype(np.float32)
    _run_tape(env, orc, tape)


def test_emulated_contact_parameter_conventions(tmp_path):
    """The kernel's row parameters under the conventions the stand-in models never exercise: negative solref (direct stiffness /
    damping), a solimp power other than 2 (the general pow() branch), geom margins, solmix-weighted mixing on one foot and
    priority-based selection on the other -- the oracle's handling of each is pinned to MuJoCo's documented formulas in
    tests/test_oracle_physics.py."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML, JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    xml = open(JVRC_STANDIN_XML).read()
    floor = '<geom name="floor" type="plane" size="0 0 0.25" contype="1" conaffinity="0"/>'
    rfoot = '<geom name="R_ANKLE_P_S-foot" type="box" size="0.1 0.05 0.01" pos="0.029 0 -0.09778" contype="0" conaffinity="1"/>'
    lfoot = '<geom name="L_ANKLE_P_S-foot" type="box" size="0.1 0.05 0.01" pos="0.029 0 -0.09778" contype="0" conaffinity="1"/>'
    assert floor in xml and rfoot in xml and lfoot in xml
    xml = xml.replace(floor, floor[:-2] + ' solref="-9000 -350" solimp="0.8 0.97 0.002 0.3 3" margin="0.002" solmix="3"/>')
    xml = xml.replace(rfoot, rfoot[:-2] + ' solref="0.015 1.1" solimp="0.85 0.96 0.0015 0.4 2.5" solmix="1"/>')
    xml = xml.replace(lfoot, lfoot[:-2] + ' solref="0.03 0.9" solimp="0.9 0.99 0.001 0.5 1" priority="2"/>')
    path = tmp_path / "jvrc_params.xml"
    path.write_text(xml)
    spec = JvrcWalkSpec(xml_path=str(path))
    n = 2
    env = emu.make_emulated(spec, n, seed=8)
    orc = [OracleJvrcWalkEnv(spec, seed=8, env_id=i) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    assert all(o.sim.ncon >= 4 for o in orc)
    tape = (np.random.default_rng(3).normal(size=(4, n, 12)) * 0.2).astype(np.float32)
    _run_tape(env, orc, tape)


def test_emulated_disable_flags(tmp_path):
    """<flag eulerdamp / refsafe / warmstart = "disable">: the three mjOption disable flags the kernel honours (explicit joint
    damping in the integrator, no clamp of solref's time constant, cold-started Newton) -- never set by the shipped models."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML, JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    xml = open(JVRC_STANDIN_XML).read()
    opt = '<option timestep="0.001"/>'
    assert opt in xml
    path = tmp_path / "jvrc_flags.xml"
    path.write_text(xml.replace(opt, '<option timestep="0.001"><flag eulerdamp="disable" refsafe="disable" warmstart="disable"/></option>'))
    spec = JvrcWalkSpec(xml_path=str(path))
    assert spec.model().disableflags != 0
    n = 2
    env = emu.make_emulated(spec, n, seed=9)
    orc = [OracleJvrcWalkEnv(spec, seed=9, env_id=i) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    tape = (np.random.default_rng(4).normal(size=(4, n, 12)) * 0.2).astype(np.float32)
    _run_tape(env, orc, tape)


def test_emulated_slide_joint_and_contact_gap(tmp_path):
    """A prismatic joint inside the tree (the kinematics / cdof / integration paths for mjJNT_SLIDE, which no shipped model
    uses in the humanoid kernels) and a contact gap (contacts detected inside margin but excluded from the constraint set
    until dist < margin - gap)."""
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML, JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    xml = open(JVRC_STANDIN_XML).read()
    knee = '<joint name="R_KNEE" type="hinge" axis="0 1 0" range="0 2.44"/>'
    floor = '<geom name="floor" type="plane" size="0 0 0.25" contype="1" conaffinity="0"/>'
    assert knee in xml and floor in xml
    xml = xml.replace(knee, '<joint name="R_KNEE" type="slide" axis="0.1 0 1" range="-0.05 0.4"/>')
    xml = xml.replace(floor, floor[:-2] + ' margin="0.004" gap="0.003"/>')
    path = tmp_path / "jvrc_slide.xml"
    path.write_text(xml)
    spec = JvrcWalkSpec(xml_path=str(path))
    n = 2
    env = emu.make_emulated(spec, n, seed=10)
    orc = [OracleJvrcWalkEnv(spec, seed=10, env_id=i) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    tape = (np.random.default_rng(6).normal(size=(4, n, 12)) * 0.2).astype(np.float32)
    _run_tape(env, orc, tape)
    assert any(o.sim.ncon > 0 and o.sim.nefc < 4 * o.sim.ncon + 40 for o in orc)


@pytest.mark.parametrize("task", ["jvrc_walk", "jvrc_step"])
def test_emulated_jvrc_init_noise(tmp_path, task):
    """init_noise in a JVRC YAML (BaseHumanoidEnv._apply_init_noise, envs/common/base_humanoid_env.py:260-263, 278-305: root z += U(0, .02),
    roll / pitch and every joint += U(-c, c)): every reset -- the explicit one and the auto-resets inside the control steps, which then
    compute the reset instead of copying the template -- draws its own pose; kernel == oracle draw for draw."""
    import yaml
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_BASE_YAML, JvrcWalkSpec
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    from oracle.env_jvrc_step import OracleJvrcStepEnv
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    cfg = yaml.safe_load(open(JVRC_BASE_YAML))
    cfg["init_noise"] = 3
    path = tmp_path / "jvrc_noise.yaml"
    path.write_text(yaml.safe_dump(cfg))
    S, O = (JvrcWalkSpec, OracleJvrcWalkEnv) if task == "jvrc_walk" else (JvrcStepSpec, OracleJvrcStepEnv)
    spec = S(yaml_path=str(path))
    assert spec.init_noise_deg == 3.0
    n, L = 3, 2
    env = emu.make_emulated(spec, n, seed=13, max_traj_len=L)
    orc = [O(spec, seed=13, env_id=i, max_traj_len=L) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    q, v = env.get_state()
    oq, ov = _states(orc)
    np.testing.assert_allclose(q, oq, rtol=0, atol=1e-12)
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    nominal = np.asarray(spec.nominal_pose)
    assert (np.abs(oq[:, 7:] - nominal[7:]).max(axis=1) > 1e-3).all() and np.abs(oq[0] - oq[1]).max() > 1e-3      # noised, and differently per env
    tape = (np.random.default_rng(2).normal(size=(5, n, 12)) * 0.1).astype(np.float32)
    for t in range(tape.shape[0]):            # episodes of two control steps: two auto-resets inside the tape
        o_dev, rew, done, tob = env.step(tape[t])
        res = [o.step_auto(tape[t, i]) for i, o in enumerate(orc)]
        q, v = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_array_equal(done, np.array([r[2] for r in res], dtype=np.uint8), err_msg=f"flags t={t}")
        np.testing.assert_allclose(q, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(v, ov, rtol=0, atol=1e-9, err_msg=f"qvel t={t}")
        np.testing.assert_allclose(o_dev, np.array([r[0] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"obs t={t}")
        np.testing.assert_allclose(rew, np.array([r[1] for r in res]), rtol=0, atol=2e-6, err_msg=f"rew t={t}")
    assert env.pop_fault_stats() == (0, 0)


@pytest.mark.parametrize("task", ["jvrc_walk", "jvrc_step"])
def test_emulated_jvrc_perturbation(tmp_path, task):
    """perturbation in a JVRC YAML (apply_perturbation, envs/common/domain_randomization.py:10-26 behind base_humanoid_env.py:86-92, 224-225):
    world-frame wrenches on two bodies, drawn after the observation with probability 1 / interval, a coin per body that clears all of them,
    cleared again by a reset.  The JVRC kernels keep them in the env's HBM record (no LDS copy); kernel == oracle draw for draw."""
    import yaml
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_BASE_YAML, JvrcWalkSpec
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    from oracle.env_jvrc_step import OracleJvrcStepEnv
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    cfg = yaml.safe_load(open(JVRC_BASE_YAML))
    cfg["perturbation"] = dict(enable=True, interval=2 * cfg["control_dt"], bodies=["PELVIS_S", "R_KNEE_S"], force_magnitude=60.0, torque_magnitude=8.0)
    path = tmp_path / "jvrc_perturb.yaml"
    path.write_text(yaml.safe_dump(cfg))
    S, O = (JvrcWalkSpec, OracleJvrcWalkEnv) if task == "jvrc_walk" else (JvrcStepSpec, OracleJvrcStepEnv)
    spec = S(yaml_path=str(path))
    assert spec.perturb_interval == 2 and spec.perturbation_config()["bodies"] == [spec.model().body_id("PELVIS_S"), spec.model().body_id("R_KNEE_S")]
    n, L = 3, 4
    env = emu.make_emulated(spec, n, seed=17, max_traj_len=L)
    orc = [O(spec, seed=17, env_id=i, max_traj_len=L) for i in range(n)]
    obs = env.reset().copy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    tape = (np.random.default_rng(4).normal(size=(9, n, 12)) * 0.1).astype(np.float32)
    pushed = 0
    for t in range(tape.shape[0]):
        o_dev, rew, done, tob = env.step(tape[t])
        res = [o.step_auto(tape[t, i]) for i, o in enumerate(orc)]
        pushed += sum(bool(np.abs(o.sim.xfrc_applied).max() > 0) for o in orc)
        q, v = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_array_equal(done, np.array([r[2] for r in res], dtype=np.uint8), err_msg=f"flags t={t}")
        np.testing.assert_allclose(q, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(v, ov, rtol=0, atol=1e-9, err_msg=f"qvel t={t}")
        np.testing.assert_allclose(o_dev, np.array([r[0] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"obs t={t}")
        np.testing.assert_allclose(rew, np.array([r[1] for r in res]), rtol=0, atol=2e-6, err_msg=f"rew t={t}")
    assert pushed >= 3, "no env carried an applied wrench through a control step"
    assert env.pop_fault_stats() == (0, 0)


def _cylinder_case(spec, env, orc):
    """shared with tests/test_model_variants_gpu.py: fallen poses on cylinder shanks, two control steps each against the oracle"""
    from tests.cyl_variant import contact_kinds, cylinder_poses
    m = spec.model()
    N = len(orc)
    env.reset()
    for o in orc:
        o.reset()
    q = cylinder_poses(spec, orc[0], N)
    v = np.random.default_rng(4).normal(size=(N, m.nv)) * 0.2
    env.set_state(q, v)
    for i, o in enumerate(orc):
        o.set_state(q[i], v[i])
    act = (np.random.default_rng(6).normal(size=(2, N, 12)) * 0.2).astype(np.float32)
    kinds = set()
    for t in range(2):
        obs, rew, done, _ = env.step(act[t])
        res = [o.step(act[t, i]) for i, o in enumerate(orc)]
        for o in orc:
            kinds |= contact_kinds(m, o.sim)
        gq, gv = env.get_state()
        oq, ov = _states(orc)
        np.testing.assert_allclose(gq, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(gv, ov, rtol=0, atol=1e-8, err_msg=f"qvel t={t}")
        np.testing.assert_allclose(obs, np.array([r[0] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"obs t={t}")
        np.testing.assert_array_equal(done & 1, np.array([int(r[2]) for r in res], dtype=np.uint8))
        env.set_state(oq, ov)
        for o in orc:
            o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    assert {(0, 5), (2, 5), (0, 4)} <= kinds, kinds
    assert env.pop_fault_stats() == (0, 0)


def test_emulated_cylinder_geoms(tmp_path):
    """cylinder shanks and ellipsoid thighs: the plane-cylinder, sphere-cylinder and plane-ellipsoid narrow phases of the kernels against the
    oracle's"""
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    from tests.cyl_variant import cylinder_spec
    spec = cylinder_spec(tmp_path)
    n = 6
    env = emu.make_emulated(spec, n, seed=3)
    orc = [OracleJvrcWalkEnv(spec, seed=3, env_id=i) for i in range(n)]
    _cylinder_case(spec, env, orc)
