"""CPU checks of the stepping-task oracle (oracle/env_jvrc_step.py, box-box narrow phase in oracle/mjc_oracle.c):
* task logic pinned against the REFERENCE's own tasks/stepping_task.py, executed in the build container on scripted
  kinematics with forced random draws (tests/golden/gen_golden.py::gen_stepping -> tests/golden/stepping.npz);
* box-box contacts on analytic invariants (resting force = weight, contact counts, edge-edge normal)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "stepping.npz")
MODE_U = {0: 0.1, 1: 0.17, 2: 0.3, 3: 0.5, 4: 0.9}


class FakeSim:
    """Kinematic state only: what SteppingTask reads through RobotInterface."""

    def __init__(self, nbody, nsite):
        self.xpos = np.zeros((nbody, 3))
        self.xquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
        self.site_xpos = np.zeros((nsite, 3))
        self.vel = {}

    def object_velocity(self, body, local):
        return np.concatenate([np.zeros(3), self.vel[int(body)]])

    def repack(self):
        pass


@pytest.fixture(scope="module")
def env():
    from oracle.env_jvrc_step import make_oracle_jvrc_step
    return make_oracle_jvrc_step(seed=0, env_id=0)


def test_task_logic_matches_reference_stepping_task(env):
    g = np.load(GOLD)
    e = env
    m = e.m
    real_sim = e.sim
    seen_modes, n_updates, n_done = set(), 0, 0
    try:
        for n in range(int(g["n"])):
            pre = f"r{n}_"
            inp = g[pre + "in"]
            mode_idx, phase_half, itr, choice, first_u, cflat = int(inp[0]), bool(inp[1]), int(inp[2]), int(inp[3]), inp[4], int(inp[5])
            sim = FakeSim(m.nbody, m.nsite)
            sim.xpos[e.root], sim.xpos[e.lfoot], sim.xpos[e.rfoot] = inp[7:10], inp[10:13], inp[13:16]
            sim.xquat[e.root] = inp[16:20]
            e.sim = sim
            e.spec.plans, e.iteration_count = [g[pre + "plan"]], itr
            e._task_reset(0, draws=dict(phase_half=phase_half, mode_u=MODE_U[mode_idx], choice=choice, plan=0, first_u=first_u, cflat=cflat))
            ref_seq, st = g[pre + "sequence"], g[pre + "state"]
            assert e.mode == mode_idx and e.nseq == len(ref_seq)
            np.testing.assert_allclose(e.sequence[: e.nseq], ref_seq, rtol=0, atol=1e-13, err_msg=f"sequence case {n}")
            assert [e.phase, e.period, e.t1, e.t2, e.spec.delay_frames, e.spec.target_radius] == list(st)
            np.testing.assert_array_equal(m.body_pos[e.floor_body], g[pre + "floor"])
            ref_pos, ref_quat = g[pre + "boxpos"], g[pre + "boxquat"]
            # identical terrain in EVERY walk mode: the sequence's boxes under their steps (coplanar with the floor outside
            # FORWARD mode, stepping_task.py:320-334), the unused ones sunk
            np.testing.assert_allclose(m.body_pos[e.box_body], ref_pos, rtol=0, atol=1e-13)
            np.testing.assert_allclose(m.body_quat[e.box_body], ref_quat, rtol=0, atol=1e-13)
            if mode_idx != 4:
                assert np.allclose(ref_pos[: e.nseq, 2], -0.1)
            seen_modes.add(mode_idx)
            # ---- scripted control steps
            kin, goal, rew, done, tst = g[pre + "kin"], g[pre + "goal"], g[pre + "rew"], g[pre + "done"], g[pre + "tstate"]
            for t in range(len(kin)):
                k = kin[t]
                sim.xpos[e.root], sim.xpos[e.head] = k[0:3], k[3:6]
                sim.site_xpos[e.lsite], sim.site_xpos[e.rsite] = k[6:9], k[9:12]
                sim.xquat[e.root] = k[12:16]
                sim.vel = {e.lfoot: k[16:19], e.rfoot: k[19:22]}
                lgrf, rgrf, selfcol, cz, hascon = k[22:27]
                e._grf = lambda foot, lg=lgrf, rg=rgrf: lg if foot == e.lfoot else rg
                e._foot_floor_contacts = lambda foot, z=cz, has=hascon: ([(0, dict(pos=np.array([0, 0, z])))] if (has and foot == e.rfoot) else [])
                e._self_collision = lambda sc=selfcol: bool(sc)
                t1_before = e.t1
                e._task_step()
                terms = e._calc_reward(None, None, None)
                np.testing.assert_allclose(e.goal, goal[t], rtol=0, atol=1e-12, err_msg=f"goal case {n} t={t}")
                np.testing.assert_allclose([terms[k2] for k2 in e.TERMS], rew[t], rtol=1e-13, atol=1e-15, err_msg=f"rewards case {n} t={t}")
                assert int(e._done()) == done[t]
                assert [e.phase, e.t1, e.t2, int(e.target_reached), e.target_reached_frames] == list(tst[t])
                n_updates += e.t1 != t1_before
                n_done += int(done[t])
            for name in ("_grf", "_foot_floor_contacts", "_self_collision"):
                del e.__dict__[name]
    finally:
        e.sim = real_sim
    assert seen_modes == {0, 1, 2, 3, 4} and n_updates > 10 and n_done > 10


def test_oracle_env_runs_and_reaches_first_target(env):
    """Standing still on the terrain: the first target lies under a foot, so after delay_frames control steps the
    target index advances; observation layout (jvrc_step.py:66-77)."""
    from oracle.env_jvrc_step import make_oracle_jvrc_step
    e = make_oracle_jvrc_step(seed=3, env_id=1)       # FORWARD mode for this key
    obs = e.reset()
    assert e.mode == 4 and obs.shape == (39,) and np.all(obs[31:] == 0)
    assert e.m.body_pos[e.floor_body][2] == -2.0
    for t in range(e.spec.delay_frames + 2):
        obs, r, done, terms = e.step(np.zeros(12, np.float32))
        assert not done and e.sim.ncon == 8           # two feet, four box corners each; the floor is 2 m below
    assert e.t1 == 1 and e.t2 == 2
    assert abs(sum(terms.values()) - r) < 1e-12 and list(terms) == e.TERMS
    np.testing.assert_allclose(obs[29] ** 2 + obs[30] ** 2, 1.0, atol=1e-12)
    assert np.all(np.abs(obs[31:35]) < 1.0) and np.all(np.abs(obs[35:37] + 0.8) < 0.05)      # targets ~0.8 m below the root


BOX = ("<mujoco><option timestep='0.001'/><worldbody><body name='stair' pos='0 0 0.1'>"
       "<geom name='stair' type='box' size='.15 1 .1'/></body>%s</worldbody></mujoco>")


def _sim(body_xml):
    from learninghumanoidwalking_amd import mjcf
    from oracle.physics import OracleSim
    return OracleSim(mjcf.compile_string(BOX % body_xml))


def test_box_resting_on_box_carries_its_weight():
    s = _sim("<body pos='0.02 0.1 0.215'><freejoint/><geom type='box' size='.1 .05 .01' mass='3'/></body>")
    s.step(600)
    assert s.ncon == 4
    f = sum(s.contact_force(i) for i in range(s.ncon))
    np.testing.assert_allclose(f[0], 3 * 9.81, rtol=1e-6)
    pts = np.array([s.contact(i)["pos"] for i in range(4)])
    np.testing.assert_allclose(sorted(pts[:, 0]), [-0.08, -0.08, 0.12, 0.12], atol=1e-6)   # the four corners of the small box
    for i in range(4):
        np.testing.assert_allclose(s.contact(i)["frame"][0], [0, 0, 1], atol=1e-9)        # geom1 (upper... lower id) -> geom2
    assert abs(s.qpos[2] - 0.21) < 2e-4 and np.abs(s.qvel).max() < 1e-6


def test_box_overhanging_an_edge_is_clipped_and_stays():
    s = _sim("<body pos='0.12 0.1 0.2095' euler='0 0 30'><freejoint/><geom type='box' size='.1 .05 .01' mass='3'/></body>")
    s.step(1)
    pts = np.array([s.contact(i)["pos"] for i in range(s.ncon)])
    assert 3 <= s.ncon <= 4 and np.all(pts[:, 0] <= 0.15 + 1e-9)                            # nothing beyond the stair edge
    s.step(500)
    f = sum(s.contact_force(i) for i in range(s.ncon))
    np.testing.assert_allclose(f[0], 3 * 9.81, rtol=0.2)                                    # the solver's softness lets it rock a little
    assert s.qpos[2] > 0.2


def test_edge_edge_contact_normal_is_perpendicular_to_both_edges():
    s = _sim("<body pos='0.2 0.0 0.36' euler='45 35 0'><freejoint/><geom type='box' size='.1 .1 .1' mass='3'/></body>")
    for k in range(300):
        s.step()
        if s.ncon:
            break
    assert s.ncon == 1
    c = s.contact(0)
    n = c["frame"][0]
    assert abs(n[1]) < 1e-9 and n[2] > 0.5 and c["dist"] < 0             # stair edge runs along y: the normal has no y component
    s.step(300)
    assert np.isfinite(s.qpos).all()
