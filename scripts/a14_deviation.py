"""How much does the stepping task's KNOWN DEVIATION matter?  (CPU oracle only.)  Outside FORWARD mode the reference leaves the
20 terrain boxes coplanar with the floor (tasks/stepping_task.py:320-334), so a foot rests on the floor AND on every box under
it; the kernels (and the oracle env they are checked against) keep the boxes out of the collision set in those modes.  This
script steps the ORACLE both ways from the same reset -- boxes sunk (shipped rule) vs boxes where the reference puts them (the
oracle has no lane limit: 64 contacts) -- under PD-hold, and prints what differs: contact count, root height, joint angles,
the GRF the reward reads (floor contacts only, robot_interface.py:278-283) and the six reward terms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
from oracle import env_jvrc_step as es

names = {es.CURVED: "CURVED", es.STANDING: "STANDING", es.BACKWARD: "BACKWARD", es.LATERAL: "LATERAL"}


def measure(T=40):
    """-> {mode name: dict(contacts=(shipped, full), dqpos, dz, grf=(shipped, full), dterm={term: max diff}, dreward)}"""
    spec = JvrcStepSpec()
    res = {}
    for mode_u, mode in ((0.1, es.CURVED), (0.17, es.STANDING), (0.3, es.BACKWARD), (0.5, es.LATERAL)):
        runs = {}
        for full in (False, True):
            env = es.OracleJvrcStepEnv(spec, seed=1, env_id=0)
            orig = env._draws
            env._draws = lambda c, o=orig: dict(o(c), mode_u=mode_u)
            env.reset()
            assert env.mode == mode
            if full:   # the reference's terrain: every box under its target step, top face at the step height (0 here)
                for k in range(es.NBOX):
                    st = env.sequence[k]
                    env.m.body_pos[env.box_body[k]] = st[0:3] - np.array([0, 0, 0.1])
                    env.m.body_quat[env.box_body[k]] = [np.cos(st[3] / 2), 0, 0, np.sin(st[3] / 2)]
                env.sim.repack()
            log = []
            for t in range(T):
                obs, r, done, terms = env.step(np.zeros(12, np.float32))
                log.append((env.sim.ncon, env.sim.qpos.copy(), env._grf(env.rfoot) + env._grf(env.lfoot), [terms[k] for k in env.TERMS], r))
            runs[full] = log
        a, b = runs[False], runs[True]
        dterm = np.max(np.abs(np.array([x[3] for x in a]) - np.array([y[3] for y in b])), axis=0)
        res[names[mode]] = dict(contacts=(max(x[0] for x in a), max(y[0] for y in b)),
                                dqpos=max(np.abs(x[1] - y[1]).max() for x, y in zip(a, b)),
                                dz=max(abs(x[1][2] - y[1][2]) for x, y in zip(a, b)),
                                grf=(a[-1][2], b[-1][2]), dterm=dict(zip(env.TERMS, dterm)),
                                dreward=max(abs(x[4] - y[4]) for x, y in zip(a, b)))
    return res


if __name__ == "__main__":
    for name, r in measure().items():
        print(f"{name:9s} contacts {r['contacts'][0]:2d} -> {r['contacts'][1]:2d}   max |dqpos| {r['dqpos']:.2e}  |d root z| {r['dz']:.2e}  "
              f"floor GRF at the last step: {r['grf'][0]:.1f} N vs {r['grf'][1]:.1f} N   max |d reward term| "
              f"{ {k: round(float(v), 5) for k, v in r['dterm'].items()} }   max |d reward| {r['dreward']:.2e}")
