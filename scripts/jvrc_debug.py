import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
np.set_printoptions(precision=6, suppress=False, linewidth=200)
from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
from oracle.env_jvrc_walk import OracleJvrcWalkEnv
spec = JvrcWalkSpec()
N, T = 4, 150
env = spec.make_batched(N, seed=9, device=0)
orc = [OracleJvrcWalkEnv(spec, seed=9, env_id=i) for i in range(N)]
env.reset(); [o.reset() for o in orc]
tape = (np.random.default_rng(1234).normal(size=(T, N, 12)) * 0.223).astype(np.float32)
m = spec.model()
for t in range(T):
    # step oracle substep by substep to log contact stats
    obs, rew, done, _ = env.step(torch.from_numpy(tape[t]).cuda())
    stats = []
    for i, o in enumerate(orc):
        r = o.step(tape[t, i])
        cons = [(m.geom_names[o.sim.contact(k)['geom1']], m.geom_names[o.sim.contact(k)['geom2']]) for k in range(o.sim.ncon)]
        stats.append((o.sim.ncon, o.sim.nefc, o.sim.niter, round(o.sim.qpos[2],3), bool(r[2])))
    q, v = env.get_state()
    oq = np.array([o.sim.qpos for o in orc]); ov = np.array([o.sim.qvel for o in orc])
    dq = np.abs(q-oq).max(1); dv = np.abs(v-ov).max(1)
    if t % 5 == 4 or dq.max() > 1e-9:
        print(t, 'dq', dq, 'dv', dv, stats)
    if dq.max() > 1e-6:
        for i, o in enumerate(orc):
            cons = sorted(set((m.geom_names[o.sim.contact(k)['geom1']], m.geom_names[o.sim.contact(k)['geom2']]) for k in range(o.sim.ncon)))
            print(' env', i, cons)
            lims = [(m.jnt_names[j], o.sim.qpos[m.jnt_qposadr[j]], m.jnt_range[j]) for j in range(1, m.njnt) if not (m.jnt_range[j][0] <= o.sim.qpos[m.jnt_qposadr[j]] <= m.jnt_range[j][1])]
            print('   limits violated', lims)
        break
    if t % 5 == 4:
        env.set_state(oq, ov)
        for o in orc: o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    fl = np.array([s[4] for s in stats])
    for i, o in enumerate(orc):
        if fl[i]: o.set_state(spec.nominal_pose, np.zeros(18))
    if fl.any():
        oq = np.array([o.sim.qpos for o in orc]); ov = np.array([o.sim.qvel for o in orc])
        env.set_state(oq, ov)
