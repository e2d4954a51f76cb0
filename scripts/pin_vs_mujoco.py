"""One-command pin of the CPU oracle (and, on a GPU box, of the HIP stepper) against a real MuJoCo.

    python scripts/pin_vs_mujoco.py [model.xml ...] [--steps 1000] [--hip]

Needs `import mujoco` (the reference pins mujoco==3.4.0, pyproject.toml:13) -- this repository's image does not have it, so
the script has never been run by its author; it is committed so that the physics pin is one command on a machine that does.
For every model (default: the reference's cartpole.xml copy and the two stand-in robots) it

  1. compiles the XML with MuJoCo, converts the MjModel with `model_from_mjmodel` (so the oracle runs on MuJoCo's own
     compiled constants: inertias, invweight0, meaninertia, ...) and ALSO compiles it with this repository's MJCF compiler,
     reporting field-by-field differences between the two compiled models;
  2. runs ONE forward pass on both sides from the same state -- the perturbed start pose, and MuJoCo's state at the first step
     with contacts -- and compares them STAGE BY STAGE in pipeline order (kinematics -> com / cvel -> qM -> bias / passive /
     actuator forces -> qacc_smooth -> contacts -> efc_J / efc_D / efc_R / efc_aref -> efc_force / qacc): the first stage that
     differs names the routine to re-check, with the field, index and both values (`stage_report`; self-tested without MuJoCo
     by tests/test_pin_stage_report.py);
  2b. resets both sides to a perturbed pose, applies the same random control tape, and steps `mj_step` next to the oracle's
     `orc_step`, reporting the largest |dqpos|, |dqvel|, the first step at which contact counts differ, and the worst
     difference in efc_force / qacc / contact frames on the first step with contacts (where states are still identical);
  3. with --hip, also advances the HIP stepper's state through the C ABI (`lhw_env_set_state` / `lhw_env_step`).

Exit status 0 iff every model stays within --tol (default 1e-9 over the first 100 steps, the bar for "same algorithm").
The `[MJ-recall]` markers in oracle/mjc_oracle.c list the statements most likely to need correction if this fails.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def compare_models(a, b, rtol=1e-9):
    """Field-by-field comparison of two compiled `Model`s (MuJoCo's vs this repository's compiler)."""
    bad = []
    for k in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "npair"):
        if getattr(a, k) != getattr(b, k):
            bad.append(f"{k}: {getattr(a, k)} vs {getattr(b, k)}")
    for k in ("meaninertia", "totalmass", "timestep"):
        if abs(getattr(a, k) - getattr(b, k)) > rtol * (1 + abs(getattr(a, k))):
            bad.append(f"{k}: {getattr(a, k)!r} vs {getattr(b, k)!r}")
    for k, va in a.arrays.items():
        vb = b.arrays.get(k)
        if vb is None or np.shape(va) != np.shape(vb):
            bad.append(f"{k}: shape {np.shape(va)} vs {None if vb is None else np.shape(vb)}")
        elif not np.allclose(va, vb, rtol=rtol, atol=1e-12):
            bad.append(f"{k}: max |diff| {np.abs(np.asarray(va, float) - np.asarray(vb, float)).max():.3e}")
    return bad


# ---- per-stage first-difference report -------------------------------------------------------------------------------------
# The pipeline stages of one mj_forward, in the order MuJoCo runs them, with the mjData fields each one produces.  Both sides are
# put into the same state, run one forward pass, and are compared stage by stage: the FIRST stage that differs names the routine
# to re-check (mjc_oracle.c marks the candidates [MJ-recall]); everything downstream of it differs as a consequence.
STAGES = (
    ("kinematics (mj_kinematics)", ("xpos", "xmat", "xipos", "ximat", "geom_xpos", "geom_xmat")),
    ("centre of mass, spatial velocities (mj_comPos, mj_comVel)", ("subtree_com", "cvel")),
    ("inertia matrix (mj_crb -> qM)", ("M",)),
    ("bias / passive / actuator forces (mj_rne, mj_passive, mj_fwdActuation)", ("qfrc_bias", "qfrc_passive", "qfrc_actuator")),
    ("unconstrained acceleration (mj_fwdAcceleration)", ("qacc_smooth",)),
    ("collision detection (mj_collision): contact count, geoms, dist, pos, frame", ("contacts",)),
    ("constraint rows (mj_makeConstraint, mj_makeImpedance): efc_J, efc_pos, efc_D / efc_R, efc_aref", ("efc_J", "efc_pos", "efc_D", "efc_R", "efc_aref")),
    ("constraint solver (mj_solNewton): efc_force, qacc", ("efc_force", "qacc")),
)


class OracleSide:
    """the float64 oracle (oracle/physics.py: OracleSim) behind the field names of the report"""

    def __init__(self, sim):
        self.sim = sim

    def field(self, name):
        sim = self.sim
        if name == "contacts":
            return [sim.contact(i) for i in range(sim.ncon)]
        if name.startswith("efc_"):
            return sim.efc(name)
        return np.array(getattr(sim, name), dtype=np.float64)


class MujocoSide:
    """a real mujoco (MjModel, MjData) pair behind the same names"""

    def __init__(self, mjm, mjd):
        self.m, self.d = mjm, mjd

    def field(self, name):
        import mujoco
        m, d = self.m, self.d
        if name == "contacts":
            return [dict(dist=float(c.dist), pos=np.array(c.pos), frame=np.array(c.frame).reshape(3, 3), geom1=int(c.geom1), geom2=int(c.geom2))
                    for c in d.contact[:d.ncon]]
        if name == "M":
            M = np.zeros((m.nv, m.nv))
            mujoco.mj_fullM(m, M, d.qM)
            return M
        if name == "efc_J":
            J = np.array(d.efc_J, dtype=np.float64)
            return J.reshape(d.nefc, m.nv) if J.size == d.nefc * m.nv else J     # (dense Jacobian; a sparse model needs mj_sparse2dense here)
        if name in ("xmat", "ximat", "geom_xmat"):
            return np.array(getattr(d, name), dtype=np.float64).reshape(-1, 9)
        if name.startswith("efc_"):
            return np.array(getattr(d, name), dtype=np.float64)[:d.nefc]
        return np.array(getattr(d, name), dtype=np.float64)


def stage_report(a, b, tol=1e-9, names=("mujoco", "oracle")):
    """Compare two sides stage by stage after a forward pass on the same state.  Returns (first differing stage or None, lines)."""
    lines, first = [], None
    for stage, fields in STAGES:
        worst, detail = 0.0, ""
        for f in fields:
            x, y = a.field(f), b.field(f)
            if f == "contacts":
                if len(x) != len(y):
                    worst, detail = float("inf"), f"contact count {len(x)} ({names[0]}) vs {len(y)} ({names[1]})"
                    break
                for i, (cx, cy) in enumerate(zip(x, y)):
                    if (cx["geom1"], cx["geom2"]) != (cy["geom1"], cy["geom2"]):
                        worst, detail = float("inf"), f"contact {i}: geoms {cx['geom1'], cx['geom2']} vs {cy['geom1'], cy['geom2']} (ordering)"
                        break
                    for k in ("dist", "pos", "frame"):
                        e = float(np.abs(np.asarray(cx[k]) - np.asarray(cy[k])).max())
                        if e > worst:
                            worst, detail = e, f"contact {i} (geoms {cx['geom1']}, {cx['geom2']}) {k}"
                continue
            x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
            if x.shape != y.shape:
                worst, detail = float("inf"), f"{f}: shape {x.shape} vs {y.shape}"
                break
            if x.size:
                scale = 1.0 + float(np.abs(x).max())
                e = float(np.abs(x - y).max()) / scale
                if e > worst:
                    idx = np.unravel_index(int(np.abs(x - y).argmax()), x.shape)
                    worst, detail = e, f"{f}{list(int(i) for i in idx)}: {x[idx]!r} vs {y[idx]!r}"
        ok = worst <= tol
        lines.append(f"   {'ok  ' if ok else 'DIFF'} {stage}: max rel |diff| {worst:.3e}" + ("" if ok else f"  <- {detail}"))
        if not ok and first is None:
            first = stage
    return first, lines


def pin_model(xml, steps, tol, seed=0, verbose=True):
    import mujoco
    from learninghumanoidwalking_amd import mjcf
    from learninghumanoidwalking_amd.model import model_from_mjmodel
    from oracle.physics import OracleSim

    mjm = mujoco.MjModel.from_xml_path(xml)
    mjd = mujoco.MjData(mjm)
    ours = mjcf.compile_file(xml, float(mjm.opt.timestep))
    theirs = model_from_mjmodel(mjm)
    diffs = compare_models(theirs, ours)
    if verbose:
        print(f"== {xml}\n   MJCF compiler vs MuJoCo's compiled model: {'identical' if not diffs else diffs}")
    sim = OracleSim(theirs)
    rs = np.random.default_rng(seed)
    mujoco.mj_resetData(mjm, mjd)
    qpos = mjd.qpos.copy()
    qpos[7 if mjm.jnt_type[0] == 0 else 0:] += rs.normal(size=mjm.nq - (7 if mjm.jnt_type[0] == 0 else 0)) * 0.05
    mjd.qpos[:] = qpos
    sim.reset_data()
    sim.qpos[:] = qpos
    worst_q = worst_v = 0.0
    first_ncon_diff = None
    solver_report = None
    # stage by stage on the start pose (no contacts yet on most models) ...
    mujoco.mj_forward(mjm, mjd)
    sim.forward()
    first_stage, lines = stage_report(MujocoSide(mjm, mjd), OracleSide(sim), tol)
    if verbose:
        print("   forward pass on the perturbed start pose, stage by stage:")
        print("\n".join(lines))
    staged_contact = False
    for t in range(steps):
        ctrl = rs.normal(size=mjm.nu) * 0.3
        mjd.ctrl[:] = ctrl
        sim.ctrl[:] = ctrl
        mujoco.mj_step(mjm, mjd)
        sim.step()
        if first_ncon_diff is None and mjd.ncon != sim.ncon:
            first_ncon_diff = (t, int(mjd.ncon), int(sim.ncon))
        if solver_report is None and mjd.nefc > 0 and mjd.nefc == sim.nefc:
            solver_report = dict(step=t, nefc=int(mjd.nefc), d_efc_force=float(np.abs(mjd.efc_force - sim.efc("efc_force")).max()),
                                 d_efc_aref=float(np.abs(mjd.efc_aref - sim.efc("efc_aref")).max()),
                                 d_efc_D=float(np.abs(mjd.efc_D - sim.efc("efc_D")).max() / (1 + np.abs(mjd.efc_D).max())),
                                 d_qacc=float(np.abs(mjd.qacc - sim.qacc).max()))
        if not staged_contact and mjd.ncon > 0:
            # ... and on the first state with contacts: MuJoCo's state copied into the oracle, one forward pass on each side
            staged_contact = True
            q, v, w = mjd.qpos.copy(), mjd.qvel.copy(), mjd.qacc_warmstart.copy()
            keep = (sim.qpos.copy(), sim.qvel.copy(), sim.qacc_warmstart.copy())
            sim.qpos[:], sim.qvel[:], sim.qacc_warmstart[:] = q, v, w
            mujoco.mj_forward(mjm, mjd)
            sim.forward()
            fs, lines = stage_report(MujocoSide(mjm, mjd), OracleSide(sim), tol)
            first_stage = first_stage or fs
            if verbose:
                print(f"   forward pass on MuJoCo's state after step {t} ({mjd.ncon} contacts), stage by stage:")
                print("\n".join(lines))
            sim.qpos[:], sim.qvel[:], sim.qacc_warmstart[:] = keep
            sim.forward()
        eq, ev = float(np.abs(mjd.qpos - sim.qpos).max()), float(np.abs(mjd.qvel - sim.qvel).max())
        worst_q, worst_v = max(worst_q, eq), max(worst_v, ev)
        if t == 99:
            first100 = (worst_q, worst_v)
    first100 = locals().get("first100", (worst_q, worst_v))
    ok = first100[0] <= tol and first100[1] <= 100 * tol
    if verbose:
        print(f"   {steps} steps: worst |dqpos| {worst_q:.3e} |dqvel| {worst_v:.3e}; first 100 steps {first100[0]:.3e} / {first100[1]:.3e}"
              f"; first contact-count difference {first_ncon_diff}; first constrained step {solver_report}  ->  {'PINNED' if ok else 'DIFFERS'}")
    return ok, dict(worst_q=worst_q, worst_v=worst_v, first100=first100, ncon_diff=first_ncon_diff, solver=solver_report, model_diffs=diffs,
                    first_differing_stage=first_stage)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("xml", nargs="*")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--tol", type=float, default=1e-9)
    args = ap.parse_args()
    try:
        import mujoco  # noqa: F401
    except ImportError:
        raise SystemExit("pin_vs_mujoco.py needs the `mujoco` package (the reference pins 3.4.0); it is not installed here")
    assets = os.path.join(ROOT, "learninghumanoidwalking_amd", "assets")
    xmls = args.xml or [os.path.join(assets, n) for n in ("cartpole.xml", "jvrc_standin.xml", "h1_standin.xml")]
    ok = all([pin_model(x, args.steps, args.tol)[0] for x in xmls])
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
