"""The stepping task's KNOWN DEVIATION, measured and bounded: outside FORWARD mode the reference leaves
the 20 terrain boxes coplanar with the floor (tasks/stepping_task.py:320-334), so a foot rests on the floor and on every box
under it; the kernels and the oracle env they are held to keep the boxes out of the collision set in those modes.  The oracle,
which has no lane limit (64 contacts), is stepped both ways from the same reset under PD-hold (scripts/a14_deviation.py): this
test keeps the size of the deviation on record -- it must stay a perturbation of the support (more, redundant contacts on the
same plane), not a different task."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_coplanar_boxes_change_contact_multiplicity_not_the_task():
    spec = importlib.util.spec_from_file_location("a14_deviation", os.path.join(ROOT, "scripts", "a14_deviation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.measure(T=40)
    assert set(res) == {"CURVED", "STANDING", "BACKWARD", "LATERAL"}
    for name, r in res.items():
        shipped, full = r["contacts"]
        assert shipped == 8 and 16 <= full <= 64, (name, r["contacts"])          # both feet flat: 8 floor contacts; the boxes add 8 .. 56
        assert r["dqpos"] < 5e-2 and r["dz"] < 5e-3, (name, r["dqpos"], r["dz"])  # same pose to within millimetres / hundredths of a radian
        assert r["dreward"] < 1e-2, (name, r["dreward"])                          # reward per step (of ~0.5) differs below 1e-2
        # the second effect: the reward's GRF counts floor contacts only (robot_interface.py:278-283) -- the boxes take over part
        # of the 608 N the floor carries alone under the shipped rule
        assert 590 < r["grf"][0] < 625 and 10 < r["grf"][1] < 0.6 * r["grf"][0], (name, r["grf"])
