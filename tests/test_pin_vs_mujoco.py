"""Physics pin against a real MuJoCo -- runs only where `mujoco` is importable (it is not in this repository's image:
skipped there, so the oracle's parity with MuJoCo stays UNPINNED).  scripts/pin_vs_mujoco.py is the
same comparison as a command."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mujoco = pytest.importorskip("mujoco", reason="mujoco is not installed: the oracle cannot be pinned against MuJoCo here")


def _pin():
    spec = importlib.util.spec_from_file_location("pin_vs_mujoco", os.path.join(ROOT, "scripts", "pin_vs_mujoco.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("xml", ["cartpole.xml", "jvrc_standin.xml", "h1_standin.xml"])
def test_oracle_matches_mj_step(xml):
    ok, report = _pin().pin_model(os.path.join(ROOT, "learninghumanoidwalking_amd", "assets", xml), steps=300, tol=1e-9, verbose=False)
    assert ok, report
