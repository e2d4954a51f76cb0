"""Generate assets/footstep_plans.txt: curved footstep plans for the stepping task's CURVED mode.

The reference ships pre-generated plans in utils/footstep_plans.txt (read by tasks/stepping_task.py:52-64: blocks of
"x,y,theta" lines separated by "---"; a block is kept when the NEXT separator is reached).  That file is data of the
reference and is not copied; this script writes plans in the same format from a constant-curvature walk: the body
centre advances by `stride` along its heading and turns by `dtheta` per step, feet alternate right/left at +-half_gap
from the centre line.  The statistics (7..18 steps per plan, ~0.2 rad turns) are chosen to resemble the reference's
plans; a real copy of the reference's file can be passed to JvrcStepSpec(plans_path=...).
"""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "learninghumanoidwalking_amd", "assets",
                   "footstep_plans.txt")


def make_plan(rng):
    n = int(rng.integers(7, 19))
    dth = float(rng.uniform(-np.pi / 16, np.pi / 16))
    stride = float(rng.uniform(0.12, 0.28))
    gap = float(rng.uniform(0.07, 0.14))
    x = y = th = 0.0
    side = -1.0                      # right foot first
    out = []
    for _ in range(n):
        out.append((x - np.sin(th) * side * gap, y + np.cos(th) * side * gap, th))
        th += dth
        x += stride * np.cos(th)
        y += stride * np.sin(th)
        side = -side
    return out


def main():
    rng = np.random.default_rng(20240501)
    lines = []
    for _ in range(110):
        lines.append("---")
        for x, y, th in make_plan(rng):
            lines.append(f"{x!r},{y!r},{th!r}".replace("np.float64(", "").replace(")", ""))
    lines.append("---")
    with open(OUT, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", OUT, len(lines), "lines")


if __name__ == "__main__":
    main()
