"""Sub-phase clock of one stepper phase (analysis build: scripts/build_variant.sh fineP -DLHW_FINEPROF=P, run with LHW_LIB=...):
python scripts/fine_phase_profile.py [N] [env]  -> ticks of env 0 per sub-step in each FINE_MARK slot of phase P."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "jvrc_walk"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = ENVIRONMENTS[name]().make_batched(N, seed=seed, device=0, max_traj_len=400)
env.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
def draw(): return torch.randn(N, env.act_dim, device="cuda", generator=gen) * 0.223
for _ in range(30): env.step(draw())
env.phase_cycles(True)
steps = 20
for _ in range(steps): env.step(draw())
c = env.phase_cycles(True)
print(os.path.basename(os.environ.get("LHW_LIB", "default")), name, "seed", seed, " ".join(f"[{i}] {c[i] / steps / 25:.0f}" for i in range(8)), "ticks / sub-step")
