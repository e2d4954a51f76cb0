"""Per-phase cycle breakdown of the wave-per-env stepper (env 0) under full load (4096 envs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "jvrc_walk"          # jvrc_walk | jvrc_step
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
spec = ENVIRONMENTS[name]()
env = spec.make_batched(N, seed=seed, device=0, max_traj_len=400)
env.reset()
if name == "jvrc_step":
    _, fz, _ = env.debug_step_record()
    print(f"jvrc_step seed {seed}: env 0 {'stands on the boxes (FORWARD mode)' if fz[0] != 0 else 'stands on the floor'}; {int((fz != 0).sum())}/{N} envs on boxes")
# "policy" regime of an untrained actor: zero-mean actions with the exploration std (0.223), redrawn every control step
# (LHW_PROFILE_STD overrides; 0.1 with a fixed draw was the quasi-static regime of round 1)
std = float(os.environ.get("LHW_PROFILE_STD", "0.223"))
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
def draw(): return torch.randn(N, 12, device="cuda", generator=gen) * std
for _ in range(30): env.step(draw())
env.phase_cycles(True)
steps = 20
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
acts = [draw() for _ in range(steps)]
torch.cuda.synchronize()
e0.record()
for a in acts: env.step(a)
e1.record(); torch.cuda.synchronize()
c = env.phase_cycles(True)
from learninghumanoidwalking_amd import _lib
print("occupancy blocks/CU (runtime query):", _lib.lib().lhw_debug_stepper_occupancy())
names = {0: "kinematics", 3: "collision", 1: "com / cinert / cdof", 5: "tree dynamics (rne, crb)", 4: "constraint rows", 11: "actuation + qfrc_smooth",
         12: "M solve", 7: "newton: cost / gradient passes", 2: "newton: Hessian + factor + solve", 15: "newton: line search", 13: "Euler solve",
         8: "integrate"}
sub = sum(c[k] for k in names) / steps / 25
print(f"N={N} ms/step {e0.elapsed_time(e1)/steps:.3f}  env0 cycles per sub-step {sub:.0f} (sum of the phases below; clock64 ticks)"
      f"  control-step prologue + epilogue per sub-step {(c[9] + c[10]) / steps / 25 - sub:.0f}")
for k, n in names.items(): print(f"  {n:34s} {c[k]/steps/25:9.0f} cyc/substep  {100*c[k]/steps/25/sub:5.1f}%")
print(f"  per control step (ticks): slot 9 (prologue + whatever the stage loop spends outside the marked phases) {c[9]/steps:.0f}, slot 10 (task step, reward, observation, reset, hand-over) {c[10]/steps:.0f}; all marked sub-step phases {sub*25:.0f}")
print(f"  newton passes per sub-step (env 0): {c[6]/steps/25:.2f}   line-search passes per sub-step: {c[14]/steps/25:.2f}")
