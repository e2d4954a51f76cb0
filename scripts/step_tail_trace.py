"""jvrc_step: per-env cycles of consecutive control steps around episode ends (what makes a wave slow: the fall or the reset?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
N, T = 2048, 50
spec = ENVIRONMENTS["jvrc_step"]()
env = spec.make_batched(N, seed=0, device=0, max_traj_len=400)
env.reset(); env.wave_cycles()
g = torch.Generator(device="cuda"); g.manual_seed(0)
C, D, Z = [], [], []
for t in range(T):
    obs, rew, done, tob = env.step(torch.randn(N, 12, device="cuda", generator=g) * 0.223)
    torch.cuda.synchronize()
    C.append(env.wave_cycles().astype(float)); D.append(done.cpu().numpy().copy()); Z.append(env.get_state()[0][:, 2].copy())
C, D, Z = np.array(C), np.array(D), np.array(Z)
mean = C.mean()
print(f"mean cycles/env/step {mean:.3e}; per-step max / mean: {np.round(C.max(1)[10:30] / mean, 1)}")
ends = np.argwhere(D[5:] != 0)
rel = {k: [] for k in range(-4, 3)}
for t, i in ends:
    t += 5
    for k in rel:
        if 0 <= t + k < T: rel[k].append(C[t + k, i])
print("cycles relative to the mean around an episode end (step offset: mean, p90, max):")
for k in sorted(rel):
    a = np.array(rel[k]) / mean
    print(f"  {k:+d}: {a.mean():.2f} {np.percentile(a, 90):.2f} {a.max():.2f}  (n {len(a)})")
t, i = np.unravel_index(np.argmax(C[5:]), C[5:].shape); t += 5
print("slowest env-step:", t, i, f"{C[t, i] / mean:.1f}x", "done flags around:", D[max(0, t - 3):t + 2, i], "z:", np.round(Z[max(0, t - 3):t + 2, i], 2), "cycles:", np.round(C[max(0, t - 3):t + 2, i] / mean, 1))
