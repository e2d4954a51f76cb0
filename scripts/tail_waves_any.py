"""Per-wave duration spread of one control-step launch for any humanoid env (random-policy regime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learninghumanoidwalking_amd import envs as lenvs
name, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 120
spec = lenvs.ENVIRONMENTS[name]()
env = spec.make_batched(N, seed=0, device=0, max_traj_len=400)
env.reset(); env.wave_cycles()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for t in range(T):
    obs, rew, done, tob = env.step(torch.randn(N, env.act_dim, device="cuda", generator=g) * 0.223)
torch.cuda.synchronize()
c = env.wave_cycles().astype(float)
d = done.cpu().numpy()
per_wave = 2 if getattr(env, "_two_per_wave", name != "jvrc_step") else 1
w = c.reshape(-1, per_wave).max(1); dd = d.reshape(-1, per_wave).max(1) > 0
print(f"{name}: waves {len(w)}  mean {w.mean():.3e}  p50/mean {np.percentile(w,50)/w.mean():.2f}  p90 {np.percentile(w,90)/w.mean():.2f}  p99 {np.percentile(w,99)/w.mean():.2f}  max {w.max()/w.mean():.2f};  "
      f"episode ends {dd.sum()}: mean {w[dd].mean()/w.mean() if dd.any() else 0:.2f} x, max {w[dd].max()/w.mean() if dd.any() else 0:.2f} x; others max {w[~dd].max()/w.mean():.2f} x; reruns {env.pop_rerun_count()}")
o = np.argsort(-w)[:6]
print("   slowest:", [(int(i), round(float(w[i] / w.mean()), 2), bool(dd[i])) for i in o])
