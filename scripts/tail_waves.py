"""Which envs make the slowest waves of a control-step launch?  Random-policy regime (as in the bench's first iterations): step
4096 envs, then print, for the slowest waves of one launch, what their envs were doing (episode end / reset in that step, root
height, self-collision, foot contacts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
spec = JvrcWalkSpec()
N = 4096
env = spec.make_batched(N, seed=0, device=0, max_traj_len=400)
env.reset()
env.enable_task_inputs(True)
env.wave_cycles()
g = torch.Generator(device="cuda"); g.manual_seed(0)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    a = torch.randn(N, 12, device="cuda", generator=g) * 0.223
    obs, rew, done, tob = env.step(a)
torch.cuda.synchronize()
c = env.wave_cycles().astype(float)
ti = env.get_task_inputs()
q, _ = env.get_state()
w = c.reshape(-1, 2).max(1)
d = done.cpu().numpy().reshape(-1, 2)
z = ti["root_xpos"][:, 2].reshape(-1, 2)
sc = ti["self_collision"].reshape(-1, 2)
fc = ti["foot_contact"].reshape(-1, 2) if "foot_contact" in ti else None
print(f"waves {len(w)}  mean {w.mean():.3e}  p50 {np.percentile(w,50):.3e}  p90 {np.percentile(w,90):.3e}  p99 {np.percentile(w,99):.3e}  max {w.max():.3e}  max/mean {w.max()/w.mean():.2f}")
order = np.argsort(-w)
qv = np.abs(ti["qvel"]).max(1).reshape(-1, 2)
grf = (ti["grf_r"] + ti["grf_l"]).reshape(-1, 2)
fcn = ti["foot_contact"].reshape(-1, 2) if "foot_contact" in ti else np.zeros_like(grf)
print("slowest waves: cycles/mean, done flags of the two envs, root z (pre-reset terminal z), self-collision flags, max |qvel|, GRF, foot-contact flag")
for i in order[:16]:
    print(f"  wave {i:5d}  {w[i]/w.mean():.2f}  done {d[i]}  z {np.round(z[i],2)}  selfcol {sc[i]}  qvel {np.round(qv[i],1)}  grf {np.round(grf[i])}  fc {fcn[i]}")
dd = d.max(1) > 0
print(f"waves with an episode end: {dd.sum()} of {len(w)}: mean {w[dd].mean()/w.mean():.2f} x mean;  without: {w[~dd].mean()/w.mean():.2f} x, max without {w[~dd].max()/w.mean():.2f} x")
ss = sc.max(1) > 0
print(f"waves with a self-collision: {ss.sum()}: mean {w[ss].mean()/w.mean() if ss.any() else 0:.2f} x; low root (z<0.5): {(z.min(1)<0.5).sum()} waves, mean {w[z.min(1)<0.5].mean()/w.mean() if (z.min(1)<0.5).any() else 0:.2f} x")

lo = order[-8:]
print("fastest waves:")
for i in lo:
    print(f"  wave {i:5d}  {w[i]/w.mean():.2f}  done {d[i]}  z {np.round(z[i],2)}  qvel {np.round(qv[i],1)}  grf {np.round(grf[i])}  fc {fcn[i]}")
both_air = (grf.max(1) == 0)
print(f"waves with both envs airborne (no GRF): {both_air.sum()}, mean {w[both_air].mean()/w.mean() if both_air.any() else 0:.2f} x;  both on the ground: {(grf.min(1)>0).sum()}, mean {w[grf.min(1)>0].mean()/w.mean():.2f} x")
