"""Isolated timing of the control-step kernel (whole-batch launches, exploration-level random actions, auto-reset):
median ms per control step over chunks -- for A/B comparisons of kernel variants (LHW_LIB) on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learninghumanoidwalking_amd.envs import ENVIRONMENTS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "jvrc_walk"
spec = ENVIRONMENTS[name]()
env = spec.make_batched(N, seed=1, device=0, max_traj_len=400)
env.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
A = env.act_dim
acts = [torch.randn(N, A, device="cuda", generator=gen) * 0.223 for _ in range(64)]
for i in range(60): env.step(acts[i % 64])
torch.cuda.synchronize()
chunks = []
for c in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(25): env.step(acts[(c * 25 + i) % 64])
    e1.record(); torch.cuda.synchronize()
    chunks.append(e0.elapsed_time(e1) / 25)
print(f"{os.path.basename(os.environ.get('LHW_LIB', 'default'))}: {name} N={N} median {np.median(chunks):.4f} ms/step  min {min(chunks):.4f}  max {max(chunks):.4f}")
