"""Run only the jvrc_walk control-step kernel (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = JvrcWalkSpec()
if len(sys.argv) > 3:      # sub-steps per control step (default 25): the difference of two runs isolates the per-sub-step counters
    FS = int(sys.argv[3])

    class _Spec(JvrcWalkSpec):
        frame_skip = property(lambda self: FS)

    spec = _Spec()
env = spec.make_batched(N, seed=1, device=0, max_traj_len=400)
env.reset()
act = torch.randn(N, 12, device="cuda") * 0.1
for _ in range(steps): env.step(act)
torch.cuda.synchronize()
