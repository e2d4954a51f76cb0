"""Run only the persistent rollout kernel of jvrc_walk (for rocprofv3 counter passes): N envs x T control steps per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ["LHW_ROLLOUT_PERSISTENT"] = "2"
from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
from learninghumanoidwalking_amd.ppo import Rollout
from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spec = JvrcWalkSpec()
env = spec.make_batched(N, seed=1, device=0, max_traj_len=400)
k = PpoKernels(spec.obs_dim, spec.act_dim, hidden=256, max_rows=32768, device=0)
k.set_tensors(reference_init(spec.obs_dim, spec.act_dim, hidden=256, generator_seed=0))
k.set_obs_norm(spec.obs_mean, spec.obs_std)
ro = Rollout(env, k, T, seed=0)
assert ro.persistent
for _ in range(launches):
    ro.collect()      # one lhw_env_rollout launch (+ the critic's batched value passes, separate kernels)
torch.cuda.synchronize()
