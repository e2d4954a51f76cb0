"""MI355X-native batched environment stepper + PPO for the humanoid-walking tasks of
rohanpsingh/LearningHumanoidWalking (cartpole, jvrc_walk ...), built from scratch for gfx950.

Only the hot path lives here: packed model + MJCF-subset compiler,
the HIP kernels behind a C ABI (``csrc/``, ``include/lhw.h``) and the Python host side that
mirrors the reference's env / PPO surface.
"""

__version__ = "0.1.0"
