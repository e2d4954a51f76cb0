// The stepping task's instantiations of the resident rollout kernel (humanoid_rollout_kernel<TASK_STEP, 64, QUEUE>) as a translation unit of
// their own: the same source as lhw_humanoid_rollout.hip, compiled with LLVM's iterative ILP scheduling strategy (_lib.EXTRA_FLAGS), which suits the
// one-env-per-wave kernels (jvrc_step rollout -3 %, same box) and not the two-envs-per-wave ones: profiles/r06_stepper_compiler_flags.txt.
#define LHW_ROLLOUT_STEP_TU 1
#include "lhw_humanoid_rollout.hip"
