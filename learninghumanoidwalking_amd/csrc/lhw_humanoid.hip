// placeholder until the wave-per-env stepper lands (next commit)
#include "lhw_internal.h"
struct HumanoidEnv { double* ep_stats; };
int humanoid_create(HumanoidEnv**, const std::vector<int32_t>&, const std::vector<double>&, const LhwEnvConfig*, int*, int*, int*) {
  return lhw_fail(LHW_ERR_UNSUPPORTED, "humanoid stepper not built yet");
}
void humanoid_destroy(HumanoidEnv*) {}
void humanoid_reset(HumanoidEnv*, const uint8_t*, float*, hipStream_t) {}
void humanoid_step(HumanoidEnv*, const float*, float*, float*, float*, uint8_t*, float*, hipStream_t) {}
void humanoid_get_state(HumanoidEnv*, double*, double*, hipStream_t) {}
void humanoid_set_state(HumanoidEnv*, const double*, const double*, hipStream_t) {}
double* humanoid_ep_stats(HumanoidEnv* h) { return h->ep_stats; }
void humanoid_set_iteration(HumanoidEnv*, int64_t) {}
