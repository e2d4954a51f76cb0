// Internal (non-ABI) interfaces between the translation units of liblhw.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/lhw.h"

int lhw_fail(int code, const char* fmt, ...);

struct HumanoidEnv;
int humanoid_create(HumanoidEnv** out, const std::vector<int32_t>& mi, const std::vector<double>& md, const LhwEnvConfig* cfg,
                    int* obs_dim, int* act_dim, int* n_terms);
void humanoid_destroy(HumanoidEnv* h);
void humanoid_reset(HumanoidEnv* h, const uint8_t* mask, float* obs, hipStream_t s);
void humanoid_step(HumanoidEnv* h, const float* act, float* obs, float* term_obs, float* rew, uint8_t* done, float* rew_terms,
                   hipStream_t s);
int humanoid_step_range(HumanoidEnv* h, int first, int count, const float* act, float* obs, float* term_obs, float* rew, uint8_t* done,
                        float* rew_terms, hipStream_t s);
void humanoid_get_state(HumanoidEnv* h, double* qpos, double* qvel, hipStream_t s);
void humanoid_set_state(HumanoidEnv* h, const double* qpos, const double* qvel, hipStream_t s);
double* humanoid_ep_stats(HumanoidEnv* h);
void humanoid_set_iteration(HumanoidEnv* h, int64_t it);
int humanoid_occupancy();
int humanoid_wave_cycles(HumanoidEnv* h, long long* out);
int humanoid_profile(HumanoidEnv* h, int enable, long long* out16);
int humanoid_task_inputs(HumanoidEnv* h, int enable /* -1: leave */, double* out_host, double** out_dev);
int humanoid_actuator_state(HumanoidEnv* h, double* pos, double* vel, double* tq);
int humanoid_step_record(HumanoidEnv* h, double* seq, double* floor_z, int32_t* istate);
