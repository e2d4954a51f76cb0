// Internal (non-ABI) interfaces between the translation units of liblhw.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/lhw.h"

int lhw_fail(int code, const char* fmt, ...);

// arguments of the persistent rollout kernel (device pointers; see humanoid_rollout_kernel)
struct RolloutArgs {
  int T, D, Dp, H, A, deterministic;
  unsigned counter0, env_id_base;      // first policy-noise counter (one per control step); global id of env 0 (noise key)
  unsigned long long seed;             // policy-noise seed
  const float *w1t, *b1, *w2t, *b2, *w3, *b3, *stds, *obs_mean, *obs_std;   // w1t [Dp][H], w2t [H][H] (k-major), w3 [A][H]
  float *obs, *act, *logp, *rew, *tob;  // [T+1][N][D], [T][N][A], [T][N], [T][N], [T][N][D]
  unsigned char* done;                  // [T][N]
  float* rew_terms;                     // [N][n_terms] of the last step (may be NULL)
};

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
int humanoid_supports_rollout(HumanoidEnv* h);
int humanoid_rollout_resident(HumanoidEnv* h);
int humanoid_rollout(HumanoidEnv* h, const RolloutArgs& ra, hipStream_t s);
double* humanoid_ep_stats(HumanoidEnv* h);
void humanoid_set_iteration(HumanoidEnv* h, int64_t it);
int humanoid_occupancy();
int humanoid_wave_cycles(HumanoidEnv* h, long long* out);
int humanoid_profile(HumanoidEnv* h, int enable, long long* out16);
int humanoid_actuator_state(HumanoidEnv* h, double* pos, double* vel, double* tq);
int humanoid_step_record(HumanoidEnv* h, double* seq, double* floor_z, int32_t* istate);
