// Parameters / state of the lane-per-env cartpole kernel (see lhw_cartpole.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct CartpoleParams {
  int n_envs, frame_skip, max_traj_len, iterations;
  int warmstart, eulerdamp;
  uint32_t env_id_base;
  uint64_t seed;
  double mc, mp, l, Iyy;        // cart mass, pole mass, pole com distance from the hinge, pole inertia about y at its com
  double arm[2], damp[2];
  double gear, h, g;
  double range_lo, range_hi, margin;
  double solref[2], solimp[5];  // joint-limit solver parameters after getsolparam (refsafe + clamps)
  double invweight0;            // dof_invweight0 of the slider
  double meaninertia, tolerance;
  double kp, kd;
};

// struct-of-arrays state: d[field][N] with fields
// 0,1 qpos  2,3 qvel  4,5 qacc_warmstart  6 actuator_length  7 actuator_velocity  8 episode return
#define CARTPOLE_NFIELDS 9
struct CartpoleState {
  double* d;
  int32_t* traj_len;
  uint32_t* reset_count;
  double* ep_stats;  // [3]: sum of returns, sum of lengths, episode count
};

void cartpole_launch_reset(const CartpoleParams& p, const CartpoleState& st, const uint8_t* mask, float* obs, hipStream_t s);
void cartpole_launch_step(const CartpoleParams& p, const CartpoleState& st, const float* act, float* obs, float* term_obs,
                          float* rew, uint8_t* done, float* rew_terms, hipStream_t s);
void cartpole_launch_set_state(const CartpoleParams& p, const CartpoleState& st, const double* qpos, const double* qvel, hipStream_t s);
void cartpole_launch_get_state(const CartpoleParams& p, const CartpoleState& st, double* qpos, double* qvel, hipStream_t s);
