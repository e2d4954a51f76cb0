// Fused cartpole control step: one environment per lane, struct-of-arrays state.
//
// Takes over, for a whole batch, the reference's CartpoleEnv.step / CartpoleRobot.step /
// _compute_reward / _check_termination / reset_model (reference
// envs/cartpole/cartpole_env.py:36-51,109-192) including the mj_step calls behind
// RobotInterface.step (reference envs/common/robot_interface.py:535-546).  The physics is the
// MuJoCo pipeline specialised to this model's topology (slide-x cart, hinge-y pole, one limited
// joint, no contacts): closed-form 2x2 CRBA/RNE, the soft joint-limit row with MuJoCo's
// impedance/reference-acceleration model, the primal Newton solver with exact line search, and
// Euler integration with implicit joint damping.  Float64 throughout: the
// reference's physics is float64 and the whole state is 8 doubles per env, so the kernel is
// bound by VALU latency, not HBM.
//
// Why lane-per-env here and wave-per-env for the humanoids: nv = 2 leaves nothing to
// parallelise inside one env, while a wavefront of 64 independent cartpoles keeps every lane busy.
#include <hip/hip_runtime.h>

#include "lhw_cartpole.h"
#include "lhw_rng.h"

#define CP_MINVAL 1e-15

struct CpRow {
  double J, pos, R, D, aref;
};

__device__ __forceinline__ double cp_impedance(const double* si, double pos, double margin) {
  if (si[0] == si[1] || si[2] <= CP_MINVAL) return 0.5 * (si[0] + si[1]);
  double x = fabs((pos - margin) / si[2]);
  if (x >= 1) return si[1];
  if (x <= 0) return si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
  else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
  return si[0] + y * (si[1] - si[0]);
}

// cost of the constraint rows at jar, optionally returning forces
__device__ __forceinline__ double cp_cost(const CpRow* row, int ne, const double* jar, double* force, int* active) {
  double c = 0;
  for (int r = 0; r < 2; r++) {
    if (r >= ne) break;
    double f = 0;
    int a = 0;
    if (jar[r] < 0) { f = -row[r].D * jar[r]; c += 0.5 * row[r].D * jar[r] * jar[r]; a = 1; }
    if (force) { force[r] = f; active[r] = a; }
  }
  return c;
}

// One mj_forward (+ optional Euler integration).  q,v,ws are updated in place when integrate != 0.
// Outputs the transmission fields of THIS forward pass (actuator_length/velocity), which are the stale
// values the reference's PD law reads during the next sub-step (note S).
__device__ void cp_forward(const CartpoleParams& p, double* q, double* v, double* ws, double force_x, int integrate,
                           double* act_len, double* act_vel) {
  const double c = cos(q[1]), s = sin(q[1]);
  const double M00 = p.mc + p.mp + p.arm[0], M01 = p.mp * p.l * c, M11 = p.Iyy + p.mp * p.l * p.l + p.arm[1];
  const double bias0 = -p.mp * p.l * s * v[1] * v[1], bias1 = -p.mp * p.g * p.l * s;
  *act_len = p.gear * q[0];
  *act_vel = p.gear * v[0];
  const double fs0 = -p.damp[0] * v[0] - bias0 + p.gear * force_x;
  const double fs1 = -p.damp[1] * v[1] - bias1;
  const double det = M00 * M11 - M01 * M01;
  double as0 = (M11 * fs0 - M01 * fs1) / det, as1 = (M00 * fs1 - M01 * fs0) / det;

  // joint-limit rows (mj_instantiateLimit): lower side first
  CpRow row[2];
  int ne = 0;
  for (int side = -1; side <= 1; side += 2) {
    double dist = side * ((side < 0 ? p.range_lo : p.range_hi) - q[0]);
    if (dist < p.margin) {
      double imp = cp_impedance(p.solimp, dist, p.margin);
      CpRow& r = row[ne++];
      r.J = -side;
      r.pos = dist;
      r.R = fmax(CP_MINVAL, (1 - imp) / imp * p.invweight0);
      r.D = 1 / r.R;
      double K, B;
      if (p.solref[0] > 0) {
        K = 1 / fmax(CP_MINVAL, p.solimp[1] * p.solimp[1] * p.solref[0] * p.solref[0] * p.solref[1] * p.solref[1]);
        B = 2 / fmax(CP_MINVAL, p.solimp[1] * p.solref[0]);
      } else {
        K = -p.solref[0] / fmax(CP_MINVAL, p.solimp[1] * p.solimp[1]);
        B = -p.solref[1] / fmax(CP_MINVAL, p.solimp[1]);
      }
      r.aref = -B * (r.J * v[0]) - K * imp * (dist - p.margin);
    }
  }

  double a0 = as0, a1 = as1, fc0 = 0; // qacc, qfrc_constraint (dof 0 only: rows act on the slider)
  if (ne > 0) {
    // ---- primal Newton (engine_solver.c), nv = 2
    double jar[2] = {0, 0}, frc[2] = {0, 0};
    int act[2] = {0, 0};
    if (p.warmstart) {
      a0 = ws[0]; a1 = ws[1];
      for (int r = 0; r < ne; r++) jar[r] = row[r].J * a0 - row[r].aref;
      double cw = cp_cost(row, ne, jar, nullptr, nullptr);
      double Ma0 = M00 * a0 + M01 * a1, Ma1 = M01 * a0 + M11 * a1;
      cw += 0.5 * ((Ma0 - fs0) * (a0 - as0) + (Ma1 - fs1) * (a1 - as1));
      for (int r = 0; r < ne; r++) jar[r] = row[r].J * as0 - row[r].aref;
      double cs = cp_cost(row, ne, jar, nullptr, nullptr);
      if (cw > cs) { a0 = as0; a1 = as1; }
    }
    const double scale = 1.0 / (p.meaninertia * 2.0);
    double cost = 0, oldcost;
    for (int iter = 0; iter <= p.iterations; iter++) {
      double Ma0 = M00 * a0 + M01 * a1, Ma1 = M01 * a0 + M11 * a1;
      for (int r = 0; r < ne; r++) jar[r] = row[r].J * a0 - row[r].aref;
      oldcost = cost;
      cost = cp_cost(row, ne, jar, frc, act);
      cost += 0.5 * ((Ma0 - fs0) * (a0 - as0) + (Ma1 - fs1) * (a1 - as1));
      fc0 = 0;
      for (int r = 0; r < ne; r++) fc0 += row[r].J * frc[r];
      double g0 = Ma0 - fs0 - fc0, g1 = Ma1 - fs1;
      double gn = sqrt(g0 * g0 + g1 * g1);
      if (iter > 0) {
        if (scale * (oldcost - cost) < p.tolerance || scale * gn < p.tolerance) break;
      } else if (scale * gn < p.tolerance) break;
      if (iter == p.iterations) break;
      double H00 = M00;
      for (int r = 0; r < ne; r++)
        if (act[r]) H00 += row[r].D * row[r].J * row[r].J;
      double hd = H00 * M11 - M01 * M01;
      double s0 = -(M11 * g0 - M01 * g1) / hd, s1 = -(H00 * g1 - M01 * g0) / hd;
      double Mv0 = M00 * s0 + M01 * s1, Mv1 = M01 * s0 + M11 * s1;
      double jv[2] = {0, 0};
      for (int r = 0; r < ne; r++) jv[r] = row[r].J * s0;
      const double qg1 = s0 * (Ma0 - fs0) + s1 * (Ma1 - fs1), qg2 = 0.5 * (s0 * Mv0 + s1 * Mv1);
      // exact line search: safeguarded Newton on the monotone piecewise-linear derivative
      double alpha = 0, lo = 0, hi = -1, d1, d2;
      {
        d1 = qg1; d2 = 2 * qg2;
        for (int r = 0; r < ne; r++)
          if (jar[r] < 0) { d1 += row[r].D * jar[r] * jv[r]; d2 += row[r].D * jv[r] * jv[r]; }
      }
      if (!(d1 >= 0 || d2 <= 0)) {
        const double d0 = fabs(d1);
        for (int it = 0; it < 40; it++) {
          double a = alpha - d1 / d2;
          if (hi >= 0 && (a <= lo || a >= hi)) a = 0.5 * (lo + hi);
          d1 = 2 * a * qg2 + qg1; d2 = 2 * qg2;
          for (int r = 0; r < ne; r++) {
            double x = jar[r] + a * jv[r];
            if (x < 0) { d1 += row[r].D * x * jv[r]; d2 += row[r].D * jv[r] * jv[r]; }
          }
          if (d1 < 0) lo = a; else hi = a;
          alpha = a;
          if (fabs(d1) <= 1e-14 * d0) break;
          if (hi >= 0 && hi - lo <= 4e-16 * hi) break;
        }
      }
      if (alpha == 0) break;
      a0 += alpha * s0; a1 += alpha * s1;
    }
  }
  ws[0] = a0; ws[1] = a1; // mj_fwdConstraint saves qacc as the next warm start
  if (!integrate) return;
  // mj_Euler: implicit in joint damping
  double n0 = a0, n1 = a1;
  if (p.eulerdamp && (p.damp[0] > 0 || p.damp[1] > 0)) {
    const double A00 = M00 + p.h * p.damp[0], A11 = M11 + p.h * p.damp[1];
    const double r0 = fs0 + fc0, r1 = fs1;
    const double dd = A00 * A11 - M01 * M01;
    n0 = (A11 * r0 - M01 * r1) / dd;
    n1 = (A00 * r1 - M01 * r0) / dd;
  }
  v[0] += p.h * n0; v[1] += p.h * n1;
  q[0] += p.h * v[0]; q[1] += p.h * v[1];
}

__device__ __forceinline__ void cp_obs(const double* q, const double* v, float* o) {
  o[0] = (float)q[0]; o[1] = (float)cos(q[1]); o[2] = (float)sin(q[1]); o[3] = (float)v[0]; o[4] = (float)v[1];
}

// reset_model (cartpole_env.py:109-121) after mj_resetData: 5 uniforms, then set_state's mj_forward
__device__ void cp_reset(const CartpoleParams& p, uint32_t genv, uint32_t reset_count, double* q, double* v, double* ws,
                         double* act_len, double* act_vel) {
  const double PI = 3.14159265358979323846;
  double pole = lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 0, -PI, PI);
  q[0] = 0.0 + lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 1, -0.1, 0.1);
  q[1] = pole + lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 2, -0.1, 0.1);
  v[0] = lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 3, -0.1, 0.1);
  v[1] = lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 4, -0.1, 0.1);
  ws[0] = ws[1] = 0;
  cp_forward(p, q, v, ws, 0.0, 0, act_len, act_vel); // actuation disabled
}

__global__ void __launch_bounds__(256) cartpole_reset_kernel(CartpoleParams p, CartpoleState st, const uint8_t* mask,
                                                             float* obs) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= p.n_envs) return;
  if (mask && !mask[n]) return;
  const int N = p.n_envs;
  double q[2], v[2], ws[2], al, av;
  uint32_t rc = st.reset_count[n];
  cp_reset(p, p.env_id_base + n, rc, q, v, ws, &al, &av);
  st.reset_count[n] = rc + 1;
  st.d[0 * N + n] = q[0]; st.d[1 * N + n] = q[1]; st.d[2 * N + n] = v[0]; st.d[3 * N + n] = v[1];
  st.d[4 * N + n] = ws[0]; st.d[5 * N + n] = ws[1]; st.d[6 * N + n] = al; st.d[7 * N + n] = av;
  st.d[8 * N + n] = 0.0; // episode return
  st.traj_len[n] = 0;
  if (obs) cp_obs(q, v, obs + 5 * n);
}

__global__ void __launch_bounds__(256) cartpole_step_kernel(CartpoleParams p, CartpoleState st, const float* __restrict__ act,
                                                            float* __restrict__ obs, float* __restrict__ term_obs,
                                                            float* __restrict__ rew, uint8_t* __restrict__ done_out,
                                                            float* __restrict__ rew_terms) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= p.n_envs) return;
  const int N = p.n_envs;
  double q[2] = {st.d[0 * N + n], st.d[1 * N + n]}, v[2] = {st.d[2 * N + n], st.d[3 * N + n]};
  double ws[2] = {st.d[4 * N + n], st.d[5 * N + n]};
  double al = st.d[6 * N + n], av = st.d[7 * N + n];
  double ep_ret = st.d[8 * N + n];
  int traj_len = st.traj_len[n];

  // np.clip(action, -0.8, 0.8) on the float32 action (cartpole_env.py:143)
  float a32 = act[n];
  a32 = fminf(fmaxf(a32, -0.8f), 0.8f);
  const double target = (double)a32;

  for (int k = 0; k < p.frame_skip; k++) {
    // RobotInterface.step_pd with the transmission fields of the previous forward pass
    double tau = p.kp * (target - al / p.gear) + p.kd * (0.0 - av / p.gear);
    cp_forward(p, q, v, ws, tau, 1, &al, &av); // ctrl = tau (cartpole does not divide by the gear)
  }

  float o[5];
  cp_obs(q, v, o);
  // _compute_reward (cartpole_env.py:155-187); python sum() order
  const double cosang = cos(q[1]);
  const double up = 0.35 * (1.0 + cosang) / 2.0 + 0.35 * exp(-2.0 * (1.0 - cosang) * (1.0 - cosang));
  const double center = 0.1 * exp(-2.0 * q[0] * q[0]);
  const double velr = 0.1 * exp(-0.05 * v[1] * v[1]);
  const double actr = 0.1 * exp(-1.0 * (double)(a32 * a32));
  const double r = ((0.0 + up) + center + velr) + actr;
  const bool terminated = fabs(q[0]) > 0.99;
  traj_len += 1;
  ep_ret += r;
  const bool truncated = p.max_traj_len > 0 && traj_len >= p.max_traj_len;
  uint8_t flags = (terminated ? 1u : 0u) | (truncated ? 2u : 0u);

  rew[n] = (float)r;
  done_out[n] = flags;
  if (rew_terms) {
    rew_terms[4 * n + 0] = (float)up; rew_terms[4 * n + 1] = (float)center;
    rew_terms[4 * n + 2] = (float)velr; rew_terms[4 * n + 3] = (float)actr;
  }
  if (term_obs)
    for (int k = 0; k < 5; k++) term_obs[5 * n + k] = o[k];

  if (p.max_traj_len > 0 && (terminated || truncated)) {
    atomicAdd(&st.ep_stats[0], ep_ret);
    atomicAdd(&st.ep_stats[1], (double)traj_len);
    atomicAdd(&st.ep_stats[2], 1.0);
    uint32_t rc = st.reset_count[n];
    cp_reset(p, p.env_id_base + n, rc, q, v, ws, &al, &av);
    st.reset_count[n] = rc + 1;
    traj_len = 0;
    ep_ret = 0;
    cp_obs(q, v, o);
  }
  for (int k = 0; k < 5; k++) obs[5 * n + k] = o[k];

  st.d[0 * N + n] = q[0]; st.d[1 * N + n] = q[1]; st.d[2 * N + n] = v[0]; st.d[3 * N + n] = v[1];
  st.d[4 * N + n] = ws[0]; st.d[5 * N + n] = ws[1]; st.d[6 * N + n] = al; st.d[7 * N + n] = av;
  st.d[8 * N + n] = ep_ret;
  st.traj_len[n] = traj_len;
}

// set_state parity hook: write qpos/qvel and re-run mj_forward with actuation disabled
__global__ void __launch_bounds__(256) cartpole_set_state_kernel(CartpoleParams p, CartpoleState st, const double* qpos,
                                                                 const double* qvel) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= p.n_envs) return;
  const int N = p.n_envs;
  double q[2] = {qpos[2 * n], qpos[2 * n + 1]}, v[2] = {qvel[2 * n], qvel[2 * n + 1]}, ws[2] = {0, 0}, al, av;
  cp_forward(p, q, v, ws, 0.0, 0, &al, &av);
  st.d[0 * N + n] = q[0]; st.d[1 * N + n] = q[1]; st.d[2 * N + n] = v[0]; st.d[3 * N + n] = v[1];
  st.d[4 * N + n] = ws[0]; st.d[5 * N + n] = ws[1]; st.d[6 * N + n] = al; st.d[7 * N + n] = av;
}

__global__ void __launch_bounds__(256) cartpole_get_state_kernel(CartpoleParams p, CartpoleState st, double* qpos, double* qvel) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= p.n_envs) return;
  const int N = p.n_envs;
  qpos[2 * n] = st.d[0 * N + n]; qpos[2 * n + 1] = st.d[1 * N + n];
  qvel[2 * n] = st.d[2 * N + n]; qvel[2 * n + 1] = st.d[3 * N + n];
}

void cartpole_launch_reset(const CartpoleParams& p, const CartpoleState& st, const uint8_t* mask, float* obs, hipStream_t s) {
  int blocks = (p.n_envs + 255) / 256;
  hipLaunchKernelGGL(cartpole_reset_kernel, dim3(blocks), dim3(256), 0, s, p, st, mask, obs);
}
void cartpole_launch_step(const CartpoleParams& p, const CartpoleState& st, const float* act, float* obs, float* term_obs,
                          float* rew, uint8_t* done, float* rew_terms, hipStream_t s) {
  int blocks = (p.n_envs + 255) / 256;
  hipLaunchKernelGGL(cartpole_step_kernel, dim3(blocks), dim3(256), 0, s, p, st, act, obs, term_obs, rew, done, rew_terms);
}
void cartpole_launch_set_state(const CartpoleParams& p, const CartpoleState& st, const double* qpos, const double* qvel, hipStream_t s) {
  int blocks = (p.n_envs + 255) / 256;
  hipLaunchKernelGGL(cartpole_set_state_kernel, dim3(blocks), dim3(256), 0, s, p, st, qpos, qvel);
}
void cartpole_launch_get_state(const CartpoleParams& p, const CartpoleState& st, double* qpos, double* qvel, hipStream_t s) {
  int blocks = (p.n_envs + 255) / 256;
  hipLaunchKernelGGL(cartpole_get_state_kernel, dim3(blocks), dim3(256), 0, s, p, st, qpos, qvel);
}
