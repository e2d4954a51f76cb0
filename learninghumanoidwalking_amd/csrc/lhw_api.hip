// extern "C" entry points of liblhw.so (declared in include/lhw.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lhw.h"
#include "lhw_cartpole.h"
#include "lhw_internal.h"

static thread_local std::string g_err;

int lhw_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

hipError_t lhw_malloc(void** p, size_t n) {
  static const bool poison = getenv("LHW_POISON") && atoi(getenv("LHW_POISON")) != 0;
  hipError_t e = hipMalloc(p, n);
  if (e == hipSuccess && poison && n) e = hipMemset(*p, 0xFF, n);
  return e;
}

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return lhw_fail(LHW_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); \
  } while (0)

struct LhwEnv {
  int task = 0, n_envs = 0, device = 0, nq = 0, nv = 0, nu = 0, obs_dim = 0, act_dim = 0, n_terms = 0;
  std::vector<int32_t> mi;
  std::vector<double> md;
  // cartpole
  CartpoleParams cp{};
  CartpoleState cps{};
  // humanoid
  HumanoidEnv* hum = nullptr;
  double* stage_q = nullptr;  // device staging for get/set state
  double* stage_v = nullptr;
};

extern "C" int lhw_version(void) { return 1; }
extern "C" const char* lhw_last_error(void) { return g_err.c_str(); }

static const int32_t* IFLD(const LhwEnv* e, int f) { return e->mi.data() + e->mi[LHW_IH_COUNT + f]; }
static const double* DFLD(const LhwEnv* e, int f) { return e->md.data() + e->mi[LHW_IH_COUNT + LHW_IF_COUNT + f]; }

static int check_model(const int32_t* mi, int64_t ni, const double* md, int64_t nd) {
  if (!mi || !md || ni < LHW_IH_COUNT + LHW_IF_COUNT + LHW_DF_COUNT || nd < LHW_DH_COUNT)
    return lhw_fail(LHW_ERR_MODEL, "model blobs too small");
  if ((uint32_t)mi[LHW_IH_MAGIC] != LHW_MODEL_MAGIC || mi[LHW_IH_VERSION] != LHW_MODEL_VERSION)
    return lhw_fail(LHW_ERR_MODEL, "bad model magic/version");
  if (mi[LHW_IH_N_IFIELDS] != LHW_IF_COUNT || mi[LHW_IH_N_DFIELDS] != LHW_DF_COUNT)
    return lhw_fail(LHW_ERR_MODEL, "model field table does not match this build");
  for (int f = 0; f < LHW_IF_COUNT; f++)
    if (mi[LHW_IH_COUNT + f] < 0 || mi[LHW_IH_COUNT + f] > ni) return lhw_fail(LHW_ERR_MODEL, "int field %d offset out of range", f);
  for (int f = 0; f < LHW_DF_COUNT; f++)
    if (mi[LHW_IH_COUNT + LHW_IF_COUNT + f] < 0 || mi[LHW_IH_COUNT + LHW_IF_COUNT + f] > nd)
      return lhw_fail(LHW_ERR_MODEL, "double field %d offset out of range", f);
  return LHW_OK;
}

// getsolparam (engine_core_constraint.c): refsafe + clamps
static void solparam(const LhwEnv* e, const double* sr, const double* si, double* solref, double* solimp) {
  const double h = e->md[LHW_DH_TIMESTEP];
  solref[0] = sr[0]; solref[1] = sr[1];
  for (int k = 0; k < 5; k++) solimp[k] = si[k];
  if (!(e->mi[LHW_IH_DISABLEFLAGS] & (1 << 11)) && solref[0] > 0 && solref[0] < 2 * h) solref[0] = 2 * h;
  solimp[0] = fmin(0.9999, fmax(0.0001, solimp[0]));
  solimp[1] = fmin(0.9999, fmax(0.0001, solimp[1]));
  solimp[2] = fmax(0.0, solimp[2]);
  solimp[3] = fmin(0.9999, fmax(0.0001, solimp[3]));
  solimp[4] = fmax(1.0, solimp[4]);
}

static int create_cartpole(LhwEnv* e, const LhwEnvConfig* cfg) {
  // the lane-per-env kernel is specialised to the slide-x / hinge-y topology of the reference's cartpole.xml
  const int32_t* jt = IFLD(e, LHW_IF_JNT_TYPE);
  const double* ax = DFLD(e, LHW_DF_JNT_AXIS);
  if (e->nq != 2 || e->nv != 2 || e->nu != 1 || e->mi[LHW_IH_NBODY] != 3 || jt[0] != 2 || jt[1] != 3)
    return lhw_fail(LHW_ERR_UNSUPPORTED, "cartpole task needs a 2-dof slide+hinge model");
  if (fabs(ax[0] - 1) > 1e-12 || fabs(ax[4] - 1) > 1e-12) return lhw_fail(LHW_ERR_UNSUPPORTED, "cartpole axes must be slide x / hinge y");
  const double* ipos = DFLD(e, LHW_DF_BODY_IPOS);
  const double* bpos = DFLD(e, LHW_DF_BODY_POS);
  const double* jpos = DFLD(e, LHW_DF_JNT_POS);
  const double* iq = DFLD(e, LHW_DF_BODY_IQUAT);
  for (int k = 0; k < 6; k++)
    if (fabs(bpos[3 + k]) > 1e-12 || fabs(jpos[k]) > 1e-12) return lhw_fail(LHW_ERR_UNSUPPORTED, "cartpole bodies/joints must sit at the origin");
  if (fabs(ipos[6]) > 1e-12 || fabs(ipos[7]) > 1e-12 || fabs(iq[8] - 1) > 1e-12)
    return lhw_fail(LHW_ERR_UNSUPPORTED, "pole inertial frame must lie on the local z axis, axis-aligned");
  if (e->mi[LHW_IH_NPAIR] != 0) return lhw_fail(LHW_ERR_UNSUPPORTED, "cartpole kernel has no contact support");
  if (cfg->frame_skip <= 0) return lhw_fail(LHW_ERR_ARG, "frame_skip must be positive");
  CartpoleParams& p = e->cp;
  const double* mass = DFLD(e, LHW_DF_BODY_MASS);
  const double* inertia = DFLD(e, LHW_DF_BODY_INERTIA);
  p.n_envs = e->n_envs; p.frame_skip = cfg->frame_skip; p.max_traj_len = cfg->max_traj_len;
  p.iterations = e->mi[LHW_IH_ITERATIONS];
  p.warmstart = !(e->mi[LHW_IH_DISABLEFLAGS] & (1 << 7));
  p.eulerdamp = !(e->mi[LHW_IH_DISABLEFLAGS] & (1 << 14));
  p.env_id_base = (uint32_t)cfg->env_id_base; p.seed = cfg->seed;
  p.mc = mass[1]; p.mp = mass[2]; p.l = ipos[8]; p.Iyy = inertia[7];
  const double *arm = DFLD(e, LHW_DF_DOF_ARMATURE), *damp = DFLD(e, LHW_DF_DOF_DAMPING);
  p.arm[0] = arm[0]; p.arm[1] = arm[1]; p.damp[0] = damp[0]; p.damp[1] = damp[1];
  p.gear = DFLD(e, LHW_DF_ACTUATOR_GEAR)[0];
  if (IFLD(e, LHW_IF_ACTUATOR_CTRLLIMITED)[0] || IFLD(e, LHW_IF_ACTUATOR_FORCELIMITED)[0])
    return lhw_fail(LHW_ERR_UNSUPPORTED, "cartpole kernel assumes an unlimited motor");
  if (IFLD(e, LHW_IF_ACTUATOR_TRNID)[0] != 0) return lhw_fail(LHW_ERR_UNSUPPORTED, "motor must drive the slider");
  if (DFLD(e, LHW_DF_DOF_FRICTIONLOSS)[0] != 0 || DFLD(e, LHW_DF_DOF_FRICTIONLOSS)[1] != 0)
    return lhw_fail(LHW_ERR_UNSUPPORTED, "cartpole kernel has no frictionloss support");
  p.h = e->md[LHW_DH_TIMESTEP];
  if (e->md[LHW_DH_GRAVITY_X] != 0 || e->md[LHW_DH_GRAVITY_Y] != 0) return lhw_fail(LHW_ERR_UNSUPPORTED, "gravity must be along z");
  p.g = -e->md[LHW_DH_GRAVITY_Z];
  const int32_t* lim = IFLD(e, LHW_IF_JNT_LIMITED);
  if (lim[1]) return lhw_fail(LHW_ERR_UNSUPPORTED, "hinge limit not supported by the cartpole kernel");
  const double* rng = DFLD(e, LHW_DF_JNT_RANGE);
  if (lim[0]) { p.range_lo = rng[0]; p.range_hi = rng[1]; } else { p.range_lo = -1e300; p.range_hi = 1e300; }
  p.margin = DFLD(e, LHW_DF_JNT_MARGIN)[0];
  solparam(e, DFLD(e, LHW_DF_JNT_SOLREF), DFLD(e, LHW_DF_JNT_SOLIMP), p.solref, p.solimp);
  p.invweight0 = DFLD(e, LHW_DF_DOF_INVWEIGHT0)[0];
  p.meaninertia = e->md[LHW_DH_MEANINERTIA]; p.tolerance = e->md[LHW_DH_TOLERANCE];
  p.kp = cfg->kp ? cfg->kp[0] : 0; p.kd = cfg->kd ? cfg->kd[0] : 0;
  e->obs_dim = 5; e->act_dim = 1; e->n_terms = 4;
  const size_t N = e->n_envs;
  HIPCHK(lhw_malloc(&e->cps.d, sizeof(double) * CARTPOLE_NFIELDS * N));
  HIPCHK(hipMemset(e->cps.d, 0, sizeof(double) * CARTPOLE_NFIELDS * N));
  HIPCHK(lhw_malloc(&e->cps.traj_len, sizeof(int32_t) * N));
  HIPCHK(hipMemset(e->cps.traj_len, 0, sizeof(int32_t) * N));
  HIPCHK(lhw_malloc(&e->cps.reset_count, sizeof(uint32_t) * N));
  HIPCHK(hipMemset(e->cps.reset_count, 0, sizeof(uint32_t) * N));
  HIPCHK(lhw_malloc(&e->cps.ep_stats, sizeof(double) * 3));
  HIPCHK(hipMemset(e->cps.ep_stats, 0, sizeof(double) * 3));
  return LHW_OK;
}

extern "C" int lhw_env_create(const int32_t* model_i, int64_t n_model_i, const double* model_d, int64_t n_model_d,
                              const LhwEnvConfig* cfg, LhwEnv** out) {
  if (!cfg || !out) return lhw_fail(LHW_ERR_ARG, "null argument");
  *out = nullptr;
  int rc = check_model(model_i, n_model_i, model_d, n_model_d);
  if (rc) return rc;
  if (cfg->n_envs <= 0) return lhw_fail(LHW_ERR_ARG, "n_envs must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return lhw_fail(LHW_ERR_NO_DEVICE, "no HIP device visible: liblhw has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return lhw_fail(LHW_ERR_ARG, "device %d out of range (%d visible)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  LhwEnv* e = new LhwEnv();
  e->mi.assign(model_i, model_i + n_model_i);
  e->md.assign(model_d, model_d + n_model_d);
  e->task = cfg->task; e->n_envs = cfg->n_envs; e->device = cfg->device;
  e->nq = model_i[LHW_IH_NQ]; e->nv = model_i[LHW_IH_NV]; e->nu = model_i[LHW_IH_NU];
  if (cfg->task == LHW_TASK_CARTPOLE) rc = create_cartpole(e, cfg);
  else if (cfg->task == LHW_TASK_JVRC_WALK || cfg->task == LHW_TASK_H1_STAND || cfg->task == LHW_TASK_JVRC_STEP || cfg->task == LHW_TASK_H1_WALK) rc = humanoid_create(&e->hum, e->mi, e->md, cfg, &e->obs_dim, &e->act_dim, &e->n_terms);
  else rc = lhw_fail(LHW_ERR_ARG, "unknown task %d", cfg->task);
  if (rc == LHW_OK) {
    if (lhw_malloc(&e->stage_q, sizeof(double) * (size_t)e->n_envs * e->nq) != hipSuccess ||
        lhw_malloc(&e->stage_v, sizeof(double) * (size_t)e->n_envs * e->nv) != hipSuccess)
      rc = lhw_fail(LHW_ERR_HIP, "lhw_malloc(staging) failed");
  }
  if (rc != LHW_OK) { lhw_env_destroy(e); return rc; }
  *out = e;
  return LHW_OK;
}

extern "C" int lhw_env_destroy(LhwEnv* e) {
  if (!e) return LHW_OK;
  (void)hipSetDevice(e->device);
  if (e->cps.d) (void)hipFree(e->cps.d);
  if (e->cps.traj_len) (void)hipFree(e->cps.traj_len);
  if (e->cps.reset_count) (void)hipFree(e->cps.reset_count);
  if (e->cps.ep_stats) (void)hipFree(e->cps.ep_stats);
  if (e->stage_q) (void)hipFree(e->stage_q);
  if (e->stage_v) (void)hipFree(e->stage_v);
  if (e->hum) humanoid_destroy(e->hum);
  delete e;
  return LHW_OK;
}

extern "C" int lhw_env_obs_dim(const LhwEnv* e) { return e ? e->obs_dim : LHW_ERR_ARG; }
extern "C" int lhw_env_act_dim(const LhwEnv* e) { return e ? e->act_dim : LHW_ERR_ARG; }
extern "C" int lhw_env_num_reward_terms(const LhwEnv* e) { return e ? e->n_terms : LHW_ERR_ARG; }
extern "C" int lhw_env_nq(const LhwEnv* e) { return e ? e->nq : LHW_ERR_ARG; }
extern "C" int lhw_env_nv(const LhwEnv* e) { return e ? e->nv : LHW_ERR_ARG; }

extern "C" int lhw_env_reset(LhwEnv* e, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  HIPCHK(hipSetDevice(e->device));
  if (e->task == LHW_TASK_CARTPOLE) cartpole_launch_reset(e->cp, e->cps, mask_dev, obs_dev, (hipStream_t)stream);
  else humanoid_reset(e->hum, mask_dev, obs_dev, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_env_step(LhwEnv* e, const float* act_dev, float* obs_dev, float* term_obs_dev, float* rew_dev,
                            uint8_t* done_dev, float* rew_terms_dev, void* stream) {
  if (!e || !act_dev || !obs_dev || !rew_dev || !done_dev) return lhw_fail(LHW_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (e->task == LHW_TASK_CARTPOLE)
    cartpole_launch_step(e->cp, e->cps, act_dev, obs_dev, term_obs_dev, rew_dev, done_dev, rew_terms_dev, (hipStream_t)stream);
  else humanoid_step(e->hum, act_dev, obs_dev, term_obs_dev, rew_dev, done_dev, rew_terms_dev, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_env_step_range(LhwEnv* e, int32_t first, int32_t count, const float* act_dev, float* obs_dev, float* term_obs_dev,
                                  float* rew_dev, uint8_t* done_dev, float* rew_terms_dev, void* stream) {
  if (!e || !act_dev || !obs_dev || !rew_dev || !done_dev) return lhw_fail(LHW_ERR_ARG, "null argument");
  if (e->task == LHW_TASK_CARTPOLE) return lhw_fail(LHW_ERR_UNSUPPORTED, "lhw_env_step_range: the lane-per-env cartpole stepper advances whole batches");
  HIPCHK(hipSetDevice(e->device));
  if (humanoid_step_range(e->hum, first, count, act_dev, obs_dev, term_obs_dev, rew_dev, done_dev, rew_terms_dev, (hipStream_t)stream))
    return lhw_fail(LHW_ERR_ARG, "env range [%d, %d) outside the batch", first, first + count);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

static int env_rollout_impl(LhwEnv* e, const LhwRolloutPolicy* policy, int32_t first, int32_t count, int32_t T, float* obs_dev, float* act_dev,
                            float* logp_dev, float* term_obs_dev, float* rew_dev, uint8_t* done_dev, float* rew_terms_dev, double* tin_dev, void* stream) {
  if (!e || !policy || !obs_dev || !act_dev || !logp_dev || !term_obs_dev || !rew_dev || !done_dev) return lhw_fail(LHW_ERR_ARG, "null argument");
  if (e->task == LHW_TASK_CARTPOLE) return lhw_fail(LHW_ERR_UNSUPPORTED, "lhw_env_rollout: wave-per-env (humanoid) steppers only");
  HIPCHK(hipSetDevice(e->device));
  const int rc = humanoid_rollout(e->hum, first, count, T, policy, obs_dev, act_dev, logp_dev, term_obs_dev, rew_dev, done_dev, rew_terms_dev, tin_dev, (hipStream_t)stream);
  if (rc == -1) return lhw_fail(LHW_ERR_ARG, "env range [%d, %d) outside the batch, or T = %d", first, first + count, T);
  if (rc == -4) return lhw_fail(LHW_ERR_HIP, "lhw_env_rollout: a HIP call failed while preparing the launch (%s)", hipGetErrorString(hipGetLastError()));
  if (rc) return lhw_fail(LHW_ERR_UNSUPPORTED, "lhw_env_rollout: needs a float32 actor obs %d -> 256 -> 256 -> act %d (<= 12) and a model that fits the task's resident kernel",
                          e->obs_dim, e->act_dim);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}
extern "C" int lhw_env_rollout(LhwEnv* e, const LhwRolloutPolicy* policy, int32_t first, int32_t count, int32_t T, float* obs_dev, float* act_dev,
                               float* logp_dev, float* term_obs_dev, float* rew_dev, uint8_t* done_dev, float* rew_terms_dev, void* stream) {
  return env_rollout_impl(e, policy, first, count, T, obs_dev, act_dev, logp_dev, term_obs_dev, rew_dev, done_dev, rew_terms_dev, nullptr, stream);
}
extern "C" int lhw_env_rollout_task_inputs(LhwEnv* e, const LhwRolloutPolicy* policy, int32_t first, int32_t count, int32_t T, float* obs_dev, float* act_dev,
                                           float* logp_dev, float* term_obs_dev, float* rew_dev, uint8_t* done_dev, float* rew_terms_dev, double* tin_dev,
                                           void* stream) {
  if (!tin_dev) return lhw_fail(LHW_ERR_ARG, "lhw_env_rollout_task_inputs: null task-input buffer");
  return env_rollout_impl(e, policy, first, count, T, obs_dev, act_dev, logp_dev, term_obs_dev, rew_dev, done_dev, rew_terms_dev, tin_dev, stream);
}

extern "C" int lhw_env_last_rollout_queued(LhwEnv* e) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  return e->task == LHW_TASK_CARTPOLE ? 0 : humanoid_last_rollout_queued(e->hum);
}

extern "C" int lhw_env_get_state(LhwEnv* e, double* qpos_host, double* qvel_host) {
  if (!e || !qpos_host || !qvel_host) return lhw_fail(LHW_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  if (e->task == LHW_TASK_CARTPOLE) cartpole_launch_get_state(e->cp, e->cps, e->stage_q, e->stage_v, 0);
  else humanoid_get_state(e->hum, e->stage_q, e->stage_v, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(qpos_host, e->stage_q, sizeof(double) * (size_t)e->n_envs * e->nq, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(qvel_host, e->stage_v, sizeof(double) * (size_t)e->n_envs * e->nv, hipMemcpyDeviceToHost));
  return LHW_OK;
}

extern "C" int lhw_env_set_state(LhwEnv* e, const double* qpos_host, const double* qvel_host) {
  if (!e || !qpos_host || !qvel_host) return lhw_fail(LHW_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(e->stage_q, qpos_host, sizeof(double) * (size_t)e->n_envs * e->nq, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->stage_v, qvel_host, sizeof(double) * (size_t)e->n_envs * e->nv, hipMemcpyHostToDevice));
  if (e->task == LHW_TASK_CARTPOLE) cartpole_launch_set_state(e->cp, e->cps, e->stage_q, e->stage_v, 0);
  else humanoid_set_state(e->hum, e->stage_q, e->stage_v, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return LHW_OK;
}

extern "C" int lhw_env_pop_episode_stats(LhwEnv* e, double* ret_sum, double* len_sum, int64_t* count) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  HIPCHK(hipSetDevice(e->device));
  double* dev = e->task == LHW_TASK_CARTPOLE ? e->cps.ep_stats : humanoid_ep_stats(e->hum);
  double h[3];
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h, dev, sizeof h, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(dev, 0, sizeof h));
  if (ret_sum) *ret_sum = h[0];
  if (len_sum) *len_sum = h[1];
  if (count) *count = (int64_t)h[2];
  return LHW_OK;
}

/* resident workgroups (= waves) per CU the runtime grants the wave-per-env step kernel */
extern "C" int lhw_debug_stepper_occupancy(void) { return humanoid_occupancy(); }

extern "C" int lhw_env_phase_cycles(LhwEnv* e, int enable, int64_t* out16) {
  if (!e || !e->hum) return lhw_fail(LHW_ERR_UNSUPPORTED, "phase profiling exists for the wave-per-env stepper only");
  HIPCHK(hipSetDevice(e->device));
  static_assert(sizeof(long long) == sizeof(int64_t), "");
  if (humanoid_profile(e->hum, enable, (long long*)out16)) return lhw_fail(LHW_ERR_HIP, "profile buffer");
  return LHW_OK;
}

/* Diagnostic: first call arms the recording; later calls return, per env, the shader-clock cycles its wavefront group spent
 * in the most recent control-step launch (host pointer [N] int64, synchronous). */
extern "C" int lhw_env_debug_wave_cycles(LhwEnv* e, int64_t* out) {
  if (!e || !e->hum) return lhw_fail(LHW_ERR_UNSUPPORTED, "wave cycles exist for the humanoid steppers only");
  HIPCHK(hipSetDevice(e->device));
  if (humanoid_wave_cycles(e->hum, (long long*)out)) return lhw_fail(LHW_ERR_HIP, "wave cycle buffer");
  return LHW_OK;
}


extern "C" int lhw_env_pop_fault_stats(LhwEnv* e, int64_t* contact_overflow, int64_t* diverged) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  if (contact_overflow) *contact_overflow = 0;
  if (diverged) *diverged = 0;
  if (!e->hum) return LHW_OK;  // the cartpole kernel has neither contacts nor a divergence path
  HIPCHK(hipSetDevice(e->device));
  double h[2];
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h, humanoid_ep_stats(e->hum) + 3, sizeof h, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(humanoid_ep_stats(e->hum) + 3, 0, sizeof h));
  if (contact_overflow) *contact_overflow = (int64_t)h[0];
  if (diverged) *diverged = (int64_t)h[1];
  return LHW_OK;
}

extern "C" int lhw_env_enable_task_inputs(LhwEnv* e, int enable) {
  if (!e || !e->hum) return lhw_fail(LHW_ERR_UNSUPPORTED, "task inputs exist for the humanoid tasks only");
  HIPCHK(hipSetDevice(e->device));
  if (humanoid_task_inputs(e->hum, enable ? 1 : 0, nullptr, nullptr)) return lhw_fail(LHW_ERR_HIP, "task input buffer");
  return LHW_OK;
}
extern "C" int lhw_env_get_task_inputs(LhwEnv* e, double* out_host) {
  if (!e || !e->hum || !out_host) return lhw_fail(LHW_ERR_ARG, "null argument / not a humanoid task");
  HIPCHK(hipSetDevice(e->device));
  const int rc = humanoid_task_inputs(e->hum, -1, out_host, nullptr);
  if (rc == -2) return lhw_fail(LHW_ERR_ARG, "lhw_env_get_task_inputs: call lhw_env_enable_task_inputs(env, 1) first");
  if (rc) return lhw_fail(LHW_ERR_HIP, "task input copy");
  return LHW_OK;
}
extern "C" int lhw_env_task_inputs_device(LhwEnv* e, double** out_dev) {
  if (!e || !e->hum || !out_dev) return lhw_fail(LHW_ERR_ARG, "null argument / not a humanoid task");
  return humanoid_task_inputs(e->hum, -1, nullptr, out_dev) ? lhw_fail(LHW_ERR_HIP, "task input buffer") : LHW_OK;
}
extern "C" int lhw_env_get_actuator_state(LhwEnv* e, double* pos_host, double* vel_host, double* torque_host) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  if (!e->hum) return lhw_fail(LHW_ERR_UNSUPPORTED, "actuator state is kept by the humanoid steppers only");
  HIPCHK(hipSetDevice(e->device));
  if (humanoid_actuator_state(e->hum, pos_host, vel_host, torque_host)) return lhw_fail(LHW_ERR_HIP, "actuator state copy failed");
  return LHW_OK;
}

extern "C" int lhw_env_pop_rerun_count(LhwEnv* e, int64_t* reruns) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  if (reruns) *reruns = 0;
  if (!e->hum) return LHW_OK;
  HIPCHK(hipSetDevice(e->device));
  double h = 0;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(&h, humanoid_ep_stats(e->hum) + 5, sizeof h, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(humanoid_ep_stats(e->hum) + 5, 0, sizeof h));
  if (reruns) *reruns = (int64_t)h;
  return LHW_OK;
}

extern "C" int lhw_env_debug_step_record(LhwEnv* e, double* seq, double* floor_z, int32_t* istate) {
  if (!e || !e->hum || e->task != LHW_TASK_JVRC_STEP) return lhw_fail(LHW_ERR_ARG, "step record: not a stepping-task env");
  if (humanoid_step_record(e->hum, seq, floor_z, istate)) return lhw_fail(LHW_ERR_HIP, "step record copy failed");
  return LHW_OK;
}

extern "C" int lhw_env_set_iteration(LhwEnv* e, int64_t iteration) {
  if (!e) return lhw_fail(LHW_ERR_ARG, "null env");
  if (e->hum) humanoid_set_iteration(e->hum, iteration);
  return LHW_OK;
}
