/* lhw.h -- C ABI of the MI355X-native batched environment stepper + PPO update
 * (liblhw.so, built from learninghumanoidwalking_amd/csrc for gfx950).
 *
 * This is the drop-in boundary of the hot path.  The reference has
 * no FFI of its own -- it is pure Python over third-party wheels -- so each entry point
 * below names the reference interface whose work it takes over:
 *
 *   lhw_env_create   <- env_fn() / BaseHumanoidEnv.__init__ + MujocoEnv.__init__
 *                       (reference envs/common/base_humanoid_env.py:38-74,
 *                        envs/common/mujoco_env.py:16-36, envs/cartpole/cartpole_env.py:82-107)
 *   lhw_env_reset    <- MujocoEnv.reset -> reset_model
 *                       (envs/common/mujoco_env.py:113-116, base_humanoid_env.py:247-276,
 *                        envs/cartpole/cartpole_env.py:109-121)
 *   lhw_env_step     <- env.step(action) for every env of the batch, plus the per-step episode
 *                       bookkeeping of RolloutWorker.sample (truncation at max_traj_len,
 *                       terminal observation for the bootstrap value, reset on episode end)
 *                       (base_humanoid_env.py:199-227, robots/robot_base.py:41-98,
 *                        envs/common/robot_interface.py:493-546 [PD + mj_step],
 *                        rl/workers/rollout_worker.py:142-181)
 *   lhw_env_get_state / lhw_env_set_state
 *                    <- data.qpos / data.qvel reads, MujocoEnv.set_state
 *                       (envs/common/mujoco_env.py:118-127); parity hooks
 *   lhw_env_set_iteration <- RolloutWorker.sync_state's env.robot.iteration_count = itr
 *                       (rl/workers/rollout_worker.py:95)
 *   lhw_gae          <- PPOBuffer.finish_path over every trajectory of the batch
 *                       (rl/storage/rollout_storage.py:53-85)
 *   lhw_mlp_*, lhw_ppo_* <- Gaussian_FF_Actor / FF_V forward and PPO.update_actor_critic
 *                       (rl/policies/actor.py:160-188, rl/policies/critic.py:41-49,
 *                        rl/algos/ppo.py:299-406)
 *
 * Conventions: every function returns 0 on success or a negative LhwStatus; the message is
 * available from lhw_last_error().  The caller owns every buffer and passes raw device (or,
 * where stated, host) pointers -- no torch types cross this boundary.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream); calls are asynchronous with respect
 * to it unless stated.  One handle belongs to one device and one host thread at a time.
 */
#ifndef LHW_H
#define LHW_H

#include <stdint.h>

#include "lhw_model_fields.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  LHW_OK = 0,
  LHW_ERR_ARG = -1,         /* bad argument / size mismatch */
  LHW_ERR_HIP = -2,         /* HIP runtime error (message has hipGetErrorString) */
  LHW_ERR_MODEL = -3,       /* model blob malformed or exceeds a compiled-in limit */
  LHW_ERR_UNSUPPORTED = -4, /* feature outside the implemented subset */
  LHW_ERR_NO_DEVICE = -5    /* no usable GPU: the product path has no CPU fallback */
} LhwStatus;

typedef enum {
  LHW_TASK_CARTPOLE = 0, /* reference envs/cartpole/cartpole_env.py */
  LHW_TASK_JVRC_WALK = 1, /* reference envs/jvrc/jvrc_walk.py + tasks/walking_task.py */
  LHW_TASK_H1_STAND = 2,  /* reference envs/h1/h1_env.py + tasks/standing_task.py (+ domain_randomization.py) */
  LHW_TASK_JVRC_STEP = 3, /* reference envs/jvrc/jvrc_step.py + tasks/stepping_task.py (box terrain, footstep targets) */
  LHW_TASK_H1_WALK = 4    /* reference envs/h1/h1_walk.py: H1 robot (noise, randomisation as H1_STAND) + tasks/walking_task.py;
                             task_params / task_iparams as LHW_TASK_H1_STAND, plus clock_lut / period */
} LhwTask;

/* done flags written by lhw_env_step */
#define LHW_DONE_TERMINATED 1u /* env.step returned done=True */
#define LHW_DONE_TRUNCATED 2u  /* trajectory reached max_traj_len (rollout_worker.py:152) */

typedef struct LhwEnv LhwEnv;

typedef struct {
  int32_t task;         /* LhwTask */
  int32_t n_envs;       /* batch size N */
  int32_t device;       /* HIP device ordinal */
  int32_t frame_skip;   /* sim sub-steps per control step (control_dt / sim_dt) */
  int32_t max_traj_len; /* >0: truncate + auto-reset like RolloutWorker.sample; 0: never reset inside step */
  int32_t env_id_base;  /* global index of env 0 (RNG keys use env_id_base + n): multi-GPU sharding */
  uint64_t seed;        /* counter-based RNG key (replaces the reference's global np.random stream) */
  double action_smoothing;       /* base_humanoid_env.py:209 (unused by cartpole) */
  const double* kp;              /* [nu] PD gains */
  const double* kd;              /* [nu] */
  const double* nominal_qpos;    /* [nq] reset pose (jvrc_base.py:52-54); NULL -> qpos0 */
  const double* action_offset;   /* [nu] nominal joint pose added to targets (base_humanoid_env.py:212) */
  const double* task_params;     /* task-specific doubles, see LHW_TP_* */
  int32_t n_task_params;
  const int32_t* task_iparams;   /* task-specific ints, see LHW_TI_* */
  int32_t n_task_iparams;
  const double* clock_lut;       /* walking: [4][period] r_frc, r_vel, l_frc, l_vel at integer phases
                                    (tasks/rewards.py:196-300 evaluated on 0..period-1) */
  int32_t period;
  double init_noise;             /* BaseHumanoidEnv._apply_init_noise (envs/common/base_humanoid_env.py:278-305) for ANY humanoid task: half-width in
                                    radians of the uniform noise on root roll / pitch and every joint at reset (root z += U(0, 0.02)); 0 = off.
                                    (H1 tasks may also give it as task_params[LHW_TP_H1_INIT_NOISE]; this field wins when > 0.) */
  /* apply_perturbation for the JVRC tasks (envs/common/domain_randomization.py:10-26 behind base_humanoid_env.py:86-92, 224-225; the H1 tasks
   * carry theirs in task_params / task_iparams): after the observation of a control step, with probability 1 / interval, every listed body
   * receives a world-frame force U(+-force)^3 and torque U(+-torque)^3 at its centre of mass, each body followed by a coin flip that clears ALL
   * applied wrenches; they act until the next draw or reset.  interval = int(cfg.interval / control_dt) control steps, 0 = off. */
  int32_t perturb_interval;
  int32_t n_perturb_bodies;      /* <= 2 */
  int32_t perturb_bodies[2];     /* body ids of the packed model */
  double perturb_force, perturb_torque;
} LhwEnvConfig;

/* task_params indices: [0] target root height (walking goal_height / standing 0.98); H1 standing adds the
 * init-noise half-width (rad), perturbation force / torque magnitudes and 35 per-entry observation-noise half-widths */
enum { LHW_TP_GOAL_HEIGHT = 0, LHW_TP_COUNT = 1, LHW_TP_H1_INIT_NOISE = 1, LHW_TP_H1_FORCE_MAG = 2, LHW_TP_H1_TORQUE_MAG = 3,
       LHW_TP_H1_OBS_NOISE = 4 };
/* stepping task: local offsets of the right / left foot force sites on the foot bodies, target radius
 * (stepping_task.py:268), half sizes of the terrain boxes (:325), then the footstep plans of the CURVED mode:
 * count, and per plan 1 + 20*3 doubles = length, (x, y, theta) rows (stepping_task.py:52-64) */
enum { LHW_TP_STEP_RSITE = 1, LHW_TP_STEP_LSITE = 4, LHW_TP_STEP_RADIUS = 7, LHW_TP_STEP_BOX_SIZE = 8, LHW_TP_STEP_NPLANS = 11,
       LHW_TP_STEP_PLANS = 12, LHW_STEP_MAX_SEQ = 20 };
/* task_iparams indices: body ids of root, head (walking) / torso (standing), right foot, left foot; H1 standing adds the
 * randomisation intervals in control steps, the perturbed bodies, and the randomised dofs (10) and bodies (11) */
enum { LHW_TI_ROOT_BODY = 0, LHW_TI_HEAD_BODY = 1, LHW_TI_RFOOT_BODY = 2, LHW_TI_LFOOT_BODY = 3, LHW_TI_COUNT = 4,
       LHW_TI_H1_DYNRAND_INTERVAL = 4, LHW_TI_H1_PERTURB_INTERVAL = 5, LHW_TI_H1_N_PBODY = 6, LHW_TI_H1_PBODY = 7,
       LHW_TI_H1_RAND_DOF = 9, LHW_TI_H1_RAND_BODY = 19 };
/* stepping task: first terrain-box geom id (boxes are contiguous), box count (<= 20), floor geom id, target dwell in
 * control steps (stepping_task.py:269) */
enum { LHW_TI_STEP_BOX_GEOM0 = 4, LHW_TI_STEP_NBOX = 5, LHW_TI_STEP_FLOOR_GEOM = 6, LHW_TI_STEP_DELAY_FRAMES = 7, LHW_TI_STEP_COUNT = 8 };

int lhw_version(void);
const char* lhw_last_error(void);

/* model_i / model_d: packed model blobs (host pointers), layout in lhw_model_fields.h */
int lhw_env_create(const int32_t* model_i, int64_t n_model_i, const double* model_d, int64_t n_model_d,
                   const LhwEnvConfig* cfg, LhwEnv** out);
int lhw_env_destroy(LhwEnv* env);

int lhw_env_obs_dim(const LhwEnv* env);
int lhw_env_act_dim(const LhwEnv* env);
int lhw_env_num_reward_terms(const LhwEnv* env);
int lhw_env_nq(const LhwEnv* env);
int lhw_env_nv(const LhwEnv* env);

/* Reset the envs whose mask byte is non-zero (mask_dev == NULL: all).  obs_dev [N][obs_dim] f32
 * receives the first observation of the reset envs (rows of other envs are left untouched). */
int lhw_env_reset(LhwEnv* env, const uint8_t* mask_dev, float* obs_dev, void* stream);

/* One control step for all N envs.
 *   act_dev      [N][act_dim] f32, in
 *   obs_dev      [N][obs_dim] f32, out: observation to act on next (after the auto-reset, if any)
 *   term_obs_dev [N][obs_dim] f32, out, nullable: observation returned by env.step itself
 *                (differs from obs_dev only where the episode ended: bootstrap input)
 *   rew_dev      [N] f32, out: sum of reward terms in the reference's dict order
 *   done_dev     [N] u8, out: LHW_DONE_* flags
 *   rew_terms_dev [N][num_reward_terms] f32, out, nullable */
int lhw_env_step(LhwEnv* env, const float* act_dev, float* obs_dev, float* term_obs_dev, float* rew_dev,
                 uint8_t* done_dev, float* rew_terms_dev, void* stream);

/* Test hook: copy the per-env stepping-task record to host: seq [N][20][6] = (x y z theta cos sin) of every target step /
 * terrain box, floor_z [N], and istate [N][5] = t1 t2 target_reached target_reached_frames sequence_length. */
int lhw_env_debug_step_record(LhwEnv* env, double* seq, double* floor_z, int32_t* istate);

/* Same as lhw_env_step for the envs [first, first + count) only; all pointers are the FULL-batch arrays (the range indexes
 * into them).  Lets the host pipeline independent groups of envs on separate HIP streams: without a batch-wide barrier per
 * control step, the tail of one group's kernel overlaps the next group's (wave-per-env steppers only). */
int lhw_env_step_range(LhwEnv* env, int32_t first, int32_t count, const float* act_dev, float* obs_dev, float* term_obs_dev,
                       float* rew_dev, uint8_t* done_dev, float* rew_terms_dev, void* stream);
/* The frozen actor as the resident rollout evaluates it inside the stepper's wavefronts: the reference's worker holds a copy of
 * the policy for the whole rollout (rl/workers/rollout_worker.py:62-77 sync_policy) and calls it once per control step
 * (:142-150, Gaussian_FF_Actor.forward rl/policies/actor.py:160-188).  Device pointers, filled by lhw_ppo_rollout_policy. */
typedef struct {
  const float *w1t, *b1, *w2t, *b2, *w3t, *b3; /* TRANSPOSED weights ([in][out]: W1^T [obs_pad][hidden], W2^T [hidden][hidden],
                                                  W3^T [hidden][act_pad]) and the biases */
  const float *stdv;                           /* [act_dim] standard deviations of the Gaussian head */
  const float *obs_mean, *obs_std;             /* [obs_dim] observation normalisation (actor.obs_mean / obs_std) */
  int32_t obs_dim, obs_pad, act_dim, act_pad, hidden;
  int32_t deterministic;                       /* != 0: act = mean (evaluation) */
  int32_t fp16_operands;                       /* != 0: weights and activations rounded to fp16 per product, float32 accumulation
                                                  (lhw_ppo_set_inference_dtype; BASELINE config 5 "fp16 actor / critic") */
  uint64_t seed;                               /* policy-noise key, as lhw_ppo_forward's */
  uint32_t counter;                            /* policy-stream counter of the FIRST control step; step t uses counter + t */
} LhwRolloutPolicy;
/* The resident rollout: T control steps of envs [first, first + count) in ONE launch -- the body of RolloutWorker.sample's loop
 * (rl/workers/rollout_worker.py:142-181: action = policy(state); env.step(action); store; reset on episode end) executed wave
 * by wave, no wavefront ever waiting for another env's control step.  Bitwise the values of T x { lhw_ppo_forward_at(act, logp
 * only) ; lhw_env_step_range } on the same buffers.  All pointers are device buffers, TIME-major over the FULL batch (N = the
 * env's n_envs; the range indexes into each time slice):
 *   obs_dev      [T + 1][N][obs_dim] f32: slice 0 in (the observation to act on first: lhw_env_reset's output or slice T of the
 *                previous rollout), slices 1 .. T out
 *   act_dev      [T][N][act_dim], logp_dev [T][N]: sampled actions and their log-densities, out
 *   term_obs_dev [T][N][obs_dim], rew_dev [T][N], done_dev [T][N] (LHW_DONE_*): as lhw_env_step, per control step, out
 *   rew_terms_dev [N][num_reward_terms] of the LAST control step, nullable
 * Humanoid tasks only, feed-forward float32 actor with hidden width 256 and act_dim <= 12 (LHW_ERR_UNSUPPORTED otherwise: the
 * caller keeps the launch-per-step pipeline).  Ranges of concurrent calls on different streams must be disjoint.  Stepping task
 * with more envs than the chip has wave slots: the resident wavefronts drain a device queue of (env, LHW_ROLLOUT_CHUNK = 10 control
 * steps) jobs instead of keeping one env each (the cost of an env follows its walking mode; same values either way). */
int lhw_env_rollout(LhwEnv* env, const LhwRolloutPolicy* policy, int32_t first, int32_t count, int32_t T, float* obs_dev, float* act_dev,
                    float* logp_dev, float* term_obs_dev, float* rew_dev, uint8_t* done_dev, float* rew_terms_dev, void* stream);
/* lhw_env_rollout that also exports the batched sim facade (below) of EVERY control step: tin_dev [T][N][LHW_TASK_INPUT_DIM] float64
 * receives, per control step and env, the record lhw_env_get_task_inputs would return after that step -- what the reference's robot
 * hands its exchangeable task once per control step (robots/robot_base.py:88-96: task.step / calc_reward / done reading RobotInterface).
 * For task code that only changes the REWARD (an edited tasks/rewards.py): the rollout runs resident with the fused termination and
 * resets, and the plug-in evaluates the whole [T * N] batch once after the launch (task_hook.py).  Everything else as lhw_env_rollout. */
int lhw_env_rollout_task_inputs(LhwEnv* env, const LhwRolloutPolicy* policy, int32_t first, int32_t count, int32_t T, float* obs_dev,
                                float* act_dev, float* logp_dev, float* term_obs_dev, float* rew_dev, uint8_t* done_dev, float* rew_terms_dev,
                                double* tin_dev, void* stream);
/* 1 if the most recent lhw_env_rollout of this env drained the job queue (rocprof name humanoid_rollout_kernel<TASK, 64, true>), 0 if
 * every wavefront kept its env group (<.., false>); what bench.py names as the dominant kernel. */
int lhw_env_last_rollout_queued(LhwEnv* env);
/* Parity hooks; HOST pointers, synchronous.  qpos [N][nq], qvel [N][nv] float64. */
int lhw_env_get_state(LhwEnv* env, double* qpos_host, double* qvel_host);
int lhw_env_set_state(LhwEnv* env, const double* qpos_host, const double* qvel_host);
/* The batched sim facade: everything the reference's tasks read through RobotInterface for one control step
 * (envs/common/robot_interface.py:75-82 qpos / qvel / qacc, :163-185 actuator position / velocity / torque, :242-250 foot body
 * positions, :269-325 foot-floor contacts and ground reaction forces, :357-380 body velocities, :382-394 object positions,
 * :472-484 self collisions) plus the arguments RobotBase.step passes to calc_reward (robots/robot_base.py:88-96), as the fused
 * kernel saw them when it evaluated the reward of the LAST control step.  Slow-path hook for task code that is not compiled
 * into the kernel (a user's BaseTask, tasks/base_task.py:12-83, or the reference's own tasks/rewards.py run as a cross-check):
 * enable once, step, then read the [N][LHW_TASK_INPUT_DIM] float64 records on the host or, for on-device consumers, through the
 * device pointer.  Derived fields follow MuJoCo's staleness (SURVEY note S): positions / velocities of bodies, contacts, qacc
 * and actuator fields describe the last forward pass, qpos / qvel the state after it. */
enum LhwTaskInput {
  LHW_TIN_GRF_R = 0,          /* get_rfoot_grf(): sum over the right foot's floor contacts of |contact wrench| */
  LHW_TIN_GRF_L = 1,
  LHW_TIN_CONTACT_Z = 2,      /* min z of the foot-floor contact points (0 when there is none) */
  LHW_TIN_FOOT_CONTACT = 3,   /* check_rfoot_floor_collision() or check_lfoot_floor_collision() */
  LHW_TIN_SELF_COLLISION = 4, /* check_self_collisions() */
  LHW_TIN_PHASE = 5,          /* task._phase, mode (WalkModes / stepping mode index) and mode_ref AFTER task.step() */
  LHW_TIN_MODE = 6,
  LHW_TIN_MODE_REF = 7,       /* 3 */
  LHW_TIN_RFOOT_VEL = 10,     /* 3: linear velocity of the body frame origin (mj_objectVelocity), world axes; the tasks use its norm */
  LHW_TIN_LFOOT_VEL = 13,     /* 3 */
  LHW_TIN_ROOT_VEL_LOCAL = 16,/* 3: get_body_vel(root, frame=1)[0] */
  LHW_TIN_ROOT_XPOS = 19,     /* 3 */
  LHW_TIN_HEAD_XPOS = 22,     /* 3 */
  LHW_TIN_RFOOT_XPOS = 25,    /* 3 */
  LHW_TIN_LFOOT_XPOS = 28,    /* 3 */
  LHW_TIN_QPOS = 32,          /* nq <= 19 */
  LHW_TIN_QVEL = 51,          /* nv <= 18 */
  LHW_TIN_QACC = 69,          /* nv */
  LHW_TIN_ACT_POS = 87,       /* nu <= 12: get_act_joint_positions() */
  LHW_TIN_ACT_VEL = 99,
  LHW_TIN_ACT_TAU = 111,      /* get_act_joint_torques() */
  LHW_TIN_PREV_TORQUE = 123,  /* calc_reward(prev_torque, prev_action, action) */
  LHW_TIN_PREV_ACTION = 135,
  LHW_TIN_ACTION = 147,
  LHW_TIN_ROOT_XMAT = 160,    /* 9: get_object_affine_by_name(root body)'s rotation, row-major (standing_task.py:76-85: the torso in the pelvis frame) */
  LHW_TASK_INPUT_DIM = 176
};
int lhw_env_enable_task_inputs(LhwEnv* env, int enable);
/* HOST pointer [N][LHW_TASK_INPUT_DIM] float64, synchronous; humanoid tasks only, after lhw_env_enable_task_inputs(env, 1). */
int lhw_env_get_task_inputs(LhwEnv* env, double* out_host);
/* The same records on the device (valid until the env is destroyed or the export disabled); NULL while disabled. */
int lhw_env_task_inputs_device(LhwEnv* env, double** out_dev);
/* Actuated-joint fields of the LAST forward pass, as the reference's RobotInterface getters return them after env.step
 * (envs/common/robot_interface.py:163-185: get_act_joint_positions / _velocities / _torques = actuator_length / gear,
 * actuator_velocity / gear, actuator_force * gear).  HOST pointers [N][nu] float64, synchronous; humanoid tasks only. */
int lhw_env_get_actuator_state(LhwEnv* env, double* pos_host, double* vel_host, double* torque_host);
/* Episode statistics accumulated on device since the last call: sum of finished-episode returns,
 * sum of finished-episode lengths, number of finished episodes (host pointers, synchronous, resets them). */
int lhw_env_pop_episode_stats(LhwEnv* env, double* ret_sum, double* len_sum, int64_t* count);
/* Fault counters since the last call (host pointers, synchronous, resets them): control steps in which contacts were
 * dropped because more than the compiled-in cap were active (16 per env; the stepping task, whose feet rest on the floor AND on
 * the terrain boxes under them, merges identical contacts and holds 192 found / 64 distinct ones per sub-step), and control steps in which an env's state became
 * non-finite (the env is flagged terminated, its outputs are zeroed, and it is reset like any finished episode). */
int lhw_env_pop_fault_stats(LhwEnv* env, int64_t* contact_overflow, int64_t* diverged);
/* Control steps since the last call that the two-envs-per-wave kernels handed to the one-env-per-wave kernel because an
 * env touched more contacts than their layout holds (8); such a step costs roughly three ordinary ones.  Diagnostic of the
 * rollout, no reference counterpart (host pointer, synchronous, resets the counter). */
int lhw_env_pop_rerun_count(LhwEnv* env, int64_t* reruns);
/* Test / tuning hook for the update's GEMM kernel (no reference counterpart): C = op(A) op(B) on device buffers.
 * a_kc: A stored [M][K] (else [K][M]); b_kc: B stored [N][K] (else [K][N]); wt: 0 = automatic tile choice, 1 = 64x64, 2 = 128x128
 * block tiles, 16 = fp16 operands (64x64 tiles, fp16 MFMA); optional epilogue bias[n], ReLU, mask (v = mask[m][n] > 0 ? v : 0).
 * With `part` (split-K scratch,
 * [ceil(K / k_chunk)][M*N] floats) the partial products are reduced in slice order and ADDED to C, as the weight-gradient GEMMs
 * of lhw_ppo_grad do; `colsum` ([slices][M] scratch, A stored [K][M] only) also ADDS sum_k A[k][m] to colsum_out[m] (the bias
 * gradient fused into the same pass).  Leading dimensions must be multiples of 4 floats (16-byte rows). */
int lhw_debug_gemm(int32_t a_kc, int32_t b_kc, int32_t wt, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                   int32_t ldb, float* C, int32_t ldc, const float* bias, int32_t relu, const float* mask, int32_t ldmask,
                   int32_t k_chunk, float* part, float* colsum, float* colsum_out, void* stream);
/* Test hook for the wide weight gradient of the update (csrc/lhw_ppo.hip: wgrad_wide_kernel; reference /root/reference/rl/algos/ppo.py:387-396,
 * the hidden layer's part of loss.backward()): per k slice z of k_chunk rows, part[z] [256][256] = A[rows of z]^T B[rows of z] and
 * colsum[z] [256] = column sums of A's rows (A, B: [K][256] device buffers, dh2 and h1; colsum may be NULL).  The caller reduces the slices. */
int lhw_debug_wgrad_wide(const float* A, const float* B, int32_t K, int32_t k_chunk, float* part, float* colsum, void* stream);
/* Test hook for the fused skinny weight gradients of the update (csrc/lhw_ppo.hip: wgrad_skinny_kernel; reference
 * /root/reference/rl/algos/ppo.py:387-396, the first- and last-layer parts of loss.backward()): dW1 [H][Dp] += dh1^T x, db1 += colsum(dh1),
 * dW3 [O][H] += dy^T h2, db3 += colsum(dy) over R rows in one launch.  H = 256, Dp <= 64 (multiple of 4), O <= Op <= 32; device buffers;
 * scratch: ceil(R / max(128, ceil(R / 256))) x (256 Dp + 256 + 256 O + O) floats. */
int lhw_debug_wgrad_skinny(int32_t H, int32_t Dp, int32_t O, int32_t Op, const float* dh1, const float* x, int32_t ldx, const float* dy,
                           const float* h2, int32_t R, float* dW1, float* db1, float* dW3, float* db3, float* scratch, void* stream);
/* Test hooks for the LDS-resident strip kernels of the update's MLPs (csrc/lhw_mlp_strip.hip; reference:
 * /root/reference/rl/algos/ppo.py:299-406, the actor / critic forward and the activation gradients of loss.backward()):
 * forward  h1 = relu(x W1^T + b1), h2 = relu(h1 W2^T + b2), y = h2 W3^T + b3 for R rows in one launch;
 * backward dh2 = (dy W3) * (h2 > 0), dh1 = (dh2 W2) * (h1 > 0).  Weights in torch Linear layout ([out][in]; W1 row stride Dp,
 * a multiple of 4; W3 [Op][H]); device buffers; H must be 256, Dp <= 64, O <= 32 (LHW_ERR_UNSUPPORTED otherwise).  wt_scratch:
 * (Dp + 256 + Op) * 256 floats for the transposed weight copies the forward kernel reads. */
int lhw_debug_mlp_strip_forward(int32_t H, int32_t Dp, int32_t O, int32_t Op, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, const float* x, int32_t ldx, int32_t R,
                                float* h1, float* h2, float* y, float* wt_scratch, void* stream);
int lhw_debug_mlp_strip_backward(int32_t H, int32_t O, int32_t Op, const float* w2, const float* w3, const float* dy, int32_t R,
                                 const float* h1, const float* h2, float* dh2, float* dh1, void* stream);
/* The same two launches with the ReLU masks handed over as BITS (round 6): the forward launch also writes, per hidden layer,
 * ceil(R / 64) * 512 words (bit set = activation positive, in the kernels' own thread-to-element order); a backward launch over the same
 * R rows given those words does not read h1 / h2 (which may then be NULL).  64-row slabs whatever the row count. */
int lhw_debug_mlp_strip_forward_bits(int32_t H, int32_t Dp, int32_t O, int32_t Op, const float* w1, const float* b1, const float* w2,
                                     const float* b2, const float* w3, const float* b3, const float* x, int32_t ldx, int32_t R, float* h1,
                                     float* h2, float* y, float* wt_scratch, uint32_t* bits1, uint32_t* bits2, void* stream);
int lhw_debug_mlp_strip_backward_bits(int32_t H, int32_t O, int32_t Op, const float* w2, const float* w3, const float* dy, int32_t R,
                                      const float* h1, const float* h2, float* dh2, float* dh1, const uint32_t* bits1, const uint32_t* bits2,
                                      void* stream);
/* Test hook: the rollout's per-control-step policy launch (observation normalisation -> actor -> Gaussian head, one strip launch;
 * what lhw_ppo_forward_at runs when only act / logp are requested) on R raw observation rows [R][obs_dim], from an actor view.
 * y [R][act_pad] receives the means.  The reference of lhw_env_rollout's in-wave policy step (bitwise). */
int lhw_debug_policy_step(const LhwRolloutPolicy* policy, const float* obs, int32_t R, uint32_t env_id_base, uint32_t counter, float* y,
                          float* act, float* logp, void* stream);
/* Diagnostic (load balance): the first call arms the recording; later calls return, per env, the shader-clock cycles its
 * wavefront group spent in the most recent control-step launch.  HOST pointer [N] int64, synchronous; humanoid tasks only. */
int lhw_env_debug_wave_cycles(LhwEnv* env, int64_t* cycles_host);
int lhw_env_set_iteration(LhwEnv* env, int64_t iteration);
/* Diagnostic: shader-clock cycles env 0 spent in each phase of the wave-per-env stepper since the last call
 * (slots: 0 kinematics, 1 com/cdof, 2 CRBA, 3 collision, 4 constraint rows, 5 velocity/RNE, 6 smooth solve,
 * 7 Newton, 8 Euler tail (integration), 9 whole control step incl. the above, 10 task/reward/reset,
 * 11 actuation + row loads + reference acceleration, 12 factor/solve of M, 13 factor/solve of M + h*damping).
 * enable != 0 turns the counters on. */
int lhw_env_phase_cycles(LhwEnv* env, int enable, int64_t* out16);
/* Diagnostic: resident workgroups per CU the HIP runtime reports for the wave-per-env step kernel. */
int lhw_debug_stepper_occupancy(void);

/* ------------------------------------------------------------------ PPO (actor/critic MLP, GAE, update) */
typedef struct LhwPpo LhwPpo;

typedef struct {
  int32_t device;
  int32_t obs_dim, act_dim, hidden; /* reference networks: 2 x 256 ReLU (rl/policies/actor.py:127) */
  int32_t learn_std;                /* --learn-std: stds are a parameter (actor.py:145-148) */
  int32_t max_rows;                 /* workspace capacity: max rows per forward / minibatch */
  float lr, eps;                    /* Adam lr and eps; eps is also the advantage-normalisation eps (ppo.py:429-430,485) */
  float clip, entropy_coeff, mirror_coeff, max_grad_norm;
  /* signed-permutation mirror tables (NULL = no mirror loss): out[j] = sign[j] * in[src[j]]
   * == x @ _get_symmetry_matrix(...) with the clock sign flip folded in (rl/envs/wrappers.py:53-85) */
  const int32_t* mirror_obs_src; const float* mirror_obs_sign; /* [obs_dim] */
  const int32_t* mirror_act_src; const float* mirror_act_sign; /* [act_dim] */
} LhwPpoConfig;

int lhw_ppo_create(const LhwPpoConfig* cfg, LhwPpo** out);
int lhw_ppo_destroy(LhwPpo* ppo);
/* number of float32 in the flat parameter vector theta (actor | stds | critic, internally padded) */
int64_t lhw_ppo_param_count(const LhwPpo* ppo);
/* offsets of actor W1,b1,W2,b2,W3,b3, stds, critic W1..b3, then padded obs width and padded actor out width */
int lhw_ppo_layout(const LhwPpo* ppo, int64_t* out15);
/* xn (and xm if non-NULL) [R][pad4(obs_dim)] <- (obs - mean)/std of R raw rows (mirrored for xm) */
int lhw_ppo_normalize(LhwPpo* ppo, const float* obs, int64_t R, const float* obs_mean, const float* obs_std, float* xn,
                      float* xm, void* stream);
/* rollout inference on N rows: mu/act/logp [N][act_dim]/[N][act_dim]/[N], value [N]; any output group may be NULL */
int lhw_ppo_forward(LhwPpo* ppo, const float* theta, const float* obs, int64_t N, const float* obs_mean,
                    const float* obs_std, uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic,
                    float* mu, float* act, float* logp, float* value, void* stream);
/* Rollout bracket.  theta does not change while a rollout is collected (the reference's workers hold a frozen copy of the policy,
 * rl/workers/rollout_worker.py:62-77 sync_policy), so the [in][out] weight copies the forward strip kernel multiplies by are made
 * ONCE here (on `stream`; streams that issue forwards afterwards must be ordered behind it) instead of in every policy step.
 * Until lhw_ppo_end_rollout -- or lhw_ppo_apply, which changes theta -- every lhw_ppo_forward / _forward_at call with this theta
 * pointer reads those copies (read-only: any number of concurrent calls on any streams).  Outside a bracket each call makes
 * its own copies, as before.  theta must NOT be written between lhw_ppo_begin_rollout and lhw_ppo_end_rollout (a checkpoint load, a
 * parameter broadcast): the bracket is keyed by the pointer, so the copies would silently go stale -- close it first (the Python
 * layer's PpoKernels.set_tensors does). */
int lhw_ppo_begin_rollout(LhwPpo* ppo, const float* theta, void* stream);
int lhw_ppo_end_rollout(LhwPpo* ppo);
/* Inside a rollout bracket opened with this theta: the actor of theta as lhw_env_rollout reads it (the bracket's [in][out]
 * weight copies, biases and stds inside theta, the caller's normalisation vectors).  Valid until lhw_ppo_end_rollout /
 * lhw_ppo_apply.  LHW_ERR_UNSUPPORTED outside a bracket or for shapes the strip kernels do not cover.  With fp16 inference selected
 * (lhw_ppo_set_inference_dtype) the view asks for fp16 operands: the in-wave policy step then rounds weights and activations to fp16 and
 * accumulates in float32 over ascending k -- the fp16 MFMA of the launch-per-step path adds its 16 products per instruction in its own
 * order, so in THAT mode the two rollouts agree to float32 rounding (1e-6), not bitwise. */
int lhw_ppo_rollout_policy(LhwPpo* ppo, const float* theta, const float* obs_mean, const float* obs_std, uint64_t seed,
                           uint32_t counter, int deterministic, LhwRolloutPolicy* out);
/* lhw_ppo_forward on workspace rows [ws_row, ws_row + N): calls issued on different streams for disjoint env groups may run
 * concurrently (env_id_base must be the global id of the group's first env) */
int lhw_ppo_forward_at(LhwPpo* ppo, const float* theta, const float* obs, int64_t N, const float* obs_mean,
                       const float* obs_std, uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic,
                       int64_t ws_row, float* mu, float* act, float* logp, float* value, void* stream);
/* fp16 != 0: lhw_ppo_forward (rollout inference) rounds weights and activations to fp16 and multiplies on the fp16 MFMA with
 * float32 accumulation (BASELINE config "fp16 actor/critic"); the update (lhw_ppo_grad) uses float32 operands unless
 * lhw_ppo_set_update_dtype selects fp16 as well */
int lhw_ppo_set_inference_dtype(LhwPpo* ppo, int fp16);
/* fp16 != 0: every GEMM of lhw_ppo_grad (forward, activation gradients, weight gradients) rounds both operands to fp16 and
 * runs on the fp16 MFMA with float32 accumulation -- BASELINE config 5 "fp16 actor/critic": fp16 weights and activations per
 * GEMM, float32 master weights, loss, gradient accumulation and Adam.  The reference has no counterpart (it trains in float32). */
int lhw_ppo_set_update_dtype(LhwPpo* ppo, int fp16);
/* time-major [T][N] GAE(lambda); done holds LHW_DONE_* flags, vterm the critic value of the terminal
 * observation, vfinal [N] the value of the observation after the last step */
int lhw_gae(int32_t T, int32_t N, const float* rew, const float* val, const uint8_t* done, const float* vterm,
            const float* vfinal, double gamma, double lam, float* ret, float* adv, void* stream);
int lhw_moments(const float* x, int64_t n, double* out2_dev, void* stream);
int lhw_scale_shift(float* x, int64_t n, float mean, float inv_scale, void* stream);
/* x <- (x - mean) / (std + eps), mean and UNBIASED std taken on the device from stats3_dev = {sum, sum of squares, count} (float64;
 * e.g. lhw_moments' output with the count appended, summed over ranks by an all-reduce): the advantage normalisation of
 * rl/algos/ppo.py:484-485 over the global batch without a device -> host round trip */
int lhw_standardize(float* x, int64_t n, const double* stats3_dev, double eps, void* stream);
/* Imitation term of the NEXT lhw_ppo_grad call (reference rl/algos/ppo.py:360-368, rl/algos/imitation.py): the host
 * evaluates env.imitation_projector() and the frozen expert policy, and passes the expert means scattered into a dense
 * [B][act_dim] target with a [B][act_dim] 0/1 mask (minibatch row order), the coefficient and the number of selected
 * entries (denominator of the mean).  NULL target disarms. */
int lhw_ppo_set_imitation(LhwPpo* ppo, const float* target, const uint8_t* mask, float coeff, int64_t n_selected);
/* forward + loss + backward of one minibatch; accumulates into grad and stats_dev[0..5]
 * (actor, critic, mirror, approx_kl, clip_fraction, imitation) */
int lhw_ppo_grad(LhwPpo* ppo, const float* theta, float* grad, const float* xn, const float* xm, const float* act,
                 const float* old_logp, const float* adv, const float* ret, const int32_t* idx, int32_t B,
                 float* stats_dev, void* stream);
/* dual clip_grad_norm_ + Adam; zeroes grad */
int lhw_ppo_apply(LhwPpo* ppo, float* theta, float* grad, float* adam_m, float* adam_v, int64_t step, float grad_scale,
                  void* stream);
/* lhw_ppo_grad followed by lhw_ppo_apply as ONE launch: the optimiser step of rl/algos/ppo.py:387-396 (zero_grad, backward, two
 * clip_grad_norm_, two Adam steps) captured once as a hipGraph per (buffers, minibatch size) and replayed, the minibatch's index pointer
 * and Adam's bias corrections patched into the graph's kernel nodes.  Bitwise the result of the two calls.  Single process only: with data
 * parallelism the gradient all-reduce belongs between lhw_ppo_grad and lhw_ppo_apply. */
int lhw_ppo_step(LhwPpo* ppo, float* theta, float* grad, float* adam_m, float* adam_v, const float* xn, const float* xm, const float* act,
                 const float* old_logp, const float* adv, const float* ret, const int32_t* idx, int32_t B, float* stats_dev, int64_t step,
                 float grad_scale, void* stream);

/* ------------------------------------------------------------------ recurrent PPO (LSTM actor / critic)
 * Gaussian_LSTM_Actor / LSTM_V: two stacked LSTMCells (hidden) + linear read-out (reference rl/policies/actor.py:191-286,
 * critic.py:52-112); rollout one cell step per control step with state reset at episode starts
 * (rl/workers/rollout_worker.py:134-137,174-177); update by BPTT over whole trajectories (rl/algos/ppo.py:512-533) --
 * a minibatch is a set of env columns of the time-major rollout, with the state zeroed where an episode starts inside a
 * column (equivalent to the reference's padded trajectory list with masked losses).  Uses LhwPpoConfig (max_rows unused). */
typedef struct LhwRnn LhwRnn;
int lhw_rnn_create(const LhwPpoConfig* cfg, int32_t seq_len, int32_t seq_cols, int32_t rollout_rows, LhwRnn** out);
int lhw_rnn_destroy(LhwRnn* rnn);
int64_t lhw_rnn_param_count(const LhwRnn* rnn);
/* offsets: [0..7] actor Wcat1 ([4H][pad4(obs)+H] = [W_ih | W_hh]) b_ih1 b_hh1 Wcat2 ([4H][2H]) b_ih2 b_hh2 Wout bout,
 * [8] stds, [9..16] critic likewise, [17] padded obs width, [18] padded actor read-out width */
int lhw_rnn_layout(const LhwRnn* rnn, int64_t* out19);
/* one rollout step on N rows; reset [N] (device, may be NULL) marks rows whose episode starts with this observation;
 * commit != 0 advances the stored hidden state, 0 evaluates only (terminal / final values) */
int lhw_rnn_forward(LhwRnn* rnn, const float* theta, const float* obs, int64_t N, const float* obs_mean, const float* obs_std,
                    const uint8_t* reset, uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic, int commit,
                    float* mu, float* act, float* logp, float* value, void* stream);
/* BPTT of one minibatch = columns cols[0..B) of the time-major [T][N] rollout (xn / xm [T*N][pad4(obs)] normalised /
 * mirrored observations from lhw_ppo_normalize-compatible layout, done = LHW_DONE_* flags); accumulates grad, stats_dev[0..5] */
int lhw_rnn_grad(LhwRnn* rnn, const float* theta, float* grad, int32_t T, int32_t N, const float* xn, const float* xm,
                 const float* act, const float* old_logp, const float* adv, const float* ret, const uint8_t* done,
                 const int32_t* cols, int32_t B, float* stats_dev, void* stream);
int lhw_rnn_apply(LhwRnn* rnn, float* theta, float* grad, float* adam_m, float* adam_v, int64_t step, float grad_scale,
                  void* stream);
/* xn (and xm) for the recurrent path: same as lhw_ppo_normalize but on an LhwRnn handle */
int lhw_rnn_normalize(LhwRnn* rnn, const float* obs, int64_t R, const float* obs_mean, const float* obs_std, float* xn,
                      float* xm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LHW_H */
