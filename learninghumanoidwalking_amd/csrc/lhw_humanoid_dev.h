// Device side of the wave-per-env humanoid stepper: layouts, the sub-step, the fused control step (control_step<MODE, TASK, W>).
// Included by lhw_humanoid.hip (one control step per launch: humanoid_kernel) and lhw_humanoid_rollout.hip (the resident rollout:
// humanoid_rollout_kernel, policy step + control step for all T steps of a rollout without leaving the wave).
#pragma once
// Rigid-body + soft-contact stepper with the tasks fused in (humanoid_kernel<MODE, TASK, W>: jvrc_walk, jvrc_step, h1,
// h1_walk).  An environment is advanced by a GROUP of W lanes: W = 32 packs two environments into one 64-lane wavefront
// (the fast path of the walking / standing tasks: at most 8 contacts per env), W = 64 gives one environment the whole
// wavefront (16 contacts; the stepping task, reset / state access, and the re-run of envs that exceeded the fast path's
// contact capacity).  The two groups of a wave share the instruction stream and nothing else: every cross-lane operation
// (DPP scans, v_readlane broadcasts, ballots) is group-local, so per-env control flow (Newton iterations, resets) is plain
// SIMT divergence at group granularity.
//
// One group advances one humanoid by a whole control step per launch:
//   frame_skip x { PD law -> forward dynamics -> constraint solve -> Euler }  then
//   task state machine, rewards, termination, observation, and (optionally) the episode
//   bookkeeping + reset of the reference's rollout worker.
// It takes over RobotBase.step/_do_simulation (reference robots/robot_base.py:41-98),
// RobotInterface.step_pd/set_motor_torque/step -> mujoco.mj_step (reference
// envs/common/robot_interface.py:493-546), BaseHumanoidEnv.step/reset_model/get_obs incl. observation / init noise
// (reference envs/common/base_humanoid_env.py:177-338), domain randomisation (envs/common/domain_randomization.py:10-56),
// WalkingTask (tasks/walking_task.py:85-205), SteppingTask (tasks/stepping_task.py:66-334), StandingTask
// (tasks/standing_task.py:49-131) and tasks/rewards.py:9-194.
//
// Physics = the MuJoCo pipeline subset of SURVEY.md Appendix A, float64: kinematics, com-based
// spatial quantities, CRBA (+armature), RNE bias, joint damping, motor actuation with ctrl/force
// clamps, applied Cartesian wrenches, primitive collisions (plane-{sphere,capsule,box}, sphere-sphere, sphere-capsule,
// capsule-capsule; box-box in the stepping-task kernels), frictionloss and joint-limit rows, pyramidal contact rows with
// MuJoCo's impedance / reference acceleration / regulariser model, the primal Newton solver with exact line search and
// warm start, Euler integration with implicit joint damping.
//
// Mapping onto CDNA4: the env's working set (body frames, spatial inertias, contact Jacobian ...) lives in LDS for the whole
// launch.  The dof-indexed work uses the CHAIN LAYOUT: half an env -- the free root's six dofs plus one leg's serial chain --
// per 16-lane DPP row (the root held by both rows), so that broadcasts are v_mov_b64_dpp row_newbcast, the tree recursions of
// kinematics / RNE / CRB are row_shr / row_shl scans, and M, M + h D and the Newton Hessian are factorised as two chains in
// lockstep (reverse L^T D L in registers, no fill-in) with one v_permlane16_swap merge at the root; sub-steps whose contacts
// couple the two legs fall back to a dense Cholesky (one dof per lane, factor row in registers).  The pyramid rows of contact c
// sit in lanes 4c..4c+3 with their Jacobian row in LDS, and frictionloss / joint-limit rows -- unit vectors -- are scalars of
// their dof's lane (no Jacobian storage, no row lanes).  Reductions are DPP scans.  The persistent state is one contiguous
// 1.3 KB record per env, read and written once per control step with lane-strided (coalesced) accesses.  
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "lhw_internal.h"
#include "lhw_rng.h"

#define NB 20   // bodies
#define NVMAX 18   // dofs (capacity of the persistent records; the kernels use the per-task width NV = L::NV_)
#define LDVMAX 19
#define NQ 19
#define NJ 14   // joints
#define NG 32   // geoms
#define NP 64   // collision candidate pairs (one per lane of the group: npair <= W is checked at create)
// contacts kept per step (NC = W / 4) and contact rows (NE = 4 NC, one per lane) follow from the group width: see LdsT
#define NU 12   // actuators
#define HMINVAL 1e-15

enum { JT_FREE = 0, JT_SLIDE = 2, JT_HINGE = 3 };
enum { G_PLANE = 0, G_SPHERE = 2, G_CAPSULE = 3, G_ELLIPSOID = 4, G_CYLINDER = 5, G_BOX = 6 };
enum { MODE_STANDING = 0, MODE_INPLACE = 1, MODE_FORWARD = 2 };
enum { WALK_CURVED = 0, WALK_STANDING = 1, WALK_BACKWARD = 2, WALK_LATERAL = 3, WALK_FORWARD = 4 };  // stepping_task.py:282-285

// persistent record (doubles)
#define R_QPOS 0
#define R_QVEL (R_QPOS + NQ)
#define R_WARM (R_QVEL + NVMAX)
#define R_SQ (R_WARM + NVMAX)       // joint position of each actuator at the last forward pass (actuator_length / gear)
#define R_SV (R_SQ + NU)         // joint velocity, likewise
#define R_FRC (R_SV + NU)        // actuator_force of the last forward pass
#define R_PREVPRED (R_FRC + NU)  // prev_prediction (action smoothing)
#define R_PREVACT (R_PREVPRED + NU)
#define R_PREVTQ (R_PREVACT + NU)
#define R_MODEREF (R_PREVTQ + NU)
#define R_EPRET (R_MODEREF + 3)
#define REC_D 168
static_assert(R_EPRET < REC_D, "record too small");
// int record
#define RI_PHASE 0
#define RI_MODE 1
#define RI_TRAJ 2
#define RI_STEPCNT 3
#define RI_RESETCNT 4
#define RI_STARTED 5  // prev_action / prev_torque initialised (robot_base.py:82-85: only once, never reset)
#define RI_OBSCNT 6   // number of get_obs calls so far (observation-noise RNG counter)
#define RI_T1 7       // stepping task: indices of the current / next target step, hit flag + dwell counter, sequence length
#define RI_T2 8
#define RI_REACHED 9
#define RI_FRAMES 10
#define RI_NSEQ 11
#define REC_I 16
// stepping-task record (doubles): 20 target steps x (x y z theta cos sin) = poses of the 20 terrain boxes, floor height
#define T_SEQ 0
#define T_FLOOR 120
#define TER_D 128
#define MAX_SEQ 20
// per-env model parameters touched by dynamics randomisation / perturbation (domain_randomization.py:10-56)
#define P_DAMP 0
#define P_FLOSS (P_DAMP + NVMAX)
#define P_MASS (P_FLOSS + NVMAX)
#define P_IPOS (P_MASS + NB)
#define P_XFRC (P_IPOS + NB * 3)  // xfrc_applied of up to two perturbed bodies: force3 torque3 each
#define PRM_D 128
static_assert(P_XFRC + 12 <= PRM_D, "parameter record too small");
enum { TASK_WALK = 1, TASK_STAND = 2, TASK_STEP = 3, TASK_H1WALK = 4 };
enum { LHW_STREAM_OBS = 4 };

// Model constants, packed host-side into one array-of-structs table per "lane role" (body, joint, dof, geom, pair,
// actuator): a lane fetches the record of its body/dof/... with one burst of independent loads, and the kernel argument
// block carries a dozen base pointers instead of seventy (the latter spilled most of the SGPR file).
// body_d: pos3 R_body9 ipos3 R_inertial9 inertia3 jnt_axis3 jnt_pos3 qpos0(joint) R_body*jnt_pos3 R_body*jnt_axis3 | mass invweight0[2]
#define BDS 44
#define BD_POS 0
#define BD_RBODY 3
#define BD_IPOS 12
#define BD_RINERT 15
#define BD_INERTIA 24
#define BD_JAXIS 27
#define BD_JPOS 30
#define BD_Q0 33
#define BD_V1 34
#define BD_V2 37
#define BD_NKIN 40
#define BD_MASS 40
#define BD_INVW 41
#define BIS 8   // body_i: parent level jnt_type(-1 welded) qposadr | rootid subtree_end dofmask jntadr
#define BI_ROOT 4
#define BI_SUBEND 5
#define BI_DOFMASK 6
#define BI_JNTADR 7
#define BI_LEVEL 1
#define JDS 10  // jnt_d: range2 solref2 solimp5 margin
#define JD_RANGE 0
#define JD_SOLREF 2
#define JD_SOLIMP 4
#define JD_MARGIN 9
#define JIS 4   // jnt_i: type limited qposadr dofadr
#define JI_TYPE 0
#define JI_LIMITED 1
#define JI_QADR 2
#define JI_DADR 3
#define DDS 16  // dof_d: armature damping invweight0 frictionloss solref2 solimp5 gear(of the dof's actuator) | range2 margin of its joint (copies) pad
#define DD_ARMATURE 0
#define DD_DAMPING 1
#define DD_INVW 2
#define DD_FLOSS 3
#define DD_SOLREF 4
#define DD_SOLIMP 6
#define DD_GEAR 11
#define DD_RANGE 12
#define DD_MARGIN 14
#define DIS 8   // dof_i: body joint kind(0 free-trans 1 free-rot 2 slide 3 hinge) prevmask actuator(-1 none) | index within its joint,
                //        joint limited, joint qposadr (copies of the joint record: one table round trip where the dof is the key)
#define DI_BODY 0
#define DI_JNT 1
#define DI_KIND 2
#define DI_PREVMASK 3
#define DI_ACT 4
#define DI_KIDX 5
#define DI_LIMITED 6
#define DI_QADR 7
#define GDS 28  // geom_d: pos3 R_local9 size3 friction3 solmix solref2 solimp5 margin gap
#define GD_POS 0
#define GD_RLOC 3
#define GD_SIZE 12
#define GD_FRICTION 15
#define GD_SOLMIX 18
#define GD_SOLREF 19
#define GD_SOLIMP 21
#define GD_MARGIN 26
#define GD_GAP 27
#define GIS 4   // geom_i: type body condim priority
#define GI_TYPE 0
#define GI_BODY 1
#define GI_CONDIM 2
#define GI_PRIORITY 3
#define ADS 6   // act_d: gear ctrlrange2 forcerange2 pad
#define AD_GEAR 0
#define AD_CTRLRANGE 1
#define AD_FORCERANGE 3
#define PIS 12  // pair_i: geom1 geom2 condim xmask(dofs moving exactly one body) mask2(dofs moving body 2) class | merge class, robot-is-geom1 |
                //         type of geom1, of geom2 (copies: the narrow phase reads its pair record only, one table round trip) |
                //         body of geom2, root body of geom1's body (the ground-reaction query of the stepping task)
#define PDS 20  // pair_d: margin includemargin friction solref2 solimp5 invweight(sum of the two bodies' translational) | size3 of geom1, of geom2 |
                //         bounding radius of geom1, of geom2 (the broad-phase test of fwd_collision) | pad
#define PD_SIZE1 11
#define PD_SIZE2 14
#define PD_RBOUND1 17
#define PD_RBOUND2 18
#define PI_TYPE1 8
#define PI_TYPE2 9
#define PI_BODY2 10
#define PI_ROOT1 11
#define AIS 6   // act_i: dof joint ctrllimited forcelimited qposadr dofadr(of the joint)
#define AI_QADR 4
#define AI_DADR 5
#define AI_DOF 0
#define AI_JNT 1
#define AI_CTRLLIMITED 2
#define AI_FORCELIMITED 3

// The model / task tables are read through pointers that are themselves loaded from HModel / HParams in device memory; the
// compiler cannot tell where such a pointer points and would read the tables with FLAT loads -- 64-bit address arithmetic per
// access, and a flat load counts on the LDS counter too, so every wait for an LDS read would also wait for the table loads in
// flight.  Declared as global-address-space pointers (device pass only; the host pass and the SIMT emulator see plain pointers)
// they become global loads off a scalar base, ordered independently of the LDS traffic.
#if defined(__HIP_DEVICE_COMPILE__)
#define LHW_GLOBAL_AS __attribute__((address_space(1)))
#else
#define LHW_GLOBAL_AS
#endif
typedef const double LHW_GLOBAL_AS* gtab_d;
typedef const int LHW_GLOBAL_AS* gtab_i;
typedef double LHW_GLOBAL_AS* gws_d;   // (the many-contact workspace of the stepping task: read and written)
typedef int LHW_GLOBAL_AS* gws_i;
template <typename T>
struct DevTab {   // what to_dev returns: converts to the table pointer type of either pass
  const T* p;
  operator const T*() const { return p; }
#if defined(__HIP_DEVICE_COMPILE__)
  operator const T LHW_GLOBAL_AS*() const { return (const T LHW_GLOBAL_AS*)p; }
#endif
};

#define KIS 4    // kin_i: body jnt_type qposadr joint
#define KDS 20   // kin_d: R0[9] p0[3] jnt_axis[3] jnt_pos[3] qpos0 pad   (R0, p0: frame relative to the previous jointed body)
#define MAX_OWNED 8   // bodies per lane of the chain layout (chain_dynamics' per-body loop; checked at create)
// Round 6: the model tables are MEMBERS of HModel (fixed capacities), not pointers to seventeen separate allocations: every table
// read of the kernels is a global load off ONE base -- the kernel's HModel argument -- plus a compile-time offset.  Before, each
// table access first fetched the table's own pointer from HModel with an s_load (about sixty scalar loads per wave and sub-step,
// each answered on the counter the LDS reads use), and the pointers that stayed live crowded the SGPR file (SGPR spills of the
// walking rollout kernel: 499 -> see DESIGN.md section 4).  The struct is 45 KB; the kernels see it through an address-space-1 reference
// (HModelRef) so that also the functions that are not inlined read it with global / scalar loads instead of flat ones.
struct HModel {
  int nq, nv, nu, nbody, njnt, ngeom, npair, nlevel, iterations, disableflags;
  double timestep, gravity[3], tolerance, meaninertia, totalmass;
  int has_primbox;     // some collision pair is sphere-box or capsule-box (collide_primbox)
  int has_cyl;         // some collision pair is plane-cylinder or sphere-cylinder (collide_cyl)
  int npb, pb_pair[4]; // plane-box pairs (floor against a foot box), in pair order, if there are at most four of them (else npb = 0):
                       // their eight corners are tested on eight lanes each instead of one after the other on the pair's lane
  int max_owned;         // bodies per lane in chain_dynamics' per-body loop (<= MAX_OWNED)
  int track_body[3];  // bodies whose spatial velocity must survive the sub-step (the task reads them afterwards)
  double track_off[9];  // local offset of the tracked point on each of them (foot force sites for the stepping task)
  double body_d[NB * BDS], jnt_d[NJ * JDS], dof_d[NVMAX * DDS], geom_d[NG * GDS], act_d[NU * ADS];
  double pair_d[NP * PDS];   // pair_i / pair_d: one record per candidate pair (mj_contactParam is a function of the pair)
  double kin_d[32 * KDS];    // kin_i [32][KIS], kin_d [32][KDS]: per lane of the chain layout, its jointed body and that body's frame relative
                             //   to the previous jointed body (fwd_kinematics)
  double fix_d[NB * 12];     // fix_i [nbody]: jointed body a welded body moves with (-1 for jointed bodies), fix_d [nbody][12]: its frame in that body's
  int body_i[NB * BIS], jnt_i[NJ * JIS], dof_i[NVMAX * DIS], geom_i[NG * GIS], act_i[NU * AIS], pair_i[NP * PIS], kin_i[32 * KIS], fix_i[NB];
  int own_tab[32 * MAX_OWNED];   // [32][max_owned]: bodies whose force / inertia the lane of the chain layout contributes (-1: none)
};

struct HParams {
  int n_envs, frame_skip, max_traj_len, period, task;
  int reset_template;                   // >= 0: record index of the template env whose freshly reset state every auto-reset copies (-1: resets are computed)
  int root_body, head_body, rfoot_body, lfoot_body;
  int env_params;                       // 1: damping / frictionloss / mass / ipos / xfrc come from the per-env record
  int dynrand_interval, perturb_interval, n_pbody, pbody[2];
  int rand_dof[10], rand_body[11], n_rand_dof, n_rand_body;
  int box_geom0, nbox, floor_geom, delay_frames, nplans;  // stepping task
  double target_radius;
  gtab_d plans;                         // [nplans][1 + MAX_SEQ * 3]: length, then (x y theta) rows
  unsigned env_id_base;
  unsigned long long seed;
  double action_smoothing, goal_height, init_noise, force_mag, torque_mag;
  gtab_d clock_lut;
  gtab_d xfrc_base;                     // JVRC tasks with perturbations: the per-env parameter records [N][PRM_D] (chain_dynamics reads P_XFRC of its env); else NULL
  double kp[NU], kd[NU], nominal_qpos[NQ], action_offset[NU], neutral_pose[NU], obs_noise[36];   // (members, like the model tables)
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const HModel LHW_GLOBAL_AS& HModelRef;
typedef const HParams LHW_GLOBAL_AS& HParamsRef;
#else
typedef const HModel& HModelRef;
typedef const HParams& HParamsRef;
#endif

// what changes from launch to launch (everything else of the task configuration sits in device memory: HumanoidEnv::p_dev)
struct HLaunch {
  int env_first, env_count;             // sub-range of envs this launch advances (lhw_env_step_range); blockIdx.x is relative to it
  int only_flagged;                     // 1: advance only the envs whose st.slow flag is set (re-run of fast-path overflows), clearing it
  int iteration;                        // training iteration (stepping-task curriculum)
  long long tin_off;                    // offset (doubles) of this control step's slice of st.tin: 0 for the per-launch record; t * N * LHW_TASK_INPUT_DIM
                                        // when a resident rollout exports the record of EVERY control step (lhw_env_rollout_task_inputs)
};

struct HState {
  double* rec;     // [N][REC_D]
  int* irec;       // [N][REC_I]
  double* prm;     // [N][PRM_D] per-env model parameters (NULL unless the task randomises them)
  double* ter;     // [N][TER_D] stepping-task record (NULL for the other tasks)
  double* ep_stats;
  unsigned char* slow;  // [N] set by the two-envs-per-wave kernel for an env that exceeded its contact capacity: nothing of that env
                        // was written, and the one-env-per-wave kernel repeats its control step
  long long* prof; // optional [16] per-phase cycle counters accumulated by env 0 (NULL = off)
  double* tin;     // optional [N][LHW_TASK_INPUT_DIM]: the task layer's inputs of the last control step (lhw_env_enable_task_inputs)
  long long* wave_cyc;  // optional [N] shader-clock cycles the env's group spent in the last control-step launch (NULL = off)
  double* bigd;    // stepping task: [N][BW_DOUBLES] / [N][BW_INTS] workspace of the many-contact path (NULL for the other tasks)
  int* bigi;
};
// Analysis builds (-DLHW_FINEPROF=<phase slot>): the phase of that slot is split further, FINE_MARK(phase, i) accumulating the
// clock of env 0 into g_fine[i] (read back through lhw_env_profile in place of the phase table).  Compiled out of the product.
#ifdef LHW_FINEPROF
static __device__ long long g_fine[16];   // (per translation unit: analysis builds read the step kernels' copy)
static __device__ long long g_fine_t;
#define FINE_ON(ph) (LHW_FINEPROF == (ph) && blockIdx.x == 0 && threadIdx.x == 0)
#define FINE_BEGIN(ph) do { if (FINE_ON(ph)) g_fine_t = (long long)clock64(); } while (0)
#define FINE_MARK(ph, i) do { if (FINE_ON(ph)) { const long long n_ = (long long)clock64(); g_fine[i] += n_ - g_fine_t; g_fine_t = n_; } } while (0)
#else
#define FINE_BEGIN(ph)
#define FINE_MARK(ph, i)
#endif
#define PROF_BEGIN() long long prof_t = (st_prof && lane == 0) ? (long long)clock64() : 0   // lane = lane within the group
#ifdef LHW_ASM_MARKS   // (analysis builds: phase boundaries as comments in the ISA listing)
#define ASM_MARK(slot) asm volatile("; LHW_PHASE " #slot)
#else
#define ASM_MARK(slot)
#endif
#define PROF_MARK(slot)                                                  \
  do {                                                                   \
    ASM_MARK(slot);                                                      \
    if (st_prof && lane == 0) {                                          \
      long long now_ = (long long)clock64();                             \
      st_prof[slot] += now_ - prof_t;                                    \
      prof_t = now_;                                                     \
    }                                                                    \
  } while (0)

// Many contacts in the stepping task (one env per wave).  Every walk mode but FORWARD leaves the 20 terrain boxes coplanar with the
// floor (tasks/stepping_task.py:320-334), so a foot rests on the floor AND on every box under it: 16 (STANDING) to ~110 (LATERAL)
// contacts per env, against 16 whose pyramid rows fit one lane each.  Two steps deal with them:
//  1. MERGING.  Most of those contacts are copies of one another -- a foot corner that lies inside twelve overlapping boxes yields
//     twelve box contacts with bitwise the same distance, position and frame, and the floor contact of that corner is the same
//     constraint once more with the roles of the two geoms swapped (frame mirrored, values within 1-2 ulp): 106 contacts, 12-14
//     distinct ones in LATERAL mode.  k identical rows of the soft-constraint problem are ONE row with k times the D (cost
//     k (1/2) D r^2, total force k f): the collision stage writes all contacts of such a sub-step to an HBM workspace ("raw" region),
//     merges the copies and hands the distinct contacts on with their multiplicity (and, for the ground-reaction query, the share
//     of the copies that are floor contacts of a foot).  If at most 16 remain -- nearly always -- they go back into the LDS arrays
//     and the ordinary one-row-per-lane solver runs, with D scaled by the multiplicity.
//  2. MANY DISTINCT CONTACTS (more than 16 after merging): they stay in the workspace ("unique" region) with the per-row solver
//     state, and newton_big walks the rows in chunks of 64.
#define NCR 192                            // raw contacts per sub-step (beyond: dropped and counted as a contact overflow)
#define AR_DIST 0
#define AR_POS (AR_DIST + NCR)
#define AR_FRAME (AR_POS + 3 * NCR)
#define AR_DOUBLES (AR_FRAME + 9 * NCR)
#define ARI_G1 0
#define ARI_G2 (ARI_G1 + NCR)
#define ARI_PAIR (ARI_G2 + NCR)
#define AR_INTS (ARI_PAIR + NCR)            // (the merge's own bookkeeping -- first copy, rank, multiplicity, foot counts -- lives in LDS: MergeLds)
#define NCB 64                             // distinct contacts the many-contact solver holds (4 NCB rows)
#define NRB (4 * NCB)
#define BW_DIST AR_DOUBLES
#define BW_POS (BW_DIST + NCB)
#define BW_FRAME (BW_POS + 3 * NCB)
#define BW_MU (BW_FRAME + 9 * NCB)
#define BW_MARGIN (BW_MU + NCB)
#define BW_TRAN (BW_MARGIN + NCB)
#define BW_SOLREF (BW_TRAN + NCB)
#define BW_SOLIMP (BW_SOLREF + 2 * NCB)
#define BW_MULT (BW_SOLIMP + 5 * NCB)      // multiplicity of the contact
#define BW_WR (BW_MULT + NCB)              // share of the copies that are floor contacts of the right / left foot
#define BW_WL (BW_WR + NCB)
#define BW_D (BW_WL + NCB)               // per row (4 c .. 4 c + 3: the pyramid edges of contact c): multiplicity / R; 0: not a row
#define BW_AREF (BW_D + NRB)
#define BW_JAR (BW_AREF + NRB)            // J a - aref at the current iterate
#define BW_JV (BW_JAR + NRB)              // J search
#define BW_FRC (BW_JV + NRB)              // efc_force (of the merged row: the sum over its copies)
#define BW_DACT (BW_FRC + NRB)            // D of the active rows, 0 for the others
#define BW_DOUBLES (BW_DACT + NRB)
#define BWI_G1 AR_INTS
#define BWI_G2 (BWI_G1 + NCB)
#define BWI_PAIR (BWI_G2 + NCB)
#define BWI_DIM (BWI_PAIR + NCB)
#define BWI_XM (BWI_DIM + NCB)
#define BWI_M2 (BWI_XM + NCB)
#define BW_INTS (BWI_M2 + NCB)

struct HumanoidEnv {
  HModel m;
  HParams p;
  HModel* m_dev;    // device copy of m (same reason)
  HParams* p_dev;   // device copy the kernels read (passed by pointer: its fields need not live in SGPRs across the sub-steps)
  int iteration;
  HState st;
  std::vector<void*> dev_allocs;
  int device;
  bool fast;   // the model fits the two-envs-per-wave kernels (W = 32)
  unsigned* ro_queue = nullptr;   // job counter + per-group progress words of the resident rollout's queue mode (lhw_humanoid_rollout.hip)
  int last_rollout_queued = 0;    // the most recent resident rollout drained the job queue (humanoid_rollout_kernel<.., QUEUE = true>)
};

// ------------------------------------------------------------------------------------------------ LDS working set
// Working set of ONE env (= one group of W lanes; a wave of the W = 32 kernels holds two of these).  Members that are
// live across stages are plain fields; everything that is only live inside one stage of the sub-step shares the region U:
//   cdof                              : com -> contact Jacobian
//   cinert                            : kinematics (rotated inertia) -> RNE -> CRBA
//   stage A  kinematics + collision   : xmat xipos xanchor xaxis gpos gmat            (over X)
//   stage B1 velocity / RNE           : cdofdot cvel cacc=cfrc, subtree sums over cvel (over X)
//   stage B2 CRBA                     : crb buf M                                     (over X; M goes to registers at once)
//   stage C  constraints + solve      : J (over cinert and X), the packed factor of the dense fallback solve (U_L)
#define NE (L::NE_)      // contact rows = lanes of the group: rows 4c .. 4c+3 belong to contact c
#define NC (L::NC_)      // contacts kept per sub-step
#define NV (L::NV_)      // dof width the kernel is compiled for (18 JVRC, 16 H1): sizes the Cholesky, the row products, the LDS matrices
#define U_CDOF (L::U_CDOF_)
#define U_CINERT (L::U_CINERT_)
#define U_XMAT (L::U_XMAT_)
#define U_XIPOS (L::U_XIPOS_)
#define U_XANCHOR (L::U_XANCHOR_)
#define U_XAXIS (L::U_XAXIS_)
#define U_GPOS (L::U_GPOS_)
#define U_GMAT (L::U_GMAT_)
#define U_CDOFDOT (L::U_CDOFDOT_)
#define U_CVEL (L::U_CVEL_)
#define U_CACC (L::U_CACC_)
#define U_CFRC (L::U_CACC_)   // cfrc overwrites cacc body by body
#define U_CSUB (L::U_CVEL_)   // subtree force sums overwrite cvel (the tracked velocities are copied out first)
#define U_CRB (L::U_CRB_)
#define U_BUF (L::U_BUF_)
#define U_M (L::U_M_)
#define U_CDIST (L::U_CDIST_)
#define U_CMARGIN (L::U_CMARGIN_)
#define U_CSOLREF (L::U_CSOLREF_)
#define U_CSOLIMP (L::U_CSOLIMP_)
#define U_CFRAME (L::U_CFRAME_)
#define U_CTRAN (L::U_CTRAN_)    // body_invweight0 (translational) of the contact's two bodies, summed
#define U_J (L::U_J_)
#define U_L (L::U_L_)
#define U_VEC (L::U_VEC_)
#define U_VEC2 (L::U_VEC2_)
#define U_DG (L::U_DG_)
#define U_EVEC (L::U_EVEC_)
#define U_DACT (L::U_DACT_)
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))   // packed lower triangle, row-major (i >= j)
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int even(int a) { return (a + 1) & ~1; }

// The layout is a template of the group width (contact / row capacity), the dof width, the geom and body capacities and the
// task features that need extra state, so that the two-envs-per-wave kernels of the walking / standing tasks stay within
// 10 KB per env: eight wavefronts = all 4096 envs of the headline batch are resident at once on the 256 CUs.
// LDS state that only the stepping-task layouts carry (an empty base otherwise: the walking / standing layouts are sized to the
// byte for eight workgroups per CU)
template <bool ON, int NC_T>
struct StepLds {
  static constexpr int NCK_ = 32;
  double con_mult[NC_T];   // number of identical contacts this one stands for (scales the D of its rows; fwd_collision's merge)
  double con_wr[NC_T], con_wl[NC_T];   // share of those copies that are floor contacts of the right / left foot (the GRF query)
  // many-contact path: number of distinct contacts of the last forward pass kept in the HBM workspace (0: they are in the LDS
  // arrays) and what the task layer reads off them (robot_interface.py:262-325, 472-484)
  int nbig, big_selfcol, big_anyfoot;
  double big_grf_r, big_grf_l, big_cz;
  // newton_big: position / frame / friction and dof masks / condim of up to NCK contacts, cached for the rebuilds of their rows
  alignas(16) double bk_rec[NCK_ * 13];
  int bk_i[NCK_ * 3];
};
template <int NC_T> struct StepLds<false, NC_T> { static constexpr int NCK_ = 1; };

template <int W_T, bool PRM_T, int NV_T, int NG_T, int NB_T, bool STEP_T>
struct LdsT : StepLds<STEP_T, W_T / 4> {
  typedef LdsT L;
  static constexpr int W_ = W_T, NC_ = W_T / 4, NE_ = W_T, NV_ = NV_T, NG_ = NG_T, NB_ = NB_T, TRI_ = NV_T * (NV_T + 1) / 2;
  static constexpr bool PRM_ = PRM_T;   // per-env model parameters are staged in LDS (else read from the model tables)
  static constexpr bool STEP_ = STEP_T;
  static constexpr int U_CDOF_ = 0, U_CINERT_ = U_CDOF_ + NV_T * 6, X_ = U_CINERT_ + NB_T * 10;
  static constexpr int U_XMAT_ = X_, U_XIPOS_ = U_XMAT_ + NB_T * 9, U_XANCHOR_ = U_XIPOS_ + NB_T * 3, U_XAXIS_ = U_XANCHOR_ + NJ * 3,
                       U_GPOS_ = U_XAXIS_ + NJ * 3, U_GMAT_ = U_GPOS_ + NG_T * 3, END_A_ = U_GMAT_ + NG_T * 9;
  static constexpr int U_CDOFDOT_ = X_, U_CVEL_ = U_CDOFDOT_ + NV_T * 6, U_CACC_ = U_CVEL_ + NB_T * 6, END_B1_ = U_CACC_ + NB_T * 6;
  static constexpr int U_CRB_ = X_, U_BUF_ = U_CRB_ + NB_T * 10, U_M_ = U_BUF_ + NV_T * 6, END_B2_ = U_M_ + TRI_ + 1;   // (M is followed by one zero: chain_idx)
  // contact records that only feed the Jacobian / row parameters: written by the collision stage, live across stage B,
  // dead once the rows are built (the solver's vectors and the Cholesky rows then reuse the space)
  static constexpr int U_J_ = U_CINERT_, U_L_ = U_J_ + W_T * NV_T;
  static constexpr int U_CDIST_ = even(cmax(cmax(cmax(END_A_, END_B1_), END_B2_), U_L_)), U_CMARGIN_ = U_CDIST_ + NC_, U_CSOLREF_ = U_CMARGIN_ + NC_,
                       U_CSOLIMP_ = U_CSOLREF_ + 2 * NC_, U_CFRAME_ = U_CSOLIMP_ + 5 * NC_, U_CTRAN_ = U_CFRAME_ + 9 * NC_, END_CON_ = U_CTRAN_ + NC_;   // (beyond J: they feed its rows)
  static constexpr int U_VEC_ = even(U_L_ + TRI_), U_VEC2_ = U_VEC_ + NV_T, U_DG_ = U_VEC2_ + NV_T,
                       U_EVEC_ = U_DG_ + NV_T, U_DACT_ = U_EVEC_ + W_T, END_C_ = U_DACT_ + W_T;   // J rows are NV long (16-byte aligned)
  static constexpr int USIZE_ = even(cmax(END_CON_, END_C_));
  double qpos[NQ], qvel[NV_T], ctrl[NU];
  double xpos[NB_T * 3];
  double rootmat[9], com[4], svel[18];   // root xmat; tree com; cvel of the three tracked bodies (root, right foot, left foot)
  double spos[STEP_T ? 9 : 1], rootquat[STEP_T ? 4 : 1];   // stepping task: world position of the tracked points (body origin + local offset); root xquat
  double qacc[NV_T];
  double efc_force[W_T];
  double con_pos[NC_ * 3], con_mu[NC_];
  int con_g1[NC_], con_g2[NC_], con_dim[NC_];
  int con_xm[NC_], con_m2[NC_], con_pair[NC_];   // dofs that move exactly one of the contact's two bodies; dofs that move body 2 (sign of the Jacobian)
  double sq[NU], sv[NU], frc[NU];
  // per-env parameters, loaded once per launch (one-element stubs when the task reads the shared model tables instead)
  double damp[PRM_T ? NV_T : 1], floss[PRM_T ? NV_T : 1], bmass[PRM_T ? NB_T : 1], bipos[PRM_T ? NB_T * 3 : 1], xfrc[PRM_T ? 12 : 1];
  alignas(16) double U[USIZE_];
  // episode / task context of the env (home of these values during the launch: nothing of it is held in registers across a sub-step)
  double cmode_ref[3], cep_ret;
  int ci[12];
  // Float32 staging that is only live BETWEEN sub-steps, in the tail of the (then dead) stage region: the env's current
  // observation (written after the last sub-step, copied out to the obs / terminal-obs buffers).
  static constexpr int OBSF_ = USIZE_ - 24;
  __device__ __forceinline__ float* obsf() { return reinterpret_cast<float*>(U + OBSF_); }
  int ncon, overflow;
  int env_id;   // index of the env this group advances (what chain_dynamics needs to find the env's applied wrenches in HBM: JVRC perturbations)
};
enum { CI_PHASE = 0, CI_MODE, CI_TRAJ, CI_STARTED, CI_STEPCNT, CI_RESETCNT, CI_OBSCNT, CI_T1, CI_T2, CI_REACHED, CI_FRAMES, CI_NSEQ };
#define CTX_LOAD()                                                                                                     \
  double mode_ref[3] = {S.cmode_ref[0], S.cmode_ref[1], S.cmode_ref[2]};                                               \
  double ep_ret = S.cep_ret;                                                                                           \
  int phase = S.ci[CI_PHASE], mode = S.ci[CI_MODE], traj_len = S.ci[CI_TRAJ], started = S.ci[CI_STARTED];              \
  unsigned step_count = (unsigned)S.ci[CI_STEPCNT], reset_count = (unsigned)S.ci[CI_RESETCNT], obs_count = (unsigned)S.ci[CI_OBSCNT]; \
  int t1 = S.ci[CI_T1], t2 = S.ci[CI_T2], reached = S.ci[CI_REACHED], frames = S.ci[CI_FRAMES], nseq = S.ci[CI_NSEQ];  \
  double goal[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CTX_STORE()                                                                                                    \
  do {                                                                                                                 \
    SYNC();                                                                                                            \
    if (lane == 0) {                                                                                                   \
      S.cmode_ref[0] = mode_ref[0]; S.cmode_ref[1] = mode_ref[1]; S.cmode_ref[2] = mode_ref[2]; S.cep_ret = ep_ret;    \
      S.ci[CI_PHASE] = phase; S.ci[CI_MODE] = mode; S.ci[CI_TRAJ] = traj_len; S.ci[CI_STARTED] = started;              \
      S.ci[CI_STEPCNT] = (int)step_count; S.ci[CI_RESETCNT] = (int)reset_count; S.ci[CI_OBSCNT] = (int)obs_count;      \
      S.ci[CI_T1] = t1; S.ci[CI_T2] = t2; S.ci[CI_REACHED] = reached; S.ci[CI_FRAMES] = frames; S.ci[CI_NSEQ] = nseq;  \
    }                                                                                                                  \
    SYNC();                                                                                                            \
  } while (0)

// 0, produced where the optimiser cannot see it (and cannot move it out of a loop)
__host__ __device__ __forceinline__ int opaque_zero() {
#if defined(__HIP_DEVICE_COMPILE__)
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
#else
  static volatile int z = 0;   // (host pass / SIMT emulator)
  return z;
#endif
}

// A value the optimiser must take as it finds it HERE: everything computed from it stays behind this point.  Used at the entry of
// rarely executed code (the dense fallback solve) so that its lane predicates -- dozens of 64-bit masks -- are formed inside the
// branch instead of being hoisted above it, held in SGPRs across the common path and spilled there (round 6: 136 v_writelane per
// sub-step in front of the Newton loop were the masks of a solve that runs in one sub-step of a thousand).
__host__ __device__ __forceinline__ int opaque_int(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}
// The lanes of a group belong to one wavefront, and a wave's LDS instructions execute in issue order, so cross-lane
// hand-offs through LDS need no hardware barrier and no s_waitcnt: __syncthreads() would add a workgroup-scope fence, i.e.
// s_waitcnt vmcnt(0) lgkmcnt(0) -- a full drain of outstanding global loads -- ~60 times per sub-step.  A wavefront-scope
// fence pair plus the compiler-only wave barrier keeps the compiler from moving LDS accesses across the hand-off and emits
// no instruction.
#define SYNC()                                               \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)
// Reconvergence point after a lane-divergent region that contains cross-lane operations (the hardware reconverges by
// itself; the SIMT emulator of tests/emu needs to be told, its lanes being free-running fibers)
#if defined(__HIP_EMU__)
#define GROUP_SYNC(W) emu_group_sync(W)
#else
#define GROUP_SYNC(W) ((void)0)
#endif
template <class L> __device__ __forceinline__ double prm_damp(HModelRef m, const L& S, int d) {
  if constexpr (L::PRM_) return S.damp[d]; else return m.dof_d[DDS * d + DD_DAMPING];
}
template <class L> __device__ __forceinline__ double prm_floss(HModelRef m, const L& S, int d) {
  if constexpr (L::PRM_) return S.floss[d]; else return m.dof_d[DDS * d + DD_FLOSS];
}
template <class L> __device__ __forceinline__ double prm_mass(HModelRef m, const L& S, int b) {
  if constexpr (L::PRM_) return S.bmass[b]; else return m.body_d[BDS * b + BD_MASS];
}
template <class L> __device__ __forceinline__ double prm_ipos(HModelRef m, const L& S, int b, int a) {
  if constexpr (L::PRM_) return S.bipos[3 * b + a]; else return m.body_d[BDS * b + BD_IPOS + a];
}

// ------------------------------------------------------------------------------------------------ small math
// Group reductions on the DPP path (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast 15 [and 31 for W = 64] across
// rows: an inclusive scan whose last lane holds the total) instead of ds_bpermute shuffles: ~20 VALU ops and no LDS round
// trips per reduction.  gfx950 is GFX9-family, so row_bcast is available.  With W = 32 the row_bcast:15 step (rows 1 and 3
// only) completes the scan of both halves at once and the total sits in lane 31 / 63; nothing crosses the half boundary.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int v, int ident) { return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROWMASK, 0xf, false); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_d(double v, double ident) {
  int lo = dpp_i<CTRL, ROWMASK>(__double2loint(v), __double2loint(ident));
  int hi = dpp_i<CTRL, ROWMASK>(__double2hiint(v), __double2hiint(ident));
  return __hiloint2double(hi, lo);
}
// index of this lane's group inside the wavefront (0 for W = 64)
template <int W> __device__ __forceinline__ int group_id() { return W == 64 ? 0 : (int)(threadIdx.x >> 5); }
// The lane's index within its wavefront, produced where the optimiser cannot see through it -- and cannot keep it: every phase of
// the sub-step derives its lane index, its group's LDS base and the addresses built from them from a FRESH copy (two VALU
// instructions), so that none of them is live across the other phases.  Left to itself the compiler computes lane, 4 lane,
// S + 4 lane, S + 8 lane, 6 lane ... once at kernel entry, runs out of registers in the solver and parks exactly these in
// scratch: ~60 of the ~200 scratch reloads per sub-step were reloads of values that cost one or two instructions to recompute
// (round 4, from the ISA: slots holding lane, lane << 2, &S + (lane << 2)).
__device__ __forceinline__ int fresh_wave_lane() {
#if defined(__HIP_DEVICE_COMPILE__)
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
#else
  return (int)threadIdx.x;   // (host pass / SIMT emulator; workgroup = one wavefront)
#endif
}
#define FRESH_GROUP(W, SG0)                      \
  const int wl_ = fresh_wave_lane();             \
  const int lane = wl_ & ((W) - 1);              \
  L& S = (SG0)[(W) == 32 ? (wl_ >> 5) : 0]

// Mask of the contact rows (= lanes) of ONE env's group: a 32-bit per-lane value in the two-envs-per-wave layouts (the two envs of a
// wave iterate over their own rows side by side), the wave's 64-bit ballot with one env per wave.
template <int W> struct RowMask { typedef unsigned long long type; };
template <> struct RowMask<32> { typedef unsigned type; };
template <int W> __device__ __forceinline__ typename RowMask<W>::type group_rows(unsigned long long ballot) {
  if constexpr (W == 32) return (unsigned)(ballot >> (32 * group_id<W>()));
  else return ballot;
}
__device__ __forceinline__ int first_row(unsigned m) { return __ffs(m) - 1; }
__device__ __forceinline__ int first_row(unsigned long long m) { return __ffsll(m) - 1; }
// value held by lane `src` (0 <= src < W, uniform) of the caller's group
template <int W>
__device__ __forceinline__ int gbcast_i(int v, int src) {
  if constexpr (W == 64) return __builtin_amdgcn_readlane(v, src);
  else {
    const int a = __builtin_amdgcn_readlane(v, src), b = __builtin_amdgcn_readlane(v, src + 32);
    return group_id<W>() ? b : a;
  }
}
// value held by lane SRC of the caller's 16-lane row: one v_mov_b64_dpp (row_newbcast, gfx90a+)
template <int SRC>
__device__ __forceinline__ double rbc(double v) {
  const long long lv = __double_as_longlong(v);
  return __longlong_as_double(__builtin_amdgcn_update_dpp(lv, lv, 0x150 + SRC, 0xf, 0xf, false));
}
// ---- v_fmac_f64 with a DPP source (round 6).  gfx90a+ encode v_fmac_f64 as VOP2, and a VOP2 instruction can take its first source
// through DPP; for 64-bit operands the one control available is row_newbcast -- exactly the chain layout's "value held by the lane of
// row position e".  acc += bcast_e(v) * c is then ONE instruction instead of v_mov_b64_dpp + v_fma_f64 (the compiler does not form it:
// its DPP combiner skips 64-bit operations, so it is written out; the product is the same single-rounding fma, bit for bit).  A DPP
// instruction must not read a VGPR that a VALU instruction wrote in the two issue slots before it -- the compiler inserts s_nop for its
// own DPP instructions but cannot see into an asm statement, so every statement starts with `s_nop 1`; inside a statement the DPP
// sources are never written (fmac_col: each accumulator is read, through DPP, by its own instruction only -- reads of a row happen
// before its writes, as in any in-place DPP reduction).  Checked on the hardware by scripts/dpp_probe.hip before any kernel used it;
// the host pass / SIMT emulator and LHW_FMAC_DPP=0 builds take the two-instruction form.
#ifndef LHW_FMAC_DPP
#define LHW_FMAC_DPP 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && LHW_FMAC_DPP
#define LHW_FMAC_ASM 1
#else
#define LHW_FMAC_ASM 0
#endif
#define LHW_FD(a) "v_fmac_f64_dpp %[" #a "], %[" #a "], %[c] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define LHW_FA(e) "v_fmac_f64_dpp %[acc], %[x], %[m" #e "] row_newbcast:" #e " row_mask:0xf bank_mask:0xf\n\t"
#define LHW_FB(e) "v_fmac_f64_dpp %[acc], %[c" #e "], %[b" #e "] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
// acc += (v held by the lane of row position SRC) * c
template <int SRC>
__device__ __forceinline__ double fma_rbc(double v, double c, double acc) {
#if LHW_FMAC_ASM
  asm("s_nop 1\n\tv_fmac_f64_dpp %[acc], %[v], %[c] row_newbcast:%[k] row_mask:0xf bank_mask:0xf" : [acc] "+v"(acc) : [v] "v"(v), [c] "v"(c), [k] "n"(SRC));
  return acc;
#else
  return fma(rbc<SRC>(v), c, acc);
#endif
}
// column step K of the chain factorisation: x += bcast_K(x) nf, R[e] += bcast_K(R[e]) nf for e < K
template <int K, int NRR>
__device__ __forceinline__ void fmac_col(double (&R)[NRR], double& x, const double nf) {
#if LHW_FMAC_ASM
  static_assert(K < 12, "fmac_col: column index");
  if constexpr (K == 0) asm("s_nop 1\n\t" LHW_FD(x) : [x] "+v"(x) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 1) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) : [x] "+v"(x), [a0] "+v"(R[0]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 2) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 3) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 4) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 5) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 6) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) LHW_FD(a5) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]), [a5] "+v"(R[5]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 7) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) LHW_FD(a5) LHW_FD(a6) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]), [a5] "+v"(R[5]), [a6] "+v"(R[6]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 8) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) LHW_FD(a5) LHW_FD(a6) LHW_FD(a7) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]), [a5] "+v"(R[5]), [a6] "+v"(R[6]), [a7] "+v"(R[7]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 9) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) LHW_FD(a5) LHW_FD(a6) LHW_FD(a7) LHW_FD(a8) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]), [a5] "+v"(R[5]), [a6] "+v"(R[6]), [a7] "+v"(R[7]), [a8] "+v"(R[8]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 10) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) LHW_FD(a5) LHW_FD(a6) LHW_FD(a7) LHW_FD(a8) LHW_FD(a9) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]), [a5] "+v"(R[5]), [a6] "+v"(R[6]), [a7] "+v"(R[7]), [a8] "+v"(R[8]), [a9] "+v"(R[9]) : [c] "v"(nf), [k] "n"(K));
  else if constexpr (K == 11) asm("s_nop 1\n\t" LHW_FD(x) LHW_FD(a0) LHW_FD(a1) LHW_FD(a2) LHW_FD(a3) LHW_FD(a4) LHW_FD(a5) LHW_FD(a6) LHW_FD(a7) LHW_FD(a8) LHW_FD(a9) LHW_FD(a10) : [x] "+v"(x), [a0] "+v"(R[0]), [a1] "+v"(R[1]), [a2] "+v"(R[2]), [a3] "+v"(R[3]), [a4] "+v"(R[4]), [a5] "+v"(R[5]), [a6] "+v"(R[6]), [a7] "+v"(R[7]), [a8] "+v"(R[8]), [a9] "+v"(R[9]), [a10] "+v"(R[10]) : [c] "v"(nf), [k] "n"(K));
#else
  x = fma(rbc<K>(x), nf, x);
#pragma unroll
  for (int e = 0; e < K; e++) R[e] = fma(rbc<K>(R[e]), nf, R[e]);
#endif
}
// H[e] += bcast_e(x) * c for every row position e: the rank-1 update of a lane's Hessian row by one constraint row whose Jacobian entries
// are held one per lane (x = the entry of this lane's own dof) -- the entries of the other columns come through DPP, not from LDS
#define LHW_FO(e) "v_fmac_f64_dpp %[h" #e "], %[x], %[c] row_newbcast:" #e " row_mask:0xf bank_mask:0xf\n\t"
template <int NRR, int E>
__device__ __forceinline__ void fmac_outer_ref(double (&H)[NRR], double x, double c) {
  if constexpr (E < NRR) {
    H[E] = fma(rbc<E>(x), c, H[E]);
    fmac_outer_ref<NRR, E + 1>(H, x, c);
  }
}
template <int NRR>
__device__ __forceinline__ void fmac_outer(double (&H)[NRR], double x, double c) {
#if LHW_FMAC_ASM
  static_assert(NRR == 11 || NRR == 12, "fmac_outer: row length");
  if constexpr (NRR == 12) asm("s_nop 1\n\t" LHW_FO(0) LHW_FO(1) LHW_FO(2) LHW_FO(3) LHW_FO(4) LHW_FO(5) LHW_FO(6) LHW_FO(7) LHW_FO(8) LHW_FO(9) LHW_FO(10) LHW_FO(11) : [h0] "+v"(H[0]), [h1] "+v"(H[1]), [h2] "+v"(H[2]), [h3] "+v"(H[3]), [h4] "+v"(H[4]), [h5] "+v"(H[5]), [h6] "+v"(H[6]), [h7] "+v"(H[7]), [h8] "+v"(H[8]), [h9] "+v"(H[9]), [h10] "+v"(H[10]), [h11] "+v"(H[11]) : [x] "v"(x), [c] "v"(c));
  else asm("s_nop 1\n\t" LHW_FO(0) LHW_FO(1) LHW_FO(2) LHW_FO(3) LHW_FO(4) LHW_FO(5) LHW_FO(6) LHW_FO(7) LHW_FO(8) LHW_FO(9) LHW_FO(10) : [h0] "+v"(H[0]), [h1] "+v"(H[1]), [h2] "+v"(H[2]), [h3] "+v"(H[3]), [h4] "+v"(H[4]), [h5] "+v"(H[5]), [h6] "+v"(H[6]), [h7] "+v"(H[7]), [h8] "+v"(H[8]), [h9] "+v"(H[9]), [h10] "+v"(H[10]) : [x] "v"(x), [c] "v"(c));
#else
  fmac_outer_ref<NRR, 0>(H, x, c);
#endif
}
template <int NRR, int E>
__device__ __forceinline__ void fmac_mrow_ref(const double (&Mr)[NRR], double x, double& acc) {
  if constexpr (E < NRR) {
    acc = fma(Mr[E], rbc<E>(x), acc);
    fmac_mrow_ref<NRR, E + 1>(Mr, x, acc);
  }
}
// acc += sum_e Mr[e] * bcast_e(x)
template <int NRR>
__device__ __forceinline__ void fmac_mrow(const double (&Mr)[NRR], double x, double& acc) {
#if LHW_FMAC_ASM
  static_assert(NRR == 11 || NRR == 12, "fmac_mrow: row length");
  if constexpr (NRR == 12) asm("s_nop 1\n\t" LHW_FA(0) LHW_FA(1) LHW_FA(2) LHW_FA(3) LHW_FA(4) LHW_FA(5) LHW_FA(6) LHW_FA(7) LHW_FA(8) LHW_FA(9) LHW_FA(10) LHW_FA(11) : [acc] "+v"(acc) : [x] "v"(x), [m0] "v"(Mr[0]), [m1] "v"(Mr[1]), [m2] "v"(Mr[2]), [m3] "v"(Mr[3]), [m4] "v"(Mr[4]), [m5] "v"(Mr[5]), [m6] "v"(Mr[6]), [m7] "v"(Mr[7]), [m8] "v"(Mr[8]), [m9] "v"(Mr[9]), [m10] "v"(Mr[10]), [m11] "v"(Mr[11]));
  else asm("s_nop 1\n\t" LHW_FA(0) LHW_FA(1) LHW_FA(2) LHW_FA(3) LHW_FA(4) LHW_FA(5) LHW_FA(6) LHW_FA(7) LHW_FA(8) LHW_FA(9) LHW_FA(10) : [acc] "+v"(acc) : [x] "v"(x), [m0] "v"(Mr[0]), [m1] "v"(Mr[1]), [m2] "v"(Mr[2]), [m3] "v"(Mr[3]), [m4] "v"(Mr[4]), [m5] "v"(Mr[5]), [m6] "v"(Mr[6]), [m7] "v"(Mr[7]), [m8] "v"(Mr[8]), [m9] "v"(Mr[9]), [m10] "v"(Mr[10]));
#else
  fmac_mrow_ref<NRR, 0>(Mr, x, acc);
#endif
}
// sum_k b[k] * bcast_E(c[k]), k = 0..5
template <int E>
__device__ __forceinline__ double fmac_dot6(const double (&b)[6], const double (&c)[6]) {
#if LHW_FMAC_ASM
  double acc = 0.0;
  asm("s_nop 1\n\t" LHW_FB(0) LHW_FB(1) LHW_FB(2) LHW_FB(3) LHW_FB(4) LHW_FB(5)
      : [acc] "+v"(acc)
      : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]),
        [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [k] "n"(E));
  return acc;
#else
  return b[0] * rbc<E>(c[0]) + b[1] * rbc<E>(c[1]) + b[2] * rbc<E>(c[2]) + b[3] * rbc<E>(c[3]) + b[4] * rbc<E>(c[4]) + b[5] * rbc<E>(c[5]);
#endif
}
// The values of this lane and of lane ^ 16 (v_permlane16_swap, gfx950), as (value of the even row, value of the odd row) of the
// row pair -- the same ordered pair in both lanes, so a reduction over it is bit-identical in the two rows.
__device__ __forceinline__ void xhalf_pair(double v, double& even, double& odd) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  even = __hiloint2double((int)b[0], (int)a[0]);
  odd = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double xhalf_sum(double v) {   // v(this lane) + v(lane ^ 16)
  double e, o;
  xhalf_pair(v, e, o);
  return e + o;
}
// likewise for lane ^ 32 (v_permlane32_swap)
__device__ __forceinline__ void x32_pair(double v, double& lower, double& upper) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  lower = __hiloint2double((int)b[0], (int)a[0]);
  upper = __hiloint2double((int)b[1], (int)a[1]);
}
// Group reductions: row_shr scan inside each 16-lane row (DPP), the row total from lane 15 (row_newbcast), then the rows of the
// group are combined with lane-swap instructions -- no v_readlane / SGPR round trip, ~18 VALU instructions for W = 32.
template <int W>
__device__ __forceinline__ double gsum(double v) {
  v += dpp_d<0x111, 0xf>(v, 0.0); v += dpp_d<0x112, 0xf>(v, 0.0); v += dpp_d<0x114, 0xf>(v, 0.0); v += dpp_d<0x118, 0xf>(v, 0.0);
  v = xhalf_sum(rbc<15>(v));
  if constexpr (W == 64) { double l, u; x32_pair(v, l, u); v = l + u; }
  return v;
}
// two group sums for the price of one scan: the rows of each pair are folded first (even row <- a, odd row <- b), one row scan
// reduces both, and a second lane swap hands every lane both totals
template <int W>
__device__ __forceinline__ void gsum2(double a, double b, double& sa, double& sb) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  // first result: [a(even row), b(even row)] per row pair, second: [a(odd row), b(odd row)]
  double v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  v += dpp_d<0x111, 0xf>(v, 0.0); v += dpp_d<0x112, 0xf>(v, 0.0); v += dpp_d<0x114, 0xf>(v, 0.0); v += dpp_d<0x118, 0xf>(v, 0.0);
  xhalf_pair(rbc<15>(v), sa, sb);
  if constexpr (W == 64) {
    double l, u;
    x32_pair(sa, l, u); sa = l + u;
    x32_pair(sb, l, u); sb = l + u;
  }
}
template <int W>
__device__ __forceinline__ double gmin(double v) {
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  v = fmin(v, dpp_d<0x111, 0xf>(v, inf)); v = fmin(v, dpp_d<0x112, 0xf>(v, inf)); v = fmin(v, dpp_d<0x114, 0xf>(v, inf));
  v = fmin(v, dpp_d<0x118, 0xf>(v, inf));
  double e, o;
  xhalf_pair(rbc<15>(v), e, o);
  v = fmin(e, o);
  if constexpr (W == 64) { double l, u; x32_pair(v, l, u); v = fmin(l, u); }
  return v;
}
// inclusive prefix sum across the group; *total receives the group total
template <int W>
__device__ __forceinline__ int gscan(int v, int* total) {
  v += dpp_i<0x111, 0xf>(v, 0); v += dpp_i<0x112, 0xf>(v, 0); v += dpp_i<0x114, 0xf>(v, 0); v += dpp_i<0x118, 0xf>(v, 0);
  v += dpp_i<0x142, 0xa>(v, 0);
  if constexpr (W == 64) v += dpp_i<0x143, 0xc>(v, 0);
  *total = gbcast_i<W>(v, W - 1);
  return v;
}
// does the predicate hold for any (active) lane of the caller's group
template <int W>
__device__ __forceinline__ bool gany(bool pred) {
  const unsigned long long b = __ballot(pred);
  if constexpr (W == 64) return b != 0;
  else return ((b >> (32 * group_id<W>())) & 0xffffffffull) != 0;
}
// ---- fp64 reciprocal, division, square root without the IEEE expansions (round 6).  The compiler expands `a / b` into v_div_scale x2,
// v_rcp_f64, eight FMAs, v_div_fmas, v_div_fixup (13 instructions) and sqrt() into v_rsq_f64 plus a scaled Goldschmidt iteration with class
// tests (17); the operands of the sub-step are lengths, masses, pivots and regularisers -- normal numbers far from the ends of the
// exponent range -- so the seed instruction plus two Newton steps (error about one ulp, not correctly rounded) is enough: the float64
// oracle keeps IEEE division and the parity tolerance (1e-12 per five-control-step segment) has four decades of room.
// LHW_FASTDIV=0 compiles the IEEE forms back in (A/B, scripts/build_variant.sh).
#ifndef LHW_FASTDIV
#define LHW_FASTDIV 1
#endif
__device__ __forceinline__ double rcp_f64(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(fma(-d, x, 1.0), x, x);
  x = fma(fma(-d, x, 1.0), x, x);
  return x;
}
__device__ __forceinline__ double qdiv(double a, double b) {
#if LHW_FASTDIV
  return a * rcp_f64(b);
#else
  return a / b;
#endif
}
// 1 / sqrt(x) for x > 0 (v_rsq_f64 + two Newton steps); x = 0 gives +inf like the instruction
__device__ __forceinline__ double rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(fma(-hx * y, y, 0.5), y, y);
  y = fma(fma(-hx * y, y, 0.5), y, y);
  return y;
}
// sqrt(x) for x >= 0: x * rsqrt(x) with the residual of the product folded back in; 0 for x = 0
__device__ __forceinline__ double qsqrt(double x) {
#if LHW_FASTDIV
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return x == 0.0 ? 0.0 : g;
#else
  return sqrt(x);
#endif
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ double normalize3(double* a) {
  double n = qsqrt(dot3(a, a));
  if (n < HMINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return n; }
  double inv = qdiv(1.0, n);
  a[0] *= inv; a[1] *= inv; a[2] *= inv;
  return n;
}
__device__ __forceinline__ void normalize4(double* q) {
  double n = qsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < HMINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double inv = qdiv(1.0, n);
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
__device__ __forceinline__ void mul_quat(double* r, const double* a, const double* b) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
__device__ __forceinline__ void quat2mat(double* R, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02); R[3] = 2 * (q12 + q03);
  R[5] = 2 * (q23 - q01); R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
__device__ __forceinline__ void mat_vec(double* r, const double* R, const double* v) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
         z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void matT_vec(double* r, const double* R, const double* v) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
         z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void mat_mul(double* C, const double* A, const double* B) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void axis_angle_quat(double* q, const double* ax, double ang) {
  double s, c;
  sincos(0.5 * ang, &s, &c);
  q[0] = c; q[1] = ax[0] * s; q[2] = ax[1] * s; q[3] = ax[2] * s;
}
// 10-number com-based inertia times spatial motion vector [rot; lin]
__device__ __forceinline__ void inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

#define NCH ((L::NV_ - 6) / 2)   // dofs per chain; NR: dof lanes per 16-lane row of the chain layout (see chain_solve)
#define NR (6 + NCH)
// ------------------------------------------------------------------------------------------------ row products
// y_lane = sum_k row[k] * v[k] with the row in registers and v broadcast from LDS
template <class L>
__device__ __forceinline__ double row_dot(const double (&row)[NV], const double* v) {
  double a0 = 0, a1 = 0;
#pragma unroll
  for (int k = 0; k < NV; k += 2) {
    a0 += row[k] * v[k];
    a1 += row[k + 1] * v[k + 1];
  }
  return a0 + a1;
}
// ------------------------------------------------------------------------------------------------ chain-structured SPD solve
// Both robots are a free root (dofs 0..5) carrying two serial chains of NCH dofs each (the legs): M, M + h D and the Newton
// Hessian M + J^T D J are block-structured [root | chain A | chain B] with no A-B block, unless a contact couples the two legs
// (then the dense solver above is used for that sub-step).  The chain solver lays one HALF of the env on each 16-lane DPP row:
//   row position p = 0..5   root dof p          (held by BOTH rows of the env: "copy A" and "copy B")
//   row position p = 6..6+NCH-1   dof p-6 of the row's chain
// so that "the value held by the lane of dof e of my half" is ONE instruction (v_mov_b64_dpp row_newbcast:e) instead of the
// four v_readlane + two v_cndmask of a 32-lane-group broadcast, and the two chains are eliminated in lockstep: a reverse
// (leaf-to-root) L^T D L factorisation -- the order in which this structure has no fill-in, as in MuJoCo's mj_factorM --
// takes 6 + NCH column steps of at most 5 + NCH updates instead of NV steps of NV.  The root-root block is split between the
// two copies (their Schur complements add up: one v_permlane16_swap exchange per value after the chain columns).
// Lane p holds row p of its half's matrix [[Krr_h, C_h^T], [C_h, T_h]] as R[0 .. NR) (full symmetric row; R[p] itself is
// never read), its diagonal entry in dg, and element p of the right-hand side.
// column step K of the factorisation with the right-hand side carried along (x <- L^-T x on the fly)
template <class L, int K>
__device__ __forceinline__ void chain_col(double (&R)[NR], double& dg, double& x, int p) {
  const double inv = rcp_f64(rbc<K>(dg));
  const double nf = (p < K) ? -(R[K] * inv) : 0.0;   // -L[K][p]
  dg = fma(nf, R[K], dg);
  fmac_col<K>(R, x, nf);
}
template <class L, int K, int KEND>
__device__ __forceinline__ void chain_cols(double (&R)[NR], double& dg, double& x, int p) {
  if constexpr (K >= KEND) {
    chain_col<L, K>(R, dg, x, p);
    chain_cols<L, K - 1, KEND>(R, dg, x, p);
  }
}
template <class L, int E>
__device__ __forceinline__ void chain_fwd(const double (&Lr)[NR], double& x) {
  if constexpr (E < NR) {
    x = fma_rbc<E>(x, Lr[E], x);
    chain_fwd<L, E + 1>(Lr, x);
  }
}
// acc += sum_e Mrow[e] * (x held by the lane of row position e)
template <class L, int E>
__device__ __forceinline__ void chain_mrow(const double (&Mr)[NR], double x, double& acc) {
  static_assert(E == 0, "chain_mrow: whole rows only");
  fmac_mrow<NR>(Mr, x, acc);
}
// x <- K^-1 x.  R, dg: this lane's row / diagonal (copy B of the root: zero; destroyed); rootb: this lane is a root dof's copy B.
// Root elements of x must be identical in the two copies on entry, and are on return.
template <class L>
__device__ __forceinline__ double chain_solve(double (&R)[NR], double dg, double x, int p, bool rootb) {
  if (rootb) x = 0.0;
  chain_cols<L, NR - 1, 6>(R, dg, x, p);
  if (p < 6) {   // Schur complements of the two chains add up on the root block
#pragma unroll
    for (int e = 0; e < 6; e++) R[e] = xhalf_sum(R[e]);
    dg = xhalf_sum(dg);
    x = xhalf_sum(x);
  }
  GROUP_SYNC(L::W_);
  chain_cols<L, 5, 0>(R, dg, x, p);
  const double myinv = rcp_f64(dg);   // (every lane's dg is frozen once its own column has been eliminated)
  x *= myinv;
  double Lr[NR];
#pragma unroll
  for (int e = 0; e < NR; e++) Lr[e] = (e < p) ? -(R[e] * myinv) : 0.0;   // row p of the unit factor, negated
  chain_fwd<L, 0>(Lr, x);
  return x;
}

// ------------------------------------------------------------------------------------------------ forward dynamics phases
// mj_kinematics.  The bodies that carry a joint (the root and the 2 NCH chain bodies) sit in the chain layout (row position 5 =
// root, 6.. = chain bodies, parents first): each lane forms the affine map of its body relative to the previous one -- the fixed
// offset composed, on the host, through any welded bodies in between, times the joint's rotation / translation -- and a
// Hillis-Steele scan of map compositions along the row (row_shr 1, 2, 4 [, 8]; identity where a lane has no source) gives every
// world frame in log2 depth: three rounds of 24 DPP moves + a 3x3 product instead of seven dependent tree levels with an LDS
// hand-off each.  Bodies without a joint (welded upper body ...) are then one product with the frame of the body they move with.
template <int CTRL>
__device__ __forceinline__ double dpp_row_id(double v, double ident) { return dpp_d<CTRL, 0xf>(v, ident); }
template <int CTRL>
__device__ __forceinline__ void affine_scan_round(double (&R)[9], double (&pw)[3]) {
  double Ra[9], pa[3];
#pragma unroll
  for (int k = 0; k < 9; k++) Ra[k] = dpp_row_id<CTRL>(R[k], (k % 4 == 0) ? 1.0 : 0.0);
#pragma unroll
  for (int k = 0; k < 3; k++) pa[k] = dpp_row_id<CTRL>(pw[k], 0.0);
  double Rn[9], t[3];
  mat_mul(Rn, Ra, R);
  mat_vec(t, Ra, pw);
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = Rn[k];
#pragma unroll
  for (int k = 0; k < 3; k++) pw[k] = pa[k] + t[k];
}
// (fwd_kinematics / fwd_com forced inline: with fwd_collision inlined the compiler had made THESE two calls in the stepping-task
// kernel instead -- generic-pointer loads of the model tables, callee-saved registers through scratch, every sub-step; round 5, same
// box, jvrc_step @ 4096: 1.287 -> 1.320 M env-steps/s.  The walking kernels inline them either way.)
template <bool STEPT, class L>
__device__ __forceinline__ void fwd_kinematics(HModelRef m, L& S, int lane) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pw[3] = {0, 0, 0}, jax[3] = {0, 0, 0}, jps[3] = {0, 0, 0};
  int kb = -1, jt = -1, jid = 0;
  if (lane < 32) {
    const int* ki = m.kin_i + KIS * lane;
    kb = ki[0]; jt = ki[1]; jid = ki[3];
    // (the lane's frame record is fetched beside its index record, not behind it: one table round trip instead of two; lanes without
    // a body hold a zero record)
    double kd[KDS];
#pragma unroll
    for (int k = 0; k < KDS; k++) kd[k] = m.kin_d[KDS * lane + k];
    if (kb >= 0) {
      const int qa = ki[2];
      for (int k = 0; k < 3; k++) { jax[k] = kd[12 + k]; jps[k] = kd[15 + k]; }
      if (jt == JT_FREE) {
        double q[4] = {S.qpos[qa + 3], S.qpos[qa + 4], S.qpos[qa + 5], S.qpos[qa + 6]};
        normalize4(q);
        if (lane < 16) for (int k = 0; k < 4; k++) { S.qpos[qa + 3 + k] = q[k]; if constexpr (STEPT) S.rootquat[k] = q[k]; }
        quat2mat(R, q);
        for (int k = 0; k < 3; k++) pw[k] = S.qpos[qa + k];
      } else {
        const double qj = S.qpos[qa] - kd[18];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = kd[k];
        if (jt == JT_HINGE) {
          double sn, cs;
          sincos(qj, &sn, &cs);
          const double ax = jax[0], ay = jax[1], az = jax[2], t = 1.0 - cs;
          const double Rj[9] = {cs + t * ax * ax, t * ax * ay - sn * az, t * ax * az + sn * ay,
                                t * ax * ay + sn * az, cs + t * ay * ay, t * ay * az - sn * ax,
                                t * ax * az - sn * ay, t * ay * az + sn * ax, cs + t * az * az};
          double v0[3], v1[3];
          mat_vec(v0, kd, jps);            // the anchor is fixed in both frames: p = p0 + R0 jpos - R jpos
          mat_mul(R, kd, Rj);
          mat_vec(v1, R, jps);
          for (int k = 0; k < 3; k++) pw[k] = kd[9 + k] + v0[k] - v1[k];
        } else {                            // slide: translation along the axis
          double v0[3];
          mat_vec(v0, kd, jax);
          for (int k = 0; k < 3; k++) pw[k] = kd[9 + k] + v0[k] * qj;
        }
      }
    }
  }
  affine_scan_round<0x111>(R, pw);
  affine_scan_round<0x112>(R, pw);
  affine_scan_round<0x114>(R, pw);
  if constexpr (NR > 13) affine_scan_round<0x118>(R, pw);
  if (lane == 0) {
    S.xpos[0] = S.xpos[1] = S.xpos[2] = 0;
    for (int k = 0; k < 9; k++) S.U[U_XMAT + k] = (k % 4 == 0) ? 1.0 : 0.0;
    S.U[U_XIPOS + 0] = S.U[U_XIPOS + 1] = S.U[U_XIPOS + 2] = 0;
  }
  if (kb >= 0 && (lane < 16 || (lane & 15) >= 6)) {     // (the root's frame: one of its two copies writes)
    double anchor[3], waxis[3];
    mat_vec(anchor, R, jps);
    mat_vec(waxis, R, jax);
    for (int k = 0; k < 3; k++) {
      S.xpos[3 * kb + k] = pw[k];
      S.U[U_XANCHOR + 3 * jid + k] = jt == JT_FREE ? pw[k] : pw[k] + anchor[k];
      S.U[U_XAXIS + 3 * jid + k] = jt == JT_FREE ? jax[k] : waxis[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) S.U[U_XMAT + 9 * kb + k] = R[k];
  }
  SYNC();
  const int b = lane;
  const bool valid = b >= 1 && b < m.nbody;
  if (valid) {
    const int ow = m.fix_i[b];             // the jointed body this one is welded to (-1: it has a joint itself)
    double Rf[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Rf[k] = m.fix_d[12 * b + k];   // (fetched beside the index, not behind it)
    const double pf[3] = {m.fix_d[12 * b + 9], m.fix_d[12 * b + 10], m.fix_d[12 * b + 11]};
    if (ow >= 0) {
      double Ro[9], Rb[9], t[3];
#pragma unroll
      for (int k = 0; k < 9; k++) Ro[k] = S.U[U_XMAT + 9 * ow + k];
      mat_mul(Rb, Ro, Rf);
      mat_vec(t, Ro, pf);
      for (int k = 0; k < 3; k++) S.xpos[3 * b + k] = S.xpos[3 * ow + k] + t[k];
#pragma unroll
      for (int k = 0; k < 9; k++) S.U[U_XMAT + 9 * b + k] = Rb[k];
    }
  }
  SYNC();
  // com of each body and its rotated inertia T = Ri diag(I) Ri^T (completed with the com offset in fwd_com): all bodies at once
  if (valid) {
    double Rb[9], Ri[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Rb[k] = S.U[U_XMAT + 9 * b + k];
    const double bip[3] = {prm_ipos(m, S, b, 0), prm_ipos(m, S, b, 1), prm_ipos(m, S, b, 2)};
    mat_vec(t, Rb, bip);
    for (int k = 0; k < 3; k++) S.U[U_XIPOS + 3 * b + k] = S.xpos[3 * b + k] + t[k];
    double ri[9];
#pragma unroll
    for (int k = 0; k < 9; k++) ri[k] = m.body_d[BDS * b + BD_RINERT + k];
    mat_mul(Ri, Rb, ri);
    const double I0 = m.body_d[BDS * b + BD_INERTIA], I1 = m.body_d[BDS * b + BD_INERTIA + 1], I2 = m.body_d[BDS * b + BD_INERTIA + 2];
    double* ci = &S.U[U_CINERT + 10 * b];
    ci[0] = Ri[0] * I0 * Ri[0] + Ri[1] * I1 * Ri[1] + Ri[2] * I2 * Ri[2];
    ci[1] = Ri[3] * I0 * Ri[3] + Ri[4] * I1 * Ri[4] + Ri[5] * I2 * Ri[5];
    ci[2] = Ri[6] * I0 * Ri[6] + Ri[7] * I1 * Ri[7] + Ri[8] * I2 * Ri[8];
    ci[3] = Ri[0] * I0 * Ri[3] + Ri[1] * I1 * Ri[4] + Ri[2] * I2 * Ri[5];
    ci[4] = Ri[0] * I0 * Ri[6] + Ri[1] * I1 * Ri[7] + Ri[2] * I2 * Ri[8];
    ci[5] = Ri[3] * I0 * Ri[6] + Ri[4] * I1 * Ri[7] + Ri[5] * I2 * Ri[8];
  }
  if (lane < 9) S.rootmat[lane] = S.U[U_XMAT + 9 * 1 + lane];
  if constexpr (STEPT) if (lane < 3) {  // tracked points (foot force sites): body origin + R * local offset, lane = point
    const int tb = lane == 0 ? m.track_body[0] : (lane == 1 ? m.track_body[1] : m.track_body[2]);
    const double off[3] = {lane == 0 ? m.track_off[0] : (lane == 1 ? m.track_off[3] : m.track_off[6]),
                           lane == 0 ? m.track_off[1] : (lane == 1 ? m.track_off[4] : m.track_off[7]),
                           lane == 0 ? m.track_off[2] : (lane == 1 ? m.track_off[5] : m.track_off[8])};
    double w[3];
    mat_vec(w, &S.U[U_XMAT + 9 * tb], off);
    for (int k = 0; k < 3; k++) S.spos[3 * lane + k] = S.xpos[3 * tb + k] + w[k];
  }
}

// subtree com of the (single) dynamic tree rooted at body 1, cinert, cdof  (mj_comPos)
template <class L>
__device__ __forceinline__ void fwd_com(HModelRef m, HParamsRef p, L& S, int lane) {
  double ms = 0, mx = 0, my = 0, mz = 0;
  if (lane >= 1 && lane < m.nbody && m.body_i[BIS * (lane) + BI_ROOT] == 1) {
    ms = prm_mass(m, S, lane);
    mx = ms * S.U[U_XIPOS + 3 * lane]; my = ms * S.U[U_XIPOS + 3 * lane + 1]; mz = ms * S.U[U_XIPOS + 3 * lane + 2];
  }
  ms = gsum<L::W_>(ms); mx = gsum<L::W_>(mx); my = gsum<L::W_>(my); mz = gsum<L::W_>(mz);
  const double com[3] = {qdiv(mx, ms), qdiv(my, ms), qdiv(mz, ms)};
  if (lane == 0) { S.com[0] = com[0]; S.com[1] = com[1]; S.com[2] = com[2]; }
  if (lane >= 1 && lane < m.nbody) {
    const int b = lane;
    const double mass = prm_mass(m, S, b);
    // static bodies (their own root) use their own com as reference; they never enter M or the bias force
    const bool dyn = m.body_i[BIS * (b) + BI_ROOT] == 1;
    double dif[3];
    for (int k = 0; k < 3; k++) dif[k] = dyn ? S.U[U_XIPOS + 3 * b + k] - com[k] : 0.0;
    double* ci = &S.U[U_CINERT + 10 * b];
    ci[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    ci[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    ci[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    ci[3] -= mass * dif[0] * dif[1];
    ci[4] -= mass * dif[0] * dif[2];
    ci[5] -= mass * dif[1] * dif[2];
    ci[6] = mass * dif[0]; ci[7] = mass * dif[1]; ci[8] = mass * dif[2]; ci[9] = mass;
  }
  if (lane < NV) {
    const int d = lane, j = m.dof_i[DIS * (d) + DI_JNT], b = m.dof_i[DIS * (d) + DI_BODY], kind = m.dof_i[DIS * (d) + DI_KIND];
    const int t = kind <= 1 ? JT_FREE : (kind == 2 ? JT_SLIDE : JT_HINGE), k = m.dof_i[DIS * (d) + DI_KIDX];
    double off[3], ax[3], c[6];
    for (int a = 0; a < 3; a++) off[a] = com[a] - S.U[U_XANCHOR + 3 * j + a];
    if (t == JT_FREE && k < 3) {
      for (int a = 0; a < 6; a++) c[a] = 0;
      c[3 + k] = 1;
    } else if (t == JT_SLIDE) {
      c[0] = c[1] = c[2] = 0;
      for (int a = 0; a < 3; a++) c[3 + a] = S.U[U_XAXIS + 3 * j + a];
    } else {
      if (t == JT_FREE) { ax[0] = S.U[U_XMAT + 9 * b + (k - 3)]; ax[1] = S.U[U_XMAT + 9 * b + 3 + (k - 3)]; ax[2] = S.U[U_XMAT + 9 * b + 6 + (k - 3)]; }
      else for (int a = 0; a < 3; a++) ax[a] = S.U[U_XAXIS + 3 * j + a];
      c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2];
      cross3(c + 3, ax, off);
    }
    for (int a = 0; a < 6; a++) S.U[U_CDOF + 6 * d + a] = c[a];
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------------ tree dynamics, chain layout
// mj_comPos (cdof), mj_comVel, mj_rne (bias force), mj_crb (joint-space inertia) in the half-env-per-DPP-row layout of the
// chain solver: row position 0..5 = the root's dofs, 6.. = the dofs of the row's chain, parents first.  Everything that the
// reference computes by walking the tree is a scan along the row here:
//   cvel(body of dof p)  = inclusive prefix sum of cdof_e qvel_e             (row_shr scan)
//   cacc                 = -g + inclusive prefix sum of cdof_dot_e qvel_e    (row_shr scan)
//   subtree force / composite inertia = suffix sum of the per-body force / cinert     (row_shl scan)
// A lane also does the per-body work of "its" bodies: a chain lane the body its dof moves; the root's 12 lanes (6 per row)
// share the bodies that move with the root (pelvis, welded upper body), so the suffix sum at position 0, added over the two
// rows, is the total of the whole tree = the root's subtree.  M[p][e] = cdof_e . (crb_p cdof_p) for e an ancestor dof of p comes
// from row broadcasts of cdof; the mirror half of the (symmetric) row is fetched through a 12 x 12 transposition buffer in LDS.
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) { return dpp_d<CTRL, 0xf>(v, 0.0); }
template <int N>
__device__ __forceinline__ void row_prefix(double (&v)[N]) {   // inclusive prefix sums along the 16-lane row
#pragma unroll
  for (int a = 0; a < N; a++) { v[a] += dpp_row<0x111>(v[a]); v[a] += dpp_row<0x112>(v[a]); v[a] += dpp_row<0x114>(v[a]); v[a] += dpp_row<0x118>(v[a]); }
}
template <int N>
__device__ __forceinline__ void row_suffix(double (&v)[N]) {   // inclusive suffix sums along the 16-lane row
#pragma unroll
  for (int a = 0; a < N; a++) { v[a] += dpp_row<0x101>(v[a]); v[a] += dpp_row<0x102>(v[a]); v[a] += dpp_row<0x104>(v[a]); v[a] += dpp_row<0x108>(v[a]); }
}
template <class L, int E>
__device__ __forceinline__ void chain_mlow(const double (&buf)[6], const double (&cd)[6], double (&t)[NR]) {
  if constexpr (E < NR) {
    t[E] = fmac_dot6<E>(buf, cd);
    chain_mlow<L, E + 1>(buf, cd, t);
  }
}
#define U_TB (L::X_)   // transposition buffer [2][NR][NR] (stage B region; dead before and after)
template <class L>
__device__ __forceinline__ void chain_dynamics(HModelRef m, HParamsRef p, L& S, const int lane, const int dof, const bool prim,
                                               double (&Mrow)[NR], double& mdiag, double& marm, double& bias, double& qapp) {
  const int cp = lane & 15, hh = (lane >> 4) & 1, dd = dof >= 0 ? dof : 0;
  const bool isdof = dof >= 0, rootb = isdof && !prim;
  // ---- cdof of this lane's dof (written by fwd_com, lane = dof)
  double cd[6] = {0, 0, 0, 0, 0, 0}, qv = 0;
  if (isdof) {
#pragma unroll
    for (int a = 0; a < 6; a++) cd[a] = S.U[U_CDOF + 6 * dd + a];
    qv = S.qvel[dd];
  }
  // ---- mj_xfrcAccumulate: Cartesian force / torque applied at the com of the perturbed bodies -> joint space
  // (H1 layouts: the wrenches sit in LDS with the env's other per-env parameters.  JVRC tasks with `perturbation` configured read them from
  // the env's HBM record -- p.xfrc_base + the env index parked in LDS -- because their two-envs-per-wave layout has no LDS left for twelve
  // more doubles per env, and nothing of it may ride through the sub-step in registers: the branch is a uniform scalar test when it is off)
  qapp = 0;
  if constexpr (L::PRM_) {
    if (p.env_params && isdof) {
      for (int k = 0; k < p.n_pbody; k++) {
        const int pb = p.pbody[k];
        if (!(((unsigned)m.body_i[BIS * pb + BI_DOFMASK] >> dd) & 1u)) continue;
        double off[3], t[3];
        for (int a = 0; a < 3; a++) off[a] = S.U[U_XIPOS + 3 * pb + a] - S.com[a];
        cross3(t, cd, off);
        for (int a = 0; a < 3; a++) qapp += (cd[3 + a] + t[a]) * S.xfrc[6 * k + a] + cd[a] * S.xfrc[6 * k + 3 + a];
      }
    }
  } else if (p.perturb_interval > 0) {
    if (isdof) {
      gtab_d xr = p.xfrc_base + (size_t)S.env_id * PRM_D + P_XFRC;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        if (k >= p.n_pbody) break;
        const int pb = p.pbody[k];
        if (!(((unsigned)m.body_i[BIS * pb + BI_DOFMASK] >> dd) & 1u)) continue;
        double off[3], t[3];
        for (int a = 0; a < 3; a++) off[a] = S.U[U_XIPOS + 3 * pb + a] - S.com[a];
        cross3(t, cd, off);
        for (int a = 0; a < 3; a++) qapp += (cd[3 + a] + t[a]) * xr[6 * k + a] + cd[a] * xr[6 * k + 3 + a];
      }
    }
  }
  // ---- velocities (mj_comVel): cvel of the body behind each dof, cdof_dot
  double cv[6];
#pragma unroll
  for (int a = 0; a < 6; a++) cv[a] = cd[a] * qv;
  double vp[6];   // velocity seen by the dof when its cdof_dot is formed
#pragma unroll
  for (int a = 0; a < 6; a++) vp[a] = -cv[a];
  row_prefix(cv);
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const double t2 = rbc<2>(cv[a]);          // free joint, rotations: the translations of the same joint only
    vp[a] = cp >= 6 ? vp[a] + cv[a] : t2;   // chain dofs: everything above them
  }
  double ca[6];
  {
    double a3[3], b3[3], c3[3];
    cross3(a3, vp, cd); cross3(b3, vp, cd + 3); cross3(c3, vp + 3, cd);
    const bool zero = cp < 3 || !isdof;       // translations of the free joint: cdof_dot = 0
    for (int a = 0; a < 3; a++) { ca[a] = zero ? 0.0 : a3[a] * qv; ca[3 + a] = zero ? 0.0 : (b3[a] + c3[a]) * qv; }
  }
  row_prefix(ca);
#pragma unroll
  for (int a = 0; a < 3; a++) ca[3 + a] -= m.gravity[a];
  // ---- per-body bias force f = I a + v x* (I v) and inertia of the bodies this lane owns (mj_rne, no acceleration)
  double bv[6], ba[6];   // velocity / bias acceleration of those bodies: own (chain) or the root's (root lanes)
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const double rv = rbc<5>(cv[a]), ra = rbc<5>(ca[a]);
    bv[a] = cp >= 6 ? cv[a] : rv;
    ba[a] = cp >= 6 ? ca[a] : ra;
  }
  double fs[6] = {0, 0, 0, 0, 0, 0}, ci[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = 0; q < m.max_owned; q++) {
    const int b = lane < 32 ? m.own_tab[(lane)*m.max_owned + q] : -1;
    if (b >= 0) {
      double I[10], t[6], t2[6], a3[3], b3[3], c3[3];
#pragma unroll
      for (int a = 0; a < 10; a++) { I[a] = S.U[U_CINERT + 10 * b + a]; ci[a] += I[a]; }
      inert_vec(t, I, ba);
      inert_vec(t2, I, bv);
      cross3(a3, bv, t2); cross3(b3, bv + 3, t2 + 3); cross3(c3, bv, t2 + 3);
      for (int a = 0; a < 3; a++) { fs[a] += a3[a] + b3[a] + t[a]; fs[3 + a] += c3[a] + t[3 + a]; }
      // the task reads the velocity of three bodies after the step (root, feet)
      for (int k = 0; k < 3; k++)
        if (b == m.track_body[k]) for (int a = 0; a < 6; a++) S.svel[6 * k + a] = bv[a];
    }
  }
  // ---- subtree sums along the row; the root's subtree is the whole tree
  row_suffix(fs);
  row_suffix(ci);
#pragma unroll
  for (int a = 0; a < 6; a++) { const double tot = xhalf_sum(rbc<0>(fs[a])); fs[a] = cp >= 6 ? fs[a] : tot; }
#pragma unroll
  for (int a = 0; a < 10; a++) { const double tot = xhalf_sum(rbc<0>(ci[a])); ci[a] = cp >= 6 ? ci[a] : tot; }
  bias = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) bias += cd[a] * fs[a];
  // ---- joint-space inertia (mj_crb): row of this lane's dof over [root | own chain]
  double buf[6], tl[NR];
  inert_vec(buf, ci, cd);
  chain_mlow<L, 0>(buf, cd, tl);
  mdiag = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) mdiag += buf[a] * cd[a];
  marm = (prim ? m.dof_d[DDS * dd + DD_ARMATURE] : 0.0);
  mdiag += marm;
  SYNC();
  if (isdof) {
    double* tb = &S.U[U_TB + (hh * NR + cp) * NR];
#pragma unroll
    for (int e = 0; e < NR; e++) tb[e] = tl[e];
  }
  SYNC();
#pragma unroll
  for (int e = 0; e < NR; e++) {
    const double tr = S.U[U_TB + (hh * NR + e) * NR + (cp < NR ? cp : 0)];   // M[e][p], computed by the lane of dof e
    double v = e <= cp ? tl[e] : tr;
    if (e < 6) v = rootb ? 0.0 : v;   // the root-root block lives in copy A of the root rows
    Mrow[e] = v;
  }
  if (rootb) mdiag = 0.0;
  SYNC();
}

// ---- collision (engine_collision_primitive.c restated).  Two passes over the same narrow phase: pass 0 counts the
// contacts of each candidate pair (lane = pair), a wave scan gives every pair its slot range in pair order, pass 1
// recomputes and writes straight into the LDS contact arrays (no per-lane contact records in scratch).  Exceptions: box-box pairs
// keep their contacts between the passes (BoxRec), and plane-box pairs -- the feet on the floor, nearly all contacts of a walking
// robot -- are evaluated once, one lane per box corner (fwd_collision).  A bounding-sphere broad phase runs in front of both passes.

// LDS working set of the merge of duplicate contacts (fwd_collision, many-contact path), laid over the contact-record cache of
// newton_big, which is idle during the collision stage.  Per raw contact: a float32 FILTER WORD v = x + 1.7 y + 64 * merge class (x, y:
// position; copies agree in it to a float32 ulp, the other corners of a foot are centimetres away -- also the ones that share x or
// y with it -- and other classes tens of units; a chance coincidence only costs the exact check), walked sixteen candidates per trip; y and the distance in float32 for the few candidates that pass (which are then compared
// exactly, in float64, from the workspace); the merge key and the pair index.  Once the groups are known, the three float arrays are
// reused for rank / multiplicity / ground-reaction counts.
template <class L>
struct MergeLds {
  float *d32, *v32, *y32;
  unsigned short *k16, *f16, *p16;
  int *rank, *cnt, *crl;
  __device__ __forceinline__ explicit MergeLds(L& S) {
    static_assert(L::NCK_ * 13 * 8 >= 3 * NCR * 4 + 2 * NCR * 2 && L::NCK_ * 3 * 4 >= NCR * 2, "merge scratch does not fit the record cache");
    v32 = reinterpret_cast<float*>(S.bk_rec); d32 = v32 + NCR; y32 = d32 + NCR;   // (v32 first: read as float4, bk_rec is 16-byte aligned)
    k16 = reinterpret_cast<unsigned short*>(y32 + NCR); f16 = k16 + NCR;
    p16 = reinterpret_cast<unsigned short*>(S.bk_i);
    rank = reinterpret_cast<int*>(v32); cnt = reinterpret_cast<int*>(d32); crl = reinterpret_cast<int*>(y32);
  }
};

template <class L>
struct ConSink {
  L* S;
  int base, n, write, g1, g2, pair;
  int key = 0;            // merge class * 2 + orientation of this lane's pair (many-contact path: the merge scan's key)
  gws_d gd = nullptr;     // non-NULL: the contacts go to the raw region of the HBM workspace (AR_* / ARI_* layout), capacity NCR
  gws_i gi = nullptr;
  __device__ __forceinline__ void emit(double dist, const double* pos, const double* nrm, const double* tan) {
    if (write) {
      const int c = base + n;
      if (c < (gd ? NCR : NC)) {
        L& Z = *S;
        double f[9];
        for (int a = 0; a < 3; a++) { f[a] = nrm[a]; f[3 + a] = tan[a]; }
        if (gd) { gd[AR_DIST + c] = dist; for (int a = 0; a < 3; a++) gd[AR_POS + 3 * c + a] = pos[a]; }
        else { Z.U[U_CDIST + c] = dist; for (int a = 0; a < 3; a++) Z.con_pos[3 * c + a] = pos[a]; }
        // mju_makeFrame
        normalize3(f);
        if (dot3(f + 3, f + 3) < 0.25) {
          f[3] = f[4] = f[5] = 0;
          if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
        }
        const double tt = dot3(f, f + 3);
        for (int a = 0; a < 3; a++) f[3 + a] -= tt * f[a];
        normalize3(f + 3);
        cross3(f + 6, f, f + 3);
        if (gd) {
          for (int a = 0; a < 9; a++) gd[AR_FRAME + 9 * c + a] = f[a];
          gi[ARI_G1 + c] = g1; gi[ARI_G2 + c] = g2; gi[ARI_PAIR + c] = pair;
          if constexpr (L::STEP_) {   // what the merge scan filters on, in LDS (see MergeLds): nothing of it is read back from the workspace
            MergeLds<L> ml(Z);
            ml.d32[c] = (float)dist; ml.v32[c] = (float)(pos[0] + 1.7 * pos[1] + 64.0 * (double)(key >> 1)); ml.y32[c] = (float)pos[1];
            ml.k16[c] = (unsigned short)key; ml.p16[c] = (unsigned short)pair;
          }
        } else {
          for (int a = 0; a < 9; a++) Z.U[U_CFRAME + 9 * c + a] = f[a];
          Z.con_g1[c] = g1; Z.con_g2[c] = g2; Z.con_pair[c] = pair;
        }
      }
    }
    n++;
  }
};

template <class L>
__device__ __forceinline__ void col_plane_sphere(ConSink<L>& k, const double* p1, const double* R1, const double* p2, double r,
                                                 double margin, const double* tan) {
  double n[3] = {R1[2], R1[5], R1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const double dist = dot3(dif, n) - r;
  if (dist > margin) return;
  double pos[3];
  for (int a = 0; a < 3; a++) pos[a] = p2[a] - n[a] * (r + 0.5 * dist);
  k.emit(dist, pos, n, tan);
}
template <class L>
__device__ __forceinline__ void col_sphere_sphere(ConSink<L>& k, const double* p1, double r1, const double* p2, double r2, double margin) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const double cd = qsqrt(dot3(dif, dif)), dist = cd - r1 - r2;
  if (dist > margin) return;
  if (cd < HMINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; } else { dif[0] = qdiv(dif[0], cd); dif[1] = qdiv(dif[1], cd); dif[2] = qdiv(dif[2], cd); }
  double pos[3];
  const double zero[3] = {0, 0, 0};
  for (int a = 0; a < 3; a++) pos[a] = p1[a] + dif[a] * (r1 + 0.5 * dist);
  k.emit(dist, pos, dif, zero);
}
// Box-box: separating-axis test over the 15 candidate axes (3 + 3 face normals, 9 edge cross products; ties go to the
// faces of geom2, an edge axis must beat the best face by 1e-6), then either a face contact -- the incident face polygon
// and the reference face rectangle are intersected (incident vertices inside the rectangle, rectangle corners inside the
// incident face, edge x side crossings) and the vertices at or below the reference face within `margin` become
// contacts, at most 4, deepest first -- or a single edge-edge contact at the
// mid-point of the closest points of the two edges.  This is the classical SAT + clipping construction (as in ODE's
// dBoxBox), NOT a restatement of MuJoCo's mjc_BoxBox, whose source could not be consulted: the two agree for face-face
// resting contacts (what the stair terrain produces) and may differ in contact count / placement in edge cases
// (DESIGN.md section 6).  The CPU checker used by the tests implements the same statements in the same order.
struct BoxRec {  // contacts of one box-box pair, kept between the counting and the writing pass of fwd_collision
  double dist[4], pos[12], n[3];
  int cnt;
  __device__ __forceinline__ void emit(double d_, const double* pos_, const double* n_, const double*) {
    if (cnt < 4) {
      dist[cnt] = d_;
      for (int a = 0; a < 3; a++) { pos[3 * cnt + a] = pos_[a]; n[a] = n_[a]; }
      cnt++;
    }
  }
};
__device__ __forceinline__ double sel3(int i, double a0, double a1, double a2) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }

// Every array below is indexed by compile-time constants only (loops fully unrolled, run-time choices through sel3), so the
// whole narrow phase lives in registers: the first version kept the clipped polygon in scratch memory and cost ~58 k
// cycles per sub-step for a wave with feet on boxes.
template <class L>
__device__ __noinline__ void col_box_box(BoxRec& k, HModelRef m, const L& S, int q, int g1, int g2, double margin) {
  double p1[3], p2[3], R1[9], R2[9], s1[3], s2[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { p1[a] = S.U[U_GPOS + 3 * g1 + a]; p2[a] = S.U[U_GPOS + 3 * g2 + a]; s1[a] = m.pair_d[PDS * q + PD_SIZE1 + a]; s2[a] = m.pair_d[PDS * q + PD_SIZE2 + a]; }
#pragma unroll
  for (int a = 0; a < 9; a++) { R1[a] = S.U[U_GMAT + 9 * g1 + a]; R2[a] = S.U[U_GMAT + 9 * g2 + a]; }
  double A[3][3], B[3][3];
  const double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int c = 0; c < 3; c++) { A[i][c] = R1[3 * c + i]; B[i][c] = R2[3 * c + i]; }
  double R[3][3], AR[3][3], ta[3], tb[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    ta[i] = dot3(d, A[i]); tb[i] = dot3(d, B[i]);
#pragma unroll
    for (int j = 0; j < 3; j++) { R[i][j] = dot3(A[i], B[j]); AR[i][j] = fabs(R[i][j]) + 1e-12; }
  }
  double best = -1e300;
  int code = -1;
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double s = fabs(ta[i]) - (s1[i] + s2[0] * AR[i][0] + s2[1] * AR[i][1] + s2[2] * AR[i][2]);
    sep = sep || s > margin;
    if (s > best) { best = s; code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double s = fabs(tb[j]) - (s2[j] + s1[0] * AR[0][j] + s1[1] * AR[1][j] + s1[2] * AR[2][j]);
    sep = sep || s > margin;
    if (s > best - 1e-9) { if (s > best) best = s; code = 3 + j; }
  }
  double ebest = -1e300;
  int ecode = -1;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double len2 = 1.0 - R[i][j] * R[i][j];
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double tl = ta[i2] * R[i1][j] - ta[i1] * R[i2][j];
      const double ra = s1[i1] * AR[i2][j] + s1[i2] * AR[i1][j], rb = s2[j1] * AR[i][j2] + s2[j2] * AR[i][j1];
      if (len2 >= 1e-12) {
        const double s = (fabs(tl) - ra - rb) * rsqrt_f64(len2);
        sep = sep || s > margin;
        if (s > ebest) { ebest = s; ecode = 6 + 3 * i + j; }
      }
    }
  if (sep) return;
  if (ecode >= 0 && ebest > best + 1e-6) {
    const int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
    const double Ai[3] = {sel3(i, A[0][0], A[1][0], A[2][0]), sel3(i, A[0][1], A[1][1], A[2][1]), sel3(i, A[0][2], A[1][2], A[2][2])};
    const double Bj[3] = {sel3(j, B[0][0], B[1][0], B[2][0]), sel3(j, B[0][1], B[1][1], B[2][1]), sel3(j, B[0][2], B[1][2], B[2][2])};
    double n[3];
    cross3(n, Ai, Bj);
    normalize3(n);
    if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    double pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (c != i) { const double sg = dot3(n, A[c]) > 0 ? 1.0 : -1.0; for (int a = 0; a < 3; a++) pa[a] += sg * s1[c] * A[c][a]; }
      if (c != j) { const double sg = dot3(n, B[c]) > 0 ? -1.0 : 1.0; for (int a = 0; a < 3; a++) pb[a] += sg * s2[c] * B[c][a]; }
    }
    const double w[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    const double bq = sel3(i, sel3(j, R[0][0], R[0][1], R[0][2]), sel3(j, R[1][0], R[1][1], R[1][2]), sel3(j, R[2][0], R[2][1], R[2][2]));
    const double dd = dot3(Ai, w), ee = dot3(Bj, w), den = 1.0 - bq * bq;
    const double rden = rcp_f64(den), al = (bq * ee - dd) * rden, be = (ee - bq * dd) * rden;
    double pos[3];
    for (int a = 0; a < 3; a++) pos[a] = 0.5 * ((pa[a] + al * Ai[a]) + (pb[a] + be * Bj[a]));
    k.emit(ebest, pos, n, nullptr);
    return;
  }
  const bool refB = code >= 3;
  const int ax = refB ? code - 3 : code;
  double Ar[3][3], Ac[3][3], pr[3], pc[3], sr[3], sc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    pr[i] = refB ? p2[i] : p1[i]; pc[i] = refB ? p1[i] : p2[i]; sr[i] = refB ? s2[i] : s1[i]; sc[i] = refB ? s1[i] : s2[i];
#pragma unroll
    for (int c = 0; c < 3; c++) { Ar[i][c] = refB ? B[i][c] : A[i][c]; Ac[i][c] = refB ? A[i][c] : B[i][c]; }
  }
  double n[3] = {sel3(ax, Ar[0][0], Ar[1][0], Ar[2][0]), sel3(ax, Ar[0][1], Ar[1][1], Ar[2][1]), sel3(ax, Ar[0][2], Ar[1][2], Ar[2][2])};
  if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  const double nr[3] = {refB ? -n[0] : n[0], refB ? -n[1] : n[1], refB ? -n[2] : n[2]};
  const double sr_ax = sel3(ax, sr[0], sr[1], sr[2]);
  const double fc[3] = {pr[0] + nr[0] * sr_ax, pr[1] + nr[1] * sr_ax, pr[2] + nr[2] * sr_ax};
  int kc = 0;
  double bestdot = -1;
#pragma unroll
  for (int c = 0; c < 3; c++) { const double v = fabs(dot3(nr, Ac[c])); if (v > bestdot) { bestdot = v; kc = c; } }
  const int ku = (kc + 1) % 3, kv = (kc + 2) % 3, ru = (ax + 1) % 3, rv = (ax + 2) % 3;
  double Akc[3], Aku[3], Akv[3], u[3], v[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    Akc[a] = sel3(kc, Ac[0][a], Ac[1][a], Ac[2][a]); Aku[a] = sel3(ku, Ac[0][a], Ac[1][a], Ac[2][a]); Akv[a] = sel3(kv, Ac[0][a], Ac[1][a], Ac[2][a]);
    u[a] = sel3(ru, Ar[0][a], Ar[1][a], Ar[2][a]); v[a] = sel3(rv, Ar[0][a], Ar[1][a], Ar[2][a]);
  }
  const double sgn = dot3(nr, Akc) > 0 ? -1.0 : 1.0;
  const double sckc = sel3(kc, sc[0], sc[1], sc[2]), scku = sel3(ku, sc[0], sc[1], sc[2]), sckv = sel3(kv, sc[0], sc[1], sc[2]);
  const double ha = sel3(ru, sr[0], sr[1], sr[2]), hb = sel3(rv, sr[0], sr[1], sr[2]);
  double P[4][3], px[4], py[4], pd[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const double su = (q == 0 || q == 3) ? 1.0 : -1.0, sv = (q < 2) ? 1.0 : -1.0;
    double w0[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      P[q][a] = pc[a] + sgn * sckc * Akc[a] + su * scku * Aku[a] + sv * sckv * Akv[a];
      w0[a] = P[q][a] - fc[a];
    }
    px[q] = dot3(w0, u); py[q] = dot3(w0, v); pd[q] = dot3(w0, nr);
  }
  // candidates (i) incident vertices inside the rectangle, (ii) rectangle corners inside the incident parallelogram,
  // (iii) incident edge x rectangle side crossings; the four deepest are kept by bubbling through a sorted 4-slot list
  double bd[4] = {1e300, 1e300, 1e300, 1e300}, bp[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  auto push = [&](double nd, double n0, double n1, double n2) {
    if (nd <= margin) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
        if (nd < bd[kk]) {
          double t = bd[kk]; bd[kk] = nd; nd = t;
          t = bp[kk][0]; bp[kk][0] = n0; n0 = t;
          t = bp[kk][1]; bp[kk][1] = n1; n1 = t;
          t = bp[kk][2]; bp[kk][2] = n2; n2 = t;
        }
    }
  };
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (fabs(px[q]) <= ha && fabs(py[q]) <= hb) push(pd[q], P[q][0], P[q][1], P[q][2]);
  {
    const double e1x = px[1] - px[0], e1y = py[1] - py[0], e2x = px[3] - px[0], e2y = py[3] - py[0];
    const double det = e1x * e2y - e1y * e2x;
    if (fabs(det) > 1e-14) {
      const double rdet = rcp_f64(det);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const double cx = (c == 0 || c == 3) ? ha : -ha, cy = (c < 2) ? hb : -hb;
        const double al = ((cx - px[0]) * e2y - (cy - py[0]) * e2x) * rdet, be = (e1x * (cy - py[0]) - e1y * (cx - px[0])) * rdet;
        if (al >= 0 && al <= 1 && be >= 0 && be <= 1) {
          const double dep = pd[0] + al * (pd[1] - pd[0]) + be * (pd[3] - pd[0]);
          push(dep, fc[0] + cx * u[0] + cy * v[0] + dep * nr[0], fc[1] + cx * u[1] + cy * v[1] + dep * nr[1],
               fc[2] + cx * u[2] + cy * v[2] + dep * nr[2]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int q1 = (q + 1) & 3;
    const double dx = px[q1] - px[q], dy = py[q1] - py[q], dd = pd[q1] - pd[q];
    const double rdx = rcp_f64(dx), rdy = rcp_f64(dy);   // (one reciprocal per edge direction instead of two divisions per side; dx == 0 / dy == 0 edges are masked by `ok`)
#pragma unroll
    for (int sd = 0; sd < 4; sd++) {
      double tt = 0, other = 0, lim = 0;
      bool ok;
      if (sd < 2) {
        ok = dx != 0;
        tt = ((sd == 0 ? ha : -ha) - px[q]) * rdx; other = py[q] + tt * dy; lim = hb;
      } else {
        ok = dy != 0;
        tt = ((sd == 2 ? hb : -hb) - py[q]) * rdy; other = px[q] + tt * dx; lim = ha;
      }
      if (ok && tt > 0 && tt < 1 && fabs(other) < lim)
        push(pd[q] + tt * dd, P[q][0] + tt * (P[q1][0] - P[q][0]), P[q][1] + tt * (P[q1][1] - P[q][1]), P[q][2] + tt * (P[q1][2] - P[q][2]));
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (bd[q] < 1e299) {
      const double pos[3] = {bp[q][0] - nr[0] * bd[q] * 0.5, bp[q][1] - nr[1] * bd[q] * 0.5, bp[q][2] - nr[2] * bd[q] * 0.5};
      k.emit(bd[q], pos, n, nullptr);
    }
}

// Sphere-box (mjc_SphereBox): the centre, in the box frame, clamped to the box; outside, the contact is along
// (centre - clamped point); inside, the sphere leaves through the nearest face.  Normal from the sphere to the box, position
// midway between the two surfaces.  Returns false beyond the margin.  (same operation order as sphereBox in the CPU checker, mjc_oracle.c)
__device__ __forceinline__ bool sphere_box_raw(const double* p1, double r, const double* p2, const double* R2, const double* size, double margin,
                                               double* dist_out, double* pos, double* nrm) {
  const double d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double ctr[3], cl[3], v[3], nl[3], pl[3];
  for (int a = 0; a < 3; a++) ctr[a] = R2[a] * d[0] + R2[3 + a] * d[1] + R2[6 + a] * d[2];   // R2^T d
  for (int a = 0; a < 3; a++) { cl[a] = ctr[a] > size[a] ? size[a] : (ctr[a] < -size[a] ? -size[a] : ctr[a]); v[a] = ctr[a] - cl[a]; }
  const double dist = qsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  double pen;
  if (dist > HMINVAL) {
    pen = dist - r;
    if (pen > margin) return false;
    for (int a = 0; a < 3; a++) { nl[a] = qdiv(v[a], dist); pl[a] = cl[a] + nl[a] * (0.5 * pen); }
  } else {
    int kf = 0;
    double depth = size[0] - fabs(ctr[0]);
    for (int a = 1; a < 3; a++) if (size[a] - fabs(ctr[a]) < depth) { depth = size[a] - fabs(ctr[a]); kf = a; }
    const double ck = kf == 0 ? ctr[0] : (kf == 1 ? ctr[1] : ctr[2]), sk = kf == 0 ? size[0] : (kf == 1 ? size[1] : size[2]);
    const double sg = ck >= 0 ? 1.0 : -1.0;
    pen = -depth - r;
    if (pen > margin) return false;
    for (int a = 0; a < 3; a++) { nl[a] = a == kf ? sg : 0.0; pl[a] = a == kf ? sg * sk : ctr[a]; }
    for (int a = 0; a < 3; a++) pl[a] += nl[a] * (0.5 * pen);
  }
  double nw[3], pw[3];
  mat_vec(nw, R2, nl);
  mat_vec(pw, R2, pl);
  *dist_out = pen;
  for (int a = 0; a < 3; a++) { nrm[a] = -nw[a]; pos[a] = p2[a] + pw[a]; }
  return true;
}
// half the slope of the squared distance from c0 + t al (box frame) to the box: nondecreasing in t
__device__ __forceinline__ double seg_box_slope(const double* c0, const double* al, const double* size, double t) {
  double g = 0;
  for (int a = 0; a < 3; a++) {
    const double x = c0[a] + t * al[a];
    if (x > size[a]) g += al[a] * (x - size[a]);
    else if (x < -size[a]) g += al[a] * (x + size[a]);
  }
  return g;
}

// Sphere-box and capsule-box pairs.  Kept out of collide_pair and called only under the model-wide flag m.has_primbox (a
// scalar branch): models without such pairs -- the stand-ins -- must not pay for this code (inside collide_pair's dispatch
// chain the compiler speculated parts of it for every pair: +8 % VALU instructions per sub-step).
template <class L>
__device__ void collide_primbox(ConSink<L>& k, HModelRef m, const L& S, int q, int g1, int g2, double margin) {
  const double zero[3] = {0, 0, 0};
  const int t1 = m.pair_i[PIS * q + PI_TYPE1];
  double p1[3], p2[3], R1[9], R2[9], s1[3], s2[3];
  for (int a = 0; a < 3; a++) { p1[a] = S.U[U_GPOS + 3 * g1 + a]; p2[a] = S.U[U_GPOS + 3 * g2 + a]; s1[a] = m.pair_d[PDS * q + PD_SIZE1 + a]; s2[a] = m.pair_d[PDS * q + PD_SIZE2 + a]; }
  for (int a = 0; a < 9; a++) { R1[a] = S.U[U_GMAT + 9 * g1 + a]; R2[a] = S.U[U_GMAT + 9 * g2 + a]; }
  if (t1 == G_SPHERE) {
    double dist, pos[3], nrm[3];
    if (sphere_box_raw(p1, s1[0], p2, R2, s2, margin, &dist, pos, nrm)) k.emit(dist, pos, nrm, zero);
  } else {
    // same construction as capsuleBox in the CPU checker, mjc_oracle.c (not MuJoCo's mjc_CapsuleBox case analysis): both ends if both touch; else the interval
    // [tlo, thi] of segment points nearest to the box (two bisections on the monotone slope), plus a touching end
    const double ax[3] = {R1[2], R1[5], R1[8]}, h = s1[1], d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const double TOL = 1e-9, FLAT = 1e-12;
    double c0[3], al[3];
    for (int a = 0; a < 3; a++) {
      c0[a] = R2[a] * d[0] + R2[3 + a] * d[1] + R2[6 + a] * d[2];
      al[a] = R2[a] * ax[0] + R2[3 + a] * ax[1] + R2[6 + a] * ax[2];
    }
    double q[3], dm, pm[3], nm[3], dp, pp[3], np_[3];
    for (int a = 0; a < 3; a++) q[a] = p1[a] - ax[a] * h;
    const bool gm = sphere_box_raw(q, s1[0], p2, R2, s2, margin, &dm, pm, nm);
    for (int a = 0; a < 3; a++) q[a] = p1[a] + ax[a] * h;
    const bool gp = sphere_box_raw(q, s1[0], p2, R2, s2, margin, &dp, pp, np_);
    if (gm && gp) { k.emit(dm, pm, nm, zero); k.emit(dp, pp, np_, zero); }
    else {
      const double sm = seg_box_slope(c0, al, s2, -h), sp = seg_box_slope(c0, al, s2, h);
      double tlo, thi;
      if (sm >= -FLAT) tlo = -h;
      else if (sp < -FLAT) tlo = h;
      else {
        double lo = -h, hi = h;
        for (int it = 0; it < 60; it++) { const double mid = 0.5 * (lo + hi); if (seg_box_slope(c0, al, s2, mid) >= -FLAT) hi = mid; else lo = mid; }
        tlo = hi;
      }
      if (sp <= FLAT) thi = h;
      else if (sm > FLAT) thi = -h;
      else {
        double lo = -h, hi = h;
        for (int it = 0; it < 60; it++) { const double mid = 0.5 * (lo + hi); if (seg_box_slope(c0, al, s2, mid) <= FLAT) lo = mid; else hi = mid; }
        thi = lo;
      }
      const int n0 = k.n;
      double ds, ps[3], ns[3];
      for (int a = 0; a < 3; a++) q[a] = p1[a] + ax[a] * tlo;
      if (sphere_box_raw(q, s1[0], p2, R2, s2, margin, &ds, ps, ns)) k.emit(ds, ps, ns, zero);
      if (thi > tlo + TOL) {
        for (int a = 0; a < 3; a++) q[a] = p1[a] + ax[a] * thi;
        if (sphere_box_raw(q, s1[0], p2, R2, s2, margin, &ds, ps, ns)) k.emit(ds, ps, ns, zero);
      } else thi = tlo;
      if (k.n - n0 < 2 && gm && tlo > -h + TOL) k.emit(dm, pm, nm, zero);
      if (k.n - n0 < 2 && gp && thi < h - TOL) k.emit(dp, pp, np_, zero);
    }
  }
}

// Plane-cylinder and sphere-cylinder pairs (round 6): the two cylinder narrow phases MuJoCo resolves analytically (mjc_PlaneCylinder,
// mjc_SphereCylinder [MJ-recall]; the other cylinder pairs go through its general convex collider and are refused at create).  Kept out of
// collide_pair and called only under the model-wide flag m.has_cyl, like collide_primbox: models without cylinders do not pay for it.
// (same operation order as planeCylinder / sphereCylinder in the CPU checker, mjc_oracle.c)
template <class L>
__device__ void collide_cyl(ConSink<L>& k, HModelRef m, const L& S, int q, int g1, int g2, double margin) {
  const double zero[3] = {0, 0, 0};
  const int t1 = m.pair_i[PIS * q + PI_TYPE1];
  double p1[3], p2[3], R1[9], R2[9];
  for (int a = 0; a < 3; a++) { p1[a] = S.U[U_GPOS + 3 * g1 + a]; p2[a] = S.U[U_GPOS + 3 * g2 + a]; }
  for (int a = 0; a < 9; a++) { R1[a] = S.U[U_GMAT + 9 * g1 + a]; R2[a] = S.U[U_GMAT + 9 * g2 + a]; }
  const double r1 = m.pair_d[PDS * q + PD_SIZE1], rad = m.pair_d[PDS * q + PD_SIZE2], hl = m.pair_d[PDS * q + PD_SIZE2 + 1];
  if (m.pair_i[PIS * q + PI_TYPE2] == G_ELLIPSOID) {   // plane-ellipsoid (mjc_PlaneConvex on a smooth geom): the support point towards the plane
    const double n[3] = {R1[2], R1[5], R1[8]}, sz[3] = {rad, hl, m.pair_d[PDS * q + PD_SIZE2 + 2]};
    double dl[3], v[3], sup[3], pos[3];
    for (int a = 0; a < 3; a++) dl[a] = -(R2[a] * n[0] + R2[3 + a] * n[1] + R2[6 + a] * n[2]);
    for (int a = 0; a < 3; a++) v[a] = sz[a] * dl[a];
    const double len = sqrt(dot3(v, v));
    for (int a = 0; a < 3; a++) v[a] = sz[a] * v[a] / len;
    for (int a = 0; a < 3; a++) sup[a] = p2[a] + R2[3 * a] * v[0] + R2[3 * a + 1] * v[1] + R2[3 * a + 2] * v[2];
    const double dist = (sup[0] - p1[0]) * n[0] + (sup[1] - p1[1]) * n[1] + (sup[2] - p1[2]) * n[2];
    if (dist > margin) return;
    for (int a = 0; a < 3; a++) pos[a] = sup[a] - n[a] * dist * 0.5;
    k.emit(dist, pos, n, zero);
  } else if (t1 == G_PLANE) {
    const double n[3] = {R1[2], R1[5], R1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double axis[3] = {R2[2], R2[5], R2[8]}, vec[3];
    double prjaxis = dot3(n, axis);
    if (prjaxis > 0) { for (int a = 0; a < 3; a++) axis[a] = -axis[a]; prjaxis = -prjaxis; }
    const double dist0 = dot3(dif, n);
    for (int a = 0; a < 3; a++) vec[a] = axis[a] * prjaxis - n[a];
    const double len_sqr = dot3(vec, vec);
    if (len_sqr >= HMINVAL * HMINVAL) { const double scl = rad / sqrt(len_sqr); for (int a = 0; a < 3; a++) vec[a] *= scl; }
    else { vec[0] = R2[0] * rad; vec[1] = R2[3] * rad; vec[2] = R2[6] * rad; }
    const double prjvec = dot3(vec, n);
    for (int a = 0; a < 3; a++) axis[a] *= hl;
    prjaxis *= hl;
    double pos[3];
    if (!(dist0 + prjaxis + prjvec <= margin)) return;
    {
      const double d = dist0 + prjaxis + prjvec;
      for (int a = 0; a < 3; a++) pos[a] = p2[a] + vec[a] + axis[a] - n[a] * d * 0.5;
      k.emit(d, pos, n, zero);
    }
    if (dist0 - prjaxis + prjvec <= margin) {
      const double d = dist0 - prjaxis + prjvec;
      for (int a = 0; a < 3; a++) pos[a] = p2[a] + vec[a] - axis[a] - n[a] * d * 0.5;
      k.emit(d, pos, n, zero);
    }
    const double prjvec1 = -prjvec * 0.5;
    if (dist0 + prjaxis + prjvec1 <= margin) {
      double vec1[3];
      cross3(vec1, vec, axis);
      normalize3(vec1);
      const double sc = rad * sqrt(3.0) / 2;
      for (int a = 0; a < 3; a++) vec1[a] *= sc;
      const double d = dist0 + prjaxis + prjvec1;
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        for (int a = 0; a < 3; a++) pos[a] = p2[a] + sgn * vec1[a] + axis[a] - 0.5 * vec[a] - n[a] * d * 0.5;
        k.emit(d, pos, n, zero);
      }
    }
  } else {   // sphere (geom1) against the cylinder: beside the lateral surface, over a cap, or off the rim
    const double axis[3] = {R2[2], R2[5], R2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const double x = dot3(vec, axis);
    double a3[3];
    for (int a = 0; a < 3; a++) a3[a] = vec[a] - axis[a] * x;
    const double a_sqr = dot3(a3, a3);
    if (x >= -hl && x <= hl) {
      const double q3[3] = {p2[0] + axis[0] * x, p2[1] + axis[1] * x, p2[2] + axis[2] * x};
      col_sphere_sphere(k, p1, r1, q3, rad, margin);
    } else {
      const double sg = x > 0 ? 1.0 : -1.0;
      if (a_sqr <= rad * rad) {
        const double dist = fabs(x) - hl - r1;
        if (dist > margin) return;
        double nrm[3], pos[3];
        for (int a = 0; a < 3; a++) { nrm[a] = -sg * axis[a]; pos[a] = p1[a] - sg * axis[a] * (r1 + 0.5 * dist); }
        k.emit(dist, pos, nrm, zero);
      } else {
        const double sc = rad / sqrt(a_sqr);
        double q3[3];
        for (int a = 0; a < 3; a++) q3[a] = p2[a] + axis[a] * sg * hl + a3[a] * sc;
        col_sphere_sphere(k, p1, r1, q3, 0.0, margin);
      }
    }
  }
}

template <class L>
__device__ void collide_pair(ConSink<L>& k, HModelRef m, const L& S, int q, int g1, int g2, double margin) {
  const double zero[3] = {0, 0, 0};
  const int t1 = m.pair_i[PIS * q + PI_TYPE1], t2 = m.pair_i[PIS * q + PI_TYPE2];
  double p1[3], p2[3], R1[9], R2[9], s1[3], s2[3];
  for (int a = 0; a < 3; a++) { p1[a] = S.U[U_GPOS + 3 * g1 + a]; p2[a] = S.U[U_GPOS + 3 * g2 + a]; s1[a] = m.pair_d[PDS * q + PD_SIZE1 + a]; s2[a] = m.pair_d[PDS * q + PD_SIZE2 + a]; }
  for (int a = 0; a < 9; a++) { R1[a] = S.U[U_GMAT + 9 * g1 + a]; R2[a] = S.U[U_GMAT + 9 * g2 + a]; }
  if (t1 == G_PLANE && t2 == G_SPHERE) col_plane_sphere(k, p1, R1, p2, s2[0], margin, zero);
  else if (t1 == G_PLANE && t2 == G_CAPSULE) {
    double ax[3] = {R2[2], R2[5], R2[8]}, e[3];
    for (int s = 1; s >= -1; s -= 2) {
      for (int a = 0; a < 3; a++) e[a] = p2[a] + s * ax[a] * s2[1];
      col_plane_sphere(k, p1, R1, e, s2[0], margin, ax);  // tangent aligned with the capsule axis
    }
  } else if (t1 == G_PLANE && t2 == G_BOX) {
    double nn[3] = {R1[2], R1[5], R1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const double dist = dot3(dif, nn);
    const int n0 = k.n;
    for (int i = 0; i < 8 && k.n - n0 < 4; i++) {
      double v[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])}, corner[3], pos[3];
      mat_vec(corner, R2, v);
      const double ld = dot3(nn, corner);
      if (dist + ld > margin || ld > 0) continue;
      for (int a = 0; a < 3; a++) pos[a] = corner[a] + p2[a] - nn[a] * (dist + ld) * 0.5;
      k.emit(dist + ld, pos, nn, zero);
    }
  } else if (t1 == G_SPHERE && t2 == G_SPHERE) col_sphere_sphere(k, p1, s1[0], p2, s2[0], margin);
  else if (t1 == G_SPHERE && t2 == G_CAPSULE) {
    double ax[3] = {R2[2], R2[5], R2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const double x = fmin(s2[1], fmax(-s2[1], dot3(ax, vec)));
    double q[3] = {p2[0] + ax[0] * x, p2[1] + ax[1] * x, p2[2] + ax[2] * x};
    col_sphere_sphere(k, p1, s1[0], q, s2[0], margin);
  } else if (t1 == G_CAPSULE && t2 == G_CAPSULE) {
    double a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]}, dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
    const double det = ma * mc - mb * mb;
    if (fabs(det) >= HMINVAL) {
      double x1 = qdiv(mc * u - mb * v, det), x2 = qdiv(ma * v - mb * u, det);
      if (x1 > s1[1]) { x1 = s1[1]; x2 = qdiv(v - mb * s1[1], mc); }
      else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = qdiv(v + mb * s1[1], mc); }
      if (x2 > s2[1]) { x2 = s2[1]; x1 = fmin(s1[1], fmax(-s1[1], qdiv(u - mb * s2[1], ma))); }
      else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = fmin(s1[1], fmax(-s1[1], qdiv(u + mb * s2[1], ma))); }
      double q1[3], q2[3];
      for (int a = 0; a < 3; a++) { q1[a] = p1[a] + a1[a] * x1; q2[a] = p2[a] + a2[a] * x2; }
      col_sphere_sphere(k, q1, s1[0], q2, s2[0], margin);
    } else {
      const int n0 = k.n;
      double q1[3], q2[3], x;
      for (int s = -1; s <= 1 && k.n - n0 < 2; s += 2) {
        for (int a = 0; a < 3; a++) q1[a] = p1[a] + s * a1[a] * s1[1];
        double vv[3] = {q1[0] - p2[0], q1[1] - p2[1], q1[2] - p2[2]};
        x = dot3(a2, vv);
        if (x >= -s2[1] && x <= s2[1]) {
          for (int a = 0; a < 3; a++) q2[a] = p2[a] + a2[a] * x;
          col_sphere_sphere(k, q1, s1[0], q2, s2[0], margin);
        }
      }
      for (int s = -1; s <= 1 && k.n - n0 < 2; s += 2) {
        for (int a = 0; a < 3; a++) q2[a] = p2[a] + s * a2[a] * s2[1];
        double vv[3] = {q2[0] - p1[0], q2[1] - p1[1], q2[2] - p1[2]};
        x = dot3(a1, vv);
        if (x >= -s1[1] && x <= s1[1]) {
          for (int a = 0; a < 3; a++) q1[a] = p1[a] + a1[a] * x;
          col_sphere_sphere(k, q1, s1[0], q2, s2[0], margin);
        }
      }
    }
  }
}

// (forced inline: left to itself the compiler inlines the walking kernels' instance and makes the stepping task's -- 9.6 k instructions --
// a call, whose prologue and epilogue save and restore 42 callee-saved VGPRs through scratch in every sub-step, besides what the
// caller spills around the call; round 5, same box, jvrc_step @ 4096: rollout 1.228 -> 1.169 s.  col_box_box inlined as well: 1.249 s.)
template <bool BOXBOX, class L>
__device__ __forceinline__ void fwd_collision(HModelRef m, HParamsRef p, L& S, int lane, gtab_d ter, gws_d bd, gws_i bi) {
  FINE_BEGIN(3);
  if (lane < m.ngeom) {
    const int g = lane, b = m.geom_i[GIS * (g) + GI_BODY];
    double gp[3] = {m.geom_d[GDS * (g) + GD_POS], m.geom_d[GDS * (g) + GD_POS + 1], m.geom_d[GDS * (g) + GD_POS + 2]}, t[3];
    double Rl[9], Rb[9], R[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rl[k] = m.geom_d[GDS * g + GD_RLOC + k]; Rb[k] = S.U[U_XMAT + 9 * b + k]; }
    mat_vec(t, Rb, gp);
    mat_mul(R, Rb, Rl);
    double wp[3] = {S.xpos[3 * b] + t[0], S.xpos[3 * b + 1] + t[1], S.xpos[3 * b + 2] + t[2]};
    if (BOXBOX && ter) {  // stepping task: the 20 boxes sit under the target steps, the floor is lowered in FORWARD mode
      const int kb = g - p.box_geom0;
      if (kb >= 0 && kb < p.nbox) {
        const double* sq = ter + T_SEQ + 6 * kb;
        const double c = sq[4], sn = sq[5];
        wp[0] = sq[0]; wp[1] = sq[1]; wp[2] = sq[2] - m.geom_d[GDS * g + GD_SIZE + 2];
        R[0] = c; R[1] = -sn; R[2] = 0; R[3] = sn; R[4] = c; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
      } else if (g == p.floor_geom) wp[2] += ter[T_FLOOR];
    }
    for (int k = 0; k < 3; k++) S.U[U_GPOS + 3 * g + k] = wp[k];
    for (int k = 0; k < 9; k++) S.U[U_GMAT + 9 * g + k] = R[k];
  }
  SYNC();
  FINE_MARK(3, 0);
  int g1 = 0, g2 = 0, ty1 = -1, ty2 = -1, pkey = 0;
  double margin = 0;
  bool have = lane < m.npair;
  if (have) {
    g1 = m.pair_i[PIS * lane]; g2 = m.pair_i[PIS * lane + 1]; ty1 = m.pair_i[PIS * lane + PI_TYPE1]; ty2 = m.pair_i[PIS * lane + PI_TYPE2];
    if constexpr (BOXBOX && L::W_ == 64) pkey = m.pair_i[PIS * lane + 6] * 2 + m.pair_i[PIS * lane + 7];
    margin = m.pair_d[PDS * lane];
    // (the terrain boxes collide in every walk mode, as the reference leaves them -- coplanar with the floor outside FORWARD mode,
    // tasks/stepping_task.py:320-334: an env with more than NC contacts takes the many-contact path below)
  }
  // Broad phase (the bounding-sphere filter of mj_collideGeoms): a pair whose bounding spheres -- or, against a plane, whose sphere and
  // the plane -- are further apart than the margin cannot yield a contact and skips both passes.  Conservative (1e-9 of slack), so the
  // contacts are the same; what it buys is that a TYPE of narrow phase none of whose pairs is in range is not executed at all -- on
  // an upright robot that is every type but the feet's (the lanes of a group walk the types one after the other).
  if (have) {
    const double rb1 = m.pair_d[PDS * lane + PD_RBOUND1], rb2 = m.pair_d[PDS * lane + PD_RBOUND2];
    double dif[3];
    for (int a = 0; a < 3; a++) dif[a] = S.U[U_GPOS + 3 * g2 + a] - S.U[U_GPOS + 3 * g1 + a];
    bool far;
    if (ty1 == G_PLANE) {
      const double nn[3] = {S.U[U_GMAT + 9 * g1 + 2], S.U[U_GMAT + 9 * g1 + 5], S.U[U_GMAT + 9 * g1 + 8]};
      far = dot3(dif, nn) - rb2 > margin + 1e-9;
    } else {
      const double lim = rb1 + rb2 + margin + 1e-9;
      const double dd = dot3(dif, dif);
      far = lim > 0 && dd > lim * lim;
      // Spheres and capsules are segments with a radius (a sphere: of length 0): the distance of two segments is at least their
      // separation along the line of centres, |d| - h1 |a1 . d^| - h2 |a2 . d^|, whatever points the narrow phase ends up with
      // (they lie on the segments).  The bounding spheres of the leg capsules -- thigh beside thigh, shank under the hip -- always
      // overlap, so without this the capsule-capsule and sphere-capsule narrow phases (divisions, square roots) ran in both passes
      // of every sub-step of an upright robot; the separation test is three dot products.  Conservative like the test above
      // (same slack), so the contacts are the same.  (multiplied through by |d|: far <=> A > 0 and A^2 > R^2 |d|^2, no root)
      if (!far && (ty1 == G_SPHERE || ty1 == G_CAPSULE) && (ty2 == G_SPHERE || ty2 == G_CAPSULE)) {
        const double h1 = ty1 == G_CAPSULE ? m.pair_d[PDS * lane + PD_SIZE1 + 1] : 0.0, h2 = ty2 == G_CAPSULE ? m.pair_d[PDS * lane + PD_SIZE2 + 1] : 0.0;
        const double a1d = S.U[U_GMAT + 9 * g1 + 2] * dif[0] + S.U[U_GMAT + 9 * g1 + 5] * dif[1] + S.U[U_GMAT + 9 * g1 + 8] * dif[2];
        const double a2d = S.U[U_GMAT + 9 * g2 + 2] * dif[0] + S.U[U_GMAT + 9 * g2 + 5] * dif[1] + S.U[U_GMAT + 9 * g2 + 8] * dif[2];
        const double A = dd - h1 * fabs(a1d) - h2 * fabs(a2d);
        const double R = m.pair_d[PDS * lane + PD_SIZE1] + m.pair_d[PDS * lane + PD_SIZE2] + margin + 1e-9;
        far = A > 0 && (R <= 0 || A * A > R * R * dd);
      }
    }
    have = !far;
  }
  ConSink<L> k{&S, 0, 0, 0, g1, g2, lane};
  // box-box pairs (stepping-task kernels only) run the SAT + clipping once, in the counting pass, and replay the recorded
  // contacts in the writing pass; every other pair type is cheap enough to be evaluated twice
  BoxRec br;
  br.cnt = 0;
  bool boxpair = false;
  if constexpr (BOXBOX) boxpair = have && ty1 == G_BOX && ty2 == G_BOX;
  bool primbox = false;
  if (m.has_primbox && have) {
    const int ta = ty1, tb = ty2;
    primbox = tb == G_BOX && (ta == G_SPHERE || ta == G_CAPSULE);
  }
  // (cylinders are compiled into the walking / standing kernels only: in the stepping kernels the extra live state cost 6.5 % of
  // jvrc_step's rate with no cylinder in the model, profiles/r06_jvrc_step_cylinder_ab.txt; humanoid_create refuses them there)
  bool cylpair = false;
  if constexpr (!BOXBOX) { if (m.has_cyl && have) cylpair = ty2 == G_CYLINDER || ty2 == G_ELLIPSOID; }
  // Plane-box pairs (the floor against a foot box; kernels without box-box pairs): lane 8 j + i tests corner i of the j-th such
  // pair -- the same expressions, corner by corner, as collide_pair's loop, which walks the eight corners one after the other on the
  // pair's own lane, twice (counting, writing).  A ballot gives every corner its rank among the corners in contact (the first four
  // count, mjc_PlaneBox) and the pair's lane its count; after the scan the corner lanes write their contacts themselves.
  bool pbpair = false, cemit = false;
  int cj = -1, crank = 0, cq = 0, cg1 = 0, cg2 = 0;
  double cdist = 0, cpos[3] = {0, 0, 0}, cnn[3] = {0, 0, 0};
  if constexpr (!BOXBOX) {
    const int npb = m.npb;
    if (npb > 0) {
      const int j = lane >> 3, i = lane & 7;
      bool pass = false;
      if (j < npb) {
        cj = j;
        cq = j == 0 ? m.pb_pair[0] : (j == 1 ? m.pb_pair[1] : (j == 2 ? m.pb_pair[2] : m.pb_pair[3]));
        cg1 = m.pair_i[PIS * cq]; cg2 = m.pair_i[PIS * cq + 1];
        const double mg = m.pair_d[PDS * cq];
        const double s2[3] = {m.pair_d[PDS * cq + PD_SIZE2], m.pair_d[PDS * cq + PD_SIZE2 + 1], m.pair_d[PDS * cq + PD_SIZE2 + 2]};
        double p1[3], p2[3], R2[9];
        for (int a = 0; a < 3; a++) { p1[a] = S.U[U_GPOS + 3 * cg1 + a]; p2[a] = S.U[U_GPOS + 3 * cg2 + a]; cnn[a] = S.U[U_GMAT + 9 * cg1 + 3 * a + 2]; }
        for (int a = 0; a < 9; a++) R2[a] = S.U[U_GMAT + 9 * cg2 + a];
        const double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
        const double dist = dot3(dif, cnn);
        const double v[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])};
        double corner[3];
        mat_vec(corner, R2, v);
        const double ld = dot3(cnn, corner);
        pass = !(dist + ld > mg || ld > 0);
        cdist = dist + ld;
        for (int a = 0; a < 3; a++) cpos[a] = corner[a] + p2[a] - cnn[a] * (dist + ld) * 0.5;
      }
      const typename RowMask<L::W_>::type bal = group_rows<L::W_>(__ballot(pass));
      if (cj >= 0) {
        const unsigned m8 = (unsigned)(bal >> (8 * cj)) & 0xffu;
        crank = __popc(m8 & ((1u << i) - 1u));
        cemit = pass && crank < 4;
      }
      if (have && ty1 == G_PLANE && ty2 == G_BOX) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
          if (jj < npb && m.pb_pair[jj] == lane) { pbpair = true; k.n = min(4, __popc((unsigned)(bal >> (8 * jj)) & 0xffu)); }
      }
    }
  }
  if (have && !boxpair && !primbox && !pbpair && !cylpair) collide_pair(k, m, S, lane, g1, g2, margin);
  if (m.has_primbox) { if (primbox) collide_primbox(k, m, S, lane, g1, g2, margin); }
  if constexpr (!BOXBOX) { if (m.has_cyl) { if (cylpair) collide_cyl(k, m, S, lane, g1, g2, margin); } }
  if constexpr (BOXBOX) {
    if (gany<L::W_>(boxpair)) {
      if (boxpair) { col_box_box(br, m, S, lane, g1, g2, margin); k.n = br.cnt; }
    }
  }
  int total;
  FINE_MARK(3, 1);
  const int mine = k.n;   // (pairs without a contact skip the writing pass: most of them, most of the time)
  const int base = gscan<L::W_>(k.n, &total) - k.n;
  FINE_MARK(3, 2);
  k.base = base; k.n = 0; k.write = 1;
  // more contacts than the group has row lanes for (stepping task, one env per wave): all of them go to the raw region of the HBM
  // workspace, exact copies are merged there, and the distinct ones come back with their multiplicity (see the workspace layout)
  bool big = false;
  if constexpr (BOXBOX && L::W_ == 64) big = bd != nullptr && total > NC;
  const int cap = big ? NCR : NC;
  if (big) { k.gd = bd; k.gi = bi; k.key = pkey; }
  if constexpr (!BOXBOX) {
    if (m.npb > 0) {   // the corner lanes write: slot = their pair's base (read off the pair's lane) + rank
      // (and only as many as the pair's lane reserved: 0 if the broad phase dropped the pair, so a corner lane can never write into
      // the slots of the next pair even if its own test and the pair lane's ever disagreed)
      int bq = 0, nq = 0;
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
        if (jj < m.npb) {
          const int bj = gbcast_i<L::W_>(base, m.pb_pair[jj]), nj = gbcast_i<L::W_>(mine, m.pb_pair[jj]);
          if (cj == jj) { bq = bj; nq = nj; }
        }
      if (cemit && crank < nq) {
        const double zero[3] = {0, 0, 0};
        ConSink<L> kc{&S, bq, crank, 1, cg1, cg2, cq};
        kc.emit(cdist, cpos, cnn, zero);
      }
    }
  }
  if (have && !boxpair && !primbox && !pbpair && !cylpair && mine > 0 && base < cap) collide_pair(k, m, S, lane, g1, g2, margin);
  if (m.has_primbox) { if (primbox && mine > 0 && base < cap) collide_primbox(k, m, S, lane, g1, g2, margin); }
  if constexpr (!BOXBOX) { if (m.has_cyl) { if (cylpair && mine > 0 && base < cap) collide_cyl(k, m, S, lane, g1, g2, margin); } }
  if constexpr (BOXBOX) {
    if (boxpair && base < cap) {
      const double zero[3] = {0, 0, 0};
      for (int q = 0; q < br.cnt; q++) k.emit(br.dist[q], &br.pos[3 * q], br.n, zero);
    }
  }
  FINE_MARK(3, 3);
  int ndist = min(total, NC);   // contacts the LDS arrays will hold
  bool merged = false;
  if constexpr (BOXBOX && L::W_ == 64) {
    if (big) {
      constexpr int W = L::W_;
      __syncthreads();   // (one wave per workgroup: orders the workspace writes above before the reads below)
      const int nr = min(total, NCR);
      // first[c]: the earliest contact c is a copy of.  Copies come in two kinds.  (a) The same pair class (contact parameters, dof
      // masks, roles of the two bodies: humanoid_create) with bitwise the same distance, position and frame: a foot corner inside
      // several overlapping boxes.  (b) The same MERGE class -- a robot body against a static geom, whichever of the two is geom1 --
      // with the frame mirrored (normal and second tangent negated: makeFrame of -n; with the roles of the bodies swapped the
      // four pyramid rows are the same four rows, the first two in the other order) and distance / position equal to within a few
      // ulp: the floor contact of that foot corner (plane-box: geom1 = floor) beside its box contacts (box-box: geom1 = foot), which
      // two narrow phases compute to within 1-2 ulp of each other (measured: |d dist| <= 1e-18, |d pos| <= 3e-17).  The group's
      // representative is its earliest member; the copies only raise its multiplicity.
      // (the scan's working arrays -- distance, class key, first -- live in LDS, in the contact-record cache of newton_big, idle here)
      MergeLds<L> ml(S);
      for (int c = lane; c < nr; c += W) {
        const float dc32 = ml.d32[c], vc32 = ml.v32[c], yc32 = ml.y32[c];
        const int key = ml.k16[c];
        // this contact's exact record (the loads are in flight while the filter walks LDS)
        const double dc = bd[AR_DIST + c];
        double pc[3], fc[9];
        for (int a = 0; a < 3; a++) pc[a] = bd[AR_POS + 3 * c + a];
        for (int a = 0; a < 9; a++) fc[a] = bd[AR_FRAME + 9 * c + a];
        const double dtol = 1e-17 + 4e-16 * fabs(dc);
        const float tolv = 4e-3f, tol32 = 1e-5f, rel32 = 2e-6f;   // (float32 ulp at |v| < 8192: 4.9e-4; of a coordinate below 30: 1.9e-6)
        int f = c;
        for (int e0 = 0; e0 < c && f == c; e0 += 16) {   // sixteen candidates per trip: four 16-byte LDS reads
          float v[16];
#pragma unroll
          for (int j = 0; j < 16; j += 4) {
            const float4 q4 = *reinterpret_cast<const float4*>(&ml.v32[e0 + j]);
            v[j] = q4.x; v[j + 1] = q4.y; v[j + 2] = q4.z; v[j + 3] = q4.w;
          }
          unsigned hits = 0;
#pragma unroll
          for (int j = 0; j < 16; j++) hits |= (fabsf(v[j] - vc32) <= tolv) ? (1u << j) : 0u;
          if (c - e0 < 16) hits &= (1u << (c - e0)) - 1u;   // candidates are the EARLIER contacts
          while (hits && f == c) {
            const int e = e0 + __ffs(hits) - 1;
            hits &= hits - 1;
            const int cle = ml.k16[e];
            if ((cle >> 1) != (key >> 1)) continue;
            if (fabsf(ml.d32[e] - dc32) > tol32 + rel32 * fabsf(dc32) || fabsf(ml.y32[e] - yc32) > tol32 + rel32 * fabsf(yc32)) continue;
            // the candidate's whole record in ONE round trip to the workspace (round 6: distance first, then position and frame, were
            // two dependent ones per verified copy -- and nearly every candidate that passes the float32 filters is a copy)
            const double dex = bd[AR_DIST + e];
            double pe[3], fe[9];
            for (int a = 0; a < 3; a++) pe[a] = bd[AR_POS + 3 * e + a];
            for (int a = 0; a < 9; a++) fe[a] = bd[AR_FRAME + 9 * e + a];
            const bool mirror = ((cle ^ key) & 1) != 0;
            bool same = fabs(dex - dc) <= dtol && (mirror || dex == dc);
            // (the two narrow phases agree to 1-2 ulp of the coordinate: an absolute floor for coordinates near zero, which are sums
            // of terms of order 0.1, and a relative part so that the copies still merge metres away from the origin -- one ulp
            // at |x| >= 4 m is 0.9e-15)
            for (int a = 0; a < 3; a++) same = same && (mirror ? fabs(pe[a] - pc[a]) <= 1e-15 + 4e-16 * fabs(pc[a]) : pe[a] == pc[a]);
            for (int a = 0; a < 9; a++) same = same && (mirror ? fabs(((a < 3 || a >= 6) ? -fe[a] : fe[a]) - fc[a]) <= 1e-15 : fe[a] == fc[a]);
            if (same) f = e;
          }
        }
        ml.f16[c] = (unsigned short)f;
      }
      SYNC();
      FINE_MARK(3, 6);
      // (equality up to a tolerance is not transitive: a contact may have matched a copy whose own match it failed on -- follow the
      // chain to the group's representative)
      int rep[(NCR + W - 1) / W];
#pragma unroll
      for (int q = 0; q < (NCR + W - 1) / W; q++) {
        const int c = q * W + lane;
        int f = c < nr ? ml.f16[c] : 0;
        if (c < nr) while (ml.f16[f] != f) f = ml.f16[f];
        rep[q] = f;
      }
      SYNC();
#pragma unroll
      for (int q = 0; q < (NCR + W - 1) / W; q++) { const int c = q * W + lane; if (c < nr) ml.f16[c] = (unsigned short)rep[q]; }
      // (the filter arrays are dead: rank / multiplicity / ground-reaction counts take their place)
      for (int c = lane; c < nr; c += W) { ml.cnt[c] = 0; ml.crl[c] = 0; }
      SYNC();
      // rank of the distinct contacts (in contact order), multiplicities by rank
      int nu = 0;
      for (int c0 = 0; c0 < nr; c0 += W) {
        const int c = c0 + lane;
        const bool uq = c < nr && ml.f16[c] == c;
        const unsigned long long bal = __ballot(uq);
        if (uq) ml.rank[c] = nu + __popcll(bal & ((1ull << lane) - 1ull));
        nu += __popcll(bal);
      }
      SYNC();
      for (int c = lane; c < nr; c += W) {   // (integer atomics in LDS: order-free)
        const int u = ml.rank[ml.f16[c]], q = ml.p16[c];
        atomicAdd(&ml.cnt[u], 1);
        // the copy's role in the ground-reaction query (robot_interface.py:269-301): geom1 off the robot, geom2 on a foot body;
        // right-foot copies count in the low half of the word, left-foot copies in the high half
        const int b2 = m.pair_i[PIS * q + PI_BODY2];
        if (m.pair_i[PIS * q + PI_ROOT1] != p.root_body) {
          if (b2 == p.rfoot_body) atomicAdd(&ml.crl[u], 1);
          if (b2 == p.lfoot_body) atomicAdd(&ml.crl[u], 1 << 16);
        }
      }
      SYNC();
      FINE_MARK(3, 7);
      merged = true;
      ndist = nu;
      const bool tolds = nu <= NC;
      // the distinct contacts, in contact order: back into the LDS arrays if they fit the row lanes, else into the unique region
      for (int c = lane; c < nr; c += W) {
        if (ml.f16[c] != c) continue;
        const int u = ml.rank[c];
        const double mult = (double)ml.cnt[u], nright = (double)(ml.crl[u] & 0xffff), nleft = (double)(ml.crl[u] >> 16);
        if (tolds) {
          S.U[U_CDIST + u] = bd[AR_DIST + c];
          for (int a = 0; a < 3; a++) S.con_pos[3 * u + a] = bd[AR_POS + 3 * c + a];
          for (int a = 0; a < 9; a++) S.U[U_CFRAME + 9 * u + a] = bd[AR_FRAME + 9 * c + a];
          S.con_g1[u] = bi[ARI_G1 + c]; S.con_g2[u] = bi[ARI_G2 + c]; S.con_pair[u] = ml.p16[c];
          if constexpr (L::STEP_) { S.con_mult[u] = mult; S.con_wr[u] = nright / mult; S.con_wl[u] = nleft / mult; }
        } else if (u < NCB) {
          const int q = ml.p16[c];
          const double* pd = m.pair_d + PDS * q;
          const double incm = pd[1], dist = bd[AR_DIST + c];
          bd[BW_DIST + u] = dist;
          for (int a = 0; a < 3; a++) bd[BW_POS + 3 * u + a] = bd[AR_POS + 3 * c + a];
          for (int a = 0; a < 9; a++) bd[BW_FRAME + 9 * u + a] = bd[AR_FRAME + 9 * c + a];
          bi[BWI_G1 + u] = bi[ARI_G1 + c]; bi[BWI_G2 + u] = bi[ARI_G2 + c]; bi[BWI_PAIR + u] = q;
          bd[BW_MULT + u] = mult; bd[BW_WR + u] = nright / mult; bd[BW_WL + u] = nleft / mult;
          // mj_contactParam, from the pair tables
          bi[BWI_XM + u] = m.pair_i[PIS * q + 3]; bi[BWI_M2 + u] = m.pair_i[PIS * q + 4];
          bd[BW_TRAN + u] = pd[10];
          bd[BW_MARGIN + u] = incm;
          bi[BWI_DIM + u] = (dist >= incm) ? 0 : m.pair_i[PIS * q + 2];
          bd[BW_MU + u] = pd[2];
          bd[BW_SOLREF + 2 * u] = pd[3]; bd[BW_SOLREF + 2 * u + 1] = pd[4];
          for (int a = 0; a < 5; a++) bd[BW_SOLIMP + 5 * u + a] = pd[5 + a];
        }
      }
      __syncthreads();
    }
  }
  FINE_MARK(3, 4);
  if (lane == 0) {
    const bool inlds = ndist <= NC;
    S.ncon = inlds ? ndist : 0;
    if constexpr (L::STEP_) S.nbig = inlds ? 0 : min(ndist, NCB);
    if (total > cap || (!inlds && ndist > NCB)) S.overflow = 1;   // sticky for the whole control step
  }
  SYNC();
  // mj_contactParam (lane = contact): a function of the geom pair, evaluated at create (humanoid_create: pair_d / pair_i)
  if (lane < S.ncon) {
    const int c = lane, q = S.con_pair[c];
    const double* pd = m.pair_d + PDS * q;
    const double incm = pd[1];
    S.con_xm[c] = m.pair_i[PIS * q + 3]; S.con_m2[c] = m.pair_i[PIS * q + 4];
    S.U[U_CTRAN + c] = pd[10];
    S.U[U_CMARGIN + c] = incm;
    S.con_dim[c] = (S.U[U_CDIST + c] >= incm) ? 0 : m.pair_i[PIS * q + 2];  // 0: excluded from the constraint set (gap)
    S.con_mu[c] = pd[2];
    S.U[U_CSOLREF + 2 * c] = pd[3]; S.U[U_CSOLREF + 2 * c + 1] = pd[4];
    for (int a = 0; a < 5; a++) S.U[U_CSOLIMP + 5 * c + a] = pd[5 + a];
    if constexpr (L::STEP_) {
      if (!merged) {   // no merge in this sub-step: every contact stands for itself
        const int b2 = m.pair_i[PIS * q + PI_BODY2];
        const bool floor1 = m.pair_i[PIS * q + PI_ROOT1] != p.root_body;
        S.con_mult[c] = 1.0; S.con_wr[c] = (floor1 && b2 == p.rfoot_body) ? 1.0 : 0.0; S.con_wl[c] = (floor1 && b2 == p.lfoot_body) ? 1.0 : 0.0;
      }
    }
  }
  SYNC();
  FINE_MARK(3, 5);
}

// getsolparam + getimpedance + KBIP + R for one row
__device__ __forceinline__ void row_params(HModelRef m, const double* sr_in, const double* si_in, double pos, double margin,
                                           double diagApprox, double* K, double* B, double* imp, double* R) {
  double sr0 = sr_in[0], sr1 = sr_in[1];
  if (!(m.disableflags & (1 << 11)) && sr0 > 0 && sr0 < 2 * m.timestep) sr0 = 2 * m.timestep;
  double s0 = fmin(0.9999, fmax(0.0001, si_in[0])), s1 = fmin(0.9999, fmax(0.0001, si_in[1])), s2 = fmax(0.0, si_in[2]);
  double s3 = fmin(0.9999, fmax(0.0001, si_in[3])), s4 = fmax(1.0, si_in[4]);
  double im;
  if (s0 == s1 || s2 <= HMINVAL) im = 0.5 * (s0 + s1);
  else {
    double x = fabs(qdiv(pos - margin, s2));
    if (x >= 1) im = s1;
    else if (x <= 0) im = s0;
    else {
      double y;
      if (s4 == 1) y = x;
      else if (s4 == 2) y = (x <= s3) ? qdiv(x * x, s3) : 1 - qdiv((1 - x) * (1 - x), 1 - s3);  // default solimp power, no pow()
      else if (x <= s3) y = pow(x, s4) / pow(s3, s4 - 1);
      else y = 1 - pow(1 - x, s4) / pow(1 - s3, s4 - 1);
      im = s0 + y * (s1 - s0);
    }
  }
  *imp = im;
  *R = fmax(HMINVAL, qdiv(1 - im, im) * diagApprox);
  if (sr0 > 0) {
    *K = qdiv(1, fmax(HMINVAL, s1 * s1 * sr0 * sr0 * sr1 * sr1));
    *B = qdiv(2, fmax(HMINVAL, s1 * sr0));
  } else {
    *K = qdiv(-sr0, fmax(HMINVAL, s1 * s1));
    *B = qdiv(-sr1, fmax(HMINVAL, s1));
  }
}
// mj_constraintUpdate for one row at residual x = J a - aref: limit / contact rows are one-sided quadratics, frictionloss
// rows (fl > 0) are Huber: quadratic for |x| < R fl, linear beyond with |force| = fl.
__device__ __forceinline__ void row_eval(bool valid, double fl, double D, double x, double* cost, double* force, double* dact) {
  double c = 0, f = 0, da = 0;
  if (valid) {
    if (fl > 0) {
      const double Rf = qdiv(fl, D);
      if (x <= -Rf) { f = fl; c = -0.5 * Rf * fl - fl * x; }
      else if (x >= Rf) { f = -fl; c = -0.5 * Rf * fl + fl * x; }
      else { f = -D * x; c = 0.5 * D * x * x; da = D; }
    } else if (x < 0) { f = -D * x; c = 0.5 * D * x * x; da = D; }
  }
  *cost = c; *force = f; *dact = da;
}
// first / second derivative contributions of one row along the search direction (jv = J search)
__device__ __forceinline__ void row_deriv(bool valid, double fl, double D, double x, double jv, double* d1, double* d2) {
  double a = 0, b = 0;
  if (valid) {
    if (fl > 0) {
      const double Rf = qdiv(fl, D);
      if (x <= -Rf) a = -fl * jv;
      else if (x >= Rf) a = fl * jv;
      else { a = D * x * jv; b = D * jv * jv; }
    } else if (x < 0) { a = D * x * jv; b = D * jv * jv; }
  }
  *d1 = a; *d2 = b;
}

// Dense Cholesky + solve with one dof per lane: `row` = this lane's row of the symmetric matrix by dof index (lower triangle: only the
// columns c < d are read), `hd` its diagonal entry, `x` its right-hand side element; returns its element of the solution.  The factor
// row stays in registers (statically indexed: the column loops are fully unrolled), the pivot row is read as LDS broadcasts off the
// dependency chain, the right-hand side is carried through the factorisation.  Packed lower triangle of the factor in the U_L region.
template <class L>
__device__ __forceinline__ double dense_factor_solve(L& S, double (&row)[NV], const double hd, const double x, const int dof_in, const bool prim_in) {
  double* A = S.U + U_L;
  double* xs = S.U + U_DG;
  const int dof = opaque_int(dof_in);
  const bool prim = opaque_int(prim_in ? 1 : 0) != 0;
  const int d = dof >= 0 ? dof : 0;
  SYNC();
  double y = x;   // right-hand side element, then y = L^-1 b
#pragma unroll
  for (int j = 0; j < NV; j++) {
    // s = H[d][j] - sum_{k<j} L[d][k] L[j][k] for the lanes d >= j (row[k] holds this lane's finished L[d][k]); lane j also
    // finishes y_j = (b_j - sum_{k<j} L[j][k] y_k) / L[j][j]
    double s = d == j ? hd : row[j], t = y;
#pragma unroll
    for (int k = 0; k < j; k++) {
      s -= row[k] * A[TRI(j, k)];
      t -= row[k] * xs[k];
    }
    if (prim && d == j) {
      const double piv = qsqrt(fmax(s, HMINVAL));
      A[TRI(j, j)] = piv;
      y = qdiv(t, piv);
      xs[j] = y;
    }
    SYNC();
    if (prim && d > j) {
      row[j] = qdiv(s, A[TRI(j, j)]);
      A[TRI(d, j)] = row[j];
    }
    SYNC();
  }
  // back substitution, column by column: x_j = y_j / L[j][j], then every lane d < j takes L[j][d] x_j off its own y
  double xd = 0.0;
#pragma unroll
  for (int j = NV - 1; j >= 0; j--) {
    if (prim && d == j) { xd = qdiv(y, A[TRI(j, j)]); xs[j] = xd; }
    SYNC();
    if (prim && d < j) y -= A[TRI(j, d)] * xs[j];
  }
  SYNC();
  return dof >= 0 ? xs[d] : 0.0;
}

// K^-1 x for a Hessian whose contacts couple the two chains (no block structure left): dense Cholesky with one dof per lane.
// It runs only in sub-steps with leg-leg contacts -- but in a batch of thousands of envs SOME wave has one in nearly every
// launch, and a launch lasts as long as its slowest wave: the first version (rolled loops, every operand through LDS, the
// chain A x chain B block summed entry by entry over all rows) took ~50 k cycles per solve and made those waves twice as long
// as the average one (scripts/tail_waves.py).  This version keeps the lane's row of the factor in registers (statically
// indexed: the column loops are fully unrolled), reads the pivot row as LDS broadcasts that do not sit on the dependency
// chain, carries the right-hand side through the factorisation (no separate forward substitution) and assembles the A x B
// block from the active rows only.  Packed lower triangle of the factor in LDS (the U_L region, which the chain solver leaves
// idle).  Hrow / hd: the chain-layout row of M + J^T D J (root + own-chain columns) as assembled for chain_solve;
// active_rows: the env's rows with a non-zero D (group_rows).
template <class L>
__device__ __forceinline__ double dense_lds_solve(L& S, const double (&Hrow)[NR], double hd, double x, int dof, bool prim, int coff,
                                                  typename RowMask<L::W_>::type active_rows) {
  const int d = dof >= 0 ? dof : 0;
  const bool chainB = prim && d >= 6 + NCH;
  // row d of H by dof index (lower triangle: only columns c <= d are used); the diagonal travels separately
  double row[NV];
#pragma unroll
  for (int c = 0; c < 6; c++) row[c] = Hrow[c];
  {
    // chain B rows: the chain A columns carry the coupling J^T D J only -- summed over the active rows
    double xb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) xb[c] = 0.0;
    while (active_rows) {
      const int r = first_row(active_rows);
      active_rows &= active_rows - 1;
      const double cj = S.U[U_DACT + r] * S.U[U_J + r * NV + d];
#pragma unroll
      for (int c = 0; c < NCH; c++) xb[c] += cj * S.U[U_J + r * NV + 6 + c];
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      row[6 + c] = chainB ? xb[c] : Hrow[6 + c];
      row[6 + NCH + c] = chainB ? Hrow[6 + c] : 0.0;
    }
  }
  return dense_factor_solve<L>(S, row, hd, x, dof, prim);
}

// ------------------------------------------------------------------------------------------------ many-contact Newton
// The constraint solve of a sub-step whose contacts live in the HBM workspace (fwd_collision: more than NC contacts; stepping task,
// one env per wave).  Same problem, same algorithm and the same stopping rules as the in-LDS solver of solve_tail -- primal Newton
// with exact line search and warm start (engine_solver.c) -- with the rows walked in strides of the 64 lanes instead of one row per
// lane: Jacobian rows, 1 / R, aref, the residual J a - aref, J search, force and active-D of every row in the workspace; the
// Hessian M + J^T D J assembled densely (lane = dof, all columns) and factorised by dense_factor_solve.  Also leaves what the task
// layer reads off the contacts of this forward pass (ground reaction forces, lowest foot contact, self-collision) in S.big_*.
// Slow by construction (~100 contacts = ~400 rows = seven chunks per sweep, five to six sweeps per sub-step): it exists so that the
// terrain of the reference is the terrain of the kernel, not to be fast.
template <class L>
__device__ __noinline__ void newton_big(HModelRef m, HParamsRef p, L& S, const int lane, gws_d __restrict__ bd, gws_i __restrict__ bi,
                                        const int dof, const bool prim, const double (&Mrow)[NR], const double mdiag, const double marm,
                                        const double fs, const double as, const bool (&uon)[3], const double (&uD)[3],
                                        const double (&uaref)[3], const double ufl, const bool anyunit, double& qacc_out, double& fcon_out) {
  constexpr int W = L::W_;
  const int dd = dof >= 0 ? dof : 0, cp = lane & 15;
  const int nc = S.nbig, nrow = 4 * nc, nchunk = (nrow + W - 1) / W;
  auto mprod = [&](double x) {
    double acc = marm * x;
    chain_mrow<L, 0>(Mrow, x, acc);
    if (cp < 6) acc = xhalf_sum(acc);
    GROUP_SYNC(W);
    return acc;
  };
  // The rows are processed in CHUNKS of W = 64 (16 contacts).  A chunk's Jacobian rows are staged in the LDS region the in-LDS solver
  // keeps its 64 rows in (U_J, idle on this path) and everything that walks rows one after the other -- J^T f, the Hessian's
  // J^T D J -- reads them from there, as the in-LDS solver does.  They are REBUILT from the contact records at every staging
  // (position, frame, friction, dof masks: 13 doubles per contact against 72 for its four rows): the first version walked the rows
  // in HBM -- a chain of ~400 dependent L2 round trips per product, jvrc_step fell from 1.5 M to 0.08 M env-steps/s -- the second
  // staged stored rows -- 61 KB per env and sweep, ~30 GB of HBM traffic per control step of 4096 envs: 0.16 M.
  // rows 4c .. 4c+3 = Jn +- mu Jt1, Jn +- mu Jt2 (condim 1: row 4c = Jn)
  // (up to NCK contacts -- the usual case: ~20 distinct contacts in LATERAL mode -- keep the 13 doubles and 3 ints a row rebuild
  // reads in LDS; beyond that the rebuilds read the workspace)
  constexpr int NCK = L::NCK_;
  const bool cached = nc <= NCK;
  if (cached) {
    for (int i = lane; i < nc * 13; i += W) {
      const int c = i / 13, a = i - 13 * c;
      S.bk_rec[i] = a < 3 ? bd[BW_POS + 3 * c + a] : (a < 12 ? bd[BW_FRAME + 9 * c + (a - 3)] : bd[BW_MU + c]);
    }
    for (int c = lane; c < nc; c += W) { S.bk_i[3 * c] = bi[BWI_XM + c]; S.bk_i[3 * c + 1] = bi[BWI_M2 + c]; S.bk_i[3 * c + 2] = bi[BWI_DIM + c]; }
    SYNC();
  }
  auto stage = [&](int ch) {
    const int c0 = ch * (W / 4), ncc = min(W / 4, nc - c0);
    for (int it = lane; it < (W / 4) * NV; it += W) {
      const int cl = it / NV, k = it - cl * NV, c = c0 + cl;
      double j0 = 0, j1 = 0, j2 = 0, j3 = 0;
      if (cl < ncc) {
        const unsigned bit = 1u << k;
        double pos[3], f[9], mu;
        int xm, m2, dim;
        if (cached) {
          const double* rc = &S.bk_rec[13 * c];
          for (int a = 0; a < 3; a++) pos[a] = rc[a];
          for (int a = 0; a < 9; a++) f[a] = rc[3 + a];
          mu = rc[12];
          xm = S.bk_i[3 * c]; m2 = S.bk_i[3 * c + 1]; dim = S.bk_i[3 * c + 2];
        } else {
          for (int a = 0; a < 3; a++) pos[a] = bd[BW_POS + 3 * c + a];
          for (int a = 0; a < 9; a++) f[a] = bd[BW_FRAME + 9 * c + a];
          mu = bd[BW_MU + c];
          xm = bi[BWI_XM + c]; m2 = bi[BWI_M2 + c]; dim = bi[BWI_DIM + c];
        }
        const bool in2 = ((unsigned)m2 & bit) != 0;
        double d[3] = {0, 0, 0};
        if ((unsigned)xm & bit) {
          double off[3], t[3];
          for (int a = 0; a < 3; a++) off[a] = pos[a] - S.com[a];
          cross3(t, &S.U[U_CDOF + 6 * k], off);
          const double sg = in2 ? 1.0 : -1.0;
          for (int a = 0; a < 3; a++) d[a] = sg * (S.U[U_CDOF + 6 * k + 3 + a] + t[a]);
        }
        const double jn = dot3(f, d);
        const bool pyr = dim != 1;
        const double t1 = mu * dot3(f + 3, d), t2 = mu * dot3(f + 6, d);
        j0 = pyr ? jn + t1 : jn; j1 = pyr ? jn - t1 : 0.0; j2 = pyr ? jn + t2 : 0.0; j3 = pyr ? jn - t2 : 0.0;
      }
      double* Jc = &S.U[U_J + 4 * cl * NV + k];
      Jc[0] = j0; Jc[NV] = j1; Jc[2 * NV] = j2; Jc[3 * NV] = j3;
    }
    SYNC();
  };
  auto jrow = [&](const double* v) {   // (this lane's staged row) . v
    double Jrow[NV];
#pragma unroll
    for (int k = 0; k < NV; k += 2) {
      const double2 ab = *reinterpret_cast<const double2*>(&S.U[U_J + lane * NV + k]);
      Jrow[k] = ab.x; Jrow[k + 1] = ab.y;
    }
    return row_dot<L>(Jrow, v);
  };
  // ---- row parameters (lane = row of the chunk): impedance / regulariser / reference acceleration
  for (int ch = 0; ch < nchunk; ch++) {
    stage(ch);
    const int r = ch * W + lane, c = r >> 2, e = r & 3;
    double D = 0, aref = 0;
    if (r < nrow) {
      const int dim = bi[BWI_DIM + c];
      if (dim == 3 || (dim == 1 && e == 0)) {
        const double jv0 = jrow(S.qvel);
        const double tran = bd[BW_TRAN + c], mu = bd[BW_MU + c];
        double K, B, imp, R;
        const double diag = dim == 1 ? tran : tran + mu * mu * tran;
        row_params(m, &bd[BW_SOLREF + 2 * c], &bd[BW_SOLIMP + 5 * c], bd[BW_DIST + c], bd[BW_MARGIN + c], diag, &K, &B, &imp, &R);
        if (dim == 3) R = fmax(HMINVAL, 2 * mu * mu * R);
        D = qdiv(1, R) * bd[BW_MULT + c];   // (k identical contacts merged by the collision stage: one row with k times the D)
        aref = -B * jv0 - K * imp * (bd[BW_DIST + c] - bd[BW_MARGIN + c]);
      }
      bd[BW_D + r] = D; bd[BW_AREF + r] = aref;   // D == 0: not a row (its cost, force and derivatives are zero)
    }
    SYNC();
  }
  __syncthreads();   // (one wave per workgroup: the workspace writes above are visible to the loads below)
  // unit rows of this lane's dof at acceleration a
  auto eval_units = [&](double a, double* cost, double* ufrc, double* udact) {
    double cu = 0, uf = 0, ud = 0;
    if (anyunit) {
      double c, f, da;
      row_eval(uon[0], ufl, uD[0], a - uaref[0], &c, &f, &da); cu += c; uf += f; ud += da;
      row_eval(uon[1], 0.0, uD[1], a - uaref[1], &c, &f, &da); cu += c; uf += f; ud += da;
      row_eval(uon[2], 0.0, uD[2], -a - uaref[2], &c, &f, &da); cu += c; uf -= f; ud += da;
    }
    *cost = prim ? cu : 0.0; *ufrc = uf; *udact = ud;
  };
  const double scale = 1.0 / (m.meaninertia * (NV > 1 ? NV : 1));
  double qacc = as, fcon = 0;
  if (!(m.disableflags & (1 << 7))) {   // warm start: cheaper of qacc_warmstart and qacc_smooth
    const double w = dof >= 0 ? S.qacc[dd] : 0.0;
    SYNC();
    if (prim) { S.U[U_VEC + dd] = w; S.U[U_VEC2 + dd] = as; }
    SYNC();
    double cw = 0, cs0 = 0;
    for (int ch = 0; ch < nchunk; ch++) {
      stage(ch);
      const int r = ch * W + lane;
      const double D = r < nrow ? bd[BW_D + r] : 0.0, aref = r < nrow ? bd[BW_AREF + r] : 0.0;
      const double jw = jrow(S.U + U_VEC), js = jrow(S.U + U_VEC2);
      double c, f, da;
      row_eval(D > 0, 0.0, D, jw - aref, &c, &f, &da); cw += c;
      row_eval(D > 0, 0.0, D, js - aref, &c, &f, &da); cs0 += c;
      SYNC();
    }
    double cu, tu, tv;
    eval_units(w, &cu, &tu, &tv); cw += cu;
    eval_units(as, &cu, &tu, &tv); cs0 += cu;
    const double Ma = mprod(w);
    if (prim) cw += 0.5 * (Ma - fs) * (w - as);
    double cs;
    gsum2<W>(cw, cs0, cw, cs);
    qacc = (cw > cs) ? as : w;
  }
  double cost = 0, oldcost = 0;
  for (int iter = 0; iter <= m.iterations; iter++) {
    SYNC();
    if (prim) S.U[U_VEC + dd] = qacc;
    SYNC();
    // one sweep over the rows: residual / force / active-D of every row, J^T f, and the Hessian's J^T D J (dense row of this lane's dof)
    double row[NV], c = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, hd = 0;
    {
      const bool chainB = prim && dd >= 6 + NCH;
#pragma unroll
      for (int k = 0; k < 6; k++) row[k] = Mrow[k];
#pragma unroll
      for (int k = 0; k < NCH; k++) { row[6 + k] = chainB ? 0.0 : Mrow[6 + k]; row[6 + NCH + k] = chainB ? Mrow[6 + k] : 0.0; }
    }
    for (int ch = 0; ch < nchunk; ch++) {
      stage(ch);
      const int r = ch * W + lane;
      const double D = r < nrow ? bd[BW_D + r] : 0.0, aref = r < nrow ? bd[BW_AREF + r] : 0.0;
      const double x = jrow(S.U + U_VEC) - aref;
      double cr, force, dactive;
      row_eval(D > 0, 0.0, D, x, &cr, &force, &dactive);
      c += cr;
      if (r < nrow) { bd[BW_JAR + r] = x; bd[BW_FRC + r] = force; bd[BW_DACT + r] = dactive; }
      S.U[U_EVEC + lane] = force; S.U[U_DACT + lane] = dactive;
      SYNC();
      for (int q = 0; q < W; q += 4) {
        f0 += S.U[U_J + q * NV + dd] * S.U[U_EVEC + q];
        f1 += S.U[U_J + (q + 1) * NV + dd] * S.U[U_EVEC + q + 1];
        f2 += S.U[U_J + (q + 2) * NV + dd] * S.U[U_EVEC + q + 2];
        f3 += S.U[U_J + (q + 3) * NV + dd] * S.U[U_EVEC + q + 3];
      }
      unsigned long long mm = __ballot(dactive != 0.0);
      while (mm) {
        const int q = __ffsll(mm) - 1;
        mm &= mm - 1;
        const double jl = S.U[U_J + q * NV + dd], cj = S.U[U_DACT + q] * jl;
        hd += cj * jl;
#pragma unroll
        for (int k = 0; k < NV; k += 2) {
          const double2 ab = *reinterpret_cast<const double2*>(&S.U[U_J + q * NV + k]);
          row[k] += cj * ab.x;
          row[k + 1] += cj * ab.y;
        }
      }
      SYNC();
    }
    double cu, ufrc, udact;
    eval_units(qacc, &cu, &ufrc, &udact);
    c += cu;
    hd += mdiag + udact;
    const double Ma = mprod(qacc);
    if (prim) c += 0.5 * (Ma - fs) * (qacc - as);
    oldcost = cost;
    double grad = 0;
    fcon = 0;
    if (dof >= 0) { fcon = ((f0 + f1) + (f2 + f3)) + ufrc; grad = Ma - fs - fcon; }
    double gn;
    gsum2<W>(c, prim ? grad * grad : 0.0, cost, gn);
    gn = qsqrt(gn);
    if (iter > 0) { if (scale * (oldcost - cost) < m.tolerance || scale * gn < m.tolerance) break; }
    else if (scale * gn < m.tolerance) break;
    if (iter == m.iterations) break;
    const double search = -dense_factor_solve<L>(S, row, hd, grad, dof, prim);
    if (prim) S.U[U_VEC2 + dd] = search;
    SYNC();
    for (int ch = 0; ch < nchunk; ch++) {   // J search of every row
      stage(ch);
      const int r = ch * W + lane;
      const double jv = jrow(S.U + U_VEC2);
      if (r < nrow) bd[BW_JV + r] = bd[BW_D + r] > 0 ? jv : 0.0;
      SYNC();
    }
    const double Mv = mprod(search);
    __syncthreads();
    double qg1, qg2;
    gsum2<W>(prim ? search * (Ma - fs) : 0.0, prim ? 0.5 * search * Mv : 0.0, qg1, qg2);
    const double xu0 = qacc - uaref[0], xu1 = qacc - uaref[1], xu2 = -qacc - uaref[2];
    // exact line search on the convex piecewise-quadratic: safeguarded Newton on its derivative.  The rows are strided over the lanes
    // (row 64 q + lane, q < NRB / 64); a lane's rows -- 1 / R, residual, J search -- are fetched ONCE into registers instead of being
    // re-read from the workspace in each of the (up to 40, typically 3-4 per Newton iteration) derivative evaluations: jvrc_step +5 %
    constexpr int RPL = NRB / W;
    double Dl[RPL], xl[RPL], vl[RPL];
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      const int r = q * W + lane;
      const bool ok = r < nrow;
      Dl[q] = ok ? bd[BW_D + r] : 0.0; xl[q] = ok ? bd[BW_JAR + r] : 0.0; vl[q] = ok ? bd[BW_JV + r] : 0.0;
    }
    auto deriv_all = [&](double a, double* d1, double* d2) {
      double r1 = 0, r2 = 0, s1, s2;
#pragma unroll
      for (int q = 0; q < RPL; q++) { row_deriv(Dl[q] > 0, 0.0, Dl[q], xl[q] + a * vl[q], vl[q], &s1, &s2); r1 += s1; r2 += s2; }
      if (anyunit && prim) {
        row_deriv(uon[0], ufl, uD[0], xu0 + a * search, search, &s1, &s2); r1 += s1; r2 += s2;
        row_deriv(uon[1], 0.0, uD[1], xu1 + a * search, search, &s1, &s2); r1 += s1; r2 += s2;
        row_deriv(uon[2], 0.0, uD[2], xu2 - a * search, -search, &s1, &s2); r1 += s1; r2 += s2;
      }
      gsum2<W>(r1, r2, *d1, *d2);
    };
    double alpha = 0;
    {
      double d1, d2;
      deriv_all(0.0, &d1, &d2);
      d1 += qg1; d2 += 2 * qg2;
      if (!(d1 >= 0 || d2 <= 0)) {
        const double d0 = fabs(d1);
        double lo = 0, hi = -1;
        for (int it = 0; it < 40; it++) {
          double a = alpha - qdiv(d1, d2);
          if (hi >= 0 && (a <= lo || a >= hi)) a = 0.5 * (lo + hi);
          deriv_all(a, &d1, &d2);
          d1 += 2 * a * qg2 + qg1;
          d2 += 2 * qg2;
          if (d1 < 0) lo = a; else hi = a;
          alpha = a;
          if (fabs(d1) <= 1e-14 * d0) break;
          if (hi >= 0 && hi - lo <= 4e-16 * hi) break;
        }
      }
    }
    if (alpha == 0) break;
    qacc += alpha * search;
  }
  __syncthreads();
  // ---- what the task layer reads off the contacts of this forward pass: GRF per foot (sum over its foot-floor contacts of the
  // norm of the decoded pyramid force), lowest foot-floor contact point, any foot contact, self-collision
  // (robot_interface.py:262-325, 472-484: "floor" = geom1 on a body outside the robot's tree, geom2 on the foot body)
  {
    double grf_r = 0, grf_l = 0, cz = 1e300;
    int anyfoot = 0, selfcol = 0;
    for (int c = lane; c < nc; c += W) {
      const int b1 = m.geom_i[GIS * (bi[BWI_G1 + c]) + GI_BODY], b2 = m.geom_i[GIS * (bi[BWI_G2 + c]) + GI_BODY];
      const bool floor1 = m.body_i[BIS * (b1) + BI_ROOT] != p.root_body;
      if (m.body_i[BIS * (b1) + BI_ROOT] == p.root_body && m.body_i[BIS * (b2) + BI_ROOT] == p.root_body) selfcol = 1;
      double fn = 0;
      const int r0 = 4 * c, dim = bi[BWI_DIM + c];
      if (dim == 3) {
        const double g0 = bd[BW_FRC + r0], g1 = bd[BW_FRC + r0 + 1], g2 = bd[BW_FRC + r0 + 2], g3 = bd[BW_FRC + r0 + 3], mu = bd[BW_MU + c];
        const double n = g0 + g1 + g2 + g3, t1f = mu * (g0 - g1), t2f = mu * (g2 - g3);
        fn = sqrt(n * n + t1f * t1f + t2f * t2f);
      } else if (dim == 1) fn = fabs(bd[BW_FRC + r0]);
      const double wr = bd[BW_WR + c], wl = bd[BW_WL + c];   // share of the merged copies that are floor contacts of the foot
      if (wr > 0) { grf_r += wr * fn; cz = fmin(cz, bd[BW_POS + 3 * c + 2]); anyfoot = 1; }
      if (wl > 0) { grf_l += wl * fn; cz = fmin(cz, bd[BW_POS + 3 * c + 2]); anyfoot = 1; }
    }
    grf_r = gsum<W>(grf_r); grf_l = gsum<W>(grf_l); cz = gmin<W>(cz);
    const bool af = gany<W>(anyfoot), sc = gany<W>(selfcol);
    SYNC();
    if (lane == 0) { S.big_grf_r = grf_r; S.big_grf_l = grf_l; S.big_cz = af ? cz : 0.0; S.big_anyfoot = af ? 1 : 0; S.big_selfcol = sc ? 1 : 0; }
    SYNC();
  }
  qacc_out = qacc; fcon_out = fcon;
}

// Everything of the sub-step behind the joint-space inertia: constraint rows, smooth acceleration, Newton, Euler.
// Dofs sit in the half-env-per-DPP-row layout of the chain solver: a root dof is held by two lanes, `prim` marks the one that
// counts in sums over dofs and writes the dof's results.  `cross`: some contact couples the two chains (group-uniform).
template <class L>
__device__ __forceinline__ void solve_tail(HModelRef m, HParamsRef p, L& S, const int lane, const int flags, long long* st_prof,
                                           const int dof, const bool prim, const bool cross, const double (&Mrow)[NR], const double mdiag,
                                           const double marm, const double qapp, const double bias, gws_d bd, gws_i bi) {
  constexpr int W = L::W_;
  PROF_BEGIN();
  const int dd = dof >= 0 ? dof : 0;   // (lanes without a dof shadow dof 0; nothing of theirs is used)
  const int cp = lane & 15;            // chain layout: position in the 16-lane row
  const bool rootb = dof >= 0 && !prim;
  const int coff = 6 + ((lane >> 4) & 1) * NCH;   // first dof of this row's chain
  // M x for the dof vector x held one element per dof lane
  auto mprod = [&](double x) {
    double acc = marm * x;   // (the row's own entry of Mrow carries the diagonal without the armature)
    chain_mrow<L, 0>(Mrow, x, acc);
    if (cp < 6) acc = xhalf_sum(acc);   // root rows: the two copies hold the coupling to one chain each
    GROUP_SYNC(W);
    return acc;
  };
  // K^-1 x for K = M + diag(extra) (+ J^T D J accumulated by the caller into Krow / kd)
  auto spd_solve = [&](double (&Krow)[NR], double kd, double x) { return chain_solve<L>(Krow, rootb ? 0.0 : kd, x, cp, rootb); };
  // ---- contact Jacobian, item = (contact, dof): rows 4c .. 4c+3 = Jn +- mu Jt1, Jn +- mu Jt2 (condim 1: row 4c = Jn)
  FINE_BEGIN(4);
  const int ncon = S.ncon, nrow = 4 * ncon;
  // Round 6: without a leg-leg contact a contact's rows are non-zero in the root's columns and in ONE chain's: the items are (contact, one of
  // those 6 + NCH columns) -- 8 x 12 = three trips for a JVRC env instead of the five of 8 x 18 -- and the lane of a chain column also
  // clears the same column of the other chain (the row loops of the solver read whole columns).  `cross`: every column is an item.
  const int ncol = cross ? NV : 6 + NCH;
  for (int it = lane; it < ncon * ncol; it += W) {
    const int c = cross ? it / NV : it / (6 + NCH), j = it - c * ncol;   // (two divisions by constants, not one by a variable)
    int k = j, kz = -1;
    if (!cross && j >= 6) {
      const unsigned cbm = (((1u << NCH) - 1u) << 6) << NCH;     // dofs of chain B
      const bool onB = ((unsigned)S.con_xm[c] & cbm) != 0;
      k = j + (onB ? NCH : 0);
      kz = j + (onB ? 0 : NCH);
    }
    const unsigned bit = 1u << k;
    const bool in2 = ((unsigned)S.con_m2[c] & bit) != 0;
    double d[3] = {0, 0, 0};
    if ((unsigned)S.con_xm[c] & bit) {
      double off[3], t[3];
      for (int a = 0; a < 3; a++) off[a] = S.con_pos[3 * c + a] - S.com[a];
      cross3(t, &S.U[U_CDOF + 6 * k], off);
      const double sg = in2 ? 1.0 : -1.0;
      for (int a = 0; a < 3; a++) d[a] = sg * (S.U[U_CDOF + 6 * k + 3 + a] + t[a]);
    }
    const double* f = &S.U[U_CFRAME + 9 * c];
    const double jn = dot3(f, d);
    const bool pyr = S.con_dim[c] != 1;
    const double mu = S.con_mu[c], t1 = mu * dot3(f + 3, d), t2 = mu * dot3(f + 6, d);
    double* Jc = &S.U[U_J + 4 * c * NV + k];
    Jc[0] = pyr ? jn + t1 : jn;
    Jc[NV] = pyr ? jn - t1 : 0.0;
    Jc[2 * NV] = pyr ? jn + t2 : 0.0;
    Jc[3 * NV] = pyr ? jn - t2 : 0.0;
    if (kz >= 0) {
      double* Jz = &S.U[U_J + 4 * c * NV + kz];
      Jz[0] = 0.0; Jz[NV] = 0.0; Jz[2 * NV] = 0.0; Jz[3 * NV] = 0.0;
    }
  }
  SYNC();
  FINE_MARK(4, 0);
  // ---- contact rows (lane = row): Jacobian row -> registers, impedance / regulariser / reference acceleration
  // (the row is re-read from LDS for each product instead of being held across the factorisations: 36 VGPRs that the
  // Cholesky needs more)
  // (round 6, measured and not adopted: the row held in registers across the products -- it fits since the v_fmac_f64_dpp change, 0.3213-0.3221 s
  // against 0.3218-0.3219 s per rollout, same box: profiles/r06_stepper_ab_jrow_other_envs.txt)
  auto jrow_dot = [&](const double* v) {
    double Jrow[NV];
#pragma unroll
    for (int k = 0; k < NV; k += 2) {
      const double2 ab = *reinterpret_cast<const double2*>(&S.U[U_J + lane * NV + k]);
      Jrow[k] = ab.x; Jrow[k + 1] = ab.y;
    }
    const double d = row_dot<L>(Jrow, v);
    return lane < nrow ? d : 0.0;   // rows beyond the last contact are not initialised
  };

  bool isrow = false;
  double D = 0, aref = 0;
  {
    const bool have = lane < nrow;
    const int c = have ? (lane >> 2) : 0, e = lane & 3, dim = S.con_dim[c];
    isrow = have && (dim == 3 || (dim == 1 && e == 0));
    const double jv0 = jrow_dot(S.qvel);
    if (isrow) {
      const double tran = S.U[U_CTRAN + c];
      const double mu = S.con_mu[c];
      double K, B, imp, R;
      const double diag = dim == 1 ? tran : tran + mu * mu * tran;
      row_params(m, &S.U[U_CSOLREF + 2 * c], &S.U[U_CSOLIMP + 5 * c], S.U[U_CDIST + c], S.U[U_CMARGIN + c], diag, &K, &B, &imp, &R);
      if (dim == 3) R = fmax(HMINVAL, 2 * mu * mu * R);  // every pyramid edge shares 2 mu^2 R(first edge)
      D = qdiv(1, R);
      if constexpr (L::STEP_) D *= S.con_mult[c];   // k identical contacts merged by the collision stage: one row with k times the D
      aref = -B * jv0 - K * imp * (S.U[U_CDIST + c] - S.U[U_CMARGIN + c]);
    }
  }
  FINE_MARK(4, 1);
  // ---- unit rows of this lane's dof: slot 0 frictionloss (Huber), 1 lower limit (J = +1), 2 upper limit (J = -1)
  bool uon[3] = {false, false, false};
  double uD[3] = {0, 0, 0}, uaref[3] = {0, 0, 0}, ufl = 0;
  const double qv = dof >= 0 ? S.qvel[dd] : 0.0;
  if (dof >= 0) {
    const int d = dof;
    // (every model constant the tests below read is requested here, in ONE batch of loads: inside their branches the limit test was a second,
    // dependent round trip to the tables behind the first -- kind / limited, then address / margin / range)
    const double fl = prm_floss(m, S, d);
    const int u_kind = m.dof_i[DIS * d + DI_KIND], u_limited = m.dof_i[DIS * d + DI_LIMITED], u_qadr = m.dof_i[DIS * d + DI_QADR];
    const double u_mg = m.dof_d[DDS * d + DD_MARGIN], u_lo = m.dof_d[DDS * d + DD_RANGE], u_hi = m.dof_d[DDS * d + DD_RANGE + 1];
    if (fl > 0) {
      double sr[2] = {m.dof_d[DDS * d + DD_SOLREF], m.dof_d[DDS * d + DD_SOLREF + 1]}, si[5], K, B, imp, R;
      for (int a = 0; a < 5; a++) si[a] = m.dof_d[DDS * d + DD_SOLIMP + a];
      row_params(m, sr, si, 0.0, 0.0, m.dof_d[DDS * d + DD_INVW], &K, &B, &imp, &R);
      uon[0] = true; uD[0] = qdiv(1, R); uaref[0] = -B * qv; ufl = fl;
    }
    const int j = m.dof_i[DIS * d + DI_JNT];
    if (u_kind >= 2 && u_limited) {
      const double q = S.qpos[u_qadr], mg = u_mg;
      const double dlo = q - u_lo, dhi = u_hi - q;
      if (dlo < mg || dhi < mg) {
        double sr[2] = {m.jnt_d[JDS * j + JD_SOLREF], m.jnt_d[JDS * j + JD_SOLREF + 1]}, si[5], K, B, imp, R;
        for (int a = 0; a < 5; a++) si[a] = m.jnt_d[JDS * j + JD_SOLIMP + a];
        if (dlo < mg) {
          row_params(m, sr, si, dlo, mg, m.dof_d[DDS * d + DD_INVW], &K, &B, &imp, &R);
          uon[1] = true; uD[1] = qdiv(1, R); uaref[1] = -B * qv - K * imp * (dlo - mg);
        }
        if (dhi < mg) {
          row_params(m, sr, si, dhi, mg, m.dof_d[DDS * d + DD_INVW], &K, &B, &imp, &R);
          uon[2] = true; uD[2] = qdiv(1, R); uaref[2] = B * qv - K * imp * (dhi - mg);
        }
      }
    }
  }
  // (most sub-steps of a walking robot have no joint at a limit and no dry friction: the per-dof unit rows are then skipped as
  // a whole in every cost / derivative evaluation of the solver)
  const bool anyunit = gany<W>(uon[0] || uon[1] || uon[2]);
  const bool anyrow = gany<W>(isrow) || anyunit;
  FINE_MARK(4, 2);
  PROF_MARK(4);
  // transmission + actuation (lane = actuator)
  if (lane < m.nu) {
    const double gear = m.act_d[ADS * (lane) + AD_GEAR];
    S.sq[lane] = qdiv(gear * S.qpos[m.act_i[AIS * lane + AI_QADR]], gear);  // actuator_length / gear, as the reference computes it
    S.sv[lane] = qdiv(gear * S.qvel[m.act_i[AIS * lane + AI_DADR]], gear);
    double f = 0;
    if (flags & 1) {
      double c = S.ctrl[lane];
      if (m.act_i[AIS * (lane) + AI_CTRLLIMITED]) c = fmin(m.act_d[ADS * (lane) + AD_CTRLRANGE + 1], fmax(m.act_d[ADS * (lane) + AD_CTRLRANGE], c));
      f = c;
      if (m.act_i[AIS * (lane) + AI_FORCELIMITED]) f = fmin(m.act_d[ADS * (lane) + AD_FORCERANGE + 1], fmax(m.act_d[ADS * (lane) + AD_FORCERANGE], f));
    }
    S.frc[lane] = f;
  }
  SYNC();
  // qfrc_smooth of this lane's dof = passive - bias + actuator + applied
  double fs = 0;
  if (dof >= 0) {
    double act = 0;
    const int u = m.dof_i[DIS * dd + DI_ACT];   // at most one actuator per dof (checked at create)
    if (u >= 0) act = m.dof_d[DDS * dd + DD_GEAR] * S.frc[u];
    fs = -prm_damp(m, S, dd) * qv - bias + act + qapp;
  }
  PROF_MARK(11);
  // qacc_smooth = M^-1 qfrc_smooth
  double as;
  {
    double r[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) r[k] = Mrow[k];
    as = spd_solve(r, mdiag, fs);
  }
  PROF_MARK(12);
  double qacc = as, fcon = 0;  // element of this lane's dof
  bool bigpath = false;
  if constexpr (L::STEP_ && W == 64) bigpath = bd != nullptr && S.nbig > 0;
  if (bigpath) {
    // more contacts than row lanes (fwd_collision left them in the HBM workspace; S.ncon is 0, the in-LDS row code above idled)
    if constexpr (L::STEP_ && W == 64) {
      // The callee is not inlined and takes its arrays by reference: it gets COPIES.  Handed the originals, they escape and every
      // access to them in this function -- Mrow in the Hessian / M-product loops, the unit-row parameters in each cost evaluation
      // of the ordinary Newton path -- is a scratch access (2048 waves x 50 KB of scratch do not fit the L2: 25 KB per sub-step
      // came back from beyond it).  Round 5, same box, jvrc_step @ 4096: rollout 1.336 -> 1.225 s (unit-row arrays alone 1.253 s;
      // the callee inlined instead 1.59 s).  Round 4 measured the same copies +3 % in the launch-per-step kernel of the time.
      double qo = as, fo = 0;
      bool uon_c[3] = {uon[0], uon[1], uon[2]};
      double uD_c[3] = {uD[0], uD[1], uD[2]}, uaref_c[3] = {uaref[0], uaref[1], uaref[2]};
      double Mrow_c[NR];
#pragma unroll
      for (int k = 0; k < NR; k++) Mrow_c[k] = Mrow[k];
      newton_big<L>(m, p, S, lane, bd, bi, dof, prim, Mrow_c, mdiag, marm, fs, as, uon_c, uD_c, uaref_c, ufl, anyunit, qo, fo);
      qacc = qo; fcon = fo;
    }
    S.efc_force[lane] = 0;
  } else if (anyrow) {
    // ------------------------------------------------------------ primal Newton (engine_solver.c)
    // cost / force / active-D of this lane's rows at acceleration a (ja = J a of the contact row, a = own dof element)
    auto eval_rows = [&](double ja, double a, double* cost, double* force, double* dactive, double* ufrc, double* udact) {
      double c, f, da;
      row_eval(isrow, 0.0, D, ja - aref, &c, force, dactive);
      double cs = c, uf = 0, ud = 0;
      if (anyunit) {
        double cu = 0;
        row_eval(uon[0], ufl, uD[0], a - uaref[0], &c, &f, &da); cu += c; uf += f; ud += da;
        row_eval(uon[1], 0.0, uD[1], a - uaref[1], &c, &f, &da); cu += c; uf += f; ud += da;
        row_eval(uon[2], 0.0, uD[2], -a - uaref[2], &c, &f, &da); cu += c; uf -= f; ud += da;
        if (prim) cs += cu;   // (the unit rows of a root dof count once)
      }
      *cost = cs; *ufrc = uf; *udact = ud;
    };
    FINE_BEGIN(7);
    const double scale = 1.0 / (m.meaninertia * (NV > 1 ? NV : 1));
    double warm_jw = 0, warm_js = 0, warm_Ma = 0;
    bool warm_pick_s = false, warm_have = false;
    // warm start: cheaper of qacc_warmstart and qacc_smooth
    if (!(m.disableflags & (1 << 7))) {
      const double w = dof >= 0 ? S.qacc[dd] : 0.0;   // qacc of the previous forward pass = qacc_warmstart
      SYNC();
      if (prim) { S.U[U_VEC + dd] = w; S.U[U_VEC2 + dd] = as; }
      SYNC();
      const double jw = jrow_dot(S.U + U_VEC), Ma = mprod(w), js = jrow_dot(S.U + U_VEC2);
      double cw, cs0, tf, td, tu, tv;
      eval_rows(jw, w, &cw, &tf, &td, &tu, &tv);
      eval_rows(js, as, &cs0, &tf, &td, &tu, &tv);
      if (prim) cw += 0.5 * (Ma - fs) * (w - as);
      double cs;
      gsum2<W>(cw, cs0, cw, cs);
      qacc = (cw > cs) ? as : w;
      warm_jw = jw; warm_js = js; warm_Ma = Ma; warm_pick_s = cw > cs; warm_have = true;
    }
    FINE_MARK(7, 0);
    double cost = 0, oldcost = 0;
    double ja_run = 0, Ma_run = 0;
    bool have_prod = false;
    // (the products of the start point were formed for the warm-start comparison: round 4, same box -0.5 %)
    if (warm_have) { ja_run = warm_pick_s ? warm_js : warm_jw; Ma_run = warm_pick_s ? mprod(as) : warm_Ma; have_prod = true; }
    for (int iter = 0; iter <= m.iterations; iter++) {
      if (st_prof && lane == 0) st_prof[6] += 1;    // diagnostic: Newton passes (cost evaluations) of env 0
      // J qacc and M qacc: computed for the first iterate, then advanced with the step (J search and M search exist for the line
      // search anyway) -- mj_solNewton's own bookkeeping (engine_solver.c: Jaref += alpha jv, Ma += alpha Mv).  Round 4, same box:
      // control step -1.5 % (round 3 measured +1 %: two more values live across the line search then cost spills)
      if (!have_prod) {
        SYNC();
        if (prim) S.U[U_VEC + dd] = qacc;
        SYNC();
        ja_run = jrow_dot(S.U + U_VEC); Ma_run = mprod(qacc);
        have_prod = true;
      }
      const double ja = ja_run, Ma = Ma_run;
      double c, force, dactive, ufrc, udact;
      eval_rows(ja, qacc, &c, &force, &dactive, &ufrc, &udact);
      if (prim) c += 0.5 * (Ma - fs) * (qacc - as);
      oldcost = cost;
      FINE_MARK(7, 1);
      S.U[U_EVEC + lane] = force; S.efc_force[lane] = force; S.U[U_DACT + lane] = dactive;   // lane = contact row (NE == W)
      SYNC();
      double grad = 0;
      fcon = 0;
      {
        double f0 = 0, f1 = 0, f2 = 0, f3 = 0;
        int r = 0;
        for (; r + 8 <= nrow; r += 8) {   // two contacts per trip: eight independent LDS reads in flight (round 4, same box: -1.1 %)
          const double a0 = S.U[U_J + r * NV + dd], a1 = S.U[U_J + (r + 1) * NV + dd], a2 = S.U[U_J + (r + 2) * NV + dd], a3 = S.U[U_J + (r + 3) * NV + dd];
          const double a4 = S.U[U_J + (r + 4) * NV + dd], a5 = S.U[U_J + (r + 5) * NV + dd], a6 = S.U[U_J + (r + 6) * NV + dd], a7 = S.U[U_J + (r + 7) * NV + dd];
          f0 += a0 * S.U[U_EVEC + r]; f1 += a1 * S.U[U_EVEC + r + 1]; f2 += a2 * S.U[U_EVEC + r + 2]; f3 += a3 * S.U[U_EVEC + r + 3];
          f0 += a4 * S.U[U_EVEC + r + 4]; f1 += a5 * S.U[U_EVEC + r + 5]; f2 += a6 * S.U[U_EVEC + r + 6]; f3 += a7 * S.U[U_EVEC + r + 7];
        }
        if (r < nrow) {
          f0 += S.U[U_J + r * NV + dd] * S.U[U_EVEC + r];
          f1 += S.U[U_J + (r + 1) * NV + dd] * S.U[U_EVEC + r + 1];
          f2 += S.U[U_J + (r + 2) * NV + dd] * S.U[U_EVEC + r + 2];
          f3 += S.U[U_J + (r + 3) * NV + dd] * S.U[U_EVEC + r + 3];
        }
        if (dof >= 0) {
          fcon = ((f0 + f1) + (f2 + f3)) + ufrc;
          grad = Ma - fs - fcon;
        }
      }
      FINE_MARK(7, 2);
      double gn;
      gsum2<W>(c, prim ? grad * grad : 0.0, cost, gn);   // the cost of this iterate and the squared gradient norm in one reduction
      gn = qsqrt(gn);
      if (iter > 0) { if (scale * (oldcost - cost) < m.tolerance || scale * gn < m.tolerance) break; }
      else if (scale * gn < m.tolerance) break;
      if (iter == m.iterations) break;
      // H = M + J^T D_active J, row of this lane's dof accumulated in the registers the factorisation works on; the diagonal
      // entry travels separately (hd)
      FINE_MARK(7, 3);
      FINE_BEGIN(2);
      PROF_MARK(7);    // (Newton: cost / gradient passes in slot 7, Hessian + factor + solve in slot 2, line search in slot 15)
      double Hrow[NR], hd = mdiag + udact;
#pragma unroll
      for (int k = 0; k < NR; k++) Hrow[k] = Mrow[k];
      // Only the env's own active rows contribute.  The two envs of a wave walk their own row lists side by side (the row index
      // is a per-lane value): max(|A|, |B|) trips instead of |A or B|, and no env ever touches a Jacobian row beyond its own
      // 4 ncon -- those are never written in a sub-step and hold whatever the LDS held before (round 4: the union loop of
      // round 3 multiplied them by D = 0, which a NaN bit pattern left by another kernel survives).
      const typename RowMask<W>::type mm_all = group_rows<W>(__ballot(dactive != 0.0));
      // Round 6: each HALF of the env walks the rows of its own chain only.  Without a leg-leg contact (`cross`) every row touches one
      // chain at most (root-only rows: counted with chain A), and the columns of the other chain are zeros: the rows of foot B used to
      // pass through the lanes of chain A (and the reverse) multiplied by 0 -- two feet on the ground = 32 trips' worth of rows where each
      // half needs 16.  The root-root block is then the sum of what the two copies of the root rows collect, which is exactly how
      // chain_solve merges the copies (xhalf_sum of the Schur complements): copy B keeps its share instead of being cleared.
      typename RowMask<W>::type mm_mine = mm_all;
      if (!cross) {
        const unsigned cbm = (((1u << NCH) - 1u) << 6) << NCH;     // dofs of chain B
        const bool rowB = lane < nrow && ((unsigned)S.con_xm[lane >> 2] & cbm) != 0;
        const typename RowMask<W>::type rowsB = group_rows<W>(__ballot(rowB));
        mm_mine = ((lane >> 4) & 1) ? (mm_all & rowsB) : (mm_all & ~rowsB);
      }
      if (rootb) hd = 0.0;   // (copy B of a root dof: M's diagonal and the unit rows live in copy A)
      {
        typename RowMask<W>::type mm = mm_mine;
        if constexpr (!L::STEP_) {
          // contact by contact (round 6): the four pyramid rows of a contact sit at fixed offsets from its first row, so the walk of the mask is
          // one step per CONTACT with at least one active row and the rows' addresses are immediates; an inactive row of such a contact
          // rides along with D = 0 (the usual contact has all four active)
          typename RowMask<W>::type mc = (mm | (mm >> 1) | (mm >> 2) | (mm >> 3)) & (typename RowMask<W>::type)0x1111111111111111ull;
          while (mc) {
            const int r0 = first_row(mc);
            mc &= mc - 1;
            const double* Jc = &S.U[U_J + r0 * NV + dd];
            const double* Dc = &S.U[U_DACT + r0];
            double jl[4], cj[4];
#pragma unroll
            for (int q = 0; q < 4; q++) jl[q] = Jc[q * NV];
#pragma unroll
            for (int q = 0; q < 4; q++) cj[q] = Dc[q] * jl[q];
#pragma unroll
            for (int q = 0; q < 4; q++) hd += cj[q] * jl[q];
            // (the row's other entries are the `jl` of the other lanes of this half -- row position e holds the dof of column e of
            // Hrow -- and come through DPP inside the multiply-add: six 16-byte LDS reads per row and their addresses are gone)
#pragma unroll
            for (int q = 0; q < 4; q++) fmac_outer<NR>(Hrow, jl[q], cj[q]);
          }
        } else {   // (the stepping task's merged contacts are more often partly active: the walk by row pairs measured 0.4 % faster there)
          while (mm) {      // two active rows per trip: their LDS reads are in flight together, the sums keep the row order
                            // (round 4, same box: -1.9 %; four rows per trip: +1 %; round 3's `#pragma unroll` of the one-row loop had lost)
            const int r = first_row(mm);
            mm &= mm - 1;
            const bool two = mm != 0;
            const int r2 = two ? first_row(mm) : r;
            if (two) mm &= mm - 1;
            const double jl = S.U[U_J + r * NV + dd], jl2 = S.U[U_J + r2 * NV + dd];
            const double cj = S.U[U_DACT + r] * jl, cj2 = two ? S.U[U_DACT + r2] * jl2 : 0.0;
            hd += cj * jl;
            hd += cj2 * jl2;
            fmac_outer<NR>(Hrow, jl, cj);
            fmac_outer<NR>(Hrow, jl2, cj2);
          }
        }
        GROUP_SYNC(W);   // (the halves leave the loop after different trip counts, and it holds cross-lane operations)
        if (rootb && cross) {   // (dense fallback: the root-root block lives in copy A)
#pragma unroll
          for (int k = 0; k < 6; k++) Hrow[k] = 0.0;
        }
      }
      FINE_MARK(2, 0);
      const double search = cross ? -dense_lds_solve<L>(S, Hrow, hd, grad, dof, prim, coff, mm_all) : -chain_solve<L>(Hrow, hd, grad, cp, rootb);
      FINE_MARK(2, 1);
      PROF_MARK(2);
      if (prim) S.U[U_VEC2 + dd] = search;
      SYNC();
      const double jv = jrow_dot(S.U + U_VEC2), Mv = mprod(search);
      double qg1, qg2;
      gsum2<W>(prim ? search * (Ma - fs) : 0.0, prim ? 0.5 * search * Mv : 0.0, qg1, qg2);
      // exact line search on the convex piecewise-quadratic: safeguarded Newton on its derivative
      const double x0 = ja - aref, xu0 = qacc - uaref[0], xu1 = qacc - uaref[1], xu2 = -qacc - uaref[2];
      auto deriv_rows = [&](double a, double* d1, double* d2) {
        double r1, r2, s1, s2;
        row_deriv(isrow, 0.0, D, x0 + a * jv, jv, &r1, &r2);
        if (anyunit && prim) {
          row_deriv(uon[0], ufl, uD[0], xu0 + a * search, search, &s1, &s2); r1 += s1; r2 += s2;
          row_deriv(uon[1], 0.0, uD[1], xu1 + a * search, search, &s1, &s2); r1 += s1; r2 += s2;
          row_deriv(uon[2], 0.0, uD[2], xu2 - a * search, -search, &s1, &s2); r1 += s1; r2 += s2;
        }
        *d1 = r1; *d2 = r2;
      };
      double alpha = 0;
      {
        double r1, r2;
        deriv_rows(0.0, &r1, &r2);
        double d1, d2;
        gsum2<W>(r1, r2, d1, d2);
        d1 += qg1; d2 += 2 * qg2;
        if (!(d1 >= 0 || d2 <= 0)) {
          const double d0 = fabs(d1);
          double lo = 0, hi = -1;
          for (int it = 0; it < 40; it++) {
            if (st_prof && lane == 0) st_prof[14] += 1;   // diagnostic: line-search passes of env 0
            double a = alpha - qdiv(d1, d2);
            if (hi >= 0 && (a <= lo || a >= hi)) a = 0.5 * (lo + hi);
            deriv_rows(a, &r1, &r2);
            gsum2<W>(r1, r2, d1, d2);
            d1 += 2 * a * qg2 + qg1;
            d2 += 2 * qg2;
            if (d1 < 0) lo = a; else hi = a;
            alpha = a;
            if (fabs(d1) <= 1e-14 * d0) break;
            if (hi >= 0 && hi - lo <= 4e-16 * hi) break;
          }
        }
      }
      PROF_MARK(15);
      FINE_BEGIN(7);
      if (alpha == 0) break;
      qacc += alpha * search;
      ja_run += alpha * jv; Ma_run += alpha * Mv;
    }
  } else {
    S.efc_force[lane] = 0;
  }
  SYNC();
  if (prim) S.qacc[dd] = qacc;   // mj_fwdConstraint: also the next warm start
  SYNC();
  FINE_MARK(7, 4);
  PROF_MARK(7);
  if (!(flags & 2)) return;
  // ------------------------------------------------------------ mj_Euler (implicit joint damping) + mj_advance
  double anew = qacc;
  const double h = m.timestep;
  const bool eulerdamp = !(m.disableflags & (1 << 14));
  if (eulerdamp) {
    double r[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) r[k] = Mrow[k];
    const double dm = dof >= 0 ? prm_damp(m, S, dd) : 0.0;
    anew = spd_solve(r, mdiag + h * dm, fs + fcon);
  }
  PROF_MARK(13);
  if (prim) S.qvel[dd] = qv + h * anew;
  SYNC();
  if (lane < m.njnt) {
    const int j = lane, qa = m.jnt_i[JIS * (j) + JI_QADR], da = m.jnt_i[JIS * (j) + JI_DADR];
    if (m.jnt_i[JIS * (j) + JI_TYPE] == JT_FREE) {
      for (int k = 0; k < 3; k++) S.qpos[qa + k] += h * S.qvel[da + k];
      double w[3] = {S.qvel[da + 3], S.qvel[da + 4], S.qvel[da + 5]};
      double ang = h * normalize3(w), qr[4], q[4] = {S.qpos[qa + 3], S.qpos[qa + 4], S.qpos[qa + 5], S.qpos[qa + 6]};
      axis_angle_quat(qr, w, ang);
      normalize4(q);
      mul_quat(q, q, qr);
      normalize4(q);
      for (int k = 0; k < 4; k++) S.qpos[qa + 3 + k] = q[k];
    } else {
      S.qpos[qa] += h * S.qvel[da];
    }
  }
  SYNC();
  PROF_MARK(8);
}

// One mj_forward (+ Euler).  flags: bit0 actuation enabled, bit1 integrate.
// On return S.qacc / S.efc_force / contacts / S.sq,sv,frc describe THIS forward pass (the "stale" fields of note S); S.qacc of
// the previous pass is the warm start of this one.
//
// Constraint rows.  MuJoCo orders them frictionloss dofs, joint limits, contacts.  Here the pyramid rows of contact c are
// rows (= lanes) 4c .. 4c+3 with their Jacobian row in LDS, and the frictionloss / limit rows -- whose
// Jacobians are +-unit vectors -- are three scalars slots of their dof's lane (0 frictionloss, 1 lower limit, 2 upper limit):
// J x is the lane's own element, J^T f lands on the lane's own dof, J^T D J on its own diagonal entry.  Only the summation
// order differs from the row order of the reference; every row is there.
template <bool BOXBOX, class L>
__device__ __forceinline__ void substep(HModelRef m, HParamsRef p, L* SG0, int flags, long long* st_prof, gtab_d ter, gws_d bd, gws_i bi) {
  constexpr int W = L::W_;
#if defined(__HIP_DEVICE_COMPILE__) && defined(LHW_SUBSTEP_PRIO)
  // (Resident rollout kernels only -- lhw_humanoid_rollout.hip defines LHW_SUBSTEP_PRIO.  In the launch-per-step pipeline the
  // policy launch of one rollout group runs beside the other group's stepper waves and must not queue behind waves that hold a
  // raised priority: measured there, rollout +6 %.)
  // The two wavefronts of a SIMD do not share its issue slots evenly: at equal priority the older one wins every arbitration tie.
  // Over a resident rollout it ran ~12 % ahead of its partner (wave totals clustered at 0.885 and 1.115 of the mean) and left it
  // to finish alone, with nobody to fill its stalls; within a control-step launch it is the spread between the fast low block
  // indices and the slow high ones.  The issue priority therefore alternates between the two on a clock BOTH read: bit
  // LHW_PRIO_SHIFT of the shader clock xor the parity of the wave's slot id in its SIMD (HW_ID[3:0]), re-evaluated at every
  // sub-step -- at any time one of the two is ahead in line, each for half the time.  Period (profiles/r05_wave_priority.txt,
  // rollout of jvrc_walk @ 4096): 2^15 / 2^16 ticks (under a sub-step) 0.412 / 0.410 s, 2^18 0.3975, 2^20 0.394, 2^22 .. 2^26
  // 0.392, 2^28 (a third of the rollout) 0.398; equal priorities 0.425.  2^24 ticks = about eight control steps.
  {
#ifndef LHW_PRIO_SHIFT
#define LHW_PRIO_SHIFT 24
#endif
    const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, bits [3:0]: wave slot within the SIMD
    const unsigned ph = (unsigned)((unsigned long long)clock64() >> LHW_PRIO_SHIFT);
#ifndef LHW_PRIO_LEVEL
#define LHW_PRIO_LEVEL 1
#endif
    if ((hw ^ ph) & 1u) __builtin_amdgcn_s_setprio(LHW_PRIO_LEVEL); else __builtin_amdgcn_s_setprio(0);
  }
#endif
  long long prof_t;
  { FRESH_GROUP(W, SG0); prof_t = (st_prof && lane == 0) ? (long long)clock64() : 0; fwd_kinematics<BOXBOX>(m, S, lane); PROF_MARK(0); }
  { FRESH_GROUP(W, SG0); fwd_collision<BOXBOX>(m, p, S, lane, ter, bd, bi); PROF_MARK(3); }   // stage A temporaries (geom frames) die with fwd_com
  FRESH_GROUP(W, SG0);
  // The chain solver needs the [root | chain A | chain B] block structure of M (checked at create).  A contact between bodies
  // of the two chains couples them in the Newton Hessian (rare: it is a self-collision, i.e. the last control step of an
  // episode): those Hessians are factorised by the looped dense Cholesky in LDS instead (dense_lds_solve).
  bool cross = false;
  if (lane < S.ncon && S.con_dim[lane] != 0) {
    const unsigned m2 = (unsigned)S.con_m2[lane], m1 = (unsigned)S.con_xm[lane] ^ m2;
    const unsigned ca = ((1u << NCH) - 1u) << 6, cb = ca << NCH;
    cross = ((m1 & ca) && (m2 & cb)) || ((m1 & cb) && (m2 & ca));
  }
  cross = gany<W>(cross);
  const int cp = lane & 15, hh = (lane >> 4) & 1;
  const int dof = (lane < 32 && cp < NR) ? (cp < 6 ? cp : 6 + hh * NCH + (cp - 6)) : -1;
  const bool prim = dof >= 0 && (cp >= 6 || hh == 0);
  fwd_com(m, p, S, lane);
  PROF_MARK(1);
  double Mrow[NR], mdiag, marm, bias, qapp;
  chain_dynamics(m, p, S, lane, dof, prim, Mrow, mdiag, marm, bias, qapp);
  PROF_MARK(5);
  solve_tail(m, p, S, lane, flags, st_prof, dof, prim, cross, Mrow, mdiag, marm, qapp, bias, bd, bi);
}
// ------------------------------------------------------------------------------------------------ task layer
__device__ __forceinline__ void sample_ref(HParamsRef p, unsigned genv, unsigned stream, unsigned counter, unsigned slot0,
                                           int mode, double* ref) {
  if (mode == MODE_STANDING) {
    for (int k = 0; k < 3; k++) ref[k] = lhw_rng_uniform(p.seed, genv, stream, counter, slot0 + k, -1.0, 1.0);
  } else if (mode == MODE_INPLACE) {
    ref[0] = lhw_rng_uniform(p.seed, genv, stream, counter, slot0, -0.5, 0.5); ref[1] = 0; ref[2] = 0;
  } else {
    ref[0] = 0; ref[1] = lhw_rng_uniform(p.seed, genv, stream, counter, slot0, 0.0, 0.4); ref[2] = 0;
  }
}

// transforms3d quat2euler 'sxyz' roll/pitch (tasks/observations.py:22)
__device__ __forceinline__ void quat_roll_pitch(const double* q, double* roll, double* pitch) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double Nq = w * w + x * x + y * y + z * z;
  const double s = Nq > 2.220446049250313e-16 ? 2.0 / Nq : 0.0;
  const double X = x * s, Y = y * s, Z = z * s;
  const double wX = w * X, wY = w * Y, wZ = w * Z, xX = x * X, xY = x * Y, xZ = x * Z, yY = y * Y, yZ = y * Z, zZ = z * Z;
  const double M00 = 1.0 - (yY + zZ), M10 = xY + wZ, M20 = xZ - wY, M21 = yZ + wX, M22 = 1.0 - (xX + yY), M11 = 1.0 - (xX + zZ), M12 = yZ - wX;
  const double cy = sqrt(M00 * M00 + M10 * M10);
  if (cy > 4.0 * 2.220446049250313e-16) { *roll = atan2(M21, M22); *pitch = atan2(-M20, cy); }
  else { *roll = atan2(-M12, M11); *pitch = atan2(-M20, cy); }
}

// (forced inline: as a call -- the compiler's choice -- it cost the walking kernel 36 B of scratch per lane and its mode_ref argument a
// home in memory; round 5, same box, jvrc_walk @ 4096: rollout 0.3931 -> 0.3862 s.  The H1 tasks' write_obs_h1 / randomize_dynamics
// inlined the same way LOSE 1.5 % on h1 @ 8192 and stay calls.)
template <class L>
__device__ __forceinline__ void write_obs(HModelRef m, HParamsRef p, L& S, int lane, int phase, int mode, const double* mode_ref,
                          float* o) {
  // get_obs (base_humanoid_env.py:177-197): fresh root quaternion / angular velocity, stale motor pos/vel
  if (lane == 0) {
    double r, pt;
    quat_roll_pitch(&S.qpos[3], &r, &pt);
    o[0] = (float)r; o[1] = (float)pt;
    o[2] = (float)S.qvel[3]; o[3] = (float)S.qvel[4]; o[4] = (float)S.qvel[5];
    const double ang = 2 * 3.141592653589793 * (double)phase / (double)p.period;
    o[29] = (float)sin(ang); o[30] = (float)cos(ang);
    o[31] = mode == MODE_FORWARD ? 1.f : 0.f; o[32] = mode == MODE_INPLACE ? 1.f : 0.f; o[33] = mode == MODE_STANDING ? 1.f : 0.f;
    o[34] = (float)mode_ref[0]; o[35] = (float)mode_ref[1]; o[36] = (float)mode_ref[2];
  }
  if (lane < 12) { o[5 + lane] = (float)S.sq[lane]; o[17 + lane] = (float)S.sv[lane]; }
}

// jvrc_step observation (jvrc_step.py:66-77): robot state, clock, goal steps x[2] y[2] z[2] theta[2]
template <class L>
__device__ void write_obs_step(HModelRef m, HParamsRef p, L& S, int lane, int phase, const double* goal, float* o) {
  if (lane == 0) {
    double r, pt;
    quat_roll_pitch(&S.qpos[3], &r, &pt);
    o[0] = (float)r; o[1] = (float)pt;
    o[2] = (float)S.qvel[3]; o[3] = (float)S.qvel[4]; o[4] = (float)S.qvel[5];
    const double ang = 2 * 3.141592653589793 * (double)phase / (double)p.period;
    o[29] = (float)sin(ang); o[30] = (float)cos(ang);
    for (int k = 0; k < 8; k++) o[31 + k] = (float)goal[k];
  }
  if (lane < 12) { o[5 + lane] = (float)S.sq[lane]; o[17 + lane] = (float)S.sv[lane]; }
}

// external state of the walking envs (jvrc_walk.py:65-67, h1_walk.py:118-123): clock, mode one-hot, mode reference
__device__ __forceinline__ void write_obs_walk_ext(HParamsRef p, int lane, int phase, int mode, const double* mode_ref, float* o) {
  if (lane == 0) {
    const double ang = 2 * 3.141592653589793 * (double)phase / (double)p.period;
    o[0] = (float)sin(ang); o[1] = (float)cos(ang);
    o[2] = mode == MODE_FORWARD ? 1.f : 0.f; o[3] = mode == MODE_INPLACE ? 1.f : 0.f; o[4] = mode == MODE_STANDING ? 1.f : 0.f;
    o[5] = (float)mode_ref[0]; o[6] = (float)mode_ref[1]; o[7] = (float)mode_ref[2];
  }
}

// H1 robot state (h1_base.py:95-119): [roll, pitch, ang vel 3, motor pos 10, motor vel 10, motor torque 10] plus uniform
// observation noise drawn per entry on every get_obs (base_humanoid_env.py:307-338); lane = observation entry
template <class L>
__device__ void write_obs_h1(HModelRef m, HParamsRef p, L& S, int lane, unsigned genv, unsigned obs_count, float* o,
                             float* o2) {
  for (int e = lane; e < 35; e += L::W_) {   // e = observation entry (35 > 32: the two-envs-per-wave groups take two passes)
    double v;
    if (e < 2) {
      double r, pt;
      quat_roll_pitch(&S.qpos[3], &r, &pt);
      v = e == 0 ? r : pt;
    } else if (e < 5) v = S.qvel[3 + (e - 2)];
    else if (e < 15) v = S.sq[e - 5];
    else if (e < 25) v = S.sv[e - 15];
    else v = S.frc[e - 25] * m.act_d[ADS * (e - 25) + AD_GEAR];
    // observation noise (base_humanoid_env.py:307-338): scale > 0 uniform in [-scale, scale]; scale < 0 Gaussian with standard
    // deviation -scale (Box-Muller on the slots e and 64 + e of the observation stream)
    const double sc = p.obs_noise[e];
    if (sc > 0) v += lhw_rng_uniform(p.seed, genv, LHW_STREAM_OBS, obs_count, e, -sc, sc);
    else if (sc < 0) {
      const double u1 = lhw_rng_u01(p.seed, genv, LHW_STREAM_OBS, obs_count, e), u2 = lhw_rng_u01(p.seed, genv, LHW_STREAM_OBS, obs_count, 64 + e);
      v += -sc * (sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2));
    }
    if (o) o[e] = (float)v;
    if (o2) o2[e] = (float)v;
  }
}

// randomize_dynamics (domain_randomization.py:29-56): leg dof frictionloss / damping, then mass scale and inertial
// offset of pelvis + leg bodies relative to the DEFAULT model.  Writes the LDS copies (used immediately on reset) and
// the per-env record.  slot0 = first RNG slot (0 on reset, 1 in step).
template <class L>
__device__ void randomize_dynamics(HModelRef m, HParamsRef p, L& S, double* prm, int lane, unsigned genv,
                                   unsigned stream, unsigned counter, unsigned slot0) {
  if (lane < p.n_rand_dof) {
    const int d = p.rand_dof[lane];
    const double fl = lhw_rng_uniform(p.seed, genv, stream, counter, slot0 + 2 * lane, 0.0, 2.0);
    const double dm = lhw_rng_uniform(p.seed, genv, stream, counter, slot0 + 2 * lane + 1, 0.02, 2.0);
    S.floss[d] = fl; S.damp[d] = dm;
    prm[P_FLOSS + d] = fl; prm[P_DAMP + d] = dm;
  }
  if (lane < p.n_rand_body) {
    const int b = p.rand_body[lane];
    const unsigned base = slot0 + 20 + 4 * lane;
    const double ms = m.body_d[BDS * b + BD_MASS] * lhw_rng_uniform(p.seed, genv, stream, counter, base, 0.95, 1.05);
    S.bmass[b] = ms; prm[P_MASS + b] = ms;
    for (int a = 0; a < 3; a++) {
      const double ip = m.body_d[BDS * b + BD_IPOS + a] + lhw_rng_uniform(p.seed, genv, stream, counter, base + 1 + a, -0.01, 0.01);
      S.bipos[3 * b + a] = ip; prm[P_IPOS + 3 * b + a] = ip;
    }
  }
}

// mj_objectVelocity(mjOBJ_XBODY): linear velocity of the body-frame origin, world orientation
template <class L>
__device__ __forceinline__ void body_linvel(const L& S, int slot /* 0 root, 1 right foot, 2 left foot */, int b, double* lin) {
  const double* cv = &S.svel[6 * slot];
  double dif[3] = {S.xpos[3 * b] - S.com[0], S.xpos[3 * b + 1] - S.com[1], S.xpos[3 * b + 2] - S.com[2]}, t[3];
  cross3(t, dif, cv);
  lin[0] = cv[3] - t[0]; lin[1] = cv[4] - t[1]; lin[2] = cv[5] - t[2];
}

// Layout of a task's kernels at group width W.  H1 tasks: 16 dofs, per-env model parameters (domain randomisation); stepping
// task: 32 geoms, foot sites; the two-envs-per-wave layouts are sized for the robots' own body counts (JVRC 18, H1 15;
// humanoid_create checks).
template <int TASK, int W>
struct LayoutOf {
  typedef LdsT<W, (TASK == TASK_STAND || TASK == TASK_H1WALK), ((TASK == TASK_STAND || TASK == TASK_H1WALK) ? 16 : 18),
               ((TASK == TASK_STEP || W == 64) ? NG : 16), (W == 64 ? NB : ((TASK == TASK_STAND || TASK == TASK_H1WALK) ? 15 : 18)), TASK == TASK_STEP> type;
};

template <int MODE, int TASK, int W>
__device__ __forceinline__ bool store_record(HModelRef m, const HLaunch& lz, const HState& st, typename LayoutOf<TASK, W>::type* SG0, const int env,
                                             const long long t_launch);

// One control step (MODE 0), reset (1), set_state (2) or get_state (3) of env `env` by the group of W lanes that calls it
// (`lane` = lane within the group, S = the group's LDS working set).  Returns true iff the env exceeded the contact capacity
// of the two-envs-per-wave layout before anything of this control step was written: the caller repeats the step with W = 64.
template <int MODE, int TASK, int W>
__device__ __forceinline__ bool control_step(HModelRef m, HParamsRef p, const HLaunch& lz, const HState& st, typename LayoutOf<TASK, W>::type* SG0,
                                             typename LayoutOf<TASK, W>::type& S, const int env,
                                             const int lane, const float* __restrict__ act, float* __restrict__ obs, float* __restrict__ term_obs,
                                             float* __restrict__ rew, unsigned char* __restrict__ done_out, float* __restrict__ rew_terms,
                                             double* __restrict__ xq, double* __restrict__ xv) {
  using L = typename LayoutOf<TASK, W>::type;
  double* rec = st.rec + (size_t)env * REC_D;
  double* prm = st.prm ? st.prm + (size_t)env * PRM_D : nullptr;
  double* ter = (TASK == TASK_STEP) ? st.ter + (size_t)env * TER_D : nullptr;
  // WALKT: WalkingTask (jvrc_walk, h1_walk); H1R: H1 robot state with observation noise + domain randomisation (h1, h1_walk)
  constexpr bool WALKT = TASK == TASK_WALK || TASK == TASK_H1WALK, H1R = TASK == TASK_STAND || TASK == TASK_H1WALK;
  constexpr unsigned WS = TASK == TASK_H1WALK ? 100u : 0u;   // RNG slot base of the walking-task draws (the H1 randomisation owns slots 0..99)
  const int OBS = TASK == TASK_WALK ? 37 : (TASK == TASK_STEP ? 39 : (TASK == TASK_H1WALK ? 43 : 35));
  int* irec = st.irec + (size_t)env * REC_I;
  const unsigned genv = p.env_id_base + env;
  long long* sprof = (env == 0) ? st.prof : nullptr;
  long long* st_prof = sprof;
  PROF_BEGIN();
  const long long t_launch = (MODE == 0 && st.wave_cyc) ? (long long)clock64() : 0;
  if (MODE == 3) {
    if (lane < m.nq) xq[(size_t)env * m.nq + lane] = rec[R_QPOS + lane];
    if (lane < NV) xv[(size_t)env * NV + lane] = rec[R_QVEL + lane];
    return false;
  }
  // ---- load the persistent record (lane-strided); the episode / task context goes to its LDS home
  if (lane < m.nq) S.qpos[lane] = rec[R_QPOS + lane];
  if (lane < NV) { S.qvel[lane] = rec[R_QVEL + lane]; S.qacc[lane] = rec[R_WARM + lane]; }   // qacc of the last pass = warm start

  if (lane < m.nu) {
    S.sq[lane] = rec[R_SQ + lane]; S.sv[lane] = rec[R_SV + lane]; S.frc[lane] = rec[R_FRC + lane];
    S.ctrl[lane] = 0;
  }
  if (lane < 3) S.cmode_ref[lane] = rec[R_MODEREF + lane];
  if (lane == 3) S.cep_ret = rec[R_EPRET];
  if (lane < 12) {
    // CI_* order: phase mode traj started stepcnt resetcnt obscnt t1 t2 reached frames nseq
    const int src = lane == CI_PHASE ? RI_PHASE : lane == CI_MODE ? RI_MODE : lane == CI_TRAJ ? RI_TRAJ : lane == CI_STARTED ? RI_STARTED
                  : lane == CI_STEPCNT ? RI_STEPCNT : lane == CI_RESETCNT ? RI_RESETCNT : lane == CI_OBSCNT ? RI_OBSCNT : lane == CI_T1 ? RI_T1
                  : lane == CI_T2 ? RI_T2 : lane == CI_REACHED ? RI_REACHED : lane == CI_FRAMES ? RI_FRAMES : RI_NSEQ;
    int v = irec[src];
    if (TASK != TASK_STEP && lane >= CI_T1) v = lane == CI_NSEQ ? 2 : 0;
    S.ci[lane] = v;
  }
  // per-env model parameters (or the shared defaults) -> LDS, once per launch
  if constexpr (L::PRM_) {
    if (lane < NV) {
      S.damp[lane] = prm ? prm[P_DAMP + lane] : m.dof_d[DDS * lane + DD_DAMPING];
      S.floss[lane] = prm ? prm[P_FLOSS + lane] : m.dof_d[DDS * lane + DD_FLOSS];
    }
    if (lane < m.nbody) {
      S.bmass[lane] = prm ? prm[P_MASS + lane] : m.body_d[BDS * lane + BD_MASS];
      for (int a = 0; a < 3; a++) S.bipos[3 * lane + a] = prm ? prm[P_IPOS + 3 * lane + a] : m.body_d[BDS * lane + BD_IPOS + a];
    }
    if (lane < 12) S.xfrc[lane] = prm ? prm[P_XFRC + lane] : 0.0;
  }
  SYNC();
  if (lane == 0) { S.overflow = 0; S.env_id = env; if constexpr (L::STEP_) S.nbig = 0; }
  SYNC();

  // One loop, one sub-step call site.  Each env walks through its stages -- the frame_skip control sub-steps, then (if
  // the episode ended) the reset's forward pass and three settle steps -- and the two envs of a wave share every
  // sub-step they both still need; an env that is done simply leaves the loop (SIMT divergence at group granularity).
  enum { ST_CONTROL = 0, ST_RESET, ST_SETTLE, ST_FORWARD, ST_LAST, ST_END };
  int stage = MODE == 0 ? ST_CONTROL : (MODE == 1 ? ST_RESET : ST_FORWARD), kstep = 0;
  bool committed = false;   // outputs / episode statistics of this control step have been written
  if (MODE == 2) {
    if (lane < m.nq) S.qpos[lane] = xq[(size_t)env * m.nq + lane];
    if (lane < NV) S.qvel[lane] = xv[(size_t)env * NV + lane];
    SYNC();
  }
  // ---- BaseHumanoidEnv.step: smoothing, offsets (base_humanoid_env.py:209-215); RobotBase.step (robot_base.py:64-98)
  double target = 0, a_in = 0;   // (the raw action stays in a register: its LDS staging is overwritten by the first sub-step)
  if (MODE == 0 && lane < m.nu) {
    a_in = (double)act[(size_t)env * m.nu + lane];
    target = p.action_smoothing * a_in + (1 - p.action_smoothing) * rec[R_PREVPRED + lane] + p.action_offset[lane];
  }
  for (;;) {
    FRESH_GROUP(W, SG0);   // (shadows the parameters inside the stage loop: nothing lane-derived is carried across a sub-step)
    int flags = 3;
    if (stage == ST_CONTROL) {
      if (kstep < p.frame_skip) {
        if (lane < m.nu) {
          // step_pd on the transmission fields of the previous forward pass (note S); ctrl = tau / gear
          const double tau = p.kp[lane] * (target - S.sq[lane]) + p.kd[lane] * (0.0 - S.sv[lane]);
          S.ctrl[lane] = qdiv(tau, m.act_d[ADS * (lane) + AD_GEAR]);
        }
        SYNC();
        kstep++;
      } else {
        bool do_reset = false;
        CTX_LOAD();
        double a_raw = 0, prevact = 0, prevtq = 0, prevpred = 0;
        if (lane < m.nu) {
          a_raw = a_in;
          prevact = rec[R_PREVACT + lane]; prevtq = rec[R_PREVTQ + lane];
          // prev_action / prev_torque are initialised once, on the first step ever (robot_base.py:82-85), from the fields of
          // the forward pass that preceded this control step (the record still holds them)
          if (!started) { prevact = target; prevtq = rec[R_FRC + lane] * m.act_d[ADS * (lane) + AD_GEAR]; }
        }
        PROF_MARK(9);  // control-step prologue (load, PD) is folded into slot 9 with the sub-step loop overheads
        double r_sum = 0, terms[10], cur_tq = 0;
        bool terminated = false;
        if (lane < m.nu) cur_tq = S.frc[lane] * m.act_d[ADS * (lane) + AD_GEAR];
        // self-collision scan of the contacts of the last forward pass (robot_interface.py:472-484)
        bool self_collision;
        {
          int selfcol = 0;
          if (lane < S.ncon) {
            const int b1 = m.geom_i[GIS * (S.con_g1[lane]) + GI_BODY], b2 = m.geom_i[GIS * (S.con_g2[lane]) + GI_BODY];
            selfcol = (m.body_i[BIS * (b1) + BI_ROOT] == p.root_body && m.body_i[BIS * (b2) + BI_ROOT] == p.root_body) ? 1 : 0;
          }
          self_collision = gany<W>(selfcol);
        }
        // ground reaction forces and lowest foot-floor contact point (robot_interface.py:269-325): lane = contact
        double grf_r = 0, grf_l = 0, cz = 1e300;
        if (TASK != TASK_STAND) {
          int anyfoot = 0;
          if (lane < S.ncon) {
            const int c = lane, b1 = m.geom_i[GIS * (S.con_g1[c]) + GI_BODY], b2 = m.geom_i[GIS * (S.con_g2[c]) + GI_BODY];
            const bool floor1 = m.body_i[BIS * (b1) + BI_ROOT] != p.root_body;
            double fn = 0;
            const int r0 = 4 * c;   // rows of contact c
            if (S.con_dim[c] != 0) {
              if (S.con_dim[c] == 3) {
                const double f0 = S.efc_force[r0], f1 = S.efc_force[r0 + 1], f2 = S.efc_force[r0 + 2], f3 = S.efc_force[r0 + 3], mu = S.con_mu[c];
                const double n = f0 + f1 + f2 + f3, t1f = mu * (f0 - f1), t2f = mu * (f2 - f3);
                fn = sqrt(n * n + t1f * t1f + t2f * t2f);
              } else fn = fabs(S.efc_force[r0]);
            }
            if constexpr (TASK == TASK_STEP) {
              // (a merged contact carries the sum of its copies' forces; the share of the copies that are floor contacts of the foot)
              const double wr = S.con_wr[c], wl = S.con_wl[c];
              if (wr > 0) { grf_r = wr * fn; cz = S.con_pos[3 * c + 2]; anyfoot = 1; }
              if (wl > 0) { grf_l = wl * fn; cz = S.con_pos[3 * c + 2]; anyfoot = 1; }
            } else {
              if (floor1 && b2 == p.rfoot_body) { grf_r = fn; cz = S.con_pos[3 * c + 2]; anyfoot = 1; }
              if (floor1 && b2 == p.lfoot_body) { grf_l = fn; cz = S.con_pos[3 * c + 2]; anyfoot = 1; }
            }
          }
          grf_r = gsum<W>(grf_r); grf_l = gsum<W>(grf_l); cz = gmin<W>(cz);
          if (!gany<W>(anyfoot)) cz = 0;
        }
        if constexpr (TASK == TASK_STEP && W == 64) {
          if (S.nbig) {   // the last forward pass took the many-contact path: its contacts are in HBM, newton_big left these
            self_collision = S.big_selfcol != 0;
            grf_r = S.big_grf_r; grf_l = S.big_grf_l; cz = S.big_cz;
          }
        }
        if (WALKT) {
        // ---- WalkingTask.step (walking_task.py:149-170)
        phase += 1;
        if (phase >= p.period) phase = 0;
        {
          const bool dbl = p.clock_lut[0 * p.period + phase] == 1.0 && p.clock_lut[2 * p.period + phase] == 1.0;
          if (lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, step_count, WS + 0, 100) == 0 && dbl) {
            if (mode == MODE_INPLACE) mode = MODE_STANDING;
            else if (mode == MODE_STANDING) mode = MODE_INPLACE;
            sample_ref(p, genv, LHW_STREAM_STEP, step_count, WS + 1, mode, mode_ref);
          }
          if (lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, step_count, WS + 4, 200) == 0 && mode != MODE_STANDING) {
            if (mode == MODE_FORWARD) mode = MODE_INPLACE;
            else if (mode == MODE_INPLACE) mode = MODE_FORWARD;
            sample_ref(p, genv, LHW_STREAM_STEP, step_count, WS + 5, mode, mode_ref);
          }
          if (!H1R) step_count++;   // h1_walk: the counter advances after the post-observation randomisation draws below
        }
        // ---- calc_reward (walking_task.py:85-147) on the fields of the last forward pass
        // joint-space sums: lane = actuator / dof
        double s_posture = 0, s_tq = 0, s_act = 0, s_rootacc = 0;
        if (lane < m.nu) {
          const double dq = p.neutral_pose[lane] - S.sq[lane];
          s_posture = dq * dq;
          s_tq = fabs(prevtq - cur_tq);
          s_act = fabs(prevact - target);
        }
        if (lane >= 3 && lane < 6) s_rootacc = fabs(S.qvel[lane]);
        if (lane < 3) s_rootacc = fabs(S.qacc[lane]);
        s_posture = gsum<W>(s_posture); s_tq = gsum<W>(s_tq); s_act = gsum<W>(s_act); s_rootacc = gsum<W>(s_rootacc);
        {
          double lv[3], rv[3], rl[3], vloc[3];
          body_linvel(S, 2, p.lfoot_body, lv); body_linvel(S, 1, p.rfoot_body, rv); body_linvel(S, 0, p.root_body, rl);
          matT_vec(vloc, S.rootmat, rl);
          double rf = p.clock_lut[0 * p.period + phase], rvc = p.clock_lut[1 * p.period + phase];
          double lf = p.clock_lut[2 * p.period + phase], lvc = p.clock_lut[3 * p.period + phase];
          if (mode == MODE_STANDING) { rf = 1; lf = 1; rvc = -1; lvc = -1; }
          double yaw_ref = mode_ref[0], vx = mode_ref[1], vy = mode_ref[2];
          if (mode == MODE_STANDING) { yaw_ref = 0; vx = 0; vy = 0; }
          else if (mode == MODE_INPLACE) { vx = 0; vy = 0; }
          else yaw_ref = 0;
          const double gs = sqrt(vx * vx + vy * vy);
          const double PI4 = 3.141592653589793 / 4;
          const double maxf = m.totalmass * 9.8 * 0.5;
          const double nl = fmin(grf_l, maxf) / maxf * 2 - 1, nr = fmin(grf_r, maxf) / maxf * 2 - 1;
          terms[0] = 0.225 * ((tan(PI4 * lf * nl) + tan(PI4 * rf * nr)) / 2);
          const double nlv = fmin(sqrt(dot3(lv, lv)), 0.2) / 0.2 * 2 - 1, nrv = fmin(sqrt(dot3(rv, rv)), 0.2) / 0.2 * 2 - 1;
          terms[1] = 0.225 * ((tan(PI4 * lvc * nlv) + tan(PI4 * rvc * nrv)) / 2);
          terms[2] = 0.050 * exp(-0.25 * s_rootacc);
          double herr = fabs(S.xpos[3 * p.root_body + 2] - cz - p.goal_height);
          if (herr < 0.01 + 0.05 * gs) herr = 0;
          terms[3] = 0.050 * exp(-40 * herr * herr);
          const double ex = vloc[0] - vx, ey = vloc[1] - vy, en = sqrt(ex * ex + ey * ey);
          terms[4] = 0.150 * exp(-10 * (en * en));
          const double ye = fabs(S.qvel[5] - yaw_ref);
          terms[5] = 0.150 * exp(-10 * (ye * ye * ye));
          const double hx = S.xpos[3 * p.head_body] - S.xpos[3 * p.root_body], hy = S.xpos[3 * p.head_body + 1] - S.xpos[3 * p.root_body + 1];
          terms[6] = 0.050 * exp(-10 * sqrt(hx * hx + hy * hy));
          terms[7] = 0.050 * exp(-sqrt(s_posture));
          terms[8] = 0.025 * exp(-0.25 * (s_tq / (double)m.nu));
          terms[9] = 0.025 * exp(-5 * s_act / (double)m.nu);
          for (int k = 0; k < 10; k++) r_sum += terms[k];  // python sum() over the dict, left to right
        }
          const double z = S.qpos[2];
          terminated = z < 0.6 || z > 1.4 || self_collision;  // walking_task.py:184-192
        } else if (TASK == TASK_STEP) {
          // ---- SteppingTask.step (stepping_task.py:211-243) on the stale site / body frames
          phase += 1;
          if (phase >= p.period) phase = 0;
          const double lp[3] = {S.spos[6], S.spos[7], S.spos[8]}, rp[3] = {S.spos[3], S.spos[4], S.spos[5]};
          const double rootp[3] = {S.xpos[3 * p.root_body], S.xpos[3 * p.root_body + 1], S.xpos[3 * p.root_body + 2]};
          {
            const double* tg = ter + T_SEQ + 6 * t1;
            const double dl = sqrt((lp[0] - tg[0]) * (lp[0] - tg[0]) + (lp[1] - tg[1]) * (lp[1] - tg[1]) + (lp[2] - tg[2]) * (lp[2] - tg[2]));
            const double dr = sqrt((rp[0] - tg[0]) * (rp[0] - tg[0]) + (rp[1] - tg[1]) * (rp[1] - tg[1]) + (rp[2] - tg[2]) * (rp[2] - tg[2]));
            if (dl < p.target_radius || dr < p.target_radius) { reached = 1; frames += 1; }
            else { reached = 0; frames = 0; }
            if (reached && frames >= p.delay_frames) {  // update_target_steps
              t1 = t2; t2 += 1;
              if (t2 == nseq) t2 = nseq - 1;
              reached = 0; frames = 0;
            }
          }
          // update_goal_steps (stepping_task.py:184-202): the two targets in the root frame
          if (mode != WALK_STANDING) {
            for (int i = 0; i < 2; i++) {
              const double* sq = ter + T_SEQ + 6 * (i ? t2 : t1);
              const double dvec[3] = {sq[0] - rootp[0], sq[1] - rootp[1], sq[2] - rootp[2]};
              double rel[3];
              matT_vec(rel, S.rootmat, dvec);
              const double M00 = S.rootmat[0] * sq[4] + S.rootmat[3] * sq[5], M10 = S.rootmat[1] * sq[4] + S.rootmat[4] * sq[5];
              const double cy = sqrt(M00 * M00 + M10 * M10);
              goal[i] = rel[0]; goal[2 + i] = rel[1]; goal[4 + i] = rel[2];
              goal[6 + i] = cy > 4.0 * 2.220446049250313e-16 ? atan2(M10, M00) : 0.0;
            }
          }
          step_count++;
          // ---- calc_reward (stepping_task.py:81-123)
          {
            const double* tg = ter + T_SEQ + 6 * t1;
            const double* tg2 = ter + T_SEQ + 6 * t2;
            double lv[3], rv[3];
            body_linvel(S, 2, p.lfoot_body, lv); body_linvel(S, 1, p.rfoot_body, rv);
            double rf = p.clock_lut[0 * p.period + phase], rvc = p.clock_lut[1 * p.period + phase];
            double lf = p.clock_lut[2 * p.period + phase], lvc = p.clock_lut[3 * p.period + phase];
            if (mode == WALK_STANDING) { rf = 1; lf = 1; rvc = -1; lvc = -1; }
            const double PI4 = 3.141592653589793 / 4;
            const double maxf = m.totalmass * 9.8 * 0.5;
            const double nl = fmin(grf_l, maxf) / maxf * 2 - 1, nr = fmin(grf_r, maxf) / maxf * 2 - 1;
            terms[0] = 0.150 * ((tan(PI4 * lf * nl) + tan(PI4 * rf * nr)) / 2);
            const double nlv = fmin(sqrt(dot3(lv, lv)), 0.2) / 0.2 * 2 - 1, nrv = fmin(sqrt(dot3(rv, rv)), 0.2) / 0.2 * 2 - 1;
            terms[1] = 0.150 * ((tan(PI4 * lvc * nlv) + tan(PI4 * rvc * nrv)) / 2);
            const double inner = cos(0.5 * tg[3]) * S.rootquat[0] + sin(0.5 * tg[3]) * S.rootquat[3];
            terms[2] = 0.050 * exp(-(10 * (1 - inner * inner)));
            double herr = fabs(rootp[2] - cz - p.goal_height);
            if (herr < 0.01) herr = 0;
            terms[3] = 0.050 * exp(-40 * herr * herr);
            const double dl = sqrt((lp[0] - tg[0]) * (lp[0] - tg[0]) + (lp[1] - tg[1]) * (lp[1] - tg[1]) + (lp[2] - tg[2]) * (lp[2] - tg[2]));
            const double dr = sqrt((rp[0] - tg[0]) * (rp[0] - tg[0]) + (rp[1] - tg[1]) * (rp[1] - tg[1]) + (rp[2] - tg[2]) * (rp[2] - tg[2]));
            const double hit = reached ? exp(-fmin(dl, dr) / 0.25) : 0.0;
            const double mx = (tg[0] + tg2[0]) / 2, my = (tg[1] + tg2[1]) / 2;
            const double progress = exp(-sqrt((rootp[0] - mx) * (rootp[0] - mx) + (rootp[1] - my) * (rootp[1] - my)) / 2);
            terms[4] = 0.450 * (0.8 * hit + 0.2 * progress);
            const double hx = S.xpos[3 * p.head_body] - rootp[0], hy = S.xpos[3 * p.head_body + 1] - rootp[1], hn = sqrt(hx * hx + hy * hy);
            terms[5] = 0.050 * exp(-10 * (hn * hn));
            for (int k = 0; k < 6; k++) r_sum += terms[k];
            for (int k = 6; k < 10; k++) terms[k] = 0;
          }
          terminated = (rootp[2] - fmin(lp[2], rp[2])) < 0.6 || self_collision;  // stepping_task.py:247-259
        } else {
          // ---- StandingTask.calc_reward / done (standing_task.py:49-131) on the fields of the last forward pass
          double s_posture = 0, s_tau = 0;
          if (lane < m.nu) {
            const double dq = S.sq[lane] - p.neutral_pose[lane];
            s_posture = dq * dq;
            s_tau = cur_tq * cur_tq;
          }
          s_posture = gsum<W>(s_posture); s_tau = gsum<W>(s_tau);
          double rl[3], vloc[3], dh[3], hloc[3];
          body_linvel(S, 0, p.root_body, rl);
          matT_vec(vloc, S.rootmat, rl);
          for (int a = 0; a < 3; a++) dh[a] = S.xpos[3 * p.head_body + a] - S.xpos[3 * p.root_body + a];
          matT_vec(hloc, S.rootmat, dh);      // torso position in the pelvis frame: inv(root_pose) . head_pose
          const double fwd = sqrt(vloc[0] * vloc[0] + vloc[1] * vloc[1]), yaw = fabs(S.qvel[5]);
          const double herr = fabs(S.xpos[3 * p.root_body + 2] - p.goal_height);
          const double uerr = sqrt(hloc[0] * hloc[0] + hloc[1] * hloc[1]);
          terms[0] = 0.3 * exp(-4 * (fwd * fwd));
          terms[1] = 0.3 * exp(-4 * (yaw * yaw));
          terms[2] = 0.1 * exp(-0.5 * (herr * herr));
          terms[3] = 0.1 * exp(-40 * (uerr * uerr));
          terms[4] = 0.1 * exp(-5e-5 * s_tau);
          terms[5] = 0.1 * exp(-1 * s_posture);
          for (int k = 0; k < 6; k++) r_sum += terms[k];
          for (int k = 6; k < 10; k++) terms[k] = 0;
          const double z = S.qpos[2];
          terminated = z < 0.9 || z > 1.4 || self_collision;  // standing_task.py:111-131
        }
        if (st.tin) {   // the batched sim facade (include/lhw.h: LhwTaskInput)
          double* ti = st.tin + lz.tin_off + (size_t)env * LHW_TASK_INPUT_DIM;
          double rv3[3], lv3[3], rl3[3], vl3[3];
          body_linvel(S, 1, p.rfoot_body, rv3); body_linvel(S, 2, p.lfoot_body, lv3); body_linvel(S, 0, p.root_body, rl3);
          matT_vec(vl3, S.rootmat, rl3);
          if (lane == 0) {
            ti[LHW_TIN_GRF_R] = grf_r; ti[LHW_TIN_GRF_L] = grf_l; ti[LHW_TIN_CONTACT_Z] = cz < 1e299 ? cz : 0.0;
            int anyc = 0;
            for (int c = 0; c < S.ncon; c++) {
              const int b1 = m.geom_i[GIS * (S.con_g1[c]) + GI_BODY], b2 = m.geom_i[GIS * (S.con_g2[c]) + GI_BODY];
              if (m.body_i[BIS * (b1) + BI_ROOT] != p.root_body && (b2 == p.rfoot_body || b2 == p.lfoot_body)) anyc = 1;
              if constexpr (TASK == TASK_STEP) { if (S.con_wr[c] > 0 || S.con_wl[c] > 0) anyc = 1; }
            }
            if constexpr (TASK == TASK_STEP && W == 64) { if (S.nbig) anyc = S.big_anyfoot; }
            ti[LHW_TIN_FOOT_CONTACT] = anyc; ti[LHW_TIN_SELF_COLLISION] = self_collision ? 1.0 : 0.0;
            ti[LHW_TIN_PHASE] = phase; ti[LHW_TIN_MODE] = mode;
            for (int a = 0; a < 3; a++) {
              ti[LHW_TIN_MODE_REF + a] = mode_ref[a];
              ti[LHW_TIN_RFOOT_VEL + a] = rv3[a]; ti[LHW_TIN_LFOOT_VEL + a] = lv3[a]; ti[LHW_TIN_ROOT_VEL_LOCAL + a] = vl3[a];
              ti[LHW_TIN_ROOT_XPOS + a] = S.xpos[3 * p.root_body + a]; ti[LHW_TIN_HEAD_XPOS + a] = S.xpos[3 * p.head_body + a];
              ti[LHW_TIN_RFOOT_XPOS + a] = S.xpos[3 * p.rfoot_body + a]; ti[LHW_TIN_LFOOT_XPOS + a] = S.xpos[3 * p.lfoot_body + a];
            }
          }
          if (lane < 9) ti[LHW_TIN_ROOT_XMAT + lane] = S.rootmat[lane];
          if (lane < m.nq) ti[LHW_TIN_QPOS + lane] = S.qpos[lane];
          if (lane < NV) { ti[LHW_TIN_QVEL + lane] = S.qvel[lane]; ti[LHW_TIN_QACC + lane] = S.qacc[lane]; }
          if (lane < m.nu) {
            ti[LHW_TIN_ACT_POS + lane] = S.sq[lane]; ti[LHW_TIN_ACT_VEL + lane] = S.sv[lane]; ti[LHW_TIN_ACT_TAU + lane] = cur_tq;
            ti[LHW_TIN_PREV_TORQUE + lane] = prevtq; ti[LHW_TIN_PREV_ACTION + lane] = prevact; ti[LHW_TIN_ACTION + lane] = target;
          }
        }
        // failure detection: a non-finite state ends the episode (counted in ep_stats[4]); outputs are sanitised so one
        // diverged env cannot poison the batch (the reference has no equivalent: a NaN there propagates into the buffers)
        {
          bool bad = false;
          if (lane < m.nq) bad = !isfinite(S.qpos[lane]);
          if (lane < NV) bad = bad || !isfinite(S.qvel[lane]) || !isfinite(S.qacc[lane]);
          if (gany<W>(bad) || !isfinite(r_sum)) {
            terminated = true;
            r_sum = 0;
            for (int k = 0; k < 10; k++) terms[k] = 0;
            if (lane < m.nq) S.qpos[lane] = p.nominal_qpos[lane];
            if (lane < NV) { S.qvel[lane] = 0; S.qacc[lane] = 0; }
            if (lane < m.nu) { S.sq[lane] = p.action_offset[lane]; S.sv[lane] = 0; S.frc[lane] = 0; cur_tq = 0; }
            if (lane == 0) atomicAdd(&st.ep_stats[4], 1.0);
            SYNC();
          }
        }
        prevact = target;
        prevtq = cur_tq;
        prevpred = a_raw;
        started = 1;
        traj_len += 1;
        ep_ret += r_sum;
        const bool truncated = p.max_traj_len > 0 && traj_len >= p.max_traj_len;
        const int NT = WALKT ? 10 : 6;
        // the observation is formed in LDS and copied out
        SYNC();
        if (TASK == TASK_WALK) write_obs(m, p, S, lane, phase, mode, mode_ref, S.obsf());
        else if (TASK == TASK_STEP) write_obs_step(m, p, S, lane, phase, goal, S.obsf());
        else {
          write_obs_h1(m, p, S, lane, genv, obs_count, S.obsf(), nullptr);
          if (TASK == TASK_H1WALK) write_obs_walk_ext(p, lane, phase, mode, mode_ref, S.obsf() + 35);
        }
        SYNC();
        for (int e = lane; e < OBS; e += W) {
          const float ov = S.obsf()[e];
          obs[(size_t)env * OBS + e] = ov;
          if (term_obs) term_obs[(size_t)env * OBS + e] = ov;
        }
        if (TASK != TASK_WALK && TASK != TASK_STEP) {
          obs_count++;
          // post-observation randomisation draws (base_humanoid_env.py:221-225): slot 0 / 70 are the interval triggers
          if (p.dynrand_interval > 0 && lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, step_count, 0, p.dynrand_interval) == 0)
            randomize_dynamics(m, p, S, prm, lane, genv, LHW_STREAM_STEP, step_count, 1);
          if (p.perturb_interval > 0 && lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, step_count, 70, p.perturb_interval) == 0) {
            // apply_perturbation (domain_randomization.py:10-26): per body force / torque, then a coin flip that clears ALL
            double xf[12];
            for (int k = 0; k < 12; k++) xf[k] = S.xfrc[k];
            for (int k = 0; k < p.n_pbody; k++) {
              for (int a = 0; a < 3; a++) {
                xf[6 * k + a] = lhw_rng_uniform(p.seed, genv, LHW_STREAM_STEP, step_count, 71 + 7 * k + a, -p.force_mag, p.force_mag);
                xf[6 * k + 3 + a] = lhw_rng_uniform(p.seed, genv, LHW_STREAM_STEP, step_count, 74 + 7 * k + a, -p.torque_mag, p.torque_mag);
              }
              if (lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, step_count, 77 + 7 * k, 2) == 0)
                for (int j = 0; j < 12; j++) xf[j] = 0;
            }
            SYNC();
            if (lane < 12) { S.xfrc[lane] = xf[lane]; prm[P_XFRC + lane] = xf[lane]; }
          }
          step_count++;
        }
        if constexpr (TASK == TASK_WALK || TASK == TASK_STEP) {
          // apply_perturbation on a JVRC task (LhwEnvConfig.perturb_*; base_humanoid_env.py:224-225 behind get_obs): the draws of the H1 tasks
          // -- STEP stream, slot 70 the trigger, 71.. per body force / torque / coin -- under the counter this control step's task draws used.
          // (the record's address is re-derived here from an opaque copy of the env index: nothing of this block is live in the sub-steps)
          if (p.perturb_interval > 0) {
            const unsigned pc = step_count - 1u;
            if (lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, pc, 70, p.perturb_interval) == 0) {
              double* xr = st.prm + (size_t)opaque_int(env) * PRM_D + P_XFRC;
              double v = lane < 12 ? xr[lane] : 0.0;      // lane = component: body lane / 6, axis lane % 6
#pragma unroll
              for (int k = 0; k < 2; k++) {
                if (k < p.n_pbody) {
                  if (lane >= 6 * k && lane < 6 * k + 6) {
                    const int a = lane - 6 * k;
                    v = a < 3 ? lhw_rng_uniform(p.seed, genv, LHW_STREAM_STEP, pc, 71 + 7 * k + a, -p.force_mag, p.force_mag)
                              : lhw_rng_uniform(p.seed, genv, LHW_STREAM_STEP, pc, 74 + 7 * k + (a - 3), -p.torque_mag, p.torque_mag);
                  }
                  if (lhw_rng_randint(p.seed, genv, LHW_STREAM_STEP, pc, 77 + 7 * k, 2) == 0) v = 0.0;   // the coin clears ALL applied wrenches
                }
              }
              if (lane < 12) xr[lane] = v;
            }
          }
        }
        if (lane == 0) {
          rew[env] = (float)r_sum;
          done_out[env] = (terminated ? 1 : 0) | (truncated ? 2 : 0);
          if (S.overflow) atomicAdd(&st.ep_stats[3], 1.0);
          if (rew_terms) for (int k = 0; k < NT; k++) rew_terms[(size_t)env * NT + k] = (float)terms[k];
        }
        if (p.max_traj_len > 0 && (terminated || truncated)) {
          if (lane == 0) { atomicAdd(&st.ep_stats[0], ep_ret); atomicAdd(&st.ep_stats[1], (double)traj_len); atomicAdd(&st.ep_stats[2], 1.0); }
          do_reset = true;
        }
        if (lane < m.nu) { rec[R_PREVPRED + lane] = prevpred; rec[R_PREVACT + lane] = prevact; rec[R_PREVTQ + lane] = prevtq; }
        CTX_STORE();
        committed = true;
        stage = do_reset ? ST_RESET : ST_END;
      }
    } else if (stage == ST_SETTLE) {
      if (kstep < 3) kstep++;   // three settle steps, ctrl = 0
      else {
        CTX_LOAD();
        double prevpred = 0;
        if (WALKT) {
          // WalkingTask.reset (walking_task.py:194-205): slot 0 mode, 1..3 mode_ref, 4 phase (+100 for h1_walk)
          const double u = lhw_rng_u01(p.seed, genv, LHW_STREAM_RESET, reset_count, WS + 0);
          mode = u < 0.6 ? MODE_STANDING : (u < 0.8 ? MODE_INPLACE : MODE_FORWARD);
          sample_ref(p, genv, LHW_STREAM_RESET, reset_count, WS + 1, mode, mode_ref);
          phase = lhw_rng_randint(p.seed, genv, LHW_STREAM_RESET, reset_count, WS + 4, p.period);
        }
        if (TASK == TASK_STEP) {
          // ---- SteppingTask.reset (stepping_task.py:261-334); RNG slots: 0 phase, 1 mode, 2 mode-specific choice, 3 first-step
          // offset, 4 number of flat steps.  Lane k builds target step k (running sums are replayed per lane so that every
          // value is produced by the same sequence of additions as in the reference's loops).
          for (int k = 0; k < 8; k++) goal[k] = 0;
          reached = 0; frames = 0;
          phase = lhw_rng_randint(p.seed, genv, LHW_STREAM_RESET, reset_count, 0, 2) == 0 ? 0 : p.period / 2;
          const double u = lhw_rng_u01(p.seed, genv, LHW_STREAM_RESET, reset_count, 1);
          mode = u < 0.15 ? WALK_CURVED : (u < 0.2 ? WALK_STANDING : (u < 0.4 ? WALK_BACKWARD : (u < 0.7 ? WALK_LATERAL : WALK_FORWARD)));
          const int k = lane;
          double sx = 0, sy = 0, sz = 0, sth = 0;
          if (mode == WALK_CURVED) {
            const double* row = p.plans + (size_t)lhw_rng_randint(p.seed, genv, LHW_STREAM_RESET, reset_count, 2, p.nplans) * (1 + MAX_SEQ * 3);
            nseq = (int)row[0];
            if (k < nseq) { sx = row[1 + 3 * k]; sy = row[2 + 3 * k]; sth = row[3 + 3 * k]; }
          } else if (mode == WALK_LATERAL) {
            const double sgn = lhw_rng_randint(p.seed, genv, LHW_STREAM_RESET, reset_count, 2, 2) == 0 ? -1.0 : 1.0;
            nseq = 19;
            double y = 0;
            for (int i = 1; i <= k + 1 && i < 20; i++) {
              if (i % 2) y += 0.4; else y -= (2.0 / 3.0) * 0.4;
            }
            sy = sgn * y;
          } else {
            const int num_steps = mode == WALK_STANDING ? 1 : 20;
            const double size = mode == WALK_BACKWARD ? -0.1 : 0.3, gap = 0.15;
            double height = 0;
            if (mode == WALK_FORWARD) {
              const double hh = fmin(1.0, fmax(0.0, ((double)lz.iteration - 3000.0) / 8000.0)) * 0.1;
              height = lhw_rng_randint(p.seed, genv, LHW_STREAM_RESET, reset_count, 2, 2) == 0 ? -hh : hh;
            }
            const double uf = lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 3, 0.095, 0.105);
            const bool neg = (double)phase == 0.5 * (double)p.period;
            const int cflat = 2 + (int)lhw_rng_randint(p.seed, genv, LHW_STREAM_RESET, reset_count, 4, 2);
            nseq = num_steps == 1 ? 2 : 20;
            if (k == 0) sy = neg ? -1 * uf : 1 * uf;
            else {
              double x = 0, y = neg ? -gap : gap, z = 0;
              const int last = (k >= nseq - 1) ? num_steps - 2 : k;   // the final step replays the whole loop
              for (int i = 1; i <= last; i++) {
                x += size; y *= -1;
                if (i > cflat) z += height;
              }
              if (k >= nseq - 1) { sx = x + size; sy = -y; sz = z; }
              else { sx = x; sy = y; sz = z; }
            }
          }
          // transform_sequence (stepping_task.py:125-138): relative to the feet mid-point and the root yaw (stale frames)
          const double mid0 = (S.xpos[3 * p.lfoot_body] + S.xpos[3 * p.rfoot_body]) / 2, mid1 = (S.xpos[3 * p.lfoot_body + 1] + S.xpos[3 * p.rfoot_body + 1]) / 2;
          double yaw;
          {
            const double w = S.rootquat[0], x = S.rootquat[1], y = S.rootquat[2], z = S.rootquat[3];
            const double Nq = w * w + x * x + y * y + z * z, sc = Nq > 2.220446049250313e-16 ? 2.0 / Nq : 0.0;
            const double Y = y * sc, Z = z * sc;
            const double M00 = 1.0 - (y * Y + z * Z), M10 = x * Y + w * Z, cy = sqrt(M00 * M00 + M10 * M10);
            yaw = cy > 4.0 * 2.220446049250313e-16 ? atan2(M10, M00) : 0.0;
          }
          if (k < MAX_SEQ) {
            double out[6] = {0.0, 0.0, -1.0, 0.0, 1.0, 0.0};
            if (k < nseq) {
              const double cyw = cos(yaw), syw = sin(yaw);
              out[0] = mid0 + sx * cyw - sy * syw; out[1] = mid1 + sx * syw + sy * cyw; out[2] = sz; out[3] = yaw + sth;
              out[4] = cos(out[3]); out[5] = sin(out[3]);
            }
            for (int a = 0; a < 6; a++) ter[T_SEQ + 6 * k + a] = out[a];
          }
          if (lane == 0) ter[T_FLOOR] = mode == WALK_FORWARD ? -2.0 : 0.0;   // stepping_task.py:330-334
          t1 = 0; t2 = 1;                                                   // update_target_steps from t1 = t2 = 0
          if (t2 == nseq) t2 = nseq - 1;
        }
        reset_count++;
        traj_len = 0;
        ep_ret = 0;
        prevpred = 0;
        SYNC();
        if (TASK == TASK_WALK) write_obs(m, p, S, lane, phase, mode, mode_ref, S.obsf());
        else if (TASK == TASK_STEP) write_obs_step(m, p, S, lane, phase, goal, S.obsf());
        else {
          write_obs_h1(m, p, S, lane, genv, obs_count, S.obsf(), nullptr);  // the counter advances whether or not the caller wants the observation
          if (TASK == TASK_H1WALK) write_obs_walk_ext(p, lane, phase, mode, mode_ref, S.obsf() + 35);
          obs_count++;
        }
        SYNC();
        if (obs) for (int e = lane; e < OBS; e += W) obs[(size_t)env * OBS + e] = S.obsf()[e];
        if (lane < m.nu) rec[R_PREVPRED + lane] = prevpred;
        CTX_STORE();
        stage = ST_END;
      }
    }
    if (TASK == TASK_WALK && MODE == 0 && stage == ST_RESET && p.reset_template >= 0) {
      // jvrc_walk: the physical state of a freshly reset env -- nominal pose, zero velocity, one forward pass, three settle
      // steps (base_humanoid_env.py:247-276) -- does not depend on the env or on any random draw, so it was computed once,
      // by this same code, for the template record at creation; an episode end copies it instead of holding its wave (and,
      // with every wave of the batch resident at once, the whole launch) for four more sub-steps.
      const double* tr = st.rec + (size_t)p.reset_template * REC_D;
      SYNC();
      if (lane < m.nq) S.qpos[lane] = tr[R_QPOS + lane];
      if (lane < NV) { S.qvel[lane] = tr[R_QVEL + lane]; S.qacc[lane] = tr[R_WARM + lane]; }
      if (lane < m.nu) { S.sq[lane] = tr[R_SQ + lane]; S.sv[lane] = tr[R_SV + lane]; S.frc[lane] = tr[R_FRC + lane]; S.ctrl[lane] = 0; }
      SYNC();
      stage = ST_SETTLE; kstep = 3;
      continue;
    }
    if (stage == ST_RESET) {   // (an env whose episode just ended enters here in the same pass)
      const unsigned reset_count = (unsigned)S.ci[CI_RESETCNT];
      // ---- MujocoEnv.reset + BaseHumanoidEnv.reset_model (mujoco_env.py:113-127, base_humanoid_env.py:247-276)
      SYNC();
      if (lane < m.nq) S.qpos[lane] = p.nominal_qpos[lane];
      if (lane < NV) S.qvel[lane] = 0;
      if (lane < m.nu) S.ctrl[lane] = 0;
      if (lane < NV) S.qacc[lane] = 0;   // mj_resetData clears qacc_warmstart
      if (H1R) {
        // mj_resetData clears xfrc_applied; dynamics randomisation on reset (base_humanoid_env.py:254-255), slots 0..63
        if (lane < 12) { S.xfrc[lane] = 0; prm[P_XFRC + lane] = 0; }
        if (p.dynrand_interval > 0) randomize_dynamics(m, p, S, prm, lane, genv, LHW_STREAM_RESET, reset_count, 0);
      } else if (p.perturb_interval > 0) {   // JVRC task with perturbations: mj_resetData clears xfrc_applied
        if (lane < 12) st.prm[(size_t)opaque_int(env) * PRM_D + P_XFRC + lane] = 0;
      }
      SYNC();   // (the nominal pose is in place before lane 0 overwrites the root's part of it)
      if (p.init_noise > 0) {  // _apply_init_noise (base_humanoid_env.py:278-305), any task: slot 64 root z, 65/66 roll/pitch, 67.. joints
        const double cn = p.init_noise;
        if (lane == 0) {
          const double z0 = p.nominal_qpos[2];
          S.qpos[2] = lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 64, z0, z0 + 0.02);
          const double ai = 0.5 * lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 65, -cn, cn);
          const double aj = 0.5 * lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 66, -cn, cn);
          const double ci = cos(ai), si = sin(ai), cj = cos(aj), sj = sin(aj);   // euler2quat(ai, aj, 0), static xyz
          S.qpos[3] = cj * ci; S.qpos[4] = cj * si; S.qpos[5] = sj * ci; S.qpos[6] = -sj * si;
        }
        if (lane >= 7 && lane < m.nq) S.qpos[lane] += lhw_rng_uniform(p.seed, genv, LHW_STREAM_RESET, reset_count, 67 + (lane - 7), -cn, cn);
      }
      SYNC();
      flags = 0;   // set_state: forward pass with actuation disabled
      stage = ST_SETTLE; kstep = 0;
    } else if (stage == ST_FORWARD) {
      flags = 0;   // lhw_env_set_state: mj_forward with actuation disabled
      stage = ST_LAST;
    }
    if (stage == ST_END) break;
    // (Every phase of the sub-step takes its lane index from fresh_wave_lane(): besides keeping the addresses built from it out of
    // scratch, the opaque value keeps the ~100 model-table loads of a sub-step -- indexed by the lane, invariant across the 25
    // sub-steps -- inside the loop.  Hoisted, they would be parked in scratch and reloaded from there, FETCH_SIZE 708 MB per launch.)
    substep<TASK == TASK_STEP>(m, p, SG0, flags, sprof, (gtab_d)ter,
                               (TASK == TASK_STEP && W == 64 && st.bigd) ? (gws_d)(st.bigd + (size_t)env * BW_DOUBLES) : (gws_d) nullptr,
                               (TASK == TASK_STEP && W == 64 && st.bigd) ? (gws_i)(st.bigi + (size_t)env * BW_INTS) : (gws_i) nullptr);
    if (stage == ST_LAST) break;
    // two envs per wave: an env that needs more contacts than this layout holds is handed to the one-env-per-wave kernel
    // untouched (nothing of it has been written yet); once its outputs are out, it can only truncate like that kernel does
    if (W == 32 && MODE == 0 && !committed && S.overflow) {
      if (lane == 0) st.slow[env] = 1;
      return true;
    }
  }
  PROF_MARK(10);
  // ---- store the record
  SYNC();
  return store_record<MODE, TASK, W>(m, lz, st, SG0, env, t_launch);
}

// The persistent record goes back to HBM (lane-strided); everything it needs is re-derived from the env index and a fresh lane.
template <int MODE, int TASK, int W>
__device__ __forceinline__ bool store_record(HModelRef m, const HLaunch& lz, const HState& st, typename LayoutOf<TASK, W>::type* SG0, const int env,
                                             const long long t_launch) {
  using L = typename LayoutOf<TASK, W>::type;
  FRESH_GROUP(W, SG0);
  double* rec = st.rec + (size_t)env * REC_D;
  int* irec = st.irec + (size_t)env * REC_I;
  if (lane < m.nq) rec[R_QPOS + lane] = S.qpos[lane];
  if (lane < NV) { rec[R_QVEL + lane] = S.qvel[lane]; rec[R_WARM + lane] = S.qacc[lane]; }
  if (lane < m.nu) { rec[R_SQ + lane] = S.sq[lane]; rec[R_SV + lane] = S.sv[lane]; rec[R_FRC + lane] = S.frc[lane]; }
  if (lane < 3) rec[R_MODEREF + lane] = S.cmode_ref[lane];
  if (lane == 3) rec[R_EPRET] = S.cep_ret;
  if (lane < (TASK == TASK_STEP ? 12 : 7)) {
    const int dst = lane == CI_PHASE ? RI_PHASE : lane == CI_MODE ? RI_MODE : lane == CI_TRAJ ? RI_TRAJ : lane == CI_STARTED ? RI_STARTED
                  : lane == CI_STEPCNT ? RI_STEPCNT : lane == CI_RESETCNT ? RI_RESETCNT : lane == CI_OBSCNT ? RI_OBSCNT : lane == CI_T1 ? RI_T1
                  : lane == CI_T2 ? RI_T2 : lane == CI_REACHED ? RI_REACHED : lane == CI_FRAMES ? RI_FRAMES : RI_NSEQ;
    irec[dst] = S.ci[lane];
  }
  if (lane == 0 && MODE == 0 && lz.only_flagged) { st.slow[env] = 0; atomicAdd(&st.ep_stats[5], 1.0); }
  if (MODE == 0 && st.wave_cyc && lane == 0) st.wave_cyc[env] = (long long)clock64() - t_launch;
  return false;
}

#ifndef LHW_WAVES_PER_SIMD
#define LHW_WAVES_PER_SIMD 2      // 256 VGPRs per lane; 8 one-wave blocks per CU (the LDS working set allows no more)
#endif
