// Control-step kernels (one launch = one control step of a range of envs) and the host side of the humanoid stepper.
// The device code -- layouts, sub-step, control_step -- lives in lhw_humanoid_dev.h (see its header for the design).
#include "lhw_humanoid_dev.h"

template <int MODE, int TASK, int W>  // MODE: 0 step, 1 reset(mask), 2 set_state, 3 get_state; TASK: TASK_WALK / TASK_STAND / TASK_STEP / TASK_H1WALK; W: lanes per env
__global__ void __launch_bounds__(64, LHW_WAVES_PER_SIMD) humanoid_kernel(const HModel* __restrict__ mp, const HParams* __restrict__ pp, HLaunch lz, HState st, const float* __restrict__ act,
                                                      float* __restrict__ obs, float* __restrict__ term_obs,
                                                      float* __restrict__ rew, unsigned char* __restrict__ done_out,
                                                      float* __restrict__ rew_terms, const unsigned char* __restrict__ mask,
                                                      double* __restrict__ xq, double* __restrict__ xv) {
  using L = typename LayoutOf<TASK, W>::type;
  constexpr int G = 64 / W;   // envs per wavefront
  __shared__ L SG[G];
  LHW_LDS_POISON(SG);
  const int lane = threadIdx.x & (W - 1);   // lane within the env's group
  HParamsRef p = *(const HParams LHW_GLOBAL_AS*)pp;
  HModelRef m = *(const HModel LHW_GLOBAL_AS*)mp;   // (a register copy of the whole record was measured: +3.5 %, its ~100 scalars crowd the SGPR file)
  if constexpr (MODE == 0 && W == 64) {
    // One call site for both uses of the one-env-per-wave step kernel.  As the re-run behind the two-envs-per-wave kernel
    // (only_flagged) it is launched with FEW workgroups, each scanning the flags of `chunk` <= 64 consecutive envs and stepping
    // the flagged ones in turn: on most control steps nothing is flagged, and a grid of one workgroup per env would have to
    // wait for wave slots (and 16 KB of LDS each) behind the other rollout group's kernel just to find that out.
    int e0 = (int)blockIdx.x;
    unsigned long long todo = e0 < lz.env_count ? 1ull : 0ull;
    if (lz.only_flagged) {
      const int chunk = (lz.env_count + (int)gridDim.x - 1) / (int)gridDim.x;
      e0 = (int)blockIdx.x * chunk;
      const int t = (int)threadIdx.x;
      todo = __ballot(t < chunk && e0 + t < lz.env_count && st.slow[lz.env_first + e0 + t] != 0);
    }
    while (todo) {
      const int i = __ffsll(todo) - 1;
      todo &= todo - 1;
      control_step<MODE, TASK, W>(m, p, lz, st, SG, SG[0], lz.env_first + e0 + i, lane, act, obs, term_obs, rew, done_out, rew_terms, xq, xv);
    }
  } else {
    const int eidx = blockIdx.x * G + group_id<W>();
    if (eidx >= lz.env_count) return;
    const int env = eidx + lz.env_first;
    if (MODE == 1 && mask && !mask[env]) return;
    control_step<MODE, TASK, W>(m, p, lz, st, SG, SG[group_id<W>()], env, lane, act, obs, term_obs, rew, done_out, rew_terms, xq, xv);
  }
}

// ------------------------------------------------------------------------------------------------ host side
static void h_quat2mat(double* R, const double* q) {
  const double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  const double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02); R[3] = 2 * (q12 + q03);
  R[5] = 2 * (q23 - q01); R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}

template <typename T>
static DevTab<T> to_dev(HumanoidEnv* h, const T* src, size_t n) {
  void* d = nullptr;
  if (lhw_malloc(&d, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess) return DevTab<T>{nullptr};
  if (n && hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return DevTab<T>{nullptr};
  h->dev_allocs.push_back(d);
  return DevTab<T>{(const T*)d};
}

static bool humanoid_upload_params(HumanoidEnv* h) {
  if (!h->p_dev) {
    void* d = nullptr;
    if (lhw_malloc(&d, sizeof(HParams)) != hipSuccess) return false;
    h->dev_allocs.push_back(d);
    h->p_dev = (HParams*)d;
  }
  if (!h->m_dev) {
    void* d = nullptr;
    if (lhw_malloc(&d, sizeof(HModel)) != hipSuccess) return false;
    h->dev_allocs.push_back(d);
    h->m_dev = (HModel*)d;
  }
  return hipMemcpy(h->p_dev, &h->p, sizeof(HParams), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(h->m_dev, &h->m, sizeof(HModel), hipMemcpyHostToDevice) == hipSuccess;
}

int humanoid_create(HumanoidEnv** out, const std::vector<int32_t>& mi, const std::vector<double>& md, const LhwEnvConfig* cfg,
                    int* obs_dim, int* act_dim, int* n_terms) {
  auto IF = [&](int f) { return mi.data() + mi[LHW_IH_COUNT + f]; };
  auto DF = [&](int f) { return md.data() + mi[LHW_IH_COUNT + LHW_IF_COUNT + f]; };
  const int nq = mi[LHW_IH_NQ], nv = mi[LHW_IH_NV], nu = mi[LHW_IH_NU], nbm = mi[LHW_IH_NBODY], nj = mi[LHW_IH_NJNT],
            ng = mi[LHW_IH_NGEOM], np = mi[LHW_IH_NPAIR];
  // Static bodies (children of the world without joints: floor, terrain boxes) never move: their geoms are attached to
  // the world body with the composed pose and the bodies themselves are dropped from the kinematic tables.
  std::vector<int> bmap(nbm, 0), bsrc(1, 0);
  for (int b = 1; b < nbm; b++) {
    if (IF(LHW_IF_BODY_ROOTID)[b] == 1) { bmap[b] = (int)bsrc.size(); bsrc.push_back(b); }
    else if (IF(LHW_IF_BODY_PARENTID)[b] != 0 || IF(LHW_IF_BODY_DOFNUM)[b] > 0)
      return lhw_fail(LHW_ERR_UNSUPPORTED, "exactly one dynamic tree (rooted at body 1) plus static children of the world is supported");
  }
  const int nb = (int)bsrc.size();
  if (nq > NQ || nv > NVMAX || nu > NU || nb > NB || nj > NJ || ng > NG || np > NP || nb > 64 || np > 64)
    return lhw_fail(LHW_ERR_MODEL, "model exceeds compiled limits (nq %d/%d nv %d/%d nu %d/%d nbody %d/%d njnt %d/%d ngeom %d/%d npair %d/%d)%s",
                    nq, NQ, nv, NVMAX, nu, NU, nb, NB, nj, NJ, ng, NG, np, NP,
                    nb > NB ? ": fold the welded (joint-less) links into their parents first -- Model.fuse_static / fit_stepper_limits, or MuJoCo's fusestatic" : "");
  const bool stepping = cfg->task == LHW_TASK_JVRC_STEP, h1walk = cfg->task == LHW_TASK_H1_WALK;
  const bool walk = cfg->task == LHW_TASK_JVRC_WALK || stepping;             // JVRC robot + gait clock
  const bool stand = cfg->task == LHW_TASK_H1_STAND || h1walk;               // H1 robot: observation noise, domain randomisation
  if (!walk && !stand) return lhw_fail(LHW_ERR_ARG, "humanoid stepper: unknown task");
  if (walk && (nu != 12 || nq != 19 || nv != 18)) return lhw_fail(LHW_ERR_UNSUPPORTED, "jvrc tasks need a free root + 12 actuated leg hinges");
  if (stand && (nu != 10 || nq != 17 || nv != 16)) return lhw_fail(LHW_ERR_UNSUPPORTED, "h1 needs a free root + 10 actuated leg hinges");
  if (!cfg->kp || !cfg->kd || !cfg->action_offset || cfg->n_task_iparams < LHW_TI_COUNT || cfg->n_task_params < LHW_TP_COUNT || cfg->frame_skip <= 0)
    return lhw_fail(LHW_ERR_ARG, "humanoid task config incomplete");
  if ((walk || h1walk) && (!cfg->clock_lut || cfg->period <= 0)) return lhw_fail(LHW_ERR_ARG, "walking / stepping tasks need the gait clock table");
  if (stand && (cfg->n_task_params < LHW_TP_H1_OBS_NOISE + 35 || cfg->n_task_iparams < LHW_TI_H1_RAND_BODY + 11))
    return lhw_fail(LHW_ERR_ARG, "h1 task parameter arrays too short");
  int nplans = 0;
  if (stepping) {
    if (cfg->n_task_iparams < LHW_TI_STEP_COUNT || cfg->n_task_params < LHW_TP_STEP_PLANS) return lhw_fail(LHW_ERR_ARG, "stepping task parameter arrays too short");
    nplans = (int)cfg->task_params[LHW_TP_STEP_NPLANS];
    if (nplans < 1 || cfg->n_task_params < LHW_TP_STEP_PLANS + nplans * (1 + LHW_STEP_MAX_SEQ * 3)) return lhw_fail(LHW_ERR_ARG, "stepping task: footstep plan table missing or short");
    for (int q = 0; q < nplans; q++) {
      const double len = cfg->task_params[LHW_TP_STEP_PLANS + (size_t)q * (1 + LHW_STEP_MAX_SEQ * 3)];
      if (!(len >= 2 && len <= LHW_STEP_MAX_SEQ)) return lhw_fail(LHW_ERR_ARG, "stepping task: plan %d has %g steps (2..%d supported)", q, len, LHW_STEP_MAX_SEQ);
    }
    const int b0 = cfg->task_iparams[LHW_TI_STEP_BOX_GEOM0], nbx = cfg->task_iparams[LHW_TI_STEP_NBOX], fg = cfg->task_iparams[LHW_TI_STEP_FLOOR_GEOM];
    if (nbx != MAX_SEQ || b0 < 0 || b0 + nbx > ng || fg < 0 || fg >= ng) return lhw_fail(LHW_ERR_ARG, "stepping task: bad terrain geom ids");
    for (int g = b0; g < b0 + nbx; g++)
      if (IF(LHW_IF_GEOM_TYPE)[g] != G_BOX || bmap[IF(LHW_IF_GEOM_BODYID)[g]] != 0) return lhw_fail(LHW_ERR_ARG, "stepping task: terrain geoms must be boxes on static bodies");
    if (IF(LHW_IF_GEOM_TYPE)[fg] != G_PLANE) return lhw_fail(LHW_ERR_ARG, "stepping task: floor geom must be a plane");
  }
  const int32_t *jtype = IF(LHW_IF_JNT_TYPE), *dparent = IF(LHW_IF_DOF_PARENTID);
  const int32_t *djnt = IF(LHW_IF_DOF_JNTID), *jdof = IF(LHW_IF_JNT_DOFADR);
  // per-body model arrays re-indexed by the compacted body id
  std::vector<int32_t> parent(nb, 0), rootid(nb, 0), bdofadr(nb, 0), bdofnum(nb, 0);
  for (int b = 1; b < nb; b++) {
    parent[b] = bmap[IF(LHW_IF_BODY_PARENTID)[bsrc[b]]]; rootid[b] = 1;
    bdofadr[b] = IF(LHW_IF_BODY_DOFADR)[bsrc[b]]; bdofnum[b] = IF(LHW_IF_BODY_DOFNUM)[bsrc[b]];
    if (parent[b] >= b) return lhw_fail(LHW_ERR_MODEL, "bodies must be in depth-first order");
  }
  for (int j = 0; j < nj; j++)
    if (jtype[j] != JT_FREE && jtype[j] != JT_SLIDE && jtype[j] != JT_HINGE) return lhw_fail(LHW_ERR_UNSUPPORTED, "joint type %d", jtype[j]);
  for (int g = 0; g < ng; g++) {
    int cd = IF(LHW_IF_GEOM_CONDIM)[g];
    if (cd != 1 && cd != 3) return lhw_fail(LHW_ERR_UNSUPPORTED, "condim %d", cd);
  }
  if (mi[LHW_IH_CONE] != 0) return lhw_fail(LHW_ERR_UNSUPPORTED, "only the pyramidal cone is implemented");
  int primbox_pairs = 0, cyl_pairs = 0;
  for (int q = 0; q < np; q++) {
    const int t1 = IF(LHW_IF_GEOM_TYPE)[IF(LHW_IF_PAIR_GEOM1)[q]], t2 = IF(LHW_IF_GEOM_TYPE)[IF(LHW_IF_PAIR_GEOM2)[q]];
    if (!stepping && t1 == G_BOX && t2 == G_BOX)
      return lhw_fail(LHW_ERR_UNSUPPORTED, "box-box pairs are only compiled into the stepping-task kernels");
    // narrow phases that exist (geom1 type <= geom2 type, as MuJoCo orders a pair): a model packed straight from an mjModel
    // (pack_from_mjmodel) does not pass through mjcf.build_pairs, which refuses the others
    const bool ok = (t1 == G_PLANE && (t2 == G_SPHERE || t2 == G_CAPSULE || t2 == G_BOX || t2 == G_CYLINDER || t2 == G_ELLIPSOID)) ||
                    (t1 == G_SPHERE && (t2 == G_SPHERE || t2 == G_CAPSULE || t2 == G_BOX || t2 == G_CYLINDER)) ||
                    (t1 == G_CAPSULE && (t2 == G_CAPSULE || t2 == G_BOX)) || (t1 == G_BOX && t2 == G_BOX);
    if (t2 == G_BOX && (t1 == G_SPHERE || t1 == G_CAPSULE)) primbox_pairs++;
    if (t2 == G_CYLINDER || t2 == G_ELLIPSOID) cyl_pairs++;
    if (stepping && (t2 == G_CYLINDER || t2 == G_ELLIPSOID)) return lhw_fail(LHW_ERR_UNSUPPORTED, "cylinder / ellipsoid geoms are not compiled into the stepping-task kernels");
    if (!ok) return lhw_fail(LHW_ERR_UNSUPPORTED, "collision pair %d (geoms %d / %d): no narrow phase for geom types %d / %d -- plane, sphere, capsule, box among "
                             "themselves, cylinders against planes and spheres, ellipsoids against planes (MuJoCo resolves the other cylinder / ellipsoid pairs and meshes through its general "
                             "convex collider): mask the pair with contype / conaffinity or replace the geom by an enclosing capsule / box",
                             q, IF(LHW_IF_PAIR_GEOM1)[q], IF(LHW_IF_PAIR_GEOM2)[q], t1, t2);
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return lhw_fail(LHW_ERR_NO_DEVICE, "no HIP device");
  HumanoidEnv* h = new HumanoidEnv();
  h->device = cfg->device;
  h->p_dev = nullptr; h->m_dev = nullptr; h->iteration = 0;
  // two envs per wave (W = 32) where the model fits half a wavefront; the stepping task needs the 16-contact layout throughout
  h->fast = !stepping && np <= 32 && ng <= 16 && nj <= 32 && nb <= (stand ? 15 : 18) && !getenv("LHW_ONE_ENV_PER_WAVE");
  HModel& m = h->m;
  memset(&m, 0, sizeof m);
  m.nq = nq; m.nv = nv; m.nu = nu; m.nbody = nb; m.njnt = nj; m.ngeom = ng; m.npair = np;
  m.iterations = mi[LHW_IH_ITERATIONS]; m.disableflags = mi[LHW_IH_DISABLEFLAGS];
  m.timestep = md[LHW_DH_TIMESTEP]; m.gravity[0] = md[LHW_DH_GRAVITY_X]; m.gravity[1] = md[LHW_DH_GRAVITY_Y]; m.gravity[2] = md[LHW_DH_GRAVITY_Z];
  m.tolerance = md[LHW_DH_TOLERANCE]; m.meaninertia = md[LHW_DH_MEANINERTIA]; m.totalmass = md[LHW_DH_TOTALMASS];
  bool ok = true;
  // ---- derived structure
  std::vector<int> level(nb, 0), subend(nb, 0);
  std::vector<unsigned> bmask(nb, 0), pmask(nv, 0);
  int nlevel = 1;
  for (int b = 1; b < nb; b++) { level[b] = level[parent[b]] + 1; nlevel = std::max(nlevel, level[b] + 1); }
  for (int b = nb - 1; b >= 0; b--) {
    subend[b] = std::max(subend[b], b + 1);
    if (b > 0) subend[parent[b]] = std::max(subend[parent[b]], subend[b]);
  }
  for (int b = 1; b < nb; b++) {
    int bb = b;
    while (bb > 0 && bdofnum[bb] == 0) bb = parent[bb];
    if (bb > 0) for (int d = bdofadr[bb] + bdofnum[bb] - 1; d >= 0; d = dparent[d]) bmask[b] |= 1u << d;
  }
  for (int d = 0; d < nv; d++) {
    const int j = djnt[d], k = d - jdof[j];
    if (jtype[j] == JT_FREE) {
      if (k < 3) pmask[d] = 0xFFFFFFFFu;           // marker: cdof_dot = 0
      else pmask[d] = 7u << jdof[j];               // translations of the same joint only
    } else {
      for (int a = dparent[d]; a >= 0; a = dparent[a]) pmask[d] |= 1u << a;
    }
  }
  // ---- packed per-role tables
  std::vector<double> body_d((size_t)nb * BDS, 0.0), jnt_d((size_t)nj * JDS, 0.0), dof_d((size_t)nv * DDS, 0.0),
      geom_d((size_t)ng * GDS, 0.0), act_d((size_t)nu * ADS, 0.0);
  std::vector<int> body_i((size_t)nb * BIS, 0), jnt_i((size_t)nj * JIS, 0), dof_i((size_t)nv * DIS, 0), geom_i((size_t)ng * GIS, 0),
      act_i((size_t)nu * AIS, 0), pair_i((size_t)np * PIS, 0);
  for (int b = 0; b < nb; b++) {
    double* k = &body_d[(size_t)BDS * b];
    const int mb = bsrc[b];
    for (int a = 0; a < 3; a++) { k[BD_POS + a] = DF(LHW_DF_BODY_POS)[3 * mb + a]; k[BD_IPOS + a] = DF(LHW_DF_BODY_IPOS)[3 * mb + a]; k[BD_INERTIA + a] = DF(LHW_DF_BODY_INERTIA)[3 * mb + a]; }
    h_quat2mat(k + BD_RBODY, DF(LHW_DF_BODY_QUAT) + 4 * mb);
    h_quat2mat(k + BD_RINERT, DF(LHW_DF_BODY_IQUAT) + 4 * mb);
    k[BD_MASS] = DF(LHW_DF_BODY_MASS)[mb];
    k[BD_INVW] = DF(LHW_DF_BODY_INVWEIGHT0)[2 * mb]; k[BD_INVW + 1] = DF(LHW_DF_BODY_INVWEIGHT0)[2 * mb + 1];
    const int jn = IF(LHW_IF_BODY_JNTNUM)[mb], ja = IF(LHW_IF_BODY_JNTADR)[mb];
    int* bi = &body_i[(size_t)BIS * b];
    bi[0] = parent[b]; bi[BI_LEVEL] = level[b]; bi[2] = -1; bi[3] = 0;
    bi[BI_ROOT] = rootid[b]; bi[BI_SUBEND] = subend[b]; bi[BI_DOFMASK] = (int)bmask[b]; bi[BI_JNTADR] = ja < 0 ? 0 : ja;
    if (jn > 1) { humanoid_destroy(h); return lhw_fail(LHW_ERR_UNSUPPORTED, "bodies with more than one joint are not supported by the wave-per-env stepper"); }
    if (jn == 1) {
      for (int a = 0; a < 3; a++) { k[BD_JAXIS + a] = DF(LHW_DF_JNT_AXIS)[3 * ja + a]; k[BD_JPOS + a] = DF(LHW_DF_JNT_POS)[3 * ja + a]; }
      for (int a = 0; a < 3; a++) {  // R_body * jnt_pos and R_body * jnt_axis (anchor / axis in the parent frame)
        k[BD_V1 + a] = k[BD_RBODY + 3 * a] * k[BD_JPOS] + k[BD_RBODY + 3 * a + 1] * k[BD_JPOS + 1] + k[BD_RBODY + 3 * a + 2] * k[BD_JPOS + 2];
        k[BD_V2 + a] = k[BD_RBODY + 3 * a] * k[BD_JAXIS] + k[BD_RBODY + 3 * a + 1] * k[BD_JAXIS + 1] + k[BD_RBODY + 3 * a + 2] * k[BD_JAXIS + 2];
      }
      bi[2] = jtype[ja]; bi[3] = IF(LHW_IF_JNT_QPOSADR)[ja];
      k[BD_Q0] = jtype[ja] == JT_FREE ? 0.0 : DF(LHW_DF_QPOS0)[IF(LHW_IF_JNT_QPOSADR)[ja]];
    }
  }
  for (int j = 0; j < nj; j++) {
    double* k = &jnt_d[(size_t)JDS * j];
    k[JD_RANGE] = DF(LHW_DF_JNT_RANGE)[2 * j]; k[JD_RANGE + 1] = DF(LHW_DF_JNT_RANGE)[2 * j + 1];
    k[JD_SOLREF] = DF(LHW_DF_JNT_SOLREF)[2 * j]; k[JD_SOLREF + 1] = DF(LHW_DF_JNT_SOLREF)[2 * j + 1];
    for (int a = 0; a < 5; a++) k[JD_SOLIMP + a] = DF(LHW_DF_JNT_SOLIMP)[5 * j + a];
    k[JD_MARGIN] = DF(LHW_DF_JNT_MARGIN)[j];
    int* ji = &jnt_i[(size_t)JIS * j];
    ji[JI_TYPE] = jtype[j]; ji[JI_LIMITED] = IF(LHW_IF_JNT_LIMITED)[j]; ji[JI_QADR] = IF(LHW_IF_JNT_QPOSADR)[j]; ji[JI_DADR] = jdof[j];
  }
  for (int d = 0; d < nv; d++) {
    double* k = &dof_d[(size_t)DDS * d];
    k[DD_ARMATURE] = DF(LHW_DF_DOF_ARMATURE)[d]; k[DD_DAMPING] = DF(LHW_DF_DOF_DAMPING)[d]; k[DD_INVW] = DF(LHW_DF_DOF_INVWEIGHT0)[d];
    k[DD_FLOSS] = DF(LHW_DF_DOF_FRICTIONLOSS)[d];
    k[DD_SOLREF] = DF(LHW_DF_DOF_SOLREF)[2 * d]; k[DD_SOLREF + 1] = DF(LHW_DF_DOF_SOLREF)[2 * d + 1];
    for (int a = 0; a < 5; a++) k[DD_SOLIMP + a] = DF(LHW_DF_DOF_SOLIMP)[5 * d + a];
    int* di = &dof_i[(size_t)DIS * d];
    const int j = djnt[d], kk = d - jdof[j];
    di[DI_BODY] = bmap[IF(LHW_IF_DOF_BODYID)[d]]; di[DI_JNT] = j;
    di[DI_KIND] = jtype[j] == JT_FREE ? (kk < 3 ? 0 : 1) : (jtype[j] == JT_SLIDE ? 2 : 3);
    di[DI_PREVMASK] = (int)pmask[d];
    di[DI_KIDX] = kk; di[DI_LIMITED] = IF(LHW_IF_JNT_LIMITED)[j]; di[DI_QADR] = IF(LHW_IF_JNT_QPOSADR)[j];
    k[DD_RANGE] = DF(LHW_DF_JNT_RANGE)[2 * j]; k[DD_RANGE + 1] = DF(LHW_DF_JNT_RANGE)[2 * j + 1]; k[DD_MARGIN] = DF(LHW_DF_JNT_MARGIN)[j];
    di[DI_ACT] = -1;
    for (int u = 0; u < nu; u++)
      if (jdof[IF(LHW_IF_ACTUATOR_TRNID)[u]] == d) {
        if (di[DI_ACT] >= 0) { humanoid_destroy(h); return lhw_fail(LHW_ERR_UNSUPPORTED, "more than one actuator on dof %d", d); }
        di[DI_ACT] = u;
        k[DD_GEAR] = DF(LHW_DF_ACTUATOR_GEAR)[u];
      }
  }
  for (int g = 0; g < ng; g++) {
    double* k = &geom_d[(size_t)GDS * g];
    for (int a = 0; a < 3; a++) { k[GD_POS + a] = DF(LHW_DF_GEOM_POS)[3 * g + a]; k[GD_SIZE + a] = DF(LHW_DF_GEOM_SIZE)[3 * g + a]; k[GD_FRICTION + a] = DF(LHW_DF_GEOM_FRICTION)[3 * g + a]; }
    h_quat2mat(k + GD_RLOC, DF(LHW_DF_GEOM_QUAT) + 4 * g);
    const int gmb = IF(LHW_IF_GEOM_BODYID)[g];
    if (gmb != 0 && bmap[gmb] == 0) {  // geom of a static body: fold the body pose in
      double Rb[9], Rg[9], pg[3];
      h_quat2mat(Rb, DF(LHW_DF_BODY_QUAT) + 4 * gmb);
      for (int a = 0; a < 9; a++) Rg[a] = k[GD_RLOC + a];
      for (int a = 0; a < 3; a++) pg[a] = k[GD_POS + a];
      for (int r = 0; r < 3; r++) {
        k[GD_POS + r] = DF(LHW_DF_BODY_POS)[3 * gmb + r] + Rb[3 * r] * pg[0] + Rb[3 * r + 1] * pg[1] + Rb[3 * r + 2] * pg[2];
        for (int c = 0; c < 3; c++) k[GD_RLOC + 3 * r + c] = Rb[3 * r] * Rg[c] + Rb[3 * r + 1] * Rg[3 + c] + Rb[3 * r + 2] * Rg[6 + c];
      }
    }
    k[GD_SOLMIX] = DF(LHW_DF_GEOM_SOLMIX)[g];
    k[GD_SOLREF] = DF(LHW_DF_GEOM_SOLREF)[2 * g]; k[GD_SOLREF + 1] = DF(LHW_DF_GEOM_SOLREF)[2 * g + 1];
    for (int a = 0; a < 5; a++) k[GD_SOLIMP + a] = DF(LHW_DF_GEOM_SOLIMP)[5 * g + a];
    k[GD_MARGIN] = DF(LHW_DF_GEOM_MARGIN)[g]; k[GD_GAP] = DF(LHW_DF_GEOM_GAP)[g];
    if (stepping && g >= cfg->task_iparams[LHW_TI_STEP_BOX_GEOM0] && g < cfg->task_iparams[LHW_TI_STEP_BOX_GEOM0] + MAX_SEQ)
      for (int a = 0; a < 3; a++) k[GD_SIZE + a] = cfg->task_params[LHW_TP_STEP_BOX_SIZE + a];  // stepping_task.py:325
    int* gi = &geom_i[(size_t)GIS * g];
    gi[GI_TYPE] = IF(LHW_IF_GEOM_TYPE)[g]; gi[GI_BODY] = bmap[gmb]; gi[GI_CONDIM] = IF(LHW_IF_GEOM_CONDIM)[g];
    gi[GI_PRIORITY] = IF(LHW_IF_GEOM_PRIORITY)[g];
  }
  std::vector<double> pair_d((size_t)np * PDS, 0.0);
  for (int q = 0; q < np; q++) {   // mj_contactParam: priority, else solmix-weighted mix; friction = max; condim = max
    const int g1 = IF(LHW_IF_PAIR_GEOM1)[q], g2 = IF(LHW_IF_PAIR_GEOM2)[q];
    const double *G1 = &geom_d[(size_t)GDS * g1], *G2 = &geom_d[(size_t)GDS * g2];
    const int *I1 = &geom_i[(size_t)GIS * g1], *I2 = &geom_i[(size_t)GIS * g2];
    double* pd = &pair_d[(size_t)PDS * q];
    int* pi = &pair_i[(size_t)PIS * q];
    pi[0] = g1; pi[1] = g2; pi[PI_TYPE1] = I1[GI_TYPE]; pi[PI_TYPE2] = I2[GI_TYPE];
    pi[PI_BODY2] = I2[GI_BODY]; pi[PI_ROOT1] = body_i[(size_t)BIS * I1[GI_BODY] + BI_ROOT];
    for (int a = 0; a < 3; a++) { pd[PD_SIZE1 + a] = G1[GD_SIZE + a]; pd[PD_SIZE2 + a] = G2[GD_SIZE + a]; }
    {  // radius of a sphere about the geom's centre that contains it (planes: unused)
      auto rbound = [](int type, const double* sz) {
        if (type == G_SPHERE) return sz[0];
        if (type == G_CAPSULE) return sz[0] + sz[1];
        if (type == G_BOX) return std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
        if (type == G_CYLINDER) return std::sqrt(sz[0] * sz[0] + sz[1] * sz[1]);
        if (type == G_ELLIPSOID) return std::max(sz[0], std::max(sz[1], sz[2]));
        return 0.0;
      };
      pd[PD_RBOUND1] = rbound(I1[GI_TYPE], G1 + GD_SIZE); pd[PD_RBOUND2] = rbound(I2[GI_TYPE], G2 + GD_SIZE);
    }
    pd[0] = std::max(G1[GD_MARGIN], G2[GD_MARGIN]);
    pd[1] = pd[0] - std::max(G1[GD_GAP], G2[GD_GAP]);
    if (I1[GI_PRIORITY] != I2[GI_PRIORITY]) {
      const double* G = I1[GI_PRIORITY] > I2[GI_PRIORITY] ? G1 : G2;
      pi[2] = (I1[GI_PRIORITY] > I2[GI_PRIORITY] ? I1 : I2)[GI_CONDIM];
      pd[2] = G[GD_FRICTION]; pd[3] = G[GD_SOLREF]; pd[4] = G[GD_SOLREF + 1];
      for (int a = 0; a < 5; a++) pd[5 + a] = G[GD_SOLIMP + a];
    } else {
      pi[2] = std::max(I1[GI_CONDIM], I2[GI_CONDIM]);
      const double m1 = G1[GD_SOLMIX], m2 = G2[GD_SOLMIX];
      double mix;
      if (m1 >= HMINVAL && m2 >= HMINVAL) mix = m1 / (m1 + m2);
      else if (m1 < HMINVAL && m2 < HMINVAL) mix = 0.5;
      else mix = m1 < HMINVAL ? 0.0 : 1.0;
      for (int a = 0; a < 2; a++)
        pd[3 + a] = (G1[GD_SOLREF] > 0 && G2[GD_SOLREF] > 0) ? mix * G1[GD_SOLREF + a] + (1 - mix) * G2[GD_SOLREF + a] : std::min(G1[GD_SOLREF + a], G2[GD_SOLREF + a]);
      for (int a = 0; a < 5; a++) pd[5 + a] = mix * G1[GD_SOLIMP + a] + (1 - mix) * G2[GD_SOLIMP + a];
      pd[2] = std::max(G1[GD_FRICTION], G2[GD_FRICTION]);
    }
    const int b1 = I1[GI_BODY], b2 = I2[GI_BODY];
    pi[3] = (int)(bmask[b1] ^ bmask[b2]); pi[4] = (int)bmask[b2];
    // pair class (fwd_collision merges bitwise identical contacts of the same class): pairs whose contact parameters and dof
    // masks are equal AND whose bodies play the same roles in the task's contact queries (geom1 on / off the robot, body of geom2)
    pi[5] = q;
    for (int e = 0; e < q; e++) {
      const double* ed = &pair_d[(size_t)PDS * e];
      const int* ei = &pair_i[(size_t)PIS * e];
      const int eb1 = geom_i[(size_t)GIS * ei[0] + GI_BODY], eb2 = geom_i[(size_t)GIS * ei[1] + GI_BODY];
      bool same = ei[2] == pi[2] && ei[3] == pi[3] && ei[4] == pi[4] && eb2 == b2 && (eb1 == 0) == (b1 == 0);
      for (int a = 0; a < 10 && same; a++) same = ed[a] == pd[a];
      same = same && ed[10] == DF(LHW_DF_GEOM_INVWEIGHT0)[2 * g1] + DF(LHW_DF_GEOM_INVWEIGHT0)[2 * g2];
      if (same) { pi[5] = ei[5]; break; }
    }
    // merge class: a robot body against a static geom, whichever of the two is geom1 (same parameters, same robot body); pi[7] = 1
    // if the robot's geom is geom1.  Other pairs (robot-robot, static-static) are a class of their own.
    pi[6] = q; pi[7] = (b1 != 0 && b2 == 0) ? 1 : 0;
    if ((b1 == 0) != (b2 == 0)) {
      const int rb = b1 != 0 ? b1 : b2;
      for (int e = 0; e < q; e++) {
        const double* ed = &pair_d[(size_t)PDS * e];
        const int* ei = &pair_i[(size_t)PIS * e];
        const int eb1 = geom_i[(size_t)GIS * ei[0] + GI_BODY], eb2 = geom_i[(size_t)GIS * ei[1] + GI_BODY];
        if ((eb1 == 0) == (eb2 == 0) || (eb1 != 0 ? eb1 : eb2) != rb || ei[2] != pi[2] || ei[3] != pi[3]) continue;
        bool same = true;
        for (int a = 0; a < 10 && same; a++) same = ed[a] == pd[a];
        same = same && ed[10] == DF(LHW_DF_GEOM_INVWEIGHT0)[2 * g1] + DF(LHW_DF_GEOM_INVWEIGHT0)[2 * g2];
        if (same) { pi[6] = ei[6]; break; }
      }
    }
    pd[10] = DF(LHW_DF_GEOM_INVWEIGHT0)[2 * g1] + DF(LHW_DF_GEOM_INVWEIGHT0)[2 * g2];   // (the geoms' own copies: Model.fuse_static keeps them)
  }
  for (int u = 0; u < nu; u++) {
    double* k = &act_d[(size_t)ADS * u];
    k[AD_GEAR] = DF(LHW_DF_ACTUATOR_GEAR)[u];
    k[AD_CTRLRANGE] = DF(LHW_DF_ACTUATOR_CTRLRANGE)[2 * u]; k[AD_CTRLRANGE + 1] = DF(LHW_DF_ACTUATOR_CTRLRANGE)[2 * u + 1];
    k[AD_FORCERANGE] = DF(LHW_DF_ACTUATOR_FORCERANGE)[2 * u]; k[AD_FORCERANGE + 1] = DF(LHW_DF_ACTUATOR_FORCERANGE)[2 * u + 1];
    int* ai = &act_i[(size_t)AIS * u];
    const int j = IF(LHW_IF_ACTUATOR_TRNID)[u];
    ai[AI_DOF] = jdof[j]; ai[AI_JNT] = j; ai[AI_CTRLLIMITED] = IF(LHW_IF_ACTUATOR_CTRLLIMITED)[u]; ai[AI_FORCELIMITED] = IF(LHW_IF_ACTUATOR_FORCELIMITED)[u];
    ai[AI_QADR] = IF(LHW_IF_JNT_QPOSADR)[j]; ai[AI_DADR] = jdof[j];
  }
  m.nlevel = nlevel;
  // chain structure (see chain_solve): dofs 0..5 one free joint, then two serial chains of equal length hanging off the root
  const int nch = (nv - 6) / 2;
  bool chain2 = nv >= 8 && nv == 6 + 2 * nch && nch <= 10 && jtype[djnt[0]] == JT_FREE;
  for (int d = 0; chain2 && d < 6; d++) chain2 = djnt[d] == djnt[0] && dparent[d] == d - 1;
  for (int c = 0; chain2 && c < 2; c++)
    for (int k = 0; chain2 && k < nch; k++) chain2 = dparent[6 + c * nch + k] == (k == 0 ? 5 : 6 + c * nch + k - 1);
  if (!chain2) { humanoid_destroy(h); return lhw_fail(LHW_ERR_UNSUPPORTED, "the humanoid stepper needs a free root joint carrying two serial chains of (nv - 6) / 2 dofs each (the legs), dofs in that order"); }
  // bodies by owner dof (the last dof on their path from the root): a chain lane owns what moves with its dof, the root's
  // twelve lanes share what moves with the root
  std::vector<std::vector<int>> owned(32);
  {
    int nroot = 0;
    for (int b = 1; b < nb; b++) {
      if (!bmask[b]) continue;
      int od = 31;
      while (!((bmask[b] >> od) & 1u)) od--;
      if (od < 6) { owned[(nroot % 6) + 16 * ((nroot / 6) % 2)].push_back(b); nroot++; }
      else { const int c = (od - 6) / nch, k = (od - 6) % nch; owned[16 * c + 6 + k].push_back(b); }
    }
  }
  // kinematic tables of the chain layout (fwd_kinematics): frame of every jointed body relative to the previous jointed body
  // (welded bodies in between folded in), frame of every welded body in the jointed body it moves with
  std::vector<int> kin_i(32 * KIS, -1), fix_i(nb, -1);
  std::vector<double> kin_d(32 * KDS, 0.0), fix_d((size_t)nb * 12, 0.0);
  {
    auto rel = [&](int b, int anc, double* Rout, double* pout) {   // frame of body b in (jointed) ancestor anc, through joint-less bodies only
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
      for (int x = b; x != anc; x = parent[x]) {                   // T <- T(x in its parent) o T
        const double* k = &body_d[(size_t)BDS * x];
        double Rn[9], pn[3];
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) Rn[3 * r + c] = k[BD_RBODY + 3 * r] * R[c] + k[BD_RBODY + 3 * r + 1] * R[3 + c] + k[BD_RBODY + 3 * r + 2] * R[6 + c];
          pn[r] = k[BD_POS + r] + k[BD_RBODY + 3 * r] * pp[0] + k[BD_RBODY + 3 * r + 1] * pp[1] + k[BD_RBODY + 3 * r + 2] * pp[2];
        }
        std::copy(Rn, Rn + 9, R); std::copy(pn, pn + 3, pp);
        if (x != b && body_i[(size_t)BIS * x + 2] >= 0) return false;   // a jointed body in between: not a chain
        if (x == 0) return false;
      }
      std::copy(R, R + 9, Rout); std::copy(pp, pp + 3, pout);
      return true;
    };
    bool okk = true;
    auto fill = [&](int lane, int b, int prevb) {
      int* ki = &kin_i[(size_t)lane * KIS];
      double* kd = &kin_d[(size_t)lane * KDS];
      const double* k = &body_d[(size_t)BDS * b];
      ki[0] = b; ki[1] = body_i[(size_t)BIS * b + 2]; ki[2] = body_i[(size_t)BIS * b + 3]; ki[3] = body_i[(size_t)BIS * b + BI_JNTADR];
      if (prevb >= 0) okk = okk && rel(b, prevb, kd, kd + 9);
      for (int a = 0; a < 3; a++) { kd[12 + a] = k[BD_JAXIS + a]; kd[15 + a] = k[BD_JPOS + a]; }
      kd[18] = k[BD_Q0];
    };
    fill(5, 1, -1); fill(21, 1, -1);                       // the root body (free joint: frame from qpos), in both rows
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < nch; k++) {
        const int b = dof_i[(size_t)DIS * (6 + c * nch + k) + DI_BODY];
        const int pb = k == 0 ? 1 : dof_i[(size_t)DIS * (6 + c * nch + k - 1) + DI_BODY];
        fill(16 * c + 6 + k, b, pb);
      }
    for (int b = 2; b < nb; b++) {
      if (body_i[(size_t)BIS * b + 2] >= 0) continue;      // has a joint
      int ow = parent[b];
      while (ow > 0 && body_i[(size_t)BIS * ow + 2] < 0) ow = parent[ow];
      if (ow <= 0) { okk = false; break; }
      fix_i[b] = ow;
      okk = okk && rel(b, ow, &fix_d[(size_t)12 * b], &fix_d[(size_t)12 * b + 9]);
    }
    if (!okk || body_i[(size_t)BIS * 1 + 2] != JT_FREE) { humanoid_destroy(h); return lhw_fail(LHW_ERR_UNSUPPORTED, "kinematic tables: the jointed bodies must form root -> chain A / chain B with only joint-less bodies in between"); }
  }
  size_t max_owned = 1;
  for (auto& o : owned) max_owned = std::max(max_owned, o.size());
  std::vector<int> own_tab(32 * max_owned, -1);
  for (int l = 0; l < 32; l++) for (size_t q = 0; q < owned[l].size(); q++) own_tab[l * max_owned + q] = owned[l][q];
  m.max_owned = (int)max_owned;
  m.has_primbox = primbox_pairs > 0;
  m.has_cyl = cyl_pairs > 0;
  {
    int cnt = 0, idx[4] = {0, 0, 0, 0};
    for (int q = 0; q < np; q++)
      if (pair_i[(size_t)PIS * q + PI_TYPE1] == G_PLANE && pair_i[(size_t)PIS * q + PI_TYPE2] == G_BOX) { if (cnt < 4) idx[cnt] = q; cnt++; }
    m.npb = cnt <= 4 ? cnt : 0;
    for (int a = 0; a < 4; a++) m.pb_pair[a] = idx[a];
  }
  auto BID = [&](int f) { const int b = cfg->task_iparams[f]; return (b >= 0 && b < nbm) ? bmap[b] : -1; };
  m.track_body[0] = BID(LHW_TI_ROOT_BODY); m.track_body[1] = BID(LHW_TI_RFOOT_BODY); m.track_body[2] = BID(LHW_TI_LFOOT_BODY);
  if (max_owned > MAX_OWNED) { humanoid_destroy(h); return lhw_fail(LHW_ERR_UNSUPPORTED, "more than %d bodies move with one dof (%d): fold welded links first (Model.fuse_static)", MAX_OWNED, (int)max_owned); }
  // the tables are members of HModel (fixed capacities; the limits were checked above)
  std::copy(body_d.begin(), body_d.end(), m.body_d); std::copy(jnt_d.begin(), jnt_d.end(), m.jnt_d); std::copy(dof_d.begin(), dof_d.end(), m.dof_d);
  std::copy(geom_d.begin(), geom_d.end(), m.geom_d); std::copy(act_d.begin(), act_d.end(), m.act_d); std::copy(pair_d.begin(), pair_d.end(), m.pair_d);
  std::copy(kin_d.begin(), kin_d.end(), m.kin_d); std::copy(fix_d.begin(), fix_d.end(), m.fix_d);
  std::copy(body_i.begin(), body_i.end(), m.body_i); std::copy(jnt_i.begin(), jnt_i.end(), m.jnt_i); std::copy(dof_i.begin(), dof_i.end(), m.dof_i);
  std::copy(geom_i.begin(), geom_i.end(), m.geom_i); std::copy(act_i.begin(), act_i.end(), m.act_i); std::copy(pair_i.begin(), pair_i.end(), m.pair_i);
  std::copy(kin_i.begin(), kin_i.end(), m.kin_i); std::copy(fix_i.begin(), fix_i.end(), m.fix_i);
  for (int a = 0; a < 32 * MAX_OWNED; a++) m.own_tab[a] = -1;
  std::copy(own_tab.begin(), own_tab.end(), m.own_tab);
  HParams& p = h->p;
  memset(&p, 0, sizeof p);
  p.n_envs = cfg->n_envs; p.frame_skip = cfg->frame_skip; p.max_traj_len = cfg->max_traj_len; p.period = cfg->period;
  p.root_body = BID(LHW_TI_ROOT_BODY); p.head_body = BID(LHW_TI_HEAD_BODY);
  p.rfoot_body = BID(LHW_TI_RFOOT_BODY); p.lfoot_body = BID(LHW_TI_LFOOT_BODY);
  if (p.root_body != 1 || p.head_body <= 0 || p.head_body >= nb || p.rfoot_body <= 0 || p.rfoot_body >= nb || p.lfoot_body <= 0 || p.lfoot_body >= nb)
    ok = false;
  p.env_id_base = (unsigned)cfg->env_id_base; p.seed = cfg->seed;
  p.action_smoothing = cfg->action_smoothing; p.goal_height = cfg->task_params[LHW_TP_GOAL_HEIGHT];
  p.task = stepping ? TASK_STEP : (walk ? TASK_WALK : (h1walk ? TASK_H1WALK : TASK_STAND));
  if (stepping) {
    p.box_geom0 = cfg->task_iparams[LHW_TI_STEP_BOX_GEOM0]; p.nbox = cfg->task_iparams[LHW_TI_STEP_NBOX];
    p.floor_geom = cfg->task_iparams[LHW_TI_STEP_FLOOR_GEOM]; p.delay_frames = cfg->task_iparams[LHW_TI_STEP_DELAY_FRAMES];
    p.nplans = nplans; p.target_radius = cfg->task_params[LHW_TP_STEP_RADIUS];
    for (int a = 0; a < 3; a++) { m.track_off[3 + a] = cfg->task_params[LHW_TP_STEP_RSITE + a]; m.track_off[6 + a] = cfg->task_params[LHW_TP_STEP_LSITE + a]; }
    ok = ok && (p.plans = to_dev<double>(h, cfg->task_params + LHW_TP_STEP_PLANS, (size_t)nplans * (1 + LHW_STEP_MAX_SEQ * 3)));
  }
  std::vector<double> obs_noise(35, 0.0);
  if (stand) {
    p.init_noise = cfg->task_params[LHW_TP_H1_INIT_NOISE]; p.force_mag = cfg->task_params[LHW_TP_H1_FORCE_MAG];
    p.torque_mag = cfg->task_params[LHW_TP_H1_TORQUE_MAG];
    for (int k = 0; k < 35; k++) obs_noise[k] = cfg->task_params[LHW_TP_H1_OBS_NOISE + k];
    const int32_t* ti = cfg->task_iparams;
    p.dynrand_interval = ti[LHW_TI_H1_DYNRAND_INTERVAL]; p.perturb_interval = ti[LHW_TI_H1_PERTURB_INTERVAL];
    p.n_pbody = ti[LHW_TI_H1_N_PBODY]; p.pbody[0] = BID(LHW_TI_H1_PBODY); p.pbody[1] = BID(LHW_TI_H1_PBODY + 1);
    p.n_rand_dof = 10; p.n_rand_body = 11;
    for (int k = 0; k < 10; k++) p.rand_dof[k] = ti[LHW_TI_H1_RAND_DOF + k];
    for (int k = 0; k < 11; k++) p.rand_body[k] = BID(LHW_TI_H1_RAND_BODY + k);
    if (p.n_pbody < 0 || p.n_pbody > 2) ok = false;
    for (int k = 0; k < p.n_pbody; k++) if (p.pbody[k] <= 0 || p.pbody[k] >= nb) ok = false;
    for (int k = 0; k < 10; k++) if (p.rand_dof[k] < 0 || p.rand_dof[k] >= nv) ok = false;
    for (int k = 0; k < 11; k++) if (p.rand_body[k] <= 0 || p.rand_body[k] >= nb) ok = false;
    p.env_params = 1;
  }
  if (cfg->init_noise > 0) p.init_noise = cfg->init_noise;   // any humanoid task (base_humanoid_env.py:260-263)
  bool jvrc_perturb = false;
  if (walk && cfg->perturb_interval > 0) {   // apply_perturbation on a JVRC task: wrenches live in the per-env record (no LDS copy: see chain_dynamics)
    if (cfg->n_perturb_bodies < 1 || cfg->n_perturb_bodies > 2) ok = false;
    p.perturb_interval = cfg->perturb_interval; p.n_pbody = cfg->n_perturb_bodies;
    p.force_mag = cfg->perturb_force; p.torque_mag = cfg->perturb_torque;
    for (int k = 0; ok && k < p.n_pbody; k++) {
      const int b = cfg->perturb_bodies[k];
      p.pbody[k] = (b > 0 && b < nbm) ? bmap[b] : -1;
      if (p.pbody[k] <= 0 || p.pbody[k] >= nb) ok = false;
    }
    jvrc_perturb = ok;
  }
  std::vector<double> nominal(nq), neutral(nu);
  for (int k = 0; k < nq; k++) nominal[k] = cfg->nominal_qpos ? cfg->nominal_qpos[k] : DF(LHW_DF_QPOS0)[k];
  for (int u = 0; u < nu; u++) neutral[u] = cfg->action_offset[u];  // task._neutral_pose == half-sitting pose == offsets (jvrc_walk.py:33)
  for (int u = 0; u < nu; u++) { p.kp[u] = cfg->kp[u]; p.kd[u] = cfg->kd[u]; p.action_offset[u] = cfg->action_offset[u]; p.neutral_pose[u] = neutral[u]; }
  for (int k = 0; k < nq; k++) p.nominal_qpos[k] = nominal[k];
  for (int k = 0; k < 35; k++) p.obs_noise[k] = obs_noise[k];
  ok = ok && (p.clock_lut = to_dev<double>(h, cfg->clock_lut, (walk || h1walk) ? (size_t)4 * cfg->period : 0));
  const size_t N = cfg->n_envs;
  h->st.prm = nullptr;
  if (ok && (p.env_params || jvrc_perturb)) {
    // per-env parameter records start from the model's defaults
    std::vector<double> one(PRM_D, 0.0), all((size_t)PRM_D * N);
    for (int d = 0; d < nv; d++) { one[P_DAMP + d] = DF(LHW_DF_DOF_DAMPING)[d]; one[P_FLOSS + d] = DF(LHW_DF_DOF_FRICTIONLOSS)[d]; }
    for (int b = 0; b < nb; b++) {
      one[P_MASS + b] = DF(LHW_DF_BODY_MASS)[bsrc[b]];
      for (int a = 0; a < 3; a++) one[P_IPOS + 3 * b + a] = DF(LHW_DF_BODY_IPOS)[3 * bsrc[b] + a];
    }
    for (size_t n = 0; n < N; n++) std::copy(one.begin(), one.end(), all.begin() + n * PRM_D);
    h->st.prm = const_cast<double*>(to_dev<double>(h, all.data(), all.size()).p);
    ok = ok && h->st.prm != nullptr;
    if (jvrc_perturb) p.xfrc_base = DevTab<double>{h->st.prm};
  }
  h->st.ter = nullptr;
  if (ok && stepping) {
    // before the first reset the boxes sit where the model file puts them (all at the pose of the first box) and the floor at 0
    std::vector<double> one(TER_D, 0.0), all((size_t)TER_D * N);
    const int g0 = p.box_geom0;
    for (int k = 0; k < MAX_SEQ; k++) {
      const double* gd = &geom_d[(size_t)GDS * (g0 + k)];
      one[T_SEQ + 6 * k] = gd[GD_POS]; one[T_SEQ + 6 * k + 1] = gd[GD_POS + 1]; one[T_SEQ + 6 * k + 2] = gd[GD_POS + 2] + gd[GD_SIZE + 2];
      one[T_SEQ + 6 * k + 3] = 0; one[T_SEQ + 6 * k + 4] = 1; one[T_SEQ + 6 * k + 5] = 0;
    }
    for (size_t n = 0; n < N; n++) std::copy(one.begin(), one.end(), all.begin() + n * TER_D);
    h->st.ter = const_cast<double*>(to_dev<double>(h, all.data(), all.size()).p);
    ok = ok && h->st.ter != nullptr;
  }
  h->st.bigd = nullptr; h->st.bigi = nullptr;
  if (ok && stepping && !getenv("LHW_STEP_NO_BIG")) {   // many-contact workspace (newton_big): 51 KB per env
    void *bdp = nullptr, *bip = nullptr;
    ok = ok && lhw_malloc(&bdp, sizeof(double) * (size_t)BW_DOUBLES * N) == hipSuccess && lhw_malloc(&bip, sizeof(int) * (size_t)BW_INTS * N) == hipSuccess;
    if (bdp) h->dev_allocs.push_back(bdp);
    if (bip) h->dev_allocs.push_back(bip);
    h->st.bigd = (double*)bdp; h->st.bigi = (int*)bip;
  }
  void *rec = nullptr, *irec = nullptr, *eps = nullptr, *slow = nullptr;
  ok = ok && lhw_malloc(&slow, N + 1) == hipSuccess && hipMemset(slow, 0, N + 1) == hipSuccess;
  if (slow) h->dev_allocs.push_back(slow);
  h->st.slow = (unsigned char*)slow;
  // (one record more than envs: the reset template of the jvrc_walk kernels)
  ok = ok && lhw_malloc(&rec, sizeof(double) * REC_D * (N + 1)) == hipSuccess && hipMemset(rec, 0, sizeof(double) * REC_D * (N + 1)) == hipSuccess &&
       lhw_malloc(&irec, sizeof(int) * REC_I * (N + 1)) == hipSuccess && hipMemset(irec, 0, sizeof(int) * REC_I * (N + 1)) == hipSuccess &&
       lhw_malloc(&eps, sizeof(double) * 8) == hipSuccess && hipMemset(eps, 0, sizeof(double) * 8) == hipSuccess;
  if (rec) h->dev_allocs.push_back(rec);
  if (irec) h->dev_allocs.push_back(irec);
  if (eps) h->dev_allocs.push_back(eps);
  h->st.rec = (double*)rec; h->st.irec = (int*)irec; h->st.ep_stats = (double*)eps; h->st.prof = nullptr; h->st.wave_cyc = nullptr; h->st.tin = nullptr;
  if (!ok) { humanoid_destroy(h); return lhw_fail(LHW_ERR_HIP, "humanoid_create: device allocation failed or bad body ids"); }
  *obs_dim = stepping ? 39 : (walk ? 37 : (h1walk ? 43 : 35)); *act_dim = nu; *n_terms = ((walk && !stepping) || h1walk) ? 10 : 6;
  p.reset_template = -1;
  if (p.task == TASK_WALK && !(p.init_noise > 0) && !jvrc_perturb && !getenv("LHW_NO_RESET_TEMPLATE")) {   // (with init noise every reset has its own state: computed)
    // reset the template record (index N) once with the ordinary reset kernel; auto-resets copy its state from then on
    if (!humanoid_upload_params(h)) { humanoid_destroy(h); return lhw_fail(LHW_ERR_HIP, "humanoid_create: parameter upload failed"); }
    const HLaunch lz{(int)N, 1, 0, 0, 0};
    hipLaunchKernelGGL((humanoid_kernel<1, TASK_WALK, 64>), dim3(1), dim3(64), 0, 0, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz, h->st, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (unsigned char*)nullptr, (float*)nullptr, (const unsigned char*)nullptr, (double*)nullptr,
                       (double*)nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { humanoid_destroy(h); return lhw_fail(LHW_ERR_HIP, "humanoid_create: reset template launch failed"); }
    p.reset_template = (int)N;
  }
  if (!humanoid_upload_params(h)) { humanoid_destroy(h); return lhw_fail(LHW_ERR_HIP, "humanoid_create: parameter upload failed"); }
  *out = h;
  return LHW_OK;
}

void humanoid_destroy(HumanoidEnv* h) {
  if (!h) return;
  for (void* d : h->dev_allocs) (void)hipFree(d);
  delete h;
}

// (LHW_ONLY_WALK: development builds that compile the jvrc_walk kernels only -- a third of the compile time)
#ifdef LHW_ONLY_WALK
#define LAUNCH_OTHER_TASKS(MODE, WIDTH, ...)
#else
#define LAUNCH_OTHER_TASKS(MODE, WIDTH, ...)                                                                        \
    else if (pp_.task == TASK_STEP) hipLaunchKernelGGL((humanoid_kernel<MODE, TASK_STEP, 64>), grid_, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz_, h->st, __VA_ARGS__); \
    else if (pp_.task == TASK_H1WALK) hipLaunchKernelGGL((humanoid_kernel<MODE, TASK_H1WALK, WIDTH>), grid_, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz_, h->st, __VA_ARGS__); \
    else hipLaunchKernelGGL((humanoid_kernel<MODE, TASK_STAND, WIDTH>), grid_, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz_, h->st, __VA_ARGS__);
#endif
#define LAUNCH_RANGE(MODE, WIDTH, FLAGGED, FIRST, COUNT, ...)                                                       \
  do {                                                                                                             \
    const HParams& pp_ = h->p;                                                                                     \
    const HLaunch lz_{(FIRST), (COUNT), (FLAGGED), h->iteration, 0};                                                \
    /* the re-run launch scans the flags with few workgroups (at most 64 envs each): see humanoid_kernel */           \
    const int full_ = (lz_.env_count + (64 / WIDTH) - 1) / (64 / WIDTH);                                            \
    const dim3 grid_((FLAGGED) ? std::min(full_, std::max(256, (lz_.env_count + 63) / 64)) : full_);                \
    if (pp_.task == TASK_WALK) hipLaunchKernelGGL((humanoid_kernel<MODE, TASK_WALK, WIDTH>), grid_, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz_, h->st, __VA_ARGS__); \
    LAUNCH_OTHER_TASKS(MODE, WIDTH, __VA_ARGS__)                                                                   \
  } while (0)
#define LAUNCH(MODE, ...) LAUNCH_RANGE(MODE, 64, 0, 0, h->p.n_envs, __VA_ARGS__)
// One control step of envs [first, first + count): two envs per wave where the model allows it, followed by the one-env-per-wave
// kernel over the same range for the envs that flagged themselves (more than 8 contacts); that launch finds nothing to do on
// most steps: a few workgroups scan the flags and exit.
#define LAUNCH_STEP(FIRST, COUNT, ...)                                                       \
  do {                                                                                       \
    if (h->fast) {                                                                           \
      LAUNCH_RANGE(0, 32, 0, FIRST, COUNT, __VA_ARGS__);                                     \
      LAUNCH_RANGE(0, 64, 1, FIRST, COUNT, __VA_ARGS__);                                     \
    } else {                                                                                 \
      LAUNCH_RANGE(0, 64, 0, FIRST, COUNT, __VA_ARGS__);                                     \
    }                                                                                        \
  } while (0)

void humanoid_reset(HumanoidEnv* h, const uint8_t* mask, float* obs, hipStream_t s) {
  LAUNCH(1, (const float*)nullptr, obs, (float*)nullptr, (float*)nullptr, (unsigned char*)nullptr, (float*)nullptr, mask,
         (double*)nullptr, (double*)nullptr);
}
void humanoid_step(HumanoidEnv* h, const float* act, float* obs, float* term_obs, float* rew, uint8_t* done, float* rew_terms,
                   hipStream_t s) {
  LAUNCH_STEP(0, h->p.n_envs, act, obs, term_obs, rew, done, rew_terms, (const unsigned char*)nullptr, (double*)nullptr, (double*)nullptr);
}
// envs [first, first + count) only; the pointers are the full-batch arrays
int humanoid_step_range(HumanoidEnv* h, int first, int count, const float* act, float* obs, float* term_obs, float* rew, uint8_t* done,
                        float* rew_terms, hipStream_t s) {
  if (first < 0 || count <= 0 || first + count > h->p.n_envs) return -1;
  LAUNCH_STEP(first, count, act, obs, term_obs, rew, done, rew_terms, (const unsigned char*)nullptr, (double*)nullptr, (double*)nullptr);
  return 0;
}
void humanoid_get_state(HumanoidEnv* h, double* qpos, double* qvel, hipStream_t s) {
  LAUNCH(3, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (unsigned char*)nullptr, (float*)nullptr,
         (const unsigned char*)nullptr, qpos, qvel);
}
void humanoid_set_state(HumanoidEnv* h, const double* qpos, const double* qvel, hipStream_t s) {
  LAUNCH(2, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (unsigned char*)nullptr, (float*)nullptr,
         (const unsigned char*)nullptr, const_cast<double*>(qpos), const_cast<double*>(qvel));
}
double* humanoid_ep_stats(HumanoidEnv* h) { return h->st.ep_stats; }
int humanoid_occupancy() {
  int nb = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, humanoid_kernel<0, TASK_WALK, 32>, 64, 0) != hipSuccess) return -1;
  return nb;
}
// diagnostic: per-env duration of the last control-step launch (load balance across wavefronts)
int humanoid_wave_cycles(HumanoidEnv* h, long long* out) {
  const size_t N = h->p.n_envs;
  if (!h->st.wave_cyc) {
    void* d = nullptr;
    if (lhw_malloc(&d, (N + 1) * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(d, 0, (N + 1) * sizeof(long long));
    h->dev_allocs.push_back(d);
    h->st.wave_cyc = (long long*)d;
    return 0;
  }
  if (!out) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpy(out, h->st.wave_cyc, N * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
// batched sim facade: per-env record of the task layer's inputs (LhwTaskInput); enable allocates, disable stops the export
int humanoid_task_inputs(HumanoidEnv* h, int enable, double* out_host, double** out_dev) {
  const size_t n = (size_t)h->p.n_envs * LHW_TASK_INPUT_DIM;
  if (enable == 1 && !h->st.tin) {
    void* d = nullptr;
    if (lhw_malloc(&d, n * sizeof(double)) != hipSuccess) return -1;
    (void)hipMemset(d, 0, n * sizeof(double));
    h->dev_allocs.push_back(d);
    h->st.tin = (double*)d;
  } else if (enable == 0) {
    h->st.tin = nullptr;   // (the buffer is released with the env)
  }
  if (out_dev) *out_dev = h->st.tin;
  if (out_host) {
    if (!h->st.tin) return -2;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(out_host, h->st.tin, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  }
  return 0;
}
int humanoid_profile(HumanoidEnv* h, int enable, long long* out16) {
  if (enable && !h->st.prof) {
    void* d = nullptr;
    if (lhw_malloc(&d, 16 * sizeof(long long)) != hipSuccess) return -1;
    (void)hipMemset(d, 0, 16 * sizeof(long long));
    h->dev_allocs.push_back(d);
    h->st.prof = (long long*)d;
  }
  if (out16 && h->st.prof) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out16, h->st.prof, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipMemset(h->st.prof, 0, 16 * sizeof(long long));
#ifdef LHW_FINEPROF
    static const long long zero16[16] = {0};
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fine), 16 * sizeof(long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fine), zero16, 16 * sizeof(long long));
#endif
  }
  return 0;
}
// curriculum input of the stepping task (stair height, stepping_task.py:305); the other tasks ignore it
void humanoid_set_iteration(HumanoidEnv* h, int64_t it) { h->iteration = (int)std::min<int64_t>(it, 1 << 30); }
// sq / sv / frc of the persistent records (fields of the last forward pass) -> host, torque = force * gear
int humanoid_actuator_state(HumanoidEnv* h, double* pos, double* vel, double* tq) {
  const size_t N = h->p.n_envs;
  const int nu = h->m.nu;
  std::vector<double> rec(N * REC_D);
  const double* act_d = h->m.act_d;   // (host copy of the model record)
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(rec.data(), h->st.rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (size_t n = 0; n < N; n++)
    for (int u = 0; u < nu; u++) {
      if (pos) pos[n * nu + u] = rec[n * REC_D + R_SQ + u];
      if (vel) vel[n * nu + u] = rec[n * REC_D + R_SV + u];
      if (tq) tq[n * nu + u] = rec[n * REC_D + R_FRC + u] * act_d[(size_t)ADS * u + AD_GEAR];
    }
  return 0;
}
int humanoid_step_record(HumanoidEnv* h, double* seq, double* floor_z, int32_t* istate) {
  if (!h->st.ter) return -1;
  const size_t N = h->p.n_envs;
  std::vector<double> ter(N * TER_D);
  std::vector<int> irec(N * REC_I);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(ter.data(), h->st.ter, ter.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemcpy(irec.data(), h->st.irec, irec.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (size_t n = 0; n < N; n++) {
    if (seq) std::copy(ter.begin() + n * TER_D + T_SEQ, ter.begin() + n * TER_D + T_SEQ + MAX_SEQ * 6, seq + n * MAX_SEQ * 6);
    if (floor_z) floor_z[n] = ter[n * TER_D + T_FLOOR];
    if (istate) { const int* r = &irec[n * REC_I]; int32_t* o = istate + n * 5; o[0] = r[RI_T1]; o[1] = r[RI_T2]; o[2] = r[RI_REACHED]; o[3] = r[RI_FRAMES]; o[4] = r[RI_NSEQ]; }
  }
  return 0;
}
