/* TEST INFRASTRUCTURE -- CPU oracle, NOT the product path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The shipped stepper is learninghumanoidwalking_amd/csrc (HIP, gfx950).
 *
 * PARITY UNPINNED: this is a float64 restatement of the subset of MuJoCo 3.4.0's
 * mj_step that the reference executes (call site: reference
 * envs/common/robot_interface.py:544; pin: reference pyproject.toml:13, uv.lock:930-941).
 * MuJoCo itself is an un-vendored third-party dependency that is absent from
 * /root/reference and not installed in this image, and the reference's tests hold no
 * golden physics vectors, so the restatement below follows
 * MuJoCo's published algorithm ("Computation" chapter of its documentation and the
 * structure of engine_forward.c / engine_core_smooth.c / engine_core_constraint.c /
 * engine_collision_primitive.c / engine_solver.c) from recall, and is validated only
 * against analytic invariants (tests/test_oracle_physics.py).  Every function names the
 * MuJoCo routine it restates; "[MJ-recall]" marks details that must be re-checked
 * against a real MuJoCo 3.4.0 before any "matches CPU MuJoCo" claim.
 *
 * Plain serial C, dense linear algebra, no attempt at speed beyond -O2: it is the
 * checker and the labelled stand-in CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lhw_model_fields.h"

#define MAXCON 256
#define MAXEFC (4 * MAXCON + 128)
#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_ELLIPSOID = 4, GEOM_CYLINDER = 5, GEOM_BOX = 6 };
enum { EFC_FRICTION = 0, EFC_LIMIT = 1, EFC_CONTACT = 2 };
#define DSBL_WARMSTART (1 << 7)
#define DSBL_REFSAFE (1 << 11)
#define DSBL_EULERDAMP (1 << 14)

typedef struct {
  const int32_t* ib;
  const double* db;
  int nq, nv, nu, nbody, njnt, ngeom, npair, nsite;
} OModel;

static inline const int32_t* IF(const OModel* m, int f) { return m->ib + m->ib[LHW_IH_COUNT + f]; }
static inline const double* DF(const OModel* m, int f) { return m->db + m->ib[LHW_IH_COUNT + LHW_IF_COUNT + f]; }

typedef struct {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], includemargin;
  int geom1, geom2, dim, efc_address, exclude;
} OContact;

typedef struct OData {
  OModel m;
  /* state */
  double *qpos, *qvel, *ctrl, *xfrc_applied, *qacc_warmstart, time;
  /* position-dependent */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat;
  double *site_xpos, *site_xmat;
  double *subtree_com, *cinert, *crb, *cdof, *cvel, *cdof_dot, *cacc, *cfrc;
  double *M, *L; /* dense nv*nv, L = Cholesky factor scratch */
  double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_applied, *qfrc_smooth, *qacc_smooth;
  double *qfrc_constraint, *qacc;
  double *actuator_length, *actuator_velocity, *actuator_force;
  /* contacts / constraints */
  int ncon, nefc, nf, nl;
  OContact contact[MAXCON];
  double* efc_J; /* MAXEFC * nv */
  double efc_pos[MAXEFC], efc_margin[MAXEFC], efc_D[MAXEFC], efc_R[MAXEFC], efc_aref[MAXEFC];
  double efc_vel[MAXEFC], efc_KBIP[MAXEFC][4], efc_diagApprox[MAXEFC], efc_frictionloss[MAXEFC];
  double efc_force[MAXEFC];
  int efc_type[MAXEFC], efc_id[MAXEFC], efc_state[MAXEFC];
  int solver_niter;
  int warning_contactfull;
} OData;

/* ------------------------------------------------------------------ small math (engine_util_*.c) */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return n; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void mulQuat(double* r, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                 a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                 a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void quat2Mat(double* R, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
static void mulMatVec3(double* r, const double* R, const double* v) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulMatTVec3(double* r, const double* R, const double* v) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void rotVecQuat(double* r, const double* v, const double* q) {
  double R[9];
  quat2Mat(R, q);
  mulMatVec3(r, R, v);
}
static void axisAngle2Quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* mju_quatIntegrate */
static void quatIntegrate(double* q, const double* vel, double scale) {
  double ax[3] = {vel[0], vel[1], vel[2]};
  double ang = scale * normalize3(ax);
  double qr[4];
  axisAngle2Quat(qr, ax, ang);
  normalize4(q);
  mulQuat(q, q, qr);
  normalize4(q);
}
/* mju_mulInertVec: 10-number com-based inertia times spatial motion [rot; lin] */
static void mulInertVec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
/* mju_crossMotion / mju_crossForce */
static void crossMotion(double* r, const double* vel, const double* v) {
  double a[3], b[3], c[3];
  cross3(a, vel, v);
  cross3(b, vel, v + 3);
  cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void crossForce(double* r, const double* vel, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, vel, f);
  cross3(b, vel + 3, f + 3);
  cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

/* ------------------------------------------------------------------ allocation */
static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

OData* orc_new(const int32_t* ib, const double* db) {
  if ((uint32_t)ib[LHW_IH_MAGIC] != LHW_MODEL_MAGIC || ib[LHW_IH_VERSION] != LHW_MODEL_VERSION) return NULL;
  OData* d = (OData*)calloc(1, sizeof(OData));
  OModel* m = &d->m;
  m->ib = ib; m->db = db;
  m->nq = ib[LHW_IH_NQ]; m->nv = ib[LHW_IH_NV]; m->nu = ib[LHW_IH_NU]; m->nbody = ib[LHW_IH_NBODY];
  m->njnt = ib[LHW_IH_NJNT]; m->ngeom = ib[LHW_IH_NGEOM]; m->npair = ib[LHW_IH_NPAIR]; m->nsite = ib[LHW_IH_NSITE];
  int nv = m->nv, nb = m->nbody;
  d->qpos = dalloc(m->nq); d->qvel = dalloc(nv); d->ctrl = dalloc(m->nu); d->xfrc_applied = dalloc(6 * nb);
  d->qacc_warmstart = dalloc(nv);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb); d->xipos = dalloc(3 * nb);
  d->ximat = dalloc(9 * nb); d->xanchor = dalloc(3 * m->njnt); d->xaxis = dalloc(3 * m->njnt);
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom);
  d->site_xpos = dalloc(3 * m->nsite); d->site_xmat = dalloc(9 * m->nsite);
  d->subtree_com = dalloc(3 * nb); d->cinert = dalloc(10 * nb); d->crb = dalloc(10 * nb);
  d->cdof = dalloc(6 * nv); d->cvel = dalloc(6 * nb); d->cdof_dot = dalloc(6 * nv); d->cacc = dalloc(6 * nb);
  d->cfrc = dalloc(6 * nb);
  d->M = dalloc((size_t)nv * nv); d->L = dalloc((size_t)nv * nv);
  d->qfrc_bias = dalloc(nv); d->qfrc_passive = dalloc(nv); d->qfrc_actuator = dalloc(nv); d->qfrc_applied = dalloc(nv);
  d->qfrc_smooth = dalloc(nv); d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qacc = dalloc(nv);
  d->actuator_length = dalloc(m->nu); d->actuator_velocity = dalloc(m->nu); d->actuator_force = dalloc(m->nu);
  d->efc_J = dalloc((size_t)MAXEFC * nv);
  memcpy(d->qpos, DF(m, LHW_DF_QPOS0), sizeof(double) * m->nq);
  d->xquat[0] = 1; d->xmat[0] = d->xmat[4] = d->xmat[8] = 1; d->ximat[0] = d->ximat[4] = d->ximat[8] = 1;
  return d;
}

void orc_free(OData* d) {
  if (!d) return;
  double** p[] = {&d->qpos, &d->qvel, &d->ctrl, &d->xfrc_applied, &d->qacc_warmstart, &d->xpos, &d->xquat, &d->xmat,
                  &d->xipos, &d->ximat, &d->xanchor, &d->xaxis, &d->geom_xpos, &d->geom_xmat, &d->site_xpos,
                  &d->site_xmat, &d->subtree_com, &d->cinert, &d->crb, &d->cdof, &d->cvel, &d->cdof_dot, &d->cacc,
                  &d->cfrc, &d->M, &d->L, &d->qfrc_bias, &d->qfrc_passive, &d->qfrc_actuator, &d->qfrc_applied,
                  &d->qfrc_smooth, &d->qacc_smooth, &d->qfrc_constraint, &d->qacc, &d->actuator_length,
                  &d->actuator_velocity, &d->actuator_force, &d->efc_J};
  for (size_t i = 0; i < sizeof p / sizeof p[0]; i++) free(*p[i]);
  free(d);
}

/* mj_resetData: qpos=qpos0, everything else zero (reference envs/common/mujoco_env.py:114) */
void orc_reset_data(OData* d) {
  const OModel* m = &d->m;
  memcpy(d->qpos, DF(m, LHW_DF_QPOS0), sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memset(d->xfrc_applied, 0, sizeof(double) * 6 * m->nbody);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->qacc, 0, sizeof(double) * m->nv);
  memset(d->actuator_force, 0, sizeof(double) * m->nu);
  d->time = 0; d->ncon = 0; d->nefc = 0;
}

/* ------------------------------------------------------------------ mj_kinematics (engine_core_smooth.c) */
static void kinematics(OData* d) {
  const OModel* m = &d->m;
  const int32_t *parent = IF(m, LHW_IF_BODY_PARENTID), *jadr = IF(m, LHW_IF_BODY_JNTADR), *jnum = IF(m, LHW_IF_BODY_JNTNUM);
  const int32_t *jtype = IF(m, LHW_IF_JNT_TYPE), *jq = IF(m, LHW_IF_JNT_QPOSADR);
  const double *bpos = DF(m, LHW_DF_BODY_POS), *bquat = DF(m, LHW_DF_BODY_QUAT), *bipos = DF(m, LHW_DF_BODY_IPOS);
  const double *biquat = DF(m, LHW_DF_BODY_IQUAT), *jpos = DF(m, LHW_DF_JNT_POS), *jaxis = DF(m, LHW_DF_JNT_AXIS);
  const double* qpos0 = DF(m, LHW_DF_QPOS0);
  for (int i = 1; i < m->nbody; i++) {
    double *xp = d->xpos + 3 * i, *xq = d->xquat + 4 * i;
    int p = parent[i];
    if (jnum[i] == 1 && jtype[jadr[i]] == JNT_FREE) {
      int qa = jq[jadr[i]];
      normalize4(d->qpos + qa + 3); /* mj_kinematics normalises the free-joint quaternion in place */
      memcpy(xp, d->qpos + qa, 3 * sizeof(double));
      memcpy(xq, d->qpos + qa + 3, 4 * sizeof(double));
      memcpy(d->xanchor + 3 * jadr[i], xp, 3 * sizeof(double));
      memcpy(d->xaxis + 3 * jadr[i], jaxis + 3 * jadr[i], 3 * sizeof(double));
    } else {
      double t[3];
      mulMatVec3(t, d->xmat + 9 * p, bpos + 3 * i);
      xp[0] = d->xpos[3 * p] + t[0]; xp[1] = d->xpos[3 * p + 1] + t[1]; xp[2] = d->xpos[3 * p + 2] + t[2];
      mulQuat(xq, d->xquat + 4 * p, bquat + 4 * i);
      for (int k = 0; k < jnum[i]; k++) {
        int j = jadr[i] + k;
        double *anchor = d->xanchor + 3 * j, *axis = d->xaxis + 3 * j;
        rotVecQuat(axis, jaxis + 3 * j, xq);
        rotVecQuat(anchor, jpos + 3 * j, xq);
        anchor[0] += xp[0]; anchor[1] += xp[1]; anchor[2] += xp[2];
        double q = d->qpos[jq[j]] - qpos0[jq[j]];
        if (jtype[j] == JNT_SLIDE) {
          xp[0] += axis[0] * q; xp[1] += axis[1] * q; xp[2] += axis[2] * q;
        } else { /* hinge */
          double ql[4], v[3];
          axisAngle2Quat(ql, jaxis + 3 * j, q);
          mulQuat(xq, xq, ql);
          rotVecQuat(v, jpos + 3 * j, xq); /* off-centre rotation correction */
          xp[0] = anchor[0] - v[0]; xp[1] = anchor[1] - v[1]; xp[2] = anchor[2] - v[2];
        }
      }
    }
    normalize4(xq);
    quat2Mat(d->xmat + 9 * i, xq);
    double t[3], qi[4];
    mulMatVec3(t, d->xmat + 9 * i, bipos + 3 * i);
    d->xipos[3 * i] = xp[0] + t[0]; d->xipos[3 * i + 1] = xp[1] + t[1]; d->xipos[3 * i + 2] = xp[2] + t[2];
    mulQuat(qi, xq, biquat + 4 * i);
    quat2Mat(d->ximat + 9 * i, qi);
  }
  const int32_t* gbody = IF(m, LHW_IF_GEOM_BODYID);
  const double *gpos = DF(m, LHW_DF_GEOM_POS), *gquat = DF(m, LHW_DF_GEOM_QUAT);
  for (int g = 0; g < m->ngeom; g++) {
    int b = gbody[g];
    double t[3], q[4];
    mulMatVec3(t, d->xmat + 9 * b, gpos + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + t[k];
    mulQuat(q, d->xquat + 4 * b, gquat + 4 * g);
    quat2Mat(d->geom_xmat + 9 * g, q);
  }
  const int32_t* sbody = IF(m, LHW_IF_SITE_BODYID);
  const double *spos = DF(m, LHW_DF_SITE_POS), *squat = DF(m, LHW_DF_SITE_QUAT);
  for (int s = 0; s < m->nsite; s++) {
    int b = sbody[s];
    double t[3], q[4];
    mulMatVec3(t, d->xmat + 9 * b, spos + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] = d->xpos[3 * b + k] + t[k];
    mulQuat(q, d->xquat + 4 * b, squat + 4 * s);
    quat2Mat(d->site_xmat + 9 * s, q);
  }
}

/* ------------------------------------------------------------------ mj_comPos */
static void comPos(OData* d) {
  const OModel* m = &d->m;
  const int32_t *parent = IF(m, LHW_IF_BODY_PARENTID), *rootid = IF(m, LHW_IF_BODY_ROOTID);
  const int32_t *jtype = IF(m, LHW_IF_JNT_TYPE), *jdof = IF(m, LHW_IF_JNT_DOFADR), *jbody = IF(m, LHW_IF_JNT_BODYID);
  const double *mass = DF(m, LHW_DF_BODY_MASS), *inertia = DF(m, LHW_DF_BODY_INERTIA);
  int nb = m->nbody;
  double* sm = (double*)calloc(nb, sizeof(double));
  for (int i = 0; i < nb; i++) {
    sm[i] = mass[i];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * i + k] = mass[i] * d->xipos[3 * i + k];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = parent[i];
    sm[p] += sm[i];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * i + k];
  }
  for (int i = 0; i < nb; i++) {
    if (sm[i] < MINVAL) memcpy(d->subtree_com + 3 * i, d->xipos + 3 * i, 3 * sizeof(double));
    else for (int k = 0; k < 3; k++) d->subtree_com[3 * i + k] /= sm[i];
  }
  free(sm);
  /* cinert: mju_inertCom */
  for (int i = 1; i < nb; i++) {
    const double *R = d->ximat + 9 * i, *I = inertia + 3 * i;
    double dif[3], T[9], *r = d->cinert + 10 * i;
    for (int k = 0; k < 3; k++) dif[k] = d->xipos[3 * i + k] - d->subtree_com[3 * rootid[i] + k];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) T[3 * a + b] = R[3 * a] * I[0] * R[3 * b] + R[3 * a + 1] * I[1] * R[3 * b + 1] + R[3 * a + 2] * I[2] * R[3 * b + 2];
    double ms = mass[i];
    r[0] = T[0] + ms * (dif[1] * dif[1] + dif[2] * dif[2]);
    r[1] = T[4] + ms * (dif[0] * dif[0] + dif[2] * dif[2]);
    r[2] = T[8] + ms * (dif[0] * dif[0] + dif[1] * dif[1]);
    r[3] = T[1] - ms * dif[0] * dif[1];
    r[4] = T[2] - ms * dif[0] * dif[2];
    r[5] = T[5] - ms * dif[1] * dif[2];
    r[6] = ms * dif[0]; r[7] = ms * dif[1]; r[8] = ms * dif[2]; r[9] = ms;
  }
  /* cdof: mju_dofCom */
  for (int j = 0; j < m->njnt; j++) {
    int b = jbody[j], da = jdof[j];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = d->subtree_com[3 * rootid[b] + k] - d->xanchor[3 * j + k];
    if (jtype[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        double* c = d->cdof + 6 * (da + k);
        memset(c, 0, 6 * sizeof(double));
        c[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) {
        double* c = d->cdof + 6 * (da + 3 + k);
        double ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        memcpy(c, ax, sizeof ax);
        cross3(c + 3, ax, off);
      }
    } else if (jtype[j] == JNT_SLIDE) {
      double* c = d->cdof + 6 * da;
      c[0] = c[1] = c[2] = 0;
      memcpy(c + 3, d->xaxis + 3 * j, 3 * sizeof(double));
    } else {
      double* c = d->cdof + 6 * da;
      memcpy(c, d->xaxis + 3 * j, 3 * sizeof(double));
      cross3(c + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* ------------------------------------------------------------------ mj_crb (dense M) + Cholesky */
static void crb(OData* d) {
  const OModel* m = &d->m;
  const int32_t *parent = IF(m, LHW_IF_BODY_PARENTID), *dbody = IF(m, LHW_IF_DOF_BODYID), *dparent = IF(m, LHW_IF_DOF_PARENTID);
  const double* arm = DF(m, LHW_DF_DOF_ARMATURE);
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * nb);
  for (int i = nb - 1; i > 0; i--)
    if (parent[i] > 0)
      for (int k = 0; k < 10; k++) d->crb[10 * parent[i] + k] += d->crb[10 * i + k];
  memset(d->M, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mulInertVec(buf, d->crb + 10 * dbody[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = dparent[j]) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += d->cdof[6 * j + k] * buf[k];
      if (j == i) s += arm[i];
      d->M[i * nv + j] = d->M[j * nv + i] = s;
    }
  }
}

/* dense Cholesky A = L L^T (lower), in place on L (n x n, row-major); returns rank deficiency count */
static int cholFactor(double* L, int n) {
  int bad = 0;
  for (int j = 0; j < n; j++) {
    double s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (s < MINVAL) { s = MINVAL; bad++; }
    double dj = sqrt(s);
    L[j * n + j] = dj;
    for (int i = j + 1; i < n; i++) {
      double t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / dj;
    }
  }
  return bad;
}
static void cholSolve(double* x, const double* L, const double* b, int n) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* ------------------------------------------------------------------ Jacobians (mj_jac) */
/* translational Jacobian (3 x nv, row-major) of a world point attached to body */
static void jacPoint(const OData* d, double* jacp, const double* point, int body) {
  const OModel* m = &d->m;
  const int32_t *rootid = IF(m, LHW_IF_BODY_ROOTID), *parent = IF(m, LHW_IF_BODY_PARENTID);
  const int32_t *bdofadr = IF(m, LHW_IF_BODY_DOFADR), *bdofnum = IF(m, LHW_IF_BODY_DOFNUM), *dparent = IF(m, LHW_IF_DOF_PARENTID);
  int nv = m->nv;
  memset(jacp, 0, sizeof(double) * 3 * nv);
  while (body > 0 && bdofnum[body] == 0) body = parent[body];
  if (body <= 0) return;
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - d->subtree_com[3 * rootid[body] + k];
  int i = bdofadr[body] + bdofnum[body] - 1;
  while (i >= 0) {
    const double* c = d->cdof + 6 * i;
    double t[3];
    cross3(t, c, off); /* cdof_rot x offset */
    jacp[0 * nv + i] = c[3] + t[0];
    jacp[1 * nv + i] = c[4] + t[1];
    jacp[2 * nv + i] = c[5] + t[2];
    i = dparent[i];
  }
}

/* ------------------------------------------------------------------ collision (engine_collision_primitive.c) */
static void makeFrame(double* f) { /* mju_makeFrame */
  normalize3(f);
  if (norm3(f + 3) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double t = dot3(f, f + 3);
  for (int k = 0; k < 3; k++) f[3 + k] -= t * f[k];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

typedef struct { double dist, pos[3], frame[6]; } RawCon;

static int planeSphere(RawCon* c, const double* p1, const double* R1, const double* p2, double r, double margin) {
  double n[3] = {R1[2], R1[5], R1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->frame[k] = n[k]; c->frame[3 + k] = 0; c->pos[k] = p2[k] - n[k] * (r + 0.5 * dist); }
  return 1;
}
static int planeCapsule(RawCon* c, const double* p1, const double* R1, const double* p2, const double* R2, const double* size, double margin) {
  double axis[3] = {R2[2], R2[5], R2[8]}, e[3];
  int n = 0;
  for (int s = 1; s >= -1; s -= 2) {
    for (int k = 0; k < 3; k++) e[k] = p2[k] + s * axis[k] * size[1];
    int got = planeSphere(c + n, p1, R1, e, size[0], margin);
    if (got) { memcpy(c[n].frame + 3, axis, sizeof axis); n++; } /* align tangent with capsule axis [MJ-recall] */
  }
  return n;
}
static int planeBox(RawCon* c, const double* p1, const double* R1, const double* p2, const double* R2, const double* size, double margin) {
  double n[3] = {R1[2], R1[5], R1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double dist = dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double v[3] = {(i & 1 ? size[0] : -size[0]), (i & 2 ? size[1] : -size[1]), (i & 4 ? size[2] : -size[2])}, corner[3];
    mulMatVec3(corner, R2, v);
    double ld = dot3(n, corner);
    if (dist + ld > margin || ld > 0) continue;
    c[cnt].dist = dist + ld;
    for (int k = 0; k < 3; k++) {
      c[cnt].frame[k] = n[k]; c[cnt].frame[3 + k] = 0;
      c[cnt].pos[k] = corner[k] + p2[k] - n[k] * c[cnt].dist * 0.5;
    }
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
static int sphereSphereRaw(RawCon* c, const double* p1, double r1, const double* p2, double r2, double margin) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double cd = norm3(dif), dist = cd - r1 - r2;
  if (dist > margin) return 0;
  c->dist = dist;
  if (cd < MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; } else { dif[0] /= cd; dif[1] /= cd; dif[2] /= cd; }
  for (int k = 0; k < 3; k++) { c->frame[k] = dif[k]; c->frame[3 + k] = 0; c->pos[k] = p1[k] + dif[k] * (r1 + 0.5 * dist); }
  return 1;
}
static int sphereCapsule(RawCon* c, const double* p1, double r1, const double* p2, const double* R2, const double* size2, double margin) {
  double axis[3] = {R2[2], R2[5], R2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double x = dot3(axis, vec);
  if (x > size2[1]) x = size2[1];
  if (x < -size2[1]) x = -size2[1];
  double q[3] = {p2[0] + axis[0] * x, p2[1] + axis[1] * x, p2[2] + axis[2] * x};
  return sphereSphereRaw(c, p1, r1, q, size2[0], margin);
}
/* mjc_PlaneCylinder [MJ-recall]: the rim point of the cap nearer the plane that lies deepest (along the plane normal with its axial part
 * removed), the corresponding rim point of the other cap, and -- when the near cap is (nearly) flat on the plane -- two more points of the
 * near rim at +-120 degrees from the first; up to four contacts, all with the plane's normal.  size2 = (radius, half length). */
/* mjc_PlaneConvex for an ellipsoid [MJ-recall]: the support point of the ellipsoid in the direction -normal (analytic: in the geom frame
 * s_i^2 d_i / |s . d| for semi-axes s and direction d), its signed distance to the plane, one contact half way between the point and the plane. */
static int planeEllipsoid(RawCon* c, const double* p1, const double* R1, const double* p2, const double* R2, const double* size2, double margin) {
  const double n[3] = {R1[2], R1[5], R1[8]};
  double dl[3], v[3], sup[3];
  for (int k = 0; k < 3; k++) dl[k] = -(R2[k] * n[0] + R2[3 + k] * n[1] + R2[6 + k] * n[2]);        /* R2^T (-n) */
  for (int k = 0; k < 3; k++) v[k] = size2[k] * dl[k];
  const double len = sqrt(dot3(v, v));
  for (int k = 0; k < 3; k++) v[k] = size2[k] * v[k] / len;                                         /* support point, geom frame */
  for (int k = 0; k < 3; k++) sup[k] = p2[k] + R2[3 * k] * v[0] + R2[3 * k + 1] * v[1] + R2[3 * k + 2] * v[2];
  const double dist = (sup[0] - p1[0]) * n[0] + (sup[1] - p1[1]) * n[1] + (sup[2] - p1[2]) * n[2];
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->pos[k] = sup[k] - n[k] * dist * 0.5; c->frame[k] = n[k]; c->frame[3 + k] = 0; }
  return 1;
}
static int planeCylinder(RawCon* c, const double* p1, const double* R1, const double* p2, const double* R2, const double* size2, double margin) {
  double n[3] = {R1[2], R1[5], R1[8]}, axis[3] = {R2[2], R2[5], R2[8]}, vec[3], dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double prjaxis = dot3(n, axis);
  if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }   /* the axis points towards the plane */
  const double dist0 = dot3(dif, n);
  for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - n[k];                              /* -normal with its axial component removed */
  const double len_sqr = dot3(vec, vec);
  if (len_sqr >= MINVAL * MINVAL) { const double scl = size2[0] / sqrt(len_sqr); for (int k = 0; k < 3; k++) vec[k] *= scl; }
  else { vec[0] = R2[0] * size2[0]; vec[1] = R2[3] * size2[0]; vec[2] = R2[6] * size2[0]; }     /* disk parallel to the plane: the cylinder's x axis */
  const double prjvec = dot3(vec, n);
  for (int k = 0; k < 3; k++) axis[k] *= size2[1];
  prjaxis *= size2[1];
  int cnt = 0;
  if (dist0 + prjaxis + prjvec <= margin) {
    c[cnt].dist = dist0 + prjaxis + prjvec;
    for (int k = 0; k < 3; k++) { c[cnt].pos[k] = p2[k] + vec[k] + axis[k] - n[k] * c[cnt].dist * 0.5; c[cnt].frame[k] = n[k]; c[cnt].frame[3 + k] = 0; }
    cnt++;
  } else return 0;                                                                            /* the nearest point is beyond the margin */
  if (dist0 - prjaxis + prjvec <= margin) {
    c[cnt].dist = dist0 - prjaxis + prjvec;
    for (int k = 0; k < 3; k++) { c[cnt].pos[k] = p2[k] + vec[k] - axis[k] - n[k] * c[cnt].dist * 0.5; c[cnt].frame[k] = n[k]; c[cnt].frame[3 + k] = 0; }
    cnt++;
  }
  const double prjvec1 = -prjvec * 0.5;
  if (dist0 + prjaxis + prjvec1 <= margin) {                                                    /* triangle points on the near cap */
    double vec1[3];
    cross3(vec1, vec, axis);
    normalize3(vec1);
    for (int k = 0; k < 3; k++) vec1[k] *= size2[0] * sqrt(3.0) / 2;
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      c[cnt].dist = dist0 + prjaxis + prjvec1;
      for (int k = 0; k < 3; k++) {
        c[cnt].pos[k] = p2[k] + sgn * vec1[k] + axis[k] - 0.5 * vec[k] - n[k] * c[cnt].dist * 0.5;
        c[cnt].frame[k] = n[k]; c[cnt].frame[3 + k] = 0;
      }
      cnt++;
    }
  }
  return cnt;
}
/* mjc_SphereCylinder [MJ-recall]: the sphere's centre relative to the cylinder decides the case -- beside the lateral surface (a sphere
 * against the axis point at its height, radius = the cylinder's), over a cap (a sphere against the cap's plane), or off the rim (a sphere
 * against the nearest rim point).  One contact, normal from the sphere (geom1) to the cylinder. */
static int sphereCylinder(RawCon* c, const double* p1, double r1, const double* p2, const double* R2, const double* size2, double margin) {
  const double axis[3] = {R2[2], R2[5], R2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const double x = dot3(vec, axis), rad = size2[0], hl = size2[1];
  double a[3];
  for (int k = 0; k < 3; k++) a[k] = vec[k] - axis[k] * x;
  const double a_sqr = dot3(a, a);
  if (x >= -hl && x <= hl) {                                    /* side */
    double q[3] = {p2[0] + axis[0] * x, p2[1] + axis[1] * x, p2[2] + axis[2] * x};
    return sphereSphereRaw(c, p1, r1, q, rad, margin);
  }
  const double sg = x > 0 ? 1.0 : -1.0;
  if (a_sqr <= rad * rad) {                                     /* cap: the plane of the nearer flat face */
    const double dist = fabs(x) - hl - r1;
    if (dist > margin) return 0;
    c->dist = dist;
    for (int k = 0; k < 3; k++) { c->frame[k] = -sg * axis[k]; c->frame[3 + k] = 0; c->pos[k] = p1[k] - sg * axis[k] * (r1 + 0.5 * dist); }
    return 1;
  }
  const double sc = rad / sqrt(a_sqr);                          /* rim */
  double q[3];
  for (int k = 0; k < 3; k++) q[k] = p2[k] + axis[k] * sg * hl + a[k] * sc;
  return sphereSphereRaw(c, p1, r1, q, 0.0, margin);
}
/* mjc_CapsuleCapsule: closest points of two segments; parallel case gives up to two contacts [MJ-recall] */
static int capsuleCapsule(RawCon* c, const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2, double margin) {
  double a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
  double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
  double u = -dot3(a1, dif), v = dot3(a2, dif);
  double det = ma * mc - mb * mb;
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * s1[1]) / mc; }
    else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb * s1[1]) / mc; }
    if (x2 > s2[1]) { x2 = s2[1]; x1 = (u - mb * s2[1]) / ma; if (x1 > s1[1]) x1 = s1[1]; else if (x1 < -s1[1]) x1 = -s1[1]; }
    else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = (u + mb * s2[1]) / ma; if (x1 > s1[1]) x1 = s1[1]; else if (x1 < -s1[1]) x1 = -s1[1]; }
    double q1[3], q2[3];
    for (int k = 0; k < 3; k++) { q1[k] = p1[k] + a1[k] * x1; q2[k] = p2[k] + a2[k] * x2; }
    return sphereSphereRaw(c, q1, s1[0], q2, s2[0], margin);
  }
  /* parallel axes: test the ends of each segment against the other */
  int n = 0;
  double q1[3], q2[3], x;
  for (int s = -1; s <= 1 && n < 2; s += 2) {
    for (int k = 0; k < 3; k++) q1[k] = p1[k] + s * a1[k] * s1[1];
    double vv[3] = {q1[0] - p2[0], q1[1] - p2[1], q1[2] - p2[2]};
    x = dot3(a2, vv);
    if (x >= -s2[1] && x <= s2[1]) {
      for (int k = 0; k < 3; k++) q2[k] = p2[k] + a2[k] * x;
      n += sphereSphereRaw(c + n, q1, s1[0], q2, s2[0], margin);
    }
  }
  for (int s = -1; s <= 1 && n < 2; s += 2) {
    for (int k = 0; k < 3; k++) q2[k] = p2[k] + s * a2[k] * s2[1];
    double vv[3] = {q2[0] - p1[0], q2[1] - p1[1], q2[2] - p1[2]};
    x = dot3(a1, vv);
    if (x >= -s1[1] && x <= s1[1]) {
      for (int k = 0; k < 3; k++) q1[k] = p1[k] + a1[k] * x;
      n += sphereSphereRaw(c + n, q1, s1[0], q2, s2[0], margin);
    }
  }
  return n;
}

/* Sphere-box (MuJoCo: mjc_SphereBox, engine_collision_box.c) [MJ-recall]: the sphere centre in the box frame is clamped to the
 * box; outside, the contact is along (centre - clamped point); inside, the sphere leaves through the nearest face.  The
 * normal points from geom1 (the sphere) to geom2 (the box); the position is midway between the two surfaces. */
static int sphereBox(RawCon* c, const double* p1, double r, const double* p2, const double* R2, const double* size, double margin) {
  double d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]}, ctr[3], cl[3], v[3], nl[3], pl[3];
  mulMatTVec3(ctr, R2, d);
  for (int k = 0; k < 3; k++) { cl[k] = ctr[k] > size[k] ? size[k] : (ctr[k] < -size[k] ? -size[k] : ctr[k]); v[k] = ctr[k] - cl[k]; }
  double dist = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), pen;
  if (dist > MINVAL) {
    pen = dist - r;
    if (pen > margin) return 0;
    for (int k = 0; k < 3; k++) { nl[k] = v[k] / dist; pl[k] = cl[k] + nl[k] * (0.5 * pen); }
  } else {
    int kf = 0;
    double depth = size[0] - fabs(ctr[0]);
    for (int k = 1; k < 3; k++) if (size[k] - fabs(ctr[k]) < depth) { depth = size[k] - fabs(ctr[k]); kf = k; }
    const double sg = ctr[kf] >= 0 ? 1.0 : -1.0;
    pen = -depth - r;
    if (pen > margin) return 0;
    for (int k = 0; k < 3; k++) { nl[k] = 0; pl[k] = ctr[k]; }
    nl[kf] = sg; pl[kf] = sg * size[kf];
    for (int k = 0; k < 3; k++) pl[k] += nl[k] * (0.5 * pen);
  }
  double nw[3], pw[3];
  mulMatVec3(nw, R2, nl);
  mulMatVec3(pw, R2, pl);
  c->dist = pen;
  for (int k = 0; k < 3; k++) { c->frame[k] = -nw[k]; c->frame[3 + k] = 0; c->pos[k] = p2[k] + pw[k]; }
  return 1;
}
/* d/dt of the squared distance from the point c0 + t al (box frame) to the box, halved: nondecreasing in t (the distance to
 * a convex set along a line is convex) */
static double segBoxSlope(const double* c0, const double* al, const double* size, double t) {
  double g = 0;
  for (int k = 0; k < 3; k++) {
    const double x = c0[k] + t * al[k];
    if (x > size[k]) g += al[k] * (x - size[k]);
    else if (x < -size[k]) g += al[k] * (x + size[k]);
  }
  return g;
}
/* Capsule-box.  NOT a restatement of MuJoCo's mjc_CapsuleBox (a long case analysis whose source could not be consulted).
 * The squared distance from a point of the capsule's segment to the box is convex along the segment; its minimisers form an
 * interval [t_lo, t_hi], found by two bisections (60 halvings each = to the last bit) on the monotone slope.  Contacts are
 * sphere-box contacts: if both segment ends touch, the two ends (a capsule lying on a face rests on its ends, as in MuJoCo);
 * otherwise at t_lo, at t_hi if it is a different point (the capsule leaves a face over an edge), and at a touching end that
 * is not one of those points.  At most 2 contacts.  The HIP kernel implements exactly this (csrc/lhw_humanoid.hip). */
#define CAPBOX_TOL 1e-9    /* two segment parameters closer than this are one point */
#define CAPBOX_FLAT 1e-12  /* |slope| below this counts as zero (a segment parallel to a face up to rounding) */
static int capsuleBox(RawCon* c, const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2, double margin) {
  const double axis[3] = {R1[2], R1[5], R1[8]}, h = s1[1];
  double d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]}, c0[3], al[3];
  mulMatTVec3(c0, R2, d);
  mulMatTVec3(al, R2, axis);
  RawCon em, ep;
  double q[3];
  for (int k = 0; k < 3; k++) q[k] = p1[k] - axis[k] * h;
  const int gm = sphereBox(&em, q, s1[0], p2, R2, s2, margin);
  for (int k = 0; k < 3; k++) q[k] = p1[k] + axis[k] * h;
  const int gp = sphereBox(&ep, q, s1[0], p2, R2, s2, margin);
  if (gm && gp) { c[0] = em; c[1] = ep; return 2; }
  const double sm = segBoxSlope(c0, al, s2, -h), sp = segBoxSlope(c0, al, s2, h);
  double tlo, thi;
  if (sm >= -CAPBOX_FLAT) tlo = -h;
  else if (sp < -CAPBOX_FLAT) tlo = h;
  else {
    double lo = -h, hi = h;
    for (int it = 0; it < 60; it++) { const double mid = 0.5 * (lo + hi); if (segBoxSlope(c0, al, s2, mid) >= -CAPBOX_FLAT) hi = mid; else lo = mid; }
    tlo = hi;
  }
  if (sp <= CAPBOX_FLAT) thi = h;
  else if (sm > CAPBOX_FLAT) thi = -h;
  else {
    double lo = -h, hi = h;
    for (int it = 0; it < 60; it++) { const double mid = 0.5 * (lo + hi); if (segBoxSlope(c0, al, s2, mid) <= CAPBOX_FLAT) lo = mid; else hi = mid; }
    thi = lo;
  }
  int n = 0;
  for (int k = 0; k < 3; k++) q[k] = p1[k] + axis[k] * tlo;
  n += sphereBox(c + n, q, s1[0], p2, R2, s2, margin);
  if (thi > tlo + CAPBOX_TOL) {
    for (int k = 0; k < 3; k++) q[k] = p1[k] + axis[k] * thi;
    n += sphereBox(c + n, q, s1[0], p2, R2, s2, margin);
  } else thi = tlo;
  if (n < 2 && gm && tlo > -h + CAPBOX_TOL) c[n++] = em;
  if (n < 2 && gp && thi < h - CAPBOX_TOL) c[n++] = ep;
  return n;
}

/* Box-box: separating-axis test over the 15 candidate axes, then either a face contact (the vertices of the intersection
 * of the incident face with the reference face rectangle, at or below the reference face within `margin`, become contacts,
 * at most 4, deepest first) or a single edge-edge contact.  This is the classical SAT + clipping construction
 * (as in ODE's dBoxBox); it is NOT a restatement of MuJoCo's mjc_BoxBox, whose source could not be consulted -- contact
 * points of the two agree for face-face resting contacts (the case the stair terrain produces) and may differ in count and
 * placement for edge cases.  The HIP kernel implements exactly this algorithm (csrc/lhw_humanoid.hip box_box). */
static int boxBox(RawCon* out, const double* p1, const double* R1, const double* s1, const double* p2, const double* R2,
                  const double* s2, double margin) {
  double A[3][3], B[3][3], d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; B[i][k] = R2[3 * k + i]; } /* axis i = column i */
  double R[3][3], AR[3][3], ta[3], tb[3];
  for (int i = 0; i < 3; i++) {
    ta[i] = dot3(d, A[i]); tb[i] = dot3(d, B[i]);
    for (int j = 0; j < 3; j++) { R[i][j] = dot3(A[i], B[j]); AR[i][j] = fabs(R[i][j]) + 1e-12; }
  }
  /* face axes */
  double best = -1e300; int code = -1; /* code 0..2: A face i, 3..5: B face j, 6..14: edge i*3+j */
  for (int i = 0; i < 3; i++) {
    double s = fabs(ta[i]) - (s1[i] + s2[0] * AR[i][0] + s2[1] * AR[i][1] + s2[2] * AR[i][2]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    double s = fabs(tb[j]) - (s2[j] + s1[0] * AR[0][j] + s1[1] * AR[1][j] + s1[2] * AR[2][j]);
    if (s > margin) return 0;
    if (s > best - 1e-9) { if (s > best) best = s; code = 3 + j; } /* ties go to the faces of geom2 */
  }
  /* edge axes a_i x b_j */
  double ebest = -1e300; int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double len2 = 1.0 - R[i][j] * R[i][j];
      if (len2 < 1e-12) continue;
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double tl = ta[i2] * R[i1][j] - ta[i1] * R[i2][j];
      double ra = s1[i1] * AR[i2][j] + s1[i2] * AR[i1][j], rb = s2[j1] * AR[i][j2] + s2[j2] * AR[i][j1];
      double s = (fabs(tl) - ra - rb) / sqrt(len2);
      if (s > margin) return 0;
      if (s > ebest) { ebest = s; ecode = 6 + 3 * i + j; }
    }
  if (ecode >= 0 && ebest > best + 1e-6) {
    /* ---- edge-edge */
    int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
    double n[3];
    cross3(n, A[i], B[j]);
    normalize3(n);
    if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    double pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
    for (int k = 0; k < 3; k++) {
      if (k != i) { double sg = dot3(n, A[k]) > 0 ? 1.0 : -1.0; for (int a = 0; a < 3; a++) pa[a] += sg * s1[k] * A[k][a]; }
      if (k != j) { double sg = dot3(n, B[k]) > 0 ? -1.0 : 1.0; for (int a = 0; a < 3; a++) pb[a] += sg * s2[k] * B[k][a]; }
    }
    /* closest points of the lines pa + al*A[i], pb + be*B[j] */
    double w[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    double bq = R[i][j], dd = dot3(A[i], w), ee = dot3(B[j], w), den = 1.0 - bq * bq;
    double al = (bq * ee - dd) / den, be = (ee - bq * dd) / den;
    double qa[3], qb[3];
    for (int a = 0; a < 3; a++) { qa[a] = pa[a] + al * A[i][a]; qb[a] = pb[a] + be * B[j][a]; }
    out[0].dist = ebest;
    for (int a = 0; a < 3; a++) { out[0].frame[a] = n[a]; out[0].frame[3 + a] = 0; out[0].pos[a] = 0.5 * (qa[a] + qb[a]); }
    return 1;
  }
  /* ---- face contact: reference box r (owner of the axis), incident box c */
  int refB = code >= 3, ax = refB ? code - 3 : code;
  const double *pr = refB ? p2 : p1, *pc = refB ? p1 : p2, *sr = refB ? s2 : s1, *sc = refB ? s1 : s2;
  double (*Ar)[3] = refB ? B : A, (*Ac)[3] = refB ? A : B;
  double n[3] = {Ar[ax][0], Ar[ax][1], Ar[ax][2]};           /* geom1 -> geom2 */
  if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  double nr[3] = {refB ? -n[0] : n[0], refB ? -n[1] : n[1], refB ? -n[2] : n[2]}; /* outward normal of the reference face */
  double fc[3];
  for (int a = 0; a < 3; a++) fc[a] = pr[a] + nr[a] * sr[ax];
  /* incident face: most anti-parallel to nr */
  int kc = 0; double bestdot = -1;
  for (int k = 0; k < 3; k++) { double v = fabs(dot3(nr, Ac[k])); if (v > bestdot) { bestdot = v; kc = k; } }
  double sgn = dot3(nr, Ac[kc]) > 0 ? -1.0 : 1.0;
  int ku = (kc + 1) % 3, kv = (kc + 2) % 3;
  /* incident face vertices (a cycle), in the reference-face frame: origin fc, axes u, v, outward normal nr */
  int ru = (ax + 1) % 3, rv = (ax + 2) % 3;
  const double *u = Ar[ru], *v = Ar[rv];
  double ha = sr[ru], hb = sr[rv];
  double P[4][3], px[4], py[4], pd[4];
  for (int q = 0; q < 4; q++) {
    double su = (q == 0 || q == 3) ? 1.0 : -1.0, sv = (q < 2) ? 1.0 : -1.0, w0[3];
    for (int a = 0; a < 3; a++) {
      P[q][a] = pc[a] + sgn * sc[kc] * Ac[kc][a] + su * sc[ku] * Ac[ku][a] + sv * sc[kv] * Ac[kv][a];
      w0[a] = P[q][a] - fc[a];
    }
    px[q] = dot3(w0, u); py[q] = dot3(w0, v); pd[q] = dot3(w0, nr);
  }
  /* Candidate contact points = vertices of (incident face) n (reference face rectangle), enumerated in a fixed order:
   * (i) incident vertices inside the rectangle, (ii) rectangle corners inside the incident parallelogram (lifted onto the
   * incident plane), (iii) incident edge x rectangle side crossings.  The four deepest (depth <= margin) are kept by
   * bubbling each candidate through a 4-slot list sorted by depth (strict <: earlier candidates win ties). */
  double bd[4] = {1e300, 1e300, 1e300, 1e300}, bp[4][3] = {{0}};
#define BB_PUSH(DEPTH, X0, X1, X2)                                              \
  do {                                                                          \
    double nd_ = (DEPTH), n0_ = (X0), n1_ = (X1), n2_ = (X2), t_;               \
    if (nd_ <= margin)                                                          \
      for (int k_ = 0; k_ < 4; k_++)                                            \
        if (nd_ < bd[k_]) {                                                     \
          t_ = bd[k_]; bd[k_] = nd_; nd_ = t_;                                  \
          t_ = bp[k_][0]; bp[k_][0] = n0_; n0_ = t_;                            \
          t_ = bp[k_][1]; bp[k_][1] = n1_; n1_ = t_;                            \
          t_ = bp[k_][2]; bp[k_][2] = n2_; n2_ = t_;                            \
        }                                                                       \
  } while (0)
  for (int q = 0; q < 4; q++)
    if (fabs(px[q]) <= ha && fabs(py[q]) <= hb) BB_PUSH(pd[q], P[q][0], P[q][1], P[q][2]);
  {
    double e1x = px[1] - px[0], e1y = py[1] - py[0], e2x = px[3] - px[0], e2y = py[3] - py[0];
    double det = e1x * e2y - e1y * e2x;
    if (fabs(det) > 1e-14)
      for (int c = 0; c < 4; c++) {
        double cx = (c == 0 || c == 3) ? ha : -ha, cy = (c < 2) ? hb : -hb;
        double al = ((cx - px[0]) * e2y - (cy - py[0]) * e2x) / det, be = (e1x * (cy - py[0]) - e1y * (cx - px[0])) / det;
        if (al >= 0 && al <= 1 && be >= 0 && be <= 1) {
          double dep = pd[0] + al * (pd[1] - pd[0]) + be * (pd[3] - pd[0]);
          BB_PUSH(dep, fc[0] + cx * u[0] + cy * v[0] + dep * nr[0], fc[1] + cx * u[1] + cy * v[1] + dep * nr[1],
                  fc[2] + cx * u[2] + cy * v[2] + dep * nr[2]);
        }
      }
  }
  for (int q = 0; q < 4; q++) {
    int q1 = (q + 1) & 3;
    double dx = px[q1] - px[q], dy = py[q1] - py[q], dd = pd[q1] - pd[q];
    for (int sd = 0; sd < 4; sd++) {
      double tt, other, lim;
      if (sd < 2) {
        if (dx == 0) continue;
        tt = ((sd == 0 ? ha : -ha) - px[q]) / dx; other = py[q] + tt * dy; lim = hb;
      } else {
        if (dy == 0) continue;
        tt = ((sd == 2 ? hb : -hb) - py[q]) / dy; other = px[q] + tt * dx; lim = ha;
      }
      if (tt > 0 && tt < 1 && fabs(other) < lim)
        BB_PUSH(pd[q] + tt * dd, P[q][0] + tt * (P[q1][0] - P[q][0]), P[q][1] + tt * (P[q1][1] - P[q][1]), P[q][2] + tt * (P[q1][2] - P[q][2]));
    }
  }
#undef BB_PUSH
  int cnt = 0;
  for (int q = 0; q < 4; q++)
    if (bd[q] < 1e299) {
      out[cnt].dist = bd[q];
      for (int a = 0; a < 3; a++) {
        out[cnt].frame[a] = n[a]; out[cnt].frame[3 + a] = 0;
        out[cnt].pos[a] = bp[q][a] - nr[a] * bd[q] * 0.5;
      }
      cnt++;
    }
  return cnt;
}

/* mj_collision + mj_setContact parameter mixing */
static void collision(OData* d) {
  const OModel* m = &d->m;
  const int32_t *g1a = IF(m, LHW_IF_PAIR_GEOM1), *g2a = IF(m, LHW_IF_PAIR_GEOM2), *gtype = IF(m, LHW_IF_GEOM_TYPE);
  const int32_t *gcondim = IF(m, LHW_IF_GEOM_CONDIM), *gprio = IF(m, LHW_IF_GEOM_PRIORITY);
  const double *gsize = DF(m, LHW_DF_GEOM_SIZE), *gmargin = DF(m, LHW_DF_GEOM_MARGIN), *ggap = DF(m, LHW_DF_GEOM_GAP);
  const double *gfri = DF(m, LHW_DF_GEOM_FRICTION), *gsolmix = DF(m, LHW_DF_GEOM_SOLMIX), *gsolref = DF(m, LHW_DF_GEOM_SOLREF);
  const double* gsolimp = DF(m, LHW_DF_GEOM_SOLIMP);
  d->ncon = 0;
  d->warning_contactfull = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = g1a[p], g2 = g2a[p], t1 = gtype[g1], t2 = gtype[g2];
    double margin = fmax(gmargin[g1], gmargin[g2]), gap = fmax(ggap[g1], ggap[g2]);
    const double *p1 = d->geom_xpos + 3 * g1, *R1 = d->geom_xmat + 9 * g1, *s1 = gsize + 3 * g1;
    const double *p2 = d->geom_xpos + 3 * g2, *R2 = d->geom_xmat + 9 * g2, *s2 = gsize + 3 * g2;
    RawCon rc[4];
    int n = 0;
    if (t1 == GEOM_PLANE && t2 == GEOM_SPHERE) n = planeSphere(rc, p1, R1, p2, s2[0], margin);
    else if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE) n = planeCapsule(rc, p1, R1, p2, R2, s2, margin);
    else if (t1 == GEOM_PLANE && t2 == GEOM_BOX) n = planeBox(rc, p1, R1, p2, R2, s2, margin);
    else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) n = sphereSphereRaw(rc, p1, s1[0], p2, s2[0], margin);
    else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) n = sphereCapsule(rc, p1, s1[0], p2, R2, s2, margin);
    else if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) n = capsuleCapsule(rc, p1, R1, s1, p2, R2, s2, margin);
    else if (t1 == GEOM_PLANE && t2 == GEOM_CYLINDER) n = planeCylinder(rc, p1, R1, p2, R2, s2, margin);
    else if (t1 == GEOM_PLANE && t2 == GEOM_ELLIPSOID) n = planeEllipsoid(rc, p1, R1, p2, R2, s2, margin);
    else if (t1 == GEOM_SPHERE && t2 == GEOM_CYLINDER) n = sphereCylinder(rc, p1, s1[0], p2, R2, s2, margin);
    else if (t1 == GEOM_SPHERE && t2 == GEOM_BOX) n = sphereBox(rc, p1, s1[0], p2, R2, s2, margin);
    else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX) n = capsuleBox(rc, p1, R1, s1, p2, R2, s2, margin);
    else if (t1 == GEOM_BOX && t2 == GEOM_BOX) n = boxBox(rc, p1, R1, s1, p2, R2, s2, margin);
    for (int i = 0; i < n; i++) {
      if (d->ncon >= MAXCON) { d->warning_contactfull = 1; return; }
      OContact* c = d->contact + d->ncon++;
      c->dist = rc[i].dist;
      memcpy(c->pos, rc[i].pos, sizeof c->pos);
      memcpy(c->frame, rc[i].frame, 6 * sizeof(double));
      makeFrame(c->frame);
      c->geom1 = g1; c->geom2 = g2;
      c->includemargin = margin - gap;
      c->exclude = (c->dist >= c->includemargin);
      /* parameter mixing (mj_contactParam): priority, then solmix-weighted average; max friction/condim */
      double mix;
      if (gprio[g1] != gprio[g2]) {
        int g = gprio[g1] > gprio[g2] ? g1 : g2;
        c->dim = gcondim[g];
        memcpy(c->solref, gsolref + 2 * g, 2 * sizeof(double));
        memcpy(c->solimp, gsolimp + 5 * g, 5 * sizeof(double));
        c->friction[0] = c->friction[1] = gfri[3 * g]; c->friction[2] = gfri[3 * g + 1];
        c->friction[3] = c->friction[4] = gfri[3 * g + 2];
      } else {
        c->dim = gcondim[g1] > gcondim[g2] ? gcondim[g1] : gcondim[g2];
        double m1 = gsolmix[g1], m2 = gsolmix[g2];
        if (m1 >= MINVAL && m2 >= MINVAL) mix = m1 / (m1 + m2);
        else if (m1 < MINVAL && m2 < MINVAL) mix = 0.5;
        else mix = m1 < MINVAL ? 0.0 : 1.0;
        if (gsolref[2 * g1] > 0 && gsolref[2 * g2] > 0)
          for (int k = 0; k < 2; k++) c->solref[k] = mix * gsolref[2 * g1 + k] + (1 - mix) * gsolref[2 * g2 + k];
        else
          for (int k = 0; k < 2; k++) c->solref[k] = fmin(gsolref[2 * g1 + k], gsolref[2 * g2 + k]);
        for (int k = 0; k < 5; k++) c->solimp[k] = mix * gsolimp[5 * g1 + k] + (1 - mix) * gsolimp[5 * g2 + k];
        double f0 = fmax(gfri[3 * g1], gfri[3 * g2]), f1 = fmax(gfri[3 * g1 + 1], gfri[3 * g2 + 1]), f2 = fmax(gfri[3 * g1 + 2], gfri[3 * g2 + 2]);
        c->friction[0] = c->friction[1] = f0; c->friction[2] = f1; c->friction[3] = c->friction[4] = f2;
      }
      c->efc_address = -1;
    }
  }
}

/* ------------------------------------------------------------------ constraints (engine_core_constraint.c) */
static void getsolparam(const OModel* m, const double* sr, const double* si, double* solref, double* solimp) {
  memcpy(solref, sr, 2 * sizeof(double));
  memcpy(solimp, si, 5 * sizeof(double));
  double h = m->db[LHW_DH_TIMESTEP];
  if (!(m->ib[LHW_IH_DISABLEFLAGS] & DSBL_REFSAFE) && solref[0] > 0 && solref[0] < 2 * h) solref[0] = 2 * h;
  solimp[0] = fmin(MAXIMP, fmax(MINIMP, solimp[0]));
  solimp[1] = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
  solimp[2] = fmax(0, solimp[2]);
  solimp[3] = fmin(MAXIMP, fmax(MINIMP, solimp[3]));
  solimp[4] = fmax(1, solimp[4]);
}
static double getimpedance(const double* solimp, double pos, double margin) {
  if (solimp[0] == solimp[1] || solimp[2] <= MINVAL) return 0.5 * (solimp[0] + solimp[1]);
  double x = fabs((pos - margin) / solimp[2]);
  if (x >= 1) return solimp[1];
  if (x <= 0) return solimp[0];
  double y;
  if (solimp[4] == 1) y = x;
  else if (x <= solimp[3]) y = pow(x, solimp[4]) / pow(solimp[3], solimp[4] - 1);
  else y = 1 - pow(1 - x, solimp[4]) / pow(1 - solimp[3], solimp[4] - 1);
  return solimp[0] + y * (solimp[1] - solimp[0]);
}

static int addRow(OData* d, int type, int id, double pos, double margin, double diagApprox, const double* solref, const double* solimp, double floss) {
  int r = d->nefc;
  if (r >= MAXEFC) return -1;
  d->nefc++;
  d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_margin[r] = margin;
  d->efc_diagApprox[r] = diagApprox; d->efc_frictionloss[r] = floss;
  double sr[2], si[5];
  getsolparam(&d->m, solref, solimp, sr, si);
  double imp = getimpedance(si, pos, margin);
  d->efc_R[r] = fmax(MINVAL, (1 - imp) / imp * diagApprox);
  double K, B;
  if (sr[0] > 0) {
    K = 1 / fmax(MINVAL, si[1] * si[1] * sr[0] * sr[0] * sr[1] * sr[1]);
    B = 2 / fmax(MINVAL, si[1] * sr[0]);
  } else {
    K = -sr[0] / fmax(MINVAL, si[1] * si[1]);
    B = -sr[1] / fmax(MINVAL, si[1]);
  }
  d->efc_KBIP[r][0] = K; d->efc_KBIP[r][1] = B; d->efc_KBIP[r][2] = imp; d->efc_KBIP[r][3] = 0;
  memset(d->efc_J + (size_t)r * d->m.nv, 0, sizeof(double) * d->m.nv);
  return r;
}

/* mj_makeConstraint: friction-loss dofs, joint limits, contacts (pyramidal), then mj_makeImpedance's R adjustment */
static void makeConstraint(OData* d) {
  const OModel* m = &d->m;
  int nv = m->nv;
  const double *floss = DF(m, LHW_DF_DOF_FRICTIONLOSS), *dinvw = DF(m, LHW_DF_DOF_INVWEIGHT0);
  const double *dsolref = DF(m, LHW_DF_DOF_SOLREF), *dsolimp = DF(m, LHW_DF_DOF_SOLIMP);
  d->nefc = 0;
  for (int i = 0; i < nv; i++)
    if (floss[i] > 0) {
      int r = addRow(d, EFC_FRICTION, i, 0, 0, dinvw[i], dsolref + 2 * i, dsolimp + 5 * i, floss[i]);
      if (r >= 0) d->efc_J[(size_t)r * nv + i] = 1;
    }
  d->nf = d->nefc;
  const int32_t *jtype = IF(m, LHW_IF_JNT_TYPE), *jlim = IF(m, LHW_IF_JNT_LIMITED), *jq = IF(m, LHW_IF_JNT_QPOSADR), *jd = IF(m, LHW_IF_JNT_DOFADR);
  const double *jrange = DF(m, LHW_DF_JNT_RANGE), *jsolref = DF(m, LHW_DF_JNT_SOLREF), *jsolimp = DF(m, LHW_DF_JNT_SOLIMP), *jmargin = DF(m, LHW_DF_JNT_MARGIN);
  for (int j = 0; j < m->njnt; j++) {
    if (!jlim[j] || (jtype[j] != JNT_HINGE && jtype[j] != JNT_SLIDE)) continue;
    double value = d->qpos[jq[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (jrange[2 * j + (side + 1) / 2] - value);
      if (dist < jmargin[j]) {
        int r = addRow(d, EFC_LIMIT, j, dist, jmargin[j], dinvw[jd[j]], jsolref + 2 * j, jsolimp + 5 * j, 0);
        if (r >= 0) d->efc_J[(size_t)r * nv + jd[j]] = -side;
      }
    }
  }
  d->nl = d->nefc - d->nf;
  const int32_t* gbody = IF(m, LHW_IF_GEOM_BODYID);
  /* engine_core_constraint.c mj_makeImpedance / diagApprox: body_invweight0 of the two geoms' bodies -- carried per geom in the
   * packed model (geom_invweight0), so that a model whose welded links were folded into their parents (Model.fuse_static)
   * keeps the regulariser of the unfused one */
  const double* ginvw = DF(m, LHW_DF_GEOM_INVWEIGHT0);
  double* jp1 = dalloc(3 * nv);
  double* jp2 = dalloc(3 * nv);
  for (int ci = 0; ci < d->ncon; ci++) {
    OContact* c = d->contact + ci;
    if (c->exclude) continue;
    int b1 = gbody[c->geom1], b2 = gbody[c->geom2];
    jacPoint(d, jp1, c->pos, b1);
    jacPoint(d, jp2, c->pos, b2);
    /* relative-velocity Jacobian in the contact frame: Jc = frame * (J2 - J1) */
    double tran = ginvw[2 * c->geom1] + ginvw[2 * c->geom2];
    if (c->dim == 1) {
      int r = addRow(d, EFC_CONTACT, ci, c->dist, c->includemargin, tran, c->solref, c->solimp, 0);
      if (r < 0) break;
      c->efc_address = r;
      for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int a = 0; a < 3; a++) s += c->frame[a] * (jp2[a * nv + k] - jp1[a * nv + k]);
        d->efc_J[(size_t)r * nv + k] = s;
      }
      continue;
    }
    /* condim 3, pyramidal: rows n + mu1 t1, n - mu1 t1, n + mu2 t2, n - mu2 t2 */
    int first = -1;
    for (int e = 0; e < 4; e++) {
      double mu = c->friction[e / 2];
      double diag = tran + mu * mu * tran;
      int r = addRow(d, EFC_CONTACT, ci, c->dist, c->includemargin, diag, c->solref, c->solimp, 0);
      if (r < 0) break;
      if (e == 0) { first = r; c->efc_address = r; }
      const double* t = c->frame + 3 * (1 + e / 2);
      double sgn = (e & 1) ? -mu : mu;
      for (int k = 0; k < nv; k++) {
        double sn = 0, st = 0;
        for (int a = 0; a < 3; a++) {
          double dj = jp2[a * nv + k] - jp1[a * nv + k];
          sn += c->frame[a] * dj;
          st += t[a] * dj;
        }
        d->efc_J[(size_t)r * nv + k] = sn + sgn * st;
      }
    }
    if (first >= 0 && d->nefc >= first + 4) {
      /* mj_makeImpedance: all pyramid edges share R = 2 mu^2 R(first edge) [MJ-recall] */
      double mu = c->friction[0];
      double Rpy = fmax(MINVAL, 2 * mu * mu * d->efc_R[first]);
      for (int e = 0; e < 4; e++) d->efc_R[first + e] = Rpy;
    }
  }
  free(jp1); free(jp2);
  for (int r = 0; r < d->nefc; r++) d->efc_D[r] = 1 / d->efc_R[r];
}

/* mj_transmission (joint transmissions only) */
static void transmission(OData* d) {
  const OModel* m = &d->m;
  const int32_t *trn = IF(m, LHW_IF_ACTUATOR_TRNID), *jq = IF(m, LHW_IF_JNT_QPOSADR);
  const double* gear = DF(m, LHW_DF_ACTUATOR_GEAR);
  for (int i = 0; i < m->nu; i++) d->actuator_length[i] = gear[i] * d->qpos[jq[trn[i]]];
}

/* ------------------------------------------------------------------ mj_fwdPosition */
static void fwdPosition(OData* d) {
  kinematics(d);
  comPos(d);
  crb(d);
  memcpy(d->L, d->M, sizeof(double) * d->m.nv * d->m.nv);
  cholFactor(d->L, d->m.nv);
  collision(d);
  makeConstraint(d);
  transmission(d);
}

/* ------------------------------------------------------------------ mj_comVel + mj_fwdVelocity */
static void comVel(OData* d) {
  const OModel* m = &d->m;
  const int32_t *parent = IF(m, LHW_IF_BODY_PARENTID), *jadr = IF(m, LHW_IF_BODY_JNTADR), *jnum = IF(m, LHW_IF_BODY_JNTNUM);
  const int32_t *jtype = IF(m, LHW_IF_JNT_TYPE), *jdof = IF(m, LHW_IF_JNT_DOFADR);
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double cv[6];
    memcpy(cv, d->cvel + 6 * parent[i], sizeof cv);
    for (int k = 0; k < jnum[i]; k++) {
      int j = jadr[i] + k, da = jdof[j];
      if (jtype[j] == JNT_FREE) {
        memset(d->cdof_dot + 6 * da, 0, 18 * sizeof(double));
        for (int a = 0; a < 3; a++)
          for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + a) + c] * d->qvel[da + a];
        for (int a = 3; a < 6; a++) crossMotion(d->cdof_dot + 6 * (da + a), cv, d->cdof + 6 * (da + a));
        for (int a = 3; a < 6; a++)
          for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + a) + c] * d->qvel[da + a];
      } else {
        crossMotion(d->cdof_dot + 6 * da, cv, d->cdof + 6 * da);
        for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * da + c] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * i, cv, sizeof cv);
  }
}

/* mj_rne with flg_acc = 0: Coriolis/centrifugal + gravity */
static void rneBias(OData* d) {
  const OModel* m = &d->m;
  const int32_t *parent = IF(m, LHW_IF_BODY_PARENTID), *bdofadr = IF(m, LHW_IF_BODY_DOFADR), *bdofnum = IF(m, LHW_IF_BODY_DOFNUM);
  const int32_t* dbody = IF(m, LHW_IF_DOF_BODYID);
  int nb = m->nbody, nv = m->nv;
  memset(d->cacc, 0, 6 * sizeof(double));
  d->cacc[3] = -m->db[LHW_DH_GRAVITY_X]; d->cacc[4] = -m->db[LHW_DH_GRAVITY_Y]; d->cacc[5] = -m->db[LHW_DH_GRAVITY_Z];
  for (int i = 1; i < nb; i++) {
    double *ca = d->cacc + 6 * i, t[6], t2[6];
    memcpy(ca, d->cacc + 6 * parent[i], 6 * sizeof(double));
    for (int k = 0; k < bdofnum[i]; k++) {
      int da = bdofadr[i] + k;
      for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[6 * da + c] * d->qvel[da];
    }
    mulInertVec(t, d->cinert + 10 * i, ca);
    mulInertVec(t2, d->cinert + 10 * i, d->cvel + 6 * i);
    crossForce(d->cfrc + 6 * i, d->cvel + 6 * i, t2);
    for (int c = 0; c < 6; c++) d->cfrc[6 * i + c] += t[c];
  }
  memset(d->cfrc, 0, 6 * sizeof(double));
  for (int i = nb - 1; i > 0; i--)
    if (parent[i] > 0)
      for (int c = 0; c < 6; c++) d->cfrc[6 * parent[i] + c] += d->cfrc[6 * i + c];
  for (int i = 0; i < nv; i++) {
    double s = 0;
    for (int c = 0; c < 6; c++) s += d->cdof[6 * i + c] * d->cfrc[6 * dbody[i] + c];
    d->qfrc_bias[i] = s;
  }
}

static void fwdVelocity(OData* d) {
  const OModel* m = &d->m;
  int nv = m->nv;
  const int32_t *trn = IF(m, LHW_IF_ACTUATOR_TRNID), *jd = IF(m, LHW_IF_JNT_DOFADR);
  const double *gear = DF(m, LHW_DF_ACTUATOR_GEAR), *damping = DF(m, LHW_DF_DOF_DAMPING);
  for (int i = 0; i < m->nu; i++) d->actuator_velocity[i] = gear[i] * d->qvel[jd[trn[i]]];
  comVel(d);
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -damping[i] * d->qvel[i];
  rneBias(d);
  /* mj_referenceConstraint */
  for (int r = 0; r < d->nefc; r++) {
    double v = 0;
    for (int k = 0; k < nv; k++) v += d->efc_J[(size_t)r * nv + k] * d->qvel[k];
    d->efc_vel[r] = v;
    d->efc_aref[r] = -d->efc_KBIP[r][1] * v - d->efc_KBIP[r][0] * d->efc_KBIP[r][2] * (d->efc_pos[r] - d->efc_margin[r]);
  }
}

/* mj_fwdActuation: motors, gain 1, no bias/dynamics; ctrl/force clamping */
static void fwdActuation(OData* d, int enabled) {
  const OModel* m = &d->m;
  const int32_t *trn = IF(m, LHW_IF_ACTUATOR_TRNID), *jd = IF(m, LHW_IF_JNT_DOFADR);
  const int32_t *climited = IF(m, LHW_IF_ACTUATOR_CTRLLIMITED), *flimited = IF(m, LHW_IF_ACTUATOR_FORCELIMITED);
  const double *gear = DF(m, LHW_DF_ACTUATOR_GEAR), *crange = DF(m, LHW_DF_ACTUATOR_CTRLRANGE), *frange = DF(m, LHW_DF_ACTUATOR_FORCERANGE);
  memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
  if (!enabled) { memset(d->actuator_force, 0, sizeof(double) * m->nu); return; }
  for (int i = 0; i < m->nu; i++) {
    double c = d->ctrl[i];
    if (climited[i]) c = fmin(crange[2 * i + 1], fmax(crange[2 * i], c));
    double f = c;
    if (flimited[i]) f = fmin(frange[2 * i + 1], fmax(frange[2 * i], f));
    d->actuator_force[i] = f;
    d->qfrc_actuator[jd[trn[i]]] += gear[i] * f;
  }
}

/* mj_fwdAcceleration (+ mj_xfrcAccumulate) */
static void fwdAcceleration(OData* d) {
  const OModel* m = &d->m;
  int nv = m->nv;
  memset(d->qfrc_applied, 0, sizeof(double) * nv);
  double* jp = NULL;
  for (int b = 1; b < m->nbody; b++) {
    const double* x = d->xfrc_applied + 6 * b;
    if (x[0] == 0 && x[1] == 0 && x[2] == 0 && x[3] == 0 && x[4] == 0 && x[5] == 0) continue;
    if (!jp) jp = dalloc(3 * nv);
    /* force at body com (xipos) + torque: qfrc += Jp^T f + Jr^T tau */
    jacPoint(d, jp, d->xipos + 3 * b, b);
    for (int k = 0; k < nv; k++) d->qfrc_applied[k] += jp[k] * x[0] + jp[nv + k] * x[1] + jp[2 * nv + k] * x[2];
    const int32_t *parent = IF(m, LHW_IF_BODY_PARENTID), *bdofadr = IF(m, LHW_IF_BODY_DOFADR), *bdofnum = IF(m, LHW_IF_BODY_DOFNUM), *dparent = IF(m, LHW_IF_DOF_PARENTID);
    int bb = b;
    while (bb > 0 && bdofnum[bb] == 0) bb = parent[bb];
    if (bb > 0)
      for (int i = bdofadr[bb] + bdofnum[bb] - 1; i >= 0; i = dparent[i])
        d->qfrc_applied[i] += d->cdof[6 * i] * x[3] + d->cdof[6 * i + 1] * x[4] + d->cdof[6 * i + 2] * x[5];
  }
  free(jp);
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
  cholSolve(d->qacc_smooth, d->L, d->qfrc_smooth, nv);
}

/* ------------------------------------------------------------------ solver (engine_solver.c, Newton) */
/* mj_constraintUpdate: force/state/cost from jar = J qacc - aref */
static double constraintUpdate(OData* d, const double* jar, int set) {
  double cost = 0;
  for (int r = 0; r < d->nefc; r++) {
    double D = d->efc_D[r], R = d->efc_R[r], f, x = jar[r];
    int state;
    if (d->efc_type[r] == EFC_FRICTION) {
      double fl = d->efc_frictionloss[r];
      if (x <= -R * fl) { f = fl; cost += -0.5 * R * fl * fl - fl * x; state = 2; }
      else if (x >= R * fl) { f = -fl; cost += -0.5 * R * fl * fl + fl * x; state = 3; }
      else { f = -D * x; cost += 0.5 * D * x * x; state = 1; }
    } else {
      if (x < 0) { f = -D * x; cost += 0.5 * D * x * x; state = 1; }
      else { f = 0; state = 0; }
    }
    if (set) { d->efc_force[r] = f; d->efc_state[r] = state; }
  }
  return cost;
}

/* exact minimiser of the 1-D piecewise-quadratic cost along `search` (PrimalSearch).  MuJoCo brackets
 * with Newton steps to a gradient tolerance; here the same convex 1-D problem is solved to machine
 * precision by safeguarded Newton on the (monotone, piecewise-linear) derivative. */
typedef struct { double quadGauss[3]; const double *jar, *jv; } LsCtx;
static void lsEval(const OData* d, const LsCtx* c, double alpha, double* deriv) {
  double d1 = 2 * alpha * c->quadGauss[2] + c->quadGauss[1], d2 = 2 * c->quadGauss[2];
  for (int r = 0; r < d->nefc; r++) {
    double x = c->jar[r] + alpha * c->jv[r], jv = c->jv[r], D = d->efc_D[r];
    if (d->efc_type[r] == EFC_FRICTION) {
      double fl = d->efc_frictionloss[r], R = d->efc_R[r];
      if (x <= -R * fl) d1 += -fl * jv;
      else if (x >= R * fl) d1 += fl * jv;
      else { d1 += D * x * jv; d2 += D * jv * jv; }
    } else if (x < 0) { d1 += D * x * jv; d2 += D * jv * jv; }
  }
  deriv[0] = d1; deriv[1] = d2;
}
static double lineSearch(const OData* d, const LsCtx* c) {
  double dv[2], lo = 0, hi = -1, alpha = 0;
  lsEval(d, c, 0, dv);
  if (dv[0] >= 0 || dv[1] <= 0) return 0; /* not a descent direction */
  double d0 = fabs(dv[0]);
  for (int it = 0; it < 40; it++) {
    double a = alpha - dv[0] / dv[1];
    if (hi >= 0 && (a <= lo || a >= hi)) a = 0.5 * (lo + hi);
    lsEval(d, c, a, dv);
    if (dv[0] < 0) lo = a; else hi = a;
    alpha = a;
    if (fabs(dv[0]) <= 1e-14 * d0) break;
    if (hi >= 0 && hi - lo <= 4e-16 * hi) break;
  }
  return alpha;
}

static void mulMv(double* r, const double* M, const double* v, int n) {
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int k = 0; k < n; k++) s += M[i * n + k] * v[k];
    r[i] = s;
  }
}

static void solveNewton(OData* d) {
  const OModel* m = &d->m;
  int nv = m->nv, ne = d->nefc;
  double meaninertia = m->db[LHW_DH_MEANINERTIA], tol = m->db[LHW_DH_TOLERANCE];
  int maxiter = m->ib[LHW_IH_ITERATIONS];
  double scale = 1 / (meaninertia * (nv > 1 ? nv : 1));
  double *Ma = dalloc(nv), *jar = dalloc(ne), *grad = dalloc(nv), *Mgrad = dalloc(nv), *search = dalloc(nv);
  double *Mv = dalloc(nv), *jv = dalloc(ne), *H = dalloc((size_t)nv * nv), *tmp = dalloc(nv > ne ? nv : ne);
  /* warmstart(): pick the cheaper of qacc_warmstart and qacc_smooth */
  if (!(m->ib[LHW_IH_DISABLEFLAGS] & DSBL_WARMSTART)) {
    memcpy(d->qacc, d->qacc_warmstart, sizeof(double) * nv);
    for (int r = 0; r < ne; r++) {
      double s = -d->efc_aref[r];
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * d->qacc[k];
      jar[r] = s;
    }
    double cw = constraintUpdate(d, jar, 0);
    mulMv(Ma, d->M, d->qacc, nv);
    for (int k = 0; k < nv; k++) cw += 0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
    for (int r = 0; r < ne; r++) {
      double s = -d->efc_aref[r];
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * d->qacc_smooth[k];
      tmp[r] = s;
    }
    double cs = constraintUpdate(d, tmp, 0);
    if (cw > cs) memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  } else memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);

  d->solver_niter = 0;
  double cost = 0, oldcost;
  for (int iter = 0; iter <= maxiter; iter++) {
    /* PrimalUpdateConstraint */
    mulMv(Ma, d->M, d->qacc, nv);
    for (int r = 0; r < ne; r++) {
      double s = -d->efc_aref[r];
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * d->qacc[k];
      jar[r] = s;
    }
    oldcost = cost;
    cost = constraintUpdate(d, jar, 1);
    for (int k = 0; k < nv; k++) cost += 0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
    /* PrimalUpdateGradient */
    for (int k = 0; k < nv; k++) {
      double s = 0;
      for (int r = 0; r < ne; r++) s += d->efc_J[(size_t)r * nv + k] * d->efc_force[r];
      d->qfrc_constraint[k] = s;
      grad[k] = Ma[k] - d->qfrc_smooth[k] - s;
    }
    double gn = 0;
    for (int k = 0; k < nv; k++) gn += grad[k] * grad[k];
    gn = sqrt(gn);
    if (iter > 0) {
      double improvement = scale * (oldcost - cost);
      if (improvement < tol || scale * gn < tol) break;
    } else if (scale * gn < tol) break;
    if (iter == maxiter) break;
    /* H = M + J^T D_active J */
    memcpy(H, d->M, sizeof(double) * nv * nv);
    for (int r = 0; r < ne; r++)
      if (d->efc_state[r] == 1) {
        const double* J = d->efc_J + (size_t)r * nv;
        double D = d->efc_D[r];
        for (int a = 0; a < nv; a++) {
          if (J[a] == 0) continue;
          double t = D * J[a];
          for (int b = 0; b < nv; b++) H[a * nv + b] += t * J[b];
        }
      }
    cholFactor(H, nv);
    cholSolve(Mgrad, H, grad, nv);
    for (int k = 0; k < nv; k++) search[k] = -Mgrad[k];
    /* line search */
    mulMv(Mv, d->M, search, nv);
    for (int r = 0; r < ne; r++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * search[k];
      jv[r] = s;
    }
    LsCtx c;
    c.jar = jar; c.jv = jv;
    c.quadGauss[0] = 0; c.quadGauss[1] = 0; c.quadGauss[2] = 0;
    for (int k = 0; k < nv; k++) {
      c.quadGauss[1] += search[k] * (Ma[k] - d->qfrc_smooth[k]);
      c.quadGauss[2] += 0.5 * search[k] * Mv[k];
    }
    double alpha = lineSearch(d, &c);
    if (alpha == 0) break;
    for (int k = 0; k < nv; k++) d->qacc[k] += alpha * search[k];
    d->solver_niter++;
  }
  free(Ma); free(jar); free(grad); free(Mgrad); free(search); free(Mv); free(jv); free(H); free(tmp);
}

static void fwdConstraint(OData* d) {
  int nv = d->m.nv;
  if (d->nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * nv);
    d->solver_niter = 0;
    return;
  }
  solveNewton(d);
  /* mj_fwdConstraint saves the solution as the next call's warm start [MJ-recall] */
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

/* ------------------------------------------------------------------ mj_forward / mj_step */
void orc_forward(OData* d, int actuation_enabled) {
  fwdPosition(d);
  fwdVelocity(d);
  fwdActuation(d, actuation_enabled);
  fwdAcceleration(d);
  fwdConstraint(d);
}

/* mj_Euler with implicit joint damping + mj_advance */
static void euler(OData* d) {
  const OModel* m = &d->m;
  int nv = m->nv;
  double h = m->db[LHW_DH_TIMESTEP];
  const double* damping = DF(m, LHW_DF_DOF_DAMPING);
  const int32_t *jtype = IF(m, LHW_IF_JNT_TYPE), *jq = IF(m, LHW_IF_JNT_QPOSADR), *jd = IF(m, LHW_IF_JNT_DOFADR);
  double* qacc = dalloc(nv);
  int anydamp = 0;
  for (int i = 0; i < nv; i++) anydamp |= damping[i] > 0;
  if (anydamp && !(m->ib[LHW_IH_DISABLEFLAGS] & DSBL_EULERDAMP)) {
    double* MM = dalloc((size_t)nv * nv);
    double* f = dalloc(nv);
    memcpy(MM, d->M, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) { MM[i * nv + i] += h * damping[i]; f[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    cholFactor(MM, nv);
    cholSolve(qacc, MM, f, nv);
    free(MM); free(f);
  } else memcpy(qacc, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = jq[j], da = jd[j];
    if (jtype[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
      quatIntegrate(d->qpos + qa + 3, d->qvel + da + 3, h);
    } else d->qpos[qa] += h * d->qvel[da];
  }
  d->time += h;
  free(qacc);
}

void orc_step(OData* d) {
  orc_forward(d, 1);
  euler(d);
}

/* ------------------------------------------------------------------ derived queries used by the reference */
/* mj_contactForce (reference robot_interface.py:311,323): pyramid decode in the contact frame */
void orc_contact_force(const OData* d, int id, double* out6) {
  memset(out6, 0, 6 * sizeof(double));
  if (id < 0 || id >= d->ncon) return;
  const OContact* c = d->contact + id;
  if (c->efc_address < 0) return;
  const double* f = d->efc_force + c->efc_address;
  if (c->dim == 1) { out6[0] = f[0]; return; }
  out6[0] = f[0] + f[1] + f[2] + f[3];
  out6[1] = c->friction[0] * (f[0] - f[1]);
  out6[2] = c->friction[1] * (f[2] - f[3]);
}

/* mj_objectVelocity(mjOBJ_XBODY) (reference robot_interface.py:363): [rot; lin] at xpos, optionally local */
void orc_object_velocity(const OData* d, int body, int flg_local, double* out6) {
  const OModel* m = &d->m;
  const int32_t* rootid = IF(m, LHW_IF_BODY_ROOTID);
  const double* cv = d->cvel + 6 * body;
  double dif[3], t[3];
  for (int k = 0; k < 3; k++) dif[k] = d->xpos[3 * body + k] - d->subtree_com[3 * rootid[body] + k];
  cross3(t, dif, cv); /* lin_new = lin - dif x omega */
  double lin[3] = {cv[3] - t[0], cv[4] - t[1], cv[5] - t[2]};
  if (flg_local) {
    mulMatTVec3(out6, d->xmat + 9 * body, cv);
    mulMatTVec3(out6 + 3, d->xmat + 9 * body, lin);
  } else {
    memcpy(out6, cv, 3 * sizeof(double));
    memcpy(out6 + 3, lin, 3 * sizeof(double));
  }
}

/* ------------------------------------------------------------------ field access for ctypes */
double* orc_field(OData* d, const char* name) {
#define F(n) if (!strcmp(name, #n)) return d->n;
  F(qpos) F(qvel) F(ctrl) F(xfrc_applied) F(qacc_warmstart) F(xpos) F(xquat) F(xmat) F(xipos) F(ximat) F(geom_xpos)
  F(geom_xmat) F(site_xpos) F(site_xmat) F(subtree_com) F(cvel) F(M) F(qfrc_bias) F(qfrc_passive) F(qfrc_actuator)
  F(qfrc_smooth) F(qacc_smooth) F(qfrc_constraint) F(qacc) F(actuator_length) F(actuator_velocity) F(actuator_force)
  F(efc_J) F(efc_pos) F(efc_force) F(efc_aref) F(efc_D) F(efc_R) F(efc_frictionloss)
#undef F
  return NULL;
}
/* Independent cross-check of the primal Newton solver (tests/test_oracle_crosscheck.py): projected Gauss-Seidel on the DUAL
 * of the same regularised problem,  min_f 1/2 f^T (A + R) f + f^T b,  A = J M^-1 J^T,  b = J qacc_smooth - aref,
 * f_i >= 0 for limit / contact rows, |f_i| <= frictionloss_i for frictionloss rows (the formulation MuJoCo's PGS option
 * uses; strictly convex, so its optimum is the Newton optimum).  AR is the dense n x n matrix A + diag(R), caller-built.
 * Returns the number of sweeps done; stops when the largest force change of a sweep is below tol. */
int orc_dual_pgs(int n, const double* AR, const double* b, const double* floss, const int* is_friction, double* f, int max_sweeps, double tol) {
  int sweep = 0;
  for (; sweep < max_sweeps; sweep++) {
    double change = 0;
    for (int i = 0; i < n; i++) {
      double res = b[i];
      for (int k = 0; k < n; k++) res += AR[(size_t)i * n + k] * f[k];
      double fi = f[i] - res / AR[(size_t)i * n + i];
      if (is_friction[i]) { if (fi > floss[i]) fi = floss[i]; else if (fi < -floss[i]) fi = -floss[i]; }
      else if (fi < 0) fi = 0;
      if (fabs(fi - f[i]) > change) change = fabs(fi - f[i]);
      f[i] = fi;
    }
    if (change < tol) { sweep++; break; }
  }
  return sweep;
}
int orc_efc_type(const OData* d, int r) { return (r >= 0 && r < d->nefc) ? d->efc_type[r] : -1; }
int orc_ncon(const OData* d) { return d->ncon; }
int orc_nefc(const OData* d) { return d->nefc; }
int orc_niter(const OData* d) { return d->solver_niter; }
double orc_time(const OData* d) { return d->time; }
/* contact record i -> out[0]=dist, [1:4]=pos, [4:13]=frame, [13]=geom1, [14]=geom2, [15]=efc_address, [16]=mu */
void orc_contact(const OData* d, int i, double* out) {
  const OContact* c = d->contact + i;
  out[0] = c->dist;
  memcpy(out + 1, c->pos, 3 * sizeof(double));
  memcpy(out + 4, c->frame, 9 * sizeof(double));
  out[13] = c->geom1; out[14] = c->geom2; out[15] = c->efc_address; out[16] = c->friction[0];
}
