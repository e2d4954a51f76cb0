// The resident rollout: policy step + control step for T control steps of a rollout in ONE launch, wave by wave.
//
// Takes over the body of the reference's worker loop (rl/workers/rollout_worker.py:142-181: `action = policy(state)`,
// `env.step(action)`, store, reset on episode end) for every env of a range.  In the reference each worker advances its own
// env without ever waiting for another one; the launch-per-control-step pipeline (lhw_env_step_range + lhw_ppo_forward_at)
// instead ends every control step on the slowest of the batch's wavefronts before the policy launch -- a launch lasts as long
// as its slowest wave (1.58x the mean wave of a 4096-env launch) and the chip drains meanwhile.  Here a wavefront keeps its
// env(s) for the whole rollout and evaluates the actor itself:
//
//   for t in 0 .. T-1:   obs[t] rows of the wave's envs  ->  policy_step  ->  act[t], logp[t]
//                        control_step<0, TASK, W>         ->  obs[t+1], term_obs[t], rew[t], done[t]   (auto-reset inside)
//
// The policy step is the fused strip launch of lhw_mlp_strip.hip restated for the 1-2 rows a wave owns: two rows x 256 hidden
// units are nowhere near an MFMA tile (1/16 of a 32x32 tile), so the layers are plain v_fma_f32 over all 64 lanes -- lane l
// owns hidden units 4l .. 4l+3 of both rows, the weights come straight from L2 in [in][out] order (one 16-byte load per lane
// and k row = 1 KB per wave, fully coalesced; 0.3 MB per wave and control step, ~2 % of a control step), the activations sit
// in the stepper's LDS stage region, which is dead between control steps.  Every value is produced by the SAME operations in
// the SAME order as mlp_fwd_strip_kernel's (normalisation expression, fmaf chains over ascending k from 0, bias added after
// the chain, the read-out cut into eight 32-k partial sums added in ascending order, lhw_policy_sample), so a rollout collected
// here is bitwise the rollout of the launch-per-step pipeline (tests/test_rollout_resident*.py).
//
// Two envs per wave (W = 32): an env that exceeds the layout's 8 contacts in a control step returns untouched from
// control_step<0, TASK, 32>; the launch-per-step path repeats its step with the one-env-per-wave kernel in a second launch.
// Here the wave does that itself, at once: both envs have left the LDS working set by then (everything persistent is in the HBM
// record between control steps), so the wave re-interprets its LDS allocation as the W = 64 layout and runs
// control_step<0, TASK, 64> for the flagged env with all 64 lanes -- same code, same bits as the second launch.
#define LHW_SUBSTEP_PRIO 1   // the sub-steps of these kernels alternate the wave's issue priority with its SIMD partner's (substep(), lhw_humanoid_dev.h)
#include "lhw_humanoid_dev.h"
#include "lhw_policy.h"

#define PH 256      // hidden width of the actor (rl/policies/actor.py:127)
#define PKQ 32      // the read-out is summed as PH / PKQ partial products of PKQ k each (SKQ of lhw_mlp_strip.hip)
#define PXK 64      // capacity of the padded observation row
#define PO_MAX 16   // action dims the read-out's lane mapping covers (JVRC 12, H1 10): four column groups of four

// agent-scope loads / stores of the job queue's progress words (a wave on another XCD wrote them: not through this XCD's L2)
#if defined(__HIP_EMU__)
__device__ __forceinline__ unsigned lhw_load_agent(const unsigned* p) { return *p; }
__device__ __forceinline__ void lhw_store_agent(unsigned* p, unsigned v) { *p = v; }
#else
__device__ __forceinline__ unsigned lhw_load_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lhw_store_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
#endif

struct HRollout {
  int T;                 // control steps of this launch
  int n_total;           // envs of the batch: row count of one time slice of the buffers below
  float* obs;            // [T + 1][n_total][obs_dim]: slice 0 is read (the observation to act on first), 1 .. T are written
  float* act;            // [T][n_total][act_dim]
  float* logp;           // [T][n_total]
  float* tob;            // [T][n_total][obs_dim] observation returned by env.step itself (bootstrap input where an episode ended)
  float* rew;            // [T][n_total]
  unsigned char* done;   // [T][n_total] LHW_DONE_* flags
  float* rew_terms;      // [n_total][n_terms] of the last control step, nullable
  // Job queue (nullable = off: wave b of the grid keeps env group b for all T steps).  When the range has more env groups than the
  // chip has wave slots, a group's rollout is cut into jobs of `chunk` control steps; the resident waves pop jobs from queue[0]
  // in the order (chunk 0 of every group, chunk 1 of every group, ...), queue[1 + group] counts the group's finished chunks.
  unsigned* queue;
  int chunk;
  long long tin_step;    // doubles per control step of the task-input export (n_total * LHW_TASK_INPUT_DIM), 0: the per-launch record (or none)
  LhwRolloutPolicy pol;
};

// One 256-wide ReLU layer for the wave's G rows: hout[r][n] = relu(chain_k fmaf(W^T[k][n], xin[r][k]) + bias[n]), n = 4 wl .. 4 wl + 3.
// The weight rows of PU consecutive k are requested as one batch of independent 16-byte loads, a batch ahead of the one being
// multiplied (left to itself the compiler waits for every load right behind its issue -- s_waitcnt vmcnt(0) 293 times per policy
// step, each a full L2 round trip; the scheduling barriers keep the batches apart, as in slab_mma of lhw_mlp_strip.hip).  Rows
// k >= K are clamped copies of row K - 1 multiplied by the zeros xin holds there (K is padded to a multiple of 2 PU by the caller's
// buffer: the observation row is zero beyond its width up to PXK), so the prefetches run past the end unconditionally.
#define PU 8
// HALF: fp16 operands (BASELINE config 5) -- every weight and every activation is rounded to fp16 before it is multiplied (the product of
// two fp16 values is exact in float32), the sums stay float32 over ascending k; biases, the ReLU and the read-out's final sum are float32.
#if defined(__HIP_EMU__)
__device__ __forceinline__ float r16(float x) { return emu_f16_round(x); }
#else
__device__ __forceinline__ float r16(float x) { return (float)(_Float16)x; }
#endif
template <int G, bool HALF>
__device__ __forceinline__ void policy_hidden(const float* __restrict__ wt, const float* __restrict__ bias, const float* xin, const int ldx, const int K,
                                              float* hout, const int wl) {
  float acc[G][4];
#pragma unroll
  for (int r = 0; r < G; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[r][c] = 0.f;
  const float4* w4 = reinterpret_cast<const float4*>(wt) + wl;
  float4 wa[PU], wb[PU];
  auto load = [&](float4 (&w)[PU], const int k0) {
#pragma unroll
    for (int j = 0; j < PU; j++) w[j] = w4[(size_t)min(k0 + j, K - 1) * (PH / 4)];
  };
  auto mul = [&](const float4 (&w)[PU], const int k0) {
#pragma unroll
    for (int j = 0; j < PU; j++)
#pragma unroll
      for (int r = 0; r < G; r++) {
        const float x = xin[r * ldx + k0 + j];   // (HALF: rounded when it was stored)
        const float w0 = HALF ? r16(w[j].x) : w[j].x, w1 = HALF ? r16(w[j].y) : w[j].y, w2 = HALF ? r16(w[j].z) : w[j].z, w3 = HALF ? r16(w[j].w) : w[j].w;
        acc[r][0] = fmaf(w0, x, acc[r][0]); acc[r][1] = fmaf(w1, x, acc[r][1]);
        acc[r][2] = fmaf(w2, x, acc[r][2]); acc[r][3] = fmaf(w3, x, acc[r][3]);
      }
  };
  load(wa, 0);
  for (int k0 = 0; k0 < K; k0 += 2 * PU) {
    load(wb, k0 + PU);
    __builtin_amdgcn_sched_barrier(0);
    mul(wa, k0);
    __builtin_amdgcn_sched_barrier(0);
    load(wa, k0 + 2 * PU);
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + PU < K) mul(wb, k0 + PU);
    __builtin_amdgcn_sched_barrier(0);
  }
  const float4 b = reinterpret_cast<const float4*>(bias)[wl];
#pragma unroll
  for (int r = 0; r < G; r++) {
    float4 v = make_float4(fmaxf(acc[r][0] + b.x, 0.f), fmaxf(acc[r][1] + b.y, 0.f), fmaxf(acc[r][2] + b.z, 0.f), fmaxf(acc[r][3] + b.w, 0.f));
    if (HALF) v = make_float4(r16(v.x), r16(v.y), r16(v.z), r16(v.w));   // the next layer's operand
    *reinterpret_cast<float4*>(hout + r * PH + 4 * wl) = v;
  }
}

// floats of LDS the policy step of a wave with G rows needs
template <int G> struct PolicyLds { static constexpr int XS = 0, H1 = XS + G * PXK, H2 = H1 + G * PH, PP = H2 + G * PH, TM = PP + 8 * G * 16, FLOATS = TM + G * 16; };

// Actor forward + Gaussian head for the wave's rows (envs env0 .. env0 + nlive - 1 of this time slice); all 64 lanes take part.
template <int G, bool HALF>
__device__ __forceinline__ void policy_step(const LhwRolloutPolicy& q, float* sc, const float* __restrict__ obs_t, float* __restrict__ act_t,
                                            float* __restrict__ logp_t, const int env0, const int nlive, const unsigned genv0, const unsigned counter) {
  typedef PolicyLds<G> PL;
  const int wl = fresh_wave_lane();
  const int D = q.obs_dim, O = q.act_dim, Op = q.act_pad;
  float *xs = sc + PL::XS, *h1 = sc + PL::H1, *h2 = sc + PL::H2, *Pp = sc + PL::PP, *Tm = sc + PL::TM;
  // the normalised observation rows (the expression of stage_input / normalize_kernel), zero beyond the observation width
  for (int i = wl; i < G * PXK; i += 64) {
    const int r = i / PXK, k = i - r * PXK;
    float v = 0.f;
    if (k < D && r < nlive) v = (obs_t[(size_t)(env0 + r) * D + k] - q.obs_mean[k]) / q.obs_std[k];
    xs[i] = HALF ? r16(v) : v;
  }
  SYNC();
  policy_hidden<G, HALF>(q.w1t, q.b1, xs, PXK, q.obs_pad, h1, wl);
  SYNC();
#ifdef LHW_DBG_POL
  if (wl == 0 && env0 == 0) printf("dbg counter %u xs %g %g %g %g h1 %g %g %g %g w1t %g %g b1 %g\n", counter, xs[0], xs[1], xs[36], xs[37], h1[0], h1[1], h1[2], h1[255], q.w1t[0], q.w1t[1], q.b1[0]);
#endif
  policy_hidden<G, HALF>(q.w2t, q.b2, h1, PH, PH, h2, wl);
  SYNC();
  // read-out: lane = (partial q8 of 8, row r, column group cg of four output units); with one row per wave the r = 1 lanes repeat
  // the r = 0 lanes' work and keep it to themselves.  One 16-byte weight load per lane and k (W3^T rows are act_pad floats), PU of
  // them in flight
  const int q8 = wl >> 3, r = (wl >> 2) & 1, rr_ = G == 2 ? r : 0, cg = wl & 3;
  const bool colsin = 4 * cg < Op;      // (act_pad = 12: the fourth column group has no columns)
  float pacc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* w3 = q.w3t + (size_t)(q8 * PKQ) * Op + 4 * (colsin ? cg : 0);
  for (int kk0 = 0; kk0 < PKQ; kk0 += PU) {
    float4 w[PU];
#pragma unroll
    for (int j = 0; j < PU; j++) w[j] = *reinterpret_cast<const float4*>(w3 + (size_t)(kk0 + j) * Op);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PU; j++) {
      const float h = h2[rr_ * PH + q8 * PKQ + kk0 + j];
      const float w0 = HALF ? r16(w[j].x) : w[j].x, w1 = HALF ? r16(w[j].y) : w[j].y, w2 = HALF ? r16(w[j].z) : w[j].z, w3 = HALF ? r16(w[j].w) : w[j].w;
      pacc[0] = fmaf(h, 4 * cg + 0 < O ? w0 : 0.f, pacc[0]); pacc[1] = fmaf(h, 4 * cg + 1 < O ? w1 : 0.f, pacc[1]);
      pacc[2] = fmaf(h, 4 * cg + 2 < O ? w2 : 0.f, pacc[2]); pacc[3] = fmaf(h, 4 * cg + 3 < O ? w3 : 0.f, pacc[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (colsin && (G == 2 || r == 0)) {
#pragma unroll
    for (int c = 0; c < 4; c++) Pp[(q8 * G + rr_) * 16 + 4 * cg + c] = pacc[c];
  }
  SYNC();
  if (wl < 16 * G) {
    const int rr = wl >> 4, col = wl & 15;
    if (col < O && rr < nlive) {
      float s = Pp[rr * 16 + col];
#pragma unroll
      for (int j = 1; j < PH / PKQ; j++) s += Pp[(j * G + rr) * 16 + col];
      s += q.b3[col];
      float term;
      const float a = lhw_policy_sample(s, q.stdv[col], q.seed, genv0 + (unsigned)rr, counter, col, q.deterministic, &term);
      act_t[(size_t)(env0 + rr) * O + col] = a;
      Tm[rr * 16 + col] = term;
    }
  }
  SYNC();
  if (wl < nlive) {
    float lp = 0.f;
    for (int k = 0; k < O; k++) lp += Tm[wl * 16 + k];     // (the order of sample_kernel's sum)
    logp_t[env0 + wl] = lp;
  }
  SYNC();
}

// Control steps [t0, t1) of env group `grp` of the range (the G envs one wave advances together).
template <int TASK, int W>
__device__ __forceinline__ void rollout_steps(HModelRef m, HParamsRef p, const HLaunch& lz, const HState& st, const HRollout& ro, unsigned char* SGraw, int grp,
                                              int t0, int t1) {
  using L = typename LayoutOf<TASK, W>::type;
  using L1 = typename LayoutOf<TASK, 64>::type;
  constexpr int G = 64 / W;   // envs per wavefront
  L* SG = reinterpret_cast<L*>(SGraw);
  const int eidx0 = grp * G;
  const int nlive = min(G, lz.env_count - eidx0);
  const int env0 = lz.env_first + eidx0;
  const int OBS = TASK == TASK_WALK ? 37 : (TASK == TASK_STEP ? 39 : (TASK == TASK_H1WALK ? 43 : 35));
  const size_t N = (size_t)ro.n_total;
  // (diagnostic, lhw_env_debug_wave_cycles: shader-clock cycles the group's control steps took, summed over the rollout's chunks;
  //  control_step leaves the last step's own figure in the same word, so the running sum is picked up before the chunk's first step)
  long long cyc_before = 0;
  if (st.wave_cyc && t0 > 0) {
    const int wl = fresh_wave_lane();
    if ((wl & (W - 1)) == 0 && (W == 32 ? (wl >> 5) : 0) < nlive) cyc_before = st.wave_cyc[env0 + (W == 32 ? (wl >> 5) : 0)];
  }
  const long long t_begin = st.wave_cyc ? (long long)clock64() : 0;
  for (int t = t0; t < t1; t++) {
    GROUP_SYNC(64);
    // what this wave wrote in the previous control step (the observation rows it now reads; after a W = 64 re-run, by other
    // lanes than the ones that read them) is visible
    __threadfence();
    const float* obs_t = ro.obs + (size_t)t * N * OBS;
    float* act_t = ro.act + (size_t)t * N * m.nu;
    if (ro.pol.fp16_operands)
      policy_step<G, true>(ro.pol, reinterpret_cast<float*>(SG[0].U), obs_t, act_t, ro.logp + (size_t)t * N, env0, nlive, p.env_id_base + (unsigned)env0,
                           ro.pol.counter + (unsigned)t);
    else
      policy_step<G, false>(ro.pol, reinterpret_cast<float*>(SG[0].U), obs_t, act_t, ro.logp + (size_t)t * N, env0, nlive, p.env_id_base + (unsigned)env0,
                            ro.pol.counter + (unsigned)t);
#ifdef LHW_RO_POLICY2X   // (analysis builds: the policy step twice -- the difference in rollout time is its cost)
    policy_step<G, false>(ro.pol, reinterpret_cast<float*>(SG[0].U), obs_t, act_t, ro.logp + (size_t)t * N, env0, nlive, p.env_id_base + (unsigned)env0,
                          ro.pol.counter + (unsigned)t);
#endif
    __threadfence();   // the action rows are read back by the lanes of their env's group
    const int wl = fresh_wave_lane();
    const int g = W == 32 ? (wl >> 5) : 0, lane = wl & (W - 1);
    float* obs_n = ro.obs + (size_t)(t + 1) * N * OBS;
    float* tob_t = ro.tob + (size_t)t * N * OBS;
    float* rew_t = ro.rew + (size_t)t * N;
    unsigned char* done_t = ro.done + (size_t)t * N;
    bool ovf = false;
    HLaunch lzt = lz;
    lzt.tin_off = (long long)t * ro.tin_step;   // (the batched sim facade of EVERY control step, for reward-only task plug-ins: lhw_env_rollout_task_inputs)
    if (g < nlive)
      ovf = control_step<0, TASK, W>(m, p, lzt, st, SG, SG[g], env0 + g, lane, act_t, obs_n, tob_t, rew_t, done_t, ro.rew_terms, nullptr, nullptr);
#ifndef LHW_RO_NO_RERUN   // (analysis builds: without the in-wave re-run, to see what its code costs the hot path -- nothing measurable)
    if constexpr (W == 32) {
      GROUP_SYNC(64);
      const unsigned long long ob = __ballot(ovf);
      if (ob) {
        HLaunch lz1 = lzt;
        lz1.only_flagged = 1;   // (store_record clears the env's flag and counts the re-run)
        L1* S1 = reinterpret_cast<L1*>(SGraw);
        for (int gg = 0; gg < 2; gg++)
          if ((ob >> (32 * gg)) & 1ull) {
            SYNC();
            // (as a call -- it is rare -- the kernel spills 12 fewer VGPRs and 200 fewer SGPRs and is 9 % SLOWER: profiles/r05_calls_and_scratch.txt)
            control_step<0, TASK, 64>(m, p, lz1, st, S1, S1[0], env0 + gg, fresh_wave_lane(), act_t, obs_n, tob_t, rew_t, done_t, ro.rew_terms, nullptr, nullptr);
          }
      }
    }
#endif
  }
  if (st.wave_cyc) {
    const int wl = fresh_wave_lane();
    const int g = W == 32 ? (wl >> 5) : 0;
    if ((wl & (W - 1)) == 0 && g < nlive) st.wave_cyc[env0 + g] = cyc_before + ((long long)clock64() - t_begin);
  }
}

// QUEUE = false: wave b of the grid keeps env group b for all T control steps.
// QUEUE = true: the grid is the chip's resident set of waves and drains a job list (HRollout::queue) -- a group whose envs are slow
// (stepping task: the walking mode and the terrain under the feet set the contact count for a whole episode; the waves of one
// launch spread +-20 % around their mean, and with two groups per wave slot the launch ends 22 % after the mean slot) no longer
// decides when its slot's NEXT group can start: the slot takes whatever job is next.  Nothing of an env lives in the wave between
// control steps (control_step reads the HBM record and writes it back), so which wave runs a job does not matter to the bits.
// A separate instantiation: wrapped into the job loop, the two-envs-per-wave kernels spill 25 more VGPRs and lose 2.8 % (round 5,
// same box, jvrc_walk @ 4096), and at 8192 envs their waves are within +-2 % of each other anyway (queue +0.1 %).
template <int TASK, int W, bool QUEUE>
__global__ void __launch_bounds__(64, LHW_WAVES_PER_SIMD) humanoid_rollout_kernel(const HModel* __restrict__ mp, const HParams* __restrict__ pp, HLaunch lz, HState st, HRollout ro) {
  using L = typename LayoutOf<TASK, W>::type;
  using L1 = typename LayoutOf<TASK, 64>::type;
  constexpr int G = 64 / W;
  constexpr size_t LDS_BYTES = sizeof(L) * G > sizeof(L1) ? sizeof(L) * G : sizeof(L1);
  static_assert(W == 64 || sizeof(L1) <= sizeof(L) * G, "the one-env-per-wave layout must fit the wave's two-env allocation (8 workgroups per CU)");
  static_assert(L::USIZE_ * 2 - 48 >= PolicyLds<G>::FLOATS, "the policy step's activations must fit the stage region in front of the observation staging");
  __shared__ __attribute__((aligned(16))) unsigned char SGraw[LDS_BYTES];
  LHW_LDS_POISON(SGraw);
  HParamsRef p = *(const HParams LHW_GLOBAL_AS*)pp;
  HModelRef m = *(const HModel LHW_GLOBAL_AS*)mp;
  const int n_groups = (lz.env_count + G - 1) / G;
  if constexpr (!QUEUE) {
    if ((int)blockIdx.x >= n_groups) return;
    rollout_steps<TASK, W>(m, p, lz, st, ro, SGraw, (int)blockIdx.x, 0, ro.T);
  } else {
    const int n_chunks = (ro.T + ro.chunk - 1) / ro.chunk;
    for (;;) {
      unsigned j = 0;
      if (fresh_wave_lane() == 0) j = atomicAdd(ro.queue, 1u);
      j = (unsigned)__builtin_amdgcn_readlane((int)j, 0);
      if (j >= (unsigned)n_groups * (unsigned)n_chunks) break;
      const int c = (int)(j / (unsigned)n_groups), grp = (int)(j % (unsigned)n_groups);
      const int t0 = c * ro.chunk, t1 = min(ro.T, t0 + ro.chunk);
      // the group's previous chunk was popped n_groups - 1 jobs ago by a wave that is running: it ends without waiting for anyone
      while (lhw_load_agent(ro.queue + 1 + grp) < (unsigned)c) __builtin_amdgcn_s_sleep(32);
      rollout_steps<TASK, W>(m, p, lz, st, ro, SGraw, grp, t0, t1);
      __threadfence();   // the group's records, observations and flags of this chunk, before the chunk counts as done
      if (fresh_wave_lane() == 0) lhw_store_agent(ro.queue + 1 + grp, (unsigned)(c + 1));
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
// The stepping task's two instantiations live in a translation unit of their own, lhw_humanoid_rollout_step.hip (this file included with
// LHW_ROLLOUT_STEP_TU defined): it is compiled with LLVM's iterative ILP scheduling strategy, which makes the one-env-per-wave kernels 3 % faster
// (the two-envs-per-wave kernels keep the default strategy: profiles/r06_stepper_compiler_flags.txt) -- and the two halves compile in parallel.
void humanoid_rollout_launch_step(bool queued, dim3 grid, hipStream_t s, const HModel* m_dev, const HParams* p_dev, HLaunch lz, HState st, HRollout ro);
#ifdef LHW_ROLLOUT_STEP_TU
void humanoid_rollout_launch_step(bool queued, dim3 grid, hipStream_t s, const HModel* m_dev, const HParams* p_dev, HLaunch lz, HState st, HRollout ro) {
  if (queued) hipLaunchKernelGGL((humanoid_rollout_kernel<TASK_STEP, 64, true>), grid, dim3(64), 0, s, m_dev, p_dev, lz, st, ro);
  else hipLaunchKernelGGL((humanoid_rollout_kernel<TASK_STEP, 64, false>), grid, dim3(64), 0, s, m_dev, p_dev, lz, st, ro);
}
#else
#ifdef LHW_ONLY_WALK
#define ROLLOUT_OTHER_TASKS(WIDTH)
#else
#define ROLLOUT_OTHER_TASKS(WIDTH)                                                                                                                              \
  else if (h->p.task == TASK_H1WALK) hipLaunchKernelGGL((humanoid_rollout_kernel<TASK_H1WALK, WIDTH, false>), grid, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz, st, ro); \
  else hipLaunchKernelGGL((humanoid_rollout_kernel<TASK_STAND, WIDTH, false>), grid, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz, st, ro);
#endif

int humanoid_last_rollout_queued(const HumanoidEnv* h) { return h->last_rollout_queued; }

int humanoid_rollout(HumanoidEnv* h, int first, int count, int T, const LhwRolloutPolicy* pol, float* obs, float* act, float* logp, float* term_obs,
                     float* rew, uint8_t* done, float* rew_terms, double* tin_all, hipStream_t s) {
  if (first < 0 || count <= 0 || first + count > h->p.n_envs || T <= 0) return -1;
  const int obs_dim = h->p.task == TASK_STEP ? 39 : (h->p.task == TASK_WALK ? 37 : (h->p.task == TASK_H1WALK ? 43 : 35));
  if (pol->hidden != PH || pol->obs_dim != obs_dim || pol->act_dim != h->m.nu || pol->act_pad > PO_MAX || (pol->act_pad & 3) || pol->act_pad < pol->act_dim ||
      pol->obs_pad < obs_dim || pol->obs_pad > PXK || (pol->obs_pad & 3))
    return -2;
  HRollout ro;
  ro.T = T; ro.n_total = h->p.n_envs;
  ro.obs = obs; ro.act = act; ro.logp = logp; ro.tob = term_obs; ro.rew = rew; ro.done = done; ro.rew_terms = rew_terms;
  ro.pol = *pol;
  ro.queue = nullptr; ro.chunk = 0;
  ro.tin_step = tin_all ? (long long)h->p.n_envs * LHW_TASK_INPUT_DIM : 0;
  HState st = h->st;
  if (tin_all) st.tin = tin_all;   // [T][n_envs][LHW_TASK_INPUT_DIM]: every control step's record instead of the last one's
  const HLaunch lz{first, count, 0, h->iteration, 0};
  if (!h->fast && h->p.task != TASK_STEP) return -3;   // a walking / standing model that does not fit the two-envs-per-wave layout (or LHW_ONE_ENV_PER_WAVE): launch-per-step only
  // Stepping task with more env groups than wave slots: the resident waves share a job queue of `chunk`-step pieces instead of a
  // group each (humanoid_rollout_kernel<.., QUEUE = true>).  LHW_ROLLOUT_CHUNK: control steps per job (default 10; 0 = one wave per
  // group whatever the batch).  LHW_ROLLOUT_SLOTS: tests only,
  // the number of wave slots to assume (so that a small batch takes the queue path).
  const int n_groups = h->fast ? (count + 1) / 2 : count;
  int grid_n = n_groups;
  {
    const int chunk_env = getenv("LHW_ROLLOUT_CHUNK") ? atoi(getenv("LHW_ROLLOUT_CHUNK")) : 10;
    int slots = getenv("LHW_ROLLOUT_SLOTS") ? atoi(getenv("LHW_ROLLOUT_SLOTS")) : 0;
    if (slots <= 0) {
      static int chip_slots = 0;
      if (!chip_slots) {
        int nb = humanoid_occupancy(), dev = 0;
        hipDeviceProp_t prop;
        if (nb <= 0 || hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -4;
        chip_slots = nb * prop.multiProcessorCount;
      }
      slots = chip_slots;
    }
    if (!h->fast && chunk_env > 0 && n_groups > slots && T > chunk_env) {
      if (!h->ro_queue) {   // two words per env: the ranges of concurrent launches (disjoint by contract) get disjoint pieces
        void* d = nullptr;
        if (lhw_malloc(&d, ((size_t)h->p.n_envs * 2 + 2) * sizeof(unsigned)) != hipSuccess) return -4;
        h->dev_allocs.push_back(d);
        h->ro_queue = (unsigned*)d;
      }
      ro.queue = h->ro_queue + 2 * (size_t)first;
      ro.chunk = chunk_env;
      if (hipMemsetAsync(ro.queue, 0, ((size_t)n_groups + 1) * sizeof(unsigned), s) != hipSuccess) return -4;
      grid_n = slots;
    }
  }
  const dim3 grid(grid_n);
  h->last_rollout_queued = ro.queue != nullptr;
  if (h->fast) {
    if (h->p.task == TASK_WALK) hipLaunchKernelGGL((humanoid_rollout_kernel<TASK_WALK, 32, false>), grid, dim3(64), 0, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz, st, ro);
    ROLLOUT_OTHER_TASKS(32)
  } else {
#ifndef LHW_ONLY_WALK
    humanoid_rollout_launch_step(ro.queue != nullptr, grid, s, (const HModel*)h->m_dev, (const HParams*)h->p_dev, lz, st, ro);
#endif
  }
  return 0;
}
#endif   // LHW_ROLLOUT_STEP_TU
