// Gaussian policy head shared by the stand-alone sampling kernel (lhw_ppo.hip) and the fused read-out of the forward strip
// kernel (lhw_mlp_strip.hip: the rollout's one-launch policy step): act = mu + std * N(0,1) with a counter-based normal
// (Box-Muller on two uniforms keyed by (seed, global env id, policy stream, step counter, action index)) and the action's
// log-density term.  One definition, so both paths produce bit-identical actions and log-probabilities (the reference samples
// torch.distributions.Normal, rl/policies/actor.py:160-188).
#pragma once
#include "lhw_rng.h"

// returns the action component; *lp_term receives its contribution to log pi(a | s)
LHW_HD float lhw_policy_sample(float mu, float sd, uint64_t seed, uint32_t genv, uint32_t counter, int a, int deterministic, float* lp_term) {
  float x = mu;
  if (!deterministic) {
    const double u1 = lhw_rng_u01(seed, genv, LHW_STREAM_POLICY, counter, 2 * a);
    const double u2 = lhw_rng_u01(seed, genv, LHW_STREAM_POLICY, counter, 2 * a + 1);
    const float z = (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2));
    x = fmaf(sd, z, mu);
  }
  const float d = (x - mu) / sd;
  *lp_term = fmaf(-0.5f * d, d, -logf(sd)) - 0.9189385332046727f;
  return x;
}
