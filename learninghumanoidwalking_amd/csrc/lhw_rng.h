// Counter-based RNG shared by all kernels (and restated in oracle/rng.py for the checker).
//
// The reference draws from the process-global np.random stream in a data-dependent order
// ("Global-np.random draw order"); a batched device implementation
// cannot reproduce a sequential Mersenne-Twister stream, so every draw is instead a pure
// function of (seed, global env index, stream, event counter, slot).  Parity tests drive the
// oracle with the same function, which makes control flow (resets, mode switches) bit-identical.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LHW_HD __host__ __device__ __forceinline__
#else
#define LHW_HD static inline
#endif

enum { LHW_STREAM_RESET = 1, LHW_STREAM_STEP = 2, LHW_STREAM_POLICY = 3 };

LHW_HD uint64_t lhw_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

LHW_HD uint64_t lhw_rng_bits(uint64_t seed, uint32_t env, uint32_t stream, uint32_t counter, uint32_t slot) {
  uint64_t k = lhw_splitmix64(seed ^ (0xD1342543DE82EF95ull * (uint64_t)(env + 1)));
  k = lhw_splitmix64(k ^ ((uint64_t)stream << 56) ^ ((uint64_t)counter << 16) ^ (uint64_t)slot);
  return k;
}

// uniform in [0,1) with 53 random bits
LHW_HD double lhw_rng_u01(uint64_t seed, uint32_t env, uint32_t stream, uint32_t counter, uint32_t slot) {
  return (double)(lhw_rng_bits(seed, env, stream, counter, slot) >> 11) * (1.0 / 9007199254740992.0);
}

// np.random.uniform(lo, hi): lo + (hi-lo)*u
LHW_HD double lhw_rng_uniform(uint64_t seed, uint32_t env, uint32_t stream, uint32_t counter, uint32_t slot,
                              double lo, double hi) {
  return lo + (hi - lo) * lhw_rng_u01(seed, env, stream, counter, slot);
}

// np.random.randint(n): floor(u*n)
LHW_HD int lhw_rng_randint(uint64_t seed, uint32_t env, uint32_t stream, uint32_t counter, uint32_t slot, int n) {
  int r = (int)(lhw_rng_u01(seed, env, stream, counter, slot) * (double)n);
  return r >= n ? n - 1 : r;
}
