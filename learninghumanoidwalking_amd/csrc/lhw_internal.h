// Internal (non-ABI) interfaces between the translation units of liblhw.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/lhw.h"

int lhw_fail(int code, const char* fmt, ...);

// POISON MODE (debug): nothing on the GPU zero-fills LDS, and what a workgroup finds there is whatever the previous occupant
// of the CU left -- another kernel, another process.  LHW_LDS_POISON(obj), placed right behind a __shared__ declaration,
// fills the object with 0xFF bytes (NaN as float / double, -1 as int) in the SIMT emulator of tests/emu (always) and in a
// -DLHW_POISON build of the GPU library (scripts/build_variant.sh), so that a read of a never-written word shows up as a NaN
// instead of depending on the CU's history.  A no-op in the product build.
#if defined(__HIP_EMU__)
#define LHW_LDS_POISON(obj) emu::poison_shared((void*)&(obj), sizeof(obj))
#elif defined(LHW_POISON)
#define LHW_LDS_POISON(obj)                                                                                            \
  do {                                                                                                                 \
    unsigned* w_ = reinterpret_cast<unsigned*>(&(obj));                                                                \
    for (unsigned i_ = threadIdx.x; i_ < sizeof(obj) / 4; i_ += blockDim.x) w_[i_] = 0xFFFFFFFFu;                      \
    __syncthreads();                                                                                                   \
  } while (0)
#else
#define LHW_LDS_POISON(obj) ((void)0)
#endif
// device allocations of the library: 0xFF-filled when LHW_POISON=1 is set in the environment (hipMalloc does not zero either)
hipError_t lhw_malloc(void** p, size_t n);
template <class T> static inline hipError_t lhw_malloc(T** p, size_t n) { return lhw_malloc((void**)p, n); }

struct HumanoidEnv;
int humanoid_create(HumanoidEnv** out, const std::vector<int32_t>& mi, const std::vector<double>& md, const LhwEnvConfig* cfg,
                    int* obs_dim, int* act_dim, int* n_terms);
void humanoid_destroy(HumanoidEnv* h);
void humanoid_reset(HumanoidEnv* h, const uint8_t* mask, float* obs, hipStream_t s);
void humanoid_step(HumanoidEnv* h, const float* act, float* obs, float* term_obs, float* rew, uint8_t* done, float* rew_terms,
                   hipStream_t s);
int humanoid_step_range(HumanoidEnv* h, int first, int count, const float* act, float* obs, float* term_obs, float* rew, uint8_t* done,
                        float* rew_terms, hipStream_t s);
int humanoid_last_rollout_queued(const HumanoidEnv* h);
int humanoid_rollout(HumanoidEnv* h, int first, int count, int T, const LhwRolloutPolicy* pol, float* obs, float* act, float* logp, float* term_obs,
                     float* rew, uint8_t* done, float* rew_terms, double* tin_all, hipStream_t s);   // lhw_humanoid_rollout.hip; -1 bad range, -2 / -3 unsupported, -4 HIP error
void humanoid_get_state(HumanoidEnv* h, double* qpos, double* qvel, hipStream_t s);
void humanoid_set_state(HumanoidEnv* h, const double* qpos, const double* qvel, hipStream_t s);
double* humanoid_ep_stats(HumanoidEnv* h);
void humanoid_set_iteration(HumanoidEnv* h, int64_t it);
int humanoid_occupancy();
int humanoid_wave_cycles(HumanoidEnv* h, long long* out);
int humanoid_profile(HumanoidEnv* h, int enable, long long* out16);
int humanoid_task_inputs(HumanoidEnv* h, int enable /* -1: leave */, double* out_host, double** out_dev);
int humanoid_actuator_state(HumanoidEnv* h, double* pos, double* vel, double* tq);
int humanoid_step_record(HumanoidEnv* h, double* seq, double* floor_z, int32_t* istate);

// LDS-resident strip kernels of the 3-layer MLPs (lhw_mlp_strip.hip); hidden width 256 only, callers fall back to the per-layer
// GEMMs otherwise
struct MlpStripFwd {
  const float *w1t, *b1, *w2t, *b2, *w3t, *b3;   // TRANSPOSED weights ([in][out]: W1^T [Dp][256], W2^T [256][256], W3^T [256][Op]) from mlp_strip_prepare
  const float* x; int ldx;                   // [R][ldx] inputs (Dp used columns)
  int Dp, O, Op, R;
  float *h1, *h2, *y;                        // [R][256], [R][256], [R][Op]; h1 / h2 may be NULL (inference: not written to HBM)
  // optional fusions for rollout inference (all NULL / 0 otherwise):
  const float *in_mean = nullptr, *in_std = nullptr; int in_dim = 0;   // x holds RAW observations [R][ldx = in_dim]: the slab is staged
                                                                      // as (x - mean) / std for columns < in_dim, zero up to Dp
  const float* stdv = nullptr;               // Gaussian head on the read-out (lhw_policy.h): act [R][O] = sample(y, stdv), logp [R]
  float *act = nullptr, *logp = nullptr;
  unsigned long long seed = 0; unsigned env_base = 0, counter = 0; int deterministic = 0;
  // optional (the update, 64-row slabs only): the ReLU masks of h1 / h2 as bits, mlp_strip_bits_words(R) words each, for the backward launch
  // over the SAME rows (same first row, same R): see store_act_t
  unsigned *bits1 = nullptr, *bits2 = nullptr;
};
struct MlpStripBwd {
  const float *w2, *w3, *dy, *h1, *h2;       // torch Linear layout [out][in]: W2 [256][256], W3 [Op][256]; dy [R][Op]
  int O, Op, R;
  float *dh2, *dh1;                          // [R][256] each
  const unsigned *bits1 = nullptr, *bits2 = nullptr;   // the forward launch's mask bits: h1 / h2 are then not read
};
size_t mlp_strip_bits_words(size_t rows);    // words per layer of the mask bits of a launch over `rows` rows (64-row slabs)
bool mlp_strip_supported(int H, int Dp, int O, int Op);
size_t mlp_strip_wt_floats(int Dp, int Op);
void mlp_strip_prepare(const float* w1, const float* w2, const float* w3, int Dp, int O, int Op, float* wt, hipStream_t s);
void mlp_strip_forward(const MlpStripFwd& a, hipStream_t s, int shape = 0);   // shape: 0 by row count, 1 small (32-row slabs), 2 big (64-row)
void mlp_strip_backward(const MlpStripBwd& a, hipStream_t s);
