// LDS-resident strip kernels for the 3-layer MLPs of the PPO update (actor / critic: in -> 256 -> 256 -> out).
//
// Takes over, for one network, the layer-by-layer passes of /root/reference/rl/algos/ppo.py:299-406 (`update_actor_critic`:
// actor / critic forward, loss.backward()) that lhw_ppo.hip otherwise runs as one GEMM launch per layer with the activations
// round-tripping HBM between the launches.  A workgroup owns a SLAB of 64 rows and keeps it in LDS through all layers:
//
//   forward   x slab -> h1 = relu(x W1^T + b1) -> h2 = relu(h1 W2^T + b2) -> y = h2 W3^T + b3
//             (h1 / h2 are written to HBM once, for the backward pass, and never read back by the forward pass)
//   backward  dy slab -> dh2 = (dy W3) * (h2 > 0) -> dh1 = (dh2 W2) * (h1 > 0)
//             (dh2 / dh1 are written once, for the weight-gradient GEMMs, which contract over the minibatch rows and stay
//             split-K GEMMs in lhw_ppo.hip)
//
// The slab sits k-major in LDS (S[k][row]): that is the A-operand layout of v_mfma_f32_32x32x2_f32 (lane l: A[row = l % 32]
// [k = l / 32]), so a layer's output tile, written back column by column, IS the next layer's A operand.  The weights are the
// B operand and come straight from global memory (L2-resident, 0.3 MB per network) in [in][out] order -- for the forward pass
// from transposed copies made once per optimiser step (mlp_strip_prepare) -- so a wave's load of one k row of its 64 output
// columns is two 128-byte segments and the K loop needs no LDS staging and no barrier: 256 threads = 4 waves, wave w owns all
// 64 rows x the 64 columns 64 w .. 64 w + 63 (2 x 2 MFMA tiles: one A read and one B load feed two MFMAs each), the operands
// of K step s + 1 are in flight while step s is multiplied, and the only barriers are the two per layer around the slab
// hand-off.  70 KB of LDS per block: two blocks per CU, so one block's epilogue (the single HBM write of h / dh) overlaps the
// other's products.  The read-out layer (N <= 32) is split over K between the four waves and summed in a fixed order.
// Float32 operands and accumulation (the f32-input MFMA is an fmaf chain over ascending k), so the hidden layers are
// bit-identical to the per-layer GEMM path and the networks keep the reference's float32 semantics; only the read-out's
// summation order differs.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/lhw.h"
#include "lhw_internal.h"
#include "lhw_policy.h"

#ifndef __HIP_EMU__
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

#define SH 256          // hidden width the strip kernels are compiled for
#define SBK 16          // K step (one register buffer of weight operands)
#define SXK 64          // capacity of the input slab (padded input width of the first layer / output width of the last)
#define SKQ 32          // the read-out is summed as SH / SKQ partial products of SKQ k each, in ascending order, whatever the shape

// Shape of a workgroup: RT row tiles of 32 slab rows, NW waves, each owning CT column tiles of 32 output units (NW * CT * 32 =
// SH).  Big: 64-row slabs, 4 waves x 64 columns -- 2 x 2 MFMA tiles per wave, one slab / weight operand feeds two MFMAs each;
// 70 KB of LDS (two workgroups per CU): the update's shape.  Small: 32-row slabs, 8 waves x 32 columns -- a quarter of the
// MFMA chain per wave: the shape of rollout inference, where a few thousand rows put fewer slabs on the chip than it has CUs
// and the latency of one slab is the latency of the policy step (41 -> ~15 us).  Results are identical between the shapes (the
// hidden layers are the same fmaf chains, the read-out has the same partial sums).
template <int RT_, int CT_, int NW_>
struct StripShape {
  static constexpr int RT = RT_, CT = CT_, NW = NW_, ROWS = 32 * RT_, LD = ROWS + 4, THR = 64 * NW_;
  static_assert(NW_ * CT_ * 32 == SH, "the waves' column tiles must cover the hidden width");
};
typedef StripShape<2, 2, 4> StripBig;
typedef StripShape<1, 1, 8> StripSmall;

// -DLHW_STRIP_CLOCK (analysis builds, scripts/strip_clock.py): every wave of the first 2048 workgroups stamps the 100 MHz wall clock at
// the phase boundaries of the strip kernels; lhw_debug_strip_clock_read copies the stamps out.
#ifdef LHW_STRIP_CLOCK
#define SCLK_N 16
__device__ unsigned long long g_strip_clk[2048 * 4 * SCLK_N];
#define SCLK(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048 && (threadIdx.x >> 6) < 4) g_strip_clk[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SCLK_N + (k)] = wall_clock64(); } while (0)
extern "C" int lhw_debug_strip_clock_read(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_strip_clk), sizeof(g_strip_clk)) == hipSuccess ? LHW_OK : LHW_ERR_HIP;
}
#else
#define SCLK(k) do { } while (0)
#endif

template <class C>
struct StripLds {
  float S[SH][C::LD];       // activation slab, k-major (Big: 69 632 B).  The input slab (x / dy, at most SXK columns) occupies
};                          // its last SXK rows until the first layer's products are done

template <class C>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[C::RT][C::CT]) {
#pragma unroll
  for (int i = 0; i < C::RT; i++)
#pragma unroll
    for (int j = 0; j < C::CT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
}

// One K step (SBK rows of k) of weights, as the MFMA wants them: lane l holds, for kk = 0 .. SBK/2, the element of row
// k0 + 2 kk + l / 32 in the wave's CT column tiles.
template <class C>
struct WOp { float v[SBK / 2][C::CT]; };

// weights of step k0 straight from global memory: stored [K][ldb] with the output unit contiguous, so one load instruction of
// the wave fetches two 128-byte segments; no LDS staging.  Rows k >= K are clamped copies of row K - 1 (the slab holds zeros
// there), so the prefetches can run past the end unconditionally.
template <class C>
__device__ __forceinline__ void wload(WOp<C>& w, const float* __restrict__ Bg, const int ldb, const int K, const int k0) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, n0 = (threadIdx.x >> 6) * 32 * C::CT + l31;
#pragma unroll
  for (int kk = 0; kk < SBK / 2; kk++) {
    const int k = min(k0 + kk * 2 + kh, K - 1);
#pragma unroll
    for (int j = 0; j < C::CT; j++) w.v[kk][j] = Bg[(size_t)k * ldb + n0 + 32 * j];
  }
}

// acc[i][j] += (A[rows 32 i .. +32][0 .. K) * B[0 .. K)[the wave's column tile j])^T.  A is the LDS slab (k-major; rows k >= K up
// to the next multiple of SBK must hold ZEROS), B the weights (wload).  No barrier in the K loop: the waves of the block own
// disjoint output columns and share only the read-only slab.  The weights of step s + 1 are in flight while step s is
// multiplied; `w0` arrives holding the weights of step 0 (the caller issues that load early, e.g. before the previous layer's
// epilogue).
// The weights are the MFMA's A operand and the slab its B operand, so the accumulators hold the TRANSPOSED output tile: lane =
// slab row, four consecutive output units per register quad (16-byte row-major stores in the epilogue).
template <class C>
__device__ __forceinline__ void slab_mma(const float (*A)[C::LD], const int K, const float* __restrict__ Bg, const int ldb, WOp<C>& w0,
                                         f32x16 (&acc)[C::RT][C::CT]) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
  WOp<C> w1;
  auto mul = [&](const WOp<C>& w, int k0) {
#pragma unroll
    for (int kk = 0; kk < SBK / 2; kk++) {
      const int k = k0 + kk * 2 + kh;
      float a[C::RT];
#pragma unroll
      for (int i = 0; i < C::RT; i++) a[i] = A[k][32 * i + l31];
#pragma unroll
      for (int i = 0; i < C::RT; i++)
#pragma unroll
        for (int j = 0; j < C::CT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v[kk][j], a[i], acc[i][j], 0, 0, 0);
    }
  };
  // (the prefetches are unconditional, so the loop body is one straight path and a multiply waits only for its own, older,
  // loads; the scheduling barriers keep the machine scheduler from sinking a prefetch down to its first use.  Prefetching the
  // slab rows a step ahead as well measured slower: 155 vs 147 us per 65536-row forward pass)
  for (int k0 = 0; k0 < K; k0 += 2 * SBK) {
    wload<C>(w1, Bg, ldb, K, k0 + SBK);
    __builtin_amdgcn_sched_barrier(0);
    mul(w0, k0);
    __builtin_amdgcn_sched_barrier(0);
    wload<C>(w0, Bg, ldb, K, k0 + 2 * SBK);
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + SBK < K) mul(w1, k0 + SBK);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Epilogue of a 256-wide layer: v = acc (+ bias[n]) (ReLU) (masked by mask[row][n] > 0); to HBM (row-major, ld SH) and, with
// TO_SLAB, into the slab as the next layer's operand.  The accumulators hold the transposed tile (see slab_mma): lane l of tile
// (i, j) owns slab row 32 i + l % 32 and, in registers 4 g .. 4 g + 3, the output units (wave's first) + 32 j + 8 g + 4 (l / 32) +
// 0..3 -- one 16-byte store (and one 16-byte mask / bias load) per register quad.  Rows beyond R are computed (finite values
// from zero inputs) but never stored to HBM.
template <class C, bool TO_SLAB, bool FULL>
__device__ __forceinline__ void store_act_t(StripLds<C>& L, const f32x16 (&acc)[C::RT][C::CT], const float* __restrict__ bias, const bool relu,
                                            const float* __restrict__ mask, float* __restrict__ out, const int row0, const int R,
                                            unsigned* __restrict__ bits_out, const unsigned* __restrict__ bits_in) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  // ReLU masks as BITS (round 6): the forward pass leaves, per thread and column tile, one word with bit (g RT + i) 4 + c set where its
  // output is positive; the backward pass -- same workgroup shape, same thread-to-element map -- reads that word instead of sixteen
  // 16-byte loads of the activations themselves (2 MB instead of 33.5 MB per layer and 32768 rows)
  unsigned wbits[C::CT];
#pragma unroll
  for (int j = 0; j < C::CT; j++) wbits[j] = bits_in ? bits_in[((size_t)blockIdx.x * C::THR + tid) * C::CT + j] : 0u;
#pragma unroll
  for (int j = 0; j < C::CT; j++) {
    const int nb = wave * 32 * C::CT + 32 * j + 4 * kh;
    // the mask values of this column tile first, as one batch of independent loads (interleaved with the stores below they
    // would be serialised: the compiler cannot prove that `out` does not alias `mask`)
    float4 mk[C::RT][4];
#pragma unroll
    for (int i = 0; i < C::RT; i++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 32 * i + l31;
        mk[i][g] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mask && !bits_in && (FULL || row0 + row < R)) mk[i][g] = *reinterpret_cast<const float4*>(mask + (size_t)(row0 + row) * SH + nb + 8 * g);
      }
    unsigned ob = 0u;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int n = nb + 8 * g;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
      for (int i = 0; i < C::RT; i++) {
        const int row = 32 * i + l31;
        const bool live = FULL || row0 + row < R;
        float4 v = make_float4(acc[i][j][4 * g] + bv.x, acc[i][j][4 * g + 1] + bv.y, acc[i][j][4 * g + 2] + bv.z, acc[i][j][4 * g + 3] + bv.w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const int b0 = (g * C::RT + i) * 4;
        if (bits_in) {
          const unsigned w = wbits[j] >> b0;
          v.x = (live && (w & 1u)) ? v.x : 0.f; v.y = (live && (w & 2u)) ? v.y : 0.f;
          v.z = (live && (w & 4u)) ? v.z : 0.f; v.w = (live && (w & 8u)) ? v.w : 0.f;
        } else if (mask) {
          v.x = (live && mk[i][g].x > 0.f) ? v.x : 0.f; v.y = (live && mk[i][g].y > 0.f) ? v.y : 0.f;
          v.z = (live && mk[i][g].z > 0.f) ? v.z : 0.f; v.w = (live && mk[i][g].w > 0.f) ? v.w : 0.f;
        }
        if (bits_out) ob |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << b0;
        if (TO_SLAB) { L.S[n][row] = v.x; L.S[n + 1][row] = v.y; L.S[n + 2][row] = v.z; L.S[n + 3][row] = v.w; }
        if (live && out) *reinterpret_cast<float4*>(out + (size_t)(row0 + row) * SH + n) = v;   // (out == NULL: inference, the hidden layers stay in LDS)
      }
    }
    if (bits_out) bits_out[((size_t)blockIdx.x * C::THR + tid) * C::CT + j] = ob;
  }
}
template <class C, bool TO_SLAB>
__device__ __forceinline__ void store_act(StripLds<C>& L, const f32x16 (&acc)[C::RT][C::CT], const float* __restrict__ bias, const bool relu,
                                          const float* __restrict__ mask, float* __restrict__ out, const int row0, const int R,
                                          unsigned* __restrict__ bits_out = nullptr, const unsigned* __restrict__ bits_in = nullptr) {
  if (row0 + C::ROWS <= R) store_act_t<C, TO_SLAB, true>(L, acc, bias, relu, mask, out, row0, R, bits_out, bits_in);   // (all but the last slab: no per-row tests)
  else store_act_t<C, TO_SLAB, false>(L, acc, bias, relu, mask, out, row0, R, bits_out, bits_in);
}

// stage a [rows][K] row-major slab (row stride ld) k-major into X, zero-padded to a multiple of SBK in k and beyond R in rows
// (mean / stdv: the input is a raw observation row of in_dim entries, normalised on the way in -- the same expression as
// normalize_kernel, so the update, which normalises its minibatches there, sees the same bits)
template <class C>
__device__ __forceinline__ void stage_input(float (*X)[C::LD], const float* __restrict__ x, const int ld, const int K, const int row0, const int R,
                                            const float* __restrict__ mean = nullptr, const float* __restrict__ stdv = nullptr, const int in_dim = 0) {
  const int Kp = (K + SBK - 1) & ~(SBK - 1);
  if (!mean && !(ld & 3) && !(reinterpret_cast<size_t>(x) & 15)) {
    // 16-byte rows (the update's minibatches: x [R][Dp], dy [R][Op]): every thread issues ALL its 16-byte loads before the first LDS write
    // -- as a loop of load -> write round trips (twelve of them, dependent) this stage took 5.4 us of a 64 us forward pass
    // (profiles/r06_strip_clock.txt)
    const int k4 = Kp >> 2;
    constexpr int NQ = (C::ROWS * (SXK / 4) + C::THR - 1) / C::THR;
    const float rk4 = 1.f / (float)k4;
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int idx = (int)threadIdx.x + q * C::THR, r = (int)(((float)idx + 0.5f) * rk4), k = 4 * (idx - r * k4);
      v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < C::ROWS * k4 && k < K && row0 + r < R) {
        v[q] = *reinterpret_cast<const float4*>(x + (size_t)(row0 + r) * ld + k);
        if (k + 1 >= K) v[q].y = 0.f;
        if (k + 2 >= K) v[q].z = 0.f;
        if (k + 3 >= K) v[q].w = 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int idx = (int)threadIdx.x + q * C::THR, r = (int)(((float)idx + 0.5f) * rk4), k = 4 * (idx - r * k4);
      if (idx < C::ROWS * k4) { X[k][r] = v[q].x; X[k + 1][r] = v[q].y; X[k + 2][r] = v[q].z; X[k + 3][r] = v[q].w; }
    }
    return;
  }
  for (int i = threadIdx.x; i < C::ROWS * Kp; i += C::THR) {
    const int r = i / Kp, k = i - r * Kp;
    float v = 0.f;
    if (mean) {
      if (k < in_dim && row0 + r < R) v = (x[(size_t)(row0 + r) * ld + k] - mean[k]) / stdv[k];
    } else if (k < K && row0 + r < R) v = x[(size_t)(row0 + r) * ld + k];
    X[k][r] = v;
  }
}

template <class C>
__global__ void __launch_bounds__(C::THR, 2) mlp_fwd_strip_kernel(MlpStripFwd a) {
  __shared__ StripLds<C> L;
  LHW_LDS_POISON(L);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int row0 = (int)blockIdx.x * C::ROWS;
  float (*X)[C::LD] = &L.S[SH - SXK];
  WOp<C> w;
  SCLK(0);
  wload<C>(w, a.w1t, SH, a.Dp, 0);   // (the first weights of a layer are in flight while the slab is staged / the previous epilogue runs)
  stage_input<C>(X, a.x, a.ldx, a.Dp, row0, a.R, a.in_mean, a.in_std, a.in_dim);
  __syncthreads();
  SCLK(1);
  f32x16 acc[C::RT][C::CT];
  zero_acc<C>(acc);
  slab_mma<C>(X, a.Dp, a.w1t, SH, w, acc);
  wload<C>(w, a.w2t, SH, SH, 0);
  SCLK(2);
  __syncthreads();   // every wave is done with the input slab
  SCLK(3);
  store_act<C, true>(L, acc, a.b1, true, nullptr, a.h1, row0, a.R, a.bits1);
  SCLK(4);
  __syncthreads();
  SCLK(5);
  zero_acc<C>(acc);
  slab_mma<C>(L.S, SH, a.w2t, SH, w, acc);
  SCLK(6);
  __syncthreads();
  SCLK(7);
  store_act<C, true>(L, acc, a.b2, true, nullptr, a.h2, row0, a.R, a.bits2);
  SCLK(8);
  __syncthreads();
  SCLK(9);
  // read-out: y = h2 W3^T + b3, N = O <= 32: one column tile.  The K range is cut into SH / SKQ partial products of SKQ k each
  // (the same cut for every workgroup shape, so every shape returns the same bits); a wave takes QW consecutive ones for all
  // row tiles, and the partials are summed in ascending k order afterwards.
  constexpr int NQ = SH / SKQ, QW = NQ / C::NW;
  f32x16 p[QW][C::RT];
#pragma unroll
  for (int q = 0; q < QW; q++)
#pragma unroll
    for (int i = 0; i < C::RT; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) p[q][i][r] = 0.f;
#pragma unroll
  for (int q = 0; q < QW; q++) {
#pragma unroll 8
    for (int kk = 0; kk < SKQ / 2; kk++) {
      const int k = (wave * QW + q) * SKQ + kk * 2 + kh;
      const float bv = a.w3t[(size_t)k * a.Op + min(l31, a.O - 1)], b = l31 < a.O ? bv : 0.f;
#pragma unroll
      for (int i = 0; i < C::RT; i++) p[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.S[k][32 * i + l31], b, p[q][i], 0, 0, 0);
    }
  }
  SCLK(10);
  __syncthreads();   // all waves are done with the slab: it now holds the partial products
  SCLK(11);
  float (*P)[C::ROWS][32] = reinterpret_cast<float (*)[C::ROWS][32]>(&L.S[0][0]);     // [NQ][ROWS][32]: NQ * ROWS * 128 B <= the slab
#pragma unroll
  for (int q = 0; q < QW; q++)
#pragma unroll
    for (int i = 0; i < C::RT; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) P[wave * QW + q][32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh][l31] = p[q][i][r];
  __syncthreads();
  for (int i = tid; i < C::ROWS * 32; i += C::THR) {
    const int row = i >> 5, col = i & 31;
    if (col < a.O && row0 + row < a.R) {
      float s = P[0][row][col];
#pragma unroll
      for (int q = 1; q < NQ; q++) s += P[q][row][col];
      s += a.b3[col];
      a.y[(size_t)(row0 + row) * a.Op + col] = s;
      if (a.act) {   // Gaussian head (rollout inference): the action component here, its log-density term through P[0] (this
        float term;  // thread's own, now spent, entry) to the row's first thread below
        a.act[(size_t)(row0 + row) * a.O + col] = lhw_policy_sample(s, a.stdv[col], a.seed, a.env_base + (unsigned)(row0 + row), a.counter, col, a.deterministic, &term);
        P[0][row][col] = term;
      }
    }
  }
  SCLK(12);
  if (a.act) {
    __syncthreads();
    for (int row = tid; row < C::ROWS; row += C::THR)
      if (row0 + row < a.R) {
        float lp = 0.f;
        for (int k = 0; k < a.O; k++) lp += P[0][row][k];     // (the order of sample_kernel's sum)
        a.logp[row0 + row] = lp;
      }
  }
}

template <class C>
__global__ void __launch_bounds__(C::THR, 2) mlp_bwd_strip_kernel(MlpStripBwd a) {
  __shared__ StripLds<C> L;
  LHW_LDS_POISON(L);
  const int row0 = (int)blockIdx.x * C::ROWS;
  float (*X)[C::LD] = &L.S[SH - SXK];
  WOp<C> w;
  SCLK(0);
  wload<C>(w, a.w3, SH, a.O, 0);
  stage_input<C>(X, a.dy, a.Op, a.O, row0, a.R);
  __syncthreads();
  SCLK(1);
  f32x16 acc[C::RT][C::CT];
  zero_acc<C>(acc);
  slab_mma<C>(X, a.O, a.w3, SH, w, acc);                                    // dy W3: B[k = o][n] = W3[o][n]
  wload<C>(w, a.w2, SH, SH, 0);
  SCLK(2);
  __syncthreads();
  SCLK(3);
  store_act<C, true>(L, acc, nullptr, false, a.h2, a.dh2, row0, a.R, nullptr, a.bits2);
  SCLK(4);
  __syncthreads();
  SCLK(5);
  zero_acc<C>(acc);
  slab_mma<C>(L.S, SH, a.w2, SH, w, acc);                                   // dh2 W2: B[k = o][n = i] = W2[o][i]
  SCLK(6);
  store_act<C, false>(L, acc, nullptr, false, a.h1, a.dh1, row0, a.R, nullptr, a.bits1);
  SCLK(7);
}

// WT [cols][rows] <- W [rows][ld] for three matrices in one launch (the forward pass multiplies by W^T: its weight operand must
// have the output unit contiguous); 32 x 32 tiles through LDS, block b of matrix m handles tile b - first[m]
struct TransposeJob { const float* W; float* WT; int rows, cols, ld, ldt, first; };
struct TransposeJobs { TransposeJob j[3]; };
__global__ void __launch_bounds__(256) transpose3_kernel(TransposeJobs J) {
  __shared__ float T[32][33];
  LHW_LDS_POISON(T);
  const int b = (int)blockIdx.x, m = b >= J.j[2].first ? 2 : (b >= J.j[1].first ? 1 : 0);
  const TransposeJob q = J.j[m];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int tc = (q.cols + 31) / 32, t = b - q.first, r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  for (int y = ty; y < 32; y += 8) T[y][tx] = (r0 + y < q.rows && c0 + tx < q.cols) ? q.W[(size_t)(r0 + y) * q.ld + c0 + tx] : 0.f;
  __syncthreads();
  for (int y = ty; y < 32; y += 8)
    if (c0 + y < q.cols && r0 + tx < q.rows) q.WT[(size_t)(c0 + y) * q.ldt + r0 + tx] = T[tx][y];
}

size_t mlp_strip_wt_floats(int Dp, int Op) { return (size_t)Dp * SH + (size_t)SH * SH + (size_t)SH * Op; }

// transposed copies of the three weight matrices ([in][out]) for the forward strip kernel, into wt (mlp_strip_wt_floats floats)
void mlp_strip_prepare(const float* w1, const float* w2, const float* w3, int Dp, int O, int Op, float* wt, hipStream_t s) {
  float *w1t = wt, *w2t = wt + (size_t)Dp * SH, *w3t = w2t + (size_t)SH * SH;
  auto tiles = [](int rows, int cols) { return ((rows + 31) / 32) * ((cols + 31) / 32); };
  TransposeJobs J;
  J.j[0] = TransposeJob{w1, w1t, SH, Dp, Dp, SH, 0};
  J.j[1] = TransposeJob{w2, w2t, SH, SH, SH, SH, tiles(SH, Dp)};
  J.j[2] = TransposeJob{w3, w3t, O, SH, SH, Op, tiles(SH, Dp) + tiles(SH, SH)};
  hipLaunchKernelGGL(transpose3_kernel, dim3(J.j[2].first + tiles(O, SH)), dim3(256), 0, s, J);
}

size_t mlp_strip_bits_words(size_t rows) { return (rows + StripBig::ROWS - 1) / StripBig::ROWS * StripBig::THR * StripBig::CT; }

bool mlp_strip_supported(int H, int Dp, int O, int Op) { return H == SH && Dp > 0 && Dp <= SXK && (Dp & 3) == 0 && O > 0 && O <= 32 && Op >= O; }

// Rows up to which the small shape is used: below it the slabs of the big shape would not even fill the CUs once, and the call
// is latency-bound (rollout inference); above it the big shape's operand reuse wins (the update's minibatches).
#define STRIP_SMALL_ROWS 16384
void mlp_strip_forward(const MlpStripFwd& a, hipStream_t s, int shape /* 0: by row count, 1: small, 2: big */) {
  if (a.R <= 0) return;
  if ((shape == 1 || (shape == 0 && a.R <= STRIP_SMALL_ROWS)) && !a.bits1 && !a.bits2)   // (mask bits: the backward kernel's shape)
    hipLaunchKernelGGL((mlp_fwd_strip_kernel<StripSmall>), dim3((a.R + StripSmall::ROWS - 1) / StripSmall::ROWS), dim3(StripSmall::THR), 0, s, a);
  else
    hipLaunchKernelGGL((mlp_fwd_strip_kernel<StripBig>), dim3((a.R + StripBig::ROWS - 1) / StripBig::ROWS), dim3(StripBig::THR), 0, s, a);
}

void mlp_strip_backward(const MlpStripBwd& a, hipStream_t s) {
  if (a.R <= 0) return;
  hipLaunchKernelGGL((mlp_bwd_strip_kernel<StripBig>), dim3((a.R + StripBig::ROWS - 1) / StripBig::ROWS), dim3(StripBig::THR), 0, s, a);
}

extern "C" int lhw_debug_mlp_strip_forward_bits(int32_t H, int32_t Dp, int32_t O, int32_t Op, const float* w1, const float* b1, const float* w2,
                                                const float* b2, const float* w3, const float* b3, const float* x, int32_t ldx, int32_t R,
                                                float* h1, float* h2, float* y, float* wt_scratch, uint32_t* bits1, uint32_t* bits2, void* stream) {
  if (!wt_scratch || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !x || !h1 || !h2 || !y) return lhw_fail(LHW_ERR_ARG, "null argument");
  if (!mlp_strip_supported(H, Dp, O, Op) || ldx < Dp) return lhw_fail(LHW_ERR_UNSUPPORTED, "strip kernels: hidden width 256, padded input width <= 64, outputs <= 32");
  mlp_strip_prepare(w1, w2, w3, Dp, O, Op, wt_scratch, (hipStream_t)stream);
  MlpStripFwd a{wt_scratch, b1, wt_scratch + (size_t)Dp * SH, b2, wt_scratch + (size_t)Dp * SH + (size_t)SH * SH, b3, x, ldx, Dp, O, Op, R, h1, h2, y};
  a.bits1 = bits1; a.bits2 = bits2;
  const char* sh = getenv("LHW_DEBUG_STRIP_SHAPE");     // (tests: "small" / "big" force the workgroup shape; default by row count)
  mlp_strip_forward(a, (hipStream_t)stream, sh ? (sh[0] == 's' ? 1 : 2) : 0);
  return hipGetLastError() == hipSuccess ? LHW_OK : lhw_fail(LHW_ERR_HIP, "mlp_fwd_strip_kernel launch failed");
}
extern "C" int lhw_debug_mlp_strip_forward(int32_t H, int32_t Dp, int32_t O, int32_t Op, const float* w1, const float* b1, const float* w2,
                                           const float* b2, const float* w3, const float* b3, const float* x, int32_t ldx, int32_t R,
                                           float* h1, float* h2, float* y, float* wt_scratch, void* stream) {
  return lhw_debug_mlp_strip_forward_bits(H, Dp, O, Op, w1, b1, w2, b2, w3, b3, x, ldx, R, h1, h2, y, wt_scratch, nullptr, nullptr, stream);
}

extern "C" int lhw_debug_mlp_strip_backward_bits(int32_t H, int32_t O, int32_t Op, const float* w2, const float* w3, const float* dy, int32_t R,
                                                 const float* h1, const float* h2, float* dh2, float* dh1, const uint32_t* bits1, const uint32_t* bits2,
                                                 void* stream) {
  if (!w2 || !w3 || !dy || !dh2 || !dh1 || (!h1 && !bits1) || (!h2 && !bits2)) return lhw_fail(LHW_ERR_ARG, "null argument");
  if (!mlp_strip_supported(H, 4, O, Op)) return lhw_fail(LHW_ERR_UNSUPPORTED, "strip kernels: hidden width 256, outputs <= 32");
  MlpStripBwd a{w2, w3, dy, h1, h2, O, Op, R, dh2, dh1};
  a.bits1 = bits1; a.bits2 = bits2;
  mlp_strip_backward(a, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? LHW_OK : lhw_fail(LHW_ERR_HIP, "mlp_bwd_strip_kernel launch failed");
}
extern "C" int lhw_debug_mlp_strip_backward(int32_t H, int32_t O, int32_t Op, const float* w2, const float* w3, const float* dy, int32_t R,
                                            const float* h1, const float* h2, float* dh2, float* dh1, void* stream) {
  return lhw_debug_mlp_strip_backward_bits(H, O, Op, w2, w3, dy, R, h1, h2, dh2, dh1, nullptr, nullptr, stream);
}

// the rollout's fused policy step (normalisation -> three layers -> Gaussian head) on R observation rows, from the actor view the
// resident rollout reads: the launch lhw_ppo_forward_at issues per control step, reachable without an LhwPpo (tests, SIMT emulator)
extern "C" int lhw_debug_policy_step(const LhwRolloutPolicy* q, const float* obs, int32_t R, uint32_t env_id_base, uint32_t counter, float* y,
                                     float* act, float* logp, void* stream) {
  if (!q || !obs || !y || !act || !logp || R <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  if (!mlp_strip_supported(q->hidden, q->obs_pad, q->act_dim, q->act_pad)) return lhw_fail(LHW_ERR_UNSUPPORTED, "strip kernels: hidden width 256, padded input width <= 64, outputs <= 32");
  MlpStripFwd a{q->w1t, q->b1, q->w2t, q->b2, q->w3t, q->b3, obs, q->obs_dim, q->obs_pad, q->act_dim, q->act_pad, R, nullptr, nullptr, y};
  a.in_mean = q->obs_mean; a.in_std = q->obs_std; a.in_dim = q->obs_dim;
  a.stdv = q->stdv; a.act = act; a.logp = logp;
  a.seed = q->seed; a.env_base = env_id_base; a.counter = counter; a.deterministic = q->deterministic;
  const char* sh = getenv("LHW_DEBUG_STRIP_SHAPE");
  mlp_strip_forward(a, (hipStream_t)stream, sh ? (sh[0] == 's' ? 1 : 2) : 0);
  return hipGetLastError() == hipSuccess ? LHW_OK : lhw_fail(LHW_ERR_HIP, "mlp_fwd_strip_kernel launch failed");
}
