// On-device PPO: actor/critic MLP forward, GAE, clipped-surrogate loss + backward, dual
// grad-norm clip + Adam.  Takes over the arithmetic of
//   Gaussian_FF_Actor / FF_V forward      (reference rl/policies/actor.py:160-188, critic.py:41-49)
//   PPOBuffer.finish_path (GAE)           (reference rl/storage/rollout_storage.py:53-85)
//   PPO.update_actor_critic               (reference rl/algos/ppo.py:299-406)
//   advantage normalisation               (reference rl/algos/ppo.py:484-485)
//
// Numerics: network math is float32 like the reference (ATen fp32).  The dense layers run on
// the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit an fmaf chain, so results
// differ from ATen only by summation order.  GAE accumulates in float64 like the reference.
//
// GEMM design (one kernel, three operand layouts): 64x64 output tile per 256-thread workgroup,
// 4 waves in a 2x2 grid of 32x32 MFMA blocks, K staged through LDS 16 at a time, K-major LDS
// tiles so the one-float-per-lane MFMA operands are conflict-free ds_read_b32; global loads are
// 16-byte vectors, register-prefetched one tile ahead.  Epilogues fuse bias+ReLU (forward),
// ReLU-mask (backward-data), column sums (bias gradients) and split-K atomic accumulation
// (backward-weight, where the contraction runs over the minibatch).
#include <hip/hip_runtime.h>

#include <mutex>

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/lhw.h"
#include "lhw_internal.h"
#include "lhw_policy.h"
#include "lhw_rng.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 16
#define LDS_LD (64 + 4)

struct GemmArgs {
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  int M, N, K;
  const float* bias;        // + bias[n]
  int relu;                 // max(0, .)
  const float* mask; int ldmask;  // *= (mask[m][n] > 0)
  float* part;              // split-K: slice z stores its partial product at part + z*M*N (row-major, ld N); reduced in slice order afterwards
  int k_chunk;              // K range per blockIdx.z
  float* colsum;            // (A stored [K][M] only) slice z also stores sum_k A[k][m] at colsum + z*M: the bias gradient of a dW GEMM
  int tiles_m, tiles_n, slices;   // filled in by launch_gemm
  // fp16 GEMM only (gemm_h_kernel): which of the buffers hold _Float16 instead of float (the pointers above are then reinterpreted;
  // leading dimensions count elements of the buffer's own type).  Operands stored as float are rounded to fp16 while they are staged.
  int a_half, b_half, c_half, mask_half;
};

// C = op(A) op(B) on v_mfma_f32_32x32x2_f32.  Block = 4 waves (2 x 2), each wave owns WT x WT MFMA tiles of 32 x 32:
// block tile 64 x 64 (WT = 1) or 128 x 128 (WT = 2; one LDS operand read per MFMA instead of two).  K advances in steps of
// BK = 16 through double-buffered LDS tiles: the global loads of step k+1 are in flight while step k is multiplied and there
// is one barrier per step.
// A_KC: A is stored [M][K] (K contiguous); else A is stored [K][M] (M contiguous) i.e. we multiply by its transpose.
// B_KC: B is stored [N][K] (K contiguous); else B is stored [K][N].
template <bool A_KC, bool B_KC, int WT>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs g) {
  constexpr int TM = 64 * WT, LD = TM + 4;
  __shared__ float As[2][BK][LD];
  LHW_LDS_POISON(As);
  __shared__ float Bs[2][BK][LD];
  LHW_LDS_POISON(Bs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, so XCD x takes the contiguous
  // range [x * per, (x + 1) * per) of the order (n tile fastest, then m tile, then k slice): the blocks that share an A tile
  // (forward / activation-gradient GEMMs: the n tiles of one m tile) or a k slice of both operands (weight-gradient GEMMs: all
  // tiles of the slice) run back to back on ONE L2 instead of being spread over all eight.
  const int per = (int)gridDim.x >> 3, v = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (v >= g.tiles_m * g.tiles_n * g.slices) return;
  const int tn = v % g.tiles_n, tm = (v / g.tiles_n) % g.tiles_m, bz = v / (g.tiles_n * g.tiles_m);
  const int m0 = tm * TM, n0 = tn * TM;
  const int kbeg = bz * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);
  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; i++)
#pragma unroll
    for (int j = 0; j < WT; j++)
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // operand staging: KC layout -> 4 / WT threads per row, each 4 * WT consecutive k; else 16 threads per k, each 4 * WT rows
  float4 ra[WT], rb[WT];
  auto load_tile = [&](float4 (&r)[WT], const float* __restrict__ P, int ld, bool kc, int x0, int X, int k0) {
#pragma unroll
    for (int q = 0; q < WT; q++) {
      r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kc) {
        const int row = x0 + tid / (4 / WT), k = k0 + (tid % (4 / WT)) * 4 * WT + 4 * q;
        if (row < X && k < kend) {
          r[q] = *reinterpret_cast<const float4*>(P + (size_t)row * ld + k);
          if (k + 1 >= kend) r[q].y = 0.f;
          if (k + 2 >= kend) r[q].z = 0.f;
          if (k + 3 >= kend) r[q].w = 0.f;
        }
      } else {
        const int k = k0 + (tid >> 4), x = x0 + (tid & 15) * 4 * WT + 4 * q;
        if (k < kend && x < X) {
          r[q] = *reinterpret_cast<const float4*>(P + (size_t)k * ld + x);
          if (x + 1 >= X) r[q].y = 0.f;
          if (x + 2 >= X) r[q].z = 0.f;
          if (x + 3 >= X) r[q].w = 0.f;
        }
      }
    }
  };
  auto store_tile = [&](float (&T)[BK][LD], const float4 (&r)[WT], bool kc) {
#pragma unroll
    for (int q = 0; q < WT; q++) {
      if (kc) {
        const int row = tid / (4 / WT), kq = (tid % (4 / WT)) * 4 * WT + 4 * q;
        T[kq + 0][row] = r[q].x; T[kq + 1][row] = r[q].y; T[kq + 2][row] = r[q].z; T[kq + 3][row] = r[q].w;
      } else {
        const int k = tid >> 4, xq = (tid & 15) * 4 * WT + 4 * q;
        *reinterpret_cast<float4*>(&T[k][xq]) = r[q];
      }
    }
  };

  const bool want_colsum = !A_KC && g.colsum != nullptr && tn == 0;
  float cs = 0.f;   // thread (m = tid % TM, k group = tid / TM): running sum of its A-tile entries
  int cur = 0;
  if (kbeg < kend) {
    load_tile(ra, g.A, g.lda, A_KC, m0, g.M, kbeg);
    load_tile(rb, g.B, g.ldb, B_KC, n0, g.N, kbeg);
    store_tile(As[0], ra, A_KC);
    store_tile(Bs[0], rb, B_KC);
  }
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      load_tile(ra, g.A, g.lda, A_KC, m0, g.M, k0 + BK);
      load_tile(rb, g.B, g.ldb, B_KC, n0, g.N, k0 + BK);
    }
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) {
      float a[WT], b[WT];
#pragma unroll
      for (int i = 0; i < WT; i++) a[i] = As[cur][kk * 2 + kh][(wm * WT + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < WT; j++) b[j] = Bs[cur][kk * 2 + kh][(wn * WT + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < WT; i++)
#pragma unroll
        for (int j = 0; j < WT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (want_colsum) {
      constexpr int KG = BK * TM / 256;   // k rows per thread
      const int m = tid % TM, kg = tid / TM;
#pragma unroll
      for (int q = 0; q < KG; q++) cs += As[cur][kg * KG + q][m];
    }
    if (more) {
      store_tile(As[cur ^ 1], ra, A_KC);
      store_tile(Bs[cur ^ 1], rb, B_KC);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue; C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* part = g.part ? g.part + (size_t)bz * g.M * g.N : nullptr;
#pragma unroll
  for (int j = 0; j < WT; j++) {
    const int col = n0 + (wn * WT + j) * 32 + (lane & 31);
    const float bias = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < WT; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (wm * WT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] + bias;
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.mask) v = g.mask[(size_t)row * g.ldmask + col] > 0.f ? v : 0.f;
          if (part) part[(size_t)row * g.N + col] = v;
          else g.C[(size_t)row * g.ldc + col] = v;
        }
      }
    }
  }
  if (want_colsum) {   // combine the k groups in a fixed order (the tiles are idle now: the loop ended with a barrier)
    constexpr int NG = 256 / TM;
    float* red = &As[0][0][0];
    red[tid] = cs;
    __syncthreads();
    if (tid < TM && m0 + tid < g.M) {
      float s = red[tid];
#pragma unroll
      for (int q = 1; q < NG; q++) s += red[q * TM + tid];
      g.colsum[(size_t)bz * g.M + m0 + tid] = s;
    }
  }
}

// Half-precision GEMM (BASELINE config 5, "fp16 actor/critic on CDNA4"): the block order, epilogues, split-K and fused column sums of
// gemm_f32_kernel on gfx950's v_mfma_f32_32x32x16_f16 with float32 accumulation.  Round 6: the operands may LIVE in fp16 in HBM
// (a_half / b_half: the update's activations x, h1, h2 and back-propagated gradients dh2, dh1 -- written as fp16 by the GEMM that
// produces them, c_half) and are then loaded as 16-byte f16x8 vectors with no conversion; operands stored as float32 (the master weights,
// the loss gradients) are rounded while they are staged, as every operand was through round 5.  K advances 32 per step (two MFMAs per
// wave and barrier instead of one per 16-k step).  Bias / ReLU / mask in float32; outputs float32 or fp16; split-K partials float32.
// Used by the rollout inference with fp16 operands (lhw_ppo_set_inference_dtype) and by every GEMM of the --fp16 update.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define HBK 32
#define HLD (HBK + 8)      // tile rows 80 bytes apart: 16-byte aligned operand reads, conflict-free across the 32 rows of a wave's read
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) gemm_h_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2][BM][HLD];
  LHW_LDS_POISON(Ah);
  __shared__ __attribute__((aligned(16))) _Float16 Bh[2][BN][HLD];
  LHW_LDS_POISON(Bh);
  __shared__ float red[256];
  LHW_LDS_POISON(red);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int per = (int)gridDim.x >> 3, v = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);   // XCD-aware order (gemm_f32_kernel)
  if (v >= g.tiles_m * g.tiles_n * g.slices) return;
  const int tn = v % g.tiles_n, tm = (v / g.tiles_n) % g.tiles_m, bz = v / (g.tiles_n * g.tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = bz * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);
  f32x16 acc;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  // staging: 8 elements per thread and operand.  KC layout ([X][K]): 4 threads per row, 8 consecutive k each; else ([K][X]): 8 threads
  // per k, 8 consecutive x each.  Elements beyond the matrix / the slice are zero.
  auto load_tile = [&](const float* __restrict__ P, int ld, bool kc, bool is_half, int x0, int X, int k0) -> f16x8 {
    f16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
    const int row = kc ? x0 + (tid >> 2) : k0 + (tid >> 3);        // index along the leading dimension
    const int col = kc ? k0 + (tid & 3) * 8 : x0 + (tid & 7) * 8;  // first of the 8 contiguous elements
    const int rlim = kc ? X : kend, clim = kc ? kend : X;
    if (row < rlim && col < clim) {
      if (is_half) {
        const _Float16* Ph = reinterpret_cast<const _Float16*>(P) + (size_t)row * ld + col;
        if (col + 8 <= clim && !(((size_t)row * ld + col) & 7)) r = *reinterpret_cast<const f16x8*>(Ph);
        else for (int j = 0; j < 8; j++) if (col + j < clim) r[j] = Ph[j];
      } else {
        const float* Pf = P + (size_t)row * ld + col;
        if (col + 8 <= clim) {
          const float4 u = *reinterpret_cast<const float4*>(Pf), w = *reinterpret_cast<const float4*>(Pf + 4);
          r = f16x8{(_Float16)u.x, (_Float16)u.y, (_Float16)u.z, (_Float16)u.w, (_Float16)w.x, (_Float16)w.y, (_Float16)w.z, (_Float16)w.w};
        } else for (int j = 0; j < 8; j++) if (col + j < clim) r[j] = (_Float16)Pf[j];
      }
    }
    return r;
  };
  auto store_tile = [&](_Float16 (&T)[BM][HLD], const f16x8& r, bool kc) {
    if (kc) *reinterpret_cast<f16x8*>(&T[tid >> 2][(tid & 3) * 8]) = r;
    else {
      const int k = tid >> 3, xq = (tid & 7) * 8;
#pragma unroll
      for (int j = 0; j < 8; j++) T[xq + j][k] = r[j];
    }
  };
  const bool want_colsum = !A_KC && g.colsum != nullptr && tn == 0;
  float cs = 0.f;
  int cur = 0;
  f16x8 ra, rb;
  if (kbeg < kend) {
    ra = load_tile(g.A, g.lda, A_KC, g.a_half != 0, m0, g.M, kbeg);
    rb = load_tile(g.B, g.ldb, B_KC, g.b_half != 0, n0, g.N, kbeg);
    store_tile(Ah[0], ra, A_KC);
    store_tile(Bh[0], rb, B_KC);
  }
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += HBK) {
    const bool more = k0 + HBK < kend;
    if (more) {
      ra = load_tile(g.A, g.lda, A_KC, g.a_half != 0, m0, g.M, k0 + HBK);
      rb = load_tile(g.B, g.ldb, B_KC, g.b_half != 0, n0, g.N, k0 + HBK);
    }
    const int am = wm * 32 + (lane & 31), bn = wn * 32 + (lane & 31), kh = lane >> 5;
    // v_mfma_f32_32x32x16_f16: lane l supplies k = 8 (l / 32) .. + 7 of its row / column: one 16-byte LDS read per operand
#pragma unroll
    for (int kk = 0; kk < HBK / 16; kk++) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(&Ah[cur][am][kk * 16 + kh * 8]);
      const f16x8 b = *reinterpret_cast<const f16x8*>(&Bh[cur][bn][kk * 16 + kh * 8]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    if (want_colsum) {   // thread (m = tid % 64, k group = tid / 64): the rounded entries it would also have multiplied
      const int m = tid & 63, kg = tid >> 6;
#pragma unroll
      for (int q = 0; q < 8; q++) cs += (float)Ah[cur][m][kg * 8 + q];
    }
    if (more) {
      store_tile(Ah[cur ^ 1], ra, A_KC);
      store_tile(Bh[cur ^ 1], rb, B_KC);
    }
    __syncthreads();
    cur ^= 1;
  }
  const int col = n0 + wn * 32 + (lane & 31);
  const float bias = (g.bias && col < g.N) ? g.bias[col] : 0.f;
  float* part = g.part ? g.part + (size_t)bz * g.M * g.N : nullptr;
  const _Float16* maskh = reinterpret_cast<const _Float16*>(g.mask);
  _Float16* Ch = reinterpret_cast<_Float16*>(g.C);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < g.M && col < g.N) {
      float v2 = acc[r] + bias;
      if (g.relu) v2 = fmaxf(v2, 0.f);
      if (g.mask) {
        const bool on = g.mask_half ? (float)maskh[(size_t)row * g.ldmask + col] > 0.f : g.mask[(size_t)row * g.ldmask + col] > 0.f;
        v2 = on ? v2 : 0.f;
      }
      if (part) part[(size_t)row * g.N + col] = v2;
      else if (g.c_half) Ch[(size_t)row * g.ldc + col] = (_Float16)v2;
      else g.C[(size_t)row * g.ldc + col] = v2;
    }
  }
  if (want_colsum) {
    red[tid] = cs;
    __syncthreads();
    if (tid < 64 && m0 + tid < g.M) g.colsum[(size_t)bz * g.M + m0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
  }
}

// out[row*ldc + col] += sum over slices (in slice order) of part[z][row*N + col]: deterministic split-K reduction
__global__ void __launch_bounds__(256) reduce_slices_kernel(const float* __restrict__ part, int nslices, int M, int N, float* __restrict__ out, int ldc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  float s = 0.f;
  for (int z = 0; z < nslices; z++) s += part[(size_t)z * M * N + i];
  const int row = i / N, col = i - row * N;
  out[(size_t)row * ldc + col] += s;
}

// Deterministic column sums, two stages.  Stage 1: block (col tile, row chunk) sums its rows of 64 columns (4 strided
// row groups combined in a fixed order) into part[chunk][col].  Stage 2: out[col] += sum over chunks in chunk order.
#define COLSUM_CHUNKS 128
__global__ void __launch_bounds__(256) colsum_det_kernel(const float* __restrict__ X, int rows, int ld, int ncols, float* __restrict__ part) {
  __shared__ float red[4][64];
  LHW_LDS_POISON(red);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const int per = (rows + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS, r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  if (c < ncols)
    for (int r = r0 + g; r < r1; r += 4) s += X[(size_t)r * ld + c];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < ncols) part[(size_t)blockIdx.y * ncols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ part, int ncols, float* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  float s = 0.f;
  for (int k = 0; k < COLSUM_CHUNKS; k++) s += part[(size_t)k * ncols + c];
  out[c] += s;
}

// Ordered reduction of split-K partials that were left in place by a series of GEMMs (the weight / bias gradients of one
// minibatch): one launch adds every segment's slices, in slice order, to its destination.
#define MAX_SEGS 16
struct Seg { const float* part; float* dst; int nslices, count, N, ldc; };   // partial z at part + z*count; element i -> dst[(i/N)*ldc + i%N]
struct SegList { Seg s[MAX_SEGS]; int first[MAX_SEGS + 1]; int n; float scale; };   // scale: applied to every total (undoes the loss scaling)
__global__ void __launch_bounds__(256) reduce_segments_kernel(SegList L) {
  // block = 64 consecutive elements of one segment x 4 slice ranges; the four partial sums are combined in a fixed order
  __shared__ float red[4][64];
  LHW_LDS_POISON(red);
  const int e0 = blockIdx.x * 64;
  int k = 0;
  while (e0 >= L.first[k + 1]) k++;
  const Seg sg = L.s[k];
  const int e = e0 - L.first[k] + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int per = (sg.nslices + 3) >> 2, z0 = q * per, z1 = min(sg.nslices, z0 + per);
  float s = 0.f;
  if (e < sg.count) {
#pragma unroll 8
    for (int z = z0; z < z1; z++) s += sg.part[(size_t)z * sg.count + e];
  }
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && e < sg.count) {
    const int t = threadIdx.x;
    const float tot = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    const int row = e / sg.N, col = e - row * sg.N;
    sg.dst[(size_t)row * sg.ldc + col] += tot * L.scale;
  }
}
static void seg_add(SegList& L, const float* part, float* dst, int nslices, int M, int N, int ldc) {
  Seg& s = L.s[L.n];
  s.part = part; s.dst = dst; s.nslices = nslices; s.count = M * N; s.N = N; s.ldc = ldc;
  if (L.n == 0) L.first[0] = 0;
  L.first[L.n + 1] = L.first[L.n] + (M * N + 63) / 64 * 64;   // (a block never straddles two segments)
  L.n++;
}
static void launch_reduce_segments(const SegList& L, hipStream_t s) {
  if (L.n == 0) return;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(L.first[L.n] / 64), dim3(256), 0, s, L);
}

// The two SKINNY weight gradients of one network in one K-streaming launch (round 5):
//   dW1 [H][Dp] = dh1^T x,  db1 = colsum(dh1),   dW3 [O][H] = dy^T h2,  db3 = colsum(dy)        (H = 256, Dp <= 64, O <= 32)
// Both contract over the minibatch rows and are bound by streaming one [R][256] activation array each (dh1, h2); as two 64 x 64-tile
// split-K GEMMs with 128-row slices they ran at 1.3 TB/s (23 + 26 us per 32768 rows: a thousand short blocks, prologue and epilogue
// per 128 rows).  Here a block of 8 waves takes `kc` consecutive rows and keeps BOTH products' whole outputs in registers -- wave w
// owns rows 32 w .. 32 w + 31 of dW1 (two 32 x 32 MFMA tiles across the padded Dp) and columns 32 w .. + 31 of dW3 (one tile) --
// while the rows stream through double-buffered LDS tiles in their row-major order ([r][256]: lane = column, the MFMA's operand
// layout for a product that contracts over r).  Partials per block: 256 x Dp + O x 256 + 256 + O floats, reduced in slice order by
// reduce_segments like every other weight gradient (same seed -> same bits).
struct WgradSkinnyArgs {
  const float *dh1, *x, *dy, *h2;   // [R][256], [R][ldx], [R][Op], [R][256]
  int ldx, Dp, O, Op, R, kc;
  float *pw1, *pb1, *pw3, *pb3;     // slice z: pw1 + z * 256 * Dp, pb1 + z * 256, pw3 + z * O * 256, pb3 + z * O
};
#define WS_KS 16
#define WS_H 256
__global__ void __launch_bounds__(512) wgrad_skinny_kernel(WgradSkinnyArgs g) {
  __shared__ float Dh[2][WS_KS][WS_H + 4];
  LHW_LDS_POISON(Dh);
  __shared__ float Hs[2][WS_KS][WS_H + 4];
  LHW_LDS_POISON(Hs);
  __shared__ float Xs[2][WS_KS][64 + 4];
  LHW_LDS_POISON(Xs);
  __shared__ float Ys[2][WS_KS][32 + 4];
  LHW_LDS_POISON(Ys);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int bz = (int)blockIdx.x, kbeg = bz * g.kc, kend = min(g.R, kbeg + g.kc);
  f32x16 a1[2], a3;
  for (int r = 0; r < 16; r++) { a1[0][r] = 0.f; a1[1][r] = 0.f; a3[r] = 0.f; }
  float4 rd[2], rh[2], rx, ry;
  auto load = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int idx = tid + 512 * q, row = idx >> 6, c4 = idx & 63, k = k0 + row;
      rd[q] = rh[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend) {
        rd[q] = *reinterpret_cast<const float4*>(g.dh1 + (size_t)k * WS_H + 4 * c4);
        rh[q] = *reinterpret_cast<const float4*>(g.h2 + (size_t)k * WS_H + 4 * c4);
      }
    }
    rx = ry = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) {
      const int row = tid >> 4, c4 = tid & 15, k = k0 + row;
      if (k < kend && 4 * c4 < g.Dp) rx = *reinterpret_cast<const float4*>(g.x + (size_t)k * g.ldx + 4 * c4);   // (Dp is a multiple of 4)
    } else if (tid < 384) {
      const int row = (tid - 256) >> 3, c4 = (tid - 256) & 7, k = k0 + row;
      if (k < kend && 4 * c4 < g.Op) {
        ry = *reinterpret_cast<const float4*>(g.dy + (size_t)k * g.Op + 4 * c4);
        if (4 * c4 + 1 >= g.O) ry.y = 0.f;
        if (4 * c4 + 2 >= g.O) ry.z = 0.f;
        if (4 * c4 + 3 >= g.O) ry.w = 0.f;
        if (4 * c4 >= g.O) ry.x = 0.f;
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int idx = tid + 512 * q, row = idx >> 6, c4 = idx & 63;
      *reinterpret_cast<float4*>(&Dh[buf][row][4 * c4]) = rd[q];
      *reinterpret_cast<float4*>(&Hs[buf][row][4 * c4]) = rh[q];
    }
    if (tid < 256) *reinterpret_cast<float4*>(&Xs[buf][tid >> 4][4 * (tid & 15)]) = rx;
    else if (tid < 384) *reinterpret_cast<float4*>(&Ys[buf][(tid - 256) >> 3][4 * ((tid - 256) & 7)]) = ry;
  };
  float cs1 = 0.f, cs3 = 0.f;   // thread (m = tid % 256, k group = tid / 256): column sum of dh1; thread t < 32: column sum of dy
  int cur = 0;
  if (kbeg < kend) { load(kbeg); store(0); }
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += WS_KS) {
    const bool more = k0 + WS_KS < kend;
    if (more) load(k0 + WS_KS);
#pragma unroll
    for (int kk = 0; kk < WS_KS / 2; kk++) {
      const int k = kk * 2 + kh;
      const float ad = Dh[cur][k][32 * wave + l31], bx0 = Xs[cur][k][l31], bx1 = Xs[cur][k][32 + l31];
      const float ay = Ys[cur][k][l31], bh = Hs[cur][k][32 * wave + l31];
      a1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad, bx0, a1[0], 0, 0, 0);
      a1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad, bx1, a1[1], 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bh, a3, 0, 0, 0);
    }
    {
      const int m = tid & 255, kg = tid >> 8;
#pragma unroll
      for (int q = 0; q < WS_KS / 2; q++) cs1 += Dh[cur][kg * (WS_KS / 2) + q][m];
      if (tid < 32) {
#pragma unroll
        for (int q = 0; q < WS_KS; q++) cs3 += Ys[cur][q][tid];
      }
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // epilogue; C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* pw1 = g.pw1 + (size_t)bz * WS_H * g.Dp;
  float* pw3 = g.pw3 + (size_t)bz * g.O * WS_H;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = 32 * j + l31;
      if (col < g.Dp) pw1[(size_t)(32 * wave + rr) * g.Dp + col] = a1[j][r];
    }
    if (rr < g.O) pw3[(size_t)rr * WS_H + 32 * wave + l31] = a3[r];
  }
  float* red = &Dh[0][0][0];   // (the tiles are idle: the loop ended with a barrier)
  red[tid] = cs1;
  __syncthreads();
  if (tid < 256) g.pb1[(size_t)bz * WS_H + tid] = red[tid] + red[256 + tid];
  if (tid < g.O) g.pb3[(size_t)bz * g.O + tid] = cs3;
}
static bool fused_skinny_on() {   // LHW_WGRAD_FUSED=0: the two split-K GEMMs instead (A/B measurements)
  static const bool on = !(getenv("LHW_WGRAD_FUSED") && atoi(getenv("LHW_WGRAD_FUSED")) == 0);
  return on;
}
static inline int wgrad_skinny_chunk(int R) {   // rows per block: about 256 blocks per launch, at least the slice length the partial regions are sized for
  const int kc = ((R + 255) / 256 + WS_KS - 1) / WS_KS * WS_KS;
  return kc < 128 ? 128 : kc;
}
static bool wgrad_skinny_supported(int H, int Dp, int O, int Op) { return H == WS_H && Dp > 0 && Dp <= 64 && (Dp & 3) == 0 && O > 0 && O <= 32 && Op >= O && Op <= 32 && (Op & 3) == 0; }

// The WIDE weight gradient dW2 [256][256] = dh2^T h1 (and db2 = colsum(dh2)) without LDS and without barriers (round 6).  Both operands
// are stored [row][256] and the product contracts over the rows, so row k of either IS the MFMA's operand layout (lane l: unit l % 32
// of row k + l / 32): a wave's load instruction fetches two 128-byte segments straight from HBM / L2, as the strip kernels load their
// weights.  Block tile 128 x 128 (four per k slice), wave tile 64 x 64 = 2 x 2 MFMA tiles: one operand load per MFMA (the LDS-staged
// 64 x 64-tile GEMM: two LDS reads per MFMA and a barrier every 16 rows, 65 us per 32768 rows = 42 % of the f32 MFMA peak).  The rows
// of chunk s + 1 are in flight while chunk s is multiplied.  XCD-aware block order: the four tiles of a k slice run on one L2.
// The bias gradient comes from the same operand registers: waves with tn == wn == 0 keep a running sum of their dh2 entries (per
// lane ascending k of one parity, the two parities added at the end: a fixed order).
struct WgradWideArgs {
  const float *A, *B;      // [K][256] each: dh2, h1
  int K, k_chunk, slices;
  float *part, *colsum;    // slice z: part + z * 256 * 256 (row-major [m][n]), colsum + z * 256
};
#define WW_H 256
// KS: rows per register buffer.  Two buffers: the loads of chunk s + 1 are issued before chunk s is multiplied, i.e. KS / 2 x 4 MFMAs
// (KS x 128 cycles when the wave has its SIMD's MFMA pipe to itself) ahead of their use.  In the update the operands come from HBM (the
// forward pass wrote h1 a few hundred MB of traffic earlier): KS = 32 (1.7 us ahead); with KS = 16 the kernel was faster than the
// LDS-staged GEMM alone on cache-warm operands and slower inside the update (profiles/r06_wgrad_wide.txt).
template <int KS>
struct WwOp { float a[KS / 2][2], b[KS / 2][2]; };
template <int KS>
__device__ __forceinline__ void ww_load(WwOp<KS>& w, const float* __restrict__ A, const float* __restrict__ B, const int k0, const int kend, const int am, const int bn) {
  const int kh = (threadIdx.x & 63) >> 5;
#pragma unroll
  for (int kk = 0; kk < KS / 2; kk++) {
    const int k = k0 + 2 * kk + kh, kc = min(k, kend - 1);
    const bool live = k < kend;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float av = A[(size_t)kc * WW_H + am + 32 * i];
      w.a[kk][i] = live ? av : 0.f;                       // (rows beyond the slice: a zero dh2 entry, times a valid row of h1)
      w.b[kk][i] = B[(size_t)kc * WW_H + bn + 32 * i];
    }
  }
}
template <int KS>
__global__ void __launch_bounds__(256, 2) wgrad_wide_kernel(WgradWideArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int per = (int)gridDim.x >> 3, v = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (v >= 4 * g.slices) return;
  const int t = v & 3, tm = t >> 1, tn = t & 1, bz = v >> 2, wm = wave & 1, wn = wave >> 1;
  const int m0 = tm * 128 + wm * 64, n0 = tn * 128 + wn * 64;
  const int kbeg = bz * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  float cs[2] = {0.f, 0.f};
  WwOp<KS> w0, w1;
  auto mul = [&](const WwOp<KS>& w) {
#pragma unroll
    for (int kk = 0; kk < KS / 2; kk++) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a[kk][i], w.b[kk][j], acc[i][j], 0, 0, 0);
      cs[0] += w.a[kk][0]; cs[1] += w.a[kk][1];
    }
  };
  if (kbeg < kend) {
    ww_load<KS>(w0, g.A, g.B, kbeg, kend, m0 + l31, n0 + l31);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * KS) {
      ww_load<KS>(w1, g.A, g.B, k0 + KS, kend, m0 + l31, n0 + l31);      // (unconditional: clamped past the end)
      __builtin_amdgcn_sched_barrier(0);
      mul(w0);
      __builtin_amdgcn_sched_barrier(0);
      ww_load<KS>(w0, g.A, g.B, k0 + 2 * KS, kend, m0 + l31, n0 + l31);
      __builtin_amdgcn_sched_barrier(0);
      if (k0 + KS < kend) mul(w1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue; C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* part = g.part + (size_t)bz * WW_H * WW_H;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) part[(size_t)(m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh) * WW_H + n0 + 32 * j + l31] = acc[i][j][r];
  if (g.colsum && tn == 0 && wn == 0) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float o = __shfl_down(cs[i], 32);
      if (kh == 0) g.colsum[(size_t)bz * WW_H + m0 + 32 * i + l31] = cs[i] + o;
    }
  }
}
static bool wgrad_wide_on() {   // LHW_WGRAD_WIDE=0: the LDS-staged split-K GEMM instead (A/B measurements)
  static const bool on = !(getenv("LHW_WGRAD_WIDE") && atoi(getenv("LHW_WGRAD_WIDE")) == 0);
  return on;
}
static void launch_wgrad_wide(const float* A, const float* B, int K, int k_chunk, float* part, float* colsum, hipStream_t s) {
  WgradWideArgs a{A, B, K, k_chunk, (K + k_chunk - 1) / k_chunk, part, colsum};
  static const int ks = getenv("LHW_WGRAD_WIDE_KS") ? atoi(getenv("LHW_WGRAD_WIDE_KS")) : 32;   // (tuning aid)
  const dim3 grid(8 * ((4 * (size_t)a.slices + 7) / 8));
  if (ks == 16) hipLaunchKernelGGL(wgrad_wide_kernel<16>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(wgrad_wide_kernel<32>, grid, dim3(256), 0, s, a);
}

// defer != 0: split-K partials (and the fused column sums) stay in g.part / g.colsum for a later reduce_segments launch
template <bool A_KC, bool B_KC>
static void launch_gemm(const GemmArgs& g, hipStream_t s, int defer = 0, int wt = 0, int half = 0) {
  GemmArgs a = g;
  if (a.k_chunk <= 0) a.k_chunk = a.K;
  const int kstep = half ? HBK : BK;
  a.k_chunk = ((a.k_chunk + kstep - 1) / kstep) * kstep;
  const int nz = (a.K + a.k_chunk - 1) / a.k_chunk;
  static const int env_wt = getenv("LHW_GEMM_WT") ? atoi(getenv("LHW_GEMM_WT")) : 0;   // tuning aid: 1 / 2 forces the tile size
  const int force_wt = wt ? wt : env_wt;
  // 64 x 64 block tiles by default: on every shape of the update they beat the 128 x 128 variant (profiles/r02_ppo_gemm_shapes.txt:
  // at K = 256 a block's whole K loop is 16 steps, so more, smaller blocks hide the load / store phases better than fewer
  // LDS reads per MFMA help); wt = 2 (LHW_GEMM_WT=2) keeps the large tile selectable for other shapes
  const bool big = force_wt == 2;
  const int tile = (big && !half) ? 128 : 64;
  a.tiles_m = (a.M + tile - 1) / tile; a.tiles_n = (a.N + tile - 1) / tile; a.slices = nz;
  const dim3 grid(8 * (((size_t)a.tiles_m * a.tiles_n * nz + 7) / 8));
  if (half) hipLaunchKernelGGL((gemm_h_kernel<A_KC, B_KC>), grid, dim3(256), 0, s, a);
  else if (big) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 1>), grid, dim3(256), 0, s, a);
  if (a.part && !defer) {  // ordered reduction of the split-K slices into the (accumulating) destination
    const int n = a.M * a.N;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a.part, nz, a.M, a.N, a.C, a.ldc);
  }
}

// ------------------------------------------------------------------------------------------- MLP plumbing
// Internal parameter layout of one 3-layer MLP (in D -> H -> H -> O), all float32:
//   W1 [H][Dp]  b1 [H]  W2 [H][H]  b2 [H]  W3 [Op][H]  b3 [Op]      Dp = pad4(D), Op = pad4(O)
// (torch Linear layout [out][in]; padded rows/columns are zero and stay zero under Adam).
struct MlpLayout {
  int D, Dp, H, O, Op;
  size_t w1, b1, w2, b2, w3, b3, total;
};
static inline int pad4(int x) { return (x + 3) & ~3; }
static MlpLayout mlp_layout(int D, int H, int O) {
  MlpLayout L;
  L.D = D; L.Dp = pad4(D); L.H = H; L.O = O; L.Op = pad4(O);
  size_t o = 0;
  L.w1 = o; o += (size_t)H * L.Dp;
  L.b1 = o; o += H;
  L.w2 = o; o += (size_t)H * H;
  L.b2 = o; o += H;
  L.w3 = o; o += (size_t)L.Op * H;
  L.b3 = o; o += L.Op;
  L.total = o;
  return L;
}

// fp16 copies of one network's minibatch activations (the --fp16 update keeps them in HBM as fp16: gemm_h_kernel)
struct HalfBufs {
  const _Float16* x; int ldx;          // gathered inputs [rows][ldx], ldx = Dp rounded up to 8 (16-byte rows), pad columns zero
  _Float16 *h1, *h2, *dh2, *dh1;       // [rows][H]
};
struct LhwPpo {
  int device, D, A, H, learn_std, max_rows;  // max_rows: capacity of the minibatch workspace (rows per net)
  _Float16 *xb_h = nullptr, *h1a_h = nullptr, *h2a_h = nullptr, *dh2a_h = nullptr, *dh1a_h = nullptr;   // --fp16 update: fp16 storage (actor: 2R rows)
  _Float16 *h1c_h = nullptr, *h2c_h = nullptr, *dh2c_h = nullptr, *dh1c_h = nullptr;                    // critic: R rows
  int ldxh = 0;
  float clip, ent_coeff, mirror_coeff, grad_clip, lr, adam_eps, beta1, beta2;
  int use_mirror;
  int infer_half = 0;     // rollout inference with fp16 operands (lhw_ppo_set_inference_dtype)
  int update_half = 0;    // every GEMM of the update with fp16 operands (lhw_ppo_set_update_dtype)
  MlpLayout la, lc;       // actor, critic
  size_t off_actor, off_std, off_critic, n_params;  // flat theta: [actor | stds(A, padded to 4) | critic]
  // mirror tables (device): obs_src[Dp], obs_sign[Dp], act_src[A], act_sign[A]
  int *d_obs_src = nullptr, *d_act_src = nullptr;
  float *d_obs_sign = nullptr, *d_act_sign = nullptr;
  // workspace
  float *xb = nullptr;   // [2R][Dp] gathered minibatch inputs (normal rows, then mirrored rows)
  float *h1a = nullptr, *h2a = nullptr, *ya = nullptr;      // actor activations [2R][H], [2R][H], [2R][Op]
  float *h1c = nullptr, *h2c = nullptr, *yc = nullptr;      // critic [R][H] [R][H] [R][4]
  float *dya = nullptr, *dh2a = nullptr, *dh1a = nullptr;   // actor grads wrt activations
  float *dyc = nullptr, *dh2c = nullptr, *dh1c = nullptr;
  float *mb_act = nullptr, *mb_logp = nullptr, *mb_adv = nullptr, *mb_ret = nullptr;
  float *stats = nullptr;  // [16] loss scalars; [8],[9] grad norm^2 actor/critic
  float *part = nullptr;       // split-K partial tiles [max slices][H*H]
  float *dstd = nullptr;       // per-row d loss / d std [R][Op]
  unsigned *bits_a = nullptr, *bits_c = nullptr;   // ReLU masks of h1 / h2 as bits, forward strip -> backward strip (2 layers x words(2R) / words(R)); NULL: max_rows % 64 != 0
  float *wt_a = nullptr, *wt_c = nullptr;   // [in][out] weight copies for the strip kernels (hidden width 256 only): the update's
  float *wt_inf = nullptr;                  // ... and WT_SLOTS pairs (actor, critic) for rollout inference, one per eighth of the
                                            // forward workspace, so that concurrent calls (disjoint row ranges, different streams) do not share one
  float *wt_roll = nullptr;                 // ... and the pair made once per rollout by lhw_ppo_begin_rollout (read-only until end_rollout / apply)
  const float* roll_theta = nullptr;        // theta the wt_roll copies were made from (NULL: no rollout bracket open)
  float *stats_part = nullptr; // per-block loss partials [blocks][NSTAT]
  const float* imit_target = nullptr;          // imitation term of the NEXT lhw_ppo_grad call (lhw_ppo_set_imitation)
  const unsigned char* imit_mask = nullptr;
  float imit_coeff = 0.f, imit_inv_count = 0.f;
  float *norm_part = nullptr;  // [2][SUMSQ_BLOCKS]
  float *bwd_part = nullptr;   // split-K partials of the weight / bias gradients: actor (two passes), then critic
  int max_slices = 0;
  // the critic's forward / backward chain runs on its own stream beside the actor's (they share only the gathered inputs and
  // the loss kernel): the load / multiply / store phases of one network's GEMMs overlap the other's
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int two_streams = 1;
  // lhw_ppo_step: one optimiser step (lhw_ppo_grad + lhw_ppo_apply) captured once as a hipGraph and replayed; the two things that change
  // from step to step -- the minibatch's index pointer (gather_kernel) and Adam's bias corrections (adam2_kernel) -- are patched into
  // the executable graph's kernel nodes before each launch
  hipGraph_t step_graph = nullptr;
  hipGraphExec_t step_exec = nullptr;
  hipGraphNode_t node_gather = nullptr, node_adam = nullptr;
  const void* step_key[14] = {nullptr};
  int step_key_b = 0, step_key_half = 0;
};

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return lhw_fail(LHW_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); \
  } while (0)

// y = mlp(x) for R rows; keeps h1/h2 for the backward pass.  half != 0: fp16 operands (rollout inference only)
// LHW_MLP_STRIP (tuning aid): 0 = per-layer GEMMs everywhere, 1 = LDS-resident strip kernels for the update's forward and
// activation-gradient passes, 2 = for the rollout inference as well (default: the rollout and the update then evaluate the
// networks with the same kernel, bit for bit)
#define WT_SLOTS 8
static int strip_mode() {
  static const int m = getenv("LHW_MLP_STRIP") ? atoi(getenv("LHW_MLP_STRIP")) : 2;
  return m;
}
static void mlp_forward(const MlpLayout& L, const float* theta, const float* x, int ldx, int R, float* h1, float* h2,
                        float* y, hipStream_t s, int half = 0, float* strip_wt = nullptr, bool wt_ready = false, bool keep_hidden = true,
                        const HalfBufs* hb = nullptr, unsigned* bits1 = nullptr, unsigned* bits2 = nullptr) {
  if (hb && half) {   // --fp16 update: x / h1 / h2 live in fp16 (hb), the weights are rounded while staged, y stays float32 for the loss
    GemmArgs g{};
    g.A = reinterpret_cast<const float*>(hb->x); g.lda = hb->ldx; g.a_half = 1; g.B = theta + L.w1; g.ldb = L.Dp;
    g.C = reinterpret_cast<float*>(hb->h1); g.ldc = L.H; g.c_half = 1; g.M = R; g.N = L.H; g.K = L.Dp; g.bias = theta + L.b1; g.relu = 1;
    launch_gemm<true, true>(g, s, 0, 0, 1);
    g = GemmArgs{};
    g.A = reinterpret_cast<const float*>(hb->h1); g.lda = L.H; g.a_half = 1; g.B = theta + L.w2; g.ldb = L.H;
    g.C = reinterpret_cast<float*>(hb->h2); g.ldc = L.H; g.c_half = 1; g.M = R; g.N = L.H; g.K = L.H; g.bias = theta + L.b2; g.relu = 1;
    launch_gemm<true, true>(g, s, 0, 0, 1);
    g = GemmArgs{};
    g.A = reinterpret_cast<const float*>(hb->h2); g.lda = L.H; g.a_half = 1; g.B = theta + L.w3; g.ldb = L.H; g.C = y; g.ldc = L.Op;
    g.M = R; g.N = L.O; g.K = L.H; g.bias = theta + L.b3;
    launch_gemm<true, true>(g, s, 0, 0, 1);
    return;
  }
  if (strip_wt && !half && mlp_strip_supported(L.H, L.Dp, L.O, L.Op)) {   // one launch, h1 / h2 stay in LDS between the layers
    if (!wt_ready) mlp_strip_prepare(theta + L.w1, theta + L.w2, theta + L.w3, L.Dp, L.O, L.Op, strip_wt, s);   // [in][out] copies of the weights
    MlpStripFwd a{strip_wt, theta + L.b1, strip_wt + (size_t)L.Dp * L.H, theta + L.b2, strip_wt + (size_t)L.Dp * L.H + (size_t)L.H * L.H,
                  theta + L.b3, x, ldx, L.Dp, L.O, L.Op, R, keep_hidden ? h1 : nullptr, keep_hidden ? h2 : nullptr, y};   // (inference: h1 / h2 never leave LDS)
    if (keep_hidden) { a.bits1 = bits1; a.bits2 = bits2; }
    mlp_strip_forward(a, s);
    return;
  }
  GemmArgs g{};
  g.A = x; g.lda = ldx; g.B = theta + L.w1; g.ldb = L.Dp; g.C = h1; g.ldc = L.H; g.M = R; g.N = L.H; g.K = L.Dp;
  g.bias = theta + L.b1; g.relu = 1;
  launch_gemm<true, true>(g, s, 0, 0, half);
  g = GemmArgs{};
  g.A = h1; g.lda = L.H; g.B = theta + L.w2; g.ldb = L.H; g.C = h2; g.ldc = L.H; g.M = R; g.N = L.H; g.K = L.H;
  g.bias = theta + L.b2; g.relu = 1;
  launch_gemm<true, true>(g, s, 0, 0, half);
  g = GemmArgs{};
  g.A = h2; g.lda = L.H; g.B = theta + L.w3; g.ldb = L.H; g.C = y; g.ldc = L.Op; g.M = R; g.N = L.O; g.K = L.H;
  g.bias = theta + L.b3;
  launch_gemm<true, true>(g, s, 0, 0, half);
}

static void colsum_det(const float* X, int rows, int ld, int ncols, float* out, float* scratch /* [COLSUM_CHUNKS][ncols] */, hipStream_t s) {
  hipLaunchKernelGGL(colsum_det_kernel, dim3((ncols + 63) / 64, COLSUM_CHUNKS), dim3(256), 0, s, X, rows, ld, ncols, scratch);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((ncols + 255) / 256), dim3(256), 0, s, scratch, ncols, out);
}
// Split-K partial regions of one network's parameter gradients ([slices][count] each), reduced by one reduce_segments launch.
// The K dimension of a weight-gradient GEMM is the minibatch: it is cut into slices so that every GEMM puts ~1000 blocks on the
// chip -- 512-row slices for the 256 x 256 layer (16 output tiles), 128-row slices for the skinny first / last layers (4 tiles),
// whose blocks would otherwise sit through 32 dependent load -> multiply steps with nothing else resident to hide them.
#define KC_WIDE 512
#define KC_SKINNY 128
struct BwdParts { float *w3, *w2, *w1, *b3, *b2, *b1; };
struct BwdSlices { int w3 = 0, w2 = 0, w1 = 0; };
static inline int nsl(int rows, int kc) { return (rows + kc - 1) / kc; }
static size_t bwd_parts_floats(const MlpLayout& L, size_t rows, int passes) {
  const size_t ss = (size_t)passes * (nsl((int)rows, KC_SKINNY) + 1), sw = (size_t)passes * (nsl((int)rows, KC_WIDE) + 1);
  return ss * ((size_t)L.Op * L.H + L.Op) + sw * ((size_t)L.H * L.H + L.H) + ss * ((size_t)L.H * L.Dp + L.H);
}
static BwdParts bwd_parts_carve(const MlpLayout& L, size_t rows, int passes, float* base) {
  const size_t ss = (size_t)passes * (nsl((int)rows, KC_SKINNY) + 1), sw = (size_t)passes * (nsl((int)rows, KC_WIDE) + 1);
  BwdParts P;
  P.w3 = base; base += ss * L.Op * L.H;
  P.b3 = base; base += ss * L.Op;
  P.w2 = base; base += sw * L.H * L.H;
  P.b2 = base; base += sw * L.H;
  P.w1 = base; base += ss * L.H * L.Dp;
  P.b1 = base;
  return P;
}
// Back-propagation through one MLP for R rows given dy [R][Op].  The weight-gradient GEMMs (K = R) leave their partial products
// -- and, from the same operand tiles, the bias gradients' partial column sums -- in P behind the z slices an earlier pass
// wrote; mlp_backward_segments then lists them for the ordered reduction into grad.  Every reduction runs in a fixed order
// (same seed -> bitwise identical weights, the property the reference's tests/test_determinism.py checks).
static void mlp_backward(const MlpLayout& L, const float* theta, const float* x, int ldx, int R, const float* h1, const float* h2,
                         const float* dy, float* dh2, float* dh1, const BwdParts& P, BwdSlices& z, hipStream_t s, int half = 0,
                         const HalfBufs* hb = nullptr, const unsigned* bits1 = nullptr, const unsigned* bits2 = nullptr) {
  GemmArgs g{};
  if (hb && half) {   // --fp16 update with fp16 storage: the same five GEMMs on the fp16 copies (dy and the weights are float32)
    auto H16 = [](const _Float16* q) { return reinterpret_cast<const float*>(q); };
    // dW3 [O][H] = dy^T h2 ; db3 = colsum(dy)
    g.A = dy; g.lda = L.Op; g.B = H16(hb->h2); g.ldb = L.H; g.b_half = 1; g.M = L.O; g.N = L.H; g.K = R;
    g.part = P.w3 + (size_t)z.w3 * L.O * L.H; g.colsum = P.b3 + (size_t)z.w3 * L.O; g.k_chunk = KC_SKINNY;
    launch_gemm<false, false>(g, s, 1, 0, 1);
    // dh2 = (dy W3) * (h2 > 0)
    g = GemmArgs{};
    g.A = dy; g.lda = L.Op; g.B = theta + L.w3; g.ldb = L.H; g.C = reinterpret_cast<float*>(hb->dh2); g.ldc = L.H; g.c_half = 1;
    g.M = R; g.N = L.H; g.K = L.O; g.mask = H16(hb->h2); g.ldmask = L.H; g.mask_half = 1;
    launch_gemm<true, false>(g, s, 0, 0, 1);
    // dW2 = dh2^T h1 ; db2 = colsum(dh2)
    g = GemmArgs{};
    g.A = H16(hb->dh2); g.lda = L.H; g.a_half = 1; g.B = H16(hb->h1); g.ldb = L.H; g.b_half = 1; g.M = L.H; g.N = L.H; g.K = R;
    g.part = P.w2 + (size_t)z.w2 * L.H * L.H; g.colsum = P.b2 + (size_t)z.w2 * L.H; g.k_chunk = KC_WIDE;
    launch_gemm<false, false>(g, s, 1, 0, 1);
    // dh1 = (dh2 W2) * (h1 > 0)
    g = GemmArgs{};
    g.A = H16(hb->dh2); g.lda = L.H; g.a_half = 1; g.B = theta + L.w2; g.ldb = L.H; g.C = reinterpret_cast<float*>(hb->dh1); g.ldc = L.H; g.c_half = 1;
    g.M = R; g.N = L.H; g.K = L.H; g.mask = H16(hb->h1); g.ldmask = L.H; g.mask_half = 1;
    launch_gemm<true, false>(g, s, 0, 0, 1);
    // dW1 [H][Dp] = dh1^T x ; db1 = colsum(dh1)
    g = GemmArgs{};
    g.A = H16(hb->dh1); g.lda = L.H; g.a_half = 1; g.B = H16(hb->x); g.ldb = hb->ldx; g.b_half = 1; g.M = L.H; g.N = L.Dp; g.K = R;
    g.part = P.w1 + (size_t)z.w1 * L.H * L.Dp; g.colsum = P.b1 + (size_t)z.w1 * L.H; g.k_chunk = KC_SKINNY;
    launch_gemm<false, false>(g, s, 1, 0, 1);
    z.w3 += nsl(R, KC_SKINNY); z.w2 += nsl(R, KC_WIDE); z.w1 += nsl(R, KC_SKINNY);
    return;
  }
  const bool strip = strip_mode() >= 1 && !half && mlp_strip_supported(L.H, L.Dp, L.O, L.Op);
  if (strip) {   // dh2 = (dy W3) * (h2 > 0) and dh1 = (dh2 W2) * (h1 > 0) in one launch, the dh2 slab staying in LDS
    MlpStripBwd a{theta + L.w2, theta + L.w3, dy, h1, h2, L.O, L.Op, R, dh2, dh1};
    a.bits1 = bits1; a.bits2 = bits2;
    mlp_strip_backward(a, s);
  }
  // The skinny weight gradients dW1 / db1 / dW3 / db3: one K-streaming launch behind the activation gradients (wgrad_skinny_kernel; its
  // slices are at least KC_SKINNY rows, so they fit the partial regions), or two split-K GEMMs (other widths, fp16 operands)
  const bool fused_skinny = strip && z.w1 == z.w3 && wgrad_skinny_supported(L.H, L.Dp, L.O, L.Op) && ldx >= L.Dp && !(ldx & 3) && fused_skinny_on();
  // dW3 [O][H] = dy^T h2 ; db3 = colsum(dy)
  g.A = dy; g.lda = L.Op; g.B = h2; g.ldb = L.H; g.M = L.O; g.N = L.H; g.K = R;
  g.part = P.w3 + (size_t)z.w3 * L.O * L.H; g.colsum = P.b3 + (size_t)z.w3 * L.O; g.k_chunk = KC_SKINNY;
  if (!fused_skinny) launch_gemm<false, false>(g, s, 1, 0, half);
  // dh2 = (dy W3) * (h2 > 0)
  if (!strip) {
    g = GemmArgs{};
    g.A = dy; g.lda = L.Op; g.B = theta + L.w3; g.ldb = L.H; g.C = dh2; g.ldc = L.H; g.M = R; g.N = L.H; g.K = L.O;
    g.mask = h2; g.ldmask = L.H;
    launch_gemm<true, false>(g, s, 0, 0, half);
  }
  // dW2 = dh2^T h1 ; db2 = colsum(dh2)
  g = GemmArgs{};
  g.A = dh2; g.lda = L.H; g.B = h1; g.ldb = L.H; g.M = L.H; g.N = L.H; g.K = R;
  g.part = P.w2 + (size_t)z.w2 * L.H * L.H; g.colsum = P.b2 + (size_t)z.w2 * L.H; g.k_chunk = KC_WIDE;
  if (!half && L.H == WW_H && wgrad_wide_on()) launch_wgrad_wide(dh2, h1, R, KC_WIDE, g.part, g.colsum, s);
  else launch_gemm<false, false>(g, s, 1, 0, half);
  // dh1 = (dh2 W2) * (h1 > 0)
  if (!strip) {
    g = GemmArgs{};
    g.A = dh2; g.lda = L.H; g.B = theta + L.w2; g.ldb = L.H; g.C = dh1; g.ldc = L.H; g.M = R; g.N = L.H; g.K = L.H;
    g.mask = h1; g.ldmask = L.H;
    launch_gemm<true, false>(g, s, 0, 0, half);
  }
  // dW1 [H][Dp] = dh1^T x ; db1 = colsum(dh1)
  if (fused_skinny) {
    const int kc = wgrad_skinny_chunk(R), ns = nsl(R, kc);
    WgradSkinnyArgs a{dh1, x, dy, h2, ldx, L.Dp, L.O, L.Op, R, kc, P.w1 + (size_t)z.w1 * L.H * L.Dp, P.b1 + (size_t)z.w1 * L.H,
                      P.w3 + (size_t)z.w3 * L.O * L.H, P.b3 + (size_t)z.w3 * L.O};
    hipLaunchKernelGGL(wgrad_skinny_kernel, dim3(ns), dim3(512), 0, s, a);
    z.w3 += ns; z.w2 += nsl(R, KC_WIDE); z.w1 += ns;
    return;
  }
  g = GemmArgs{};
  g.A = dh1; g.lda = L.H; g.B = x; g.ldb = ldx; g.M = L.H; g.N = L.Dp; g.K = R;
  g.part = P.w1 + (size_t)z.w1 * L.H * L.Dp; g.colsum = P.b1 + (size_t)z.w1 * L.H; g.k_chunk = KC_SKINNY;
  launch_gemm<false, false>(g, s, 1, 0, half);
  z.w3 += nsl(R, KC_SKINNY); z.w2 += nsl(R, KC_WIDE); z.w1 += nsl(R, KC_SKINNY);
}
static void mlp_backward_segments(SegList& S, const MlpLayout& L, float* grad, const BwdParts& P, const BwdSlices& z) {
  seg_add(S, P.w3, grad + L.w3, z.w3, L.O, L.H, L.H);
  seg_add(S, P.b3, grad + L.b3, z.w3, L.O, 1, 1);
  seg_add(S, P.w2, grad + L.w2, z.w2, L.H, L.H, L.H);
  seg_add(S, P.b2, grad + L.b2, z.w2, L.H, 1, 1);
  seg_add(S, P.w1, grad + L.w1, z.w1, L.H, L.Dp, L.Dp);
  seg_add(S, P.b1, grad + L.b1, z.w1, L.H, 1, 1);
}

// ------------------------------------------------------------------------------------------- elementwise kernels
// float32 rows [rows][ld] -> fp16 rows [rows][ldh] (ldh >= cols, pad columns zero): the gathered minibatch inputs of the --fp16 update
__global__ void __launch_bounds__(256) rows_to_half_kernel(const float* __restrict__ src, int ld, int cols, size_t rows, _Float16* __restrict__ dst, int ldh) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * (size_t)ldh) return;
  const size_t r = i / ldh;
  const int c = (int)(i - r * ldh);
  dst[i] = c < cols ? (_Float16)src[r * ld + c] : (_Float16)0.f;
}
// (x - mean)/std into a [R][Dp] buffer (pad columns zero); optional mirrored copy
// mirror: out[j] = sign[j] * obs[src[j]]  == obs @ M with the clock sign flip folded in
// (reference rl/envs/wrappers.py:53-85: sin(arcsin(c)+pi) == -c)
__global__ void normalize_kernel(const float* __restrict__ obs, int D, int Dp, size_t R, const float* __restrict__ mean,
                                 const float* __restrict__ stdv, float* __restrict__ xn, float* __restrict__ xm,
                                 const int* __restrict__ src, const float* __restrict__ sign) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * (size_t)Dp) return;
  size_t r = i / Dp;
  int j = (int)(i - r * Dp);
  float v = 0.f, vm = 0.f;
  if (j < D) {
    v = (obs[r * D + j] - mean[j]) / stdv[j];
    if (xm) vm = (sign[j] * obs[r * D + src[j]] - mean[j]) / stdv[j];
  }
  xn[i] = v;
  if (xm) xm[i] = vm;
}

// gather a minibatch: rows idx[0..B) of xn -> xb[0..B), of xm -> xb[R..R+B) (if mirror), plus act/logp/adv/ret
__global__ void gather_kernel(const int* __restrict__ idx, int B, int Rcap, int Dp, int A, const float* __restrict__ xn,
                              const float* __restrict__ xm, const float* __restrict__ act, const float* __restrict__ logp,
                              const float* __restrict__ adv, const float* __restrict__ ret, float* __restrict__ xb,
                              float* __restrict__ mact, float* __restrict__ mlogp, float* __restrict__ madv,
                              float* __restrict__ mret) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Dp) return;
  int m = (int)(i / Dp), j = (int)(i - (size_t)m * Dp);
  size_t s = (size_t)idx[m];
  xb[(size_t)m * Dp + j] = xn[s * Dp + j];
  if (xm) xb[((size_t)Rcap + m) * Dp + j] = xm[s * Dp + j];
  if (j < A) mact[(size_t)m * A + j] = act[s * A + j];
  if (j == 0) { mlogp[m] = logp[s]; madv[m] = adv[s]; mret[m] = ret[s]; }
}

// rollout sampling: act = mu + std * N(0,1) (or mu), logp of the sampled action under (mu, std)
// One (row, action component) per lane, 32 lanes per row (act_dim <= 32): the Box-Muller draw is ~400 instructions of float64
// transcendentals per component, so a thread per row (12 draws in sequence, 8 blocks for a 2048-row group) left this kernel
// latency-bound at 15 us on the rollout's critical path.  The log-density terms are summed by the row's first lane in
// component order, as the fused read-out of the forward strip kernel does (bit-identical log-probabilities).
__global__ void __launch_bounds__(256) sample_kernel(const float* __restrict__ mu, int ldmu, int A, int N, const float* __restrict__ stdv,
                                                     uint64_t seed, uint32_t env_base, uint32_t counter, int deterministic,
                                                     float* __restrict__ act, float* __restrict__ logp) {
  __shared__ float terms[8][32];
  LHW_LDS_POISON(terms);
  const int r = threadIdx.x >> 5, a = threadIdx.x & 31, n = blockIdx.x * 8 + r;
  if (n < N && a < A) {
    float term;
    act[(size_t)n * A + a] = lhw_policy_sample(mu[(size_t)n * ldmu + a], stdv[a], seed, env_base + n, counter, a, deterministic, &term);
    terms[r][a] = term;
  }
  __syncthreads();
  if (n < N && a == 0) {
    float lp = 0.f;
    for (int k = 0; k < A; k++) lp += terms[r][k];
    logp[n] = lp;
  }
}

// PPO losses and their gradients wrt network outputs (reference rl/algos/ppo.py:302-384, FF path, mask = 1).
// No atomics: bias / std gradients are column sums of dya / dyc / dstd taken afterwards in a fixed order, and the loss
// scalars are written as per-block partials [gridDim.x][6]: 0 actor_loss 1 critic_loss 2 mirror_loss 3 approx_kl
// 4 clip_fraction (already divided by B) 5 imitation_loss.
// Imitation term (ppo.py:360-368): imitation_loss = mean over the selected (sample, action dim) entries of
// (mu - expert)^2; the host evaluates the env's projector and the frozen expert and hands over the dense target / mask
// in minibatch order (lhw_ppo_set_imitation); here the term enters the loss scalar and d loss / d mu.
#define NSTAT 6
__global__ void __launch_bounds__(256) ppo_loss_kernel(int B, int Rcap, int A, int Op, const float* __restrict__ ya,
                                                       const float* __restrict__ yc, const float* __restrict__ act,
                                                       const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                       const float* __restrict__ ret, const float* __restrict__ stdv,
                                                       float clip, float mirror_coeff, int use_mirror,
                                                       const int* __restrict__ act_src, const float* __restrict__ act_sign,
                                                       float* __restrict__ dya, float* __restrict__ dyc,
                                                       float* __restrict__ dstd /* [B][Op] or NULL */, float* __restrict__ stats_part,
                                                       const float* __restrict__ imit_target /* [B][A] or NULL */,
                                                       const unsigned char* __restrict__ imit_mask /* [B][A] */, float imit_coeff,
                                                       float imit_inv_count, int seqB,
                                                       float gscale /* power of two applied to dya / dyc (fp16 update: loss scaling) */) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  // row of sample m in the actor output buffer, and of its mirrored twin: FF minibatch: m and Rcap + m; recurrent minibatch
  // (time-major, seqB columns per step, mirrored columns appended per step): t * 2 seqB + b and + seqB
  size_t rn = (size_t)m, rm = (size_t)Rcap + m;
  if (seqB > 0 && use_mirror) { rn = (size_t)(m / seqB) * (2 * (size_t)seqB) + (size_t)(m % seqB); rm = rn + seqB; }
  float s_actor = 0, s_critic = 0, s_mirror = 0, s_kl = 0, s_cf = 0, s_imit = 0;
  const float invB = 1.f / (float)B, invBA = 1.f / ((float)B * (float)A);
  __shared__ float red[NSTAT][4];
  LHW_LDS_POISON(red);
  if (m < B) {
    float lp = 0.f;
    for (int a = 0; a < A; a++) {
      float d = (act[(size_t)m * A + a] - ya[rn * Op + a]) / stdv[a];
      lp += -0.5f * d * d - logf(stdv[a]) - 0.9189385332046727f;
    }
    float logr = lp - old_logp[m];
    float ratio = expf(logr);
    float ad = adv[m];
    float cl = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    float cpi = ratio * ad, cll = cl * ad;
    s_actor = -fminf(cpi, cll);
    float dratio = (cpi <= cll) ? ad : 0.f;  // torch.min backward; ties inside the clip range carry the full gradient
    float dlp = -invB * dratio * ratio;
    s_kl = (ratio - 1.f) - logr;
    s_cf = fabsf(ratio - 1.f) > clip ? 1.f : 0.f;
    float v = yc[(size_t)m * 4];
    float e = ret[m] - v;
    s_critic = e * e;
    dyc[(size_t)m * 4] = -2.f * e * invB * gscale;
    dyc[(size_t)m * 4 + 1] = 0.f; dyc[(size_t)m * 4 + 2] = 0.f; dyc[(size_t)m * 4 + 3] = 0.f;
    if (use_mirror) for (int a = 0; a < Op; a++) dya[rm * Op + a] = 0.f;
    for (int a = 0; a < Op; a++) {
      float g = 0.f, gs = 0.f;
      if (a < A) {
        float mu = ya[rn * Op + a], sd = stdv[a], x = act[(size_t)m * A + a];
        g = dlp * (x - mu) / (sd * sd);
        gs = dlp * ((x - mu) * (x - mu) / (sd * sd * sd) - 1.f / sd);
        if (use_mirror) {
          // mirror_actions[a] = sign[a] * mu_mir[src[a]]  (== mu_mir @ M_a, rl/envs/wrappers.py:49-51)
          float mm = act_sign[a] * ya[rm * Op + act_src[a]];
          float diff = mu - mm;
          s_mirror += diff * diff;
          g += mirror_coeff * 2.f * diff * invBA;
          // gradient wrt the mirrored-pass output it came from (act_src is a permutation: each slot written once)
          dya[rm * Op + act_src[a]] = -mirror_coeff * 2.f * diff * invBA * act_sign[a] * gscale;
        }
        if (imit_target && imit_mask[(size_t)m * A + a]) {
          float diff = mu - imit_target[(size_t)m * A + a];
          s_imit += diff * diff;
          g += imit_coeff * 2.f * diff * imit_inv_count;
        }
      }
      dya[rn * Op + a] = g * gscale;
      if (dstd) dstd[(size_t)m * Op + a] = gs;
    }
  }
  // block reduction of the scalars in a fixed order (xor butterfly inside the wave, then waves 0..3)
  float vals[NSTAT] = {s_actor * invB, s_critic * invB, s_mirror * invBA, s_kl * invB, s_cf * invB, s_imit * imit_inv_count};
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < NSTAT; k++) vals[k] += __shfl_xor(vals[k], o);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int k = 0; k < NSTAT; k++) red[k][wave] = vals[k];
  __syncthreads();
  if (threadIdx.x < NSTAT) stats_part[(size_t)blockIdx.x * NSTAT + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// out[k] += sum_b part[b][n] in block order (single block; n small)
__global__ void reduce_rows_kernel(const float* __restrict__ part, int nrows, int n, float* __restrict__ out) {
  int k = threadIdx.x;
  if (k >= n) return;
  float s = 0.f;
  for (int b = 0; b < nrows; b++) s += part[(size_t)b * n + k];
  out[k] += s;
}

// entropy_penalty = -mean(entropy) = -mean_a(0.5 + 0.5 log 2pi + log std_a): d/d std_a = -1/(A std_a) (ppo.py:343,380)
__global__ void entropy_grad_kernel(const float* __restrict__ stdv, int A, float ent_coeff, float* __restrict__ grad_std) {
  int a = threadIdx.x;
  if (a < A) grad_std[a] += -ent_coeff / ((float)A * stdv[a]);
}

// sum of squares of a flat range (grad norm), with pre-scale: per-block partials (fixed grid), summed in block order
#define SUMSQ_BLOCKS 128
// clip_grad_norm_ (coef = max_norm/(norm+1e-6), applied only if < 1) + torch.optim.Adam step; zeroes the gradient
// The two parameter groups (actor [+ stds], critic) in two launches instead of six: the per-block sums of squares of both groups
// from one grid, and one Adam grid over both groups whose blocks each add their group's SUMSQ_BLOCKS partials in the order
// sumsq_final_kernel used (same bits), instead of waiting for a one-thread launch per group to do it.
__global__ void __launch_bounds__(256) sumsq2_kernel(const float* __restrict__ g0, size_t n0, const float* __restrict__ g1, size_t n1, float scale,
                                                     float* __restrict__ part /* [2][SUMSQ_BLOCKS] */) {
  const int grp = blockIdx.x / SUMSQ_BLOCKS, b = blockIdx.x - grp * SUMSQ_BLOCKS;
  const float* __restrict__ g = grp ? g1 : g0;
  const size_t n = grp ? n1 : n0;
  float s = 0.f;
  for (size_t i = (size_t)b * blockDim.x + threadIdx.x; i < n; i += (size_t)SUMSQ_BLOCKS * blockDim.x) {
    float v = g[i] * scale;
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float red[4];
  LHW_LDS_POISON(red);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) adam2_kernel(float* __restrict__ theta, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                                    size_t n0, size_t off1, size_t n1, int blocks0, float gscale, const float* __restrict__ part,
                                                    float* __restrict__ normsq_out /* [2] */, float max_norm, float lr, float beta1, float beta2,
                                                    float eps, float bc1, float bc2sqrt) {
  const int grp = (int)blockIdx.x >= blocks0 ? 1 : 0;
  __shared__ float nsq;
  LHW_LDS_POISON(nsq);
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < SUMSQ_BLOCKS; b++) s += part[grp * SUMSQ_BLOCKS + b];
    nsq = s;
    if ((int)blockIdx.x == (grp ? blocks0 : 0)) normsq_out[grp] = s;
  }
  __syncthreads();
  const size_t i = (size_t)((int)blockIdx.x - (grp ? blocks0 : 0)) * blockDim.x + threadIdx.x;
  if (i >= (grp ? n1 : n0)) return;
  const size_t e = (grp ? off1 : 0) + i;
  float norm = sqrtf(nsq);
  float coef = max_norm / (norm + 1e-6f);
  coef = coef < 1.f ? coef : 1.f;
  float g = grad[e] * gscale * coef;
  float mi = beta1 * m[e] + (1.f - beta1) * g;
  float vi = beta2 * v[e] + (1.f - beta2) * g * g;
  m[e] = mi; v[e] = vi;
  float denom = sqrtf(vi) / bc2sqrt + eps;
  theta[e] -= (lr / bc1) * (mi / denom);
  grad[e] = 0.f;
}

// GAE(lambda) over a time-major rollout, one lane per env, float64 accumulation
// (reference rl/storage/rollout_storage.py:53-85 + the bootstrap rules of rl/workers/rollout_worker.py:163-190)
__global__ void __launch_bounds__(256) gae_kernel(int T, int N, const float* __restrict__ rew, const float* __restrict__ val,
                                                  const uint8_t* __restrict__ done, const float* __restrict__ vterm,
                                                  const float* __restrict__ vfinal, double gamma, double lam,
                                                  float* __restrict__ ret, float* __restrict__ adv) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double gae = 0.0, nextv = (double)vfinal[n];
  for (int t = T - 1; t >= 0; t--) {
    size_t i = (size_t)t * N + n;
    uint8_t f = done[i];
    if (f) {  // trajectory ends here: bootstrap (not done) * V(terminal obs), advantage recursion restarts
      nextv = (f & 1) ? 0.0 : (double)vterm[i];
      gae = 0.0;
    }
    double v = (double)val[i];
    double delta = (double)rew[i] + gamma * nextv - v;
    gae = delta + gamma * lam * gae;
    double r = gae + v;
    ret[i] = (float)r;
    adv[i] = (float)r - val[i];  // advantages = returns.float() - values.float() (ppo.py:484)
    nextv = v;
  }
}

// advantage normalisation: (a - mean) / (std_unbiased + eps) from global moments
#define MOM_BLOCKS 256
__global__ void __launch_bounds__(256) moments_kernel(const float* __restrict__ x, size_t n, double* __restrict__ part) {
  double s = 0, s2 = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double v = x[i];
    s += v; s2 += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); s2 += __shfl_xor(s2, o); }
  __shared__ double red[2][4];
  LHW_LDS_POISON(red);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
__global__ void moments_final_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
  if (threadIdx.x < 2) {
    double s = 0;
    for (int b = 0; b < n; b++) s += part[2 * b + threadIdx.x];
    out[threadIdx.x] = s;
  }
}
__global__ void scale_shift_kernel(float* __restrict__ x, size_t n, float mean, float inv) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (x[i] - mean) * inv;
}

// x <- (x - mean) / (std + eps) with mean / UNBIASED std formed on the device from (sum, sum of squares, count) -- the same double
// arithmetic, rounded to float32 at the same point, as the host path of lhw_scale_shift's callers (global_mean_std)
__global__ void standardize_kernel(float* __restrict__ x, size_t n, const double* __restrict__ st, double eps) {
  const double cnt = st[2], mean = st[0] / cnt;
  const double var = fmax(0.0, (st[1] - cnt * mean * mean) / fmax(1.0, cnt - 1.0));
  const float mean_f = (float)mean, inv_f = (float)(1.0 / (sqrt(var) + eps));
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (x[i] - mean_f) * inv_f;
}

// ------------------------------------------------------------------------------------------- C ABI
// Test / tuning hook: one GEMM of the update path on caller-provided device buffers (see include/lhw.h).
extern "C" int lhw_debug_gemm(int32_t a_kc, int32_t b_kc, int32_t wt, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                              const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int32_t relu, const float* mask,
                              int32_t ldmask, int32_t k_chunk, float* part, float* colsum, float* colsum_out, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || (lda | ldb | ldc) & 3) return lhw_fail(LHW_ERR_ARG, "lhw_debug_gemm: bad argument");
  if ((colsum && (a_kc || !part || !colsum_out)) || (k_chunk > 0 && k_chunk < K && !part)) return lhw_fail(LHW_ERR_ARG, "lhw_debug_gemm: split-K needs part; colsum needs A stored [K][M]");
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu;
  g.mask = mask; g.ldmask = ldmask; g.part = part; g.k_chunk = k_chunk; g.colsum = colsum;
  // wt = 16 .. 31: the fp16 GEMM; bits 0..3 of (wt - 16): A / B / C / mask are STORED as fp16 (else float32, rounded while staged)
  const int defer = part != nullptr, half = wt >= 16 && wt < 32;
  if (half) { g.a_half = (wt - 16) & 1; g.b_half = ((wt - 16) >> 1) & 1; g.c_half = ((wt - 16) >> 2) & 1; g.mask_half = ((wt - 16) >> 3) & 1; wt = 1; }
  if (a_kc && b_kc) launch_gemm<true, true>(g, s, defer, wt, half);
  else if (a_kc && !b_kc) launch_gemm<true, false>(g, s, defer, wt, half);
  else if (!a_kc && !b_kc) launch_gemm<false, false>(g, s, defer, wt, half);
  else return lhw_fail(LHW_ERR_UNSUPPORTED, "lhw_debug_gemm: A [K][M] with B [N][K] is not used by the update");
  if (part) {   // the deferred path of the update: partials (and column sums) reduced by one launch, accumulating into C / colsum_out
    const int kstep = half ? HBK : BK;
    const int kc = ((std::max(1, k_chunk > 0 ? k_chunk : K) + kstep - 1) / kstep) * kstep;
    SegList S;
    S.n = 0; S.scale = 1.f;
    seg_add(S, part, C, (K + kc - 1) / kc, M, N, ldc);
    if (colsum) seg_add(S, colsum, colsum_out, (K + kc - 1) / kc, M, 1, 1);
    launch_reduce_segments(S, s);
  }
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// Test hook: dW1 / db1 / dW3 / db3 of one network by the fused K-streaming kernel, ADDED to the outputs (scratch: slices x (256 Dp + 256 + 256 O + O) floats)
extern "C" int lhw_debug_wgrad_wide(const float* A, const float* B, int32_t K, int32_t k_chunk, float* part, float* colsum, void* stream) {
  if (!A || !B || !part || K <= 0 || k_chunk <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  launch_wgrad_wide(A, B, K, k_chunk, part, colsum, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? LHW_OK : lhw_fail(LHW_ERR_HIP, "wgrad_wide_kernel launch failed");
}
extern "C" int lhw_debug_wgrad_skinny(int32_t H, int32_t Dp, int32_t O, int32_t Op, const float* dh1, const float* x, int32_t ldx, const float* dy,
                                      const float* h2, int32_t R, float* dW1, float* db1, float* dW3, float* db3, float* scratch, void* stream) {
  if (!dh1 || !x || !dy || !h2 || !dW1 || !db1 || !dW3 || !db3 || !scratch || R <= 0) return lhw_fail(LHW_ERR_ARG, "lhw_debug_wgrad_skinny: bad argument");
  if (!wgrad_skinny_supported(H, Dp, O, Op) || ldx < Dp || (ldx & 3)) return lhw_fail(LHW_ERR_UNSUPPORTED, "fused skinny weight gradients: hidden 256, Dp <= 64, O <= 32");
  hipStream_t s = (hipStream_t)stream;
  const int kc = wgrad_skinny_chunk(R), ns = (R + kc - 1) / kc;
  WgradSkinnyArgs g{dh1, x, dy, h2, ldx, Dp, O, Op, R, kc, scratch, scratch + (size_t)ns * H * Dp, scratch + (size_t)ns * (H * Dp + H), scratch + (size_t)ns * (H * Dp + H + O * H)};
  hipLaunchKernelGGL(wgrad_skinny_kernel, dim3(ns), dim3(512), 0, s, g);
  SegList S;
  S.n = 0; S.scale = 1.f;
  seg_add(S, g.pw1, dW1, ns, H, Dp, Dp);
  seg_add(S, g.pb1, db1, ns, H, 1, 1);
  seg_add(S, g.pw3, dW3, ns, O, H, H);
  seg_add(S, g.pb3, db3, ns, O, 1, 1);
  launch_reduce_segments(S, s);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_ppo_create(const LhwPpoConfig* c, LhwPpo** out) {
  if (!c || !out) return lhw_fail(LHW_ERR_ARG, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return lhw_fail(LHW_ERR_NO_DEVICE, "no HIP device visible: liblhw has no CPU fallback");
  if (c->obs_dim <= 0 || c->act_dim <= 0 || c->act_dim > 32 || c->hidden <= 0 || c->hidden % 4 || c->max_rows <= 0)
    return lhw_fail(LHW_ERR_ARG, "bad PPO dimensions (act_dim <= 32, hidden %% 4 == 0)");
  HIPCHK(hipSetDevice(c->device));
  LhwPpo* p = new LhwPpo();
  p->device = c->device; p->D = c->obs_dim; p->A = c->act_dim; p->H = c->hidden; p->learn_std = c->learn_std;
  p->max_rows = c->max_rows;
  p->clip = c->clip; p->ent_coeff = c->entropy_coeff; p->mirror_coeff = c->mirror_coeff; p->grad_clip = c->max_grad_norm;
  p->lr = c->lr; p->adam_eps = c->eps; p->beta1 = 0.9f; p->beta2 = 0.999f;
  p->use_mirror = c->mirror_obs_src != nullptr;
  p->la = mlp_layout(p->D, p->H, p->A);
  p->lc = mlp_layout(p->D, p->H, 1);
  p->off_actor = 0;
  p->off_std = p->la.total;
  p->off_critic = p->off_std + pad4(p->A);
  p->n_params = p->off_critic + p->lc.total;
  const size_t R = p->max_rows, Dp = p->la.Dp, H = p->H, Op = p->la.Op;
  auto alloc = [&](float** ptr, size_t n) { return lhw_malloc(ptr, sizeof(float) * n) == hipSuccess && hipMemset(*ptr, 0, sizeof(float) * n) == hipSuccess; };
  bool ok = alloc(&p->xb, 2 * R * Dp) && alloc(&p->h1a, 2 * R * H) && alloc(&p->h2a, 2 * R * H) && alloc(&p->ya, 2 * R * Op) &&
            alloc(&p->h1c, R * H) && alloc(&p->h2c, R * H) && alloc(&p->yc, R * 4) && alloc(&p->dya, 2 * R * Op) &&
            alloc(&p->dh2a, 2 * R * H) && alloc(&p->dh1a, 2 * R * H) && alloc(&p->dyc, R * 4) && alloc(&p->dh2c, R * H) &&
            alloc(&p->dh1c, R * H) && alloc(&p->mb_act, R * p->A) && alloc(&p->mb_logp, R) && alloc(&p->mb_adv, R) &&
            alloc(&p->mb_ret, R) && alloc(&p->stats, 16) && alloc(&p->dstd, R * Op) && alloc(&p->stats_part, ((R + 255) / 256) * NSTAT) &&
            alloc(&p->norm_part, 2 * SUMSQ_BLOCKS);
  p->max_slices = (int)((R + 511) / 512);
  ok = ok && alloc(&p->part, std::max<size_t>((size_t)p->max_slices * H * std::max<size_t>(H, Dp), (size_t)COLSUM_CHUNKS * H));
  ok = ok && alloc(&p->bwd_part, bwd_parts_floats(p->la, R, 2) + bwd_parts_floats(p->lc, R, 1));
  if (mlp_strip_supported(p->la.H, p->la.Dp, p->la.O, p->la.Op)) ok = ok && alloc(&p->wt_a, mlp_strip_wt_floats(p->la.Dp, p->la.Op));
  if (mlp_strip_supported(p->lc.H, p->lc.Dp, p->lc.O, p->lc.Op)) ok = ok && alloc(&p->wt_c, mlp_strip_wt_floats(p->lc.Dp, p->lc.Op));
  if (p->wt_a && p->wt_c && R % 64 == 0 && !(getenv("LHW_STRIP_BITS") && atoi(getenv("LHW_STRIP_BITS")) == 0))
    ok = ok && alloc(reinterpret_cast<float**>(&p->bits_a), 4 * mlp_strip_bits_words(R)) && alloc(reinterpret_cast<float**>(&p->bits_c), 2 * mlp_strip_bits_words(R));
  if (p->wt_a && p->wt_c) ok = ok && alloc(&p->wt_inf, WT_SLOTS * (mlp_strip_wt_floats(p->la.Dp, p->la.Op) + mlp_strip_wt_floats(p->lc.Dp, p->lc.Op))) &&
                               alloc(&p->wt_roll, mlp_strip_wt_floats(p->la.Dp, p->la.Op) + mlp_strip_wt_floats(p->lc.Dp, p->lc.Op));
  if (ok && p->use_mirror) {
    std::vector<int> osrc(Dp, 0), asrc(p->A, 0);
    std::vector<float> osgn(Dp, 0.f), asgn(p->A, 0.f);
    for (int j = 0; j < p->D; j++) { osrc[j] = c->mirror_obs_src[j]; osgn[j] = c->mirror_obs_sign[j]; }
    for (int j = 0; j < p->A; j++) { asrc[j] = c->mirror_act_src[j]; asgn[j] = c->mirror_act_sign[j]; }
    for (int j = 0; j < p->D; j++) if (osrc[j] < 0 || osrc[j] >= p->D) { ok = false; }
    for (int j = 0; j < p->A; j++) if (asrc[j] < 0 || asrc[j] >= p->A) { ok = false; }
    ok = ok && lhw_malloc(&p->d_obs_src, sizeof(int) * Dp) == hipSuccess && lhw_malloc(&p->d_act_src, sizeof(int) * p->A) == hipSuccess &&
         lhw_malloc(&p->d_obs_sign, sizeof(float) * Dp) == hipSuccess && lhw_malloc(&p->d_act_sign, sizeof(float) * p->A) == hipSuccess;
    if (ok) {
      (void)hipMemcpy(p->d_obs_src, osrc.data(), sizeof(int) * Dp, hipMemcpyHostToDevice);
      (void)hipMemcpy(p->d_act_src, asrc.data(), sizeof(int) * p->A, hipMemcpyHostToDevice);
      (void)hipMemcpy(p->d_obs_sign, osgn.data(), sizeof(float) * Dp, hipMemcpyHostToDevice);
      (void)hipMemcpy(p->d_act_sign, asgn.data(), sizeof(float) * p->A, hipMemcpyHostToDevice);
    }
  }
  p->two_streams = !(getenv("LHW_PPO_TWO_STREAMS") && atoi(getenv("LHW_PPO_TWO_STREAMS")) == 0);
  ok = ok && hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess &&
       hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    lhw_ppo_destroy(p);
    return lhw_fail(LHW_ERR_HIP, "PPO workspace allocation failed (max_rows=%d) or bad mirror table", c->max_rows);
  }
  *out = p;
  return LHW_OK;
}

extern "C" int lhw_ppo_destroy(LhwPpo* p) {
  if (!p) return LHW_OK;
  (void)hipSetDevice(p->device);
  float* bufs[] = {p->xb, p->h1a, p->h2a, p->ya, p->h1c, p->h2c, p->yc, p->dya, p->dh2a, p->dh1a, p->dyc, p->dh2c, p->dh1c,
                   p->mb_act, p->mb_logp, p->mb_adv, p->mb_ret, p->stats, p->d_obs_sign, p->d_act_sign, p->part, p->dstd, p->stats_part,
                   p->norm_part, p->bwd_part, p->wt_a, p->wt_c, p->wt_inf, p->wt_roll, reinterpret_cast<float*>(p->bits_a), reinterpret_cast<float*>(p->bits_c)};
  for (float* b : bufs) if (b) (void)hipFree(b);
  _Float16* hbufs[] = {p->xb_h, p->h1a_h, p->h2a_h, p->dh2a_h, p->dh1a_h, p->h1c_h, p->h2c_h, p->dh2c_h, p->dh1c_h};
  for (_Float16* b : hbufs) if (b) (void)hipFree(b);
  if (p->d_obs_src) (void)hipFree(p->d_obs_src);
  if (p->d_act_src) (void)hipFree(p->d_act_src);
  if (p->step_exec) (void)hipGraphExecDestroy(p->step_exec);
  if (p->step_graph) (void)hipGraphDestroy(p->step_graph);
  if (p->side) { (void)hipStreamSynchronize(p->side); (void)hipStreamDestroy(p->side); }
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  delete p;
  return LHW_OK;
}

extern "C" int lhw_ppo_set_inference_dtype(LhwPpo* p, int fp16) {
  if (!p) return lhw_fail(LHW_ERR_ARG, "null ppo");
  p->infer_half = fp16 ? 1 : 0;
  return LHW_OK;
}

extern "C" int lhw_ppo_set_update_dtype(LhwPpo* p, int fp16) {
  if (!p) return lhw_fail(LHW_ERR_ARG, "null ppo");
  if (fp16 && !p->xb_h && !(getenv("LHW_FP16_STORAGE") && atoi(getenv("LHW_FP16_STORAGE")) == 0)) {
    // fp16 HBM storage of the minibatch activations (round 6; LHW_FP16_STORAGE=0: float32 storage rounded per GEMM, as through round 5)
    HIPCHK(hipSetDevice(p->device));
    const size_t R = p->max_rows, H = p->H;
    p->ldxh = (p->la.Dp + 7) & ~7;
    auto alloch = [&](_Float16** ptr, size_t n) { return lhw_malloc(ptr, sizeof(_Float16) * n) == hipSuccess && hipMemset(*ptr, 0, sizeof(_Float16) * n) == hipSuccess; };
    const bool ok = alloch(&p->xb_h, 2 * R * p->ldxh) && alloch(&p->h1a_h, 2 * R * H) && alloch(&p->h2a_h, 2 * R * H) && alloch(&p->dh2a_h, 2 * R * H) &&
                    alloch(&p->dh1a_h, 2 * R * H) && alloch(&p->h1c_h, R * H) && alloch(&p->h2c_h, R * H) && alloch(&p->dh2c_h, R * H) && alloch(&p->dh1c_h, R * H);
    if (!ok) return lhw_fail(LHW_ERR_HIP, "fp16 workspace allocation failed (max_rows=%d)", p->max_rows);
  }
  p->update_half = fp16 ? 1 : 0;
  return LHW_OK;
}

extern "C" int64_t lhw_ppo_param_count(const LhwPpo* p) { return p ? (int64_t)p->n_params : LHW_ERR_ARG; }

// offsets (in floats) of each tensor inside the flat parameter vector:
// out[0..5] actor W1,b1,W2,b2,W3,b3 ; out[6] stds ; out[7..12] critic W1..b3 ; out[13] obs pad width Dp ; out[14] actor Op
extern "C" int lhw_ppo_layout(const LhwPpo* p, int64_t* out15) {
  if (!p || !out15) return lhw_fail(LHW_ERR_ARG, "null argument");
  const MlpLayout &a = p->la, &c = p->lc;
  int64_t v[15] = {(int64_t)(p->off_actor + a.w1), (int64_t)(p->off_actor + a.b1), (int64_t)(p->off_actor + a.w2),
                   (int64_t)(p->off_actor + a.b2), (int64_t)(p->off_actor + a.w3), (int64_t)(p->off_actor + a.b3),
                   (int64_t)p->off_std,
                   (int64_t)(p->off_critic + c.w1), (int64_t)(p->off_critic + c.b1), (int64_t)(p->off_critic + c.w2),
                   (int64_t)(p->off_critic + c.b2), (int64_t)(p->off_critic + c.w3), (int64_t)(p->off_critic + c.b3),
                   (int64_t)a.Dp, (int64_t)a.Op};
  memcpy(out15, v, sizeof v);
  return LHW_OK;
}

// normalised (and mirrored) copies of R raw observation rows: xn/xm [R][Dp]
extern "C" int lhw_ppo_normalize(LhwPpo* p, const float* obs, int64_t R, const float* obs_mean, const float* obs_std, float* xn,
                                 float* xm, void* stream) {
  if (!p || !obs || !xn || R <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  if (xm && !p->use_mirror) return lhw_fail(LHW_ERR_ARG, "mirror output requested but no mirror tables configured");
  HIPCHK(hipSetDevice(p->device));
  size_t n = (size_t)R * p->la.Dp;
  hipLaunchKernelGGL(normalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs, p->D, p->la.Dp, (size_t)R,
                     obs_mean, obs_std, xn, xm, p->d_obs_src, p->d_obs_sign);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// ws_row: first row of the forward workspace to use (concurrent calls on different streams must use disjoint row ranges)
static int ppo_forward_impl(LhwPpo* p, const float* theta, const float* obs, int64_t N, const float* obs_mean, const float* obs_std,
                            uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic, int64_t ws_row, float* mu,
                            float* act, float* logp, float* value, void* stream) {
  if (!p || !theta || !obs || N <= 0 || ws_row < 0 || ws_row + N > p->max_rows)
    return lhw_fail(LHW_ERR_ARG, "bad argument (rows [%lld, %lld), capacity %d)", (long long)ws_row, (long long)(ws_row + N), p ? p->max_rows : 0);
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t Dp = p->la.Dp, H = p->H, Op = p->la.Op, r0 = (size_t)ws_row;
  float *xb = p->xb + r0 * Dp, *h1a = p->h1a + r0 * H, *h2a = p->h2a + r0 * H, *ya = p->ya + r0 * Op;
  float *h1c = p->h1c + r0 * H, *h2c = p->h2c + r0 * H, *yc = p->yc + r0 * 4;
  size_t n = (size_t)N * Dp;
  float *wta = nullptr, *wtc = nullptr;   // this call's weight copies for the strip kernel
  const bool wt_ready = p->roll_theta != nullptr && p->roll_theta == theta;   // inside a rollout bracket: made once by lhw_ppo_begin_rollout
  if (strip_mode() >= 2 && p->wt_inf) {
    const size_t fa = mlp_strip_wt_floats(p->la.Dp, p->la.Op), fc = mlp_strip_wt_floats(p->lc.Dp, p->lc.Op);
    wta = wt_ready ? p->wt_roll : p->wt_inf + (size_t)(ws_row * WT_SLOTS / p->max_rows) * (fa + fc);
    wtc = wta + fa;
  }
  // the rollout's policy step (actions + log-densities only) as ONE strip launch: observation normalisation on the way into the
  // slab, the three layers, the Gaussian head on the read-out -- instead of normalise / forward / sample launches
  // (the fused staging reads RAW observation rows of width obs_dim: without the normalisation vectors it would have to copy rows
  // of width Dp, which the caller's buffer does not have -- those calls take the three-launch path)
  const bool fused = act && logp && !mu && !value && wta && obs_mean && obs_std && !p->infer_half && mlp_strip_supported(p->la.H, p->la.Dp, p->la.O, p->la.Op);
  if (fused) {
    const MlpLayout& La = p->la;
    const float* th = theta + p->off_actor;
    if (!wt_ready) mlp_strip_prepare(th + La.w1, th + La.w2, th + La.w3, La.Dp, La.O, La.Op, wta, s);
    MlpStripFwd a{wta, th + La.b1, wta + (size_t)La.Dp * La.H, th + La.b2, wta + (size_t)La.Dp * La.H + (size_t)La.H * La.H, th + La.b3,
                  obs, p->D, La.Dp, La.O, La.Op, (int)N, nullptr, nullptr, ya};
    a.in_mean = obs_mean; a.in_std = obs_std; a.in_dim = p->D;
    a.stdv = theta + p->off_std; a.act = act; a.logp = logp;
    a.seed = seed; a.env_base = env_id_base; a.counter = counter; a.deterministic = deterministic;
    mlp_strip_forward(a, s);
    HIPCHK(hipGetLastError());
    return LHW_OK;
  }
  hipLaunchKernelGGL(normalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, obs, p->D, p->la.Dp, (size_t)N, obs_mean, obs_std,
                     xb, (float*)nullptr, (const int*)nullptr, (const float*)nullptr);
  if (act || mu) {
    mlp_forward(p->la, theta + p->off_actor, xb, p->la.Dp, (int)N, h1a, h2a, ya, s, p->infer_half, wta, wt_ready, false);
    if (mu) HIPCHK(hipMemcpy2DAsync(mu, sizeof(float) * p->A, ya, sizeof(float) * p->la.Op, sizeof(float) * p->A, N, hipMemcpyDeviceToDevice, s));
    if (act) {
      if (!logp) return lhw_fail(LHW_ERR_ARG, "logp required with act");
      hipLaunchKernelGGL(sample_kernel, dim3((N + 7) / 8), dim3(256), 0, s, ya, p->la.Op, p->A, (int)N, theta + p->off_std,
                         seed, env_id_base, counter, deterministic, act, logp);
    }
  }
  if (value) {
    mlp_forward(p->lc, theta + p->off_critic, xb, p->la.Dp, (int)N, h1c, h2c, yc, s, p->infer_half, wtc, wt_ready, false);
    HIPCHK(hipMemcpy2DAsync(value, sizeof(float), yc, sizeof(float) * 4, sizeof(float), N, hipMemcpyDeviceToDevice, s));
  }
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_ppo_begin_rollout(LhwPpo* p, const float* theta, void* stream) {
  if (!p || !theta) return lhw_fail(LHW_ERR_ARG, "null argument");
  p->roll_theta = nullptr;
  if (strip_mode() < 2 || !p->wt_roll) return LHW_OK;   // per-layer GEMM inference reads theta itself: nothing to prepare
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = (hipStream_t)stream;
  const MlpLayout &La = p->la, &Lc = p->lc;
  const float *tha = theta + p->off_actor, *thc = theta + p->off_critic;
  mlp_strip_prepare(tha + La.w1, tha + La.w2, tha + La.w3, La.Dp, La.O, La.Op, p->wt_roll, s);
  mlp_strip_prepare(thc + Lc.w1, thc + Lc.w2, thc + Lc.w3, Lc.Dp, Lc.O, Lc.Op, p->wt_roll + mlp_strip_wt_floats(La.Dp, La.Op), s);
  HIPCHK(hipGetLastError());
  p->roll_theta = theta;
  return LHW_OK;
}
extern "C" int lhw_ppo_end_rollout(LhwPpo* p) {
  if (!p) return lhw_fail(LHW_ERR_ARG, "null ppo");
  p->roll_theta = nullptr;
  return LHW_OK;
}

extern "C" int lhw_ppo_rollout_policy(LhwPpo* p, const float* theta, const float* obs_mean, const float* obs_std, uint64_t seed,
                                      uint32_t counter, int deterministic, LhwRolloutPolicy* out) {
  if (!p || !theta || !obs_mean || !obs_std || !out) return lhw_fail(LHW_ERR_ARG, "null argument");
  const MlpLayout& La = p->la;
  if (p->roll_theta == nullptr || p->roll_theta != theta || !p->wt_roll)
    return lhw_fail(LHW_ERR_UNSUPPORTED, "lhw_ppo_rollout_policy: no rollout bracket open for this theta (lhw_ppo_begin_rollout)");
  if (!mlp_strip_supported(La.H, La.Dp, La.O, La.Op))
    return lhw_fail(LHW_ERR_UNSUPPORTED, "lhw_ppo_rollout_policy: actor with hidden width 256 only");
  const float* th = theta + p->off_actor;
  const float* wt = p->wt_roll;
  out->w1t = wt; out->b1 = th + La.b1;
  out->w2t = wt + (size_t)La.Dp * La.H; out->b2 = th + La.b2;
  out->w3t = wt + (size_t)La.Dp * La.H + (size_t)La.H * La.H; out->b3 = th + La.b3;
  out->stdv = theta + p->off_std; out->obs_mean = obs_mean; out->obs_std = obs_std;
  out->obs_dim = p->D; out->obs_pad = La.Dp; out->act_dim = La.O; out->act_pad = La.Op; out->hidden = La.H;
  out->deterministic = deterministic; out->seed = seed; out->counter = counter;
  out->fp16_operands = p->infer_half ? 1 : 0;
  return LHW_OK;
}

// Rollout inference for N rows (N <= max_rows): normalise, actor + critic forward, sample.
//   act/logp/mu may be NULL to run the critic only; value may be NULL to run the actor only.
extern "C" int lhw_ppo_forward(LhwPpo* p, const float* theta, const float* obs, int64_t N, const float* obs_mean,
                               const float* obs_std, uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic,
                               float* mu, float* act, float* logp, float* value, void* stream) {
  return ppo_forward_impl(p, theta, obs, N, obs_mean, obs_std, seed, env_id_base, counter, deterministic, 0, mu, act, logp, value, stream);
}
extern "C" int lhw_ppo_forward_at(LhwPpo* p, const float* theta, const float* obs, int64_t N, const float* obs_mean,
                                  const float* obs_std, uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic,
                                  int64_t ws_row, float* mu, float* act, float* logp, float* value, void* stream) {
  return ppo_forward_impl(p, theta, obs, N, obs_mean, obs_std, seed, env_id_base, counter, deterministic, ws_row, mu, act, logp, value, stream);
}

#define LHW_MAX_DEVICES 64
// device that owns a device pointer; makes it current (the handle-free entry points below have no LhwPpo to ask)
static int device_of(const void* ptr, int* dev) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, ptr) != hipSuccess) return lhw_fail(LHW_ERR_ARG, "not a device pointer");
  if (at.device < 0 || at.device >= LHW_MAX_DEVICES) return lhw_fail(LHW_ERR_ARG, "device %d out of range", at.device);
  if (hipSetDevice(at.device) != hipSuccess) return lhw_fail(LHW_ERR_HIP, "hipSetDevice(%d) failed", at.device);
  *dev = at.device;
  return 0;
}

extern "C" int lhw_gae(int32_t T, int32_t N, const float* rew, const float* val, const uint8_t* done, const float* vterm,
                       const float* vfinal, double gamma, double lam, float* ret, float* adv, void* stream) {
  if (T <= 0 || N <= 0 || !rew || !val || !done || !vterm || !vfinal || !ret || !adv) return lhw_fail(LHW_ERR_ARG, "bad argument");
  int dev = 0;
  if (device_of(rew, &dev)) return LHW_ERR_HIP;
  hipLaunchKernelGGL(gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, N, rew, val, done, vterm, vfinal,
                     gamma, lam, ret, adv);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// sum and sum of squares (float64) of x[0..n): the caller all-reduces them across GPUs, then calls lhw_scale_shift
extern "C" int lhw_moments(const float* x, int64_t n, double* out2_dev, void* stream) {
  if (!x || !out2_dev || n <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  // handle-free entry point: run on the device that owns x, with that device's own partial-sum buffer (kept for the
  // process lifetime; one per device, so a process driving several GPUs never hands a kernel a foreign-device pointer)
  int dev = 0;
  if (device_of(x, &dev)) return LHW_ERR_HIP;
  static double* scratch_of[LHW_MAX_DEVICES] = {nullptr};
  static std::mutex mu;
  double* scratch;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!scratch_of[dev]) HIPCHK(lhw_malloc(&scratch_of[dev], sizeof(double) * 2 * MOM_BLOCKS));
    scratch = scratch_of[dev];
  }
  hipLaunchKernelGGL(moments_kernel, dim3(MOM_BLOCKS), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, scratch);
  hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, MOM_BLOCKS, out2_dev);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}
extern "C" int lhw_scale_shift(float* x, int64_t n, float mean, float inv_scale, void* stream) {
  if (!x || n <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  int dev = 0;
  if (device_of(x, &dev)) return LHW_ERR_HIP;
  hipLaunchKernelGGL(scale_shift_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, mean, inv_scale);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_standardize(float* x, int64_t n, const double* stats3_dev, double eps, void* stream) {
  if (!x || !stats3_dev || n <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  int dev = 0;
  if (device_of(x, &dev)) return LHW_ERR_HIP;
  hipLaunchKernelGGL(standardize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, stats3_dev, eps);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// One minibatch: gather rows idx[0..B) from the iteration's buffers, forward (policy on obs and on mirrored
// obs, critic), losses, backward.  Gradients are ACCUMULATED into grad (flat, same layout as theta);
// stats_dev[0..4] += actor_loss, critic_loss, mirror_loss, approx_kl, clip_fraction of this minibatch.
// Arms the imitation term for the next lhw_ppo_grad call: target / mask are device arrays [B][act_dim] in minibatch row
// order (row r belongs to idx[r]); n_selected = number of set mask entries (the mean's denominator).
extern "C" int lhw_ppo_set_imitation(LhwPpo* p, const float* target, const uint8_t* mask, float coeff, int64_t n_selected) {
  if (!p) return lhw_fail(LHW_ERR_ARG, "null ppo");
  if (!target || !mask || n_selected <= 0) { p->imit_target = nullptr; p->imit_mask = nullptr; return LHW_OK; }
  p->imit_target = target; p->imit_mask = mask; p->imit_coeff = coeff; p->imit_inv_count = 1.f / (float)n_selected;
  return LHW_OK;
}

extern "C" int lhw_ppo_grad(LhwPpo* p, const float* theta, float* grad, const float* xn, const float* xm, const float* act,
                            const float* old_logp, const float* adv, const float* ret, const int32_t* idx, int32_t B,
                            float* stats_dev, void* stream) {
  if (!p || !theta || !grad || !xn || !act || !old_logp || !adv || !ret || !idx || !stats_dev) return lhw_fail(LHW_ERR_ARG, "null argument");
  if (B <= 0 || B > p->max_rows) return lhw_fail(LHW_ERR_ARG, "minibatch %d exceeds workspace capacity %d", B, p->max_rows);
  const int mir = p->use_mirror && xm != nullptr;
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = (hipStream_t)stream;
  const int R = p->max_rows, Dp = p->la.Dp, Op = p->la.Op;
  size_t n = (size_t)B * Dp;
  const float* th_a = theta + p->off_actor;
  const float* th_c = theta + p->off_critic;
  hipStream_t sc = p->two_streams ? p->side : s;   // the critic's chain
  auto fork = [&]() { if (sc != s) { (void)hipEventRecord(p->ev_fork, s); (void)hipStreamWaitEvent(sc, p->ev_fork, 0); } };
  auto join = [&]() { if (sc != s) { (void)hipEventRecord(p->ev_join, sc); (void)hipStreamWaitEvent(s, p->ev_join, 0); } };
  // the [in][out] weight copies of the forward strips are made on the side stream while the minibatch is gathered (round 6: the two
  // 9 us transposes were the first links of the step's chain)
  const bool strips = strip_mode() >= 1 && !p->update_half && p->wt_a && p->wt_c && mlp_strip_supported(p->la.H, p->la.Dp, p->la.O, p->la.Op) &&
                      mlp_strip_supported(p->lc.H, p->lc.Dp, p->lc.O, p->lc.Op);
  if (strips) {
    fork();
    mlp_strip_prepare(th_a + p->la.w1, th_a + p->la.w2, th_a + p->la.w3, p->la.Dp, p->la.O, p->la.Op, p->wt_a, sc);
    mlp_strip_prepare(th_c + p->lc.w1, th_c + p->lc.w2, th_c + p->lc.w3, p->lc.Dp, p->lc.O, p->lc.Op, p->wt_c, sc);
  }
  hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, idx, B, R, Dp, p->A, xn, mir ? xm : nullptr, act, old_logp,
                     adv, ret, p->xb, p->mb_act, p->mb_logp, p->mb_adv, p->mb_ret);
  if (strips) join();
  // --fp16 update with fp16 storage: fp16 copies of the gathered rows; every activation the GEMMs exchange stays fp16 in HBM
  const bool hstore = p->update_half && p->xb_h != nullptr;
  HalfBufs ha{}, hc{}, ham{};
  if (hstore) {
    const int ldh = p->ldxh;
    if (mir && B == R) {
      const size_t nn = (size_t)2 * B * ldh;
      hipLaunchKernelGGL(rows_to_half_kernel, dim3((nn + 255) / 256), dim3(256), 0, s, p->xb, Dp, Dp, (size_t)2 * B, p->xb_h, ldh);
    } else {
      const size_t nn = (size_t)B * ldh;
      hipLaunchKernelGGL(rows_to_half_kernel, dim3((nn + 255) / 256), dim3(256), 0, s, p->xb, Dp, Dp, (size_t)B, p->xb_h, ldh);
      if (mir) hipLaunchKernelGGL(rows_to_half_kernel, dim3((nn + 255) / 256), dim3(256), 0, s, p->xb + (size_t)R * Dp, Dp, Dp, (size_t)B, p->xb_h + (size_t)R * ldh, ldh);
    }
    ha = HalfBufs{p->xb_h, ldh, p->h1a_h, p->h2a_h, p->dh2a_h, p->dh1a_h};
    hc = HalfBufs{p->xb_h, ldh, p->h1c_h, p->h2c_h, p->dh2c_h, p->dh1c_h};
    ham = HalfBufs{p->xb_h + (size_t)R * ldh, ldh, p->h1a_h + (size_t)R * p->H, p->h2a_h + (size_t)R * p->H, p->dh2a_h + (size_t)R * p->H, p->dh1a_h + (size_t)R * p->H};
  }
  const HalfBufs *pha = hstore ? &ha : nullptr, *phc = hstore ? &hc : nullptr, *pham = hstore ? &ham : nullptr;
  // forward: rows [0,B) and, if mirroring, rows [R, R+B)
  // ReLU masks as bits from the forward strips to the backward strips (per layer: normal rows, then the mirrored rows' launch)
  const size_t bw = mlp_strip_bits_words(R);
  unsigned *ba1 = p->bits_a, *ba2 = p->bits_a ? p->bits_a + 2 * bw : nullptr, *bc1 = p->bits_c, *bc2 = p->bits_c ? p->bits_c + bw : nullptr;
  fork();
  mlp_forward(p->lc, th_c, p->xb, Dp, B, p->h1c, p->h2c, p->yc, sc, p->update_half, strip_mode() >= 1 ? p->wt_c : nullptr, strips, true, phc, bc1, bc2);
  if (mir && B == R) {
    mlp_forward(p->la, th_a, p->xb, Dp, 2 * B, p->h1a, p->h2a, p->ya, s, p->update_half, strip_mode() >= 1 ? p->wt_a : nullptr, strips, true, pha, ba1, ba2);   // mirrored rows follow without a gap
  } else {
    mlp_forward(p->la, th_a, p->xb, Dp, B, p->h1a, p->h2a, p->ya, s, p->update_half, strip_mode() >= 1 ? p->wt_a : nullptr, strips, true, pha, ba1, ba2);
    if (mir)
      mlp_forward(p->la, th_a, p->xb + (size_t)R * Dp, Dp, B, p->h1a + (size_t)R * p->H, p->h2a + (size_t)R * p->H, p->ya + (size_t)R * Op, s, p->update_half, strip_mode() >= 1 ? p->wt_a : nullptr, strips, true, pham,
                  ba1 ? ba1 + bw : nullptr, ba2 ? ba2 + bw : nullptr);
  }
  join();
  const int nblk = (B + 255) / 256;
  // fp16 update: the back-propagated gradients are rounded to fp16 per GEMM, and d loss / d output carries 1 / B -- at B = 32768
  // most of it would fall into the fp16 subnormal range.  Loss scaling by a power of two (exact in float32): the read-out
  // gradients are multiplied by 2^ceil(log2 B) here and the weight-gradient totals divided by it in the final ordered reduction.
  const float lscale = p->update_half ? exp2f(ceilf(log2f((float)B))) : 1.f;
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(nblk), dim3(256), 0, s, B, R, p->A, Op, p->ya, p->yc, p->mb_act, p->mb_logp,
                     p->mb_adv, p->mb_ret, theta + p->off_std, p->clip, p->mirror_coeff, mir, p->d_act_src, p->d_act_sign, p->dya,
                     p->dyc, p->learn_std ? p->dstd : (float*)nullptr, p->stats_part, p->imit_target, p->imit_mask, p->imit_coeff,
                     p->imit_inv_count, 0, lscale);
  p->imit_target = nullptr; p->imit_mask = nullptr;   // armed for one call only
  if (p->learn_std) {
    colsum_det(p->dstd, B, Op, p->A, grad + p->off_std, p->part, s);
    hipLaunchKernelGGL(entropy_grad_kernel, dim3(1), dim3(64), 0, s, theta + p->off_std, p->A, p->ent_coeff, grad + p->off_std);
  }
  const BwdParts Pa = bwd_parts_carve(p->la, R, 2, p->bwd_part);
  const BwdParts Pc = bwd_parts_carve(p->lc, R, 1, p->bwd_part + bwd_parts_floats(p->la, R, 2));
  BwdSlices za, zc;
  fork();
  // (the step's loss statistics -- logging only -- are summed at the head of the side stream, off the actor's chain)
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, sc, p->stats_part, nblk, NSTAT, stats_dev);
  mlp_backward(p->lc, th_c, p->xb, Dp, B, p->h1c, p->h2c, p->dyc, p->dh2c, p->dh1c, Pc, zc, sc, p->update_half, phc, bc1, bc2);
  if (mir && B == R) {
    // the mirrored rows follow the normal ones without a gap: one pass over 2B rows
    mlp_backward(p->la, th_a, p->xb, Dp, 2 * B, p->h1a, p->h2a, p->dya, p->dh2a, p->dh1a, Pa, za, s, p->update_half, pha, ba1, ba2);
  } else {
    mlp_backward(p->la, th_a, p->xb, Dp, B, p->h1a, p->h2a, p->dya, p->dh2a, p->dh1a, Pa, za, s, p->update_half, pha, ba1, ba2);
    if (mir)
      mlp_backward(p->la, th_a, p->xb + (size_t)R * Dp, Dp, B, p->h1a + (size_t)R * p->H, p->h2a + (size_t)R * p->H,
                   p->dya + (size_t)R * Op, p->dh2a + (size_t)R * p->H, p->dh1a + (size_t)R * p->H, Pa, za, s, p->update_half, pham,
                   ba1 ? ba1 + bw : nullptr, ba2 ? ba2 + bw : nullptr);
  }
  join();
  SegList S;
  S.n = 0; S.scale = 1.f / lscale;
  mlp_backward_segments(S, p->la, grad + p->off_actor, Pa, za);
  mlp_backward_segments(S, p->lc, grad + p->off_critic, Pc, zc);
  launch_reduce_segments(S, s);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// dual clip_grad_norm_ + Adam on the two parameter groups [0, na) and [off_critic, off_critic + nc) of a flat vector
static void clip_and_adam(float* theta, float* grad, float* adam_m, float* adam_v, size_t na, size_t off_critic, size_t nc,
                          int64_t step, float grad_scale, float* norm_part, float* stats, float grad_clip, float lr, float beta1,
                          float beta2, float adam_eps, hipStream_t s) {
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const int blocks0 = (int)((na + 255) / 256), blocks1 = (int)((nc + 255) / 256);
  hipLaunchKernelGGL(sumsq2_kernel, dim3(2 * SUMSQ_BLOCKS), dim3(256), 0, s, grad, na, grad + off_critic, nc, grad_scale, norm_part);
  hipLaunchKernelGGL(adam2_kernel, dim3(blocks0 + blocks1), dim3(256), 0, s, theta, grad, adam_m, adam_v, na, off_critic, nc, blocks0, grad_scale,
                     norm_part, stats + 8, grad_clip, lr, beta1, beta2, adam_eps, bc1, sqrtf(bc2));
}

// clip_grad_norm_ on the actor and critic parameter groups separately, then one Adam step each; zeroes grad.
// grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).  step is the 1-based Adam step count.
extern "C" int lhw_ppo_apply(LhwPpo* p, float* theta, float* grad, float* adam_m, float* adam_v, int64_t step, float grad_scale,
                             void* stream) {
  if (!p || !theta || !grad || !adam_m || !adam_v || step <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(p->device));
  p->roll_theta = nullptr;   // theta changes: the weight copies of an open rollout bracket are stale
  hipStream_t s = (hipStream_t)stream;
  const size_t na = p->learn_std ? p->off_std + p->A : p->off_std;  // actor group (+ stds if they are parameters)
  clip_and_adam(theta, grad, adam_m, adam_v, na, p->off_critic, p->lc.total, step, grad_scale, p->norm_part, p->stats, p->grad_clip,
                p->lr, p->beta1, p->beta2, p->adam_eps, s);
  if (!p->learn_std) HIPCHK(hipMemsetAsync(grad + p->off_std, 0, sizeof(float) * pad4(p->A), s));
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// One optimiser step as ONE graph launch (round 6).  lhw_ppo_grad + lhw_ppo_apply are some forty launches on two streams, a dozen of
// them small (gather, loss, ordered reductions, transposes, clip, Adam: 5-20 us of work each behind a launch gap of the same order);
// captured once per (buffers, minibatch size) as a hipGraph they replay with one host call and the runtime's graph scheduling between the
// nodes.  Same kernels, same order, same arithmetic: bitwise the weights of the two-call path (tests/test_ppo_gpu.py).  What changes from
// step to step is patched into the executable graph: the minibatch's index pointer (gather_kernel's first argument) and Adam's bias
// corrections (adam2_kernel's last two).  Single process only -- with data parallelism the gradient all-reduce sits between the two halves
// (the Python layer then keeps lhw_ppo_grad / all-reduce / lhw_ppo_apply).  LHW_PPO_GRAPH=0 turns it off (the two calls, eagerly).
static bool ppo_graph_on() {
  static const bool on = !(getenv("LHW_PPO_GRAPH") && atoi(getenv("LHW_PPO_GRAPH")) == 0);
  return on;
}
extern "C" int lhw_ppo_step(LhwPpo* p, float* theta, float* grad, float* adam_m, float* adam_v, const float* xn, const float* xm, const float* act,
                            const float* old_logp, const float* adv, const float* ret, const int32_t* idx, int32_t B, float* stats_dev,
                            int64_t step, float grad_scale, void* stream) {
  if (!p || !theta || !grad || !adam_m || !adam_v || !idx || step <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  // (an armed imitation term is a one-shot argument of the next lhw_ppo_grad: not captured; the legacy default stream cannot be captured)
  if (!ppo_graph_on() || p->imit_target != nullptr || s == nullptr) {
    const int rc = lhw_ppo_grad(p, theta, grad, xn, xm, act, old_logp, adv, ret, idx, B, stats_dev, stream);
    return rc ? rc : lhw_ppo_apply(p, theta, grad, adam_m, adam_v, step, grad_scale, stream);
  }
  HIPCHK(hipSetDevice(p->device));
  const void* key[14] = {theta, grad, adam_m, adam_v, xn, xm, act, old_logp, adv, ret, stats_dev, stream, nullptr, nullptr};
  float gs_key; memcpy(&gs_key, &grad_scale, sizeof gs_key);
  bool same = p->step_exec != nullptr && p->step_key_b == B && p->step_key_half == p->update_half;
  for (int i = 0; same && i < 12; i++) same = p->step_key[i] == key[i];
  const size_t na = p->learn_std ? p->off_std + p->A : p->off_std;
  const float bc1 = 1.f - powf(p->beta1, (float)step), bc2s = sqrtf(1.f - powf(p->beta2, (float)step));
  if (!same) {
    if (p->step_exec) { (void)hipGraphExecDestroy(p->step_exec); p->step_exec = nullptr; }
    if (p->step_graph) { (void)hipGraphDestroy(p->step_graph); p->step_graph = nullptr; }
    p->node_gather = p->node_adam = nullptr;
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = lhw_ppo_grad(p, theta, grad, xn, xm, act, old_logp, adv, ret, idx, B, stats_dev, stream);
    if (!rc) rc = lhw_ppo_apply(p, theta, grad, adam_m, adam_v, step, grad_scale, stream);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (ec != hipSuccess || !g) return lhw_fail(LHW_ERR_HIP, "lhw_ppo_step: stream capture failed: %s", hipGetErrorString(ec));
    p->step_graph = g;
    size_t nn = 0;
    HIPCHK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    HIPCHK(hipGraphGetNodes(g, nodes.data(), &nn));
    for (hipGraphNode_t nd : nodes) {
      hipGraphNodeType ty;
      if (hipGraphNodeGetType(nd, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams kp;
      if (hipGraphKernelNodeGetParams(nd, &kp) != hipSuccess) continue;
      if (kp.func == (void*)gather_kernel) p->node_gather = nd;
      else if (kp.func == (void*)adam2_kernel) p->node_adam = nd;
    }
    if (!p->node_gather || !p->node_adam) return lhw_fail(LHW_ERR_HIP, "lhw_ppo_step: gather / Adam nodes not found in the captured graph (%zu nodes)", nn);
    HIPCHK(hipGraphInstantiate(&p->step_exec, g, nullptr, nullptr, 0));
    for (int i = 0; i < 12; i++) p->step_key[i] = key[i];
    p->step_key_b = B; p->step_key_half = p->update_half;
  }
  // patch the two nodes: the kernels' full argument lists, as lhw_ppo_grad / clip_and_adam pass them
  {
    const int mir = p->use_mirror && xm != nullptr;
    const int* a_idx = idx; int a_B = B, a_R = p->max_rows, a_Dp = p->la.Dp, a_A = p->A;
    const float *a_xn = xn, *a_xm = mir ? xm : nullptr, *a_act = act, *a_lp = old_logp, *a_adv = adv, *a_ret = ret;
    float *a_xb = p->xb, *a_ma = p->mb_act, *a_ml = p->mb_logp, *a_mv = p->mb_adv, *a_mr = p->mb_ret;
    void* args[16] = {&a_idx, &a_B, &a_R, &a_Dp, &a_A, &a_xn, &a_xm, &a_act, &a_lp, &a_adv, &a_ret, &a_xb, &a_ma, &a_ml, &a_mv, &a_mr};
    hipKernelNodeParams kp;
    HIPCHK(hipGraphKernelNodeGetParams(p->node_gather, &kp));
    kp.kernelParams = args; kp.extra = nullptr;
    HIPCHK(hipGraphExecKernelNodeSetParams(p->step_exec, p->node_gather, &kp));
  }
  {
    float *a_th = theta, *a_g = grad, *a_m = adam_m, *a_v = adam_v;
    size_t a_n0 = na, a_off1 = p->off_critic, a_n1 = p->lc.total;
    int a_b0 = (int)((na + 255) / 256);
    float a_gs = grad_scale, a_clip = p->grad_clip, a_lr = p->lr, a_b1 = p->beta1, a_b2 = p->beta2, a_eps = p->adam_eps, a_bc1 = bc1, a_bc2 = bc2s;
    const float* a_part = p->norm_part;
    float* a_nout = p->stats + 8;
    void* args[18] = {&a_th, &a_g, &a_m, &a_v, &a_n0, &a_off1, &a_n1, &a_b0, &a_gs, &a_part, &a_nout, &a_clip, &a_lr, &a_b1, &a_b2, &a_eps, &a_bc1, &a_bc2};
    hipKernelNodeParams kp;
    HIPCHK(hipGraphKernelNodeGetParams(p->node_adam, &kp));
    kp.kernelParams = args; kp.extra = nullptr;
    HIPCHK(hipGraphExecKernelNodeSetParams(p->step_exec, p->node_adam, &kp));
  }
  p->roll_theta = nullptr;
  HIPCHK(hipGraphLaunch(p->step_exec, s));
  return LHW_OK;
}

// =========================================================================================== recurrent (LSTM) PPO
// Gaussian_LSTM_Actor / LSTM_V (reference rl/policies/actor.py:191-286, critic.py:52-112): two stacked LSTMCells and a
// linear read-out per network; rollout = one cell step per control step with the hidden state reset at episode starts
// (rl/workers/rollout_worker.py:134-137,174-177); update = back-propagation through time over whole trajectories
// (rl/algos/ppo.py:512-533).  Where the reference pads a list of trajectories to a common length and masks the losses,
// the device keeps the rollout's time-major layout: a minibatch is a set of env columns over all T steps, the hidden and
// cell state are zeroed wherever an episode starts inside a column, and every (t, column) sample is valid -- the same
// per-trajectory computation and the same loss mean, without padding.
//
// Per cell the input and recurrent weights are stored side by side, W = [W_ih | W_hh] ([4H][K], gate order i f g o as
// in torch), so one MFMA GEMM over the concatenated input [x_t | h_{t-1}] gives the gate pre-activations; the two bias
// vectors stay separate parameters (they receive the same gradient).
struct LstmLayout {
  int D, Dp, H, O, Op, K1;
  size_t w1, bi1, bh1, w2, bi2, bh2, wo, bo, total;
};
static LstmLayout lstm_layout(int D, int H, int O) {
  LstmLayout L;
  L.D = D; L.Dp = pad4(D); L.H = H; L.O = O; L.Op = pad4(O); L.K1 = L.Dp + H;
  size_t o = 0;
  L.w1 = o; o += (size_t)4 * H * L.K1;
  L.bi1 = o; o += 4 * H;
  L.bh1 = o; o += 4 * H;
  L.w2 = o; o += (size_t)4 * H * 2 * H;
  L.bi2 = o; o += 4 * H;
  L.bh2 = o; o += 4 * H;
  L.wo = o; o += (size_t)L.Op * H;
  L.bo = o; o += L.Op;
  L.total = o;
  return L;
}

struct SeqWs {  // activations of one network over a [T][Bt] minibatch (rows r = t * Bt + b)
  float *xh1 = nullptr, *xh2 = nullptr;  // [R][K1] = [x_t | h1_{t-1}], [R][2H] = [h1_t | h2_{t-1}]
  float *g1 = nullptr, *g2 = nullptr;    // [R][4H] activated gates (overwritten by d loss / d pre-activation in the backward pass)
  float *c1 = nullptr, *c2 = nullptr;    // [R][H] cell states
  float *h2 = nullptr, *y = nullptr;     // [R][H] top hidden state, [R][Op] read-out
  float *dy = nullptr, *dh2 = nullptr;   // [R][Op], [R][H]
  float *dx2 = nullptr, *dx1h = nullptr, *dcar1 = nullptr, *dcar2 = nullptr;  // per-step scratch [Bt][2H], [Bt][H], [Bt][H] x2
  int Bt = 0;
};

struct LhwRnn {
  int device, D, A, H, learn_std, T, Bmax, Nroll, use_mirror;
  float clip, ent_coeff, mirror_coeff, grad_clip, lr, adam_eps, beta1, beta2;
  LstmLayout la, lc;
  size_t off_actor, off_std, off_critic, n_params;
  int *d_obs_src = nullptr, *d_act_src = nullptr;
  float *d_obs_sign = nullptr, *d_act_sign = nullptr;
  // rollout: per network the concatenated step inputs hold the hidden state between calls, cells in rc
  float *rxh1[2] = {nullptr, nullptr}, *rxh2[2] = {nullptr, nullptr}, *rc1[2] = {nullptr, nullptr}, *rc2[2] = {nullptr, nullptr};
  float *rg = nullptr, *rh2 = nullptr, *ry = nullptr, *rcs = nullptr;  // step scratch: gates [N][4H], top hidden [N][H], read-out [N][Op], cells [N][H]
  SeqWs wa, wc;
  unsigned char* reset = nullptr;  // [T][Bmax]
  float *mb_act = nullptr, *mb_logp = nullptr, *mb_adv = nullptr, *mb_ret = nullptr, *dstd = nullptr;
  float *stats = nullptr, *stats_part = nullptr, *norm_part = nullptr, *part = nullptr;
  std::vector<void*> allocs;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gates G [B][4H] (pre-activation, biases not yet added) -> activated in place; c, h of this step.
// h goes to dest_a (always) and dest_b (zeroed for rows whose NEXT step starts an episode: the recurrent slot).
__global__ void __launch_bounds__(256) lstm_cell_fwd_kernel(int B, int H, float* __restrict__ G, const float* __restrict__ bi,
                                                            const float* __restrict__ bh, const float* __restrict__ c_prev,
                                                            const unsigned char* __restrict__ reset_t, float* __restrict__ c_out,
                                                            float* __restrict__ dest_a, int lda, float* __restrict__ dest_b, int ldb,
                                                            const unsigned char* __restrict__ reset_next) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i - (size_t)b * H);
  float* g = G + (size_t)b * 4 * H;
  const float gi = sigmoidf_(g[j] + bi[j] + bh[j]);
  const float gf = sigmoidf_(g[H + j] + bi[H + j] + bh[H + j]);
  const float gg = tanhf(g[2 * H + j] + bi[2 * H + j] + bh[2 * H + j]);
  const float go = sigmoidf_(g[3 * H + j] + bi[3 * H + j] + bh[3 * H + j]);
  const float cp = (c_prev && !(reset_t && reset_t[b])) ? c_prev[(size_t)b * H + j] : 0.f;
  const float c = gf * cp + gi * gg;
  const float h = go * tanhf(c);
  g[j] = gi; g[H + j] = gf; g[2 * H + j] = gg; g[3 * H + j] = go;
  c_out[(size_t)b * H + j] = c;
  dest_a[(size_t)b * lda + j] = h;
  if (dest_b) dest_b[(size_t)b * ldb + j] = (reset_next && reset_next[b]) ? 0.f : h;
}

// backward of one cell step: G holds the activated gates and receives d loss / d pre-activation; dcar carries d loss / d c
// to the previous step (zero across an episode start)
__global__ void __launch_bounds__(256) lstm_cell_bwd_kernel(int B, int H, float* __restrict__ G, const float* __restrict__ c,
                                                            const float* __restrict__ c_prev, const unsigned char* __restrict__ reset_t,
                                                            const float* __restrict__ dh_a, int lda, const float* __restrict__ dh_b, int ldb,
                                                            const unsigned char* __restrict__ reset_next, float* __restrict__ dcar) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i - (size_t)b * H);
  float* g = G + (size_t)b * 4 * H;
  const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
  const bool rst = reset_t && reset_t[b];
  const float cp = (c_prev && !rst) ? c_prev[(size_t)b * H + j] : 0.f;
  float dh = dh_a[(size_t)b * lda + j];
  if (dh_b && !(reset_next && reset_next[b])) dh += dh_b[(size_t)b * ldb + j];
  const float tc = tanhf(c[(size_t)b * H + j]);
  const float dct = dh * go * (1.f - tc * tc) + dcar[(size_t)b * H + j];
  dcar[(size_t)b * H + j] = rst ? 0.f : dct * gf;
  g[j] = dct * gg * gi * (1.f - gi);
  g[H + j] = dct * cp * gf * (1.f - gf);
  g[2 * H + j] = dct * gi * (1.f - gg * gg);
  g[3 * H + j] = dh * tc * go * (1.f - go);
}

// (obs - mean) / std written into the x part of a concatenated input buffer (row stride ld)
__global__ void normalize_ld_kernel(const float* __restrict__ obs, int D, int Dp, size_t R, const float* __restrict__ mean,
                                    const float* __restrict__ stdv, float* __restrict__ out, int ld) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * (size_t)Dp) return;
  size_t r = i / Dp;
  int j = (int)(i - r * Dp);
  out[r * ld + j] = j < D ? (obs[r * D + j] - mean[j]) / stdv[j] : 0.f;
}
// zero the hidden / cell state of rows starting an episode
__global__ void rnn_reset_kernel(int N, int H, const unsigned char* __restrict__ reset, float* __restrict__ h1, int ld1,
                                 float* __restrict__ h2, int ld2, float* __restrict__ c1, float* __restrict__ c2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * H) return;
  const int b = (int)(i / H), j = (int)(i - (size_t)b * H);
  if (reset[b]) { h1[(size_t)b * ld1 + j] = 0.f; h2[(size_t)b * ld2 + j] = 0.f; c1[i] = 0.f; c2[i] = 0.f; }
}
// sequence minibatch gather: columns idx[0..B) of the time-major rollout [T][N] -> rows (t, b) of the workspaces
__global__ void seq_gather_kernel(const int* __restrict__ idx, int T, int N, int B, int Bt, int Dp, int K1, int A,
                                  const float* __restrict__ xn, const float* __restrict__ xm, const float* __restrict__ act,
                                  const float* __restrict__ logp, const float* __restrict__ adv, const float* __restrict__ ret,
                                  const unsigned char* __restrict__ done, float* __restrict__ xa, float* __restrict__ xc,
                                  float* __restrict__ mact, float* __restrict__ mlogp, float* __restrict__ madv,
                                  float* __restrict__ mret, unsigned char* __restrict__ reset_a, unsigned char* __restrict__ reset_c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)T * B * Dp) return;
  const size_t m = i / Dp;
  const int j = (int)(i - m * Dp), t = (int)(m / B), b = (int)(m - (size_t)t * B);
  const size_t src = (size_t)t * N + idx[b];
  const float v = xn[src * Dp + j];
  xa[((size_t)t * Bt + b) * K1 + j] = v;
  if (xm) xa[((size_t)t * Bt + B + b) * K1 + j] = xm[src * Dp + j];
  xc[m * K1 + j] = v;
  if (j < A) mact[m * A + j] = act[src * A + j];
  if (j == 0) {
    mlogp[m] = logp[src]; madv[m] = adv[src]; mret[m] = ret[src];
    const unsigned char r = (t == 0 || done[(size_t)(t - 1) * N + idx[b]]) ? 1 : 0;   // an episode starts at step t of this column
    reset_a[(size_t)t * Bt + b] = r;
    if (xm) reset_a[(size_t)t * Bt + B + b] = r;
    reset_c[m] = r;
  }
}

static void lstm_seq_forward(const LstmLayout& L, const float* th, SeqWs& w, int T, const unsigned char* reset, hipStream_t s) {
  const int Bt = w.Bt, H = L.H, K1 = L.K1;
  const int nb = (int)(((size_t)Bt * H + 255) / 256);
  // the recurrent slots of step 0 start from zero
  (void)hipMemset2DAsync(w.xh1 + L.Dp, sizeof(float) * K1, 0, sizeof(float) * H, Bt, s);
  (void)hipMemset2DAsync(w.xh2 + H, sizeof(float) * 2 * H, 0, sizeof(float) * H, Bt, s);
  for (int t = 0; t < T; t++) {
    const size_t r0 = (size_t)t * Bt;
    const bool last = t + 1 == T;
    GemmArgs g{};
    g.A = w.xh1 + r0 * K1; g.lda = K1; g.B = th + L.w1; g.ldb = K1; g.C = w.g1 + r0 * 4 * H; g.ldc = 4 * H; g.M = Bt; g.N = 4 * H; g.K = K1;
    launch_gemm<true, true>(g, s);
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(nb), dim3(256), 0, s, Bt, H, w.g1 + r0 * 4 * H, th + L.bi1, th + L.bh1,
                       t ? w.c1 + (r0 - Bt) * H : (const float*)nullptr, reset + r0, w.c1 + r0 * H, w.xh2 + r0 * 2 * H, 2 * H,
                       last ? (float*)nullptr : w.xh1 + (r0 + Bt) * K1 + L.Dp, K1, last ? (const unsigned char*)nullptr : reset + r0 + Bt);
    g = GemmArgs{};
    g.A = w.xh2 + r0 * 2 * H; g.lda = 2 * H; g.B = th + L.w2; g.ldb = 2 * H; g.C = w.g2 + r0 * 4 * H; g.ldc = 4 * H; g.M = Bt; g.N = 4 * H; g.K = 2 * H;
    launch_gemm<true, true>(g, s);
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(nb), dim3(256), 0, s, Bt, H, w.g2 + r0 * 4 * H, th + L.bi2, th + L.bh2,
                       t ? w.c2 + (r0 - Bt) * H : (const float*)nullptr, reset + r0, w.c2 + r0 * H, w.h2 + r0 * H, H,
                       last ? (float*)nullptr : w.xh2 + (r0 + Bt) * 2 * H + H, 2 * H, last ? (const unsigned char*)nullptr : reset + r0 + Bt);
  }
  GemmArgs g{};
  g.A = w.h2; g.lda = H; g.B = th + L.wo; g.ldb = H; g.C = w.y; g.ldc = L.Op; g.M = T * Bt; g.N = L.O; g.K = H; g.bias = th + L.bo;
  launch_gemm<true, true>(g, s);
}

// BPTT given w.dy; accumulates the parameter gradients of this network into grad (same layout as theta)
static void lstm_seq_backward(const LstmLayout& L, const float* th, float* grad, SeqWs& w, int T, const unsigned char* reset,
                              float* part, int k_chunk, hipStream_t s) {
  const int Bt = w.Bt, H = L.H, K1 = L.K1, R = T * Bt;
  const int nb = (int)(((size_t)Bt * H + 255) / 256);
  GemmArgs g{};
  g.A = w.dy; g.lda = L.Op; g.B = th + L.wo; g.ldb = H; g.C = w.dh2; g.ldc = H; g.M = R; g.N = H; g.K = L.O;
  launch_gemm<true, false>(g, s);
  (void)hipMemsetAsync(w.dcar1, 0, sizeof(float) * Bt * H, s);
  (void)hipMemsetAsync(w.dcar2, 0, sizeof(float) * Bt * H, s);
  for (int t = T - 1; t >= 0; t--) {
    const size_t r0 = (size_t)t * Bt;
    const bool last = t + 1 == T;
    const unsigned char* rnext = last ? nullptr : reset + r0 + Bt;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(nb), dim3(256), 0, s, Bt, H, w.g2 + r0 * 4 * H, w.c2 + r0 * H,
                       t ? w.c2 + (r0 - Bt) * H : (const float*)nullptr, reset + r0, w.dh2 + r0 * H, H,
                       last ? (const float*)nullptr : w.dx2 + H, 2 * H, rnext, w.dcar2);
    g = GemmArgs{};   // d [h1_t | h2_{t-1}] = dG2 W2
    g.A = w.g2 + r0 * 4 * H; g.lda = 4 * H; g.B = th + L.w2; g.ldb = 2 * H; g.C = w.dx2; g.ldc = 2 * H; g.M = Bt; g.N = 2 * H; g.K = 4 * H;
    launch_gemm<true, false>(g, s);
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(nb), dim3(256), 0, s, Bt, H, w.g1 + r0 * 4 * H, w.c1 + r0 * H,
                       t ? w.c1 + (r0 - Bt) * H : (const float*)nullptr, reset + r0, w.dx2, 2 * H,
                       last ? (const float*)nullptr : w.dx1h, H, rnext, w.dcar1);
    g = GemmArgs{};   // d h1_{t-1} = dG1 W1[:, Dp:]
    g.A = w.g1 + r0 * 4 * H; g.lda = 4 * H; g.B = th + L.w1 + L.Dp; g.ldb = K1; g.C = w.dx1h; g.ldc = H; g.M = Bt; g.N = H; g.K = 4 * H;
    launch_gemm<true, false>(g, s);
  }
  // parameter gradients: contractions over all R = T * Bt rows at once
  g = GemmArgs{};
  g.A = w.g2; g.lda = 4 * H; g.B = w.xh2; g.ldb = 2 * H; g.C = grad + L.w2; g.ldc = 2 * H; g.M = 4 * H; g.N = 2 * H; g.K = R; g.part = part; g.k_chunk = k_chunk;
  launch_gemm<false, false>(g, s);
  colsum_det(w.g2, R, 4 * H, 4 * H, grad + L.bi2, part, s);
  colsum_det(w.g2, R, 4 * H, 4 * H, grad + L.bh2, part, s);
  g = GemmArgs{};
  g.A = w.g1; g.lda = 4 * H; g.B = w.xh1; g.ldb = K1; g.C = grad + L.w1; g.ldc = K1; g.M = 4 * H; g.N = K1; g.K = R; g.part = part; g.k_chunk = k_chunk;
  launch_gemm<false, false>(g, s);
  colsum_det(w.g1, R, 4 * H, 4 * H, grad + L.bi1, part, s);
  colsum_det(w.g1, R, 4 * H, 4 * H, grad + L.bh1, part, s);
  g = GemmArgs{};
  g.A = w.dy; g.lda = L.Op; g.B = w.h2; g.ldb = H; g.C = grad + L.wo; g.ldc = H; g.M = L.O; g.N = H; g.K = R; g.part = part; g.k_chunk = k_chunk;
  launch_gemm<false, false>(g, s);
  colsum_det(w.dy, R, L.Op, L.O, grad + L.bo, part, s);
}

extern "C" int lhw_rnn_destroy(LhwRnn* p) {
  if (!p) return LHW_OK;
  (void)hipSetDevice(p->device);
  for (void* a : p->allocs) (void)hipFree(a);
  delete p;
  return LHW_OK;
}

// seq_len / seq_cols: capacity of a BPTT minibatch (time steps x env columns); rollout_rows: envs stepped per call
extern "C" int lhw_rnn_create(const LhwPpoConfig* c, int32_t seq_len, int32_t seq_cols, int32_t rollout_rows, LhwRnn** out) {
  if (!c || !out) return lhw_fail(LHW_ERR_ARG, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return lhw_fail(LHW_ERR_NO_DEVICE, "no HIP device visible: liblhw has no CPU fallback");
  if (c->obs_dim <= 0 || c->act_dim <= 0 || c->act_dim > 32 || c->hidden <= 0 || c->hidden % 4 || seq_len <= 0 || seq_cols <= 0 || rollout_rows <= 0)
    return lhw_fail(LHW_ERR_ARG, "bad recurrent PPO dimensions");
  HIPCHK(hipSetDevice(c->device));
  LhwRnn* p = new LhwRnn();
  p->device = c->device; p->D = c->obs_dim; p->A = c->act_dim; p->H = c->hidden; p->learn_std = c->learn_std;
  p->T = seq_len; p->Bmax = seq_cols; p->Nroll = rollout_rows;
  p->clip = c->clip; p->ent_coeff = c->entropy_coeff; p->mirror_coeff = c->mirror_coeff; p->grad_clip = c->max_grad_norm;
  p->lr = c->lr; p->adam_eps = c->eps; p->beta1 = 0.9f; p->beta2 = 0.999f;
  p->use_mirror = c->mirror_obs_src != nullptr;
  p->la = lstm_layout(p->D, p->H, p->A);
  p->lc = lstm_layout(p->D, p->H, 1);
  p->off_actor = 0; p->off_std = p->la.total; p->off_critic = p->off_std + pad4(p->A); p->n_params = p->off_critic + p->lc.total;
  bool ok = true;
  auto alloc = [&](auto** ptr, size_t n) {
    void* d = nullptr;
    if (!ok || lhw_malloc(&d, sizeof(**ptr) * std::max<size_t>(n, 1)) != hipSuccess || hipMemset(d, 0, sizeof(**ptr) * std::max<size_t>(n, 1)) != hipSuccess) { ok = false; return; }
    p->allocs.push_back(d);
    *ptr = (decltype(*ptr))d;
  };
  const size_t H = p->H, K1 = p->la.K1, Op = p->la.Op, N = p->Nroll;
  for (int n = 0; n < 2; n++) { alloc(&p->rxh1[n], N * K1); alloc(&p->rxh2[n], N * 2 * H); alloc(&p->rc1[n], N * H); alloc(&p->rc2[n], N * H); }
  alloc(&p->rg, N * 4 * H); alloc(&p->rh2, N * H); alloc(&p->ry, N * Op); alloc(&p->rcs, N * H);
  auto alloc_ws = [&](SeqWs& w, const LstmLayout& L, int Bt) {
    const size_t R = (size_t)p->T * Bt;
    w.Bt = Bt;
    alloc(&w.xh1, R * L.K1); alloc(&w.xh2, R * 2 * H); alloc(&w.g1, R * 4 * H); alloc(&w.g2, R * 4 * H); alloc(&w.c1, R * H); alloc(&w.c2, R * H);
    alloc(&w.h2, R * H); alloc(&w.y, R * L.Op); alloc(&w.dy, R * L.Op); alloc(&w.dh2, R * H);
    alloc(&w.dx2, (size_t)Bt * 2 * H); alloc(&w.dx1h, (size_t)Bt * H); alloc(&w.dcar1, (size_t)Bt * H); alloc(&w.dcar2, (size_t)Bt * H);
  };
  alloc_ws(p->wa, p->la, p->use_mirror ? 2 * p->Bmax : p->Bmax);
  alloc_ws(p->wc, p->lc, p->Bmax);
  const size_t Rm = (size_t)p->T * p->Bmax;
  alloc(&p->reset, 3 * Rm);   // actor rows (up to 2 Bmax per step) then critic rows
  alloc(&p->mb_act, Rm * p->A); alloc(&p->mb_logp, Rm); alloc(&p->mb_adv, Rm); alloc(&p->mb_ret, Rm); alloc(&p->dstd, Rm * Op);
  alloc(&p->stats, 16); alloc(&p->stats_part, ((Rm + 255) / 256) * NSTAT); alloc(&p->norm_part, 2 * SUMSQ_BLOCKS);
  const size_t max_slices = (2 * Rm + 2047) / 2048;
  alloc(&p->part, std::max<size_t>(max_slices * 4 * H * std::max<size_t>(K1, 2 * H), (size_t)COLSUM_CHUNKS * 4 * H));
  if (ok && p->use_mirror) {
    const size_t Dp = p->la.Dp;
    std::vector<int> osrc(Dp, 0), asrc(p->A, 0);
    std::vector<float> osgn(Dp, 0.f), asgn(p->A, 0.f);
    for (int j = 0; j < p->D; j++) { osrc[j] = c->mirror_obs_src[j]; osgn[j] = c->mirror_obs_sign[j]; if (osrc[j] < 0 || osrc[j] >= p->D) ok = false; }
    for (int j = 0; j < p->A; j++) { asrc[j] = c->mirror_act_src[j]; asgn[j] = c->mirror_act_sign[j]; if (asrc[j] < 0 || asrc[j] >= p->A) ok = false; }
    alloc(&p->d_obs_src, Dp); alloc(&p->d_act_src, p->A); alloc(&p->d_obs_sign, Dp); alloc(&p->d_act_sign, p->A);
    if (ok) {
      (void)hipMemcpy(p->d_obs_src, osrc.data(), sizeof(int) * Dp, hipMemcpyHostToDevice);
      (void)hipMemcpy(p->d_act_src, asrc.data(), sizeof(int) * p->A, hipMemcpyHostToDevice);
      (void)hipMemcpy(p->d_obs_sign, osgn.data(), sizeof(float) * Dp, hipMemcpyHostToDevice);
      (void)hipMemcpy(p->d_act_sign, asgn.data(), sizeof(float) * p->A, hipMemcpyHostToDevice);
    }
  }
  if (!ok) { lhw_rnn_destroy(p); return lhw_fail(LHW_ERR_HIP, "recurrent PPO workspace allocation failed (T=%d cols=%d) or bad mirror table", seq_len, seq_cols); }
  *out = p;
  return LHW_OK;
}

extern "C" int64_t lhw_rnn_param_count(const LhwRnn* p) { return p ? (int64_t)p->n_params : LHW_ERR_ARG; }

// offsets in the flat parameter vector: out[0..7] actor W1 (cat) b_ih1 b_hh1 W2 (cat) b_ih2 b_hh2 Wout bout; out[8] stds;
// out[9..16] critic likewise; out[17] padded obs width Dp; out[18] padded actor read-out width Op
extern "C" int lhw_rnn_layout(const LhwRnn* p, int64_t* out19) {
  if (!p || !out19) return lhw_fail(LHW_ERR_ARG, "null argument");
  const LstmLayout* Ls[2] = {&p->la, &p->lc};
  const size_t off[2] = {p->off_actor, p->off_critic};
  for (int n = 0; n < 2; n++) {
    const LstmLayout& L = *Ls[n];
    const size_t v[8] = {L.w1, L.bi1, L.bh1, L.w2, L.bi2, L.bh2, L.wo, L.bo};
    for (int k = 0; k < 8; k++) out19[n * 9 + k] = (int64_t)(off[n] + v[k]);
  }
  out19[8] = (int64_t)p->off_std;
  out19[17] = p->la.Dp; out19[18] = p->la.Op;
  return LHW_OK;
}

// One rollout step for N rows.  reset (device, [N], may be NULL): rows that start an episode with this observation.
// commit != 0 advances the stored hidden / cell state (the reference's policy(state) / critic(state) calls in
// RolloutWorker.sample); commit == 0 evaluates without touching it (value of a terminal / final observation).
extern "C" int lhw_rnn_forward(LhwRnn* p, const float* theta, const float* obs, int64_t N, const float* obs_mean, const float* obs_std,
                               const uint8_t* reset, uint64_t seed, uint32_t env_id_base, uint32_t counter, int deterministic,
                               int commit, float* mu, float* act, float* logp, float* value, void* stream) {
  if (!p || !theta || !obs || N <= 0 || N > p->Nroll) return lhw_fail(LHW_ERR_ARG, "bad argument (N=%lld, capacity %d)", (long long)N, p ? p->Nroll : 0);
  if (act && !logp) return lhw_fail(LHW_ERR_ARG, "logp required with act");
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = (hipStream_t)stream;
  const int H = p->H, K1 = p->la.K1, Dp = p->la.Dp;
  const int nb = (int)(((size_t)N * H + 255) / 256);
  const bool want[2] = {act != nullptr || mu != nullptr, value != nullptr};
  for (int n = 0; n < 2; n++) {
    if (!want[n]) continue;
    const LstmLayout& L = n ? p->lc : p->la;
    const float* th = theta + (n ? p->off_critic : p->off_actor);
    if (reset && commit)
      hipLaunchKernelGGL(rnn_reset_kernel, dim3(nb), dim3(256), 0, s, (int)N, H, reset, p->rxh1[n] + Dp, K1, p->rxh2[n] + H, 2 * H, p->rc1[n], p->rc2[n]);
    const size_t nn = (size_t)N * Dp;
    hipLaunchKernelGGL(normalize_ld_kernel, dim3((nn + 255) / 256), dim3(256), 0, s, obs, p->D, Dp, (size_t)N, obs_mean, obs_std, p->rxh1[n], K1);
    GemmArgs g{};
    g.A = p->rxh1[n]; g.lda = K1; g.B = th + L.w1; g.ldb = K1; g.C = p->rg; g.ldc = 4 * H; g.M = (int)N; g.N = 4 * H; g.K = K1;
    launch_gemm<true, true>(g, s);
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(nb), dim3(256), 0, s, (int)N, H, p->rg, th + L.bi1, th + L.bh1, (const float*)p->rc1[n],
                       (const unsigned char*)nullptr, commit ? p->rc1[n] : p->rcs, p->rxh2[n], 2 * H, commit ? p->rxh1[n] + Dp : (float*)nullptr, K1,
                       (const unsigned char*)nullptr);
    g = GemmArgs{};
    g.A = p->rxh2[n]; g.lda = 2 * H; g.B = th + L.w2; g.ldb = 2 * H; g.C = p->rg; g.ldc = 4 * H; g.M = (int)N; g.N = 4 * H; g.K = 2 * H;
    launch_gemm<true, true>(g, s);
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(nb), dim3(256), 0, s, (int)N, H, p->rg, th + L.bi2, th + L.bh2, (const float*)p->rc2[n],
                       (const unsigned char*)nullptr, commit ? p->rc2[n] : p->rcs, p->rh2, H, commit ? p->rxh2[n] + H : (float*)nullptr, 2 * H,
                       (const unsigned char*)nullptr);
    g = GemmArgs{};
    g.A = p->rh2; g.lda = H; g.B = th + L.wo; g.ldb = H; g.C = p->ry; g.ldc = L.Op; g.M = (int)N; g.N = L.O; g.K = H; g.bias = th + L.bo;
    launch_gemm<true, true>(g, s);
    if (n == 0) {
      if (mu) HIPCHK(hipMemcpy2DAsync(mu, sizeof(float) * p->A, p->ry, sizeof(float) * L.Op, sizeof(float) * p->A, N, hipMemcpyDeviceToDevice, s));
      if (act)
        hipLaunchKernelGGL(sample_kernel, dim3((N + 7) / 8), dim3(256), 0, s, p->ry, L.Op, p->A, (int)N, theta + p->off_std, seed, env_id_base,
                           counter, deterministic, act, logp);
    } else {
      HIPCHK(hipMemcpy2DAsync(value, sizeof(float), p->ry, sizeof(float) * L.Op, sizeof(float), N, hipMemcpyDeviceToDevice, s));
    }
  }
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

// BPTT over one minibatch of B env columns of the time-major rollout ([T][N] buffers; xn / xm = normalised (mirrored)
// observations [T*N][Dp], done = LHW_DONE_* flags).  Accumulates into grad and stats_dev[0..5] like lhw_ppo_grad.
extern "C" int lhw_rnn_grad(LhwRnn* p, const float* theta, float* grad, int32_t T, int32_t N, const float* xn, const float* xm,
                            const float* act, const float* old_logp, const float* adv, const float* ret, const uint8_t* done,
                            const int32_t* cols, int32_t B, float* stats_dev, void* stream) {
  if (!p || !theta || !grad || !xn || !act || !old_logp || !adv || !ret || !done || !cols || !stats_dev) return lhw_fail(LHW_ERR_ARG, "null argument");
  if (T <= 0 || T > p->T || B <= 0 || B > p->Bmax || N <= 0) return lhw_fail(LHW_ERR_ARG, "sequence minibatch %d x %d exceeds capacity %d x %d", T, B, p->T, p->Bmax);
  const int mir = p->use_mirror && xm != nullptr;
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = (hipStream_t)stream;
  const int Dp = p->la.Dp, K1 = p->la.K1, Op = p->la.Op;
  const int Bt = mir ? 2 * B : B, R = T * B;
  p->wa.Bt = Bt; p->wc.Bt = B;
  unsigned char *reset_a = p->reset, *reset_c = p->reset + (size_t)2 * p->T * p->Bmax;
  const size_t n = (size_t)R * Dp;
  hipLaunchKernelGGL(seq_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cols, T, N, B, Bt, Dp, K1, p->A, xn, mir ? xm : (const float*)nullptr,
                     act, old_logp, adv, ret, done, p->wa.xh1, p->wc.xh1, p->mb_act, p->mb_logp, p->mb_adv, p->mb_ret, reset_a, reset_c);
  const float *th_a = theta + p->off_actor, *th_c = theta + p->off_critic;
  lstm_seq_forward(p->la, th_a, p->wa, T, reset_a, s);
  lstm_seq_forward(p->lc, th_c, p->wc, T, reset_c, s);
  const int nblk = (R + 255) / 256;
  // the loss kernel writes d loss / d read-out for the normal rows (and the mirrored rows); rows it does not own stay zero
  HIPCHK(hipMemsetAsync(p->wa.dy, 0, sizeof(float) * (size_t)T * Bt * Op, s));
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(nblk), dim3(256), 0, s, R, 0, p->A, Op, p->wa.y, p->wc.y, p->mb_act, p->mb_logp, p->mb_adv,
                     p->mb_ret, theta + p->off_std, p->clip, p->mirror_coeff, mir, p->d_act_src, p->d_act_sign, p->wa.dy, p->wc.dy,
                     p->learn_std ? p->dstd : (float*)nullptr, p->stats_part, (const float*)nullptr, (const unsigned char*)nullptr, 0.f, 0.f,
                     mir ? B : 0, 1.f);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, s, p->stats_part, nblk, NSTAT, stats_dev);
  if (p->learn_std) {
    colsum_det(p->dstd, R, Op, p->A, grad + p->off_std, p->part, s);
    hipLaunchKernelGGL(entropy_grad_kernel, dim3(1), dim3(64), 0, s, theta + p->off_std, p->A, p->ent_coeff, grad + p->off_std);
  }
  const int kc = 2048;
  lstm_seq_backward(p->la, th_a, grad + p->off_actor, p->wa, T, reset_a, p->part, kc, s);
  lstm_seq_backward(p->lc, th_c, grad + p->off_critic, p->wc, T, reset_c, p->part, kc, s);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_rnn_apply(LhwRnn* p, float* theta, float* grad, float* adam_m, float* adam_v, int64_t step, float grad_scale,
                             void* stream) {
  if (!p || !theta || !grad || !adam_m || !adam_v || step <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t na = p->learn_std ? p->off_std + p->A : p->off_std;
  clip_and_adam(theta, grad, adam_m, adam_v, na, p->off_critic, p->lc.total, step, grad_scale, p->norm_part, p->stats, p->grad_clip,
                p->lr, p->beta1, p->beta2, p->adam_eps, s);
  if (!p->learn_std) HIPCHK(hipMemsetAsync(grad + p->off_std, 0, sizeof(float) * pad4(p->A), s));
  HIPCHK(hipGetLastError());
  return LHW_OK;
}

extern "C" int lhw_rnn_normalize(LhwRnn* p, const float* obs, int64_t R, const float* obs_mean, const float* obs_std, float* xn,
                                 float* xm, void* stream) {
  if (!p || !obs || !xn || R <= 0) return lhw_fail(LHW_ERR_ARG, "bad argument");
  if (xm && !p->use_mirror) return lhw_fail(LHW_ERR_ARG, "mirror output requested but no mirror tables configured");
  HIPCHK(hipSetDevice(p->device));
  size_t n = (size_t)R * p->la.Dp;
  hipLaunchKernelGGL(normalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs, p->D, p->la.Dp, (size_t)R,
                     obs_mean, obs_std, xn, xm, p->d_obs_src, p->d_obs_sign);
  HIPCHK(hipGetLastError());
  return LHW_OK;
}
