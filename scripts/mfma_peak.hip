// Practical ceiling of v_mfma_f32_32x32x2_f32 on this box: waves that do nothing but MFMAs on independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o scripts/_bin/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char* tag) {
  float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * NACC * 4096.0;
  printf("%s: blocks %d (x4 waves) iters %d nacc %d: %.3f ms  %.1f TFLOP/s\n", tag, blocks, iters, NACC, ms, flop / ms / 1e9);
  hipFree(d);
}
int main() {
  run<4>(256 * 2, 4000, "2 waves/SIMD");
  run<4>(256 * 1, 4000, "1 wave/SIMD ");
  run<4>(256 * 4, 4000, "4 waves/SIMD");
  run<1>(256 * 2, 16000, "2 waves/SIMD, one dependent chain each");
  run<4>(256 * 2, 40000, "2 waves/SIMD, 10x longer");
  return 0;
}
