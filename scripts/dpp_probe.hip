// Hardware check of the cross-lane primitives the chain solver / chain dynamics rely on (run on the GPU box):
// v_mov_b64_dpp row_newbcast:N, v_permlane16_swap, v_permlane32_swap, row_shr / row_shl scans on doubles -- against the semantics
// the SIMT emulator of tests/emu implements.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SRC>
__device__ double rbc(double v) {
  const long long lv = __double_as_longlong(v);
  return __longlong_as_double(__builtin_amdgcn_update_dpp(lv, lv, 0x150 + SRC, 0xf, 0xf, false));
}
__device__ double xhalf_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ double x32_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
template <int CTRL>
__device__ double dpp_row(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
  const int l = threadIdx.x;
  const double v = 1000.0 + l;
  out[l] = rbc<3>(v);
  out[64 + l] = rbc<11>(v);
  out[128 + l] = xhalf_sum(v);
  double w = v;
  if ((l & 15) < 6) w = xhalf_sum(v);   // under a partial exec mask
  out[192 + l] = w;
  double p = v, s = v;
  p += dpp_row<0x111>(p); p += dpp_row<0x112>(p); p += dpp_row<0x114>(p); p += dpp_row<0x118>(p);
  s += dpp_row<0x101>(s); s += dpp_row<0x102>(s); s += dpp_row<0x104>(s); s += dpp_row<0x108>(s);
  out[256 + l] = p;
  out[320 + l] = s;
  out[384 + l] = x32_sum(v);
}
int main() {
  double* d;
  (void)hipMalloc(&d, 448 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[448];
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const int row = l & ~15;
    const double other = 1000.0 + (l ^ 16), own = 1000.0 + l;
    if (h[l] != 1000.0 + row + 3) bad++;
    if (h[64 + l] != 1000.0 + row + 11) bad++;
    if (h[128 + l] != own + other) bad++;
    if (h[192 + l] != (((l & 15) < 6) ? own + other : own)) bad++;
    double pre = 0, suf = 0;
    for (int j = row; j <= l; j++) pre += 1000.0 + j;
    for (int j = l; j < row + 16; j++) suf += 1000.0 + j;
    if (h[256 + l] != pre) { bad++; printf("prefix lane %d: %g vs %g\n", l, h[256 + l], pre); }
    if (h[320 + l] != suf) { bad++; printf("suffix lane %d: %g vs %g\n", l, h[320 + l], suf); }
    if (h[384 + l] != own + 1000.0 + (l ^ 32)) bad++;
  }
  printf("dpp_probe: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
  return bad != 0;
}
