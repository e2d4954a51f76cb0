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
// round 6: v_fmac_f64 with a DPP (row_newbcast) source, as csrc/lhw_humanoid_dev.h writes it out (fma_rbc / fmac_col / fmac_mrow)
template <int SRC>
__device__ double fma_rbc(double v, double c, double acc) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %[acc], %[v], %[c] row_newbcast:%[k] row_mask:0xf bank_mask:0xf" : [acc] "+v"(acc) : [v] "v"(v), [c] "v"(c), [k] "n"(SRC));
  return acc;
}
__global__ void k2(double* out) {
  const int l = threadIdx.x;
  const double v = 1.0 + l / 64.0, c = 3.0 - l / 32.0, a = 0.5 + l / 16.0;   // (dyadic: every product below is exact, so a mismatch is a wrong lane, not a rounding)
  out[l] = fma_rbc<3>(v, c, a);                 // a + v[row + 3] * c
  out[64 + l] = fma_rbc<11>(v, c, a);
  double x = v;                                 // in place: x += x[row + 5] * c  (every lane reads the OLD x of lane 5 of its row)
  asm("s_nop 1\n\tv_fmac_f64_dpp %[x], %[x], %[c] row_newbcast:5 row_mask:0xf bank_mask:0xf" : [x] "+v"(x) : [c] "v"(c));
  out[128 + l] = x;
  double y = v, z = a;                          // two accumulators in one statement, then a dependent chain (x read through DPP right after it was written)
  asm("s_nop 1\n\tv_fmac_f64_dpp %[y], %[y], %[c] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %[z], %[z], %[c] row_newbcast:2 row_mask:0xf bank_mask:0xf" : [y] "+v"(y), [z] "+v"(z) : [c] "v"(c));
  out[192 + l] = y; out[256 + l] = z;
  double w = v;
  w = fma_rbc<0>(w, c, w); w = fma_rbc<1>(w, c, w); w = fma_rbc<2>(w, c, w);
  out[320 + l] = w;
  double u = a;                                 // under a partial exec mask (the upper 32 lanes idle, as when one env of a wave has left a loop)
  if (l < 32) u = fma_rbc<7>(v, c, a);
  out[384 + l] = u;
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
  {
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad2 = 0;
    auto V = [](int l) { return 1.0 + l / 64.0; };
    auto C = [](int l) { return 3.0 - l / 32.0; };
    auto A = [](int l) { return 0.5 + l / 16.0; };
    for (int l = 0; l < 64; l++) {
      const int row = l & ~15;
      auto chk = [&](const char* what, double got, double want) { if (got != want) { bad2++; printf("fmac_dpp %s lane %d: %.17g vs %.17g\n", what, l, got, want); } };
      chk("bcast3", h[l], fma(V(row + 3), C(l), A(l)));
      chk("bcast11", h[64 + l], fma(V(row + 11), C(l), A(l)));
      chk("inplace", h[128 + l], fma(V(row + 5), C(l), V(l)));
      chk("pair y", h[192 + l], fma(V(row + 2), C(l), V(l)));
      chk("pair z", h[256 + l], fma(A(row + 2), C(l), A(l)));
      double w[16];
      for (int j = 0; j < 16; j++) w[j] = V(row + j);
      for (int e = 0; e < 3; e++) { double we = w[e]; for (int j = 0; j < 16; j++) w[j] = fma(we, C(row + j), w[j]); }
      chk("chain", h[320 + l], w[l & 15]);
      chk("partial exec", h[384 + l], l < 32 ? fma(V(row + 7), C(l), A(l)) : A(l));
    }
    printf("dpp_probe v_fmac_f64_dpp: %s (%d mismatches)\n", bad2 ? "FAIL" : "ok", bad2);
    bad += bad2;
  }
  return bad != 0;
}
