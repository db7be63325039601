// Diagnostic microbenchmark (not part of the product): why does a dependent exp() chain cost ~115 ticks at one
// wave per SIMD but ~626 at two?  Times the individual instructions of the ocml exp expansion (v_rndne_f64,
// v_cvt_i32_f64, v_ldexp_f64, v_cmp+v_cndmask, Horner chain with literal constants) and an alternative exp that
// avoids the suspects.   hipcc --offload-arch=gfx950 -O3 tools/exp_anomaly_bench.hip -o tools/bin/expbench
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline double exp_alt(double x) {
  // round-to-nearest via the 1.5*2^52 trick, 2^k by integer add into the exponent field
  const double L2E = 1.4426950408889634, C1 = 6.93147180369123816490e-01, C2 = 1.90821492927058770002e-10;
  double t = fma(x, L2E, 6755399441055744.0);
  int k = __double2loint(t);
  double kd = t - 6755399441055744.0;
  double r = fma(-kd, C1, x);
  r = fma(-kd, C2, r);
  double p = 2.08767569878680989792e-09;  // 1/12!
  p = fma(p, r, 2.50521083854417187751e-08);
  p = fma(p, r, 2.75573192239858906526e-07);
  p = fma(p, r, 2.75573192239858906526e-06);
  p = fma(p, r, 2.48015873015873015873e-05);
  p = fma(p, r, 1.98412698412698412698e-04);
  p = fma(p, r, 1.38888888888888888889e-03);
  p = fma(p, r, 8.33333333333333333333e-03);
  p = fma(p, r, 4.16666666666666666667e-02);
  p = fma(p, r, 1.66666666666666666667e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  int hi = __double2hiint(p) + (k << 20);
  return __hiloint2double(hi, __double2loint(p));
}

#define TIME(slot, body)                     \
  t0 = clock64();                            \
  for (int i = 0; i < iters; ++i) { body }   \
  t1 = clock64();                            \
  if (tid == 0) cyc[slot] = t1 - t0;

__global__ void bench(double* out, unsigned long long* cyc, int iters) {
  const int tid = threadIdx.x;
  unsigned long long t0, t1;
  double x = 1.0 + tid * 1e-3;
  int ie = 0;
  TIME(0, _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));)
  TIME(1, _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(ie));)
  TIME(2, _Pragma("unroll") for (int u = 0; u < 16; ++u) { int q; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q) : "v"(x)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(q)); })
  TIME(3, _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile("v_cmp_ngt_f64 vcc, 0, %0\n\ts_nop 0\n\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(x), "+v"(ie) : : "vcc");)
  TIME(4, _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile("v_mul_f64 %0, %0, 1.0" : "+v"(x));)
  double e = 0.001 * (tid & 7);
  TIME(5, _Pragma("unroll") for (int u = 0; u < 4; ++u) e = exp(-e);)
  double e2 = 0.001 * (tid & 7);
  TIME(6, _Pragma("unroll") for (int u = 0; u < 4; ++u) e2 = exp_alt(-e2);)
  double e3 = 0.001 * (tid & 7);
  TIME(7, _Pragma("unroll") for (int u = 0; u < 4; ++u) e3 = __expf((float)-e3);)
  double s3 = 0.001 * (tid & 7), c3 = 0;
  TIME(8, _Pragma("unroll") for (int u = 0; u < 4; ++u) { sincos(s3 + c3, &s3, &c3); })
  out[blockIdx.x * blockDim.x + tid] = x + ie + e + e2 + e3 + s3;
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 8 * 1024 * 1024);
  hipMalloc(&cyc, 8 * 16);
  const int iters = 200;
  for (int threads : {64, 256, 320, 512, 1024}) {
    hipLaunchKernelGGL(bench, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[9];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %4d: rndne %.1f | ldexp %.1f | cvt i32<->f64 pair %.1f | cmp+cndmask %.1f | mul %.1f | exp %.1f | exp_alt %.1f | expf %.1f | sincos %.1f   (ticks per dependent op)\n",
           threads, h[0] / (16.0 * iters), h[1] / (16.0 * iters), h[2] / (16.0 * iters), h[3] / (16.0 * iters), h[4] / (16.0 * iters), h[5] / (4.0 * iters),
           h[6] / (4.0 * iters), h[7] / (4.0 * iters), h[8] / (4.0 * iters));
  }
  // accuracy of exp_alt vs exp on a grid is checked on the host side of the real kernels' tests, not here
  return 0;
}
