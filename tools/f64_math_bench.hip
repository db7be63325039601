// Dependent-chain cost (ticks per call, one wave, s_memtime) of the fp64 library functions on the critical path of the per-step phases:
// exp, tanh, sincos, sqrt, division, and cheaper formulations of tanh / reciprocal.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(int n, double x0, double* out, unsigned long long* cyc) {
  double x = x0 + threadIdx.x * 1e-6, acc = 0.0;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    double y;
    if (OP == 0) y = exp(-x);
    if (OP == 1) y = tanh(x);
    if (OP == 2) { double s, c; sincos(x, &s, &c); y = s + c; }
    if (OP == 3) y = sqrt(x);
    if (OP == 4) y = 1.25 / x;
    if (OP == 5) { double e = exp(-2.0 * fabs(x)); y = copysign((1.0 - e) / (1.0 + e), x); }  // tanh via exp + one division
    if (OP == 6) y = __drcp_rn(x);
    if (OP == 7) y = 10.0 * tanh(x / 10.0);
    acc += y;
    x = x * 0.999 + y * 1e-3;  // keeps the chain dependent and the argument in a sane range
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = acc + x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc; unsigned long long h;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
  const char* names[] = {"exp(-x)", "tanh(x)", "sincos(x)", "sqrt(x)", "1.25/x", "tanh via exp+div", "__drcp_rn", "10*tanh(x/10)"};
  const int n = 4000;
#define RUN(OP) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, n, 0.7, out, cyc); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-18s %7.1f ticks per call (incl. ~14 of loop overhead)\n", names[OP], (double)h / n);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  return 0;
}
