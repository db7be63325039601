// Issue cost of v_fmac_f64 with and without a DPP row_newbcast operand (one wave, 16 independent accumulators).
//   hipcc --offload-arch=gfx950 -O3 -o dpp_fma_bench tools/dpp_fma_bench.hip && ./dpp_fma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ void k(double* out, long long* cyc, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  double kv = a + threadIdx.x, av = b;
  long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(kv), "v"(av));
  }
  long long t1 = clock64();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(kv), "v"(av));
  }
  long long t2 = clock64();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  double* o; long long* c;
  hipMalloc(&o, 1 << 20); hipMalloc(&c, 64);
  for (int waves = 1; waves <= 8; waves *= 2) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, o, c, 1.0, 1e-9);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("waves/WG %d: plain %.2f cycles per FMA, dpp %.2f cycles per FMA (wave 0's clock, %d FMAs)\n", waves, (double)h[0] / (REP * 16), (double)h[1] / (REP * 16), REP * 16);
  }
  return 0;
}
