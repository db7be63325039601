// Diagnostic microbenchmark (not part of the product): cycles per instruction of the fp64 building blocks on one
// wave / several waves per SIMD -- v_fma_f64 (dependent and independent), v_mfma_f64_16x16x4_f64 (dependent chain and
// 4 independent accumulators), exp(), LDS read round trip.  s_memtime around unrolled loops; lane 0 of wave 0 reports.
//   hipcc --offload-arch=gfx950 -O3 tools/f64_rate_bench.hip -o tools/bin/f64bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void bench(double* out, unsigned long long* cyc, int iters) {
  __shared__ double lds[1024];
  const int tid = threadIdx.x;
  lds[tid & 1023] = tid * 0.001;
  __syncthreads();
  double x = 1.0 + tid * 1e-9, y = 0.5, z = 0.25;
  unsigned long long t0, t1;
  // 1. dependent fma chain
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) x = fma(x, y, z);
  }
  t1 = clock64();
  if (tid == 0) cyc[0] = t1 - t0;
  // 2. 8 independent fma chains
  double a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = x + k;
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = fma(a[k], y, z);
  }
  t1 = clock64();
  if (tid == 0) cyc[1] = t1 - t0;
#pragma unroll
  for (int k = 0; k < 8; ++k) x += a[k];
  // 3. dependent mfma chain
  v4d acc = {0, 0, 0, 0};
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
  }
  t1 = clock64();
  if (tid == 0) cyc[2] = t1 - t0;
  // 4. 4 independent mfma accumulators
  v4d ac[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) ac[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, ac[k], 0, 0, 0);
  }
  t1 = clock64();
  if (tid == 0) cyc[3] = t1 - t0;
  // 5. dependent exp chain
  double e = 0.001 * (tid & 7);
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) e = exp(-e);
  }
  t1 = clock64();
  if (tid == 0) cyc[4] = t1 - t0;
  // 6. dependent LDS read chain (pointer chasing through indices)
  int idx = tid & 1023;
  t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) idx = ((int)(lds[idx] * 1000.0) + 1) & 1023;
  }
  t1 = clock64();
  if (tid == 0) cyc[5] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = x + acc[0] + ac[0][0] + ac[1][1] + ac[2][2] + ac[3][3] + e + idx;
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 8 * 1024 * 1024);
  hipMalloc(&cyc, 8 * 16);
  const int iters = 200;
  for (int threads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(bench, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[6];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %4d (%d waves/SIMD): dep fma %.1f cyc | indep fma %.1f cyc/instr | dep mfma f64 16x16x4 %.1f cyc | indep mfma %.1f cyc/instr | dep exp %.1f cyc | "
           "dep LDS read(+cvt) %.1f cyc\n",
           threads, (threads + 255) / 256, h[0] / (16.0 * iters), h[1] / (16.0 * iters), h[2] / (16.0 * iters), h[3] / (16.0 * iters), h[4] / (4.0 * iters),
           h[5] / (16.0 * iters));
  }
  return 0;
}
