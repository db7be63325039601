// Do the fp64 matrix pipe and the fp64 vector pipe of a SIMD run side by side?  512-thread workgroups (2 waves per SIMD: waves w
// and w + 4 share a SIMD).  Mode 0: all 8 waves issue v_mfma_f64_16x16x4_f64; mode 1: all 8 issue v_fmac_f64_dpp (row_newbcast);
// mode 2: waves 0-3 MFMA, waves 4-7 DPP-FMA; mode 3: waves 0-3 MFMA, 4-7 idle; mode 4: waves 0-3 idle, 4-7 DPP-FMA; mode 5: plain
// v_fma_f64 on all 8.  Work per wave: NIT x (16 MFMAs | 256 FMAs) = the same flops.  Prints cycles per wave (s_memtime) and the flop rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void fmac_bcast(double& acc, double bc, double own) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(own), "n"(N));
}
__global__ __launch_bounds__(512) void k(int mode, int nit, double* out, unsigned long long* cyc) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mf = mode == 0 || ((mode == 2 || mode == 3) && wv < 4);
  const bool va = mode == 1 || ((mode == 2 || mode == 4) && wv >= 4);
  const bool pl = mode == 5;
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  double x = 1.0 + lane * 1e-3, y = 0.5 + lane * 1e-4;
  __syncthreads();
  unsigned long long t0 = clock64();
  if (mf) {
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a3, 0, 0, 0);
      }
    }
  } else if (va) {
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        fmac_bcast<0>(acc[0], x, y); fmac_bcast<1>(acc[1], x, y); fmac_bcast<2>(acc[2], x, y); fmac_bcast<3>(acc[3], x, y);
        fmac_bcast<4>(acc[4], x, y); fmac_bcast<5>(acc[5], x, y); fmac_bcast<6>(acc[6], x, y); fmac_bcast<7>(acc[7], x, y);
        fmac_bcast<8>(acc[8], x, y); fmac_bcast<9>(acc[9], x, y); fmac_bcast<10>(acc[10], x, y); fmac_bcast<11>(acc[11], x, y);
        fmac_bcast<12>(acc[12], x, y); fmac_bcast<13>(acc[13], x, y); fmac_bcast<14>(acc[14], x, y); fmac_bcast<15>(acc[15], x, y);
      }
    }
  } else if (pl) {
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
      }
    }
  }
  unsigned long long t1 = clock64();
  double s = a0[0] + a1[1] + a2[2] + a3[3];
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[wv] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 512 * 8); hipMalloc(&cyc, 64);
  const int nit = 2000;
  for (int grid : {1, 256}) {
    for (int mode = 0; mode < 6; ++mode) {
      hipMemset(cyc, 0, 64);
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, mode, nit, out, cyc);
      hipDeviceSynchronize();
      unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      printf("grid %3d mode %d: ticks/wave:", grid, mode);
      for (int w = 0; w < 8; ++w) printf(" %llu", h[w]);
      // flops per wave = nit * 16 MFMA * 2048 = nit * 256 FMA * 128
      double mx = 0; for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
      int active = (mode == 0 || mode == 1 || mode == 5) ? 8 : (mode == 2 ? 8 : 4);
      printf("  | per-CU flop/tick %.1f (peak 128)\n", active * (double)nit * 16 * 2048 / mx);
    }
  }
  return 0;
}
