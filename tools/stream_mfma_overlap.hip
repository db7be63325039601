// Do a wave's operand stream and its fp64 MFMAs overlap when a SIMD holds FOUR waves instead of two?  (round 6: in both matrix-core kernels the
// stream time and the MFMA time of phase V add up at two waves per SIMD -- profiles/NOTES.md.)  One workgroup per CU streams a 1.25 MB matrix that
// all workgroups share (L2 resident) in the access pattern of the tile kernel's phase V -- per batch 4 global_load_dwordx4 per lane, double
// buffered -- and issues NMF v_mfma_f64_16x16x4_f64 per batch (8 = the kernel's ratio, 0 = the stream alone; with the loads compiled out: the
// MFMAs alone).  Variants: 8 waves (512 threads) and 16 waves (1024 threads, <= 128 VGPRs) per workgroup, the same bytes and MFMAs per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/smo tools/stream_mfma_overlap.hip && /tmp/smo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
#define NROW 400
template <int NT, int NMF, bool LOADS>
__global__ __launch_bounds__(NT) void k(const double* mat, int nrep, double* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, NW = NT / 64;
  const int m = lane & 15, kk = lane >> 4;
  // a "block" = 32 columns x all rows (batches of 16 rows); blocks dealt round-robin to the waves: 12.5 blocks of 32 -> 400 / 32
  v4d acc_e = {0, 0, 0, 0}, acc_o = {0, 0, 0, 0};
  const double bq = 1.0 + lane * 1e-3;
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int rep = 0; rep < nrep; ++rep) {
    // total work per CU and rep: 12 blocks x 25 batches = 300 batches of 4 KB = 1.2 MB, split evenly over the waves
    const int nbat = 300 / NW + (wv < 300 % NW ? 1 : 0);
    const int b0 = wv * (300 / NW) + (wv < 300 % NW ? wv : 300 % NW);
    v2d A0[4], A1[4];
    auto ld = [&](v2d (&A)[4], int b) {
      const int blk = b / 25, bat = b - blk * 25;
      gptr2_t p = (gptr2_t)((const double __attribute__((address_space(1)))*)mat + (size_t)(16 * bat + kk) * NROW + 32 * blk + 2 * m);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (LOADS) A[u] = p[(size_t)u * (4 * NROW / 2)];
        else asm volatile("" : "=v"(A[u]) : "v"(p));
      }
    };
    auto mf = [&](const v2d (&A)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (2 * u < NMF) acc_e = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u].x, bq, acc_e, 0, 0, 0);
        if (2 * u + 1 < NMF) acc_o = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u].y, bq, acc_o, 0, 0, 0);
        if (NMF == 0) asm volatile("" ::"v"(A[u]));
      }
    };
    ld(A0, b0);
    int b = 0;
    for (; b + 1 < nbat; b += 2) {
      ld(A1, b0 + b + 1);
      mf(A0);
      ld(A0, b0 + (b + 2 < nbat ? b + 2 : b + 1));
      mf(A1);
    }
    if (b < nbat) mf(A0);
  }
  const unsigned long long t1 = clock64();
  out[(size_t)blockIdx.x * NT + threadIdx.x] = acc_e[0] + acc_o[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NT, int NMF, bool LOADS>
static void run(const char* name, const double* mat, double* out, unsigned long long* cyc) {
  const int nrep = 50;
  hipLaunchKernelGGL((k<NT, NMF, LOADS>), dim3(256), dim3(NT), 0, 0, mat, nrep, out, cyc);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NT, NMF, LOADS>), dim3(256), dim3(NT), 0, 0, mat, nrep, out, cyc);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0;
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double bytes = 300.0 * 4096.0 * nrep;
  printf("%-62s %8.0f cycles per 1.2 MB pass | %5.1f B/clk per CU | %.2f GHz | %.3f ms\n", name, (double)h / nrep, bytes / (double)h, (double)h / (ms * 1e-3) * 1e-9, ms);
}
int main() {
  double *mat, *out; unsigned long long* cyc;
  (void)hipMalloc(&mat, (size_t)NROW * NROW * 8 + 65536); (void)hipMemset(mat, 0, (size_t)NROW * NROW * 8 + 65536);
  (void)hipMalloc(&out, (size_t)256 * 1024 * 8); (void)hipMalloc(&cyc, 64);
  run<512, 0, true>("8 waves, stream alone", mat, out, cyc);
  run<512, 8, false>("8 waves, MFMAs alone (8 per 4 KB batch)", mat, out, cyc);
  run<512, 8, true>("8 waves, stream + MFMAs (the tile kernel's phase V)", mat, out, cyc);
  run<512, 4, true>("8 waves, stream + half the MFMAs", mat, out, cyc);
  run<1024, 0, true>("16 waves, stream alone", mat, out, cyc);
  run<1024, 8, false>("16 waves, MFMAs alone", mat, out, cyc);
  run<1024, 8, true>("16 waves, stream + MFMAs", mat, out, cyc);
  run<1024, 4, true>("16 waves, stream + half the MFMAs", mat, out, cyc);
  return 0;
}
