// Does a Kinv tile read from LDS (ds_read_b128) come back beside the global stream, or through the same return path into the VGPRs?
// Model of the lean kernel's phase V: 8 waves, each streams NG buffers of 6 tiles (1 KB per tile and wave: one dwordx4 per lane) from an
// L2-resident matrix, double buffered, two v_mfma_f64_4x4x4_4b per tile; NL further buffers per wave come from LDS instead.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_mix_bench tools/lds_mix_bench.hip && /tmp/lds_mix_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
#define NT 6
__device__ __forceinline__ void gload(v2d (&A)[NT], gptr2_t p) {
#pragma unroll
  for (int r = 0; r < NT; ++r) A[r] = p[r * 64];
}
__device__ __forceinline__ void lload(v2d (&A)[NT], const v2d* p) {
#pragma unroll
  for (int r = 0; r < NT; ++r) A[r] = p[r * 64];
}
template <bool MF>
__device__ __forceinline__ void use(const v2d (&A)[NT], double kx, double ky, double (&acc)[NT]) {
  if (!MF) {
    asm volatile("" ::"v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]));
    return;
  }
#pragma unroll
  for (int r = 0; r < NT; ++r) acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[r].x, kx, acc[r], 0, 0, 0);
#pragma unroll
  for (int r = 0; r < NT; ++r) acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[r].y, ky, acc[r], 0, 0, 0);
}
// NGH / NGL: global buffers of the three heavy / five light waves; NLH / NLL: their LDS buffers
template <bool MF>
__global__ __launch_bounds__(512) void bench(const double* tiles, int nstep, int NGH, int NGL, int NLH, int NLL, double* out, unsigned long long* cyc) {
  extern __shared__ v2d lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool heavy = wv < 3;
  const int NG = heavy ? NGH : NGL, NL = heavy ? NLH : NLL;
  // this wave's LDS buffers: NL x 6 tiles x 64 lanes
  const int lbase = (wv < 3 ? wv * NLH : 3 * NLH + (wv - 3) * NLL) * NT * 64;
  for (int i = lane; i < NL * NT * 64; i += 64) lds[lbase + i] = (v2d){1e-3 * i, 2e-3 * i};
  gptr2_t g = (gptr2_t)tiles + (size_t)wv * 24 * NT * 64 + lane;  // (each wave its own part of the matrix, as in the kernel)
  __syncthreads();
  double acc[NT];
#pragma unroll
  for (int r = 0; r < NT; ++r) acc[r] = 0.0;
  const unsigned long long t0 = clock64();
  for (int t = 0; t < nstep; ++t) {
    v2d A[NT], B[NT], L[NT];
    const double kx = 1.0 + 1e-9 * t, ky = 0.5;
    int b = 0;
    if (NG > 0) gload(A, g);
    if (NG > 1) gload(B, g + NT * 64);
    int l = 0;
    for (; b + 2 <= NG; b += 2) {
      use<MF>(A, kx, ky, acc);
      if (b + 2 < NG) gload(A, g + (size_t)(b + 2) * NT * 64);
      if (l < NL) {  // one LDS buffer per pair of global ones while there are any
        lload(L, lds + lbase + l * NT * 64 + lane);
        use<MF>(L, kx, ky, acc);
        ++l;
      }
      use<MF>(B, kx, ky, acc);
      if (b + 3 < NG) gload(B, g + (size_t)(b + 3) * NT * 64);
    }
    if (b < NG) use<MF>(A, kx, ky, acc);
    for (; l < NL; ++l) {
      lload(L, lds + lbase + l * NT * 64 + lane);
      use<MF>(L, kx, ky, acc);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < NT; ++r) s += acc[r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const size_t ndbl = (size_t)8 * 24 * NT * 128 + 4096;
  double* tiles; double* out; unsigned long long* cyc;
  (void)hipMalloc(&tiles, ndbl * 8); (void)hipMalloc(&out, 256 * 512 * 8); (void)hipMalloc(&cyc, 256 * 8);
  std::vector<double> h(ndbl);
  for (size_t i = 0; i < ndbl; ++i) h[i] = 1e-6 * (i % 1000);
  (void)hipMemcpy(tiles, h.data(), ndbl * 8, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)bench<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)bench<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nstep = 200, grid = 200;
  struct Cfg { int ngh, ngl, nlh, nll; const char* what; };
  const Cfg cfgs[] = {
      {16, 10, 0, 0, "today: heavy 19 - 3 resident = 16 global buffers, light 13 - 3 = 10"},
      {10, 10, 0, 0, "heavy waves cut to the light waves' stream (what the LDS buffers must not slow down)"},
      {10, 10, 6, 0, "heavy: 10 global + 6 from LDS (108 KB of LDS)"},
      {12, 10, 4, 0, "heavy: 12 global + 4 from LDS (72 KB)"},
      {13, 10, 3, 0, "heavy: 13 global + 3 from LDS (54 KB)"},
      {12, 9, 4, 1, "heavy 12 + 4, light 9 + 1 (102 KB)"},
      {0, 0, 6, 2, "LDS only: heavy 6, light 2"},
      {16, 16, 0, 0, "16 global buffers per wave (768 KB per step)"},
  };
  for (int mf = 0; mf < 2; ++mf)
    for (const Cfg& c : cfgs) {
      const size_t lds = (size_t)(3 * c.nlh + 5 * c.nll) * NT * 1024 + 1024;
      if (lds > 160 * 1024) continue;
      for (int rep = 0; rep < 2; ++rep) {
        if (mf) hipLaunchKernelGGL(bench<true>, dim3(grid), dim3(512), lds, 0, tiles, nstep, c.ngh, c.ngl, c.nlh, c.nll, out, cyc);
        else hipLaunchKernelGGL(bench<false>, dim3(grid), dim3(512), lds, 0, tiles, nstep, c.ngh, c.ngl, c.nlh, c.nll, out, cyc);
        (void)hipDeviceSynchronize();
      }
      std::vector<unsigned long long> hc(grid);
      (void)hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : hc) s += (double)v;
      printf("%s  G %2d/%2d  L %d/%d : %8.0f cycles per step   %s\n", mf ? "loads + MFMAs" : "loads only   ", c.ngh, c.ngl, c.nlh, c.nll, s / grid / nstep, c.what);
    }
  return 0;
}
