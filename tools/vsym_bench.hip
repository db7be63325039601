// Stand-alone model of a SYMMETRIC phase V of the lean small-swarm kernel (round 5 experiment): stream only the upper triangle of Kinv
// (tiles of 16 rows x 8 columns in MFMA operand order) and use every strictly-upper tile twice --
//   direct      v[rows of the tile]    += K_tile   k[its 8 columns]          (2 x v_mfma_f64_4x4x4_4b, as today)
//   transposed  v[its 8 columns]       += K_tile^T k[rows of the tile]       (the tile transposed between the lane bit pairs (5:4) <-> (1:0) by
//                                                                             4 ds_bpermute_b32, 2 more MFMAs, partial sums over the 4 row blocks
//                                                                             by DPP, 16 lanes store them to an LDS slot per column group)
// against the full stream (tools/v4_bench.hip: loads only 12.7 k, MFMAs only 8.4 k, together 11.4 - 14.0 k cycles per step at N = 304).
// The model keeps the kernel's geometry: 8 waves, buffers of 6 tiles (2 column groups x 3 row tiles), double buffered, NB buffers per wave
// (the triangle + the bands' diagonal squares: ~420 of 722 tiles -> 9 buffers per wave), the first NDIAG of them direct-only.
//   mode 0: everything   mode 1: no loads (operands stay in registers)   mode 2: loads only   mode 3: no transposed half (direct on half the tiles)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/vsym_bench tools/vsym_bench.hip && tools/bin/vsym_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
#ifndef NB
#define NB 9
#endif
#ifndef NDIAG
#define NDIAG 1
#endif
#define NPAD 304

__device__ __forceinline__ void mfma4(double& acc, double a, double b) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ double bperm(int addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_shr(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__device__ __forceinline__ void load6(v2d (&A)[6], gptr2_t p) {
  if (MODE == 1) return;
#pragma unroll
  for (int s = 0; s < 6; ++s) A[s] = p[s * 64];
}
template <int MODE>
__device__ __forceinline__ void use6(const v2d (&A)[6], const v2d (&K)[2], const double (&kR)[3], double (&acc)[2][3], bool transposed, int paddr, double* slot, int lane) {
  asm volatile("" ::"v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(K[0]), "v"(K[1]));
  if (MODE == 2) return;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int r = 0; r < 3; ++r) mfma4(acc[0][r], A[3 * q + r].x, K[q].x);
#pragma unroll
    for (int r = 0; r < 3; ++r) mfma4(acc[1][r], A[3 * q + r].y, K[q].y);
  }
  if (transposed && MODE != 3) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      double tx = 0.0, ty = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double ax = bperm(paddr, A[3 * q + r].x), ay = bperm(paddr, A[3 * q + r].y);
        mfma4(tx, ax, kR[r]);
        mfma4(ty, ay, kR[r]);
      }
      tx += dpp_shr<0x114>(tx);  // row_shr:4
      ty += dpp_shr<0x114>(ty);
      tx += dpp_shr<0x118>(tx);  // row_shr:8
      ty += dpp_shr<0x118>(ty);
      if ((lane & 12) == 12) {  // D lane = 16 i + 4 blk + p: the block sum sits in blk = 3; element (column i [+ 4], particle p)
        v2d w;
        w.x = tx;
        w.y = ty;
        *reinterpret_cast<v2d*>(slot + q * 32 + ((lane >> 4) * 4 + (lane & 3)) * 2) = w;
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void vbench(const double* tiles, int nstep, double* out, unsigned long long* cyc) {
  __shared__ double kb[(NPAD + 32) * 4];
  __shared__ __attribute__((aligned(16))) double slots[8][NB][64];
  __shared__ double part[8][64 * 3];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (NPAD + 32) * 4; i += 512) kb[i] = 1e-3 * (i % 17) + 1e-4 * blockIdx.x;
  gptr2_t base = (gptr2_t)(tiles + (size_t)wv * NB * 6 * 128) + lane;
  const int paddr = 4 * (16 * (lane & 3) + (lane & 12) + (lane >> 4));  // lane (k, blk, e) takes from lane (e, blk, k)
  const double* ka = kb + (lane >> 4) * 4 + (lane & 3);
  __syncthreads();
  double tot = 0.0;
  const unsigned long long t0 = clock64();
  for (int t = 0; t < nstep; ++t) {
    double acc[2][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[0][r] = acc[1][r] = 0.0;
    double kR[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) kR[r] = kb[(16 * (3 * wv + r) + (lane & 12) + (lane >> 4)) * 4 + (lane & 3)];
    v2d bufA[6], bufB[6], kA[2], kBq[2];
    if (MODE == 1) {
#pragma unroll
      for (int s = 0; s < 6; ++s) bufA[s] = bufB[s] = (v2d){1.0 + lane, 0.5 * lane};
    }
    load6<MODE>(bufA, base);
    load6<MODE>(bufB, base + 6 * 64);
    auto readk = [&](v2d(&K)[2], int b) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        K[q].x = ka[(b * 2 + q) * 32];
        K[q].y = ka[(b * 2 + q) * 32 + 16];
      }
    };
    readk(kA, 0);
    readk(kBq, 1);
    int b = 0;
    for (; b + 2 < NB - 1; b += 2) {
      use6<MODE>(bufA, kA, kR, acc, b >= NDIAG, paddr, &slots[wv][b][0], lane);
      load6<MODE>(bufA, base + (size_t)(b + 2) * 6 * 64);
      readk(kA, b + 2);
      use6<MODE>(bufB, kBq, kR, acc, b + 1 >= NDIAG, paddr, &slots[wv][b + 1][0], lane);
      load6<MODE>(bufB, base + (size_t)(b + 3) * 6 * 64);
      readk(kBq, b + 3);
    }
    use6<MODE>(bufA, kA, kR, acc, true, paddr, &slots[wv][b][0], lane);
    if (b + 1 < NB) use6<MODE>(bufB, kBq, kR, acc, true, paddr, &slots[wv][b + 1][0], lane);
    if (b + 2 < NB) {  // (odd NB: one more)
      load6<MODE>(bufA, base + (size_t)(b + 2) * 6 * 64);
      readk(kA, b + 2);
      use6<MODE>(bufA, kA, kR, acc, true, paddr, &slots[wv][b + 2][0], lane);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) part[wv][r * 64 + lane] = acc[0][r] + acc[1][r];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // the owner's sum: direct + the other waves' transposed partials (a few slot reads per lane), then feedback into k
    if (tid < NPAD) {
      double s = part[tid & 7][tid & 127];
#pragma unroll
      for (int w = 0; w < 7; ++w) s += slots[w][(tid >> 6) % NB][tid & 63];
      kb[tid] = 1e-3 + 1e-9 * s;
      tot += s;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 512 + tid] = tot;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int nwg = 200, nstep = 150;
  std::vector<double> h((size_t)8 * (NB + 2) * 6 * 128);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)(i % 1013) - 0.5;
  double *tiles, *out;
  unsigned long long* cyc;
  hipMalloc(&tiles, h.size() * 8);
  hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&out, nwg * 512 * 8);
  hipMalloc(&cyc, 8);
  const char* names[4] = {"loads + direct + transposed", "no loads", "loads only", "loads + direct only"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(vbench<0>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      if (mode == 1) hipLaunchKernelGGL(vbench<1>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      if (mode == 2) hipLaunchKernelGGL(vbench<2>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      if (mode == 3) hipLaunchKernelGGL(vbench<3>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc = 0;
      hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
      if (rep) printf("NB %d  %-30s %8.0f cycles per step (workgroup 0), kernel %.3f ms for %d steps, %d workgroups\n", NB, names[mode], (double)hc / nstep, ms, nstep, nwg);
    }
  return 0;
}
