// Stand-alone model of phase V of the lean small-swarm kernel (v = Kinv k for P = 4 particles, one GP per workgroup) to price
// v_mfma_f64_4x4x4_4b_f64 against the v_fmac_f64_dpp form the kernel uses:
//   Kinv is stored as tiles of 16 rows x 8 columns in MFMA operand order (lane l = 16 k + 4 blk + i holds K[I0 + 4 blk + i][J0 + k]
//   and K[..][J0 + 4 + k]): one global_load_dwordx4 per lane = 1 KB contiguous per wave feeds two MFMAs; the B operand is
//   k[J0 + (l >> 4)][l & 3] from LDS (two reads per group of 8 columns, shared by the chunk's 8 row tiles).
//   mode 0: loads + MFMAs   mode 1: MFMAs only (operands stay in registers)   mode 2: loads only
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/v4_bench tools/v4_bench.hip && tools/bin/v4_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
#define NPAD 304
#define NJG (NPAD / 8)
#ifndef NRES
#define NRES 3
#endif

template <int NRT>
__device__ __forceinline__ void load_group(v2d (&A)[8], gptr2_t p) {
#pragma unroll
  for (int r = 0; r < NRT; ++r) A[r] = p[r * 64];
}
template <int NRT, int MODE>
__device__ __forceinline__ void use_group(const v2d (&A)[8], v2d kk, double (&acc)[8]) {
  if (MODE == 2) {
    asm volatile("" ::"v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(kk));
    if (NRT == 8) asm volatile("" ::"v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]));
    return;
  }
  // (the two MFMAs of a row tile depend on each other through its accumulator: all first halves, then all second halves)
  // in place (inline asm): the builtin form makes the compiler rename the accumulators and copy them back with v_mov_b64 +
  // wait states between the groups, which halves the issue rate
#pragma unroll
  for (int r = 0; r < NRT; ++r) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(A[r].x), "v"(kk.x));
#pragma unroll
  for (int r = 0; r < NRT; ++r) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(A[r].y), "v"(kk.y));
}

// one wave's share: column groups [g0, g1) of a chunk with NRT row tiles; tiles of a group are contiguous (NRT KB)
template <int NRT, int MODE>
__device__ __forceinline__ void stream(gptr2_t base, int g0, int g1, const double* kb, int lane, double (&acc)[8], const v2d (&res)[NRES > 0 ? NRES : 1][8]) {
  const int nres = min(NRES, g1 - g0);
  const int kofs = (lane >> 4) * 4 + (lane & 3);
  v2d A[8], B[8], kA, kB;
  int g = g0 + nres;
  gptr2_t p = base + (size_t)g * NRT * 64 + lane;
  const int n = g1 - g;
  if (MODE != 1) {
    if (n > 0) load_group<NRT>(A, p);
    if (n > 1) load_group<NRT>(B, p + NRT * 64);
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) A[r] = B[r] = (v2d){1.0 + lane, 0.5 * lane};
  }
  p += 2 * NRT * 64;
  if (n > 0) kA = (v2d){kb[g * 32 + kofs], kb[g * 32 + 16 + kofs]};
  if (n > 1) kB = (v2d){kb[(g + 1) * 32 + kofs], kb[(g + 1) * 32 + 16 + kofs]};
#pragma unroll
  for (int r = 0; r < NRES; ++r)
    if (r < nres) {
      const v2d kR = (v2d){kb[(g0 + r) * 32 + kofs], kb[(g0 + r) * 32 + 16 + kofs]};
      use_group<NRT, MODE>(res[NRES > 0 ? r : 0], kR, acc);
    }
#ifdef KPRE
  v2d kq[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) kq[i] = (v2d){kb[min(g + i, g1 - 1) * 32 + kofs], kb[min(g + i, g1 - 1) * 32 + 16 + kofs]};
  asm volatile("" : "+v"(kq[0]), "+v"(kq[1]), "+v"(kq[2]), "+v"(kq[3]), "+v"(kq[4]), "+v"(kq[5]), "+v"(kq[6]), "+v"(kq[7]), "+v"(kq[8]), "+v"(kq[9]));
#pragma unroll
  for (int i = 0; i < 10; i += 2) {
    if (i < n) {
      use_group<NRT, MODE>(A, kq[i], acc);
      if (i + 2 < n) {
        if (MODE != 1) load_group<NRT>(A, p);
        p += NRT * 64;
      }
    }
    if (i + 1 < n) {
      use_group<NRT, MODE>(B, kq[i + 1], acc);
      if (i + 3 < n) {
        if (MODE != 1) load_group<NRT>(B, p);
        p += NRT * 64;
      }
    }
  }
#else
  for (int i = 0; i < n; i += 2) {
    use_group<NRT, MODE>(A, kA, acc);
    if (i + 2 < n) {
      if (MODE != 1) load_group<NRT>(A, p);
      kA = (v2d){kb[(g + 2) * 32 + kofs], kb[(g + 2) * 32 + 16 + kofs]};
      p += NRT * 64;
    }
    ++g;
    if (i + 1 < n) {
      use_group<NRT, MODE>(B, kB, acc);
      if (i + 3 < n) {
        if (MODE != 1) load_group<NRT>(B, p);
        kB = (v2d){kb[(g + 2) * 32 + kofs], kb[(g + 2) * 32 + 16 + kofs]};
        p += NRT * 64;
      }
      ++g;
    }
  }
#endif
}

template <int MODE>
__global__ __launch_bounds__(512) void vbench(const double* tiles, int nstep, double* out, unsigned long long* cyc) {
  __shared__ double kb[NPAD * 4];
  __shared__ double part[8][128 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < NPAD * 4; i += 512) kb[i] = 1e-3 * (i % 17) + 1e-4 * blockIdx.x;
  // chunk 0: waves 0-2, chunk 1: waves 3-5 (8 row tiles, 38 groups: 13/13/12), chunk 2: waves 6-7 (3 row tiles, 19/19)
  const int c = wv < 3 ? 0 : (wv < 6 ? 1 : 2);
  const int wi = wv - (c == 0 ? 0 : (c == 1 ? 3 : 6));
  const int g0 = c < 2 ? (NJG * wi) / 3 : (NJG * wi) / 2, g1 = c < 2 ? (NJG * (wi + 1)) / 3 : (NJG * (wi + 1)) / 2;
  gptr2_t base = (gptr2_t)(tiles + (size_t)c * 128 * NPAD);
  v2d res[NRES > 0 ? NRES : 1][8];
#pragma unroll
  for (int r = 0; r < NRES; ++r) {
    if (c < 2) load_group<8>(res[NRES > 0 ? r : 0], base + (size_t)(g0 + r) * 8 * 64 + lane);
    else load_group<3>(res[NRES > 0 ? r : 0], base + (size_t)(g0 + r) * 3 * 64 + lane);
  }
  __syncthreads();
  double tot = 0.0;
  const unsigned long long t0 = clock64();
  for (int t = 0; t < nstep; ++t) {
    double acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.0;
    if (c < 2) stream<8, MODE>(base, g0, g1, kb, lane, acc, res);
    else stream<3, MODE>(base, g0, g1, kb, lane, acc, res);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (MFMA results read by the stores below: the hazard is ours to cover)
    // D lane = 16 i + 4 blk + p: row 16 rt + 4 blk + i of the chunk, particle p
#pragma unroll
    for (int r = 0; r < 8; ++r) part[wv][(16 * r + 4 * ((lane >> 2) & 3) + (lane >> 4)) * 4 + (lane & 3)] = acc[r];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < NPAD * 4 / 8) {  // a little feedback so that nothing is hoisted: k depends on the previous step's sums
      double s = part[0][tid] + part[3][tid] + part[6][tid & 127];
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
  std::vector<double> h((size_t)NPAD * NPAD);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)(i % 1013) - 0.5;
  double *tiles, *out;
  unsigned long long* cyc;
  hipMalloc(&tiles, h.size() * 8);
  hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&out, nwg * 512 * 8);
  hipMalloc(&cyc, 8);
  const char* names[3] = {"loads + 4x4x4 MFMAs", "4x4x4 MFMAs only", "loads only"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(vbench<0>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      if (mode == 1) hipLaunchKernelGGL(vbench<1>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      if (mode == 2) hipLaunchKernelGGL(vbench<2>, dim3(nwg), dim3(512), 0, 0, tiles, nstep, out, cyc);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc = 0;
      hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
      if (rep) printf("NRES %d  %-24s %8.0f cycles per step (workgroup 0), kernel %.3f ms for %d steps, %d workgroups\n", NRES, names[mode], (double)hc / nstep, ms, nstep, nwg);
    }
  return 0;
}
