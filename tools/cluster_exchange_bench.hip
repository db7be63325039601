// What would the all-gather of k cost in an N+G-sharded cluster?  Clusters of CS workgroups; every step each
// workgroup publishes its block of k (ROWS x PART doubles) and needs the blocks of the CS - 1 others before it can go on.
//   mode 0: payload by plain stores, then a release fence and ONE flag per workgroup and step; readers poll the partners' flags
//           (acquire), then read the payload with agent-scope loads (form R1 of cdna_hip_programming.md, Guideline 16).
//   mode 1: no flags: every double travels as two 8-byte {tag, half} granules (the form the GP-sharded rollout uses for its
//           P doubles per step), readers re-read until every tag matches.
// All workgroups are resident (clusters * CS <= CUs).  Prints the cycles per step of wave 0 (s_memtime), i.e. publish + wait + read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef u64 __attribute__((address_space(1))) * gu64_t;
#define SPIN (1u << 22)
__global__ __launch_bounds__(512) void xbench(int mode, int CS, int nd /* doubles per block */, int steps, double* pay, u64* gran, unsigned* flag,
                                              double* out, u64* cyc, unsigned* fail) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cluster = blockIdx.x / CS, me = blockIdx.x % CS;
  double acc = 0.0;
  __shared__ volatile int abortw;
  if (tid == 0) abortw = 0;
  __syncthreads();
  u64 t0 = clock64();
  for (int t = 0; t < steps && !abortw; ++t) {
    const int par = t & 1;
    if (mode == 0) {
      double* mine = pay + ((size_t)(cluster * 2 + par) * CS + me) * nd;
      for (int e = tid; e < nd; e += 512) mine[e] = 1e-3 * e + t;
      __threadfence();  // release at agent scope
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flag[(cluster * 2 + par) * CS + me], (unsigned)t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (wv == 0) {  // one wave polls the partners' flags
        bool done = false;
        for (unsigned s = 0; s < SPIN; ++s) {
          bool ok = true;
          if (lane < CS && lane != me) ok = __hip_atomic_load(&flag[(cluster * 2 + par) * CS + lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)t + 1u;
          if (__all(ok)) { done = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!done && lane == 0) abortw = 1;
      }
      __syncthreads();
      if (!abortw)
        for (int p = 0; p < CS; ++p) {
          if (p == me) continue;
          const double* theirs = pay + ((size_t)(cluster * 2 + par) * CS + p) * nd;
          for (int e = tid; e < nd; e += 512) acc += __hip_atomic_load(&theirs[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
      gu64_t mine = (gu64_t)gran + ((size_t)(cluster * 2 + par) * CS + me) * nd * 2;
      for (int e = tid; e < nd; e += 512) {
        const u64 bits = (u64)__double_as_longlong(1e-3 * e + t);
        __hip_atomic_store(mine + 2 * e, ((u64)(t + 1) << 32) | (unsigned)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2 * e + 1, ((u64)(t + 1) << 32) | (unsigned)(bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int p = 0; p < CS && !abortw; ++p) {
        if (p == me) continue;
        gu64_t theirs = (gu64_t)gran + ((size_t)(cluster * 2 + par) * CS + p) * nd * 2;
        for (int e = tid; e < 2 * nd && !abortw; e += 512) {
          u64 x = 0;
          bool ok = false;
          for (unsigned s = 0; s < SPIN && !ok; ++s) {
            x = __hip_atomic_load(theirs + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(x >> 32) == (unsigned)t + 1u;
          }
          if (!ok) abortw = 1;
          acc += (double)(unsigned)x;
        }
      }
      __syncthreads();
    }
  }
  u64 t1 = clock64();
  out[(size_t)blockIdx.x * 512 + tid] = acc;
  if (abortw && tid == 0) atomicAdd(fail, 1u);
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
int main(int argc, char** argv) {
  const int CS = argc > 1 ? atoi(argv[1]) : 8, rows = argc > 2 ? atoi(argv[2]) : 38, part = argc > 3 ? atoi(argv[3]) : 32;
  const int nd = rows * part, clusters = 208 / CS, steps = 300;
  double *pay, *out;
  u64 *gran, *cyc;
  unsigned *flag, *fail;
  hipMalloc(&pay, (size_t)clusters * 2 * CS * nd * 8);
  hipMalloc(&gran, (size_t)clusters * 2 * CS * nd * 16);
  hipMalloc(&flag, (size_t)clusters * 2 * CS * 4);
  hipMalloc(&out, (size_t)clusters * CS * 512 * 8);
  hipMalloc(&cyc, 8);
  hipMalloc(&fail, 4);
  printf("clusters of %d workgroups (%d clusters resident), block = %d rows x %d particles = %d doubles (%.1f KB), a workgroup reads %.1f KB per step\n", CS,
         clusters, rows, part, nd, nd * 8 / 1024.0, (CS - 1) * nd * 8 / 1024.0);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(flag, 0, (size_t)clusters * 2 * CS * 4);
    hipMemset(gran, 0, (size_t)clusters * 2 * CS * nd * 16);
    hipMemset(fail, 0, 4);
    hipLaunchKernelGGL(xbench, dim3(clusters * CS), dim3(512), 0, 0, mode, CS, nd, steps, pay, gran, flag, out, cyc, fail);
    hipDeviceSynchronize();
    u64 h = 0;
    unsigned f = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("%-58s %8.0f cycles per step%s\n", mode == 0 ? "payload + release fence + one flag, acquire poll, sc1 reads" : "tagged 8-byte granules (two per double), no flag",
           (double)h / steps, f ? "  (TIMED OUT)" : "");
  }
  return 0;
}
