// Stand-alone model of phase V of the 16-particle kernel (v = Kinv k, [N x N] x [N x 16]) to compare instruction forms:
//   mode 0: v_mfma_f64_16x16x4_f64, one dwordx4 of Kinv + one panel read feed 2 MFMAs            (the round-1/2 form)
//   mode 1: v_mfma_f64_4x4x4_4b_f64, the same loads feed 8 MFMAs, B rotated by v_mov_b32_dpp row_ror
//   mode 2: as 1, the rotated B operands read from the panel instead (4 ds_read per step)
//   mode 3: as 1 without the Kinv stream (A operands stay in registers): the instruction mix alone
//   mode 4: as 0 without the Kinv stream
//   mode 5 / 6: as 1 / 3 with the instruction stream grouped per 16-row batch: all 12 rotations, then the 32 MFMAs back to back
//               (a vector instruction between two 4x4x4 MFMAs of a wave costs ~14 cycles: tools/mfma4x4_probe.hip)
// 512-thread workgroups, one per CU, every wave takes 32-row blocks w, w + 8, ...; `steps` repetitions (the time loop) so that
// Kinv is streamed from L2 as in the rollout.  Prints cycles per repetition (wave 0 of workgroup 0) and the flop rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const v2d __attribute__((address_space(1))) * gptr2_t;
#define KR 18
template <int CTRL>
__device__ __forceinline__ double rot(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ __launch_bounds__(512) void vbench(const double* Kinv, int Npad, int steps, double* out, unsigned long long* cyc) {
  extern __shared__ double smem[];
  double* kv = smem;  // [Npad][KR]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int e = tid; e < Npad * KR; e += 512) kv[e] = 1e-3 * (e % 97);
  __syncthreads();
  const int m = lane & 15, kk = lane >> 4;
  const int nblk = (Npad + 31) >> 5;
  const size_t astep = (size_t)4 * Npad / 2;
  double sink = 0.0;
  unsigned long long t0 = clock64();
  for (int st = 0; st < steps; ++st) {
    for (int blk = wv; blk < nblk; blk += 8) {
      const int col = blk * 32 + 2 * m;
      gptr2_t a0 = (gptr2_t)((const double __attribute__((address_space(1)))*)Kinv + (size_t)kk * Npad + (col < Npad ? col : 0));
      const double* b0 = kv + kk * KR + m;
      v4d ae = {0, 0, 0, 0}, ao = {0, 0, 0, 0};
      v2d A0[4], A1[4];
      double B0[4], B1[4];
      const int nb = Npad >> 4;
      auto load = [&](v2d(&A)[4], double(&B)[4], int b) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (MODE < 3 || MODE == 5) A[u] = a0[(size_t)(4 * b + u) * astep];
          B[u] = b0[(4 * b + u) * 4 * KR];
        }
      };
      auto mma = [&](const v2d(&A)[4], const double(&B)[4], int b) {
        if (MODE == 5 || MODE == 6) {  // all rotations first, then 32 MFMAs with nothing between them
          double r1[4], r2[4], r3[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            r1[u] = rot<0x124>(B[u]);
            r2[u] = rot<0x128>(B[u]);
            r3[u] = rot<0x12C>(B[u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            ae[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, B[u], ae[0], 0, 0, 0);
            ao[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, B[u], ao[0], 0, 0, 0);
            ae[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, r1[u], ae[1], 0, 0, 0);
            ao[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, r1[u], ao[1], 0, 0, 0);
            ae[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, r2[u], ae[2], 0, 0, 0);
            ao[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, r2[u], ao[2], 0, 0, 0);
            ae[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, r3[u], ae[3], 0, 0, 0);
            ao[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, r3[u], ao[3], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          return;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (MODE == 0 || MODE == 4) {
            ae = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u].x, B[u], ae, 0, 0, 0);
            ao = __builtin_amdgcn_mfma_f64_16x16x4f64(A[u].y, B[u], ao, 0, 0, 0);
          } else {
            double b1, b2, b3;
            if (MODE == 2) {
              const double* r = kv + ((4 * b + u) * 4 + kk) * KR;
              b1 = r[(m + 4) & 15];
              b2 = r[(m + 8) & 15];
              b3 = r[(m + 12) & 15];
            } else {
              b1 = rot<0x124>(B[u]);
              b2 = rot<0x128>(B[u]);
              b3 = rot<0x12C>(B[u]);
            }
            ae[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, B[u], ae[0], 0, 0, 0);
            ao[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, B[u], ao[0], 0, 0, 0);
            ae[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, b1, ae[1], 0, 0, 0);
            ao[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, b1, ao[1], 0, 0, 0);
            ae[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, b2, ae[2], 0, 0, 0);
            ao[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, b2, ao[2], 0, 0, 0);
            ae[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].x, b3, ae[3], 0, 0, 0);
            ao[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[u].y, b3, ao[3], 0, 0, 0);
          }
        }
      };
      if (MODE == 3 || MODE == 4 || MODE == 6) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          A0[u] = (v2d){1.0 + lane * 1e-3 + u, 0.5 + lane * 1e-4};
          A1[u] = (v2d){0.7 + lane * 1e-3, 0.2 + lane * 1e-4 + u};
        }
      }
      load(A0, B0, 0);
      for (int b = 0; b + 1 < nb; b += 2) {
        load(A1, B1, b + 1);
        mma(A0, B0, b);
        load(A0, B0, b + 2 < nb ? b + 2 : nb - 1);
        mma(A1, B1, b + 1);
      }
      if (nb & 1) mma(A0, B0, nb - 1);
      sink += ae[0] + ae[1] + ae[2] + ae[3] + ao[0] + ao[1] + ao[2] + ao[3];
    }
    __syncthreads();
  }
  unsigned long long t1 = clock64();
  out[(size_t)blockIdx.x * 512 + tid] = sink;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

// k-split form: the four blocks of the 4x4x4 MFMA hold four different j-quads of the SAME 4 rows x 4 particle columns, so no operand is
// replicated (no rotations): lane l = 16 k + 4 blk + e,
//   A (one dwordx4 per row quad R): Kinv[row0 + 4 R + e][jb + 2 (4 k + blk)], [.. + 1]        (.x even j, .y odd j; 256 B contiguous per row)
//   B (one panel read per column quad q and parity): k[jb + 2 (4 k + blk) + par][4 q + e]       (conflict-free at pitch 18)
//   acc[R][q]: partial sums over this block's j-quads of v[row0 + 4 R + i][4 q + j], lane 16 i + 4 blk + j; the 4 blocks are added once at the end.
// A wave takes 16-row units w, w + 8, ...; batch = 32 rows of j: 4 Kinv loads + 8 panel reads feed 32 MFMAs with no vector instruction between.
template <bool STREAM, bool LDSB = true, int NT = 512, int DEPTH = 2>
__global__ __launch_bounds__(NT) void vbench_ks(const double* Kinv, int Npad, int steps, double* out, unsigned long long* cyc) {
  extern __shared__ double smem[];
  double* kv = smem;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int e = tid; e < (Npad + 32) * KR; e += NT) kv[e] = 1e-3 * (e % 97);
  __syncthreads();
  const int k = lane >> 4, blk = (lane >> 2) & 3, e = lane & 3;
  const int jp = 2 * (4 * k + blk);
  const int nunit = Npad >> 4, nb = (Npad + 31) >> 5;
  double sink = 0.0;
  unsigned long long t0 = clock64();
  for (int st = 0; st < steps; ++st) {
    for (int un = wv; un < nunit; un += NT / 64) {
      gptr2_t a0 = (gptr2_t)((const double __attribute__((address_space(1)))*)Kinv + (size_t)(un * 16 + e) * Npad + jp);
      const size_t rstep = (size_t)4 * Npad / 2;  // 4 rows in v2d units
      const double* b0 = kv + jp * KR + e;
      double acc[4][4];
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[R][q] = 0.0;
      v2d A0[4], A1[4];
      double B0[4][2], B1[4][2];
      auto load = [&](v2d(&A)[4], double(&B)[4][2], int b) {
#pragma unroll
        for (int R = 0; R < 4; ++R)
          if (STREAM) A[R] = a0[(size_t)R * rstep + 16 * b];
        if (LDSB) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            B[q][0] = b0[(32 * b) * KR + 4 * q];
            B[q][1] = b0[(32 * b + 1) * KR + 4 * q];
          }
        }
      };
      auto mma = [&](const v2d(&A)[4], const double(&B)[4][2]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int R = 0; R < 4; ++R)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[R][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[R].x, B[q][0], acc[R][q], 0, 0, 0);
#pragma unroll
        for (int R = 0; R < 4; ++R)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[R][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[R].y, B[q][1], acc[R][q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      if (!STREAM) {
#pragma unroll
        for (int R = 0; R < 4; ++R) {
          A0[R] = (v2d){1.0 + lane * 1e-3 + R, 0.5 + lane * 1e-4};
          A1[R] = (v2d){0.7 + lane * 1e-3, 0.2 + lane * 1e-4 + R};
        }
      }
      if (!LDSB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          B0[q][0] = 0.3 + lane * 1e-3 + q; B0[q][1] = 0.1 + lane * 1e-4 - q;
          B1[q][0] = 0.4 + lane * 1e-3 - q; B1[q][1] = 0.2 + lane * 1e-4 + q;
        }
      }
      load(A0, B0, 0);
      const unsigned long long u0 = clock64();
      if (DEPTH == 3) {  // loads two batches ahead of the MFMA run that uses them
        v2d A2[4];
        double B2[4][2];
        load(A1, B1, nb > 1 ? 1 : 0);
        int b = 0;
        for (; b + 2 < nb; b += 3) {
          load(A2, B2, b + 2);
          mma(A0, B0);
          load(A0, B0, b + 3 < nb ? b + 3 : nb - 1);
          mma(A1, B1);
          load(A1, B1, b + 4 < nb ? b + 4 : nb - 1);
          mma(A2, B2);
        }
        if (b < nb) mma(A0, B0);
        if (b + 1 < nb) mma(A1, B1);
      } else {
        for (int b = 0; b + 1 < nb; b += 2) {
          load(A1, B1, b + 1);
          mma(A0, B0);
          load(A0, B0, b + 2 < nb ? b + 2 : nb - 1);
          mma(A1, B1);
        }
        if (nb & 1) mma(A0, B0);
      }
      if (blockIdx.x == 0 && tid == 0 && un == wv && st == steps - 1) cyc[1] = clock64() - u0;  // wave 0's first unit of the last repetition
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int q = 0; q < 4; ++q) sink += acc[R][q];
    }
    __syncthreads();
  }
  unsigned long long t1 = clock64();
  out[(size_t)blockIdx.x * NT + tid] = sink;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <bool STREAM, bool LDSB = true, int NT = 512, int DEPTH = 2>
static void run_ks(const double* Kinv, int Npad, int steps, double* out, unsigned long long* cyc, const char* name) {
  const size_t lds = (size_t)(Npad + 32) * KR * 8;
  hipFuncSetAttribute((const void*)vbench_ks<STREAM, LDSB, NT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((vbench_ks<STREAM, LDSB, NT, DEPTH>), dim3(250), dim3(NT), lds, 0, Kinv, Npad, steps, out, cyc);
  hipDeviceSynchronize();
  unsigned long long h = 0, h1 = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&h1, cyc + 1, 8, hipMemcpyDeviceToHost);
  const double per = (double)h / steps;
  printf("   (wave 0, one 16-row unit: %.1f ticks per MFMA)\n", (double)h1 / (((Npad + 31) / 32) * 32.0));
  const double flop = 2.0 * (double)Npad * (32.0 * ((Npad + 31) / 32)) * 16.0;
  printf("%-62s %9.0f ticks per repetition, %6.1f flop/tick/CU, Kinv stream %5.1f B/tick/CU\n", name, per, flop / per, STREAM ? 8.0 * Npad * (32.0 * ((Npad + 31) / 32)) / per : 0.0);
}
template <int MODE>
static void run(const double* Kinv, int Npad, int steps, double* out, unsigned long long* cyc, const char* name) {
  const size_t lds = (size_t)Npad * KR * 8;
  hipFuncSetAttribute((const void*)vbench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(vbench<MODE>, dim3(250), dim3(512), lds, 0, Kinv, Npad, steps, out, cyc);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)h / steps;
  const int nblk = (Npad + 31) / 32;
  const double flop = 2.0 * 32.0 * nblk * Npad * 16.0;  // per workgroup and repetition (padded rows included)
  printf("%-62s %9.0f ticks per repetition, %6.1f flop/tick/CU, Kinv stream %5.1f B/tick/CU\n", name, per, flop / per, (MODE < 3 || MODE == 5) ? 8.0 * 32 * nblk * Npad / per : 0.0);
}
int main(int argc, char** argv) {
  const int Npad = argc > 1 ? atoi(argv[1]) : 304, steps = argc > 2 ? atoi(argv[2]) : 100;
  double *Kinv, *out;
  unsigned long long* cyc;
  const size_t n = (size_t)Npad * (Npad + 64) + 64;
  hipMalloc(&Kinv, n * 8); hipMalloc(&out, 250 * 1024 * 8); hipMalloc(&cyc, 64);
  double* h = (double*)malloc(n * 8);
  for (size_t i = 0; i < n; ++i) h[i] = 1e-3 * (double)(i % 1013);
  hipMemcpy(Kinv, h, n * 8, hipMemcpyHostToDevice);
  printf("Npad %d, %d repetitions, 250 workgroups of 8 waves\n", Npad, steps);
  run<0>(Kinv, Npad, steps, out, cyc, "16x16x4, Kinv streamed");
  run<1>(Kinv, Npad, steps, out, cyc, "4x4x4_4b + dpp-rotated B, Kinv streamed");
  run<2>(Kinv, Npad, steps, out, cyc, "4x4x4_4b + rotated B read from the panel, Kinv streamed");
  run<3>(Kinv, Npad, steps, out, cyc, "4x4x4_4b + dpp-rotated B, A in registers");
  run<4>(Kinv, Npad, steps, out, cyc, "16x16x4, A in registers");
  run<5>(Kinv, Npad, steps, out, cyc, "4x4x4_4b grouped (rotations, then 32 MFMAs), Kinv streamed");
  run<6>(Kinv, Npad, steps, out, cyc, "4x4x4_4b grouped, A in registers");
  run_ks<true>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split blocks (no rotations), Kinv streamed");
  run_ks<false>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split blocks, A in registers");
  run_ks<false, false>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split blocks, A and B in registers");
  run_ks<false, false, 1024>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split, A and B in registers, 16 waves per workgroup");
  run_ks<false, true, 1024>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split, A in registers, 16 waves per workgroup");
  run_ks<true, true, 1024>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split, Kinv streamed, 16 waves per workgroup");
  run_ks<false, true, 512, 3>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split, A in registers, panel reads 2 batches ahead");
  run_ks<true, true, 512, 3>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split, Kinv streamed, loads 2 batches ahead");
  run_ks<true, true, 1024, 3>(Kinv, Npad, steps, out, cyc, "4x4x4_4b k-split, Kinv streamed, 2 batches ahead, 16 waves");
  return 0;
}
