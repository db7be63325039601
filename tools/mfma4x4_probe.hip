// Operand layout and rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction) on gfx950.
//   part 1: lane l supplies A = code(l), B = code'(l); the host tries the layout hypothesis
//             A: lane = 16 k + 4 blk + i   B: lane = 16 k + 4 blk + j   D: lane = 16 i + 4 blk + j      (D_blk[i][j] = sum_k A_blk[i][k] B_blk[k][j])
//           against the device result and prints whether it holds.
//   part 2: cycles per instruction, 8 waves per CU, independent accumulators, beside v_mfma_f64_16x16x4_f64.
// Layout found (and used by tools/v_phase_bench.hip): lane l = 16 k + 4 blk + e;  A: A_blk[i = e][k]   B: B_blk[k][j = e]   D: lane 16 i + 4 blk + j.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void layout(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int CTRL>
__device__ __forceinline__ double rot(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(512) void rate(int mode, int nit, double* out, unsigned long long* cyc) {
  __shared__ double lds[64][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int e = threadIdx.x; e < 4096; e += 512) lds[e >> 6][e & 63] = 1e-3 * e;
  double x = 1.0 + lane * 1e-3, y = 0.5 + lane * 1e-4, x1 = x + 0.25, y1 = y + 0.125;  // distinct operands: identical calls would be merged
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  __syncthreads();
  unsigned long long t0 = clock64();
  if (mode == 0) {
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y, s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, s2, 0, 0, 0);
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, x, s3, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y1, s4, 0, 0, 0);
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y1, s5, 0, 0, 0);
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x1, s6, 0, 0, 0);
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, x1, s7, 0, 0, 0);
      }
    }
  } else if (mode == 1) {
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, x, a3, 0, 0, 0);
      }
    }
  } else if (mode == 3) {  // 8 independent accumulators, 6 32-bit DPP moves per 8 MFMAs (the rotated-B form of phase V)
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double b1 = rot<0x124>(y), b2 = rot<0x128>(y), b3 = rot<0x12C>(y);
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y, s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b1, s2, 0, 0, 0);
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, b1, s3, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b2, s4, 0, 0, 0);
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, b2, s5, 0, 0, 0);
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b3, s6, 0, 0, 0);
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, b3, s7, 0, 0, 0);
        y += 1.0;  // (keeps the rotations inside the loop)
      }
    }
  } else if (mode == 4) {  // 8 independent accumulators, one panel read (LDS) per 8 MFMAs
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double bb = lds[(it * 4 + u) & 63][lane];
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, bb, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, bb, s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, bb, s2, 0, 0, 0);
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, bb, s3, 0, 0, 0);
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(bb, y1, s4, 0, 0, 0);
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(bb, y, s5, 0, 0, 0);
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(bb, x1, s6, 0, 0, 0);
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(bb, x, s7, 0, 0, 0);
      }
    }
  } else if (mode == 5) {  // 8 independent accumulators, one plain 32-bit VALU instruction between MFMAs
    int c = lane;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s0, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y, s1, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, s2, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, x, s3, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y1, s4, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, y1, s5, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x1, s6, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
        s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, x1, s7, 0, 0, 0);
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(c));
      }
    }
    s0 += c;
  } else if (mode == 6) {  // 16 independent accumulators, 4 A x 4 B operand registers (the k-split form of tools/v_phase_bench.hip)
    double acc[4][4];
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[R][q] = 0.0;
    const double A[4] = {x, x1, x + 0.5, x1 + 0.5}, B[4] = {y, y1, y + 0.5, y1 + 0.5};
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int R = 0; R < 4; ++R)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[R][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[R], B[q], acc[R][q], 0, 0, 0);
    }
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
      for (int q = 0; q < 4; ++q) s0 += acc[R][q];
  } else if (mode == 7) {  // 8 accumulators, result written to a DIFFERENT register than the addend (two register sets, ping-pong)
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, d0, d1, d2, d3, d4, d5, d6, d7;
#define MF(D, A, B, C) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "v"(C))
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        MF(d0, x, y, c0); MF(d1, x1, y, c1); MF(d2, y, x, c2); MF(d3, y1, x, c3); MF(d4, x, y1, c4); MF(d5, x1, y1, c5); MF(d6, y, x1, c6); MF(d7, y1, x1, c7);
        MF(c0, x, y, d0); MF(c1, x1, y, d1); MF(c2, y, x, d2); MF(c3, y1, x, d3); MF(c4, x, y1, d4); MF(c5, x1, y1, d5); MF(c6, y, x1, d6); MF(c7, y1, x1, d7);
      }
    }
    s0 = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else if (mode == 8) {  // as mode 0 (in place) through the same inline asm, for comparison
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
#define MFI(C, A, B) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B))
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        MFI(c0, x, y); MFI(c1, x1, y); MFI(c2, y, x); MFI(c3, y1, x); MFI(c4, x, y1); MFI(c5, x1, y1); MFI(c6, y, x1); MFI(c7, y1, x1);
      }
    }
    s0 = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else if (mode >= 9 && mode <= 14) {
    // round 6: what separates the 8.2 cycles per SIMD of mode 0 from the 17.3 phase V of the lean kernel sees (tools/vissue_bench.hip)?  The number of
    // accumulators in rotation (NA) and the number of distinct operand registers (NO A operands, NO / 2 B operands), separately
    //   9: NA 4, 4 operand registers    10: NA 8, 12 + 6 distinct    11: NA 4, 12 + 6 (the kernel at two row tiles)    12: NA 6, 12 + 6 (three row tiles)
    //  13: NA 12, 12 + 6                14: NA 2, 12 + 6
    double A[12], B[6], c[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { A[i] = x + 0.01 * i; c[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 6; ++i) B[i] = y + 0.02 * i;
    const int NA = mode == 9 ? 4 : (mode == 10 ? 8 : (mode == 11 ? 4 : (mode == 12 ? 6 : (mode == 13 ? 12 : 2))));
    for (int it = 0; it < nit; ++it) {
      if (mode == 9) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { MFI(c[0], x, y); MFI(c[1], x1, y); MFI(c[2], y, x); MFI(c[3], y1, x); }
      } else if (NA == 8) {
#pragma unroll
        for (int u = 0; u < 32; ++u) MFI(c[u % 8], A[u % 12], B[(u / 2) % 6]);
      } else if (NA == 4) {
#pragma unroll
        for (int u = 0; u < 32; ++u) MFI(c[u % 4], A[u % 12], B[(u / 2) % 6]);
      } else if (NA == 6) {
#pragma unroll
        for (int u = 0; u < 32; ++u) MFI(c[u % 6], A[u % 12], B[(u / 3) % 6]);
      } else if (NA == 12) {
#pragma unroll
        for (int u = 0; u < 32; ++u) MFI(c[u % 12], A[u % 12], B[(u / 2) % 6]);
      } else {
#pragma unroll
        for (int u = 0; u < 32; ++u) MFI(c[u % 2], A[u % 12], B[(u / 2) % 6]);
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s0 += c[i];
  } else {  // dependent chain of 4x4x4 on one accumulator
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s0, 0, 0, 0);
    }
  }
  unsigned long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + a0[0] + a1[1] + a2[2] + a3[3];
  if (lane == 0 && blockIdx.x == 0) cyc[wv] = t1 - t0;
}
int main() {
  double ha[64], hb[64], hd[64], *a, *b, *d;
  for (int l = 0; l < 64; ++l) {
    ha[l] = 1.0 + l;          // distinct codes: products identify the pairing
    hb[l] = 1.0 + 0.01 * l;
  }
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int blk = 0; blk < 4; ++blk)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double ref = 0;
        for (int k = 0; k < 4; ++k) ref += ha[16 * k + 4 * blk + i] * hb[16 * k + 4 * blk + j];
        worst = fmax(worst, fabs(ref - hd[16 * i + 4 * blk + j]));
      }
  printf("layout A: lane = 16 k + 4 blk + i | B: lane = 16 k + 4 blk + j | D: lane = 16 i + 4 blk + j  ->  max |diff| = %.3e  (%s)\n", worst,
         worst < 1e-9 ? "HOLDS" : "does NOT hold");
  if (worst >= 1e-9) {
    printf("device D:");
    for (int l = 0; l < 64; ++l) printf(" %.4f", hd[l]);
    printf("\n");
  }
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 512 * 8); hipMalloc(&cyc, 64);
  const int nit = 200;
  const char* names[15] = {"4x4x4_4b, 8 independent accumulators", "16x16x4, 4 independent accumulators", "4x4x4_4b, one dependent chain",
                          "4x4x4_4b, 8 acc + 6 v_mov_b32_dpp per 8", "4x4x4_4b, 8 acc + one LDS read per 8", "4x4x4_4b, 8 acc + one v_add_u32 per MFMA", "4x4x4_4b, 16 acc, 4 A x 4 B operands", "4x4x4_4b, 8 acc, result register != addend register", "4x4x4_4b, 8 acc in place (inline asm)", "4 acc, 4 operand registers", "8 acc, 12 A + 6 B distinct operand registers", "4 acc, 12 A + 6 B", "6 acc, 12 A + 6 B", "12 acc, 12 A + 6 B", "2 acc, 12 A + 6 B"};
  for (int mode = 0; mode < 15; ++mode) {
    hipLaunchKernelGGL(rate, dim3(256), dim3(512), 0, 0, mode, nit, out, cyc);
    hipLaunchKernelGGL(rate, dim3(256), dim3(512), 0, 0, mode, nit, out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double per_wave = (double)h[0] / (nit * 32.0);
    // 2 waves per SIMD share the pipe: per-SIMD ticks per instruction = per_wave / 2
    const double flop = mode == 1 ? 2048.0 : 512.0;
    printf("%-42s %7.1f ticks per instruction and wave, %6.1f per SIMD -> %5.1f flop/tick/SIMD\n", names[mode], per_wave, per_wave / 2, flop / (per_wave / 2));
  }
  return 0;
}
