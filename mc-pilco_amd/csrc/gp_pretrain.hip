// Gram / Cholesky / triangular inverse / alpha / packing / subset-of-data selection for gfx950.
// Replaces GP_prior.forward, get_alpha and get_SOD of the reference (gpr_lib/GP_prior/GP_prior.py:
// 91-135, 232-257) and the kernel classes' get_covariance (Stationary_GP.py:162-170,
// Sparse_GP.py:426-441,625-646, GP_prior.py:314-335).  These run once per GP per trial
// (Model_learning.pretrain_gp), so they are written for clarity and fp64 accuracy; the
// per-step hot path is rollout.hip.
#include <type_traits>

#include "mcp_device.h"
#include "../../include/mcpilco_hip_debug.h"

using namespace mcp;

// ---------------------------------------------------------------------------------------
// covariance matrices
// ---------------------------------------------------------------------------------------
__global__ void cov_build_kernel(mcp_kernel kn, int N1, const double* __restrict__ X1, int N2, const double* __restrict__ X2,
                                 int add_noise, double* __restrict__ K, int ldk) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (i >= N1 || j >= N2) return;
  double k = kern_eval(kn, X1 + (size_t)i * kn.D, 1, X2 + (size_t)j * kn.D, 1);
  if (add_noise && i == j) k += kern_sigma_n2(kn);
  K[(size_t)i * ldk + j] = k;
}

__global__ void cov_diag_kernel(mcp_kernel kn, int N, const double* __restrict__ X, int add_noise, double* __restrict__ diag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double k = kern_diag(kn, X + (size_t)i * kn.D, 1);
  diag[i] = add_noise ? k + kern_sigma_n2(kn) : k;
}

// ---------------------------------------------------------------------------------------
// Cholesky A = U^T U, upper, in place, one workgroup, blocked right-looking (NB = 16):
//   per block row kb:  (1) 16x16 diagonal block factored in LDS by wave 0,
//                      (2) row panel U[kb:kb+16, kb+16:N] = U_kk^-T A[...]  (thread per column),
//                          kept in LDS for (3) the trailing update A[i][j] -= sum_m U[m][i] U[m][j].
// ---------------------------------------------------------------------------------------
#define CH_NB 16
#define CH_NT 1024

__device__ __forceinline__ int imin_d(int a, int b) { return a < b ? a : b; }

__global__ __launch_bounds__(CH_NT) void chol_factor_kernel(int N, double* __restrict__ A, int lda, double* __restrict__ logdet,
                                                            uint32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* dg = smem;                     // [NB][NB+1] diagonal block
  double* pn = smem + CH_NB * (CH_NB + 1);  // [NB][N] row panel
  const int tid = threadIdx.x;
  double ld_acc = 0.0;  // thread 0 only
  uint32_t bad = 0;

  for (int kb = 0; kb < N; kb += CH_NB) {
    const int nb = min(CH_NB, N - kb);
    // (1) diagonal block -> LDS, factor with the first nb lanes of wave 0
    if (tid < CH_NB * CH_NB) {
      int r = tid / CH_NB, c = tid % CH_NB;
      dg[r * (CH_NB + 1) + c] = (r < nb && c < nb && c >= r) ? A[(size_t)(kb + r) * lda + kb + c] : 0.0;
    }
    __syncthreads();
    if (tid < MCP_WAVE) {
      volatile double* dgv = dg;  // single-wave section: LDS accesses must stay in program order
      for (int k = 0; k < nb; ++k) {
        double d = dgv[k * (CH_NB + 1) + k];
        if (!(d > 0.0)) bad |= MCP_STATUS_NOT_SPD;
        double sd = sqrt(d);
        if (tid == 0) ld_acc += log(sd);
        // scale row k, then rank-1 update of the rows below (lanes <-> columns)
        double ukc = 0.0;
        if (tid < nb && tid >= k) {
          ukc = (tid == k) ? sd : dgv[k * (CH_NB + 1) + tid] / sd;
          dgv[k * (CH_NB + 1) + tid] = ukc;
        }
        __builtin_amdgcn_wave_barrier();
        if (tid < nb && tid > k) {
          for (int r = k + 1; r <= tid; ++r) dgv[r * (CH_NB + 1) + tid] = dgv[r * (CH_NB + 1) + tid] - dgv[k * (CH_NB + 1) + r] * ukc;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    // write the factored diagonal block back (and zero below the diagonal)
    if (tid < CH_NB * CH_NB) {
      int r = tid / CH_NB, c = tid % CH_NB;
      if (r < nb && c < nb) A[(size_t)(kb + r) * lda + kb + c] = (c >= r) ? dg[r * (CH_NB + 1) + c] : 0.0;
    }
    const int j0 = kb + nb;
    const int nc = N - j0;
    // (2) panel solve: column j of the panel solves U_kk^T x = a  (forward substitution)
    for (int c = tid; c < nc; c += CH_NT) {
      double x[CH_NB];
#pragma unroll
      for (int r = 0; r < CH_NB; ++r) {
        if (r < nb) {
          double s = A[(size_t)(kb + r) * lda + j0 + c];
          for (int m = 0; m < r; ++m) s = fma(-dg[m * (CH_NB + 1) + r], x[m], s);
          x[r] = s / dg[r * (CH_NB + 1) + r];
          A[(size_t)(kb + r) * lda + j0 + c] = x[r];
          pn[(size_t)r * nc + c] = x[r];
        }
      }
    }
    __syncthreads();
    // (3) trailing update over the upper triangle (i <= j)
    //     4x4 register tiles: per panel row m a thread reads 4 + 4 panel values from LDS for 16 multiply-adds (one element per
    //     thread needed 2 LDS reads per multiply-add and made the phase instruction bound); only tiles on or above the diagonal
    const int nts = (nc + 3) >> 2;
    for (int t = tid; t < nts * nts; t += CH_NT) {
      const int ti = t / nts, tj = t - ti * nts;
      if (tj < ti) continue;
      const int i0 = 4 * ti, jj0 = 4 * tj;
      double acc[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = imin_d(i0 + r, nc - 1), j = imin_d(jj0 + c, nc - 1);
          acc[r][c] = A[(size_t)(j0 + i) * lda + j0 + j];
        }
#pragma unroll
      for (int m = 0; m < CH_NB; ++m) {
        if (m < nb) {
          double pi[4], pj[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pi[r] = pn[(size_t)m * nc + imin_d(i0 + r, nc - 1)];
            pj[r] = pn[(size_t)m * nc + imin_d(jj0 + r, nc - 1)];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fma(-pi[r], pj[c], acc[r][c]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = i0 + r, j = jj0 + c;
          if (i < nc && j < nc && j >= i) A[(size_t)(j0 + i) * lda + j0 + j] = acc[r][c];
        }
    }
    __syncthreads();
  }
  // zero the strictly-lower part that lies outside the diagonal blocks
  for (int idx = tid; idx < N * N; idx += CH_NT) {
    int r = idx / N, c = idx % N;
    if (c < r && (c / CH_NB) != (r / CH_NB)) A[(size_t)r * lda + c] = 0.0;
  }
  if (tid == 0) *logdet = 2.0 * ld_acc;
  if (bad) atomicOr(status, bad);
}

// ---------------------------------------------------------------------------------------
// Uinv = U^-1 : one thread per column, rows swept bottom-up with the row of U broadcast from LDS
// ---------------------------------------------------------------------------------------
#define TI_NT 64
__global__ __launch_bounds__(TI_NT) void tri_inverse_kernel(int N, const double* __restrict__ U, int ldu, double* __restrict__ Ui,
                                                            int ldi) {
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [N] one row of U
  const int j = blockIdx.x * TI_NT + threadIdx.x;                // my column
  const int jmax = min(N - 1, blockIdx.x * TI_NT + TI_NT - 1);   // last column of this block
  for (int i = jmax; i >= 0; --i) {
    __syncthreads();
    for (int m = i + threadIdx.x; m <= jmax; m += TI_NT) smem[m] = U[(size_t)i * ldu + m];
    __syncthreads();
    if (j < N) {
      double x;
      if (j < i) {
        x = 0.0;
      } else {
        double s0 = (j == i) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int m = i + 1;
        for (; m + 3 <= j; m += 4) {
          s0 = fma(-smem[m], Ui[(size_t)m * ldi + j], s0);
          s1 = fma(-smem[m + 1], Ui[(size_t)(m + 1) * ldi + j], s1);
          s2 = fma(-smem[m + 2], Ui[(size_t)(m + 2) * ldi + j], s2);
          s3 = fma(-smem[m + 3], Ui[(size_t)(m + 3) * ldi + j], s3);
        }
        for (; m <= j; ++m) s0 = fma(-smem[m], Ui[(size_t)m * ldi + j], s0);
        x = ((s0 + s1) + (s2 + s3)) / smem[i];
      }
      Ui[(size_t)i * ldi + j] = x;
    }
  }
  // rows below the block's last column are zero for these columns
  if (j < N)
    for (int i = jmax + 1; i < N; ++i) Ui[(size_t)i * ldi + j] = 0.0;
}

// Uinv = U^-1 for N <= 1024: one WAVE per column j.  Back substitution x_i = (delta_ij - sum_{i<m<=j} U[i][m] x_m) / U[i][i],
// i = j .. 0: lane l keeps x_m for m = l (mod 64) in registers, the row of U is one coalesced read per 64 columns (the next
// row is fetched while the current dot product is reduced), the dot product a wave64 DPP sum.  N columns run in parallel
// (the thread-per-column kernel above walks its own earlier results through global memory: 2 ms at N=300 against ~0.1 ms).
#define TW_KM 16  // 64 * TW_KM >= N
__global__ __launch_bounds__(256) void tri_inverse_wave_kernel(int N, const double* __restrict__ U, int ldu, double* __restrict__ Ui, int ldi) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);  // this wave's column
  if (j >= N) return;
  double x[TW_KM];
#pragma unroll
  for (int k = 0; k < TW_KM; ++k) x[k] = 0.0;
  const int kj = j >> 6;  // register slots 0..kj can hold a nonzero
  double urow[TW_KM], unext[TW_KM];
  auto load_row = [&](double (&r)[TW_KM], int i) {
#pragma unroll
    for (int k = 0; k < TW_KM; ++k) {
      const int m = k * 64 + lane;
      r[k] = (k <= kj && i >= 0 && m <= j) ? U[(size_t)i * ldu + m] : 0.0;
    }
  };
  load_row(urow, j);
  for (int i = j; i >= 0; --i) {
    load_row(unext, i - 1);
    double part = 0.0, uii = 0.0;
#pragma unroll
    for (int k = 0; k < TW_KM; ++k) {
      if (k <= kj) {
        const int m = k * 64 + lane;
        if (m > i) part = fma(urow[k], x[k], part);  // x_m is still 0 for m > j
        if (m == i) uii = urow[k];
      }
    }
    const double dot = wave_sum(part);
    const double d = wave_sum(uii);  // the diagonal element, from the lane that owns column i
    const double xi = ((i == j ? 1.0 : 0.0) - dot) / d;
#pragma unroll
    for (int k = 0; k < TW_KM; ++k)
      if (k * 64 + lane == i) x[k] = xi;
#pragma unroll
    for (int k = 0; k < TW_KM; ++k) urow[k] = unext[k];
  }
  // column j of the result: rows <= j from the registers, zeros below
#pragma unroll
  for (int k = 0; k < TW_KM; ++k) {
    const int m = k * 64 + lane;
    if (m < N) Ui[(size_t)m * ldi + j] = m <= j ? x[k] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------
// MFMA-blocked forms of the two kernels above (round 3; GP_prior.forward's torch.cholesky / torch.inverse, GP_prior.py:106-110,
// once per GP per trial in pretrain and once per EPOCH in GP_prior.fit_model, GP_prior.py:179-230).  One workgroup of 16 waves
// per matrix; the matrix stays in L2 / the CU's L1, every 16x16 block product runs on v_mfma_f64_16x16x4_f64:
//   A operand  lane l -> A[i = l & 15][k = l >> 4],   B operand  lane l -> B[k = l >> 4][j = l & 15],
//   accumulator register r of lane l -> D[(l >> 4) + 4 r][l & 15]          (so register u of an accumulator IS the B operand of
//   step u of a following product: D2 = A2 * D needs no data movement).
// The 16x16 diagonal blocks are factored / inverted by ONE wave in registers: lane c holds column c, scalars travel by v_readlane
// (the LDS form above spends ~13 k cycles per block in volatile round trips; this one ~5 k).
// Measured at N = 300 (tools/time_fit_model.py, rocprofv3): see DESIGN.md 4.5.
// ---------------------------------------------------------------------------------------
#define CM_NT 512  // (1024 threads = 128 registers: the in-register diagonal block of wave 0 spills 26 of them)
#define CM_NW (CM_NT / 64)
typedef double v4d_p __attribute__((ext_vector_type(4)));
typedef double __attribute__((address_space(1))) * gdp_t;         // explicit global pointers: a noinline device function would otherwise
typedef const double __attribute__((address_space(1))) * gcdp_t;  // address its pointer arguments as flat (64-bit address per lane and load)
// arguments of a non-kernel function arrive in vector registers even when they are uniform: back to scalars
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ gdp_t uniform_ptr(gdp_t p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (gdp_t)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double lane_get(double v, int l) {  // l: a compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// the value of lane R of each 16-lane row, in every lane of that row (DPP row_newbcast: stays in the vector registers -- the 136
// scalars a 16x16 block needs through v_readlane overflowed the scalar file into v_writelane / v_readlane spill pairs)
template <int R>
__device__ __forceinline__ double row_get(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + R, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + R, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Column steps k = K..15 of the in-register Cholesky of a 16x16 block (lane c: column c in x[]; rows below the diagonal are scratch)
// with the inverse riding along: after step k row k of U is final, which is all that row k of L^-1 = U^-T needs,
//   Linv[k][c] = (delta_kc - sum_{m < k} U[m][k] Linv[m][c]) / U[k][k]          (lane c: v[m] = Linv[m][c] = Uinv[c][m]),
// k independent FMAs that fill the stalls of the step's dependent chain (rsq, Newton steps, the pivot row) instead of a second
// serial pass of sixteen rows after it.
template <int K>
struct Chol16Col {
  template <int Rr>
  static __device__ __forceinline__ void update(double uk, double (&x)[16]) {
    if constexpr (Rr < 16) {
      x[Rr] = fma(-row_get<Rr>(uk), uk, x[Rr]);
      update<Rr + 1>(uk, x);
    }
  }
  template <int M>
  static __device__ __forceinline__ void dotl(const double (&x)[16], const double (&v)[16], double& s0, double& s1) {
    if constexpr (M < K) {
      if constexpr (M & 1)
        s1 = fma(-row_get<K>(x[M]), v[M], s1);
      else
        s0 = fma(-row_get<K>(x[M]), v[M], s0);
      dotl<M + 1>(x, v, s0, s1);
    }
  }
  static __device__ __forceinline__ void run(double (&x)[16], double (&v)[16], int c, uint32_t& bad) {
    const double dk = row_get<K>(x[K]);
    if (!(dk > 0.0)) bad |= MCP_STATUS_NOT_SPD;
    // sqrt(dk) and 1 / sqrt(dk) together (v_rsq_f64 seed, coupled Goldschmidt step, two Newton steps each): the pivots of a Gram
    // matrix are far from the denormal / overflow ranges the library forms rescale for, and a pivot <= 0 or NaN still gives NaN
    const double y = __builtin_amdgcn_rsq(dk);
    double g = dk * y, h = 0.5 * y;
    const double r0 = fma(-h, g, 0.5);
    g = fma(g, r0, g);
    h = fma(h, r0, h);
    g = fma(fma(-g, g, dk), h, g);
    g = fma(fma(-g, g, dk), h, g);
    double is = h + h;
    is = fma(is, fma(-g, is, 1.0), is);
    is = fma(is, fma(-g, is, 1.0), is);
    const double uk = c == K ? g : (c > K ? x[K] * is : 0.0);
    x[K] = uk;
    update<K + 1>(uk, x);
    double s0 = c == K ? 1.0 : 0.0, s1 = 0.0;
    dotl<0>(x, v, s0, s1);
    double vk = (s0 + s1) * is;
    asm volatile("" : "+v"(vk));  // (pins the step's work before the scheduling fence below)
    v[K] = vk;
    __builtin_amdgcn_sched_barrier(0);  // (or the scheduler hoists every broadcast of every step to the top and spills a hundred registers)
    if constexpr (K < 15) Chol16Col<K + 1>::run(x, v, c, bad);
  }
};

// lane c (= lane & 15; the four 16-lane rows of the wave work redundantly) holds column c of an upper-triangular 16x16 block in
// x[0..15] (x[r] = U[r][c], 0 below the diagonal).  Returns column c of U^-1 in w[].  inv_d[r] = 1 / U[r][r].
__device__ __forceinline__ void tri16_inverse(const double (&x)[16], const double (&inv_d)[16], int c, double (&w)[16]) {
#pragma unroll
  for (int m = 0; m < 16; ++m) w[m] = 0.0;
#pragma unroll
  for (int r = 15; r >= 0; --r) {
    double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int m = r + 1; m < 16; ++m) s = fma(-lane_get(x[r], m), w[m], s);  // U[r][m] w[m]  (w[m] = 0 for m > c)
    w[r] = s * inv_d[r];
  }
}

__global__ __launch_bounds__(CM_NT) void chol_factor_mfma_kernel(int N, double* __restrict__ A, int lda, double* __restrict__ logdet,
                                                                 uint32_t* __restrict__ status, size_t a_stride, size_t ld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  A += (size_t)blockIdx.x * a_stride;  // (one workgroup per matrix of a batch: the G GPs of a training epoch, mcp_nll_epoch)
  logdet += (size_t)blockIdx.x * ld_stride;
  double* ui = smem;        // [16][16]  U_kk^-1 (row m, column r at ui[m * 16 + r])
  double* pn = smem + 256;  // [16][ncp] the row panel of this block row
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NBK = (N + 15) >> 4;
  double ld_acc = 0.0;  // wave 0, lanes 0..15: sum of log U[c][c] over the block rows
  uint32_t bad = 0;
  for (int kbk = 0; kbk < NBK; ++kbk) {
    const int kb = kbk << 4, nb = min(16, N - kb);
    const int j0 = kb + 16, nc = N > j0 ? N - j0 : 0, nct = (nc + 15) >> 4, ncp = nct << 4;
    // (1) diagonal block: factor in registers, write back, invert
    if (wv == 0) {
      const int c = li;
      double x[16], inv_d[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = (r <= c && c < nb) ? A[(size_t)(kb + r) * lda + kb + c] : (r == c ? 1.0 : 0.0);  // identity beyond N
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const double dk = lane_get(x[k], k);
        if (!(dk > 0.0)) bad |= MCP_STATUS_NOT_SPD;
        const double sd = sqrt(dk), is = 1.0 / sd;
        inv_d[k] = is;
        const double uk = c == k ? sd : (c > k ? x[k] * is : 0.0);
        x[k] = uk;
#pragma unroll
        for (int r = k + 1; r < 16; ++r) x[r] = fma(-lane_get(uk, r), uk, x[r]);  // (only rows r <= c are meaningful)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r > c) x[r] = 0.0;
        if (lane < 16 && r < nb && c < nb) A[(size_t)(kb + r) * lda + kb + c] = x[r];
      }
      if (lane < 16 && c < nb) {
        double dcc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) dcc = r == c ? x[r] : dcc;
        ld_acc += log(dcc);
      }
      double w[16];
      tri16_inverse(x, inv_d, c, w);
      if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ui[r * 16 + c] = w[r];
      }
    }
    __syncthreads();
    // (2) row panel P = U_kk^-T A[kb:kb+16, j0:N]:  D[i = r][j] = sum_m W[m][r] A[kb + m][j0 + j]
    for (int tile = wv; tile < nct; tile += CM_NW) {
      const int col = j0 + 16 * tile + li;
      v4d_p acc = {0.0, 0.0, 0.0, 0.0};
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = ui[(4 * u + kq) * 16 + li];
        const int row = kb + 4 * u + kq;
        bv[u] = (col < N && row < N) ? A[(size_t)row * lda + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq + 4 * r;
        const bool ok = col < N && kb + i < N;
        if (ok) A[(size_t)(kb + i) * lda + col] = acc[r];
        pn[i * ncp + 16 * tile + li] = ok ? acc[r] : 0.0;
      }
    }
    __syncthreads();
    // (3) trailing update, blocks (I, J), I <= J, of the upper triangle:  T[i][j] -= sum_m P[m][16 I + i] P[m][16 J + j]
    // (round 4: the tiles of a wave are independent, but each one was load -> MFMA -> store with the matrix in global memory: a memory
    //  round trip per tile on every wave, 38 of them in a row in the first block step at N = 400.  Now the NEXT tile's accumulator
    //  loads are in flight while the current one is multiplied and stored.)
    const int nblk = nct * (nct + 1) / 2;
    auto tile_of = [&](int bidx, int& I, int& J) {
      I = 0;
      int rem = bidx;
      while (rem >= nct - I) {
        rem -= nct - I;
        ++I;
      }
      J = I + rem;
    };
    auto load_acc = [&](int I, int J, v4d_p& acc) {
      const int col = j0 + 16 * J + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = j0 + 16 * I + kq + 4 * r;
        acc[r] = (row < N && col < N) ? A[(size_t)row * lda + col] : 0.0;
      }
    };
    int I = 0, J = 0, In = 0, Jn = 0;
    v4d_p acc = {0.0, 0.0, 0.0, 0.0}, accn = {0.0, 0.0, 0.0, 0.0};
    if (wv < nblk) {
      tile_of(wv, I, J);
      load_acc(I, J, acc);
    }
    for (int bidx = wv; bidx < nblk; bidx += CM_NW) {
      const bool more = bidx + CM_NW < nblk;
      if (more) {
        tile_of(bidx + CM_NW, In, Jn);
        load_acc(In, Jn, accn);
      }
      const int col = j0 + 16 * J + li;
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = -pn[(4 * u + kq) * ncp + 16 * I + li];
        bv[u] = pn[(4 * u + kq) * ncp + 16 * J + li];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = j0 + 16 * I + kq + 4 * r;
        if (row < N && col < N) A[(size_t)row * lda + col] = acc[r];
      }
      I = In;
      J = Jn;
      acc = accn;
    }
    __syncthreads();
  }
  // zero the strictly lower part (torch.cholesky(upper=True) returns zeros there; the trailing updates wrote scratch into the
  // lower halves of the diagonal blocks)
  for (int idx = tid; idx < N * N; idx += CM_NT) {
    const int r = idx / N, c = idx - r * N;
    if (c < r) A[(size_t)r * lda + c] = 0.0;
  }
  if (wv == 0) {
    const double tot = wave_sum(lane < 16 ? ld_acc : 0.0);
    if (lane == 0) *logdet = 2.0 * tot;
  }
  if (bad) atomicOr(status, bad);
}

// The same factorization LEFT-looking (round 4): block row I of U is finished in one go,
//   T_IJ = A_IJ - sum_{k < I} U_kI^T U_kJ,   U_II = chol(T_II),   U_IJ = U_II^-T T_IJ   (J > I),
// so every 16x16 tile of the matrix is written ONCE (the right-looking kernel above reads, updates and writes every trailing tile in
// every block step: a store -> barrier -> load chain per step that its MFMAs wait behind), the sums over k stream finished, read-only
// rows with their loads two k-steps ahead, and the one serial chain -- factoring and inverting the diagonal block, wave 0 -- runs
// BESIDE the other waves' sums, which do not need it until their last four MFMAs:
//   wave 0:       T_II = P_I - U_(I-1)I^T U_(I-1)I  (P_I and the tile both wait in LDS, see below) -> columns in registers -> U_II, W_I = U_II^-1
//   tile waves:   tiles J = I + 1 + hw + 6 s, up to CL_GS at a time in the same k loop (their loads overlap), T_IJ kept in registers
//   wave 1 also:  P_(I+1) = A_(I+1)(I+1) - sum_{k < I} U_k(I+1)^T U_k(I+1): all of the NEXT diagonal block's sum that can be had
//                 before this row is finished; both of its MFMA operands are the B operand of the wave's tile J = I + 1 -- no loads
//                 (in the last block rows, where some waves have no tile, the first of those does it instead)
//   barrier;  U_IJ = W_I^T T_IJ (register u of T IS the B operand of step u), wave 1 leaves U_I(I+1) in LDS for the next row;  barrier.
// The chain per block row is then: 4 MFMAs, two LDS round trips, the 16 column steps, the inverse, two barriers and one tile product.
// CL_MAXS tile slots per tile wave, up to CL_GS of them in one k loop (registers).
#ifdef CLX_STAMPS  // experiment build only (build.py --variant-gp): cycle stamps of wave 0 / wave 1 per block row
__device__ unsigned long long g_clx[16 * 128];
#define CLX_T(w, slot) do { if (blockIdx.x == 0 && wv == (w) && lane == 0 && I < 128) g_clx[I * 16 + (slot)] = clock64(); } while (0)
extern "C" int mcp_debug_read_chol_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clx), sizeof(g_clx)); }
#else
#define CLX_T(w, slot) do { } while (0)
#endif
// the role of wave 0 (a function of its own: its registers are then allocated apart from the tile role's)
__device__ __noinline__ void chol_left_diag_role(int N_, gdp_t A_, int lda_, double* __restrict__ logdet, uint32_t* __restrict__ status,
                                                 double* ui, double* dg, double (*dgp)[256], double* nt) {
  const int N = uniform_int(N_), lda = uniform_int(lda_);
  const gdp_t A = uniform_ptr(A_);
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, li = lane & 15;
  const int NBK = (N + 15) >> 4;
  const int wv = 0;
  (void)wv;
  {
    double ld_acc = 0.0;
    uint32_t bad = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // P_0 = A_00 (identity beyond N), no tile above it
      const int row = kq + 4 * r;
      dgp[0][row * 16 + li] = (row < N && li < N) ? A[(size_t)row * lda + li] : (row == li ? 1.0 : 0.0);
      nt[row * 16 + li] = 0.0;
    }
    for (int I = 0; I < NBK; ++I) {
      int kb = I << 4;
      asm volatile("" : "+s"(kb));  // (or the sixteen store addresses of the block become 64-bit induction variables, spilled and reloaded every row)
      const int nb = min(16, N - kb);
      CLX_T(0, 0);
      v4d_p acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = dgp[I & 1][(kq + 4 * r) * 16 + li];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double av = nt[(4 * u + kq) * 16 + li];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, av, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dg[(kq + 4 * r) * 16 + li] = acc[r];
      __builtin_amdgcn_wave_barrier();
      int c = li;
      asm volatile("" : "+v"(c));  // (or sixty loop-invariant masks and constants of c are hoisted out of the row loop and spilled)
      double x[16], w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const double v = dg[r * 16 + c];
        x[r] = r <= c ? v : 0.0;
      }
      CLX_T(0, 1);
      Chol16Col<0>::run(x, w, c, bad);  // w[m] = Uinv[c][m]
      if (lane < 16 && c < nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r < nb) A[(size_t)(kb + r) * lda + kb + c] = r <= c ? x[r] : 0.0;  // (zeros below the diagonal, as torch.cholesky(upper=True) returns)
        double dcc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) dcc = r == c ? x[r] : dcc;
        ld_acc += log(dcc);
      }
      CLX_T(0, 2);
      if (lane < 16) {
#pragma unroll
        for (int m = 0; m < 16; ++m) ui[c * 16 + m] = w[m];
      }
      CLX_T(0, 3);
      __syncthreads();
      CLX_T(0, 4);
      __syncthreads();
      CLX_T(0, 5);
    }
    const double tot = wave_sum(lane < 16 ? ld_acc : 0.0);
    if (lane == 0) *logdet = 2.0 * tot;
    if (bad) atomicOr(status, bad);
  }
}

// the k loop of NA tiles of one wave:  acc[q] = sum_{k < I} U_kI^T U_kJq.  The loads are unconditional (a lane beyond column N reads a
// valid address of no consequence: its sums reach only outputs that are never stored; rows are always inside, off-diagonal tiles
// exist only in full block rows).  Operands of NS consecutive k-steps wait in NS register sets that take turns: the loads of a set are
// issued right after its MFMAs, NS k-steps before they are needed.
// WITH_P: tile 0 of the group is J = I + 1, whose B operand U_k(I+1) is both operands of the next diagonal block's sum
// accp = sum_{k < I} U_k(I+1)^T U_k(I+1)  (four more MFMAs per k-step, no more loads; okc: this lane's column of that block is inside N).
template <int NA, bool WITH_P, bool PANEL>
__device__ __forceinline__ void chol_left_sums(gcdp_t A, int lda, int I, unsigned aoff, const double* panel, const unsigned (&off)[NA], v4d_p (&acc)[NA],
                                               v4d_p& accp, bool okc) {
  // register sets in flight (a wave with one tile has only its loads to wait for; deeper measured slower: every row starts with NS
  // sets of loads whether it has that many k-steps or not)
  constexpr int NS = NA == 1 ? 6 : (NA == 2 ? 4 : 3);
  const int kmax = max(I - 1, 0);
  double av[PANEL ? 1 : NS][4], bv[NS][NA][4];
  auto fetch = [&](int set, int k) {
    gcdp_t rp = A + (size_t)(16 * min(k, kmax)) * lda;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (!PANEL) av[set][u] = (rp + (size_t)(4 * u) * lda)[aoff];
#pragma unroll
      for (int q = 0; q < NA; ++q) bv[set][q][u] = (rp + (size_t)(4 * u) * lda)[off[q]];
    }
  };
  auto mult = [&](int set, int k) {
    if constexpr (PANEL) {  // the A operand of the whole row waits in LDS (aoff: this lane's element of a 4-row slab there)
#pragma unroll
      for (int u = 0; u < 4; ++u) av[0][u] = panel[(16 * k + 4 * u) * 16 + aoff];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double a = av[PANEL ? 0 : set][u];
#pragma unroll
      for (int q = 0; q < NA; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[set][q][u], acc[q], 0, 0, 0);
      if constexpr (WITH_P) {
        const double v = okc ? bv[set][0][u] : 0.0;
        accp = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, accp, 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NA; ++q) acc[q] = (v4d_p){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int st = 0; st < NS; ++st) fetch(st, st);
  // whole rounds without a branch (a conditional step would make the number of loads in flight unknown to the compiler's wait-count
  // bookkeeping, which then waits for all of them: one memory latency per k-step), then the last I % NS steps
  int k = 0;
  for (; k + NS <= I; k += NS) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      mult(st, k + st);
      fetch(st, k + st + NS);
    }
  }
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (k + st < I) mult(st, k + st);  // (uniform)
}

// the same sum for a wave that has no tile in the row (the last five block rows): accp = sum_{k < I} U_k(I+1)^T U_k(I+1), own loads
__device__ __forceinline__ void chol_left_psum(gcdp_t A, int lda, int I, unsigned offn, bool okc, v4d_p& accp) {
  constexpr int NS = 6;
  const int kmax = max(I - 1, 0);
  double pv[NS][4];
  auto fetch = [&](int set, int k) {
    gcdp_t rp = A + (size_t)(16 * min(k, kmax)) * lda;
#pragma unroll
    for (int u = 0; u < 4; ++u) pv[set][u] = (rp + (size_t)(4 * u) * lda)[offn];
  };
  auto mult = [&](int set) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double v = okc ? pv[set][u] : 0.0;
      accp = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, accp, 0, 0, 0);
    }
  };
#pragma unroll
  for (int st = 0; st < NS; ++st) fetch(st, st);
  int k = 0;
  for (; k + NS <= I; k += NS) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      mult(st);
      fetch(st, k + st + NS);
    }
  }
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (k + st < I) mult(st);  // (uniform)
}

// tile t (J = I + t) of slot s of tile wave hw: the first six tiles go round once, then wave hw = 0 -- which carries the next diagonal
// block's sum with its first tile -- sits out one turn:  hw 0: 1, 12, 18, ...;  hw 1..5: 1 + hw, 6 + hw, 12 + hw, ...
__device__ __forceinline__ int chol_left_tile_of(int hw, int s) { return s == 0 ? 1 + hw : (hw == 0 ? 6 + 6 * s : hw + 6 * s); }

// The role of the six tile waves 1, 2, 3, 5, 6, 7 (helper index hw = 0..5; wave 4 shares its SIMD with wave 0, whose double-precision
// FMAs wait behind any MFMA issued there -- fp64 MFMA and VALU share the DP units -- so it only keeps the barriers company):
// slots s in groups of CL_GS, each group's k loop instantiated for the number of tiles it really has.
template <int CL_MAXS, int CL_GS, bool PANEL>
__device__ __noinline__ void chol_left_tile_role(int N_, gdp_t A_, int lda_, int wv_, const double* ui, double (*dgp)[256], double* nt, double* panels) {
  const int N = uniform_int(N_), lda = uniform_int(lda_), wv = uniform_int(wv_);
  const gdp_t A = uniform_ptr(A_);
  static_assert(CL_MAXS % CL_GS == 0 && CL_GS <= 4, "slots come in whole groups");
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, li = lane & 15;
  const int NBK = (N + 15) >> 4;
  const int hw = wv < 4 ? wv - 1 : wv - 2;
  const unsigned lrow = (unsigned)(kq * lda);
  const int pstride = 16 * (NBK << 4);  // doubles per LDS panel
  v4d_p T[CL_MAXS];
  // the original entries A_IJ of this wave's tiles of block row I: loaded a row ahead (behind the previous row's stores, in front of
  // its second barrier), so that their latency is not part of the row
  auto load_tiles = [&](int I) {
    const int kb = I << 4, ntile = NBK - I;
#pragma unroll
    for (int s = 0; s < CL_MAXS; ++s) {
      const int t = chol_left_tile_of(hw, s);
      if (t < ntile) {  // (uniform)
        const unsigned o = lrow + (unsigned)min(kb + 16 * t + li, N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) T[s][r] = (A + (size_t)(kb + 4 * r) * lda)[o];
      }
    }
  };
  if (wv != 4) load_tiles(0);
  for (int I = 0; I < NBK; ++I) {
    int kb = I << 4;
    asm volatile("" : "+s"(kb));  // (keeps the per-slot addresses from becoming spilled 64-bit induction variables of the row loop)
    const int ntile = NBK - I;
    CLX_T(1, 8);
    if (wv != 4) {
      // the mirror tiles below the diagonal: zeros, as torch.cholesky(upper=True) returns (nothing reads them; stored here, the stores
      // drain behind the k loops instead of in front of the row's second barrier)
#pragma unroll
      for (int s = 0; s < CL_MAXS; ++s) {
        const int t = chol_left_tile_of(hw, s);
        if (t < ntile) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mrow = kb + 16 * t + kq + 4 * r;
#ifndef CLX_NOMIRROR
            if (mrow < N) A[(size_t)mrow * lda + kb + li] = 0.0;
#endif
          }
        }
      }
      CLX_T(1, 11);
      const unsigned aoff = PANEL ? (unsigned)(kq * 16 + li) : lrow + (unsigned)min(kb + li, N - 1);
      const double* panel = panels + (I & 1) * pstride;
      // P_(I+1): with six or more tiles in the row every wave has one, and the sum rides in wave hw = 0's first k loop on the operand
      // that is there anyway; with fewer, the first wave WITHOUT a tile takes it as a job of its own
      const bool okc = kb + 16 + li < N;
      const int p_owner = ntile - 1 >= 6 ? 0 : ntile - 1;  // (the last row, ntile = 1, has no next block: owner 0 finds no tile)
      double an[4] = {0.0, 0.0, 0.0, 0.0};
      if (hw == p_owner && ntile > 1) {
        const int cn = kb + 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = kq + 4 * r, rown = kb + 16 + row;
          an[r] = (cn < N && rown < N) ? A[(size_t)rown * lda + cn] : (row == li ? 1.0 : 0.0);  // identity beyond N
        }
      }
      if (hw == p_owner && p_owner > 0) {
        v4d_p accp = {0.0, 0.0, 0.0, 0.0};
        chol_left_psum(A, lda, I, lrow + (unsigned)(okc ? kb + 16 + li : 0), okc, accp);
#pragma unroll
        for (int r = 0; r < 4; ++r) dgp[(I + 1) & 1][(kq + 4 * r) * 16 + li] = an[r] - accp[r];
      }
#pragma unroll
      for (int g0 = 0; g0 < CL_MAXS; g0 += CL_GS) {
        int na = 0;  // (uniform) tiles of this group
#pragma unroll
        for (int q = 0; q < CL_GS; ++q) na += chol_left_tile_of(hw, g0 + q) < ntile ? 1 : 0;
        if (na > 0) {
          unsigned off[CL_GS];
#pragma unroll
          for (int q = 0; q < CL_GS; ++q) {
            const int t = chol_left_tile_of(hw, g0 + q);
            off[q] = lrow + (unsigned)min(kb + 16 * (t < ntile ? t : 0) + li, N - 1);
          }
          v4d_p accp = {0.0, 0.0, 0.0, 0.0};
          const bool with_p = g0 == 0 && hw == 0 && p_owner == 0;  // (this group holds tile 1)
          auto run = [&](auto na_c) {
            constexpr int NA = decltype(na_c)::value;
            unsigned o[NA];
            v4d_p acc[NA];
#pragma unroll
            for (int q = 0; q < NA; ++q) o[q] = off[q];
            if (with_p)
              chol_left_sums<NA, true, PANEL>(A, lda, I, aoff, panel, o, acc, accp, okc);
            else
              chol_left_sums<NA, false, PANEL>(A, lda, I, aoff, panel, o, acc, accp, okc);
#pragma unroll
            for (int q = 0; q < NA; ++q) T[g0 + q] -= acc[q];
          };
          if (na == 1) run(std::integral_constant<int, 1>());
          if constexpr (CL_GS >= 2) {
            if (na == 2) run(std::integral_constant<int, 2>());
          }
          if constexpr (CL_GS >= 3) {
            if (na == 3) run(std::integral_constant<int, 3>());
          }
          if constexpr (CL_GS >= 4) {
            if (na == 4) run(std::integral_constant<int, 4>());
          }
          if (with_p) {  // P_(I+1)
#pragma unroll
            for (int r = 0; r < 4; ++r) dgp[(I + 1) & 1][(kq + 4 * r) * 16 + li] = an[r] - accp[r];
          }
        }
      }
    }
    if constexpr (PANEL) {
      // the column panel U[0 : 16 I][block column I + 1], the A operand of every tile of the NEXT row, into the other LDS buffer (these rows
      // are final; the last sixteen, U_I(I+1), follow from wave 1 below): one pass by the seven waves here while the chain finishes
      if (I + 1 < NBK) {
        double* pn = panels + ((I + 1) & 1) * pstride;
        const int cnt = kb * 16, c0 = kb + 16;
        for (int base = (wv - 1) * 64 + lane; base < cnt; base += 8 * 448) {
          double v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int idx = base + e * 448, r = idx >> 4, c = c0 + (idx & 15);
            v[e] = (idx < cnt && c < N) ? A[(size_t)r * lda + c] : 0.0;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int idx = base + e * 448;
            if (idx < cnt) pn[idx] = v[e];
          }
        }
      }
    }
    CLX_T(1, 9);
    __syncthreads();
    if (wv != 4) {
      double wa[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wa[u] = ui[(4 * u + kq) * 16 + li];
#pragma unroll
      for (int s = 0; s < CL_MAXS; ++s) {
        const int t = chol_left_tile_of(hw, s);
        if (t < ntile) {  // (uniform)
          const int col = kb + 16 * t + li;
          v4d_p o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int u = 0; u < 4; ++u) o = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[u], T[s][u], o, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = kb + kq + 4 * r;
            if (col < N) A[(size_t)row * lda + col] = o[r];
            if (t == 1) {
              nt[(kq + 4 * r) * 16 + li] = col < N ? o[r] : 0.0;
              if constexpr (PANEL) (panels + ((I + 1) & 1) * pstride)[(kb + kq + 4 * r) * 16 + li] = col < N ? o[r] : 0.0;
            }
          }
        }
      }
      if (I + 1 < NBK) load_tiles(I + 1);
    }
    CLX_T(1, 10);
    __syncthreads();
  }
}

#define CL_LDS_FIXED 1280  // doubles
template <int CL_MAXS, int CL_GS, bool PANEL>
__global__ __launch_bounds__(CM_NT) void chol_left_mfma_kernel(int N, double* __restrict__ A, int lda, double* __restrict__ logdet,
                                                               uint32_t* __restrict__ status, size_t a_stride, size_t ld_stride) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* ui = smem;                                   // [256]    W_I (row m, column r at ui[m * 16 + r])
  double* dg = smem + 256;                             // [256]    T_II on its way from accumulator layout to one column per lane
  double(*dgp)[256] = (double(*)[256])(smem + 512);    // [2][256] P_I (row I & 1), accumulator layout unfolded: [row][column]
  double* nt = smem + 1024;                            // [256]    U_(I-1)I
  double* panels = smem + CL_LDS_FIXED;                // PANEL: two column panels [16 NBK][16] (this row's and the next one's)
  A += (size_t)blockIdx.x * a_stride;
  logdet += (size_t)blockIdx.x * ld_stride;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // two roles with the same two barriers per block row
  if (wv == 0)
    chol_left_diag_role(N, (gdp_t)A, lda, logdet, status, ui, dg, dgp, nt);
  else
    chol_left_tile_role<CL_MAXS, CL_GS, PANEL>(N, (gdp_t)A, lda, wv, ui, dgp, nt, panels);
}

static int launch_chol_mfma(int form, int N, double* A, int lda, double* logdet, uint32_t* status, int batch, size_t a_stride, size_t ld_stride,
                             hipStream_t st) {
  if (form == 1) {
    const size_t fixed_lds = sizeof(double) * CL_LDS_FIXED, panel_lds = fixed_lds + sizeof(double) * 2 * 16 * (size_t)(((N + 15) >> 4) << 4);
    if (N <= 400) {  // 1 + 6 * 4 tiles in the first block row
      MCP_ENSURE_MAX_LDS((chol_left_mfma_kernel<4, 4, true>));
      hipLaunchKernelGGL((chol_left_mfma_kernel<4, 4, true>), dim3(batch), dim3(CM_NT), panel_lds, st, N, A, lda, logdet, status, a_stride, ld_stride);
    } else if (N <= 576) {  // (two panels of 16 x 576 doubles: 144 KiB, + 10 KiB, of the 160)
      MCP_ENSURE_MAX_LDS((chol_left_mfma_kernel<8, 2, true>));
      hipLaunchKernelGGL((chol_left_mfma_kernel<8, 2, true>), dim3(batch), dim3(CM_NT), panel_lds, st, N, A, lda, logdet, status, a_stride, ld_stride);
    } else if (N <= 784) {  // 1 + 6 * 8
      hipLaunchKernelGGL((chol_left_mfma_kernel<8, 2, false>), dim3(batch), dim3(CM_NT), fixed_lds, st, N, A, lda, logdet, status, a_stride, ld_stride);
    } else {  // 1 + 6 * 12 >= 72 (N <= 1152)
      hipLaunchKernelGGL((chol_left_mfma_kernel<12, 2, false>), dim3(batch), dim3(CM_NT), fixed_lds, st, N, A, lda, logdet, status, a_stride, ld_stride);
    }
  } else {
    const size_t lds = sizeof(double) * (256 + (size_t)16 * (N + 16));
    MCP_ENSURE_MAX_LDS(chol_factor_mfma_kernel);
    hipLaunchKernelGGL(chol_factor_mfma_kernel, dim3(batch), dim3(CM_NT), lds, st, N, A, lda, logdet, status, a_stride, ld_stride);
  }
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// Uinv = U^-1 by 16x16 blocks, one workgroup: the diagonal blocks W_I = U_II^-1 in registers (one wave each), then block diagonal
// d = 1, 2, ...:  Uinv[I][J] = - W_I sum_{K = I+1..J} U[I][K] Uinv[K][J],  J = I + d  (every term was finished in an earlier stage).
__global__ __launch_bounds__(CM_NT) void tri_inverse_block_kernel(int N, const double* __restrict__ U, int ldu, double* __restrict__ Ui, int ldi,
                                                                  size_t u_stride, size_t ui_stride) {
  U += (size_t)blockIdx.x * u_stride;
  Ui += (size_t)blockIdx.x * ui_stride;
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NBK = (N + 15) >> 4;
  for (int idx = tid; idx < N * N; idx += CM_NT) {  // zeros below the diagonal (and everywhere a block is not written)
    const int r = idx / N, c = idx - r * N;
    if (c < r) Ui[(size_t)r * ldi + c] = 0.0;
  }
  for (int kbk = wv; kbk < NBK; kbk += CM_NW) {
    const int kb = kbk << 4, nb = min(16, N - kb), c = li;
    double x[16], inv_d[16], w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (r <= c && c < nb) ? U[(size_t)(kb + r) * ldu + kb + c] : (r == c ? 1.0 : 0.0);
#pragma unroll
    for (int r = 0; r < 16; ++r) inv_d[r] = 1.0 / lane_get(x[r], r);
    tri16_inverse(x, inv_d, c, w);
    if (lane < 16 && c < nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r < nb && r <= c) Ui[(size_t)(kb + r) * ldi + kb + c] = w[r];
    }
  }
  __syncthreads();
  for (int d = 1; d < NBK; ++d) {
    for (int I = wv; I + d < NBK; I += CM_NW) {
      const int J = I + d;
      const int col = 16 * J + li;
      v4d_p acc = {0.0, 0.0, 0.0, 0.0};
      for (int K = I + 1; K <= J; ++K) {
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ar = 16 * I + li, ac = 16 * K + 4 * u + kq;   // A[i = li][k]   = U[16 I + i][16 K + k]
          const int br = 16 * K + 4 * u + kq;                     // B[k][j = li]   = Uinv[16 K + k][16 J + j]
          av[u] = (ar < N && ac < N) ? U[(size_t)ar * ldu + ac] : 0.0;
          bv[u] = (br < N && col < N) ? Ui[(size_t)br * ldi + col] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
      }
      // - W_I S: register u of the accumulator is row 4 u + kq of S -- the B operand of step u
      v4d_p out = {0.0, 0.0, 0.0, 0.0};
      double wa[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ar = 16 * I + li, ac = 16 * I + 4 * u + kq;
        wa[u] = (ar < N && ac < N) ? -Ui[(size_t)ar * ldi + ac] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) out = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[u], acc[u], out, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + kq + 4 * r;
        if (row < N && col < N) Ui[(size_t)row * ldi + col] = out[r];
      }
    }
    __syncthreads();
  }
}

// Round 4: U^-1 by BLOCK COLUMNS.  Column J of X = U^-1 depends on U alone:  X[J][J] = W_J = U_JJ^-1,  X[I][J] = - W_I sum_{K = I+1..J} U[I][K]
// X[K][J]  for I = J-1 .. 0 -- a serial chain over I inside a column, no dependence between columns.  So: one launch inverts the
// diagonal blocks (one wave each), a second one gives every block column its own one-wave workgroup, which keeps the column's finished
// blocks in LDS (the B operands of its later products): no workgroup barrier anywhere, NBK x G workgroups in flight instead of one
// (the block-diagonal sweep above: NBK stages of at most NBK / 8 tile products per wave behind a barrier each -- 0.52 ms at N = 400).
__global__ __launch_bounds__(64) void tri_diag_inverse_kernel(int N, const double* __restrict__ U, int ldu, double* __restrict__ Ui, int ldi,
                                                              size_t u_stride, size_t ui_stride) {
  U += (size_t)blockIdx.y * u_stride;
  Ui += (size_t)blockIdx.y * ui_stride;
  const int lane = threadIdx.x, c = lane & 15, kb = (int)blockIdx.x << 4, nb = min(16, N - kb);
  double x[16], inv_d[16], w[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = (r <= c && c < nb) ? U[(size_t)(kb + r) * ldu + kb + c] : (r == c ? 1.0 : 0.0);
#pragma unroll
  for (int r = 0; r < 16; ++r) inv_d[r] = 1.0 / lane_get(x[r], r);
  tri16_inverse(x, inv_d, c, w);
  if (lane < 16 && c < nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r < nb) Ui[(size_t)(kb + r) * ldi + kb + c] = r <= c ? w[r] : 0.0;
  }
}
__global__ __launch_bounds__(64) void tri_inverse_cols_kernel(int N, const double* __restrict__ U, int ldu, double* __restrict__ Ui, int ldi,
                                                              size_t u_stride, size_t ui_stride) {
  extern __shared__ __attribute__((aligned(16))) double xs[];  // [J + 1][16][16]: the finished blocks of this column
  U += (size_t)blockIdx.y * u_stride;
  Ui += (size_t)blockIdx.y * ui_stride;
  const int NBK = (N + 15) >> 4;
  const int J = NBK - 1 - (int)blockIdx.x;  // (the longest columns start first)
  const int lane = threadIdx.x, kq = lane >> 4, li = lane & 15;
  const int col = 16 * J + li;
  // X[J][J] = W_J (written by tri_diag_inverse_kernel)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * J + kq + 4 * r;
    xs[J * 256 + (kq + 4 * r) * 16 + li] = (row < N && col < N) ? Ui[(size_t)row * ldi + col] : 0.0;
  }
  // The chain of the column as one stream of ITEMS: for I = J-1 .. 0 the products S += U[I][K] X[K][J], K = I+1 .. J, then the item that
  // closes the step, X[I][J] = -W_I S.  Every item is one 16x16 A operand from global memory (a block of U, or the diagonal inverse W_I)
  // that does not depend on the chain: they are loaded TC_PF items ahead into a ring of register sets.  The loop body has no memory
  // operation under a branch -- results stay in LDS until the end -- so the compiler can count the loads in flight.  (Round 4 first form:
  // the prefetch sat under wave-uniform branches together with the stores of the results, the wait-count bookkeeping gave up and every
  // item paid an L2 round trip: 78 us at N = 300 for a chain whose MFMAs take 18.)
  constexpr int TC_PF = 6;
  const gcdp_t Ug = (gcdp_t)U, Wg = (gcdp_t)Ui;
  int If = J - 1, Kf = J, Ip = J - 1, Kp = J;  // fetch and process positions; K = J + 1 stands for the closing item of the step
  double buf[TC_PF][4];
  auto fetch = [&](double (&dst)[4]) {
    const bool live = If >= 0;                 // (past the end of the stream: block (0, 0) of U, never used)
    const int I = live ? If : 0;
    const bool closing = live && Kf > J;
    const gcdp_t base = closing ? Wg : Ug;     // (uniform)
    const int ld = closing ? ldi : ldu, kc = live ? (closing ? I : Kf) : 0;
    const unsigned rowoff = (unsigned)((16 * I + li) * ld);  // A[i = li][k]: block row 16 I + i (< N: I < J)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ac = 16 * kc + 4 * u + kq;
      const double v = base[rowoff + (unsigned)min(ac, N - 1)];
      dst[u] = ac < N ? v : 0.0;
    }
    if (live && ++Kf > J + 1) {  // (scalar bookkeeping)
      --If;
      Kf = If + 1;
    }
  };
#pragma unroll
  for (int q = 0; q < TC_PF; ++q) fetch(buf[q]);
  v4d_p acc = {0.0, 0.0, 0.0, 0.0};
  const int items = J * (J + 1) / 2 + J;
  for (int q0 = 0; q0 < items; q0 += TC_PF) {
#pragma unroll
    for (int q = 0; q < TC_PF; ++q) {
      if (Ip >= 0) {  // (wave-uniform; MFMAs and LDS only)
        if (Kp <= J) {
          double bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) bv[u] = xs[Kp * 256 + (4 * u + kq) * 16 + li];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[q][u], bv[u], acc, 0, 0, 0);
          ++Kp;
        } else {  // X[I][J] = - W_I S  (register u of the accumulator is row 4 u + kq of S: the B operand of step u)
          v4d_p out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int u = 0; u < 4; ++u) out = __builtin_amdgcn_mfma_f64_16x16x4f64(-buf[q][u], acc[u], out, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) xs[Ip * 256 + (kq + 4 * r) * 16 + li] = col < N ? out[r] : 0.0;
          acc = (v4d_p){0.0, 0.0, 0.0, 0.0};
          --Ip;
          Kp = Ip + 1;
        }
      }
      fetch(buf[q]);  // this register set: the item TC_PF further on
    }
  }
  // the column, from LDS: blocks 0 .. J-1 (block J is already there), zeros below
  for (int idx = lane; idx < 16 * J * 16; idx += 64) {
    const int row = idx >> 4, c = 16 * J + (idx & 15);
    if (c < N) Ui[(size_t)row * ldi + c] = xs[idx];
  }
  for (int row = 16 * (J + 1) + kq; row < N; row += 4)
    if (col < N) Ui[(size_t)row * ldi + col] = 0.0;
}
// Kinv = Uinv Uinv^T by 16x16 tiles on the matrix cores, one wave per tile (I <= J) of the upper triangle, mirrored into the lower:
//   Kinv[I][J] = sum_{K >= J} Uinv[I][K] Uinv[J][K]^T      (both operands read rows of Uinv: A[i][k] = Ui[16 I + i][16 K + k], B[k][j] = Ui[16 J + j][16 K + k])
// The same column by FOUR waves (round 4, second form).  Step I of the chain is a sum of J - I block products followed by one closing
// product: the products of a step are dealt to the four waves (item m = J - K of the step to wave m mod 4, oldest blocks first, so that the
// one product that needs the block finished in the previous step, K = I + 1, is the last of its wave), partial sums meet in LDS, wave 0
// adds them and closes the step while the others are already in the next one.  Per step two LDS-only barriers (s_waitcnt lgkmcnt(0) +
// s_barrier: the prefetched global operands stay in flight across them; __syncthreads would drain them twice per step).
// The chain of the last column, 171 products at N = 300, becomes 18 steps of ceil(n / 4) products + close.
#define TC4_PF 4
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(256) void tri_inverse_cols4_kernel(int N, const double* __restrict__ U, int ldu, double* __restrict__ Ui, int ldi,
                                                                size_t u_stride, size_t ui_stride) {
  extern __shared__ __attribute__((aligned(16))) double xs[];  // [J + 1][256] finished blocks | [3][256] partial sums of waves 1..3
  U += (size_t)blockIdx.y * u_stride;
  Ui += (size_t)blockIdx.y * ui_stride;
  const int NBK = (N + 15) >> 4;
  const int J = NBK - 1 - (int)blockIdx.x;  // (the longest columns start first)
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = 16 * J + li;
  double* part = xs + (J + 1) * 256;
  if (w == 0) {  // X[J][J] = W_J (written by tri_diag_inverse_kernel)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * J + kq + 4 * r;
      xs[J * 256 + (kq + 4 * r) * 16 + li] = (row < N && col < N) ? Ui[(size_t)row * ldi + col] : 0.0;
    }
  }
  const gcdp_t Ug = (gcdp_t)U, Wg = (gcdp_t)Ui;
  // this wave's stream of items: per step I = J-1 .. 0 the products m = w, w + 4, ... < n = J - I (block K = J - m), then, wave 0 only, the
  // closing item (m = -1 stands for it).  Steps in which the wave has no product (n <= w) have no item.
  int If = J - 1, mf = w;  // fetch position
  auto skip_empty = [&](int& I, int& m) {
    while (I >= 0 && m >= 0 && m >= J - I) {  // no (more) product of mine in step I
      if (w == 0) {
        m = -1;  // the closing item comes next
        return;
      }
      --I;
      m = w;
    }
  };
  skip_empty(If, mf);
  double buf[TC4_PF][4];
  auto fetch = [&](double (&dst)[4]) {
    const bool live = If >= 0;  // (past the end of the stream: block (0, 0) of U, never used)
    const int I = live ? If : 0;
    const bool closing = live && mf < 0;
    const gcdp_t base = closing ? Wg : Ug;  // (uniform)
    const int ld = closing ? ldi : ldu, kc = live ? (closing ? I : J - mf) : 0;
    const unsigned rowoff = (unsigned)((16 * I + li) * ld);  // A[i = li][k]: block row 16 I + i (< N: I < J)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ac = 16 * kc + 4 * u + kq;
      const double v = base[rowoff + (unsigned)min(ac, N - 1)];
      dst[u] = ac < N ? v : 0.0;
    }
    if (live) {  // (scalar bookkeeping)
      if (mf < 0) {
        --If;
        mf = w;
      } else {
        mf += 4;
      }
      skip_empty(If, mf);
    }
  };
#pragma unroll
  for (int q = 0; q < TC4_PF; ++q) fetch(buf[q]);
  lds_barrier();  // X[J][J] is in LDS
  v4d_p acc = {0.0, 0.0, 0.0, 0.0};
  int Ip = J - 1, mp = w;   // process position
  bool newest_seen = false;  // this step's barrier B passed
  // the end of a wave's products of step I: barrier B if it has not passed it yet, partial sum to LDS (waves 1..3), barrier A
  auto end_products = [&]() {
    if (!newest_seen) lds_barrier();  // B: the block of the previous step is in LDS (this wave did not need it)
    if (w > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(w - 1) * 256 + (kq + 4 * r) * 16 + li] = acc[r];
      acc = (v4d_p){0.0, 0.0, 0.0, 0.0};
    }
    lds_barrier();  // A: the partial sums of the step are in LDS
    newest_seen = false;
  };
  // steps without a product of this wave: their two barriers
  auto idle_steps = [&]() {
    while (Ip >= 0 && mp >= 0 && mp >= J - Ip) {
      if (w == 0) {
        end_products();
        mp = -1;
        return;
      }
      end_products();
      --Ip;
      mp = w;
    }
  };
  idle_steps();
  while (Ip >= 0) {
#pragma unroll
    for (int q = 0; q < TC4_PF; ++q) {
      if (Ip >= 0) {  // (wave-uniform; MFMAs, LDS and barriers only)
        if (mp >= 0) {
          const int n = J - Ip, K = J - mp;
          if (mp == n - 1) {  // the product on the block of the previous step
            lds_barrier();    // B
            newest_seen = true;
          }
          double bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) bv[u] = xs[K * 256 + (4 * u + kq) * 16 + li];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[q][u], bv[u], acc, 0, 0, 0);
          mp += 4;
          if (mp >= n) {
            end_products();
            if (w == 0) {
              mp = -1;
            } else {
              --Ip;
              mp = w;
              idle_steps();
            }
          }
        } else {  // wave 0: S = sum of the partial sums;  X[I][J] = - W_I S  (register u of S is the B operand of step u)
          const int n = J - Ip;
#pragma unroll
          for (int pw = 1; pw < 4; ++pw) {
            if (pw < n) {  // (waves beyond the step's products wrote zeros: skip the read)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[r] += part[(pw - 1) * 256 + (kq + 4 * r) * 16 + li];
            }
          }
          v4d_p out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int u = 0; u < 4; ++u) out = __builtin_amdgcn_mfma_f64_16x16x4f64(-buf[q][u], acc[u], out, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) xs[Ip * 256 + (kq + 4 * r) * 16 + li] = col < N ? out[r] : 0.0;
          acc = (v4d_p){0.0, 0.0, 0.0, 0.0};
          --Ip;
          mp = 0;
          idle_steps();
        }
      }
      fetch(buf[q]);  // this register set: the item TC4_PF further on
    }
  }
  lds_barrier();  // the last block of the column is in LDS
  // the column, from LDS: blocks 0 .. J-1 (block J is already there), zeros below
  for (int idx = tid; idx < 16 * J * 16; idx += 256) {
    const int row = idx >> 4, c = 16 * J + (idx & 15);
    if (c < N) Ui[(size_t)row * ldi + c] = xs[idx];
  }
  for (int row = 16 * (J + 1) + (tid >> 4); row < N; row += 16)
    if (col < N) Ui[(size_t)row * ldi + col] = 0.0;
}

__global__ __launch_bounds__(256) void kinv_tiles_kernel(int N, const double* __restrict__ Ui, int ldi, double* __restrict__ Kinv, int ldk,
                                                         size_t ui_stride, size_t k_stride) {
  Ui += (size_t)blockIdx.y * ui_stride;
  Kinv += (size_t)blockIdx.y * k_stride;
  const int NBK = (N + 15) >> 4, nt = NBK * (NBK + 1) / 2;
  const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15;
  const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (t >= nt) return;
  int I = 0, rem = t;
  while (rem >= NBK - I) {
    rem -= NBK - I;
    ++I;
  }
  const int J = I + rem;
  const bool ra = 16 * I + li < N, rb = 16 * J + li < N;
  const double* pa = Ui + (size_t)(ra ? 16 * I + li : 0) * ldi + kq;
  const double* pb = Ui + (size_t)(rb ? 16 * J + li : 0) * ldi + kq;
  v4d_p acc = {0.0, 0.0, 0.0, 0.0};
  for (int K = J; K < NBK; ++K) {
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 16 * K + 4 * u + kq;
      av[u] = (ra && c < N) ? pa[16 * K + 4 * u] : 0.0;
      bv[u] = (rb && c < N) ? pb[16 * K + 4 * u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * I + kq + 4 * r, col = 16 * J + li;
    if (row < N && col < N) {
      Kinv[(size_t)row * ldk + col] = acc[r];
      if (I != J) Kinv[(size_t)col * ldk + row] = acc[r];
    }
  }
}

// Kinv[i][j] = sum_{m >= max(i,j)} Ui[i][m] Ui[j][m]
__global__ void kinv_from_uinv_kernel(int N, const double* __restrict__ Ui, int ldi, double* __restrict__ Kinv, int ldk, size_t ui_stride,
                                      size_t k_stride) {
  Ui += (size_t)blockIdx.z * ui_stride;
  Kinv += (size_t)blockIdx.z * k_stride;
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (i >= N || j >= N) return;
  int m0 = max(i, j);
  const double* a = Ui + (size_t)i * ldi;
  const double* b = Ui + (size_t)j * ldi;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int m = m0;
  for (; m + 3 < N; m += 4) {
    s0 = fma(a[m], b[m], s0);
    s1 = fma(a[m + 1], b[m + 1], s1);
    s2 = fma(a[m + 2], b[m + 2], s2);
    s3 = fma(a[m + 3], b[m + 3], s3);
  }
  for (; m < N; ++m) s0 = fma(a[m], b[m], s0);
  Kinv[(size_t)i * ldk + j] = (s0 + s1) + (s2 + s3);
}

__global__ void gp_alpha_kernel(int N, const double* __restrict__ Kinv, int ldk, const double* __restrict__ Y, double mean,
                                double* __restrict__ alpha) {
  // one wave per row: lanes stride the columns, DPP reduction
  int row = blockIdx.x * (blockDim.x / MCP_WAVE) + (threadIdx.x / MCP_WAVE);
  int lane = threadIdx.x % MCP_WAVE;
  if (row >= N) return;
  double s = 0.0;
  for (int m = lane; m < N; m += MCP_WAVE) s = fma(Kinv[(size_t)row * ldk + m], Y[m] - mean, s);
  s = wave_sum(s);
  if (lane == 0) alpha[row] = s;
}

__global__ void gp_pack_kernel(int N, int D, const double* __restrict__ X, const double* __restrict__ alpha,
                               const double* __restrict__ Kinv, int ldk, int Npad, double* __restrict__ Xt_out,
                               double* __restrict__ X_out, double* __restrict__ alpha_out, double* __restrict__ Kinv_out,
                               double* __restrict__ aX_out) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = gid; idx < (size_t)Npad * Npad; idx += stride) {
    int i = (int)(idx / Npad), j = (int)(idx % Npad);
    Kinv_out[idx] = (i < N && j < N) ? Kinv[(size_t)i * ldk + j] : 0.0;
  }
  for (size_t idx = gid; idx < (size_t)Npad * D; idx += stride) {
    int j = (int)(idx / D), d = (int)(idx % D);
    double v = j < N ? X[(size_t)j * D + d] : 0.0;
    X_out[idx] = v;
    Xt_out[(size_t)d * Npad + j] = v;
  }
  for (size_t idx = gid; idx < (size_t)Npad; idx += stride) alpha_out[idx] = idx < (size_t)N ? alpha[idx] : 0.0;
  if (gid < (size_t)D) {
    double s = 0.0;
    for (int j = 0; j < N; ++j) s = fma(alpha[j], X[(size_t)j * D + gid], s);
    aX_out[gid] = s;
  }
}

// ---------------------------------------------------------------------------------------
// Greedy subset-of-data selection (GP_prior.get_SOD, GP_prior.py:232-257): incremental Cholesky, parallel over the CANDIDATES.
// The reference refactors the subset from scratch for every candidate; the decisions are the same comparisons in exact arithmetic
// (the fixtures record the smallest margin).  A candidate x_c is tested with  var = k(x_c,x_c) - ||w_c||^2,  L w_c = k_S(x_c),
// L L^T = K_S + sigma_n^2 I.  Forward substitution row by row gives  w_c[j] = (k(x_c, x_pj) - sum_{i<j} w_c[i] w_pj[i]) / d_j  with
// w_pj the vector of the j-th accepted point itself and d_j its pivot sqrt(k_pp + sigma_n^2 - ||w_p||^2): component j of EVERY later
// candidate can be formed the moment point j is accepted -- one dot product per candidate, all candidates at once (round 4 walked the
// substitution of one candidate with one wave: 23 ms at N = 300, 162 ms at N = 600, 99 % of pretrain_gp).  One 1024-thread workgroup:
//   W [n][N] (workspace): W[j][c] = w_c[j], coalesced over c;  nrm[c] = ||w_c||^2 so far, kd[c] = k(x_c, x_c)  (workspace tail);
//   per accepted point p: publish w_p (LDS) and its pivot; thread (g, c), c > p: partial dot over the g-th share of j (N <= 1024: the
//   1024 threads are JS = 1024 / roundup64(N) groups of candidates; larger N: one group, several candidates per thread); group 0 adds
//   the shares in group order, appends w_c[n], updates nrm[c] and -- the same thread, the value still in a register -- tests the
//   candidate: the smallest accepted index (LDS atomicMin) is the next point; everything in between was rejected against the subset
//   it was tested with, as in the sequential scan.
// ---------------------------------------------------------------------------------------
constexpr int SOD_NT = 1024;
#define SOD_PIN16(v)                                                                                                                      \
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), \
               "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))
__global__ __launch_bounds__(SOD_NT) void sod_select_kernel(mcp_kernel kn, int N, const double* __restrict__ X, double thr,
                                                            int32_t* __restrict__ idx_out, int32_t* __restrict__ n_out,
                                                            double* __restrict__ W) {
  extern __shared__ __attribute__((aligned(16))) double sod_smem[];
  double* piv = sod_smem;               // [2] 1 / pivot of the accepted point, pivot = sqrt(k_pp + sigma_n^2 - ||w_p||^2)
  double* wp = sod_smem + 2;            // [N] the accepted point's own vector
  double* part = wp + N;                // [SOD_NT] partial dot products of groups 1..JS-1
  int* s_next = (int*)(part + SOD_NT);  // [3] smallest accepted candidate; round k uses slot k % 3
  double* nrm = W + (size_t)N * N;
  double* kd = nrm + N;
  const int tid = threadIdx.x, D = kn.D;
  const int C = (N + MCP_WAVE - 1) / MCP_WAVE * MCP_WAVE;
  const int JS = C <= SOD_NT ? SOD_NT / C : 1;                  // groups that share the j range of a dot product
  const int CPT = C <= SOD_NT ? 1 : (N + SOD_NT - 1) / SOD_NT;  // candidates per thread (JS == 1 then)
  const int grp = C <= SOD_NT ? tid / C : 0, c0 = C <= SOD_NT ? tid % C : tid;
  const double s2 = kern_sigma_n2(kn);
  double kd0 = 0.0, nrm0 = 0.0;  // k(x_c, x_c) and ||w_c||^2 of the thread's first candidate stay in registers; further ones in the workspace
  for (int c = tid; c < N; c += SOD_NT) {
    kd[c] = kern_diag(kn, X + (size_t)c * D, 1);
    nrm[c] = 0.0;
  }
  if (grp == 0 && c0 < N) kd0 = kern_diag(kn, X + (size_t)c0 * D, 1);
  if (tid == 0) s_next[0] = s_next[1] = s_next[2] = N;
  __syncthreads();
  // the scan: point p (the n-th of the subset) has just been accepted; every later candidate gets its component n, is tested, and the
  // smallest index that passes is the next p
  int n = 0, p = 0, slot = 0;
#ifdef SOD_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq;
#define SOD_STAMP(k)                                       \
  {                                                        \
    unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
    st[k] += tn_ - tq;                                     \
    tq = tn_;                                              \
  }
#else
#define SOD_STAMP(k)
#endif
  while (true) {
#ifdef SOD_STAMPS
    tq = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0) {
      idx_out[n] = p;
      s_next[slot == 2 ? 0 : slot + 1] = N;  // next round's slot: last read two rounds ago
    }
    // the pivot's reciprocal, by the thread that holds p's sums (read by everybody behind the barrier below)
    if (grp == 0 && (p - c0) % SOD_NT == 0 && c0 <= p) {
      const bool first = p == c0;
      piv[0] = 1.0 / sqrt((first ? kd0 : kd[p]) + s2 - (first ? nrm0 : nrm[p]));
    }
    for (int j = tid; j < n; j += SOD_NT) wp[j] = W[(size_t)j * N + p];
    // k(x_c, x_p) of the thread's first candidate needs nothing of the above: its loads travel with the gather's
    const double* xp = X + (size_t)p * D;
    const bool own0 = grp == 0 && c0 < N && c0 > p;
    double kcp0 = 0.0;
    if (own0) kcp0 = kern_eval(kn, X + (size_t)c0 * D, 1, xp, 1);
    SOD_STAMP(0)
    __syncthreads();
    SOD_STAMP(1)
    const double rd = piv[0];
    const int per = (n + JS - 1) / JS, j0 = grp * per, j1 = min(n, j0 + per);
    for (int q = 0; q < CPT; ++q) {
      const int c = c0 + q * SOD_NT;
      const bool live = grp < JS && c < N && c > p;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      if (live) {
        const double* wc = W + c;
        int j = j0;
        // (the 16 loads of a batch are pinned in flight together: left to the compiler they go out four at a time with a full wait
        //  behind each group -- an L2 round trip per four rows)
        for (; j + 16 <= j1; j += 16) {
          double v[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) v[t] = wc[(size_t)(j + t) * N];
          SOD_PIN16(v);
#pragma unroll
          for (int t = 0; t < 16; t += 4) {
            a0 = fma(v[t], wp[j + t], a0);
            a1 = fma(v[t + 1], wp[j + t + 1], a1);
            a2 = fma(v[t + 2], wp[j + t + 2], a2);
            a3 = fma(v[t + 3], wp[j + t + 3], a3);
          }
        }
        if (j < j1) {
          double v[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) v[t] = wc[(size_t)min(j + t, j1 - 1) * N];
          SOD_PIN16(v);
#pragma unroll
          for (int t = 0; t < 16; t += 4) {
            a0 = fma(j + t < j1 ? v[t] : 0.0, wp[min(j + t, j1 - 1)], a0);
            a1 = fma(j + t + 1 < j1 ? v[t + 1] : 0.0, wp[min(j + t + 1, j1 - 1)], a1);
            a2 = fma(j + t + 2 < j1 ? v[t + 2] : 0.0, wp[min(j + t + 2, j1 - 1)], a2);
            a3 = fma(j + t + 3 < j1 ? v[t + 3] : 0.0, wp[min(j + t + 3, j1 - 1)], a3);
          }
        }
      }
      double dot = (a0 + a1) + (a2 + a3);
      SOD_STAMP(2)
      if (JS > 1) {
        if (grp > 0) part[tid] = dot;
        __syncthreads();
        SOD_STAMP(3)
        if (grp == 0)
          for (int g = 1; g < JS; ++g) dot += part[g * C + c0];
      }
      bool pass = false;
      if (live && grp == 0) {
        const double kcp = q == 0 ? kcp0 : kern_eval(kn, X + (size_t)c * D, 1, xp, 1);
        const double w = (kcp - dot) * rd;
        W[(size_t)n * N + c] = w;
        const double nc = fma(w, w, q == 0 ? nrm0 : nrm[c]), kc = q == 0 ? kd0 : kd[c];
        if (q == 0)
          nrm0 = nc;
        else
          nrm[c] = nc;
        pass = sqrt(kc - nc) > thr;
      }
      // candidates grow with the lane: the wave's first lane that passed speaks for it (one LDS atomic per wave, not one per lane)
      const unsigned long long bal = __ballot(pass);
      if (pass && (bal & ((1ull << (tid & 63)) - 1ull)) == 0ull) atomicMin(&s_next[slot], c);
    }
    SOD_STAMP(4)
    __syncthreads();
    SOD_STAMP(5)
    n += 1;
    p = s_next[slot];
    slot = slot == 2 ? 0 : slot + 1;
    if (p >= N) break;
  }
  if (tid == 0) *n_out = n;
#ifdef SOD_STAMPS
  if (tid == 256 || tid == 0 || tid == 256 + C) {  // into the unused tail rows of W (n < N rows are written when anything was rejected)
    unsigned long long* o = (unsigned long long*)(W + (size_t)(N - 1) * N) + (tid == 0 ? 0 : (tid == 256 ? 8 : 16));
    for (int k = 0; k < 6; ++k) o[k] = st[k];
    o[6] = (unsigned long long)n;
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---------------------------------------------------------------------------------------
// The same selection ACROSS workgroups (round 5, N >= SODM_MIN = 256): one 1024-thread workgroup per 64 candidates (wave 0 holds them, one per lane,
// with k(x_c, x_c) and ||w_c||^2 in registers; the 16 waves share the j range of the block's dot products).  At N = 600 the one-workgroup kernel
// spends 18.7 us per accepted point streaming W (2.9 MB) through ONE CU's L2 path; here every workgroup streams its own 64 columns.
// Per accepted point there is ONE exchange: every workgroup publishes its own first passing candidate -- index, reciprocal pivot and the
// candidate's vector w_c[0 .. n] -- as tagged 8-byte granules (tag = round + 1, value = half a double; relaxed agent-scope stores, no fences: W
// stays in its owner's L2), reads the G indices, takes the smallest and reads that workgroup's vector.  Two granule sets by round parity: a
// workgroup can be at most one round ahead (it needs every index of a round to leave it).
//   gx: hdr[g][parity][4] = {index, 1/pivot lo, hi, -} | vec[g][parity][N][2]
// Every poll is bounded; a partner that never arrives ends the kernel with *n_out = -1 (the grid must be resident: G <= 64 workgroups).
// ---------------------------------------------------------------------------------------
#ifdef SODM_MIN_OVERRIDE  // (experiment builds: where the multi-workgroup form starts to pay)
constexpr int SODM_MIN = SODM_MIN_OVERRIDE;
#else
constexpr int SODM_MIN = 256;  // (N = 300: 1.32 -> 0.83 ms; below, the one-workgroup kernel's 12 k cycles per point are at the exchange's level)
#endif
constexpr unsigned SODM_SPIN = 1u << 22;
typedef unsigned long long __attribute__((address_space(1))) * sod_gu64_t;
__device__ __forceinline__ void sod_put(sod_gu64_t g, unsigned tag, unsigned v) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sod_put_double(sod_gu64_t g, unsigned tag, double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  sod_put(g, tag, (unsigned)b);
  sod_put(g + 1, tag, (unsigned)(b >> 32));
}
// both halves of a double, re-read until both carry the tag (false: the spin limit ran out)
__device__ __forceinline__ bool sod_get_double(sod_gu64_t g, unsigned tag, double& x) {
  for (unsigned spins = 0; spins < SODM_SPIN; ++spins) {
    const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag) {
      x = __longlong_as_double((long long)((b << 32) | (a & 0xffffffffull)));
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
static size_t sod_multi_granules(int N) { const size_t G = ((size_t)N + 63) / 64; return G * 2 * 4 + G * 2 * (size_t)N * 2; }
__global__ __launch_bounds__(SOD_NT) void sod_select_multi_kernel(mcp_kernel kn, int N, const double* __restrict__ X, double thr,
                                                                  int32_t* __restrict__ idx_out, int32_t* __restrict__ n_out,
                                                                  double* __restrict__ W, unsigned long long* __restrict__ gx, int LR,
                                                                  const double* __restrict__ Kg) {
  extern __shared__ __attribute__((aligned(16))) double sod_smem[];
  double* wp = sod_smem;          // [N] the accepted point's vector
  double* part = wp + N;          // [SOD_NT] the 16 waves' shares of the block's dot products
  double* wnew = part + SOD_NT;   // [64] component n of the block's candidates
  double* shd = wnew + 64;        // [2] 1 / pivot of the accepted point
  int* shi = (int*)(shd + 2);     // [4] my first passing candidate | the accepted one | abort | the round's verdict on the predicted owner
  double* wl = shd + 4;           // [LR][64] the first LR rows of the block's columns of W (what the LDS has room for): LDS latency instead of L2's
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6, D = kn.D;
  const int G = gridDim.x, g = blockIdx.x;
  const int c = 64 * g + lane;
  const double s2 = kern_sigma_n2(kn);
  sod_gu64_t hdr = (sod_gu64_t)gx;
  sod_gu64_t vec = (sod_gu64_t)gx + (size_t)G * 2 * 4;
  double kd0 = 0.0, nrm0 = 0.0;
  if (grp == 0 && c < N) kd0 = kern_diag(kn, X + (size_t)c * D, 1);
  if (tid == 0) {
    shd[0] = 1.0 / sqrt(kern_diag(kn, X, 1) + s2);  // the first point is always kept (GP_prior.py:240): its pivot needs no exchange
    shi[2] = 0;
  }
  __syncthreads();
  int n = 0, p = 0;
  bool failed = false;
#ifdef SODM_STAMPS
  unsigned long long sst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stq = clock64();
#define SODM_STAMP(k)                      \
  {                                        \
    unsigned long long tn_ = clock64();    \
    sst[k] += tn_ - stq;                   \
    stq = tn_;                             \
  }
#else
#define SODM_STAMP(k)
#endif
  while (true) {
    const unsigned tag = (unsigned)n + 1u;
    const int par = n & 1;
    if (g == 0 && tid == 0) idx_out[n] = p;
    const double rd = shd[0];
    const bool live = c < N && c > p;
    // this wave's share of the dot products  sum_{j < n} W[j][c] w_p[j]
    const int per = (n + 15) >> 4, j0 = grp * per, j1 = min(n, j0 + per);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // k(x_c, x_p) of wave 0's candidates: row p of the Gram matrix the launch function built beforehand (cov_build_kernel, all CUs; evaluated
    // here -- 2 D strided loads and an exp per lane -- it was 6 k of the 19 k cycles a round took, on the one wave every other waits for)
    double kcp = 0.0;
    if (grp == 0 && live) kcp = Kg[(size_t)p * N + c];
    if (live) {
      const double* wc = W + c;
      const int jl = min(j1, LR);  // rows below LR: from the LDS copy
      int j = j0;
      for (; j + 4 <= jl; j += 4) {
        a0 = fma(wl[j * 64 + lane], wp[j], a0);
        a1 = fma(wl[(j + 1) * 64 + lane], wp[j + 1], a1);
        a2 = fma(wl[(j + 2) * 64 + lane], wp[j + 2], a2);
        a3 = fma(wl[(j + 3) * 64 + lane], wp[j + 3], a3);
      }
      for (; j < jl; ++j) a0 = fma(wl[j * 64 + lane], wp[j], a0);
      for (; j + 16 <= j1; j += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = wc[(size_t)(j + t) * N];
        SOD_PIN16(v);
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
          a0 = fma(v[t], wp[j + t], a0);
          a1 = fma(v[t + 1], wp[j + t + 1], a1);
          a2 = fma(v[t + 2], wp[j + t + 2], a2);
          a3 = fma(v[t + 3], wp[j + t + 3], a3);
        }
      }
      if (j < j1) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = wc[(size_t)min(j + t, j1 - 1) * N];
        SOD_PIN16(v);
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
          a0 = fma(j + t < j1 ? v[t] : 0.0, wp[min(j + t, j1 - 1)], a0);
          a1 = fma(j + t + 1 < j1 ? v[t + 1] : 0.0, wp[min(j + t + 1, j1 - 1)], a1);
          a2 = fma(j + t + 2 < j1 ? v[t + 2] : 0.0, wp[min(j + t + 2, j1 - 1)], a2);
          a3 = fma(j + t + 3 < j1 ? v[t + 3] : 0.0, wp[min(j + t + 3, j1 - 1)], a3);
        }
      }
    }
    part[tid] = (a0 + a1) + (a2 + a3);
    SODM_STAMP(0)
    __syncthreads();
    SODM_STAMP(1)
    if (grp == 0) {  // wave 0: the shares in wave order, component n, the test, my first passing candidate
      double dot = part[lane];
#pragma unroll
      for (int q = 1; q < 16; ++q) dot += part[q * 64 + lane];
      bool pass = false;
      double w = 0.0;
      if (live) {
        w = (kcp - dot) * rd;
        W[(size_t)n * N + c] = w;
        nrm0 = fma(w, w, nrm0);
        pass = sqrt(kd0 - nrm0) > thr;
      }
      wnew[lane] = w;
      if (n < LR) wl[n * 64 + lane] = w;
      const unsigned long long bal = __ballot(pass);
      const int first = bal ? (int)__builtin_ctzll(bal) : -1;
      sod_gu64_t h = hdr + ((size_t)g * 2 + par) * 4;
      if (lane == (first < 0 ? 0 : first)) {
        if (first >= 0) sod_put_double(h + 1, tag, 1.0 / sqrt(kd0 + s2 - nrm0));
        sod_put(h, tag, (unsigned)(first < 0 ? N : 64 * g + first));
        shi[0] = first < 0 ? N : 64 * g + first;
      }
    }
    SODM_STAMP(2)
    __syncthreads();
    SODM_STAMP(3)
    {  // my candidate's vector w_c[0 .. n] (row n from LDS: it was formed a moment ago)
      const int cg = shi[0];
      if (cg < N) {
        sod_gu64_t vm = vec + (((size_t)g * 2 + par) * N) * 2;
        for (int j = tid; j <= n; j += SOD_NT) sod_put_double(vm + 2 * (size_t)j, tag, j == n ? wnew[cg - 64 * g] : (j < LR ? wl[j * 64 + cg - 64 * g] : W[(size_t)j * N + cg]));
      }
    }
    // Whose candidate wins is known before the indices are: the live candidates are c > p, so the lowest block that has any is block (p + 1) / 64, and
    // if that block has a passing candidate it is the smallest.  Waves 1-15 therefore ask for THAT workgroup's vector while wave 0 collects the
    // indices -- one round trip instead of two; when the block had no passing candidate (it publishes no vector) wave 0's verdict in LDS ends their
    // wait and everybody reads the true owner's.
    SODM_STAMP(4)
    const int gpred = min(G - 1, (p + 1) >> 6);
    volatile int* verdict = shi + 3;
    if (grp == 0) {  // every workgroup's index of this round; the smallest is the next point
      int best = N, mine = N;
      double rmine = 0.0;  // (the pivot travels with the index: the lane that polled the winner hands it on)
      for (int q = lane; q < G; q += 64) {
        sod_gu64_t h = hdr + ((size_t)q * 2 + par) * 4;
        bool ok = false;
        for (unsigned spins = 0; spins < SODM_SPIN && !ok; ++spins) {
          const unsigned long long a = __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long b0 = __hip_atomic_load(h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long b1 = __hip_atomic_load(h + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int cq = (int)(unsigned)a;
          ok = (unsigned)(a >> 32) == tag && (cq >= N || ((unsigned)(b0 >> 32) == tag && (unsigned)(b1 >> 32) == tag));
          if (ok) {
            if (cq < mine) {
              mine = cq;
              rmine = __longlong_as_double((long long)((b1 << 32) | (b0 & 0xffffffffull)));
            }
          } else {
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (!ok) shi[2] = 1;
      }
      best = mine;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o));
      if (best < N && mine == best) shd[0] = rmine;
      if (lane == 0) {
        shi[1] = best;
        *verdict = (int)(2u * tag + ((best < N && (best >> 6) == gpred) ? 0u : 1u));  // (even: the predicted owner it is)
      }
    } else {
      sod_gu64_t vq = vec + (((size_t)gpred * 2 + par) * N) * 2;
      for (int j = tid - 64; j <= n; j += SOD_NT - 64) {
        bool done = false;
        for (unsigned spins = 0; spins < SODM_SPIN && !done; ++spins) {
          const unsigned long long x0 = __hip_atomic_load(vq + 2 * (size_t)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long x1 = __hip_atomic_load(vq + 2 * (size_t)j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(x0 >> 32) == tag && (unsigned)(x1 >> 32) == tag) {
            wp[j] = __longlong_as_double((long long)((x1 << 32) | (x0 & 0xffffffffull)));
            done = true;
          } else if (*verdict == (int)(2u * tag + 1u)) {
            done = true;  // not the predicted owner (or nobody): no vector will come from there
          } else {
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (!done) shi[2] = 1;
      }
    }
    SODM_STAMP(5)
    __syncthreads();
    SODM_STAMP(6)
    const int pn = shi[1];
    const bool hit = *verdict == (int)(2u * tag);
    if (shi[2]) {
      failed = true;
      break;
    }
    n += 1;
    if (pn >= N) break;
    {  // after a miss: the accepted point's vector from the workgroup that owns it
      const int gw = pn >> 6;
      sod_gu64_t vw = vec + (((size_t)gw * 2 + par) * N) * 2;
      if (!hit) {
        for (int j = tid; j < n; j += SOD_NT) {
          double x = 0.0;
          if (!sod_get_double(vw + 2 * (size_t)j, tag, x)) shi[2] = 1;
          wp[j] = x;
        }
      }
    }
    p = pn;
    __syncthreads();
    SODM_STAMP(7)
    if (shi[2]) {
      failed = true;
      break;
    }
  }
#ifdef SODM_STAMPS
  if (g == gridDim.x / 2 && (tid == 0 || tid == 512)) {  // (wave 0's lane 0 and a thread of wave 8, workgroup G / 2: into the one-workgroup kernel's tail of the workspace)
    unsigned long long* o = (unsigned long long*)(W + (size_t)N * N) + (tid == 0 ? 0 : 8);
    for (int k = 0; k < 8; ++k) o[k] = sst[k];
  }
#endif
  if (g == 0 && tid == 0) *n_out = failed ? -1 : n;
}

// ---------------------------------------------------------------------------------------
// Marginal-likelihood gradient (GP_prior.fit_model's objective, Gaussian_likelihood.py:15-24):
//   L = 1/2 (r^T Kinv r + logdet K),   dL/dtheta = 1/2 sum_ij Wm_ij dK_ij/dtheta,   Wm = Kinv - alpha alpha^T.
// One workgroup per row i: (1) threads over j stage Wm_ij, Wm_ij*kse_ij and the two MPK_2 factor values in LDS,
// (2) one thread per hyper-parameter sums over j.  slab[i][p]; nll_colsum_kernel adds the rows in a fixed order.
// Parameter layout (NP = 4D+3): [0,D) log lengthscales | D log lambda | D+1 noise (1/2 tr Wm) | [D+2,2D+3) MPK_1 (D+1)
//                               | [2D+3,3D+3) MPK_2 factor 0 | [3D+3,4D+3) MPK_2 factor 1
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void nll_grad_row(const mcp_kernel& kn, int N, const double* __restrict__ X, const double* __restrict__ Kinv, int ldk,
                                             const double* __restrict__ alpha, double* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* wm = sm;          // [N] Wm_ij
  double* wk = sm + N;      // [N] Wm_ij * kse_ij
  double* fa = sm + 2 * N;  // [N] MPK_2 factor A_ij
  double* fb = sm + 3 * N;  // [N] MPK_2 factor B_ij
  const int i = blockIdx.x, tid = threadIdx.x, D = kn.D;
  const double* xi = X + (size_t)i * D;
  const double ai = alpha[i];
  for (int j = tid; j < N; j += 256) {
    const double* xj = X + (size_t)j * D;
    double dist = 0.0, A = 0.0, Bv = 0.0;
    for (int d = 0; d < D; ++d) {
      double r = (xi[d] - xj[d]) * kn.inv_ls[d];
      dist = fma(r, r, dist);
      if (kn.poly_deg >= 2) {
        double xx = xi[d] * xj[d];
        A = fma(kn.w20[d], xx, A);
        Bv = fma(kn.w21[d], xx, Bv);
      }
    }
    double w = Kinv[(size_t)i * ldk + j] - ai * alpha[j];
    wm[j] = w;
    wk[j] = w * kern_lambda(kn) * exp(-dist);
    fa[j] = A;
    fb[j] = Bv;
  }
  __syncthreads();
  // (2) the sums over j, one per hyper-parameter: the 256 threads as nseg segments of NPpad >= NP lanes, segment s taking j = s, s + nseg, ...
  // (one thread per parameter left 157 of the 256 idle at D = 24 and walked 400 dependent global loads each); partial sums meet in LDS
  // and are added in segment order
  const int NP = 4 * D + 3;
  double* red = sm + 4 * N;  // [256]
  int NPpad = 32;
  while (NPpad < NP) NPpad <<= 1;
  if (NPpad <= 256) {
    const int nseg = 256 / NPpad, p = tid % NPpad, seg = tid / NPpad;
    double s0 = 0.0, s1 = 0.0;
    auto over_j = [&](auto term) {  // two accumulators: consecutive loads do not wait for each other's FMA
      int j = seg;
      for (; j + nseg < N; j += 2 * nseg) {
        s0 += term(j);
        s1 += term(j + nseg);
      }
      if (j < N) s0 += term(j);
    };
    if (p < D) {  // d/d log l_p :  kse * 2 (dx/l)^2
      const double il2 = kn.inv_ls[p] * kn.inv_ls[p], xip = xi[p];
      over_j([&](int j) {
        const double dx = xip - X[(size_t)j * D + p];
        return wk[j] * (2.0 * dx * dx * il2);
      });
    } else if (p == D) {  // d/d log lambda
      over_j([&](int j) { return wk[j]; });
    } else if (p == D + 1) {  // 1/2 tr Wm (the caller multiplies by d sigma_n^2 / d sigma_n_log)
      s0 = seg == 0 ? wm[i] : 0.0;
    } else if (p < 2 * D + 3) {  // MPK_1, feature e (e == D: the offset feature)
      const int e = p - (D + 2);
      if (kn.poly_deg >= 1) {
        const double c = 2.0 * kn.w1[e] * (e < D ? xi[e] : 1.0);
        if (e < D)
          over_j([&](int j) { return wm[j] * (c * X[(size_t)j * D + e]); });
        else
          over_j([&](int j) { return wm[j] * c; });
      }
    } else if (p < 3 * D + 3) {  // MPK_2 factor 0 parameter e: 2 w20_e x_ie x_je * B_ij
      const int e = p - (2 * D + 3);
      if (kn.poly_deg >= 2) {
        const double c = 2.0 * kn.w20[e] * xi[e];
        over_j([&](int j) { return (wm[j] * fb[j]) * (c * X[(size_t)j * D + e]); });
      }
    } else if (p < NP) {  // MPK_2 factor 1 parameter e: 2 w21_e x_ie x_je * A_ij
      const int e = p - (3 * D + 3);
      if (kn.poly_deg >= 2) {
        const double c = 2.0 * kn.w21[e] * xi[e];
        over_j([&](int j) { return (wm[j] * fa[j]) * (c * X[(size_t)j * D + e]); });
      }
    }
    red[tid] = s0 + s1;
    __syncthreads();
    if (tid < NP) {
      double s = 0.0;
      for (int sg = 0; sg < nseg; ++sg) s += red[sg * NPpad + tid];
      slab[(size_t)i * NP + tid] = 0.5 * s;
    }
    return;
  }
  for (int p = tid; p < NP; p += 256) {  // (more than 256 hyper-parameters: D > 63 -- beyond MCP_MAX_GPDIM today)
    double s = 0.0;
    if (p < D) {
      const double il2 = kn.inv_ls[p] * kn.inv_ls[p];
      for (int j = 0; j < N; ++j) {
        double dx = xi[p] - X[(size_t)j * D + p];
        s = fma(wk[j], 2.0 * dx * dx * il2, s);
      }
    } else if (p == D) {
      for (int j = 0; j < N; ++j) s += wk[j];
    } else if (p == D + 1) {
      s = wm[i];
    } else if (p < 2 * D + 3) {
      const int e = p - (D + 2);
      if (kn.poly_deg >= 1) {
        const double we = 2.0 * kn.w1[e];
        const double pie = e < D ? xi[e] : 1.0;
        for (int j = 0; j < N; ++j) s = fma(wm[j], we * pie * (e < D ? X[(size_t)j * D + e] : 1.0), s);
      }
    } else if (p < 3 * D + 3) {
      const int e = p - (2 * D + 3);
      if (kn.poly_deg >= 2) {
        const double we = 2.0 * kn.w20[e] * xi[e];
        for (int j = 0; j < N; ++j) s = fma(wm[j] * fb[j], we * X[(size_t)j * D + e], s);
      }
    } else {
      const int e = p - (3 * D + 3);
      if (kn.poly_deg >= 2) {
        const double we = 2.0 * kn.w21[e] * xi[e];
        for (int j = 0; j < N; ++j) s = fma(wm[j] * fa[j], we * X[(size_t)j * D + e], s);
      }
    }
    slab[(size_t)i * NP + p] = 0.5 * s;
  }
}
__global__ __launch_bounds__(256) void nll_grad_kernel(mcp_kernel kn, int N, const double* __restrict__ X, const double* __restrict__ Kinv,
                                                       int ldk, const double* __restrict__ alpha, double* __restrict__ slab) {
  nll_grad_row(kn, N, X, Kinv, ldk, alpha, slab);
}

__global__ void nll_colsum_kernel(int rows, int cols, const double* __restrict__ slab, double* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int r = 0;
  for (; r + 3 < rows; r += 4) {
    s0 += slab[(size_t)r * cols + c];
    s1 += slab[(size_t)(r + 1) * cols + c];
    s2 += slab[(size_t)(r + 2) * cols + c];
    s3 += slab[(size_t)(r + 3) * cols + c];
  }
  for (; r < rows; ++r) s0 += slab[(size_t)r * cols + c];
  out[c] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static inline bool kernel_ok(const mcp_kernel* k) {
  return k && k->D > 0 && k->D <= MCP_MAX_GPDIM && k->poly_deg >= 0 && k->poly_deg <= 2 && k->inv_ls &&
         (k->poly_deg < 1 || k->w1) && (k->poly_deg < 2 || (k->w20 && k->w21));
}

extern "C" int mcp_cov_build(const mcp_kernel* kern, int N1, const double* X1, int N2, const double* X2, int add_noise, double* K,
                             int ldk, void* stream) {
  if (!kernel_ok(kern) || !X1 || !X2 || !K || N1 <= 0 || N2 <= 0 || ldk < N2) return MCP_ERR_ARG;
  dim3 grid((N2 + 255) / 256, N1);
  hipLaunchKernelGGL(cov_build_kernel, grid, dim3(256), 0, (hipStream_t)stream, *kern, N1, X1, N2, X2, add_noise, K, ldk);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_cov_diag(const mcp_kernel* kern, int N, const double* X, int add_noise, double* diag, void* stream) {
  if (!kernel_ok(kern) || !X || !diag || N <= 0) return MCP_ERR_ARG;
  hipLaunchKernelGGL(cov_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *kern, N, X, add_noise, diag);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// U^-1 and K^-1 = U^-1 U^-T of `batch` matrices (strides in doubles): diagonal blocks, block columns, tiles -- three launches
static int launch_inverse_mfma(int N, const double* U, int ldu, double* Ui, int ldi, double* Kinv, int ldk, int batch, size_t u_stride,
                               size_t ui_stride, size_t k_stride, hipStream_t st, bool one_wave_columns = false) {
  const int NBK = (N + 15) >> 4, nt = NBK * (NBK + 1) / 2;
  hipLaunchKernelGGL(tri_diag_inverse_kernel, dim3(NBK, batch), dim3(64), 0, st, N, U, ldu, Ui, ldi, u_stride, ui_stride);
  MCP_LAUNCH_CHECK();
  if (one_wave_columns) {
    MCP_ENSURE_MAX_LDS(tri_inverse_cols_kernel);
    hipLaunchKernelGGL(tri_inverse_cols_kernel, dim3(NBK, batch), dim3(64), sizeof(double) * 256 * (size_t)NBK, st, N, U, ldu, Ui, ldi, u_stride,
                       ui_stride);
  } else {
    MCP_ENSURE_MAX_LDS(tri_inverse_cols4_kernel);
    hipLaunchKernelGGL(tri_inverse_cols4_kernel, dim3(NBK, batch), dim3(256), sizeof(double) * 256 * (size_t)(NBK + 3), st, N, U, ldu, Ui, ldi,
                       u_stride, ui_stride);
  }
  MCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(kinv_tiles_kernel, dim3((nt + 3) / 4, batch), dim3(256), 0, st, N, Ui, ldi, Kinv, ldk, ui_stride, k_stride);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// ---------------------------------------------------------------------------------------
// Factorisation across workgroups (round 5; N >= CHB_MIN): right-looking by panels of CHB_NB rows.  Per panel k
//   1. U_kk = chol(A_kk)                     the one-workgroup left-looking kernel above on the diagonal block
//   2. W = U_kk^-1                           tri_diag_inverse + tri_inverse_cols4 on that block (8 block columns)
//   3. U_kj = W^T A_kj   (j > k, in place)   chol_panel_solve_kernel:  one wave per 16 columns, the whole 128-row column slab in registers
//   4. A_ij -= U_ki^T U_kj  (k < i <= j)     chol_trailing_update_kernel: one wave per 32 x 32 tile, 64 x 64 per workgroup, upper tiles only
// Every product is  C[m][n] = sum_k P[k][m] Q[k][n]  on v_mfma_f64_16x16x4_f64 (A operand lane (kq, li) = P[k0 + kq][m0 + li], B likewise from Q,
// result register r = C[m0 + kq + 4 r][n0 + li]), operands straight from L2.  The one-workgroup kernel (chain-bound: 8.2 k cycles per 16 rows,
// one CU) took 3.1 ms at N = 1000 and stopped at 1152 rows; the panels' chain is 8 block rows each and the O(N^3) part runs on the whole chip.
// Scratch: mcp_chol_factor takes no workspace, but the strictly-lower triangle of A is output-zero by contract -- W_k lives in block (k, 0) of it
// ((1, 0) for k = 0), the panels' logdet terms in its last row; chol_finish_kernel sums those and zeroes the triangle.
// ---------------------------------------------------------------------------------------
#define CHB_NB 128
#define CHB_MIN 600
__global__ __launch_bounds__(256) void chol_panel_solve_kernel(int R, const double* __restrict__ W, int ldw, double* __restrict__ B, int ldb) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15;
  const int col = ((int)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + li;
  const bool ok = col < R;
  double b[CHB_NB / 4];
#pragma unroll
  for (int s = 0; s < CHB_NB / 4; ++s) b[s] = ok ? B[(size_t)(4 * s + kq) * ldb + col] : 0.0;
  v4d_p out[CHB_NB / 16];
#pragma unroll
  for (int mt = 0; mt < CHB_NB / 16; ++mt) {
    v4d_p acc = {0.0, 0.0, 0.0, 0.0};
    // (W is upper triangular: rows k > 16 mt + 15 of its columns [16 mt, 16 mt + 16) are zero -- and were never written)
#pragma unroll
    for (int s = 0; s < 4 * (mt + 1); ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(size_t)(4 * s + kq) * ldw + 16 * mt + li], b[s], acc, 0, 0, 0);
    out[mt] = acc;
  }
  if (ok) {
#pragma unroll
    for (int mt = 0; mt < CHB_NB / 16; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) B[(size_t)(16 * mt + kq + 4 * r) * ldb + col] = out[mt][r];
  }
}

__global__ __launch_bounds__(256) void chol_trailing_update_kernel(int R, const double* __restrict__ Up, int ldu, double* __restrict__ C, int ldc) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (ti > tj) return;  // (the upper tiles only)
  const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15, w = threadIdx.x >> 6;
  const int m0 = 64 * ti + 32 * (w >> 1), n0 = 64 * tj + 32 * (w & 1);
  if (m0 >= R || n0 >= R) return;
  const int ma = min(m0 + li, R - 1), mb = min(m0 + 16 + li, R - 1), na = min(n0 + li, R - 1), nb = min(n0 + 16 + li, R - 1);  // (clamped: such columns are not stored)
  v4d_p acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = v4d_p{0.0, 0.0, 0.0, 0.0};
  for (int s0 = 0; s0 < CHB_NB / 4; s0 += 8) {  // 8 k-steps per batch: 32 loads in flight, then 32 MFMAs
    double a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double* row = Up + (size_t)(4 * (s0 + u) + kq) * ldu;
      a0[u] = row[ma];
      a1[u] = row[mb];
      b0[u] = row[na];
      b1[u] = row[nb];
    }
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a0[4]), "+v"(a0[5]), "+v"(a0[6]), "+v"(a0[7]), "+v"(a1[0]), "+v"(a1[1]),
                 "+v"(a1[2]), "+v"(a1[3]), "+v"(a1[4]), "+v"(a1[5]), "+v"(a1[6]), "+v"(a1[7]));
    asm volatile("" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b0[4]), "+v"(b0[5]), "+v"(b0[6]), "+v"(b0[7]), "+v"(b1[0]), "+v"(b1[1]),
                 "+v"(b1[2]), "+v"(b1[3]), "+v"(b1[4]), "+v"(b1[5]), "+v"(b1[6]), "+v"(b1[7]));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + kq + 4 * r, col = n0 + 16 * j + li;
        if (row < R && col < R) C[(size_t)row * ldc + col] -= acc[i][j][r];
      }
}

// logdet = sum of the panels' terms (parked in the last row of the lower triangle), then the strictly-lower triangle back to zero
__global__ void chol_finish_kernel(int N, double* __restrict__ A, int lda, int np, double* __restrict__ logdet) {
  __shared__ double tot;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < np; ++k) s += A[(size_t)(N - 1) * lda + k];
    tot = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) *logdet = tot;
  __syncthreads();  // (every block has read the terms of the last row before any block zeroes it: the last row belongs to the LAST block)
  for (int row = blockIdx.x; row < N; row += gridDim.x) {
    if (row == N - 1 && gridDim.x > 1) continue;  // (left to the tail kernel)
    for (int c = threadIdx.x; c < row; c += blockDim.x) A[(size_t)row * lda + c] = 0.0;
  }
}
__global__ void chol_finish_tail_kernel(int N, double* __restrict__ A, int lda) {
  for (int c = threadIdx.x; c < N - 1; c += blockDim.x) A[(size_t)(N - 1) * lda + c] = 0.0;
}

static int launch_chol_blocked(int N, double* A, int lda, double* logdet, uint32_t* status, hipStream_t st) {
  const int NB = CHB_NB, np = (N + NB - 1) / NB;
  if (N < 2 * NB + 1 || np > NB) return MCP_ERR_LIMIT;
  for (int k = 0; k < np; ++k) {
    const int k0 = k * NB, nb = N - k0 < NB ? N - k0 : NB, R = N - k0 - nb;
    double* Akk = A + (size_t)k0 * lda + k0;
    double* ldk = A + (size_t)(N - 1) * lda + k;
    if (nb > 16) {
      const int rc = launch_chol_mfma(1, nb, Akk, lda, ldk, status, 1, 0, 0, st);
      if (rc != MCP_OK) return rc;
    } else {
      const size_t lds = sizeof(double) * ((size_t)CH_NB * (CH_NB + 1) + (size_t)CH_NB * nb);
      hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(CH_NT), lds, st, nb, Akk, lda, ldk, status);
      MCP_LAUNCH_CHECK();
    }
    if (R > 0) {  // (nb == NB here)
      double* W = A + (size_t)(k == 0 ? NB : k0) * lda;
      const int NBK = NB / 16;
      hipLaunchKernelGGL(tri_diag_inverse_kernel, dim3(NBK, 1), dim3(64), 0, st, NB, Akk, lda, W, lda, (size_t)0, (size_t)0);
      MCP_LAUNCH_CHECK();
      MCP_ENSURE_MAX_LDS(tri_inverse_cols4_kernel);
      hipLaunchKernelGGL(tri_inverse_cols4_kernel, dim3(NBK, 1), dim3(256), sizeof(double) * 256 * (size_t)(NBK + 3), st, NB, Akk, lda, W, lda,
                         (size_t)0, (size_t)0);
      MCP_LAUNCH_CHECK();
      double* Akj = Akk + nb;
      hipLaunchKernelGGL(chol_panel_solve_kernel, dim3((R + 63) / 64), dim3(256), 0, st, R, W, lda, Akj, lda);
      MCP_LAUNCH_CHECK();
      const int T = (R + 63) / 64;
      hipLaunchKernelGGL(chol_trailing_update_kernel, dim3(T, T), dim3(256), 0, st, R, Akj, lda, Akk + (size_t)nb * lda + nb, lda);
      MCP_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(chol_finish_kernel, dim3(256), dim3(256), 0, st, N, A, lda, np, logdet);
  MCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(chol_finish_tail_kernel, dim3(1), dim3(256), 0, st, N, A, lda);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// ---------------------------------------------------------------------------------------
// U^-1 and K^-1 beyond the one-launch forms' 1152 rows (round 5): Y = U^-T by block forward substitution, every product in the transposed-
// left form the MFMA operands load coalesced from row-major storage,  C = +- P^T Q:
//   W_I = U_II^-1                                     all diagonal blocks at once (tri_diag_inverse + tri_inverse_cols4, batched), into Uinv
//   Y[I][I] = W_I^T;  T = U[0:I0, I]^T Y[0:I0, 0:I0]  (K = I0; Y lower triangular: rows above a column block are skipped)
//   Y[I][0:I0] = - W_I^T T                            (K = 128)
// Y is built in the Kinv buffer, T in the lower triangle of Uinv (zero at the end by contract); then Uinv = Y^T by tiles and
// Kinv = Uinv Uinv^T (kinv_tiles_kernel) over the whole matrix.  Replaces the round-1 column kernels there: N = 2048 94 -> 1.5 ms, 4096 659 -> 6.6 ms.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tn_gemm_kernel(int M, int Nn, int K, const double* __restrict__ P, int ldp, const double* __restrict__ Q,
                                                      int ldq, double* __restrict__ C, int ldc, double sign, int q_lower) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15, w = threadIdx.x >> 6;
  const int m0 = 64 * (int)blockIdx.y + 32 * (w >> 1), n0 = 64 * (int)blockIdx.x + 32 * (w & 1);
  if (m0 >= M || n0 >= Nn) return;
  const int ma = min(m0 + li, M - 1), mb = min(m0 + 16 + li, M - 1), na = min(n0 + li, Nn - 1), nb = min(n0 + 16 + li, Nn - 1);
  v4d_p acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = v4d_p{0.0, 0.0, 0.0, 0.0};
  for (int k0 = q_lower ? (n0 & ~31) : 0; k0 < K; k0 += 32) {  // (Q lower triangular: its rows above column n0 are zero)
    double a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kr = min(k0 + 4 * u + kq, K - 1);
      const double* pr = P + (size_t)kr * ldp;
      const double* qr = Q + (size_t)kr * ldq;
      a0[u] = pr[ma];
      a1[u] = pr[mb];
      b0[u] = qr[na];
      b1[u] = qr[nb];
    }
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a0[4]), "+v"(a0[5]), "+v"(a0[6]), "+v"(a0[7]), "+v"(a1[0]), "+v"(a1[1]),
                 "+v"(a1[2]), "+v"(a1[3]), "+v"(a1[4]), "+v"(a1[5]), "+v"(a1[6]), "+v"(a1[7]));
    asm volatile("" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b0[4]), "+v"(b0[5]), "+v"(b0[6]), "+v"(b0[7]), "+v"(b1[0]), "+v"(b1[1]),
                 "+v"(b1[2]), "+v"(b1[3]), "+v"(b1[4]), "+v"(b1[5]), "+v"(b1[6]), "+v"(b1[7]));
    if (k0 + 32 > K) {  // (the last, partial batch: the clamped rows count once)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + 4 * u + kq >= K) a0[u] = a1[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + kq + 4 * r, col = n0 + 16 * j + li;
        if (row < M && col < Nn) C[(size_t)row * ldc + col] = sign * acc[i][j][r];
      }
}
// Y[I][I] = W_I^T for every diagonal block (W_I: the diagonal blocks of Ui)
__global__ void diag_blocks_transpose_kernel(int N, const double* __restrict__ Ui, int ldi, double* __restrict__ Y, int ldy) {
  const int I0 = (int)blockIdx.x * CHB_NB, nb = min(CHB_NB, N - I0);
  for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
    const int r = e / nb, c = e - r * nb;
    Y[(size_t)(I0 + r) * ldy + I0 + c] = Ui[(size_t)(I0 + c) * ldi + I0 + r];
  }
}
// Ui = Y^T outside the diagonal blocks (32 x 32 tiles through LDS), the strictly-lower blocks of Ui back to zero
__global__ __launch_bounds__(256) void uinv_from_y_kernel(int N, const double* __restrict__ Y, int ldy, double* __restrict__ Ui, int ldi) {
  __shared__ double tile[32][33];
  const int br = blockIdx.y, bc = blockIdx.x;  // tile (br, bc) of Y, br >= bc
  if (br < bc) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bool same_block = (32 * br) / CHB_NB == (32 * bc) / CHB_NB;  // inside a diagonal block of 128: Ui holds W_I already
  if (same_block) return;
  for (int r = ty; r < 32; r += 8) {
    const int row = 32 * br + r, col = 32 * bc + tx;
    tile[r][tx] = (row < N && col < N) ? Y[(size_t)row * ldy + col] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int row = 32 * bc + r, col = 32 * br + tx;  // Ui[col of Y][row of Y]
    if (row < N && col < N) Ui[(size_t)row * ldi + col] = tile[tx][r];
    const int lr = 32 * br + r, lc = 32 * bc + tx;    // the mirrored (strictly-lower) tile: the scratch of T -> zero
    if (lr < N && lc < N) Ui[(size_t)lr * ldi + lc] = 0.0;
  }
}

static int launch_inverse_blocked(int N, const double* U, int ldu, double* Ui, int ldi, double* Kinv, int ldk, hipStream_t st) {
  const int NB = CHB_NB, np = (N + NB - 1) / NB, nfull = N / NB, nlast = N - nfull * NB;
  const int NBK = NB / 16;
  MCP_ENSURE_MAX_LDS(tri_inverse_cols4_kernel);
  if (nfull > 0) {  // W_I of the full blocks, batched over I (block I at offset I (NB ld + NB))
    hipLaunchKernelGGL(tri_diag_inverse_kernel, dim3(NBK, nfull), dim3(64), 0, st, NB, U, ldu, Ui, ldi, (size_t)NB * ldu + NB, (size_t)NB * ldi + NB);
    MCP_LAUNCH_CHECK();
    hipLaunchKernelGGL(tri_inverse_cols4_kernel, dim3(NBK, nfull), dim3(256), sizeof(double) * 256 * (size_t)(NBK + 3), st, NB, U, ldu, Ui, ldi,
                       (size_t)NB * ldu + NB, (size_t)NB * ldi + NB);
    MCP_LAUNCH_CHECK();
  }
  if (nlast > 0) {
    const size_t off_u = (size_t)nfull * ((size_t)NB * ldu + NB), off_i = (size_t)nfull * ((size_t)NB * ldi + NB);
    const int nbk = (nlast + 15) / 16;
    hipLaunchKernelGGL(tri_diag_inverse_kernel, dim3(nbk, 1), dim3(64), 0, st, nlast, U + off_u, ldu, Ui + off_i, ldi, (size_t)0, (size_t)0);
    MCP_LAUNCH_CHECK();
    hipLaunchKernelGGL(tri_inverse_cols4_kernel, dim3(nbk, 1), dim3(256), sizeof(double) * 256 * (size_t)(nbk + 3), st, nlast, U + off_u, ldu, Ui + off_i,
                       ldi, (size_t)0, (size_t)0);
    MCP_LAUNCH_CHECK();
  }
  double* Y = Kinv;
  hipLaunchKernelGGL(diag_blocks_transpose_kernel, dim3(np), dim3(256), 0, st, N, Ui, ldi, Y, ldk);
  MCP_LAUNCH_CHECK();
  for (int I = 1; I < np; ++I) {
    const int I0 = I * NB, nb = N - I0 < NB ? N - I0 : NB;
    double* T = Ui + (size_t)I0 * ldi;  // rows I0.., columns 0..I0 of the lower triangle
    // T [nb x I0] = U[0:I0, I0:I0+nb]^T Y[0:I0, 0:I0]
    hipLaunchKernelGGL(tn_gemm_kernel, dim3((I0 + 63) / 64, (nb + 63) / 64), dim3(256), 0, st, nb, I0, I0, U + I0, ldu, Y, ldk, T, ldi, 1.0, 1);
    MCP_LAUNCH_CHECK();
    // Y[I][0:I0] = - W_I^T T
    hipLaunchKernelGGL(tn_gemm_kernel, dim3((I0 + 63) / 64, (nb + 63) / 64), dim3(256), 0, st, nb, I0, nb, Ui + (size_t)I0 * ldi + I0, ldi, T, ldi,
                       Y + (size_t)I0 * ldk, ldk, -1.0, 0);
    MCP_LAUNCH_CHECK();
  }
  const int nt32 = (N + 31) / 32;
  hipLaunchKernelGGL(uinv_from_y_kernel, dim3(nt32, nt32), dim3(256), 0, st, N, Y, ldk, Ui, ldi);
  MCP_LAUNCH_CHECK();
  const int NBK16 = (N + 15) >> 4, nt = NBK16 * (NBK16 + 1) / 2;
  hipLaunchKernelGGL(kinv_tiles_kernel, dim3((nt + 3) / 4, 1), dim3(256), 0, st, N, Ui, ldi, Kinv, ldk, (size_t)0, (size_t)0);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// which form of the factorisation / inverse a call runs: 1 (default) the round-4/5 kernels, 0 the round-1/2 ones, 2 the round-3 one-workgroup
// forms, 3 the round-4 forms with one-wave inverse columns -- requested per call (mcp_dispatch.chol_form, include/mcpilco_hip_debug.h)
static int chol_form_of(const mcp_dispatch* d) { return !d ? 1 : (d->chol_form == 1 ? 0 : (d->chol_form == 2 ? 2 : (d->chol_form == 3 ? 3 : 1))); }

extern "C" int mcp_chol_factor_ex(int N, double* A, int lda, double* logdet, uint32_t* status, void* stream, const mcp_dispatch* d) {
  const int g_chol_mfma = chol_form_of(d);
  if (!A || !logdet || !status || N <= 0 || lda < N) return MCP_ERR_ARG;
  if (N > 8192) return MCP_ERR_LIMIT;
  if (g_chol_mfma == 1 && N >= CHB_MIN) return launch_chol_blocked(N, A, lda, logdet, status, (hipStream_t)stream);  // panels across the chip
  if (N > 1152) return MCP_ERR_LIMIT;  // (the one-workgroup forms: test hooks 0, 2, 3)
  if (g_chol_mfma && N > 16) {
    return launch_chol_mfma(g_chol_mfma == 3 ? 1 : g_chol_mfma, N, A, lda, logdet, status, 1, 0, 0, (hipStream_t)stream);
  }
  size_t lds = sizeof(double) * ((size_t)CH_NB * (CH_NB + 1) + (size_t)CH_NB * N);
  MCP_ENSURE_MAX_LDS(chol_factor_kernel);
  hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(CH_NT), lds, (hipStream_t)stream, N, A, lda, logdet, status);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_chol_factor(int N, double* A, int lda, double* logdet, uint32_t* status, void* stream) {
  return mcp_chol_factor_ex(N, A, lda, logdet, status, stream, nullptr);
}

extern "C" int mcp_chol_inverse_ex(int N, const double* U, int ldu, double* Uinv, int ldi, double* Kinv, int ldk, void* stream, const mcp_dispatch* d) {
  const int g_chol_mfma = chol_form_of(d);
  if (!U || !Uinv || !Kinv || N <= 0 || ldu < N || ldi < N || ldk < N) return MCP_ERR_ARG;
  if (N > 16384) return MCP_ERR_LIMIT;
  if (g_chol_mfma == 1 && N > 1152) return launch_inverse_blocked(N, U, ldu, Uinv, ldi, Kinv, ldk, (hipStream_t)stream);
  if ((g_chol_mfma == 1 || g_chol_mfma == 3) && N > 16 && N <= 1152)  // (3: the one-wave-per-column form of the inverse)
    return launch_inverse_mfma(N, U, ldu, Uinv, ldi, Kinv, ldk, 1, 0, 0, 0, (hipStream_t)stream, g_chol_mfma == 3);
  if (g_chol_mfma && N > 16 && N <= 1152)  // (2: the round-3 block-diagonal sweep of one workgroup, kept as a comparison form)
    hipLaunchKernelGGL(tri_inverse_block_kernel, dim3(1), dim3(CM_NT), 0, (hipStream_t)stream, N, U, ldu, Uinv, ldi, (size_t)0, (size_t)0);
  else if (N <= 64 * TW_KM)
    hipLaunchKernelGGL(tri_inverse_wave_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, U, ldu, Uinv, ldi);
  else
    hipLaunchKernelGGL(tri_inverse_kernel, dim3((N + TI_NT - 1) / TI_NT), dim3(TI_NT), sizeof(double) * N, (hipStream_t)stream, N, U,
                       ldu, Uinv, ldi);
  MCP_LAUNCH_CHECK();
  dim3 grid((N + 255) / 256, N);
  hipLaunchKernelGGL(kinv_from_uinv_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, Uinv, ldi, Kinv, ldk, (size_t)0, (size_t)0);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_chol_inverse(int N, const double* U, int ldu, double* Uinv, int ldi, double* Kinv, int ldk, void* stream) {
  return mcp_chol_inverse_ex(N, U, ldu, Uinv, ldi, Kinv, ldk, stream, nullptr);
}

// out = A G A for a SYMMETRIC A (K^-1) and any G: S = G^T A, out = S^T A -- two products in the transposed-left form of tn_gemm_kernel
// (coalesced MFMA operands from row-major storage).  The chain rule through K^-1 of GP_prior.forward's autograd graph:
// d K^-1 = - K^-1 dK K^-1  (GP_prior.py:109-110 under autograd; _ForwardFunction.backward).
extern "C" int mcp_sym_sandwich(int N, const double* A, int lda, const double* G, int ldg, double* out, int ldo, double* scratch, void* stream) {
  if (!A || !G || !out || !scratch || N <= 0 || lda < N || ldg < N || ldo < N) return MCP_ERR_ARG;
  if (N > 16384) return MCP_ERR_LIMIT;
  if (out == A || out == G || scratch == A || scratch == G || scratch == out) return MCP_ERR_ARG;
  const dim3 grid((N + 63) / 64, (N + 63) / 64);
  hipLaunchKernelGGL(tn_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, N, N, G, ldg, A, lda, scratch, N, 1.0, 0);
  MCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tn_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, N, N, scratch, N, A, lda, out, ldo, 1.0, 0);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_gp_alpha(int N, const double* Kinv, int ldk, const double* Y, double mean, double* alpha, void* stream) {
  if (!Kinv || !Y || !alpha || N <= 0 || ldk < N) return MCP_ERR_ARG;
  hipLaunchKernelGGL(gp_alpha_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, Kinv, ldk, Y, mean, alpha);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" int mcp_gp_pack(int N, int D, const double* X, const double* alpha, const double* Kinv, int ldk, int Npad, double* Xt_out,
                           double* X_out, double* alpha_out, double* Kinv_out, double* aX_out, void* stream) {
  if (!X || !alpha || !Kinv || !Xt_out || !X_out || !alpha_out || !Kinv_out || !aX_out) return MCP_ERR_ARG;
  if (N <= 0 || D <= 0 || D > MCP_MAX_GPDIM || Npad < N || (Npad % 16) != 0 || ldk < N) return MCP_ERR_ARG;
  hipLaunchKernelGGL(gp_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, N, D, X, alpha, Kinv, ldk, Npad, Xt_out, X_out,
                     alpha_out, Kinv_out, aX_out);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// W [N][N] + the one-workgroup kernel's running sums [2 N]; from SODM_MIN candidates on also the granules of the multi-workgroup kernel (a caller that
// passes the first part only gets the one-workgroup kernel)
static size_t sod_base_bytes(int N) { return sizeof(double) * ((size_t)N * N + 2 * (size_t)N); }
static bool sod_multi_applies(int N) { return N >= SODM_MIN && N <= 4096; }
// (the multi-workgroup kernel's part: the exchange granules and the Gram matrix of the candidates)
static size_t sod_multi_bytes(int N) { return sizeof(unsigned long long) * sod_multi_granules(N) + sizeof(double) * (size_t)N * N; }
extern "C" size_t mcp_sod_workspace_bytes(int N) {
  if (N <= 0) return 0;
  return sod_base_bytes(N) + (sod_multi_applies(N) ? sod_multi_bytes(N) : 0);
}

extern "C" int mcp_sod_select(const mcp_kernel* kern, int N, const double* X, double threshold, int32_t* idx_out, int32_t* n_out,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!kernel_ok(kern) || !X || !idx_out || !n_out || !workspace || N <= 0) return MCP_ERR_ARG;
  if (workspace_bytes < sod_base_bytes(N)) return MCP_ERR_WORKSPACE;
  if (N > 16384) return MCP_ERR_LIMIT;  // the accepted point's vector [N] lives in LDS
  double* Uw = (double*)workspace;
  if (sod_multi_applies(N) && workspace_bytes >= mcp_sod_workspace_bytes(N)) {
    unsigned long long* gx = (unsigned long long*)((char*)workspace + sod_base_bytes(N));
    if (hipMemsetAsync(gx, 0, sizeof(unsigned long long) * sod_multi_granules(N), (hipStream_t)stream) != hipSuccess) return MCP_ERR_LAUNCH;
    double* Kg = (double*)(gx + sod_multi_granules(N));
    hipLaunchKernelGGL(cov_build_kernel, dim3((N + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, *kern, N, X, N, X, 0, Kg, N);
    MCP_LAUNCH_CHECK();
    const size_t fixed = sizeof(double) * ((size_t)N + SOD_NT + 64 + 4);
    const int LR = (int)std::min<size_t>((size_t)N, ((size_t)160 * 1024 - fixed) / (64 * sizeof(double)));  // rows of the block's columns kept in LDS
    MCP_ENSURE_MAX_LDS(sod_select_multi_kernel);
    hipLaunchKernelGGL(sod_select_multi_kernel, dim3((N + 63) / 64), dim3(SOD_NT), fixed + (size_t)LR * 64 * sizeof(double), (hipStream_t)stream, *kern, N,
                       X, threshold, idx_out, n_out, Uw, gx, LR, Kg);
    MCP_LAUNCH_CHECK();
    return MCP_OK;
  }
  MCP_ENSURE_MAX_LDS(sod_select_kernel);
  hipLaunchKernelGGL(sod_select_kernel, dim3(1), dim3(SOD_NT), sizeof(double) * ((size_t)N + SOD_NT + 4), (hipStream_t)stream, *kern, N, X,
                     threshold, idx_out, n_out, Uw);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

extern "C" size_t mcp_nll_workspace_bytes(int N, int D) { return (N > 0 && D > 0) ? sizeof(double) * (size_t)N * (4 * D + 3) : 0; }

extern "C" int mcp_nll_grad(const mcp_kernel* kern, int N, const double* X, const double* Kinv, int ldk, const double* alpha, double* grad,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!kernel_ok(kern) || !X || !Kinv || !alpha || !grad || !workspace || N <= 0 || ldk < N) return MCP_ERR_ARG;
  if (N > 4096) return MCP_ERR_LIMIT;  // four [N] row buffers live in LDS
  if (workspace_bytes < mcp_nll_workspace_bytes(N, kern->D)) return MCP_ERR_WORKSPACE;
  MCP_ENSURE_MAX_LDS(nll_grad_kernel);
  const int NP = 4 * kern->D + 3;
  double* slab = (double*)workspace;
  hipLaunchKernelGGL(nll_grad_kernel, dim3(N), dim3(256), sizeof(double) * (4 * (size_t)N + 256), (hipStream_t)stream, *kern, N, X, Kinv, ldk, alpha,
                     slab);
  MCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(nll_colsum_kernel, dim3((NP + 127) / 128), dim3(128), 0, (hipStream_t)stream, N, NP, slab, grad);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}

// ---------------------------------------------------------------------------------------
// One epoch of GP hyper-parameter training for the G GPs of a model at once (mcp_nll_epoch): what GP_prior.fit_model does per epoch
// through forward + Marginal_log_likelihood + autograd (gpr_lib/GP_prior/GP_prior.py:91-115,179-230; Gaussian_likelihood.py:15-24;
// Model_learning.train_gp_likelihood, model_learning/Model_learning.py:398-421), from the optimizer's RAW parameters to their gradients
// without a host round trip: the GPs are independent, so every stage is ONE launch whose grid carries the GP index.
// Workspace, per GP (doubles): K -> U [N N] | Uinv [N N] | Kinv [N N] | alpha [N] | r [N] | slab [N NP] | grad [NP] | inv_ls [D] |
// w1 [D+1] | w20 [D] | w21 [D] | scal [3] | logdet [1]; in front of all of them the G mcp_kernel descriptors the stages read.
// ---------------------------------------------------------------------------------------
struct NllBatch {
  mcp_nll_gp gp[MCP_MAX_GP];
};
struct NllWs {
  size_t kn, K, Ui, Kinv, alpha, r, slab, grad, invls, w1, w20, w21, scal, logdet, per_gp, total;  // offsets in doubles
};
static inline NllWs nll_ws_layout(int G, int N, int D) {
  NllWs w;
  const size_t NP = 4 * (size_t)D + 3, NN = (size_t)N * N;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += (n + 1) & ~(size_t)1;
    return r;
  };
  w.K = take(NN);
  w.Ui = take(NN);
  w.Kinv = take(NN);
  w.alpha = take(N);
  w.r = take(N);
  w.slab = take((size_t)N * NP);
  w.grad = take(NP);
  w.invls = take(D);
  w.w1 = take(D + 1);
  w.w20 = take(D);
  w.w21 = take(D);
  w.scal = take(4);
  w.logdet = take(2);
  w.per_gp = o;
  w.kn = 0;  // the descriptors come first
  const size_t knd = ((size_t)G * sizeof(mcp_kernel) + 15) / 16 * 2;
  w.total = knd + (size_t)G * w.per_gp;
  return w;
}
__device__ __forceinline__ double* nll_gp_base(double* ws, int G, size_t per_gp, int g) {
  const size_t knd = ((size_t)G * sizeof(mcp_kernel) + 15) / 16 * 2;
  return ws + knd + (size_t)g * per_gp;
}

// raw parameters -> the kernels' operands: 1 / l, lambda = exp(log_lambda), sigma_n^2 = exp(sigma_n_log)^2 + sigma_n_num^2, the MPK weights
// s^2 with s_d = (k - d) exp(par_d) (Sparse_GP.py:613-623: the reference's get_Sigma), and the mcp_kernel descriptor that points at them
__global__ void nll_prep_kernel(NllBatch b, int G, int N, int D, int deg, int ard, double* __restrict__ ws, NllWs L) {
  const int g = blockIdx.x, tid = threadIdx.x;
  const mcp_nll_gp& gp = b.gp[g];
  double* base = nll_gp_base(ws, G, L.per_gp, g);
  double *invls = base + L.invls, *w1 = base + L.w1, *w20 = base + L.w20, *w21 = base + L.w21, *scal = base + L.scal;
  for (int d = tid; d < D; d += blockDim.x) {
    invls[d] = exp(-gp.log_ls[ard ? d : 0]);
    if (deg >= 2) {
      const double s0 = 2.0 * exp(gp.mpk2[d]), s1 = exp(gp.mpk2[D + d]);
      w20[d] = s0 * s0;
      w21[d] = s1 * s1;
    }
  }
  if (deg >= 1)
    for (int d = tid; d <= D; d += blockDim.x) {
      const double s = gp.mpk1 ? exp(gp.mpk1[d]) : 0.0;  // (a degree-2 term without a degree-1 term: zero weights)
      w1[d] = s * s;
    }
  if (tid == 0) {
    scal[0] = gp.log_lambda ? exp(gp.log_lambda[0]) : 0.0;
    const double sn = gp.sigma_n_log ? exp(gp.sigma_n_log[0]) : 0.0;
    scal[1] = sn * sn + gp.sigma_n_num2;
    scal[2] = gp.mean ? gp.mean[0] : 0.0;
    mcp_kernel kn;
    kn.D = D;
    kn.poly_deg = deg;
    kn.lambda = kn.sigma_n2 = kn.mean = 0.0;
    kn.inv_ls = invls;
    kn.w1 = deg >= 1 ? w1 : nullptr;
    kn.w20 = deg >= 2 ? w20 : nullptr;
    kn.w21 = deg >= 2 ? w21 : nullptr;
    kn.scal = scal;
    reinterpret_cast<mcp_kernel*>(ws)[g] = kn;
  }
}
// 16 rows x 256 columns of one GP's Gram matrix per workgroup.  The 256 inputs x_j of the columns (transposed: xj[d][j], lanes read
// consecutive addresses), the 16 inputs x_i of the rows and the kernel's weights are staged in LDS once, so a thread's D-long loops run
// on LDS instead of on strided global loads whose latency they could not hide (one thread per entry with both inputs in global memory:
// 124 us for the six 400 x 400, D = 24 matrices of the UR5 model).  The arithmetic is kern_eval's, operation for operation.
#define CB_ROWS 16
// (DEG: the polynomial degree as a template parameter -- as a run-time test inside the unrolled row loop it was a scalar branch behind every LDS
//  read, each waiting for its own operand: 41 us for what takes 11 without the loop)
template <int DEG>
__global__ __launch_bounds__(256) void cov_build_batch_kernel(const mcp_kernel* __restrict__ kns, int N, const double* __restrict__ X,
                                                              double* __restrict__ ws, int G, NllWs L) {
  extern __shared__ __attribute__((aligned(16))) double cb[];
  const int g = blockIdx.z, i0 = blockIdx.y * CB_ROWS, j0 = blockIdx.x * 256, tid = threadIdx.x;
  const mcp_kernel kn = kns[g];
  const int D = kn.D;
  constexpr int deg = DEG;
  double* xj = cb;                       // [D][257]
  double* xi = xj + 257 * D;             // [CB_ROWS][D]
  double* par = xi + CB_ROWS * D;        // inv_ls[D] | w1[D + 1] | w20[D] | w21[D]
  const int nj = min(256, N - j0), ni = min(CB_ROWS, N - i0);
  for (int base = tid; base < nj * D; base += 8 * 256) {  // (eight loads in flight per thread: four waves alone on a CU hide nothing else)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * 256;
      v[u] = e < nj * D ? X[(size_t)j0 * D + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * 256;
      if (e < nj * D) {
        const int r = e / D, d = e - r * D;
        xj[d * 257 + r] = v[u];
      }
    }
  }
  for (int e = tid; e < ni * D; e += 256) xi[e] = X[(size_t)i0 * D + e];
  for (int d = tid; d < D; d += 256) {
    par[d] = kn.inv_ls[d];
    par[2 * D + 1 + d] = deg >= 2 ? kn.w20[d] : 0.0;
    par[3 * D + 1 + d] = deg >= 2 ? kn.w21[d] : 0.0;
  }
  for (int d = tid; d <= D; d += 256) par[D + d] = deg >= 1 ? kn.w1[d] : 0.0;
  __syncthreads();
  if (tid >= nj) return;
  const double *inv_ls = par, *w1 = par + D, *w20 = par + 2 * D + 1, *w21 = par + 3 * D + 1;
  double* Kg = nll_gp_base(ws, G, L.per_gp, g) + L.K;
  const double sn2 = kern_sigma_n2(kn), lam = kern_lambda(kn);
  // feature by feature, the CB_ROWS rows side by side: this column's x_jd and the weights are read once per feature instead of once per
  // (row, feature), and the rows' sums are CB_ROWS independent chains; every sum still runs over d in kern_eval's order
  double dist[CB_ROWS], p1[CB_ROWS], pa[CB_ROWS], pb[CB_ROWS];
#pragma unroll
  for (int r = 0; r < CB_ROWS; ++r) {
    dist[r] = 0.0;
    p1[r] = w1[D];
    pa[r] = pb[r] = 0.0;
  }
  for (int d = 0; d < D; ++d) {
    const double xjd = xj[d * 257 + tid], il = inv_ls[d], wd = w1[d], wa = w20[d], wb = w21[d];
#pragma unroll
    for (int r = 0; r < CB_ROWS; ++r) {
      const double ad = xi[(r < ni ? r : 0) * D + d];
      const double q = (ad - xjd) * il;
      dist[r] = fma(q, q, dist[r]);
      if (deg >= 1) p1[r] = fma(wd * ad, xjd, p1[r]);
      if (deg >= 2) {
        const double ab = ad * xjd;
        pa[r] = fma(wa, ab, pa[r]);
        pb[r] = fma(wb, ab, pb[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < CB_ROWS; ++r) {
    if (r < ni) {
      double k = lam * exp(-dist[r]);
      if (deg >= 1) {
        k += p1[r];
        if (deg >= 2) k = fma(pa[r], pb[r], k);
      }
      if (i0 + r == j0 + tid) k += sn2;
      Kg[(size_t)(i0 + r) * N + j0 + tid] = k;
    }
  }
}
static inline size_t cov_build_batch_lds(int D) { return sizeof(double) * ((size_t)257 * D + (size_t)CB_ROWS * D + 4 * (size_t)D + 2); }
// r = Y y_scale - mean;  alpha = Kinv r  (one wave per row)
__global__ void nll_alpha_batch_kernel(NllBatch b, int G, int N, double* __restrict__ ws, NllWs L) {
  const int g = blockIdx.y, row = blockIdx.x * (blockDim.x / MCP_WAVE) + (threadIdx.x / MCP_WAVE), lane = threadIdx.x % MCP_WAVE;
  if (row >= N) return;
  const mcp_nll_gp& gp = b.gp[g];
  double* base = nll_gp_base(ws, G, L.per_gp, g);
  const double* Kinv = base + L.Kinv;
  const double mean = base[L.scal + 2];
  double s = 0.0;
  for (int m = lane; m < N; m += MCP_WAVE) s = fma(Kinv[(size_t)row * N + m], gp.Y[m] * gp.y_scale - mean, s);
  s = wave_sum(s);
  if (lane == 0) {
    base[L.alpha + row] = s;
    base[L.r + row] = gp.Y[row] * gp.y_scale - mean;
  }
}
__global__ __launch_bounds__(256) void nll_grad_batch_kernel(const mcp_kernel* __restrict__ kns, int N, const double* __restrict__ X,
                                                             double* __restrict__ ws, int G, NllWs L) {
  const int g = blockIdx.y;
  double* base = nll_gp_base(ws, G, L.per_gp, g);
  nll_grad_row(kns[g], N, X, base + L.Kinv, N, base + L.alpha, base + L.slab);
}
// The same rows by a workgroup of 16 waves that takes `rows` consecutive rows i, with the inputs staged in LDS once, TRANSPOSED
// (xs[d][j], odd pitch: lanes j read consecutive addresses), and the kernel's weights beside them.
//   (1) thread j:  Wm_ij, Wm_ij kse_ij, A_ij, B_ij                          (the D-long loops run on LDS)
//   (2) wave w:    hyper-parameters p = w, w + 16, ...  -- one parameter at a time, the SAME for all lanes (no divergence between the
//                  parameter classes), lanes over j, one wave reduction per parameter; summed over the workgroup's rows in registers:
//                  slab[workgroup][p], added up by nll_finish_kernel in workgroup order.
// The row kernel above walks D- and N-long loops of global loads per thread (strided, or one element per iteration, with one thread per
// parameter and the classes diverging inside a wave): 128 us per epoch at the UR5 shape (six GPs, N = 400, D = 24), where the
// arithmetic is a few microseconds.
#define NG_NT 1024
#define NG_KMAX ((4 * MCP_MAX_GPDIM + 3 + NG_NT / 64 - 1) / (NG_NT / 64))  // parameters per wave
template <int DEG>  // (the polynomial degree at compile time: no scalar branch inside the feature loops)
__global__ __launch_bounds__(NG_NT) void nll_grad_rows_kernel(const mcp_kernel* __restrict__ kns, int N, const double* __restrict__ X,
                                                              double* __restrict__ ws, int G, NllWs L, int rows) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i0 = blockIdx.x * rows;
  const mcp_kernel kn = kns[g];
  const int D = kn.D, NP = 4 * D + 3, Np = N | 1;
  constexpr int deg = DEG;
  double* wm = sm;               // [N] Wm_ij
  double* wk = sm + N;           // [N] Wm_ij * kse_ij
  double* fa = sm + 2 * N;       // [N] Wm_ij * A_ij   (MPK_2 factors)
  double* fb = sm + 3 * N;       // [N] Wm_ij * B_ij
  double* par = sm + 4 * N;      // inv_ls[D] | w1[D + 1] | w20[D] | w21[D]
  double* xs = par + 4 * D + 2;  // [D][Np]
  const double* base = nll_gp_base(ws, G, L.per_gp, g);
  const double *Kinv = base + L.Kinv, *alpha = base + L.alpha;
  double* slab = nll_gp_base(ws, G, L.per_gp, g) + L.slab;
  for (int base = tid; base < N * D; base += 12 * NG_NT) {  // (twelve loads in flight per thread: the staging is a chain of round trips otherwise)
    double v[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int e = base + u * NG_NT;
      v[u] = e < N * D ? X[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int e = base + u * NG_NT;
      if (e < N * D) {
        const int r = e / D, d = e - r * D;
        xs[d * Np + r] = v[u];
      }
    }
  }
  for (int d = tid; d < D; d += NG_NT) {
    par[d] = kn.inv_ls[d];
    par[2 * D + 1 + d] = deg >= 2 ? kn.w20[d] : 0.0;
    par[3 * D + 1 + d] = deg >= 2 ? kn.w21[d] : 0.0;
  }
  for (int d = tid; d <= D; d += NG_NT) par[D + d] = deg >= 1 ? kn.w1[d] : 0.0;
  const double lam = kern_lambda(kn);
  const double *inv_ls = par, *w1 = par + D, *w20 = par + 2 * D + 1, *w21 = par + 3 * D + 1;
  double tot[NG_KMAX];  // this wave's parameters, this lane's share, summed over the workgroup's rows
#pragma unroll
  for (int kx = 0; kx < NG_KMAX; ++kx) tot[kx] = 0.0;
  __syncthreads();
  for (int i = i0; i < min(i0 + rows, N); ++i) {
    const double ai = alpha[i];
    for (int j = tid; j < N; j += NG_NT) {
      double dist = 0.0, A = 0.0, Bv = 0.0;
#pragma unroll 6
      for (int d = 0; d < D; ++d) {
        const double xid = xs[d * Np + i], xjd = xs[d * Np + j];
        double r = (xid - xjd) * inv_ls[d];
        dist = fma(r, r, dist);
        if (deg >= 2) {
          double xx = xid * xjd;
          A = fma(w20[d], xx, A);
          Bv = fma(w21[d], xx, Bv);
        }
      }
      double w = Kinv[(size_t)i * N + j] - ai * alpha[j];
      wm[j] = w;
      wk[j] = w * lam * exp(-dist);
      fa[j] = w * A;
      fb[j] = w * Bv;
    }
    __syncthreads();
#pragma unroll
    for (int kx = 0; kx < NG_KMAX; ++kx) {
      const int p = wv + kx * (NG_NT / 64);  // (wave-uniform)
      if (p >= NP) break;
      double s = 0.0;
      if (p < D) {  // d/d log l_p :  kse * 2 (dx/l)^2
        const double il2 = inv_ls[p] * inv_ls[p], xip = xs[p * Np + i];
#pragma unroll 4
        for (int j = lane; j < N; j += 64) {
          const double dx = xip - xs[p * Np + j];
          s = fma(wk[j], 2.0 * dx * dx * il2, s);
        }
      } else if (p == D) {  // d/d log lambda
#pragma unroll 4
        for (int j = lane; j < N; j += 64) s += wk[j];
      } else if (p == D + 1) {  // 1/2 tr Wm (the caller multiplies by d sigma_n^2 / d sigma_n_log)
        s = lane == 0 ? wm[i] : 0.0;
      } else if (p < 2 * D + 3) {  // MPK_1, feature e (e == D: the offset feature)
        const int e = p - (D + 2);
        if (deg >= 1) {
          const double c = 2.0 * w1[e] * (e < D ? xs[e * Np + i] : 1.0);
          if (e < D)
#pragma unroll 4
          for (int j = lane; j < N; j += 64) s = fma(wm[j], c * xs[e * Np + j], s);
          else
#pragma unroll 4
          for (int j = lane; j < N; j += 64) s = fma(wm[j], c, s);
        }
      } else if (p < 3 * D + 3) {  // MPK_2 factor 0 parameter e: 2 w20_e x_ie x_je * B_ij
        const int e = p - (2 * D + 3);
        if (deg >= 2) {
          const double c = 2.0 * w20[e] * xs[e * Np + i];
#pragma unroll 4
          for (int j = lane; j < N; j += 64) s = fma(fb[j], c * xs[e * Np + j], s);
        }
      } else {  // MPK_2 factor 1 parameter e: 2 w21_e x_ie x_je * A_ij
        const int e = p - (3 * D + 3);
        if (deg >= 2) {
          const double c = 2.0 * w21[e] * xs[e * Np + i];
#pragma unroll 4
          for (int j = lane; j < N; j += 64) s = fma(fa[j], c * xs[e * Np + j], s);
        }
      }
      tot[kx] += s;  // per lane, rows in order; the lanes meet once, below (the sum does not depend on how many GPs share the launch)
    }
    __syncthreads();  // (wm .. fb are rewritten by the next row)
  }
#pragma unroll
  for (int kx = 0; kx < NG_KMAX; ++kx) {
    const int p = wv + kx * (NG_NT / 64);
    if (p < NP) {  // (wave-uniform)
      const double t = 0.5 * wave_sum(tot[kx]);
      if (lane == 0) slab[(size_t)blockIdx.x * NP + p] = t;
    }
  }
}
// rows per workgroup: a function of N alone, so that a GP's sums are the same whether it is trained alone or in a batch
static inline int nll_grad_rows_per_wg(int N) { return (N + 127) / 128; }
static inline size_t nll_grad_rows_lds(int N, int D) { return sizeof(double) * (4 * (size_t)N + 4 * (size_t)D + 2 + (size_t)D * (N | 1)); }
__global__ __launch_bounds__(256) void nll_finish_kernel(NllBatch b, int G, int N, int D, int deg, int ard, double* __restrict__ ws, NllWs L,
                                                         int slab_rows) {
  __shared__ double sh[4 * MCP_MAX_GPDIM + 3];
  __shared__ double red[4];
  const int g = blockIdx.x, tid = threadIdx.x, NP = 4 * D + 3;
  const mcp_nll_gp& gp = b.gp[g];
  double* base = nll_gp_base(ws, G, L.per_gp, g);
  const double* slab = base + L.slab;
  for (int c = tid; c < NP; c += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = 0;
    for (; r + 3 < slab_rows; r += 4) {  // (slab: one row per row of K, or per workgroup of nll_grad_rows_kernel)
      s0 += slab[(size_t)r * NP + c];
      s1 += slab[(size_t)(r + 1) * NP + c];
      s2 += slab[(size_t)(r + 2) * NP + c];
      s3 += slab[(size_t)(r + 3) * NP + c];
    }
    for (; r < slab_rows; ++r) s0 += slab[(size_t)r * NP + c];
    sh[c] = (s0 + s1) + (s2 + s3);
  }
  // r . alpha and sum alpha
  double ra = 0.0, sa = 0.0;
  for (int j = tid; j < N; j += 256) {
    const double a = base[L.alpha + j];
    ra = fma(base[L.r + j], a, ra);
    sa += a;
  }
  ra = wave_sum(ra);
  sa = wave_sum(sa);
  if ((tid & 63) == 0) red[tid >> 6] = ra;
  __syncthreads();
  const double rdot = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = sa;
  __syncthreads();
  const double asum = ((red[0] + red[1]) + red[2]) + red[3];
  if (tid == 0 && gp.loss) gp.loss[0] = 0.5 * (rdot + base[L.logdet]);
  if (gp.g_log_ls) {
    if (ard) {
      for (int d = tid; d < D; d += 256) gp.g_log_ls[d] = sh[d];
    } else if (tid == 0) {
      double s = 0.0;
      for (int d = 0; d < D; ++d) s += sh[d];
      gp.g_log_ls[0] = s;
    }
  }
  if (tid == 0) {
    if (gp.g_log_lambda) gp.g_log_lambda[0] = sh[D];
    if (gp.g_sigma_n_log && gp.sigma_n_log) gp.g_sigma_n_log[0] = sh[D + 1] * 2.0 * exp(2.0 * gp.sigma_n_log[0]);
    if (gp.g_mean) gp.g_mean[0] = -asum;
  }
  if (gp.g_mpk1 && deg >= 1)
    for (int e = tid; e <= D; e += 256) gp.g_mpk1[e] = sh[D + 2 + e];
  if (gp.g_mpk2 && deg >= 2)
    for (int e = tid; e < 2 * D; e += 256) gp.g_mpk2[e] = sh[2 * D + 3 + e];
}

extern "C" size_t mcp_nll_epoch_workspace_bytes(int G, int N, int D) {
  if (G <= 0 || G > MCP_MAX_GP || N <= 0 || D <= 0 || D > MCP_MAX_GPDIM) return 0;
  return sizeof(double) * nll_ws_layout(G, N, D).total;
}

extern "C" int mcp_nll_epoch(int G, const mcp_nll_gp* gps, int N, int D, int poly_deg, int ard, const double* X, uint32_t* status,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!gps || !X || !status || !workspace || G <= 0 || N <= 0 || D <= 0) return MCP_ERR_ARG;
  if (G > MCP_MAX_GP || D > MCP_MAX_GPDIM || N > 1152 || N <= 16) return MCP_ERR_LIMIT;  // (the MFMA-blocked factorisations: 16 < N, row panel in LDS)
  if (poly_deg < 0 || poly_deg > 2) return MCP_ERR_ARG;
  const NllWs L = nll_ws_layout(G, N, D);
  if (workspace_bytes < sizeof(double) * L.total) return MCP_ERR_WORKSPACE;
  NllBatch b;
  for (int g = 0; g < MCP_MAX_GP; ++g) b.gp[g] = gps[g < G ? g : 0];
  for (int g = 0; g < G; ++g) {
    if (!gps[g].log_ls || !gps[g].log_lambda || !gps[g].Y) return MCP_ERR_ARG;
    if (poly_deg >= 2 && !gps[g].mpk2) return MCP_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  double* ws = (double*)workspace;
  const mcp_kernel* kns = (const mcp_kernel*)workspace;
  const size_t knd = ((size_t)G * sizeof(mcp_kernel) + 15) / 16 * 2;
  double* g0 = ws + knd;
  hipLaunchKernelGGL(nll_prep_kernel, dim3(G), dim3(64), 0, st, b, G, N, D, poly_deg, ard, ws, L);
  MCP_LAUNCH_CHECK();
  if (cov_build_batch_lds(D) > 150 * 1024) return MCP_ERR_LIMIT;
  {
    const dim3 cgrid((N + 255) / 256, (N + CB_ROWS - 1) / CB_ROWS, G);
    if (poly_deg >= 2) {
      MCP_ENSURE_MAX_LDS(cov_build_batch_kernel<2>);
      hipLaunchKernelGGL(cov_build_batch_kernel<2>, cgrid, dim3(256), cov_build_batch_lds(D), st, kns, N, X, ws, G, L);
    } else if (poly_deg == 1) {
      MCP_ENSURE_MAX_LDS(cov_build_batch_kernel<1>);
      hipLaunchKernelGGL(cov_build_batch_kernel<1>, cgrid, dim3(256), cov_build_batch_lds(D), st, kns, N, X, ws, G, L);
    } else {
      MCP_ENSURE_MAX_LDS(cov_build_batch_kernel<0>);
      hipLaunchKernelGGL(cov_build_batch_kernel<0>, cgrid, dim3(256), cov_build_batch_lds(D), st, kns, N, X, ws, G, L);
    }
  }
  MCP_LAUNCH_CHECK();
  {
    const int rc = launch_chol_mfma(1, N, g0 + L.K, N, g0 + L.logdet, status, G, L.per_gp, L.per_gp, st);
    if (rc != MCP_OK) return rc;
  }
  {
    const int rc = launch_inverse_mfma(N, g0 + L.K, N, g0 + L.Ui, N, g0 + L.Kinv, N, G, L.per_gp, L.per_gp, L.per_gp, st);
    if (rc != MCP_OK) return rc;
  }
  hipLaunchKernelGGL(nll_alpha_batch_kernel, dim3((N + 3) / 4, G), dim3(256), 0, st, b, G, N, ws, L);
  MCP_LAUNCH_CHECK();
  int slab_rows = N;
  if (nll_grad_rows_lds(N, D) <= 150 * 1024) {
    const int rows = nll_grad_rows_per_wg(N);
    slab_rows = (N + rows - 1) / rows;
    if (poly_deg >= 2) {
      MCP_ENSURE_MAX_LDS(nll_grad_rows_kernel<2>);
      hipLaunchKernelGGL(nll_grad_rows_kernel<2>, dim3(slab_rows, G), dim3(NG_NT), nll_grad_rows_lds(N, D), st, kns, N, X, ws, G, L, rows);
    } else if (poly_deg == 1) {
      MCP_ENSURE_MAX_LDS(nll_grad_rows_kernel<1>);
      hipLaunchKernelGGL(nll_grad_rows_kernel<1>, dim3(slab_rows, G), dim3(NG_NT), nll_grad_rows_lds(N, D), st, kns, N, X, ws, G, L, rows);
    } else {
      MCP_ENSURE_MAX_LDS(nll_grad_rows_kernel<0>);
      hipLaunchKernelGGL(nll_grad_rows_kernel<0>, dim3(slab_rows, G), dim3(NG_NT), nll_grad_rows_lds(N, D), st, kns, N, X, ws, G, L, rows);
    }
  } else {
    MCP_ENSURE_MAX_LDS(nll_grad_batch_kernel);
    hipLaunchKernelGGL(nll_grad_batch_kernel, dim3(N, G), dim3(256), sizeof(double) * (4 * (size_t)N + 256), st, kns, N, X, ws, G, L);
  }
  MCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(nll_finish_kernel, dim3(G), dim3(256), 0, st, b, G, N, D, poly_deg, ard, ws, L, slab_rows);
  MCP_LAUNCH_CHECK();
  return MCP_OK;
}
