// Device-side helpers shared by the gfx950 kernels: wave64 DPP reductions, Philox4x32-10,
// and the covariance function of include/mcpilco_hip.h's mcp_kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcpilco_hip.h"

// Experiment build (-DMCPX_WALL_STAMPS: python mc-pilco_amd/build.py --variant wall MCPX_WALL_STAMPS): every diagnostic stamp of the kernels reads the
// constant 100 MHz counter instead of the shader clock -- tools/phase_stamps.py then prints 10 ns ticks.  The shader clock follows the load (fp64
// MFMA phases run at half the clock of the others: profiles/NOTES.md), so shares of CYCLES are not shares of TIME.
#ifdef MCPX_WALL_STAMPS
#define clock64() wall_clock64()
#endif

#define MCP_WAVE 64

#define MCP_LAUNCH_CHECK()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return MCP_ERR_LAUNCH; \
  } while (0)

// Opts a kernel in to the CU's full 160 KiB of dynamic LDS -- once per DEVICE (a process may drive several) and per kernel
// (the static lives in the enclosing function, i.e. per template instantiation); the runtime's refusal is MCP_ERR_LAUNCH.
#define MCP_ENSURE_MAX_LDS(...)                                                                                             \
  do {                                                                                                                      \
    static bool done_[64];                                                                                                  \
    int dev_ = 0;                                                                                                           \
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return MCP_ERR_LAUNCH;                                 \
    if (!done_[dev_]) {                                                                                                     \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              160 * 1024) != hipSuccess)                                                                    \
        return MCP_ERR_LAUNCH;                                                                                              \
      done_[dev_] = true;                                                                                                   \
    }                                                                                                                       \
  } while (0)

namespace mcp {

// ---------------------------------------------------------------------------------------
// wave64 sum of a double via DPP (gfx9 row_shr / row_bcast controls); result broadcast to
// every lane through v_readlane of lane 63.  No LDS traffic.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0x111, 0xf>(v);  // row_shr:1
  v += dpp_take<0x112, 0xf>(v);  // row_shr:2
  v += dpp_take<0x114, 0xf>(v);  // row_shr:4
  v += dpp_take<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_take<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v += dpp_take<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the total
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// N independent wave sums at once: the DPP steps of different values interleave (ILP), so N values
// cost far less than N dependent wave_sum() calls.  Result valid in every lane.
template <int N>
__device__ __forceinline__ void wave_sum_multi(double (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_take<0x111, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_take<0x112, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_take<0x114, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_take<0x118, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_take<0x142, 0xa>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_take<0x143, 0xc>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v[i]), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v[i]), 63);
    v[i] = __hiloint2double(hi, lo);
  }
}

// the first n (wave-uniform, <= N) of N values: same interleaving, no work for the unused tail
template <int N>
__device__ __forceinline__ void wave_sum_first(double (&v)[N], int n) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[i] += dpp_take<0x111, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[i] += dpp_take<0x112, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[i] += dpp_take<0x114, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[i] += dpp_take<0x118, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[i] += dpp_take<0x142, 0xa>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[i] += dpp_take<0x143, 0xc>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i < n) {
      int lo = __builtin_amdgcn_readlane(__double2loint(v[i]), 63);
      int hi = __builtin_amdgcn_readlane(__double2hiint(v[i]), 63);
      v[i] = __hiloint2double(hi, lo);
    }
  }
}

// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] in every lane, with gfx950's row / half swaps (v_permlane16_swap: odd rows of the first operand <->
// even rows of the second; v_permlane32_swap: upper half <-> lower half) on two copies of v: VALU speed, where __shfl_xor (ds_bpermute)
// is an LDS round trip.  Same bits as v + __shfl_xor(v, 16 / 32) (the addition commutes).
__device__ __forceinline__ double sum_xor16(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double sum_xor32(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// within every quad of lanes: lane i receives v of lane (i - k4) & 3 (k4 a constant after unrolling) -- DPP quad_perm, VALU speed, where
// __shfl is an LDS round trip (ds_bpermute).  Used to hand the four words of a Philox block round the four lanes that share it.
__device__ __forceinline__ uint32_t quad_from_back(uint32_t v, int k4) {
  k4 &= 3;
  if (k4 == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x93, 0xf, 0xf, false);  // quad_perm [3,0,1,2]
  if (k4 == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // [2,3,0,1]
  if (k4 == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x39, 0xf, 0xf, false);  // [1,2,3,0]
  return v;
}
// Eight wave sums at once by pair-halving: at every stage two values are folded into one register -- a lane keeps the value its
// own lane-index bit selects and receives the partner lane's copy of the same value -- so the live values halve (8 -> 4 -> 2 -> 1)
// while the lanes that hold a given value's partial sums spread over the wave: 7 exchanges + 3 single-value all-reduce stages
// instead of 8 x 6 (wave_sum_multi).  On return lane l holds the wave total of value (l & 7), in every lane.
// Exchanges: lane ^ 1, ^ 2 by quad_perm; ^ 4 by row_shl:4 / row_shr:4 under bank masks (which also pick, per bank, WHICH of the
// two values is fetched: no select on the receive side); ^ 8 by row_ror:8; ^ 16, ^ 32 by the row / half swap instructions.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_merge(double old, double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_pack8(const double (&v)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  double w[4], x[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // lane ^ 1: quad_perm [1,0,3,2]
    const double keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
    w[i] = keep + dpp_take<0xB1, 0xf>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // lane ^ 2: quad_perm [2,3,0,1]
    const double keep = b1 ? w[2 * i + 1] : w[2 * i], send = b1 ? w[2 * i] : w[2 * i + 1];
    x[i] = keep + dpp_take<0x4E, 0xf>(send);
  }
  // lane ^ 4: banks 0, 2 (bit 2 clear) keep x[0] and fetch x[0] of lane + 4; banks 1, 3 keep x[1] and fetch x[1] of lane - 4
  double y = (b2 ? x[1] : x[0]) + dpp_merge<0x114, 0xA>(dpp_merge<0x104, 0x5>(0.0, x[0]), x[1]);
  y += dpp_take<0x128, 0xf>(y);  // lane ^ 8: row_ror:8
  return sum_xor32(sum_xor16(y));  // lane ^ 16, ^ 32
}

// sin and cos of a state angle.  The library's sincos carries the Payne-Hanek reduction for huge arguments (~200 instructions, 40 bytes
// of scratch per lane) on the one-wave chains that bound the small-swarm kernels (phase S forward, the adjoint chain backward: ~1.5 k
// cycles per step each).  Angles of a rollout are O(10): three-term Cody-Waite reduction by pi/2 (exact first product for |n| < 2^20)
// and the fdlibm kernel polynomials -- absolute error <= 2.1e-16 (<= 2.4 ulp next to the zeros) for |x| < 1e5, checked against long
// double over 2e7 arguments; any lane beyond that (or NaN / inf) sends the whole wave through the library routine.
__device__ __forceinline__ void sincos_fast(double x, double* sp, double* cp) {
  if (__builtin_expect(__ballot(!(fabs(x) < 1.0e5)) != 0ull, 0)) {
    sincos(x, sp, cp);
    return;
  }
  const double fn = rint(x * 6.36619772367581382433e-01);
  double r = fma(-fn, 1.57079632673412561417e+00, x);
  r = fma(-fn, 6.07710050630396597660e-11, r);
  r = fma(-fn, 2.02226624879595063154e-21, r);
  const int n = (int)fn;
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s = (n & 1) ? cr : sr, c = (n & 1) ? sr : cr;
  *sp = (n & 2) ? -s : s;
  *cp = ((n + 1) & 2) ? -c : c;
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: draws depend only on
// (seed, call, global particle, time step, stream, index) -- never on launch geometry.
// ---------------------------------------------------------------------------------------
struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

#define MCP_STREAM_EPS 0u
#define MCP_STREAM_MASK 1u
#define MCP_STREAM_POS 2u  // position measurement noise of the partially-measurable-system rollout

// the launch's noise descriptor with the device counter (mcp_noise.call_dev, if any) folded into `call` -- read ONCE at the top of a kernel; the
// draws below see a plain descriptor (graph replays: include/mcpilco_hip.h)
__device__ __forceinline__ mcp_noise noise_of_launch(const mcp_noise& nz) {
  mcp_noise r = nz;
  if (nz.call_dev) r.call += *nz.call_dev;
  return r;
}

__device__ __forceinline__ u32x4 philox_draw(const mcp_noise& nz, int64_t particle, int t, uint32_t stream, uint32_t index) {
  uint64_t gp = (uint64_t)(particle + nz.particle_offset);
  u32x4 c;
  c.x = (uint32_t)gp;
  c.y = (uint32_t)(gp >> 32) ^ (stream << 30) ^ ((uint32_t)t << 8);
  c.z = index;
  c.w = (uint32_t)nz.call;
  uint32_t k0 = (uint32_t)nz.seed ^ (uint32_t)(nz.call >> 32);
  uint32_t k1 = (uint32_t)(nz.seed >> 32);
  return philox4x32_10(c, k0, k1);
}

// standard normal for (particle, t, gp index g): Box-Muller on two 52-bit uniforms
__device__ __forceinline__ double philox_normal(const mcp_noise& nz, int64_t particle, int t, int g, uint32_t stream = MCP_STREAM_EPS) {
  u32x4 r = philox_draw(nz, particle, t, stream, (uint32_t)g);
  const double two_m52 = 2.220446049250313e-16;
  double u1 = ((double)(((uint64_t)r.x << 20) | (r.y >> 12)) + 0.5) * two_m52;  // (0,1)
  double u2 = ((double)(((uint64_t)r.z << 20) | (r.w >> 12)) + 0.5) * two_m52;
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

// dropout keep decision for (particle, t, basis b): keep with probability 1-p
__device__ __forceinline__ bool philox_keep(const mcp_noise& nz, int64_t particle, int t, int b, uint32_t drop_thresh) {
  u32x4 r = philox_draw(nz, particle, t, MCP_STREAM_MASK, (uint32_t)(b >> 2));
  uint32_t v = (b & 3) == 0 ? r.x : (b & 3) == 1 ? r.y : (b & 3) == 2 ? r.z : r.w;
  return v >= drop_thresh;
}

__host__ __device__ inline uint32_t drop_threshold(double p) {
  double t = p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// ---------------------------------------------------------------------------------------
// covariance function  k(a, b)  with a, b given as strided vectors (element d at a[d*sa])
// ---------------------------------------------------------------------------------------
// the descriptor's three scalars: by value, or from device memory when kn.scal is set (include/mcpilco_hip.h)
__device__ __forceinline__ double kern_lambda(const mcp_kernel& kn) { return kn.scal ? kn.scal[0] : kn.lambda; }
__device__ __forceinline__ double kern_sigma_n2(const mcp_kernel& kn) { return kn.scal ? kn.scal[1] : kn.sigma_n2; }
__device__ __forceinline__ double kern_mean(const mcp_kernel& kn) { return kn.scal ? kn.scal[2] : kn.mean; }

__device__ __forceinline__ double kern_eval(const mcp_kernel& kn, const double* a, int sa, const double* b, int sb) {
  double dist = 0.0;
  for (int d = 0; d < kn.D; ++d) {
    double r = (a[d * sa] - b[d * sb]) * kn.inv_ls[d];
    dist = fma(r, r, dist);
  }
  double k = kern_lambda(kn) * exp(-dist);
  if (kn.poly_deg >= 1) {
    double p1 = kn.w1[kn.D];
    for (int d = 0; d < kn.D; ++d) p1 = fma(kn.w1[d] * a[d * sa], b[d * sb], p1);
    k += p1;
    if (kn.poly_deg >= 2) {
      double pa = 0.0, pb = 0.0;
      for (int d = 0; d < kn.D; ++d) {
        double ab = a[d * sa] * b[d * sb];
        pa = fma(kn.w20[d], ab, pa);
        pb = fma(kn.w21[d], ab, pb);
      }
      k = fma(pa, pb, k);
    }
  }
  return k;
}

__device__ __forceinline__ double kern_diag(const mcp_kernel& kn, const double* a, int sa) {
  double k = kern_lambda(kn);
  if (kn.poly_deg >= 1) {
    double p1 = kn.w1[kn.D];
    for (int d = 0; d < kn.D; ++d) p1 = fma(kn.w1[d] * a[d * sa], a[d * sa], p1);
    k += p1;
    if (kn.poly_deg >= 2) {
      double pa = 0.0, pb = 0.0;
      for (int d = 0; d < kn.D; ++d) {
        double aa = a[d * sa] * a[d * sa];
        pa = fma(kn.w20[d], aa, pa);
        pb = fma(kn.w21[d], aa, pb);
      }
      k = fma(pa, pb, k);
    }
  }
  return k;
}

// tanh(x) = -t / (2 + t),  t = expm1(-2|x|) in (-1, 0]: accurate to a couple of ulp over the whole range (no cancellation near 0,
// saturates cleanly) at a third of the library tanh's dependent latency (tools/f64_math_bench.hip: 704 vs ~250 ticks) -- it sits on the
// critical path of every time step (policy squashing, Policy.py:52-60)
__device__ __forceinline__ double fast_tanh(double x) {
  const double t = expm1(-2.0 * fabs(x));
  return copysign(-t / (2.0 + t), x);
}

__device__ __forceinline__ bool is_bad(double v) { return !(fabs(v) <= 1.79769313486231570815e308); }

}  // namespace mcp
