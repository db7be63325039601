// Diagnostic microbenchmark (not part of the product): per-CU streaming rate of an L2-resident
// matrix, the access pattern of the rollout's phase V.  Every workgroup re-reads the same
// rows x pitch fp64 matrix `iters` times.  Variants: bytes per lane (8/16), waves per workgroup,
// loads in flight, row-strided (column chunk) vs fully contiguous.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_stream_bench.hip -o /tmp/l2bench && /tmp/l2bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// pattern A: like phase V -- wave owns a 64-column chunk (8 B/lane) and walks rows (stride = pitch)
template <int INFLIGHT>
__global__ __launch_bounds__(1024) void colchunk_x2(const double* __restrict__ A, int rows, int pitch, int iters, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nchunk = pitch / 64;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    // units = nchunk * rows, split evenly over waves
    int total = nchunk * rows, L = (total + nw - 1) / nw;
    int u = wv * L, u1 = min(total, u + L);
    while (u < u1) {
      int c = u / rows, j = u - c * rows;
      int jb = min(rows, j + (u1 - u));
      const double* p = A + (size_t)j * pitch + c * 64 + lane;
      int n = jb - j;
      int k = 0;
      for (; k + INFLIGHT <= n; k += INFLIGHT) {
        double v[INFLIGHT];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) v[q] = p[(size_t)(k + q) * pitch];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) acc += v[q];
      }
      for (; k < n; ++k) acc += p[(size_t)k * pitch];
      u += n;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

// pattern B: 128-column chunk, 16 B/lane
template <int INFLIGHT>
__global__ __launch_bounds__(1024) void colchunk_x4(const double* __restrict__ A, int rows, int pitch, int iters, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nchunk = pitch / 128;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    int total = nchunk * rows, L = (total + nw - 1) / nw;
    int u = wv * L, u1 = min(total, u + L);
    while (u < u1) {
      int c = u / rows, j = u - c * rows;
      int jb = min(rows, j + (u1 - u));
      const double2* p = reinterpret_cast<const double2*>(A + (size_t)j * pitch + c * 128) + lane;
      int n = jb - j;
      int k = 0;
      const size_t pp = pitch / 2;
      for (; k + INFLIGHT <= n; k += INFLIGHT) {
        double2 v[INFLIGHT];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) v[q] = p[(size_t)(k + q) * pp];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) acc += v[q].x + v[q].y;
      }
      for (; k < n; ++k) { double2 t = p[(size_t)k * pp]; acc += t.x + t.y; }
      u += n;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

// pattern C: fully contiguous per wave, 16 B/lane
template <int INFLIGHT>
__global__ __launch_bounds__(1024) void contiguous_x4(const double* __restrict__ A, int rows, int pitch, int iters, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t total2 = (size_t)rows * pitch / 2;  // double2 elements
  const size_t per = total2 / nw;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    const double2* p = reinterpret_cast<const double2*>(A) + wv * per + lane;
    size_t n = per / 64;
    size_t k = 0;
    for (; k + INFLIGHT <= n; k += INFLIGHT) {
      double2 v[INFLIGHT];
#pragma unroll
      for (int q = 0; q < INFLIGHT; ++q) v[q] = p[(k + q) * 64];
#pragma unroll
      for (int q = 0; q < INFLIGHT; ++q) acc += v[q].x + v[q].y;
    }
  }
  if (acc == 123.456) out[0] = acc;
}

template <typename K>
static int run(const char* name, K kern, int grid, int threads, const double* A, int rows, int pitch, int iters, double* out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, A, rows, pitch, 2, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, A, rows, pitch, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)rows * pitch * 8.0 * iters;  // per workgroup
  double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-28s grid %4d thr %4d : %8.3f ms  %7.1f GB/s per WG  (%5.1f B/clk @2.4GHz)  aggregate %6.2f TB/s\n", name, grid, threads, ms, gbs,
         gbs / 2.4, gbs * grid / 1e3);
  return 0;
}

int main() {
  const int rows = 600, pitch = 384;  // ~1.84 MB: two 300-row matrices, pitch a multiple of 128
  std::vector<double> h((size_t)rows * pitch, 1.0);
  double *A, *out;
  CHECK(hipMalloc(&A, h.size() * 8));
  CHECK(hipMalloc(&out, 8));
  CHECK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  const int iters = 100;
  for (int grid : {200, 100, 25, 1}) {
    for (int thr : {512, 1024}) {
      run("colchunk_x2 inflight 8", colchunk_x2<8>, grid, thr, A, rows, pitch, iters, out);
      run("colchunk_x2 inflight 16", colchunk_x2<16>, grid, thr, A, rows, pitch, iters, out);
      run("colchunk_x4 inflight 8", colchunk_x4<8>, grid, thr, A, rows, pitch, iters, out);
      run("colchunk_x4 inflight 16", colchunk_x4<16>, grid, thr, A, rows, pitch, iters, out);
      run("contiguous_x4 inflight 8", contiguous_x4<8>, grid, thr, A, rows, pitch, iters, out);
      run("contiguous_x4 inflight 16", contiguous_x4<16>, grid, thr, A, rows, pitch, iters, out);
    }
  }
  return 0;
}
