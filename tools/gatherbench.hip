// gatherbench.hip -- is the SAT sampler bound by its access shape?  Compares, on a 7680x3840x12 B
// table, (a) the walker's shape: per touched row one 12-byte gather per lattice column, and
// (b) streaming the same rows with coalesced 16-byte loads.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "f360.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void gather_rows(const uint32_t *__restrict__ sat, int W, const int *__restrict__ xs, int ncols,
                                                   const int *__restrict__ ys, int nrows, int rows_per_wave, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const int c = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
  if ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 >= ncols) return;
  const int x = xs[min(c, ncols - 1)];
  const int r0 = blockIdx.y * rows_per_wave;
  uint32_t acc = 0;
  uint3 v[8];
  for (int rb = r0; rb < min(r0 + rows_per_wave, nrows); rb += 8) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int y = ys[min(rb + r, nrows - 1)];
      const uint32_t *p = sat + ((size_t)y * W + x) * 3;
      v[r] = make_uint3(p[0], p[1], p[2]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += v[r].x ^ v[r].y ^ v[r].z;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// each wave streams a 3 KiB segment (256 texels) of `rows_per_wave` listed rows
__global__ __launch_bounds__(256) void stream_rows(const uint32_t *__restrict__ sat, int W, const int *__restrict__ ys, int nrows,
                                                   int rows_per_wave, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);  // 30 segments per row
  if (seg >= W / 256) return;
  const int r0 = blockIdx.y * rows_per_wave;
  uint32_t acc = 0;
  for (int rb = r0; rb < min(r0 + rows_per_wave, nrows); rb += 4) {
    uint4 v[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ys[min(rb + r, nrows - 1)];
      const uint4 *p = reinterpret_cast<const uint4 *>(sat + ((size_t)y * W + seg * 256) * 3);
#pragma unroll
      for (int q = 0; q < 3; ++q) v[r][q] = p[q * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc += v[r][q].x ^ v[r][q].y ^ v[r][q].z ^ v[r][q].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
float time_us(F f, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best; sum += ms;
  }
  printf("   best %.1f us, mean %.1f us\n", best * 1e3f, sum / reps * 1e3f);
  return best * 1e3f;
}

int main() {
  const int W = 7680, H = 3840, Wr = 4272, Hr = 2144, NT = 3;
  std::vector<int16_t> gx(Wr + 1), gy(Hr + 1);
  f360_tables_satdec_grid_axis(gx.data(), Wr, W);
  f360_tables_satdec_grid_axis(gy.data(), Hr, H);
  std::vector<int> xs, ys;
  for (int i = 0; i <= Wr; ++i) { int x = W / 2 + gx[i]; x = ((x % W) + W) % W; xs.push_back(x); }
  for (int j = 0; j <= Hr; ++j) { int y = H / 2 + gy[j]; if (y >= 0 && y < H) ys.push_back(y); }
  printf("lattice: %zu columns, %zu rows\n", xs.size(), ys.size());
  uint32_t *sat, *sink; int *dxs, *dys;
  const size_t tb = (size_t)W * H * 12;
  CK(hipMalloc(&sat, tb * NT)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&dxs, xs.size() * 4)); CK(hipMalloc(&dys, ys.size() * 4));
  CK(hipMemset(sat, 1, tb * NT));
  CK(hipMemcpy(dxs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dys, ys.data(), ys.size() * 4, hipMemcpyHostToDevice));
  int t = 0;
  const int ncols = (int)xs.size(), nrows = (int)ys.size();
  for (int rpw : {8, 16, 32}) {
    printf("gather (walker shape) rows/wave %d:", rpw);
    time_us([&] { hipLaunchKernelGGL(gather_rows, dim3((ncols + 255) / 256, (nrows + rpw - 1) / rpw), dim3(256), 0, 0,
                                     sat + (tb / 4) * (t++ % NT), W, dxs, ncols, dys, nrows, rpw, sink); }, 12);
  }
  for (int rpw : {4, 8, 16}) {
    printf("stream the same %d rows, rows/wave %d (%.0f MB):", nrows, rpw, nrows * 92160.0 / 1e6);
    time_us([&] { hipLaunchKernelGGL(stream_rows, dim3(8, (nrows + rpw - 1) / rpw), dim3(256), 0, 0,
                                     sat + (tb / 4) * (t++ % NT), W, dys, nrows, rpw, sink); }, 12);
  }
  return 0;
}
