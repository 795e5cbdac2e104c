// rereadbench.hip -- would a fused reduce+write kernel get its second read of a tile from cache?
// Each wave owns a tile of `rows` x 256 px (1 KiB per row) of a 7680x3840 RGB0 frame:
//   mode 0: read the tile once, write 3 KiB per row (the table writer's traffic)
//   mode 1: read the tile, reduce it, then read it AGAIN (data-dependent on the first pass) and write
//   mode 2: read the tile twice, no writes
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void tile_kernel(const uint8_t *__restrict__ src, uint4 *__restrict__ dst, int linesize,
                                                   int height, int nstrips, int rows) {
  const int lane = threadIdx.x & 63;
  const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (strip >= nstrips) return;
  const int y0 = blockIdx.y * rows, y1 = min(y0 + rows, height);
  const uint8_t *base = src + (size_t)strip * 1024 + lane * 16;
  uint32_t acc = 0;
  for (int y = y0; y < y1; y += 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4 *>(base + (size_t)min(y + u, height - 1) * linesize);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (MODE == 0) {
    for (int y = y0; y < y1; ++y) {
      uint4 *o = dst + ((size_t)y * nstrips + strip) * 192 + lane;
      o[0] = make_uint4(acc, y, 1, 2); o[64] = make_uint4(acc, y, 3, 4); o[128] = make_uint4(acc, y, 5, 6);
    }
    return;
  }
  // second pass: the address depends on the first pass' result (acc & 0 == 0) so it cannot be merged
  const uint8_t *base2 = base + (acc & 0u);
  for (int y = y0; y < y1; y += 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4 *>(base2 + (size_t)min(y + u, height - 1) * linesize);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (y + u >= y1) break;
      acc += v[u].x;
      if (MODE == 1) {
        uint4 *o = dst + ((size_t)(y + u) * nstrips + strip) * 192 + lane;
        o[0] = make_uint4(acc, v[u].y, 1, 2); o[64] = make_uint4(acc, v[u].z, 3, 4); o[128] = make_uint4(acc, v[u].w, 5, 6);
      }
    }
  }
  if (MODE == 2 && acc == 0x12345678u) dst[0] = make_uint4(acc, 0, 0, 0);
}

template <class F>
float time_us(F f, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; sum += ms;
  }
  printf(" best %.1f us, mean %.1f us\n", best * 1e3f, sum / reps * 1e3f);
  return best;
}

int main() {
  const int W = 7680, H = 3840, NF = 12;
  const size_t fb = (size_t)W * H * 4;
  uint8_t *frames; uint4 *out;
  CK(hipMalloc(&frames, fb * NF)); CK(hipMalloc(&out, (size_t)W * H * 12));
  CK(hipMemset(frames, 1, fb * NF));
  int f = 0;
  for (int rows : {8, 16, 32, 64}) {
    const dim3 grid(8, (H + rows - 1) / rows);
    printf("rows %2d  read once + write 3x :", rows);
    time_us([&] { hipLaunchKernelGGL(tile_kernel<0>, grid, dim3(256), 0, 0, frames + fb * (f++ % NF), out, W * 4, H, 30, rows); }, 12);
    printf("rows %2d  read TWICE + write 3x:", rows);
    time_us([&] { hipLaunchKernelGGL(tile_kernel<1>, grid, dim3(256), 0, 0, frames + fb * (f++ % NF), out, W * 4, H, 30, rows); }, 12);
    printf("rows %2d  read twice, no write :", rows);
    time_us([&] { hipLaunchKernelGGL(tile_kernel<2>, grid, dim3(256), 0, 0, frames + fb * (f++ % NF), out, W * 4, H, 30, rows); }, 12);
  }
  return 0;
}
