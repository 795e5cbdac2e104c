// membench.hip -- ceilings for the access shapes the SAT kernels use (not part of the product).
// hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench && tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// linear grid-stride read, 16 B per lane, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void read_linear(const uint4 *__restrict__ src, size_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + (size_t)(U - 1) * 256 < n16; i += stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// tile read like the SAT kernels: a wave reads 1 KiB (64 x 16 B) per row, `rows` rows at a row
// stride of `linesize`, U rows in flight; 4 waves of a block take 4 adjacent 1-KiB strips
template <int U>
__global__ __launch_bounds__(256) void read_tiles(const uint8_t *__restrict__ src, int linesize, int height,
                                                  int nstrips, int rows, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (strip >= nstrips) return;
  const int y0 = blockIdx.y * rows;
  const int y1 = min(y0 + rows, height);
  uint32_t acc = 0;
  for (int y = y0; y < y1; y += U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = (y + u < y1) ? *reinterpret_cast<const uint4 *>(src + (size_t)(y + u) * linesize + (size_t)strip * 1024 + lane * 16)
                          : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// linear write, 16 B per lane
__global__ __launch_bounds__(256) void write_linear(uint4 *__restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n16; i += stride) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

template <class F>
float time_us(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best; sum += ms;
  }
  printf("   best %.1f us, mean %.1f us", best * 1e3f, sum / reps * 1e3f);
  return best * 1e3f;
}

int main() {
  const int W = 7680, H = 3840, NF = 24;               // 24 distinct frames so nothing is cache-resident
  const size_t fb = (size_t)W * H * 4;
  uint8_t *frames; uint32_t *sink; uint4 *out;
  CK(hipMalloc(&frames, fb * NF)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, (size_t)W * H * 12));
  CK(hipMemset(frames, 1, fb * NF));
  int f = 0;
  for (int grid : {1024, 2048, 4096, 8192}) {
    printf("read_linear<8> grid %d:", grid);
    float us = time_us([&] { hipLaunchKernelGGL(read_linear<8>, dim3(grid), dim3(256), 0, 0, (const uint4 *)(frames + fb * (f++ % NF)), fb / 16, sink); }, 20);
    printf("  -> %.2f TB/s\n", fb / us / 1e6);
  }
  for (int rows : {16, 32, 64, 128}) {
    printf("read_tiles<8> rows %d:", rows);
    float us = time_us([&] { hipLaunchKernelGGL(read_tiles<8>, dim3(8, (H + rows - 1) / rows), dim3(256), 0, 0, frames + fb * (f++ % NF), W * 4, H, 30, rows, sink); }, 20);
    printf("  -> %.2f TB/s\n", fb / us / 1e6);
  }
  for (int rows : {32, 64}) {
    printf("read_tiles<16> rows %d:", rows);
    float us = time_us([&] { hipLaunchKernelGGL(read_tiles<16>, dim3(8, (H + rows - 1) / rows), dim3(256), 0, 0, frames + fb * (f++ % NF), W * 4, H, 30, rows, sink); }, 20);
    printf("  -> %.2f TB/s\n", fb / us / 1e6);
  }
  {
    printf("read same frame again (MALL-resident?) tiles<8> rows 32:");
    float us = time_us([&] { hipLaunchKernelGGL(read_tiles<8>, dim3(8, 120), dim3(256), 0, 0, frames, W * 4, H, 30, 32, sink); }, 20);
    printf("  -> %.2f TB/s\n", fb / us / 1e6);
  }
  {
    // does a big write slow the NEXT kernel's cold read (deferred write-back)?
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f, sum = 0;
    for (int r = 0; r < 12; ++r) {
      hipLaunchKernelGGL(write_linear, dim3(8192), dim3(256), 0, 0, out, (size_t)W * H * 12 / 16);
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(read_tiles<8>, dim3(8, 120), dim3(256), 0, 0, frames + fb * (f++ % NF), W * 4, H, 30, 32, sink);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; sum += ms;
    }
    printf("read_tiles<8> rows 32 right after a 354 MB write: best %.1f us, mean %.1f us\n", best * 1e3f, sum / 12 * 1e3f);
  }
  for (int grid : {2048, 8192}) {
    printf("write_linear 354 MB grid %d:", grid);
    float us = time_us([&] { hipLaunchKernelGGL(write_linear, dim3(grid), dim3(256), 0, 0, out, (size_t)W * H * 12 / 16); }, 10);
    printf("  -> %.2f TB/s\n", (double)W * H * 12 / us / 1e6);
  }
  return 0;
}
