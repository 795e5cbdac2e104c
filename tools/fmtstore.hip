// fmtstore -- does buffer_store_format_xyz through an 8_8_8_8_UINT descriptor write three bytes
// of a four-byte texel and leave the fourth alone (what OpenCL's uchar3 `.xyz` store needs)?
// Prints the first pixels and counts texels whose 4th byte changed; times it against the
// short + byte pair and a whole-dword store on a 4272x2144 frame.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t *dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = i & 255, g = (i >> 8) & 255, b = (i * 7) & 255;
  if (MODE == 0) {
    u32x4 rs;
    const uint64_t p = (uint64_t)dst;
    rs.x = (uint32_t)p;
    rs.y = (uint32_t)(p >> 32) | (4u << 16);  // stride 4 (bits 61:48 of the 128-bit word)
    rs.z = (uint32_t)n;                        // num_records (elements, structured)
    rs.w = 4u | (5u << 3) | (6u << 6) | (7u << 9) | (4u << 12) | (10u << 15);  // xyzw, UINT, 8_8_8_8
    rs.x = __builtin_amdgcn_readfirstlane(rs.x);
    rs.y = __builtin_amdgcn_readfirstlane(rs.y);
    rs.z = __builtin_amdgcn_readfirstlane(rs.z);
    rs.w = __builtin_amdgcn_readfirstlane(rs.w);
    asm volatile("buffer_store_format_xyz %0, %1, %2, 0 idxen" ::"v"(u32x3{r, g, b}), "v"((uint32_t)i), "s"(rs) : "memory");
  } else if (MODE == 1) {
    asm volatile("global_store_short %0, %1, %3\n\tglobal_store_byte %0, %2, %3 offset:2" ::"v"((uint32_t)i * 4), "v"(r | (g << 8)), "v"(b), "s"(dst) : "memory");
  } else {
    asm volatile("global_store_dword %0, %1, %2" ::"v"((uint32_t)i * 4), "v"(r | (g << 8) | (b << 16)), "s"(dst) : "memory");
  }
}

int main() {
  const int n = 4272 * 2144;
  uint8_t *d;
  hipMalloc(&d, (size_t)n * 4);
  std::vector<uint8_t> h((size_t)n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(d, 0xA5, (size_t)n * 4);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    hipMemcpy(h.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost);
    long bad_rgb = 0, byte3_changed = 0;
    for (int i = 0; i < n; ++i) {
      const uint8_t r = i & 255, g = (i >> 8) & 255, b = (i * 7) & 255;
      if (h[4 * i] != r || h[4 * i + 1] != g || h[4 * i + 2] != b) ++bad_rgb;
      if (h[4 * i + 3] != 0xA5) ++byte3_changed;
    }
    printf("mode %d (%s): %.1f us, wrong rgb %ld, byte 3 changed %ld; first texel %02x %02x %02x %02x, texel 1000 %02x %02x %02x %02x\n",
           mode, mode == 0 ? "buffer_store_format_xyz" : mode == 1 ? "short + byte" : "dword", best * 1e3f,
           bad_rgb, byte3_changed, h[0], h[1], h[2], h[3], h[4000], h[4001], h[4002], h[4003]);
  }
  return 0;
}
