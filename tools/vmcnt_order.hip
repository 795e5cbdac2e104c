// vmcnt_order.hip -- do global loads and stores retire from vmcnt in issue order on gfx950?
// Each wave issues a cold 16-byte load (1 GiB buffer, pseudo-random lines), then K small stores
// to a hot line, then `s_waitcnt vmcnt(K)` -- "all but the K youngest are done" -- and checks
// the loaded value.  If stores could retire ahead of the older load, the wait would pass with
// the load still in flight and the register would hold the poison it was pre-set to.
// Not part of the product; profiles/round2_vmcnt_order.txt holds the result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void fill(u32x4 *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = u32x4{(uint32_t)i, (uint32_t)(i >> 32) ^ 0x9e3779b9u, (uint32_t)i * 2654435761u, 0x600dbeefu};
}

template <int K>
__global__ __launch_bounds__(256) void probe(const u32x4 *big, size_t n, uint32_t *hot, unsigned long long *bad, int iters) {
  const int lane = threadIdx.x & 63;
  uint64_t s = (uint64_t)(blockIdx.x * 256 + threadIdx.x) * 0x9e3779b97f4a7c15ull + 12345;
  unsigned long long wrong = 0;
  uint32_t *h = hot + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
  for (int it = 0; it < iters; ++it) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const size_t idx = (size_t)((s >> 20) % n);
    const u32x4 *src = big + idx;
    u32x4 v = {0xdeadu, 0xdeadu, 0xdeadu, 0xdeadu};
    const uint32_t d = (uint32_t)it;
    if (K == 1)
      asm volatile("global_load_dwordx4 %0, %1, off\n\tglobal_store_dword %2, %3, off\n\ts_waitcnt vmcnt(1)"
                   : "+v"(v) : "v"(src), "v"(h), "v"(d) : "memory");
    else
      asm volatile("global_load_dwordx4 %0, %1, off\n\tglobal_store_dword %2, %3, off\n\tglobal_store_short %2, %3, off nt\n\t"
                   "global_store_byte %2, %3, off offset:2 nt\n\tglobal_store_dword %2, %3, off\n\ts_waitcnt vmcnt(4)"
                   : "+v"(v) : "v"(src), "v"(h), "v"(d) : "memory");
    if (v.x != (uint32_t)idx || v.w != 0x600dbeefu) ++wrong;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (wrong) atomicAdd(bad, wrong);
}

int main() {
  const size_t n = (size_t)1 << 26;  // 1 GiB of 16-byte items
  u32x4 *big; uint32_t *hot; unsigned long long *bad;
  CK(hipMalloc(&big, n * 16)); CK(hipMalloc(&hot, 4096 * 4 * 64 * 4)); CK(hipMalloc(&bad, 8));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, big, n);
  CK(hipMemset(bad, 0, 8)); CK(hipDeviceSynchronize());
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe<1>, dim3(4096), dim3(256), 0, 0, big, n, hot, bad, iters);
    hipLaunchKernelGGL(probe<4>, dim3(4096), dim3(256), 0, 0, big, n, hot, bad, iters);
  }
  CK(hipDeviceSynchronize());
  unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
  printf("cold load, then hot stores, then vmcnt(#stores): %llu stale loads in %.0f million lane-trials\n", h,
         6.0 * 4096 * 256 * iters / 1e6);
  return 0;
}
