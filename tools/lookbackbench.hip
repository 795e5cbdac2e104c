// lookbackbench.hip -- what would a single-pass table build cost in carry traffic and waiting?
// Emulates its data flow on a 7680x3840 RGB0 frame (not its arithmetic): one wave per tile of
// 256 px x TB rows holds the tile in registers, publishes the tile's column aggregate (768
// dwords) and row aggregate (TB x 3 dwords), obtains the sum of the aggregates of all tiles above
// / to the left by decoupled look-back (tickets from an atomic counter, so every predecessor is
// already running; spins are bounded and report through an error word), publishes its inclusive
// sums and writes 3 KiB per row.  Compared with "no carries at all" (mode 0 of rereadbench: read
// once, write 3x).  The look-back result is checked against a direct sum on the host.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { kEmpty = 0, kAggregate = 1, kInclusive = 2 };
constexpr int kSpinLimit = 1 << 20;

struct Args {
  const uint8_t *src;
  uint4 *dst;
  int linesize, height, nstrips, nbands;
  uint32_t *ticket;
  uint32_t *vflag, *hflag;   // [nbands][nstrips], value = epoch * 4 + state
  uint32_t *colagg, *colinc; // [nbands][nstrips][768]
  uint32_t *rowagg, *rowinc; // [nbands][nstrips][TB*3 padded to 256]
  uint32_t *error;
  uint32_t epoch;
  int use_fence;             // 1: normal memory + __threadfence(); 0: fine-grained memory, waitcnt only
  int do_lookback;           // 0: none, 1: serial walk, 2: windowed (64 predecessors polled at once)
};

__device__ __forceinline__ uint32_t flag_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-uniform wait until the flag of `epoch` reaches at least `state`; returns the state seen
__device__ __forceinline__ uint32_t wait_flag(const Args &a, const uint32_t *p, uint32_t state) {
  for (int spin = 0; spin < kSpinLimit; ++spin) {
    const uint32_t v = flag_load(p);
    if ((v >> 2) == a.epoch && (v & 3u) >= state) return v & 3u;
    __builtin_amdgcn_s_sleep(2);
  }
  atomicOr(a.error, 1u);
  return kInclusive;  // give up: results are wrong, the error word says so
}
__device__ __forceinline__ void publish_fence(const Args &a) {
  if (a.use_fence) __threadfence();
  else __builtin_amdgcn_s_waitcnt(0);  // stores to fine-grained memory: acknowledged = visible
}
__device__ __forceinline__ void consume_fence(const Args &a) {
  if (a.use_fence) __threadfence();
}

// Windowed look-back: lane l polls the flag of predecessor `first - l * stride` (first, first -
// stride, ... down to `last`), until a contiguous run of ready predecessors ends in one with
// inclusive sums (or at the last one).  Returns how many predecessors to add (the nearest
// `count - 1` by their aggregate, the farthest by `*inclusive ? inclusive : aggregate`).
__device__ __forceinline__ int window_wait(const Args &a, const uint32_t *flags, long first,
                                           long stride, int avail, bool *inclusive) {
  const int lane = threadIdx.x & 63;
  const int n = min(avail, 64);
  for (int spin = 0; spin < kSpinLimit; ++spin) {
    uint32_t st = kEmpty;
    if (lane < n) {
      const uint32_t v = flag_load(flags + (first - (long)lane * stride));
      st = (v >> 2) == a.epoch ? (v & 3u) : kEmpty;
    }
    const unsigned long long inc = __ballot(st == kInclusive);
    const unsigned long long rdy = __ballot(st >= kAggregate);
    const int first_inc = inc ? __ffsll((long long)inc) - 1 : 64;
    const int need = min(first_inc + 1, n);
    const unsigned long long mask = need >= 64 ? ~0ull : ((1ull << need) - 1);
    if ((rdy & mask) == mask) {
      *inclusive = first_inc < n;
      return need;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) atomicOr(a.error, 1u);
  *inclusive = true;
  return 0;
}

template <int TB>
__global__ __launch_bounds__(256) void single_pass_kernel(const Args a) {
  __shared__ uint32_t tick[1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) tick[0] = atomicAdd(a.ticket, 4u);
  __syncthreads();
  const int tile = (int)tick[0] + wave;
  if (tile >= a.nstrips * a.nbands) return;
  const int band = tile / a.nstrips, strip = tile - band * a.nstrips;
  const int y0 = band * TB;

  // phase A: the tile
  uint4 px[TB];
  const uint8_t *base = a.src + (size_t)strip * 1024 + lane * 16;
#pragma unroll
  for (int r = 0; r < TB; ++r)
    px[r] = *reinterpret_cast<const uint4 *>(base + (size_t)min(y0 + r, a.height - 1) * a.linesize);

  // phase B: aggregates (12 "column sums" per lane, 3 "row sums" per row held by lane r)
  uint32_t col[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) col[e] = 0;
  uint32_t rowv[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < TB; ++r) {
    col[0] += px[r].x & 0xff; col[1] += (px[r].x >> 8) & 0xff; col[2] += (px[r].x >> 16) & 0xff;
    col[3] += px[r].y & 0xff; col[4] += (px[r].y >> 8) & 0xff; col[5] += (px[r].y >> 16) & 0xff;
    col[6] += px[r].z & 0xff; col[7] += (px[r].z >> 8) & 0xff; col[8] += (px[r].z >> 16) & 0xff;
    col[9] += px[r].w & 0xff; col[10] += (px[r].w >> 8) & 0xff; col[11] += (px[r].w >> 16) & 0xff;
    // a cheap stand-in for the wave reduction of the row: lane r keeps its own pixel sum
    if (lane == r) { rowv[0] = px[r].x & 0xff; rowv[1] = px[r].y & 0xff; rowv[2] = px[r].z & 0xff; }
  }
  const size_t t = (size_t)band * a.nstrips + strip;
  uint32_t cprefix[12], rprefix[3] = {0, 0, 0};
#pragma unroll
  for (int e = 0; e < 12; ++e) cprefix[e] = 0;

  if (a.do_lookback) {
    uint4 *ca = reinterpret_cast<uint4 *>(a.colagg + t * 768) + lane * 3;
    ca[0] = make_uint4(col[0], col[1], col[2], col[3]);
    ca[1] = make_uint4(col[4], col[5], col[6], col[7]);
    ca[2] = make_uint4(col[8], col[9], col[10], col[11]);
    if (lane < TB) {
      uint32_t *ra = a.rowagg + t * 256 + lane * 3;
      ra[0] = rowv[0]; ra[1] = rowv[1]; ra[2] = rowv[2];
    }
    publish_fence(a);
    if (lane == 0) {
      flag_store(a.vflag + t, a.epoch * 4 + kAggregate);
      flag_store(a.hflag + t, a.epoch * 4 + kAggregate);
    }
    // phase C: vertical look-back
    if (a.do_lookback == 2) {
      int b_hi = band - 1;
      while (b_hi >= 0) {
        bool inc;
        const int need = window_wait(a, a.vflag, (long)b_hi * a.nstrips + strip, a.nstrips, b_hi + 1, &inc);
        if (need == 0) break;
        for (int l = 0; l < need; ++l) {  // loads of one window are independent of each other
          const size_t p = (size_t)(b_hi - l) * a.nstrips + strip;
          const uint4 *v = reinterpret_cast<const uint4 *>(((inc && l == need - 1) ? a.colinc : a.colagg) + p * 768) + lane * 3;
          const uint4 v0 = v[0], v1 = v[1], v2 = v[2];
          cprefix[0] += v0.x; cprefix[1] += v0.y; cprefix[2] += v0.z; cprefix[3] += v0.w;
          cprefix[4] += v1.x; cprefix[5] += v1.y; cprefix[6] += v1.z; cprefix[7] += v1.w;
          cprefix[8] += v2.x; cprefix[9] += v2.y; cprefix[10] += v2.z; cprefix[11] += v2.w;
        }
        if (inc) break;
        b_hi -= need;
      }
    } else
    for (int b = band - 1; b >= 0; --b) {
      const size_t p = (size_t)b * a.nstrips + strip;
      const uint32_t st = wait_flag(a, a.vflag + p, kAggregate);
      consume_fence(a);
      const uint4 *v = reinterpret_cast<const uint4 *>((st == kInclusive ? a.colinc : a.colagg) + p * 768) + lane * 3;
      const uint4 v0 = v[0], v1 = v[1], v2 = v[2];
      cprefix[0] += v0.x; cprefix[1] += v0.y; cprefix[2] += v0.z; cprefix[3] += v0.w;
      cprefix[4] += v1.x; cprefix[5] += v1.y; cprefix[6] += v1.z; cprefix[7] += v1.w;
      cprefix[8] += v2.x; cprefix[9] += v2.y; cprefix[10] += v2.z; cprefix[11] += v2.w;
      if (st == kInclusive) break;
    }
    {
      uint4 *ci = reinterpret_cast<uint4 *>(a.colinc + t * 768) + lane * 3;
      ci[0] = make_uint4(cprefix[0] + col[0], cprefix[1] + col[1], cprefix[2] + col[2], cprefix[3] + col[3]);
      ci[1] = make_uint4(cprefix[4] + col[4], cprefix[5] + col[5], cprefix[6] + col[6], cprefix[7] + col[7]);
      ci[2] = make_uint4(cprefix[8] + col[8], cprefix[9] + col[9], cprefix[10] + col[10], cprefix[11] + col[11]);
      publish_fence(a);
      if (lane == 0) flag_store(a.vflag + t, a.epoch * 4 + kInclusive);
    }
    // horizontal look-back
    if (a.do_lookback == 2) {
      int s_hi = strip - 1;
      while (s_hi >= 0) {
        bool inc;
        const int need = window_wait(a, a.hflag, (long)band * a.nstrips + s_hi, 1, s_hi + 1, &inc);
        if (need == 0) break;
        for (int l = 0; l < need; ++l) {
          const size_t p = (size_t)band * a.nstrips + (s_hi - l);
          if (lane < TB) {
            const uint32_t *v = ((inc && l == need - 1) ? a.rowinc : a.rowagg) + p * 256 + lane * 3;
            rprefix[0] += v[0]; rprefix[1] += v[1]; rprefix[2] += v[2];
          }
        }
        if (inc) break;
        s_hi -= need;
      }
    } else
    for (int s = strip - 1; s >= 0; --s) {
      const size_t p = (size_t)band * a.nstrips + s;
      const uint32_t st = wait_flag(a, a.hflag + p, kAggregate);
      consume_fence(a);
      if (lane < TB) {
        const uint32_t *v = (st == kInclusive ? a.rowinc : a.rowagg) + p * 256 + lane * 3;
        rprefix[0] += v[0]; rprefix[1] += v[1]; rprefix[2] += v[2];
      }
      if (st == kInclusive) break;
    }
    if (lane < TB) {
      uint32_t *ri = a.rowinc + t * 256 + lane * 3;
      ri[0] = rprefix[0] + rowv[0]; ri[1] = rprefix[1] + rowv[1]; ri[2] = rprefix[2] + rowv[2];
    }
    publish_fence(a);
    if (lane == 0) flag_store(a.hflag + t, a.epoch * 4 + kInclusive);
  }

  // phase D: 3 KiB per row
  const uint32_t mix = cprefix[0] + cprefix[5] + cprefix[11] + rprefix[0] + rprefix[2];
#pragma unroll
  for (int r = 0; r < TB; ++r) {
    if (y0 + r >= a.height) break;
    uint4 *o = a.dst + ((size_t)(y0 + r) * a.nstrips + strip) * 192 + lane;
    o[0] = make_uint4(px[r].x + mix, px[r].y, 1, 2);
    o[64] = make_uint4(px[r].z + mix, px[r].w, 3, 4);
    o[128] = make_uint4(px[r].x ^ mix, 5, 6, 7);
  }
}

template <int TB>
int run(int use_fence, int finegrained) {
  const int W = 7680, H = 3840, nstrips = W / 256, nbands = (H + TB - 1) / TB;
  const size_t ntiles = (size_t)nstrips * nbands;
  uint8_t *src; uint4 *dst;
  CK(hipMalloc(&src, (size_t)W * H * 4));
  CK(hipMalloc(&dst, (size_t)W * H * 12));
  std::vector<uint8_t> h((size_t)W * H * 4);
  uint32_t s = 12345;
  for (auto &b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
  CK(hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice));
  Args a{};
  a.src = src; a.dst = dst; a.linesize = W * 4; a.height = H; a.nstrips = nstrips; a.nbands = nbands;
  a.use_fence = use_fence;
  const size_t carry_bytes = ntiles * (768 * 2 + 256 * 2) * 4, flag_bytes = ntiles * 2 * 4 + 64;
  void *carry, *flags;
  if (finegrained) {
    CK(hipExtMallocWithFlags(&carry, carry_bytes, hipDeviceMallocFinegrained));
    CK(hipExtMallocWithFlags(&flags, flag_bytes, hipDeviceMallocFinegrained));
  } else {
    CK(hipMalloc(&carry, carry_bytes));
    CK(hipMalloc(&flags, flag_bytes));
  }
  CK(hipMemset(flags, 0, flag_bytes));
  a.colagg = (uint32_t *)carry; a.colinc = a.colagg + ntiles * 768;
  a.rowagg = a.colinc + ntiles * 768; a.rowinc = a.rowagg + ntiles * 256;
  a.vflag = (uint32_t *)flags; a.hflag = a.vflag + ntiles; a.ticket = a.hflag + ntiles; a.error = a.ticket + 1;
  const dim3 grid((unsigned)((ntiles + 3) / 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int lb = 0; lb < 3; ++lb) {
    if (use_fence && lb == 2) continue;
    a.do_lookback = lb;
    float best = 1e30f, sum = 0; const int reps = 12;
    for (int r = 0; r < reps + 2; ++r) {
      a.epoch += 1;
      CK(hipMemsetAsync(a.ticket, 0, 4, 0));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(single_pass_kernel<TB>, grid, dim3(256), 0, 0, a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    uint32_t err = 0; CK(hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost));
    printf("TB=%d %s %s lookback=%d: avg %.1f us  best %.1f us  error=%u\n", TB,
           finegrained ? "finegrained" : "coarse", use_fence ? "fences" : "waitcnt", lb,
           sum / reps * 1e3, best * 1e3, err);
  }
  // check: the inclusive column vector of the last band of strip 3 against a host sum
  std::vector<uint32_t> got(768);
  CK(hipMemcpy(got.data(), a.colinc + ((size_t)(nbands - 1) * nstrips + 3) * 768, 768 * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int x = 0; x < 256; ++x)
    for (int c = 0; c < 3; ++c) {
      uint32_t want = 0;
      for (int y = 0; y < H; ++y) want += h[((size_t)y * W + 3 * 256 + x) * 4 + c];
      // rows past the frame are clamped copies of the last row in the last band
      for (int y = H; y < nbands * TB; ++y) want += h[((size_t)(H - 1) * W + 3 * 256 + x) * 4 + c];
      if (got[x * 3 + c] != want) ++bad;
    }
  std::vector<uint32_t> rgot(256);
  CK(hipMemcpy(rgot.data(), a.rowinc + ((size_t)5 * nstrips + (nstrips - 1)) * 256, 256 * 4, hipMemcpyDeviceToHost));
  for (int r = 0; r < TB; ++r) {
    uint32_t want[3] = {0, 0, 0};
    for (int sidx = 0; sidx < nstrips; ++sidx) {
      const size_t p = ((size_t)(5 * TB + r) * W + sidx * 256 + r * 4) * 4;  // lane r's first three pixels' red
      want[0] += h[p]; want[1] += h[p + 4]; want[2] += h[p + 8];
    }
    for (int c = 0; c < 3; ++c) if (rgot[r * 3 + c] != want[c]) ++bad;
  }
  printf("   look-back sums %s (%d mismatches)\n", bad ? "WRONG" : "correct", bad);
  (void)hipFree(src); (void)hipFree(dst); (void)hipFree(carry); (void)hipFree(flags);
  return 0;
}

int main() {
  if (run<64>(1, 0)) return 1;
  if (run<64>(0, 1)) return 1;
  if (run<32>(1, 0)) return 1;
  if (run<32>(0, 1)) return 1;
  return 0;
}
