// sat_walk_nodrain.h -- the read-once strip walker without the per-batch store drain
// (sat_walk2_kernel; RGB0 sources; "sat.walk_variant" 2).  Included once by sat_encode.hip,
// inside its anonymous namespace, after sat_walk_kernel and its helpers.
//
// Same units, tickets, granules and arithmetic as sat_walk_kernel.  What differs is how a wave
// waits.  vmcnt retires loads, stores and LDS-DMA together, in issue order, and a wave of the
// first kernel waits for its next batch of pixels (and for its hand-off poll) with a count of the
// YOUNGER LOADS only -- which also waits for every older store: once per batch the wave idles
// until all 24 table stores of the previous batch are acknowledged.  With five waves per SIMD
// (the three-kernel writer) nobody notices; a strip owner is alone, or one of two, on its SIMD,
// so its batch time is arithmetic + store-acknowledge latency -- and that latency differs from
// box to box (the same library: 80 us per 8K frame on most, 93-98 on some, while the writer ran
// at 73.5 us on all of them).  Here
//   * pixels and hand-off polls arrive by LDS-DMA (global_load_lds: no destination registers the
//     compiler could touch before the data is there), pixels two batches ahead;
//   * the poll of batch t + 1 is issued BEFORE the stores of batch t, so waiting for it leaves
//     those stores in flight;
//   * every wait states exactly how many younger operations may still be outstanding -- the
//     stores included: 59 behind a pixel batch, 24 behind a poll (the order of issue is fixed:
//     see the table in the kernel).
// The exact counts hold for strips that lie wholly inside the frame and from the third batch on;
// a ragged last strip, the first two batches and the timing ablations wait with vmcnt(0).
// A failed poll still takes walk_repoll (a drain); a strip that has fallen one batch behind its
// left neighbour -- which is where failed polls put it -- never fails again.
#pragma once

constexpr int kW2Waves = 4;
constexpr int kW2Slot = kRowUnroll * 1024;  // one batch of a strip: 8 rows x 1 KiB
constexpr int kW2Poll = 256;                // 24 granules (192 bytes), padded
constexpr int kW2Stage = 3 * kStripPx * 4;  // one table row of a strip
constexpr int kW2Wave = 2 * kW2Slot + 2 * kW2Poll + kW2Stage;
typedef const __attribute__((address_space(1))) void *w2_gptr;
typedef __attribute__((address_space(3))) void *w2_lptr;

template <int N>
__device__ __forceinline__ void w2_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// eight rows of a lane's four pixels out of a slot (row r at r KiB), one wait
__device__ __forceinline__ void w2_read_rows(uint32_t addr, u32x4 (&p)[kRowUnroll]) {
  asm volatile(
      "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\t"
      "ds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
      "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
      "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]),
        "=&v"(p[7])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ unsigned long long w2_read_granule(uint32_t addr) {
  unsigned long long g;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(g) : "v"(addr) : "memory");
  return g;
}

__global__ __launch_bounds__(64 * kW2Waves) void sat_walk2_kernel(const EncodeArgs a,
                                                                const WalkBatch wb) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kW2Waves * kW2Wave + 16];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t *wg_ticket = reinterpret_cast<uint32_t *>(lds + kW2Waves * kW2Wave);
  if (threadIdx.x == 0)
    *wg_ticket = __hip_atomic_fetch_add(&a.walk->ticket, 1u, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned long long serial = a.walk->serial;  // written by the previous launch
  const int unit = __builtin_amdgcn_readfirstlane((int)(*wg_ticket * (uint32_t)kW2Waves) + wave);
  // (the ticket, the serial and nothing else came through the compiler's vector-memory
  // bookkeeping; from here to the retirement every vector-memory operation is counted by hand)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (unit < a.walk_units) {
    const int f = unit / a.nstrips;
    const int strip = unit - f * a.nstrips;
    const uint8_t *src = wb.src[f];
    uint32_t *sat = wb.sat[f];
    const int x0 = strip * kStripPx + lane * kLanePx;
    bool need = strip > 0 && !(a.ablate & 64);  // timing experiment: nobody waits
    const bool no_stores = a.ablate & 128;      // timing experiment: the table is not written
    const unsigned long long tag = serial & kWalkTagMask;
    const int nb = a.walk_nbatches;
    unsigned long long *out =
        a.walk_chain + (size_t)unit * nb * kWalkLanes + min(lane, kWalkLanes - 1);
    // the left neighbour's granules (strip 0 polls its own, ignored): 16 bytes = 2 per lane
    const unsigned long long *in_base =
        a.walk_chain + (size_t)(need ? unit - 1 : unit) * nb * kWalkLanes;
    uint8_t *my_ptr = lds + wave * kW2Wave;
    const uint32_t mine = (uint32_t)reinterpret_cast<uintptr_t>(my_ptr);
    const uint32_t stage = mine + 2 * kW2Slot + 2 * kW2Poll;
    const int row_dwords = a.width * 3;
    const int base = strip * kStripPx * 3;
    const int y_last = a.height - 1;
    const bool full_strip = base + 3 * kStripPx <= row_dwords;
    const bool fast = full_strip && !no_stores;  // the exact counts below hold
    const uint8_t *srcp = src + (size_t)min(x0, a.width - kLanePx) * 4;
    uint32_t *rowp = sat + base + lane * 4;  // this lane's first 16 bytes of the current row

    uint32_t acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0;
    uint32_t slow_polls = 0, spun = 0;
    const unsigned long long t_start = (a.ablate & 256) ? __builtin_amdgcn_s_memrealtime() : 0;

    // 8 LDS-DMA loads: batch t's rows (clamped to the last row) into pixel slot sl
    auto issue_pixels = [&](int t, int sl) {
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r) {
        const int y = min(t * kRowUnroll + r, y_last);
        __builtin_amdgcn_global_load_lds((w2_gptr)(srcp + (size_t)y * a.linesize),
                                         (w2_lptr)(my_ptr + sl * kW2Slot + r * 1024), 16, 0, 0);
      }
    };
    // 1 LDS-DMA load (sc1: past the L1): batch t's 24 granules into poll slot sl, 2 per lane
    auto issue_poll = [&](int t, int sl) {
      const unsigned long long *p = in_base + (size_t)min(t, nb - 1) * kWalkLanes + 2 * min(lane, 11);
      if (lane < 12)
        __builtin_amdgcn_global_load_lds((w2_gptr)p, (w2_lptr)(my_ptr + 2 * kW2Slot + sl * kW2Poll),
                                         16, 0, 16);
    };

    // Issue order of a wave's vector-memory operations (L = pixel batch, 8 ops; P = poll, 1;
    // G = granule store, 1; S = table stores of a batch, 24):
    //   prologue  L(0) P(0) L(1)
    //   batch t   [wait L(t)] scan [wait P(t)] resolve  G(t)  L(t+2) P(t+1)  S(t)
    // Behind L(t) (issued in batch t-2): P(t-1) S(t-2) | G(t-1) L(t+1) P(t) S(t-1) = 59 younger
    // operations; behind P(t) (issued in batch t-1, after L(t+1)): S(t-1) = 24.
    auto batch = [&](int t, int sl) {
      const bool steady = fast && t >= 2;
      const int y = t * kRowUnroll;
      if (steady) w2_wait_vm<59>(); else w2_wait_vm<0>();
      u32x4 px[kRowUnroll];
      w2_read_rows(mine + sl * kW2Slot + lane * 16, px);
      uint32_t inc_rg[kRowUnroll], inc_b[kRowUnroll];
      uint32_t tot = 0;
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r) {
        const uint32_t v[4] = {px[r].x, px[r].y, px[r].z, px[r].w};
        uint32_t rb = 0, gg = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          rb += v[k] & 0x00ff00ffu;
          gg += v[k] & 0x0000ff00u;
        }
        inc_rg[r] = wave_scan_incl((rb & 0xffffu) | (gg << 8));
        inc_b[r] = wave_scan_incl(rb >> 16);
        const uint32_t s_rg = (uint32_t)__builtin_amdgcn_readlane((int)inc_rg[r], 63);
        const uint32_t s_b = (uint32_t)__builtin_amdgcn_readlane((int)inc_b[r], 63);
        walk_writelane(tot, s_rg & 0xffffu, 3 * r);
        walk_writelane(tot, s_rg >> 16, 3 * r + 1);
        walk_writelane(tot, s_b, 3 * r + 2);
      }
      // hand-off
      if (steady) w2_wait_vm<24>(); else w2_wait_vm<0>();
      uint32_t lin = 0;
      if (need) {
        unsigned long long g =
            w2_read_granule(mine + 2 * kW2Slot + sl * kW2Poll + min(lane, kWalkLanes - 1) * 8);
        if (!__all((g >> 24) == tag || lane >= kWalkLanes)) {
          ++slow_polls;
          g = walk_repoll(in_base + (size_t)t * kWalkLanes + min(lane, kWalkLanes - 1), tag, lane,
                          spun);
          if (!__all((g >> 24) == tag || lane >= kWalkLanes)) {  // gave up: say so, stop waiting
            if (lane == 0)
              __hip_atomic_store(a.walk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            need = false;
          }
        }
        lin = (uint32_t)g & 0xffffffu;
      }
      if (lane < kWalkLanes)  // (the last strip publishes too: the counts above rely on it)
        walk_store_granule(out + (size_t)t * kWalkLanes,
                           (tag << 24) | (unsigned long long)((lin + tot) & 0xffffffu));
      issue_pixels(t + 2, sl);  // this slot's pixels are in registers now
      issue_poll(t + 1, sl ^ 1);
      // table rows
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r) {
        if (y + r > y_last) break;
        uint32_t c[12];
        unpack_px4(make_uint4(px[r].x, px[r].y, px[r].z, px[r].w), c);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          c[3 * k + 0] += c[3 * k - 3];
          c[3 * k + 1] += c[3 * k - 2];
          c[3 * k + 2] += c[3 * k - 1];
        }
        const uint32_t base_r =
            (inc_rg[r] & 0xffffu) - c[9] + (uint32_t)__builtin_amdgcn_readlane((int)lin, 3 * r);
        const uint32_t base_g =
            (inc_rg[r] >> 16) - c[10] + (uint32_t)__builtin_amdgcn_readlane((int)lin, 3 * r + 1);
        const uint32_t base_b =
            inc_b[r] - c[11] + (uint32_t)__builtin_amdgcn_readlane((int)lin, 3 * r + 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[3 * k + 0] += c[3 * k + 0] + base_r;
          acc[3 * k + 1] += c[3 * k + 1] + base_g;
          acc[3 * k + 2] += c[3 * k + 2] + base_b;
        }
        lds_write_b128(stage + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
        lds_write_b128(stage + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
        lds_write_b128(stage + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
        u32x4 q[3];
        lds_read3_b128(stage + lane * 16, q[0], q[1], q[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int off = k * 256 + lane * 4;
          if (base + off < row_dwords && !no_stores)  // width % 4 == 0 -> whole 16 B in range
            global_store_b128_uncounted_nt(rowp + k * 256, q[k]);
        }
        rowp += row_dwords;  // rows are written strictly in order
      }
    };

    issue_pixels(0, 0);
    issue_poll(0, 0);
    issue_pixels(1, 1);
    for (int t0 = 0; t0 < nb; t0 += 2) {
      batch(t0, 0);
      if (t0 + 1 < nb) batch(t0 + 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // hand back to the compiler's counts
    if ((a.ablate & 256) && lane == 0) {
      ulonglong2 *st = reinterpret_cast<ulonglong2 *>(a.walk_stats + (size_t)unit * 4);
      st[0] = make_ulonglong2(t_start, __builtin_amdgcn_s_memrealtime());
      st[1] = make_ulonglong2(slow_polls, spun);
    }
  }
  if (lane == 0) {
    const uint32_t waves = gridDim.x * kW2Waves;
    const uint32_t before = __hip_atomic_fetch_add(&a.walk->done, 1u, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    if (before == waves - 1) {
      unsigned long long next = serial + 1;
      if ((next & kWalkTagMask) == 0) ++next;
      __hip_atomic_store(&a.walk->serial, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.walk->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.walk->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
