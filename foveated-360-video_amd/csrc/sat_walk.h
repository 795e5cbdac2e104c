// sat_walk.h -- the read-once batched SAT encoder: strip owners walk down their frames
// (sat_walk_kernel), with or without the one-pass encode + sample machinery of sat_fuse_dev.h.
// Instantiated by sat_walk.hip (tables only) and sat_fuse.hip (tables + reduced frames).
#pragma once

#include "sat_common.h"

namespace f360 {
namespace sat {

// ---- read-once batched encoder: strip owners walk down their frames --------------
// f360_sat_encode_batch with enough frames to fill the device.  One wave owns a 256-pixel strip
// of ONE frame and walks all its rows top to bottom: the vertical running sums never leave its
// registers, the strip-local row prefix is the DPP scan of the table writer, and the only thing
// a strip needs from outside is, per row, the sum of that row over the strips to its left
// (3 dwords).  That prefix travels left to right from strip to strip through global memory:
// per batch of 8 rows a strip reads its left neighbour's 24 running row prefixes, adds its own
// 24 row sums and publishes the result for its right neighbour BEFORE it does the heavy part
// (scan results are kept in registers), so the chain advances at hand-off latency, not at
// table-writing speed.  The frame is read once: no reducer pass, no carry kernel
// (sat_encoder_encode_kernels.cl:44-74 done in one pass over the pixels).
//
// Hand-off (MI355X_MICROARCH.md "Workgroup dispatch ... visibility", form R2): the payload IS
// the flag.  A granule is one naturally aligned 8-byte word {tag: 40 bits | row prefix: 24
// bits} written by one lane with ONE sc1 (write-through) store and polled with sc1 loads that
// bypass the reader's L1; the tag is the launch's serial number, so entries of earlier launches
// never match and nothing is cleared between launches.  A row prefix is < 65536 * 255 < 2^24.
// No fence, no release, no acquire: a granule is either this launch's (tag matches) or not yet.
//
// Forward progress: workgroups draw a ticket (atomic counter) and unit = ticket order, frames
// major, strips left to right.  A strip only ever waits for the unit one ticket position before
// it, whose workgroup drew its ticket earlier, i.e. is resident and running; strip 0 of a frame
// waits for nobody.  Every wait is bounded (a.walk_spin polls, a tenth of a second), and a strip
// whose wait runs into the bound does not guess: it leaves the hand-off chain and finishes its
// rows alone, computing the row sums of everything to its left from the source pixels itself
// (slow -- strip s reads s strips per batch -- and exact), publishes correct prefixes for its
// right neighbour as before, and counts itself in *walk_err (host-mapped;
// f360_debug_walk_recoveries).  So the grid always drains AND the tables are always right: a
// timeout costs time, never a result.  The state words live in device memory and are advanced by the launch itself
// (the last wave to retire zeroes the ticket and bumps the serial), so nothing per launch
// comes from the host: a captured launch replays correctly.
struct WalkState {
  uint32_t ticket;            // workgroups of the current launch that have started
  uint32_t done;              // waves of the current launch that have retired
  unsigned long long serial;  // launch number = tag of this launch's granules; never 0
};
constexpr int kWalkFrames = 64;              // frames per launch
#ifndef F360_WALK_WAVES
#define F360_WALK_WAVES 4
#endif
constexpr int kWalkWaves = F360_WALK_WAVES;  // strip owners (consecutive units) per workgroup
constexpr int kWalkLanes = 3 * kRowUnroll;   // granules per batch: 8 rows x 3 channels
constexpr uint32_t kWalkSpinDefault = 1u << 16;  // ~2 us per poll under load
constexpr unsigned long long kWalkTagMask = (1ull << 40) - 1;

struct WalkBatch {
  const uint8_t *src[kWalkFrames];  // packed source, or the luma plane
  uint32_t *sat[kWalkFrames];
  const uint8_t *u[kWalkFrames], *v[kWalkFrames];  // planar sources
};

// v[lane] = s (a wave-uniform value) for one constant lane
__device__ __forceinline__ void walk_writelane(uint32_t &v, uint32_t s, int lane) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(lane));
}

__device__ __forceinline__ void walk_store_granule(unsigned long long *p, unsigned long long g) {
  // one lane, one 8-byte write-through store; hidden from the compiler's vmcnt bookkeeping
  // like the table stores (global_store_b128_uncounted)
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(g) : "memory");
}

// The slow path of a hand-off wait: the first poll (issued a batch's worth of scans earlier)
// did not find this launch's tag in all 24 granules.  Self-contained asm loads with their own
// full wait, so the compiler's count of the pixel loads in flight is the fast path's.
__device__ __forceinline__ unsigned long long walk_repoll(const unsigned long long *p,
                                                          unsigned long long tag, int lane,
                                                          uint32_t limit, uint32_t &spun) {
  unsigned long long g = 0;
  for (uint32_t spins = 0; spins < limit; ++spins) {
    ++spun;
    __builtin_amdgcn_s_sleep(4);
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(g)
                 : "v"(p)
                 : "memory");
    if (__all((g >> 24) == tag || lane >= kWalkLanes)) break;
  }
  return g;  // the caller checks the tags once more: a mismatch now means the bound was hit
}

// A batch of the walker's pixel loads: read exactly once, by exactly one wave
template <int SRC>
__device__ __forceinline__ void walk_load_batch(const EncodeArgs &a, const EncodeFrame &fr,
                                                RowBatch<SRC> &b, int y, int x0, int y_last) {
#ifdef F360_WALK_NT_LOADS
  if constexpr (SRC == kSrcRgb0) {
    const int xc = min(x0, a.width - kLanePx);
    const uint8_t *p = fr.src + (size_t)xc * 4;
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) {
      const u32x4 v = __builtin_nontemporal_load(
          reinterpret_cast<const u32x4 *>(p + (size_t)min(y + r, y_last) * a.linesize));
      b.raw[r] = make_uint4(v.x, v.y, v.z, v.w);
    }
    return;
  }
#endif
  reduce_load_batch<SRC>(a, fr, b, y, x0, y_last);
}

// The same batch for a strip that finishes alone (left_of_me): every load an asm of its own with
// its own full wait, like walk_repoll's -- a load the compiler can see inside the row loop makes
// it drain the prefetched batches at the loop head (the ISA guard found exactly that).
__device__ __forceinline__ uint4 walk_alone_load16(const void *p) {
  uint4 v;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t walk_alone_load4(const void *p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t walk_alone_load2(const void *p) {
  uint32_t v;
  asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
template <int SRC>
__device__ __forceinline__ void walk_alone_load_batch(const EncodeArgs &a, const EncodeFrame &fr,
                                                      RowBatch<SRC> &b, int y, int x0,
                                                      int y_last) {
  const int xc = min(x0, a.width - kLanePx);
  if constexpr (SRC >= kSrcYuvSwsC) {  // load_yuv_batch
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.y4[r] = walk_alone_load4(fr.y + (size_t)min(y + r, y_last) * a.yuv.y_linesize + xc);
#pragma unroll
    for (int r = 0; r < kRowUnroll / 2; ++r) {
      const size_t crow = (size_t)min((y >> 1) + r, y_last >> 1);
      const uint32_t u = walk_alone_load2(fr.u + crow * a.yuv.u_linesize + (xc >> 1));
      const uint32_t v = walk_alone_load2(fr.v + crow * a.yuv.v_linesize + (xc >> 1));
      b.uv[r] = u | (v << 16);
    }
  } else {  // reduce_load_batch<kSrcRgb0>
    static_assert(SRC == kSrcRgb0, "the strip walker reads RGB0 or planar sources");
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.raw[r] = walk_alone_load16(fr.src + (size_t)xc * 4 + (size_t)min(y + r, y_last) * a.linesize);
  }
}

}  // namespace sat
}  // namespace f360

#include "sat_fuse_dev.h"

namespace f360 {
namespace sat {

template <int SRC, int DEPTH, bool FUSE = false>
__global__ __launch_bounds__(FUSE ? 128 * kFuseOwners : 64 * kWalkWaves) void sat_walk_kernel(
    const EncodeArgs a, const WalkBatch wb, const typename WalkFuseArg<FUSE>::type wf) {
  constexpr int OW = FUSE ? kFuseOwners : kWalkWaves;  // strip owners per workgroup
  // one LDS object: the waves' 3 KiB store-staging slices, then the workgroup's ticket
  __shared__ __attribute__((aligned(16))) uint32_t stage[OW * 3 * kStripPx + 4];
  // encode + sample: per strip owner the reduced pixels it owns, the D rows of the current
  // batch, and the mailbox through which the owner hands them to its helper wave
  __shared__ __attribute__((aligned(16))) uint32_t fuse_lds[FUSE ? OW * kFuseWaveDwords : 4];
  __shared__ uint32_t fuse_box[FUSE ? OW * kRowUnroll : 4];
  const int lane = threadIdx.x & 63;
  // encode + sample: waves OW.. are helpers -- helper k turns the D rows of strip owner
  // k into reduced pixels while the owner walks on (its SIMD has issue slots to spare: the
  // owner alone uses a quarter of them)
  const bool helper = FUSE && (int)(threadIdx.x >> 6) >= OW;
  const int wave = (threadIdx.x >> 6) & (OW - 1);
  uint32_t *wg_ticket = stage + OW * 3 * kStripPx;
  if (threadIdx.x == 0)
    *wg_ticket = __hip_atomic_fetch_add(&a.walk->ticket, 1u, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
  if (FUSE && threadIdx.x < OW * kRowUnroll) fuse_box[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long serial = a.walk->serial;  // written by the previous launch
  const int unit =
      __builtin_amdgcn_readfirstlane((int)(*wg_ticket * (uint32_t)OW) + wave);
  if constexpr (FUSE) {
    if (helper) {
      if (unit < a.walk_units)
        walk_fuse_helper<SRC == kSrcRgb0>(a, wf, unit, lane, fuse_lds + wave * kFuseWaveDwords,
                         fuse_box + wave * kRowUnroll);
      return;
    }
  }
  if (unit < a.walk_units) {
    const int f = unit / a.nstrips;
    const int strip = unit - f * a.nstrips;
    EncodeFrame fr;
    fr.src = wb.src[f];
    fr.sat = wb.sat[f];
    fr.ws = 0;
    fr.y = fr.src;
    fr.u = wb.u[f];
    fr.v = wb.v[f];
    const int x0 = strip * kStripPx + lane * kLanePx;
    const bool pub = strip + 1 < a.nstrips && unit != a.walk_mute;
    const bool need = strip > 0 && !(a.ablate & 64);   // timing experiment: nobody waits
    bool alone = false;  // a hand-off wait timed out: the rest of the strip without the chain
    // no table wanted (f360_satdec_foveate_rect_frames: reduced frames only), or the timing
    // experiment of the same effect
    const bool no_stores = (a.ablate & 128) || fr.sat == nullptr;
    const unsigned long long tag = serial & kWalkTagMask;
    const int nb = a.walk_nbatches;
    // my granules; the left neighbour's are one unit earlier (strip 0 polls its own: ignored)
    unsigned long long *out =
        a.walk_chain + (size_t)unit * nb * kWalkLanes + min(lane, kWalkLanes - 1);
    const unsigned long long *in = need ? out - (size_t)nb * kWalkLanes : out;
    const uint32_t mine =
        (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * 3 * kStripPx * 4;
    const int row_dwords = a.width * 3;
    const int base = strip * kStripPx * 3;
    const int y_last = a.height - 1;

    uint32_t acc[12];  // the table row above, for this lane's 4 pixels (0 above the frame)
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0;
    uint32_t slow_polls = 0, spun = 0;  // hand-off waits that took the slow path, their polls
    // encode + sample: the table row at the last snapshot; D rows and their mailbox (LDS)
    // (planar sources convert in registers and have none to spare: their snapshot lives in LDS,
    // in place of the source-pixel rows, which they do not post)
    constexpr bool kPix = FUSE && SRC == kSrcRgb0;
    constexpr bool kLdsSnap = FUSE && SRC != kSrcRgb0;
    uint32_t snap[kLdsSnap ? 1 : 12];
    uint32_t box_spins = 0;  // polls spent waiting for the helper to hand a D-row slot back
    const uint32_t dbase =
        (uint32_t)reinterpret_cast<uintptr_t>(fuse_lds + wave * kFuseWaveDwords + kFuseEntries);
    const uint32_t mbox = (uint32_t)reinterpret_cast<uintptr_t>(fuse_box + wave * kRowUnroll);
    const uint32_t *plan = nullptr;
    if constexpr (FUSE) {
      if constexpr (kLdsSnap) {
        const uint32_t sa = dbase + (uint32_t)(kRowUnroll * 3) * (kStripPx * 4) + lane * 48;
        lds_write_b128(sa, u32x4{0, 0, 0, 0});
        lds_write_b128(sa + 16, u32x4{0, 0, 0, 0});
        lds_write_b128(sa + 32, u32x4{0, 0, 0, 0});
      } else {
#pragma unroll
        for (int e = 0; e < 12; ++e) snap[e] = 0;
      }
      plan = wf.rowplan + (size_t)f * wf.plan_stride;
    }
    const unsigned long long t_start = (a.ablate & 256) ? __builtin_amdgcn_s_memrealtime() : 0;
    const unsigned long long c_start = (a.ablate & 256) ? __builtin_amdgcn_s_memtime() : 0;

    // --- a batch in three steps.  scan: the strip's own sums of the 8 rows -- lane totals, wave
    // scans (kept for the row step), row totals in lanes 3r + c of the returned register
    auto scan_batch = [&](const RowBatch<SRC> &raw, uint32_t (&inc_rg)[kRowUnroll],
                          uint32_t (&inc_b)[kRowUnroll]) -> uint32_t {
      uint32_t tot = 0;  // lane 3r + c: this strip's sum of row y + r, channel c
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r) {
        uint32_t v[4];
        batch_pixels<SRC>(a, raw, r, v);
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
      return tot;
    };
    // publish: running row prefixes out to the right (lin: those that came in from the left)
    auto publish = [&](int t, uint32_t lin, uint32_t tot) {
      if (pub && lane < kWalkLanes)
        walk_store_granule(out + (size_t)t * kWalkLanes,
                           (tag << 24) | (unsigned long long)((lin + tot) & 0xffffffu));
    };
    // write: the table rows (sat_write_kernel's row step with the scans already done)
    auto write_batch = [&](const RowBatch<SRC> &raw, const uint32_t (&inc_rg)[kRowUnroll],
                           const uint32_t (&inc_b)[kRowUnroll], uint32_t lin, int t,
                           const u32x8 &pw) {
      const int y = t * kRowUnroll;
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r) {
        if (y + r > y_last) break;
        // encode + sample: a first look at this row's D-row slot, issued now and read after the
        // staging round trip below has waited for it anyway
        uint32_t slot_word = 0;
        if constexpr (FUSE) {
          if (pw[r] & kFuseEmit)
            asm volatile("ds_read_b32 %0, %1" : "=v"(slot_word) : "v"(mbox + r * 4) : "memory");
        }
        uint32_t c[12], px[4];
        batch_pixels<SRC>(a, raw, r, px);
        unpack_px4(make_uint4(px[0], px[1], px[2], px[3]), c);
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
        uint32_t *row = fr.sat + (size_t)(y + r) * a.width * 3;
        lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
        lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
        lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
        u32x4 q[3];
        lds_read3_b128(mine + lane * 16, q[0], q[1], q[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int off = k * 256 + lane * 4;
          if (base + off < row_dwords && !no_stores)  // width % 4 == 0 -> whole 16 B in range
            global_store_b128_uncounted_nt(row + base + off, q[k]);
        }
        if constexpr (FUSE) {
          const uint32_t pr = pw[r];  // wave-uniform (scalar registers)
          if (pr & kFuseEmit) {
            // (a reduced row one table row high: where its boxes are also one column wide --
            // the fovea -- a reduced pixel is the source pixel itself, so the source row goes
            // along and saves the helper five LDS reads and three subtractions per pixel)
            const bool one_row = ((pr >> 16) & 0x3ffu) == 1u;
            // D = this row - snapshot into slot r, once the helper is done with the slot's
            // previous row (a batch ago), then the plan word into the mailbox: the payload is
            // in LDS before its flag (one wave's LDS operations execute in order)
            asm volatile("" : "+v"(slot_word));  // (returned before lds_read3_b128's wait)
            while (slot_word != 0) {
              ++box_spins;
              __builtin_amdgcn_s_sleep(1);
              slot_word = lds_read_b32(mbox + r * 4);
            }
            const uint32_t da = dbase + (uint32_t)r * (3 * kStripPx * 4) + lane * 48;
            uint32_t sv[12];
            if constexpr (kLdsSnap) {
              u32x4 s0, s1, s2;
              asm volatile(
                  "ds_read_b128 %0, %3\n\t"
                  "ds_read_b128 %1, %3 offset:16\n\t"
                  "ds_read_b128 %2, %3 offset:32\n\t"
                  "s_waitcnt lgkmcnt(0)"
                  : "=&v"(s0), "=&v"(s1), "=&v"(s2)
                  : "v"(dbase + (uint32_t)(kRowUnroll * 3) * (kStripPx * 4) + lane * 48)
                  : "memory");
              sv[0] = s0.x, sv[1] = s0.y, sv[2] = s0.z, sv[3] = s0.w;
              sv[4] = s1.x, sv[5] = s1.y, sv[6] = s1.z, sv[7] = s1.w;
              sv[8] = s2.x, sv[9] = s2.y, sv[10] = s2.z, sv[11] = s2.w;
            } else {
#pragma unroll
              for (int e = 0; e < 12; ++e) sv[e] = snap[e];
            }
            lds_write_b128(da, u32x4{acc[0] - sv[0], acc[1] - sv[1], acc[2] - sv[2],
                                     acc[3] - sv[3]});
            lds_write_b128(da + 16, u32x4{acc[4] - sv[4], acc[5] - sv[5], acc[6] - sv[6],
                                          acc[7] - sv[7]});
            lds_write_b128(da + 32, u32x4{acc[8] - sv[8], acc[9] - sv[9],
                                          acc[10] - sv[10], acc[11] - sv[11]});
            if constexpr (kPix)
              if (one_row)
                lds_write_b128(dbase + (uint32_t)(kRowUnroll * 3 + r) * (kStripPx * 4) + lane * 16,
                               u32x4{px[0], px[1], px[2], px[3]});
            // (no wait between payload and flag: LDS executes one wave's operations in order)
            asm volatile("ds_write_b32 %0, %1" ::"v"(mbox + r * 4), "v"(pr) : "memory");
          }
          if (pr & kFuseSnap) {
            if constexpr (kLdsSnap) {
              const uint32_t sa = dbase + (uint32_t)(kRowUnroll * 3) * (kStripPx * 4) + lane * 48;
              lds_write_b128(sa, u32x4{acc[0], acc[1], acc[2], acc[3]});
              lds_write_b128(sa + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
              lds_write_b128(sa + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
            } else {
#pragma unroll
              for (int e = 0; e < 12; ++e) snap[e] = acc[e];
            }
          }
        }
      }
    };
    // The sums of rows [8t, 8t + 8) over all strips to the left, recomputed from the source with
    // the very scan the neighbours use (lane 3r + c, like a granule's payload): what a strip
    // whose hand-off did not come takes instead.  Slow (strip s reads s strips) and exact.
    auto left_of_me = [&](int t) -> uint32_t {
      uint32_t lin = 0;
      for (int s = 0; s < strip; ++s) {
        RowBatch<SRC> left;
        uint32_t sc_rg[kRowUnroll], sc_b[kRowUnroll];
        walk_alone_load_batch<SRC>(a, fr, left, t * kRowUnroll, s * kStripPx + lane * kLanePx,
                                   y_last);
        lin += scan_batch(left, sc_rg, sc_b);
      }
      return lin & 0xffffffu;
    };
    // One batch.  A wait for the left neighbour that runs into its bound (it cannot, short of a
    // hung or descheduled neighbour) takes the strip off the hand-off chain for the rest of its
    // rows: nothing is guessed, what is stored and what is published to the right are the values
    // the chain would have delivered.
    auto walk_batch = [&](const RowBatch<SRC> &raw, unsigned long long g, int t) {
      uint32_t inc_rg[kRowUnroll], inc_b[kRowUnroll];
      // encode + sample: the plan words of the batch's rows, a scalar load hidden from the
      // compiler like every other memory operation of this loop; waited for before the rows
      u32x8 pw = {0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (FUSE) {
        const uint32_t *pp = plan + (size_t)t * kRowUnroll;
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(pw) : "s"(pp));
      }
      const uint32_t tot = scan_batch(raw, inc_rg, inc_b);
      uint32_t lin = 0;
      if (need) {
        if (!alone && !__all((g >> 24) == tag || lane >= kWalkLanes)) {
          ++slow_polls;
          g = walk_repoll(in + (size_t)t * kWalkLanes, tag, lane, a.walk_spin, spun);
          if (!__all((g >> 24) == tag || lane >= kWalkLanes)) {
            alone = true;
            if (lane == 0)
              __hip_atomic_fetch_add(a.walk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        lin = alone ? left_of_me(t) : (uint32_t)g & 0xffffffu;
      }
      // out to the right BEFORE the heavy part: the chain advances at hand-off latency
      publish(t, lin, tot);
      if constexpr (FUSE) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pw));
      write_batch(raw, inc_rg, inc_b, lin, t, pw);
    };

    // DEPTH batches of 8 rows rotate through static buffers, DEPTH - 1 of them in flight.  The
    // poll of batch t is issued BEFORE the pixel loads of batch t + DEPTH - 1, so waiting for
    // it leaves those in flight; every load is unconditional (rows clamped to the last row).
    // (Consuming the hand-off one iteration AFTER producing it -- scan and publish batch t + 1,
    // then write the rows of batch t -- was built and measured: 82.3 against 80.3 us per frame,
    // the waiting path still taken in 60 % of the batches.  Slack does not help: a strip cannot
    // pass its left neighbour, so the gap between two neighbours is a random walk with a
    // reflecting barrier, and moving the barrier by one batch moves the walk, not its spread.)
    RowBatch<SRC> buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
      walk_load_batch<SRC>(a, fr, buf[d], d * kRowUnroll, x0, y_last);
    for (int t0 = 0; t0 < nb; t0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int t = t0 + d;
        const unsigned long long g = __hip_atomic_load(
            in + (size_t)min(t, nb - 1) * kWalkLanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        walk_load_batch<SRC>(a, fr, buf[(d + DEPTH - 1) % DEPTH], (t + DEPTH - 1) * kRowUnroll,
                             x0, y_last);
        if (t < nb) walk_batch(buf[d], g, t);
      }
    }
    if ((a.ablate & 256) && lane == 0) {
      // (two 16-byte stores: 8-byte stores are reserved for the hand-off granules, whose sc1
      // bit the ISA guard checks)
      ulonglong2 *st = reinterpret_cast<ulonglong2 *>(a.walk_stats + (size_t)unit * 8);
      st[0] = make_ulonglong2(t_start, __builtin_amdgcn_s_memrealtime());
      // (slow waits in 16 bits, above them the shader-clock cycles of the walk: boxes differ)
      st[1] = make_ulonglong2(
          slow_polls | ((__builtin_amdgcn_s_memtime() - c_start) << 16),
          spun | ((unsigned long long)box_spins << 32));
    }
  }
  // retire: the last wave of the launch re-arms the state for the next one
  if (lane == 0) {
    const uint32_t waves = gridDim.x * OW;
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

// ---- host side (sat_walk.hip) ----------------------------------------------------------------
// What a call on the read-once encoder works out before its launches (walk_prepare).
struct WalkSetup {
  EncodeArgs a;  // everything but the launch's own walk_units
  int nstrips, nb, per_launch, yuv_src;
  // encode + sample (sat_fuse.hip): the plan / side buffers' carving
  int plan_stride, pmax;
  size_t plan_words, side_stride;
};
bool walk_wanted(const f360_ctx *ctx, int count, int width);
int walk_frames_per_launch(const f360_ctx *ctx, int count, int width);
int walk_prepare(f360_ctx *ctx, int count, const f360::YuvPlanes *yuvs, int width, int height,
                 int linesize, const f360::SatFuse *fuse, WalkSetup &ws);
void walk_fill_batch(WalkBatch &wb, int k0, int n, uint32_t *const *sats,
                     const uint8_t *const *srcs, const f360::YuvPlanes *yuvs);
int sat_encode_walk(f360_ctx *ctx, int count, uint32_t *const *sats, const uint8_t *const *srcs,
                    const f360::YuvPlanes *yuvs, int width, int height, int linesize, bool prof);

}  // namespace sat
}  // namespace f360
