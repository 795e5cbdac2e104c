// sat_reduce.h -- K1 of the three-kernel SAT encoder: column / row / tile sums of a frame
// (sat_three.hip launches it).
#pragma once

#include "sat_common.h"
#include "sat_walk.h"

namespace f360 {
namespace sat {

// ---- K1: column / row / tile sums ------------------------------------------
// One wave per (strip, super-band), no workgroup synchronisation (a
// __syncthreads() would drain the loads that are kept in flight).  A
// super-band is a contiguous run of rows; they are loaded in batches of
// kRowUnroll, double-buffered, without any per-row branch: addresses are
// clamped into the frame and rows past the end are masked to zero (the masks
// are wave-uniform, i.e. scalar registers).  Lanes past the right edge exist
// only in the last strip, where nothing to their right consumes their sums, so
// they need no mask at all.  Red and blue travel together as two 16-bit fields
// (x & 0x00ff00ff), green as x & 0xff00: within one band (<= 64 rows) a lane's
// column sums and a strip's row sums cannot carry from one field into the next.
struct ReduceState {
  uint32_t col[12];        // column sums of the rows since the super-band began
  uint32_t crb[4], cg[4];  // current band: packed R|B<<16 and G<<8 column sums
  uint32_t tile[3];        // lane 63: strip sums of the rows of the current band
};


// sums rows [y, y + kRowUnroll) that lie below y_stop; the strip's row sums go to the wave's
// LDS slice `rows_lds` (byte address of the band's first row), see reduce_store_rowsums
template <int SRC>
__device__ __forceinline__ void reduce_rows(const EncodeArgs &a, ReduceState &st,
                                            const RowBatch<SRC> &raw, int y, int y_stop,
                                            int row_in_band, uint32_t rows_lds, int lane) {
#pragma unroll
  for (int r = 0; r < kRowUnroll; r += 2) {
    uint32_t rb[2], g[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t live = (y + r + h < y_stop) ? 0xffffffffu : 0u;  // scalar
      const uint32_t mrb = 0x00ff00ffu & live, mg = 0x0000ff00u & live;
      uint32_t v[4];
      batch_pixels<SRC>(a, raw, r + h, v);
      rb[h] = 0;
      g[h] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t m = v[k] & mrb;
        const uint32_t gg = v[k] & mg;
        st.crb[k] += m;
        st.cg[k] += gg;
        rb[h] += m;
        g[h] += gg;
      }
    }
    // strip sums end up in lane 63; the two rows' greens share one scan
    uint32_t t0 = rb[0], t1 = rb[1], tg = (g[0] >> 8) | (g[1] << 8);
    if (!(a.ablate & 1)) {
      t0 = wave_scan_incl(t0);
      t1 = wave_scan_incl(t1);
      tg = wave_scan_incl(tg);
    }
    if (lane == 63 && y + r < y_stop && !(a.ablate & 2)) {
      // rows past y_stop were masked to zero above, so both rows can be written
      const uint32_t at = rows_lds + (uint32_t)(row_in_band + r) * 12;
      lds_write_b64(at, t0 & 0xffffu, tg & 0xffffu);
      lds_write_b64(at + 8, t0 >> 16, t1 & 0xffffu);
      lds_write_b64(at + 16, tg >> 16, t1 >> 16);
      st.tile[0] += (t0 & 0xffffu) + (t1 & 0xffffu);
      st.tile[1] += (tg & 0xffffu) + (tg >> 16);
      st.tile[2] += (t0 >> 16) + (t1 >> 16);
    }
  }
}

// One band's row sums, LDS -> rowsum[strip][y][3], 256 contiguous bytes per store instruction.
// They are kept out of the row loop on purpose: a global store between the loads and their use
// makes the compiler wait for (almost) everything in flight, because loads and stores share
// vmcnt on gfx9-class hardware and complete out of order relative to each other.
__device__ __forceinline__ void reduce_store_rowsums(const EncodeArgs &a, const EncodeFrame &fr,
                                                     uint32_t rows_lds, int strip, int band_y0,
                                                     int y_stop, int lane) {
  const int n = min(a.band_rows, y_stop - band_y0) * 3;
  uint32_t *dst = a.rowsum + fr.ws + ((size_t)strip * a.height + band_y0) * 3;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int i = q * 64 + lane;
    if (i < n) dst[i] = lds_read_b32(rows_lds + (uint32_t)i * 4);
  }
}

// fold the current band's packed sums into the 32-bit running sums
__device__ __forceinline__ void reduce_flush_band(ReduceState &st) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st.col[3 * k + 0] += st.crb[k] & 0xffffu;
    st.col[3 * k + 1] += st.cg[k] >> 8;
    st.col[3 * k + 2] += st.crb[k] >> 16;
    st.crb[k] = 0;
    st.cg[k] = 0;
  }
}

// (Round 5 tried an EMIT form in which the reducer -- bound by its reads, with the store path to
// spare -- also copied the fovea's reduced pixels, which are source pixels: 30 -> 75 us for
// 20 MB of three-byte stores, eight sub-dword store instructions per row and wave.  Removed;
// profiles/round5_band_one_pass.txt.)
template <int SRC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sat_reduce_kernel(
    const EncodeArgs a, const EncodeBatch eb) {
  __shared__ uint32_t rowsum_stage[kWavesPerBlock * 64 * 3];  // one band of row sums per wave
  if ((int)blockIdx.x >= a.reduce_blocks) {  // only when a.has_maps: see fov_maps.h
    __shared__ uint8_t fov_flags[f360::kFovLdsEntries];
    __shared__ int16_t fov_ranks[f360::kFovLdsEntries];
    __shared__ int fov_part[4];
    f360::fov_maps_axis(a.maps, (int)blockIdx.x - a.reduce_blocks, fov_flags, fov_ranks, fov_part);
    return;
  }
  const int lane = threadIdx.x & 63;
  F360_ENCODE_FRAME(fr, a, eb)
  const uint32_t rows_lds = (uint32_t)reinterpret_cast<uintptr_t>(rowsum_stage) +
                            (uint32_t)(threadIdx.x >> 6) * 64 * 3 * 4;
  // 1-D grid over the tiles in row-major order, 4 consecutive tiles per workgroup: every
  // workgroup is full, so the round-robin of workgroups over the 8 XCDs stays balanced (a 2-D
  // grid with 8 workgroup columns pins each strip group to one XCD, the ragged last one too)
  const int tile = __builtin_amdgcn_readfirstlane(
      (int)blockIdx.x * kWavesPerBlock + (int)(threadIdx.x >> 6));
  if (tile >= a.nstrips * a.nsb) return;
  const int sb = tile / a.nstrips;
  const int strip = tile - sb * a.nstrips;
  const int x0 = strip * kStripPx + lane * kLanePx;

  ReduceState st;
#pragma unroll
  for (int e = 0; e < 12; ++e) st.col[e] = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) st.crb[k] = st.cg[k] = 0;
  st.tile[0] = st.tile[1] = st.tile[2] = 0;

  const int band0 = sb * a.sb_bands;
  const int band_end = min(band0 + a.sb_bands, a.nbands);
  const int y_stop = min(band_end * a.band_rows, a.height);

  // The super-band is a run of batches of kRowUnroll rows with kReduceDepth of them in flight.
  // Loads are unconditional (clamped to the wave's last row): a load inside a branch makes the
  // compiler wait for ALL outstanding loads at the next use (it cannot count what is in flight
  // on both paths), which silently turns any depth into one.  Buffers rotate with static
  // indices; band boundaries fall on batch boundaries (band_rows is 16, 32 or 64).
  RowBatch<SRC> buf[kReduceDepth];
  const int y_first = band0 * a.band_rows;
  const int y_last = y_stop - 1;
  const int bpb = a.band_rows / kRowUnroll;  // batches per band
  const int nbatch = (band_end - band0) * bpb;
#pragma unroll
  for (int d = 0; d < kReduceDepth - 1; ++d)
    reduce_load_batch<SRC>(a, fr, buf[d], y_first + d * kRowUnroll, x0, y_last);
  for (int t0 = 0; t0 < nbatch; t0 += kReduceDepth) {
#pragma unroll
    for (int d = 0; d < kReduceDepth; ++d) {
      const int t = t0 + d;
      reduce_load_batch<SRC>(a, fr, buf[(d + kReduceDepth - 1) % kReduceDepth],
                             y_first + (t + kReduceDepth - 1) * kRowUnroll, x0, y_last);
      if (t < nbatch) {
        const int band = band0 + t / bpb;
        const int in_band = t % bpb;
        if (in_band == 0 && a.sb_bands != 1)
          store12(a.lp + fr.ws + (size_t)band * a.wp3 + (size_t)x0 * 3, st.col);
        reduce_rows<SRC>(a, st, buf[d], y_first + t * kRowUnroll, y_stop,
                         in_band * kRowUnroll, rows_lds, lane);
        if (in_band == bpb - 1) {
          if (!(a.ablate & 2))
            reduce_store_rowsums(a, fr, rows_lds, strip, band * a.band_rows, y_stop, lane);
          reduce_flush_band(st);
          if (lane == 63) {
            uint32_t *tt = a.tiletotal + fr.ws + ((size_t)strip * a.nbands + band) * 3;
            tt[0] = st.tile[0];
            tt[1] = st.tile[1];
            tt[2] = st.tile[2];
          }
          st.tile[0] = st.tile[1] = st.tile[2] = 0;
        }
      }
    }
  }
  store12(a.sbtotal + fr.ws + (size_t)sb * a.wp3 + (size_t)x0 * 3, st.col);
}


}  // namespace sat
}  // namespace f360
