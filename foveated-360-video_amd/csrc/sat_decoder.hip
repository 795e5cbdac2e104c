// sat_decoder.hip -- SATDecoder device side: log-rectilinear SAT sampler
// (forward warp), bilinear un-warp, identity decode.
//
// Replaces SATDecoder::{InitializeGrid, SampleFrameRectGPU,
// InterpolateFrameRectGPU, DecodeFrameGPU} (src/sat_decoder.cc:139-210,301-348,
// 887-928) and their kernels (src/sat_decoder_sample_rect_kernel.cl:138-295,
// src/sat_decoder_interpolate_kernel.cl, src/sat_decoder_decode_kernel.cl).
//
// The reference's 2-D grid and all of its per-pixel exp/pow/log calls are
// separable (x depends on the column only, y on the row only); they live in
// 1-D tables built on the host (host_tables.cpp) so the kernels below are pure
// integer / IEEE-float gather kernels.
#include <algorithm>
#include <numeric>
#include <cmath>
#include <cstring>

#include "f360_internal.h"
#include "fov_maps.h"
#include "host_tables.h"

namespace {

// Streaming outputs (written once, consumed later by another kernel or a copy engine) are stored
// non-temporally: see global_store_b128_uncounted_nt in sat_common.h.
#ifdef F360_NO_NT_STORES
#define F360_STREAM_STORE(ptr, value) (*(ptr) = (value))
#else
#define F360_STREAM_STORE(ptr, value) __builtin_nontemporal_store((value), (ptr))
#endif

using f360::AxisBox;
using f360::FovMaps;
using f360::sample_axis;  // one axis of sample_rect_kernel's box rule (fov_maps.h)
using f360::udiv3_exact;  // the three channel quotients of one box (fov_maps.h)

__device__ __forceinline__ uint3 load_sat3(const uint32_t *sat, size_t texel) {
  const uint32_t *p = sat + texel * 3;
  return make_uint3(p[0], p[1], p[2]);
}

// q = n / d for uint32, exact.  The common case (n < 2^23: a box of 8-bit
// samples) goes through one correctly rounded float division, which is exact
// there; anything larger (a wrapped table read with a degenerate box) takes
// the integer path.
__device__ __forceinline__ uint32_t udiv_exact(uint32_t n, uint32_t d) {
  if (n < (1u << 23) && d < (1u << 23)) {
    return (uint32_t)((float)n / (float)d);
  }
  return n / d;
}

// 12-byte texel at a 32-bit BYTE offset from the table base (the table is < 4 GiB on
// this path): scalar base + 32-bit vector offset addressing, no 64-bit arithmetic
__device__ __forceinline__ uint3 load_sat3_at(const uint32_t *sat, uint32_t byte_off) {
  const uint32_t *p = reinterpret_cast<const uint32_t *>(
      reinterpret_cast<const char *>(sat) + byte_off);
  return make_uint3(p[0], p[1], p[2]);
}

// Variant 0: one thread per reduced pixel.
__global__ __launch_bounds__(256) void sample_rect_kernel(
    uint8_t *__restrict__ dst, int out_w, int out_h, int out_stride_px,
    const uint32_t *__restrict__ sat, int src_w, int src_h,
    const int16_t *__restrict__ gx, const int16_t *__restrict__ gy, int cxp,
    int cyp) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= out_w || j >= out_h) return;
  const AxisBox bx = sample_axis(cxp, gx[i + 1], gx[i], src_w, true);
  const AxisBox by = sample_axis(cyp, gy[j + 1], gy[j], src_h, false);
  if (!(bx.ok && by.ok)) return;
  const uint3 br = load_sat3(sat, (size_t)by.hi * src_w + bx.hi);
  const uint3 tr = load_sat3(sat, (size_t)by.lo * src_w + bx.hi);
  const uint3 tl = load_sat3(sat, (size_t)by.lo * src_w + bx.lo);
  const uint3 bl = load_sat3(sat, (size_t)by.hi * src_w + bx.lo);
  const uint32_t area = (uint32_t)((bx.hi - bx.lo) * (by.hi - by.lo));
  uint8_t *o = dst + ((size_t)j * out_stride_px + i) * 4;
  o[0] = (uint8_t)udiv_exact(br.x - tr.x + tl.x - bl.x, area);
  o[1] = (uint8_t)udiv_exact(br.y - tr.y + tl.y - bl.y, area);
  o[2] = (uint8_t)udiv_exact(br.z - tr.z + tl.z - bl.z, area);
}

// Variant 1 ("column walker"): a wave owns 63 adjacent reduced columns (lane 0
// is a halo lane for the column to the left) and walks down `rows` reduced rows.
// Adjacent boxes share corners -- the lower-left corner of pixel i is the
// lower-right corner of pixel i-1, and the top corners of row j are the bottom
// corners of row j-1 -- so each output pixel costs ONE 12-byte gather: the left
// corner arrives by DPP from the neighbouring lane and the top corners stay in
// registers.  Where the reference's wrap / clamp rules break that sharing
// (frame seam, first row of a run) the corner is loaded explicitly, so the
// result is the reference's for every pixel.
__device__ __forceinline__ uint32_t dpp_from_lane_below(uint32_t v) {
  // wave_shr:1 -- lane l receives lane l-1's value, lane 0 keeps its own
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint3 dpp_from_lane_below(const uint3 &v) {
  return make_uint3(dpp_from_lane_below(v.x), dpp_from_lane_below(v.y),
                    dpp_from_lane_below(v.z));
}

constexpr int kWalkCols = 63;
constexpr int kWalkBatch = 8;  // rows whose gathers are issued together

struct SampleArgs {
  uint8_t *dst;
  int out_w, out_h, out_stride_px;
  const uint32_t *sat;
  int src_w, src_h;
  const int16_t *gx, *gy;
  int cxp, cyp;
  // tile streamer: inverse of the x grid (lbx[d - lb_dmin] = first grid index whose offset is
  // >= d), largest corner step (+1, rounded to 4)
  const int *lbx;
  int lb_dmin, lb_n, halo;
  int ablate;                  // timing experiments only
};

// Non-temporal: the reduced frame is read next by a copy engine or another kernel, and dirty lines
// left in L2 / Infinity Cache are written back at the expense of the next frame's reducer.
__device__ __forceinline__ void store_rgb(uint8_t *o, uint32_t r, uint32_t g, uint32_t b) {
  __builtin_nontemporal_store((uint16_t)((r & 0xffu) | ((g & 0xffu) << 8)),
                              reinterpret_cast<uint16_t *>(o));
  __builtin_nontemporal_store((uint8_t)b, o + 2);
}

// One wave: reduced columns [c0, c0 + 63) clipped to [col_begin, col_end), rows [j0, j1).
__device__ __forceinline__ void walk_body(const SampleArgs &a, int c0, int col_begin,
                                          int col_end, int j0, int j1) {
  const int lane = threadIdx.x & 63;
  const int i = c0 - 1 + lane;
  const int ic = min(max(i, 0), a.out_w - 1);
  const AxisBox bx = sample_axis(a.cxp, a.gx[ic + 1], a.gx[ic], a.src_w, true);
  const bool writes = lane > 0 && i >= col_begin && i < col_end && bx.ok;
  // lane 0 never writes, so it never needs a left corner of its own
  const bool left_shared =
      lane == 0 || dpp_from_lane_below((uint32_t)bx.hi) == (uint32_t)bx.lo;
  const uint32_t dxw = (uint32_t)(bx.hi - bx.lo);
  const uint32_t *sat = a.sat;
  const uint32_t row_bytes = (uint32_t)a.src_w * 12u;
  const uint32_t x_hi = (uint32_t)bx.hi * 12u, x_lo = (uint32_t)bx.lo * 12u;
  uint8_t *out = a.dst + (size_t)i * 4;

  int prev_hi = -1;
  uint3 p_br = make_uint3(0, 0, 0), p_bl = make_uint3(0, 0, 0);
  for (int jb = j0; jb < j1; jb += kWalkBatch) {
    // wave-uniform row boxes of the batch, then all of its gathers at once
    AxisBox by[kWalkBatch];
    bool any = false;
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r) {
      const int j = min(jb + r, a.out_h - 1);
      by[r] = sample_axis(a.cyp, a.gy[j + 1], a.gy[j], a.src_h, false);
      by[r].ok = by[r].ok && (jb + r < j1);
      any = any || by[r].ok;
    }
    if (!any) {
      prev_hi = -1;
      continue;
    }
    uint3 brs[kWalkBatch];
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r)  // clamped corners are always in range
      brs[r] = load_sat3_at(sat, (uint32_t)by[r].hi * row_bytes + x_hi);
    // the top row of the batch's first box, when the previous batch does not provide it: issued
    // with the batch's gathers instead of costing a round trip of its own later
    int first = kWalkBatch, top_row = 0;
#pragma unroll
    for (int r = kWalkBatch - 1; r >= 0; --r)
      if (by[r].ok) {
        first = r;
        top_row = by[r].lo;
      }
    const bool have_top = top_row != prev_hi;
    uint3 top = make_uint3(0, 0, 0);
    if (have_top) top = load_sat3_at(sat, (uint32_t)top_row * row_bytes + x_hi);
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r) {
      if (!by[r].ok) {
        prev_hi = -1;
        continue;
      }
      const uint3 br = brs[r];
      uint3 bl = dpp_from_lane_below(br);
      if (!left_shared) bl = load_sat3_at(sat, (uint32_t)by[r].hi * row_bytes + x_lo);
      uint3 tr, tl;
      if (by[r].lo == prev_hi) {
        tr = p_br;
        tl = p_bl;
      } else {
        tr = (have_top && r == first)
                 ? top
                 : load_sat3_at(sat, (uint32_t)by[r].lo * row_bytes + x_hi);
        tl = dpp_from_lane_below(tr);
        if (!left_shared) tl = load_sat3_at(sat, (uint32_t)by[r].lo * row_bytes + x_lo);
      }
      if (writes) {
        const uint3 q = udiv3_exact(make_uint3(br.x - tr.x + tl.x - bl.x, br.y - tr.y + tl.y - bl.y,
                                               br.z - tr.z + tl.z - bl.z),
                                    dxw * (uint32_t)(by[r].hi - by[r].lo));
        store_rgb(out + (size_t)(jb + r) * ((size_t)a.out_stride_px * 4), q.x, q.y, q.z);
      }
      p_br = br;
      p_bl = bl;
      prev_hi = by[r].hi;
    }
  }
}

__global__ __launch_bounds__(256) void sample_rect_walk_kernel(const SampleArgs a, int rows) {
  const int c0 = __builtin_amdgcn_readfirstlane(
      ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * kWalkCols);
  if (c0 >= a.out_w) return;  // whole wave
  const int by = (int)blockIdx.y;
  const int j0 = by * rows;
  walk_body(a, c0, 0, a.out_w, j0, min(j0 + rows, a.out_h));
}

// Several gaze points against ONE table (clients that watch the same video share the
// encode): blockIdx.z selects the client.
constexpr int kMaxBatch = 64;    // frames (or gaze points) one launch can take
constexpr int kMaxClients = 16;  // f360_satdec_sample_rect_batch's documented limit
struct SampleBatch {
  uint8_t *dst[kMaxBatch];
  const uint32_t *sat[kMaxBatch];  // the same table for every client, or one per frame
  int cxp[kMaxBatch], cyp[kMaxBatch];
};

__global__ __launch_bounds__(256) void sample_rect_walk_batch_kernel(SampleArgs a,
                                                                      const SampleBatch b,
                                                                      int rows) {
  const int c0 = __builtin_amdgcn_readfirstlane(
      ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * kWalkCols);
  if (c0 >= a.out_w) return;  // whole wave
  const int z = blockIdx.z;
  a.dst = b.dst[z];
  a.sat = b.sat[z];
  a.cxp = b.cxp[z];
  a.cyp = b.cyp[z];
  const int j0 = (int)blockIdx.y * rows;
  walk_body(a, c0, 0, a.out_w, j0, min(j0 + rows, a.out_h));
}

// Variant 2 ("tile streamer") lives in sample_stream.h.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// Wave-private LDS exchange goes through inline asm: the compiler's memory
// model is per lane and it deletes plain LDS stores that only OTHER lanes read.
// One wave's LDS operations execute in order, so no barrier is needed.
__device__ __forceinline__ void lds_store4(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_load4(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}

#include "sample_stream.h"

template <int NS>
__global__ __launch_bounds__(256) void sample_rect_stream_batch_kernel(SampleArgs a,
                                                                        const SampleBatch b,
                                                                        int rows, int nblocks) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][NS * kTsSlotBytes + kTsStageBytes];
  const int wave = threadIdx.x >> 6;
  const int z = blockIdx.y;
  a.dst = b.dst[z];
  a.sat = b.sat[z];
  a.cxp = b.cxp[z];
  a.cyp = b.cyp[z];
  const int ntiles = (a.src_w + kTsTile - 1) / kTsTile;
  const int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
  if (g >= ntiles * nblocks) return;
  const int blk = g / ntiles;
  tile_stream_body<NS>(a, g - blk * ntiles, blk * rows, rows,
                              (uint32_t)reinterpret_cast<uintptr_t>(&stage[wave][0]),
                              &stage[wave][0]);
}

// Host check of what tile_stream_body assumes for a gaze: per 128-texel tile the largest of the
// candidate ranges fits two passes (128 pixels) and the others together one (64).
bool tile_stream_fits(const f360_sat_decoder *dec, int cxp, int source_width, int target_width) {
  const int ntiles = (source_width + kTsTile - 1) / kTsTile;
  const std::vector<int> &lb = dec->lbx_host;
  auto lb_at = [&](long d) {
    return lb[(size_t)std::min<long>(std::max<long>(d - dec->lb_dmin, 0), (long)lb.size() - 1)];
  };
  for (int t = 0; t < ntiles; ++t) {
    int largest = 0, total = 0;
    for (int q = 0; q < kTsRanges; ++q) {
      const int k = (q == 1 || q == 4) ? 1 : (q == 2 ? -1 : 0);
      long lo_b = (long)t * kTsTile + (long)k * source_width - cxp, hi_b = lo_b + kTsTile;
      if (q >= 3) {
        lo_b = hi_b;
        hi_b = lo_b + (t == ntiles - 1 ? dec->halo : 0);
      }
      const int i_lo = std::max(lb_at(lo_b) - 1, 0);
      const int i_hi = std::max(std::min(lb_at(hi_b) - 1, target_width), i_lo);
      largest = std::max(largest, i_hi - i_lo);
      total += i_hi - i_lo;
    }
    if (largest > kTsTile || total - largest > 64) return false;
  }
  return true;
}
// ... and of the call as a whole: the per-geometry tables (inverse grid, halo) belong to the
// source size the grid was initialised for; any other size takes the walker
bool tile_stream_applies(const f360_sat_decoder *dec, const uint32_t *sat_dev, int source_width,
                         int source_height, int target_linesize, int target_height) {
  return dec->stream_ok && (source_width % 4) == 0 && source_width == dec->sw &&
         source_height == dec->sh &&
         (size_t)source_width * source_height * 12 < ((size_t)1 << 32) &&
         (size_t)target_linesize * target_height < ((size_t)1 << 32) &&
         ((uintptr_t)sat_dev % 16) == 0 && source_height <= 0xffff && target_height < 0xffff;
}

// ---------------------------------------------------------------------------
// Fused foveation (SURVEY.md 8f-1 i): frame -> reduced frame without materialising the table.
// When the gaze is known before the encode (the offline modes of the reference take it from a
// trace, run_satlogrectilinear.cc:926-938), only the table entries at the lattice rows / columns
// that gaze samples are ever read.  foveate_maps_kernel numbers those rows and columns, the table
// writer (sat_three.hip, STORE == 2) emits just those entries into a compact array, and
// sample_compact_kernel forms the box means from it -- the same integers as encode + sample.
// The lattice maps as a kernel of their own (two workgroups, one per axis).  Normally they run as
// two extra workgroups of the reducer's launch instead ("fov.piggyback", sat_three.hip).
__global__ __launch_bounds__(f360::kFovThreads) void foveate_maps_kernel(const FovMaps m) {
  __shared__ uint8_t flags[f360::kFovLdsEntries];
  __shared__ int16_t ranks[f360::kFovLdsEntries];
  __shared__ int part[4];
  f360::fov_maps_axis(m, (int)blockIdx.x, flags, ranks, part);
}

// The walker on the compact corner array: a box is (ih, il) in compact indices instead of
// table coordinates; the sharing tests between neighbouring lanes / rows are the same.
__global__ __launch_bounds__(256) void sample_compact_walk_kernel(
    uint8_t *__restrict__ dst, int out_w, int out_h, int out_stride_px,
    const uint32_t *__restrict__ corners, int corner_stride, const int *__restrict__ ihx,
    const int *__restrict__ ilx, const int *__restrict__ dxw, const int *__restrict__ ihy,
    const int *__restrict__ ily, const int *__restrict__ dyw, int rows) {
  const int lane = threadIdx.x & 63;
  const int c0 = __builtin_amdgcn_readfirstlane(
      ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * kWalkCols);
  if (c0 >= out_w) return;  // whole wave
  const int i = c0 - 1 + lane;
  const int ic = min(max(i, 0), out_w - 1);
  const int hx = ihx[ic], lx = ilx[ic];
  const bool writes = lane > 0 && i < out_w && hx >= 0;
  const uint32_t x_hi = (uint32_t)max(hx, 0) * 12u, x_lo = (uint32_t)max(lx, 0) * 12u;
  const bool left_shared = lane == 0 || dpp_from_lane_below(x_hi) == x_lo;
  const uint32_t dx = (uint32_t)dxw[ic];
  const uint32_t row_bytes = (uint32_t)corner_stride * 12u;
  uint8_t *out = dst + (size_t)i * 4;
  const int j0 = (int)blockIdx.y * rows, j1 = min(j0 + rows, out_h);

  int prev_hi = -1;
  uint3 p_br = make_uint3(0, 0, 0), p_bl = make_uint3(0, 0, 0);
  for (int jb = j0; jb < j1; jb += kWalkBatch) {
    int hy[kWalkBatch], ly[kWalkBatch];
    bool any = false;
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r) {
      const int j = min(jb + r, out_h - 1);
      hy[r] = (jb + r < j1) ? ihy[j] : -1;  // wave-uniform
      ly[r] = ily[j];
      any = any || hy[r] >= 0;
    }
    if (!any) {
      prev_hi = -1;
      continue;
    }
    uint3 brs[kWalkBatch];
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r)
      brs[r] = load_sat3_at(corners, (uint32_t)max(hy[r], 0) * row_bytes + x_hi);
    int first = kWalkBatch, top_row = 0;  // top row of the batch's first box, see walk_body
#pragma unroll
    for (int r = kWalkBatch - 1; r >= 0; --r)
      if (hy[r] >= 0) {
        first = r;
        top_row = ly[r];
      }
    const bool have_top = top_row != prev_hi;
    uint3 top = make_uint3(0, 0, 0);
    if (have_top) top = load_sat3_at(corners, (uint32_t)top_row * row_bytes + x_hi);
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r) {
      if (hy[r] < 0) {
        prev_hi = -1;
        continue;
      }
      const uint3 br = brs[r];
      uint3 bl = dpp_from_lane_below(br);
      if (!left_shared) bl = load_sat3_at(corners, (uint32_t)hy[r] * row_bytes + x_lo);
      uint3 tr, tl;
      if (ly[r] == prev_hi) {
        tr = p_br;
        tl = p_bl;
      } else {
        tr = (have_top && r == first)
                 ? top
                 : load_sat3_at(corners, (uint32_t)ly[r] * row_bytes + x_hi);
        tl = dpp_from_lane_below(tr);
        if (!left_shared) tl = load_sat3_at(corners, (uint32_t)ly[r] * row_bytes + x_lo);
      }
      if (writes) {
        const int j = min(jb + r, out_h - 1);
        const uint3 q = udiv3_exact(make_uint3(br.x - tr.x + tl.x - bl.x, br.y - tr.y + tl.y - bl.y,
                                               br.z - tr.z + tl.z - bl.z),
                                    dx * (uint32_t)dyw[j]);
        store_rgb(out + (size_t)j * ((size_t)out_stride_px * 4), q.x, q.y, q.z);
      }
      p_br = br;
      p_bl = bl;
      prev_hi = hy[r];
    }
  }
}

// ---------------------------------------------------------------------------
// decode_kernel (src/sat_decoder_decode_kernel.cl:1-58): 1x1 boxes.
__global__ __launch_bounds__(256) void decode_kernel(
    uint8_t *__restrict__ dst, int dst_linesize, int dst_bpp,
    const uint32_t *__restrict__ sat, int width, int height) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;
  const size_t sl = (size_t)3 * width;
  uint8_t *o = dst + (size_t)y * dst_linesize + (size_t)x * dst_bpp;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t v = sat[y * sl + 3 * x + c];
    if (x > 0 && y > 0)
      v = v - sat[(y - 1) * sl + 3 * x + c] + sat[(y - 1) * sl + 3 * (x - 1) + c] -
          sat[y * sl + 3 * (x - 1) + c];
    else if (x > 0)
      v -= sat[y * sl + 3 * (x - 1) + c];
    else if (y > 0)
      v -= sat[(y - 1) * sl + 3 * x + c];
    else
      v = sat[c];
    o[c] = (uint8_t)min(v, 255u);
  }
}

// The same for 4-byte pixels and a width that is a multiple of 4, as a row streamer.  A wave
// owns a 256-pixel strip (4 pixels = 48 table bytes per lane) and walks kDecodeRows rows with the
// previous row in registers, so a table row is read once (plus one halo row per run); the left
// neighbours arrive by DPP.  Round 4: what a row needs -- its 3 KiB table segment, the 1 KiB of
// the destination whose 4th bytes must survive, the texel left of the strip -- goes memory ->
// LDS directly (global_load_lds: contiguous 1 KiB requests instead of three 16-byte loads per
// lane at a 48-byte stride, no staging registers) into a ring of kDecodeSlots slots, three rows
// ahead of the row being decoded; the wait is a counted vmcnt and the stores are issued from
// inline asm (sample_stream.h has the reasons).  Texels outside the frame count as zero, which
// turns the kernel's four cases (:20-57) into one expression in modular u32 arithmetic.  The
// fourth byte of every pixel is preserved by a 16-byte read-modify-write owned by one lane.
// (runs of 16 / 64 / 128 rows and rings of 3 / 6 slots measure the same 110-114 us at 8K: the
// kernel moves 354 MB of table + 118 MB of old pixels + 118 MB of new ones at the 5.3 TB/s this
// device sustains on mixed traffic, profiles/round4_streaming_siblings.txt)
constexpr int kDecodeRows = 32;
constexpr int kDecodeSlots = 4;
constexpr int kDecodeSlotBytes = 3072 + 1024 + 16;  // table segment | old pixels | left texel
constexpr int kDecodeLoadsPerRow = 5;

struct DecodeRow {
  uint32_t t[12];  // 4 texels x 3 channels
  uint32_t l[3];   // the texel left of the first one
};

__global__ __launch_bounds__(256) void decode_strip_kernel(
    uint8_t *__restrict__ dst, int dst_linesize, const uint32_t *__restrict__ sat, int width,
    int height, int strips) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[4][kDecodeSlots * kDecodeSlotBytes];
  const int wslot = threadIdx.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4) + wslot);
  const int lane = threadIdx.x & 63;
  const int strip = wave % strips;
  const int y0 = (wave / strips) * kDecodeRows;
  if (y0 >= height) return;
  const int x_own = strip * 256 + lane * 4;
  const bool writes = x_own < width;
  const bool has_left_strip = strip > 0, has_up = y0 > 0;
  const int n = min(kDecodeRows, height - y0) + 1;  // streamed rows: the row above, then mine
  // per-lane source offsets inside a table row / a destination row; chunks past the row's end
  // (the last strip of a width that is not a multiple of 256) re-read its last chunk
  const uint32_t row_bytes = (uint32_t)width * 12u;
  const uint32_t seg = (uint32_t)strip * 3072u;
  const uint32_t last_chunk = row_bytes - 16u;
  uint32_t tab_off[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) tab_off[k] = min(seg + (uint32_t)(k * 64 + lane) * 16u, last_chunk);
  const uint32_t old_off = (uint32_t)min(x_own, width - 4) * 4u;
  const uint32_t left_off = has_left_strip ? seg - 12u + (uint32_t)min(lane, 2) * 4u : 0u;
  const char *tab = reinterpret_cast<const char *>(sat);
  uint8_t *my = &ring[wslot][0];
  const uint32_t my_lds = (uint32_t)reinterpret_cast<uintptr_t>(my);

  auto issue = [&](int slot, int i) {  // streamed row i = frame row y0 - 1 + i (clamped)
    const int y = min(max(y0 - 1 + i, 0), height - 1);
    const char *row = tab + (size_t)y * row_bytes;
    uint8_t *d = my + slot * kDecodeSlotBytes;
    __builtin_amdgcn_global_load_lds((ts_gptr)(row + tab_off[0]), (ts_lptr)d, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((ts_gptr)(row + tab_off[1]), (ts_lptr)(d + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((ts_gptr)(row + tab_off[2]), (ts_lptr)(d + 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((ts_gptr)(dst + (size_t)y * dst_linesize + old_off),
                                     (ts_lptr)(d + 3072), 16, 0, 0);
    if (lane < 3)
      __builtin_amdgcn_global_load_lds((ts_gptr)(row + left_off), (ts_lptr)(d + 4096), 4, 0, 0);
  };
  constexpr int D = kDecodeSlots - 1;
#pragma unroll
  for (int k = 0; k < D; ++k) issue(k, min(k, n - 1));
  DecodeRow up;
  int slot = 0, slot_pf = D;
  for (int i = 0; i < n; ++i) {
    issue(slot_pf, min(i + D, n - 1));  // unconditional: every row is exactly 5 operations
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * kDecodeLoadsPerRow) : "memory");
    const uint32_t base = my_lds + (uint32_t)slot * kDecodeSlotBytes;
    slot = slot + 1 == kDecodeSlots ? 0 : slot + 1;
    slot_pf = slot_pf + 1 == kDecodeSlots ? 0 : slot_pf + 1;
    u32x4_t q0, q1, q2, old;
    u32x2_t l01;
    uint32_t l2;
    asm volatile(
        "ds_read_b128 %0, %6\n\t"
        "ds_read_b128 %1, %6 offset:16\n\t"
        "ds_read_b128 %2, %6 offset:32\n\t"
        "ds_read_b128 %3, %7\n\t"
        "ds_read_b64 %4, %8\n\t"
        "ds_read_b32 %5, %8 offset:8\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(old), "=&v"(l01), "=&v"(l2)
        : "v"(base + (uint32_t)lane * 48u), "v"(base + 3072u + (uint32_t)lane * 16u),
          "v"(base + 4096u)
        : "memory");
    DecodeRow c;
    c.t[0] = q0.x; c.t[1] = q0.y; c.t[2] = q0.z; c.t[3] = q0.w;
    c.t[4] = q1.x; c.t[5] = q1.y; c.t[6] = q1.z; c.t[7] = q1.w;
    c.t[8] = q2.x; c.t[9] = q2.y; c.t[10] = q2.z; c.t[11] = q2.w;
    c.l[0] = has_left_strip ? l01.x : 0u;
    c.l[1] = has_left_strip ? l01.y : 0u;
    c.l[2] = has_left_strip ? l2 : 0u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const uint32_t from_left = dpp_from_lane_below(c.t[9 + ch]);
      if (lane != 0) c.l[ch] = from_left;
    }
    if (i == 0) {  // the row above my first one: no output; above the frame it counts as zero
      if (!has_up) {
#pragma unroll
        for (int k = 0; k < 12; ++k) c.t[k] = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) c.l[k] = 0;
      }
      up = c;
      continue;
    }
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t v[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const uint32_t left = k ? c.t[3 * (k - 1) + ch] : c.l[ch];
        const uint32_t up_left = k ? up.t[3 * (k - 1) + ch] : up.l[ch];
        v[ch] = min(c.t[3 * k + ch] - up.t[3 * k + ch] + up_left - left, 255u);
      }
      px[k] = (old[k] & 0xff000000u) | v[0] | (v[1] << 8) | (v[2] << 16);
    }
    if (writes)
      store_b128_uncounted(dst, (uint32_t)(y0 - 1 + i) * (uint32_t)dst_linesize + (uint32_t)x_own * 4u,
                           u32x4_t{px[0], px[1], px[2], px[3]});
    up = c;
  }
  // nothing may still be landing in this wave's LDS when the workgroup's allocation is reused
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// interpolate_rect_kernel (src/sat_decoder_interpolate_kernel.cl:1-152).
// tx / ty hold, per pixel offset from the gaze centre, the inverse map u, the
// forward map of u and of its neighbour (host_tables.h: InterpAxisEntry).
// Everything the kernel text computes from one axis only (:26-65, :75-142), for one pixel
// position: the reduced-buffer index of the exact hit, the two bilinear neighbours and the
// interpolation weight.  `pos` is the (x: already wrapped) coordinate, `entry` its row of the
// host table, `wrapped` the kernel's x_offset flag (always false for y).
struct InterpAxis {
  int exact_idx, lo, hi;
  float ratio;
  bool exact;
};
__device__ __forceinline__ InterpAxis interp_axis(int pos, int centre, int4 entry, bool wrapped,
                                                  int out_size, int rsize, int src_size) {
  const int u = entry.x, dcalc = entry.y, dmin = entry.z, du = entry.w;
  InterpAxis r;
  r.exact = dcalc == pos - centre;
  const int a0 = centre + dmin, a1 = centre + dcalc;
  const int mn = min(a0, a1), mx = max(a0, a1);
  int min_u = min(u, u + du), max_u = max(u, u + du);
  if (mn < 0 && !wrapped) min_u = max_u;  // :105-116
  if (mx >= out_size && !wrapped) max_u = min_u;
  r.lo = min(max(min_u + rsize / 2, 0), src_size - 1);
  r.hi = min(max(max_u + rsize / 2, 0), src_size - 1);
  r.exact_idx = min(max(u + rsize / 2, 0), src_size - 1);
  r.ratio = mx == mn ? 0.0f
                     : fminf(fmaxf((float)(pos - mn) / (float)(mx - mn), 0.0f), 1.0f);
  return r;
}

// A lane owns 4 adjacent output columns (their axis records stay in registers, one 16-byte
// store per row), a wave 256 columns x `rows` rows; the row's axis record is wave-uniform.
constexpr int kInterpCols = 4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// STAGED ("interp.staged", the default): away from the fovea up to 23 adjacent output columns
// lie between the same two reduced columns, and the vertical lerps of the kernel text (:143-146:
// mix(top, bottom, y_ratio) for the left and for the right neighbour) depend on the reduced
// column only -- so the wave computes them ONCE per reduced column of its span (at most
// kInterpSpan x 64 of them, loaded as coalesced rows instead of four gathers per pixel), parks
// them in wave-private LDS as float3, and an output pixel is two 16-byte LDS reads and the
// horizontal lerp.  The arithmetic per value is the kernel text's, operation for operation
// (unfused, same order), so the bytes are unchanged; a wave whose columns straddle the wrap seam
// (reduced columns from both ends of the row) keeps the per-pixel gathers.
constexpr int kInterpSpan = 5;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void interp_lds_write(uint32_t addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// eight 16-byte reads (left / right vertical lerps of a lane's four pixels), one wait
__device__ __forceinline__ void interp_lds_read8(const uint32_t (&a)[4], const uint32_t (&b)[4],
                                                 f32x4 (&l)[4], f32x4 (&r)[4]) {
  asm volatile(
      "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\t"
      "ds_read_b128 %3, %11\n\tds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\t"
      "ds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]),
        "=&v"(r[3])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
      : "memory");
}

template <bool STAGED>
__global__ __launch_bounds__(256) void interpolate_rect_kernel(
    uint32_t *__restrict__ dst, int out_w, int out_h,
    const uint32_t *__restrict__ src, int src_w, int src_h,
    const int4 *__restrict__ tx, int range_x, const int4 *__restrict__ ty,
    int range_y, int cxp, int cyp, int rows) {
  __shared__ __attribute__((aligned(16))) float lerp_stage[STAGED ? 4 * kInterpSpan * 64 * 4 : 4];
  const int lane = threadIdx.x & 63;
  const int x_base = ((int)blockIdx.x * 64 + lane) * kInterpCols;
  const int y0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * rows);
  if (y0 >= out_h) return;
  const bool vec_store = (out_w % kInterpCols) == 0;

  InterpAxis ax[kInterpCols];
#pragma unroll
  for (int k = 0; k < kInterpCols; ++k) {
    int x = min(x_base + k, out_w - 1);
    bool x_offset = false;
    if (x - cxp > out_w / 2) {  // :27-33
      x -= out_w;
      x_offset = true;
    } else if (x - cxp < (-out_w) / 2) {
      x += out_w;
      x_offset = true;
    }
    ax[k] = interp_axis(x, cxp, tx[x - cxp + range_x], x_offset, out_w, src_w, src_w);
  }

  // 32-bit byte offsets into the reduced frame (it is far below 4 GiB)
  uint32_t off_lo[kInterpCols], off_hi[kInterpCols], off_ex[kInterpCols];
  bool all_exact_x = true;
#pragma unroll
  for (int k = 0; k < kInterpCols; ++k) {
    off_lo[k] = (uint32_t)ax[k].lo * 4u;
    off_hi[k] = (uint32_t)ax[k].hi * 4u;
    off_ex[k] = (uint32_t)ax[k].exact_idx * 4u;
    all_exact_x = all_exact_x && ax[k].exact;
  }
  const bool wave_exact_x = __all(all_exact_x);  // the fovea: every column is an exact hit
  const char *srcb = reinterpret_cast<const char *>(src);
  const uint32_t row_bytes = (uint32_t)src_w * 4u;
  auto texel = [&](uint32_t byte_off) {
    return *reinterpret_cast<const uint32_t *>(srcb + byte_off);
  };

  // The four neighbours of a pixel are fetched once per distinct (column pair, row pair): in the
  // periphery several adjacent output columns (rows) map to the same pair of reduced columns
  // (rows), so a column whose pair equals its left neighbour's copies it, and a row whose pair
  // equals the previous row's keeps the registers.  The kernel is bound by the number of
  // gathered lanes, not by bytes.
  bool same_as_left[kInterpCols];
  same_as_left[0] = false;
#pragma unroll
  for (int k = 1; k < kInterpCols; ++k)
    same_as_left[k] = off_lo[k] == off_lo[k - 1] && off_hi[k] == off_hi[k - 1];
  uint32_t tl[kInterpCols], tr[kInterpCols], bl[kInterpCols], br[kInterpCols];
  int held_lo = -1, held_hi = -1;  // reduced rows the registers above hold

  const int y1 = min(y0 + rows, out_h);
  if constexpr (STAGED) {
    // the wave's span of reduced columns
    int c_lo = ax[0].lo, c_hi = ax[0].hi;
#pragma unroll
    for (int k = 1; k < kInterpCols; ++k) {
      c_lo = min(c_lo, ax[k].lo);
      c_hi = max(c_hi, ax[k].hi);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      c_lo = min(c_lo, __shfl_xor(c_lo, m, 64));
      c_hi = max(c_hi, __shfl_xor(c_hi, m, 64));
    }
    const int cmin = __builtin_amdgcn_readfirstlane(c_lo);
    const int cmax = __builtin_amdgcn_readfirstlane(c_hi);
    if (cmax - cmin < kInterpSpan * 64) {
      const int ngroups = (cmax - cmin) / 64 + 1;  // 64-column groups the wave stages
      const uint32_t mine = (uint32_t)reinterpret_cast<uintptr_t>(lerp_stage) +
                            (uint32_t)(threadIdx.x >> 6) * (kInterpSpan * 64 * 16);
      uint32_t at_lo[kInterpCols], at_hi[kInterpCols];
#pragma unroll
      for (int k = 0; k < kInterpCols; ++k) {
        at_lo[k] = mine + (uint32_t)(ax[k].lo - cmin) * 16u;
        at_hi[k] = mine + (uint32_t)(ax[k].hi - cmin) * 16u;
      }
      // Away from the fovea the span is one or two groups: then the texels of ALL the wave's rows
      // are requested before the first is used (a row pair that repeats is a cache hit), so the
      // wave pays one memory round trip instead of one per distinct pair of reduced rows.
      constexpr int kPre = 8;  // rows whose texels are requested together
      if (ngroups <= 2) {
       for (int yc = y0; yc < y1; yc += kPre) {  // chunks of kPre rows
        uint32_t ptop[kPre][2], pbot[kPre][2];
#pragma unroll
        for (int r = 0; r < kPre; ++r) {
          const int y = min(yc + r, y1 - 1);
          const InterpAxis ay = interp_axis(y, cyp, ty[y - cyp + range_y], false, out_h, src_h, src_h);
          const uint32_t r_lo = (uint32_t)ay.lo * row_bytes, r_hi = (uint32_t)ay.hi * row_bytes;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t c = (uint32_t)min(cmin + lane + 64 * j, cmax) * 4u;
            ptop[r][j] = texel(r_lo + c);
            pbot[r][j] = texel(r_hi + c);
          }
        }
#pragma unroll
        for (int r = 0; r < kPre; ++r) {
          const int y = yc + r;
          if (y >= y1) break;
          const InterpAxis ay = interp_axis(y, cyp, ty[y - cyp + range_y], false, out_h, src_h, src_h);
          uint32_t out[kInterpCols];
          const float yr = ay.ratio;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (j >= ngroups) break;  // wave-uniform
            f32x4 v;
            v.w = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float ft = (float)((ptop[r][j] >> (8 * c)) & 0xffu);
              const float fb = (float)((pbot[r][j] >> (8 * c)) & 0xffu);
              const float lerp = ft + (fb - ft) * yr;
              if (c == 0) v.x = lerp;
              if (c == 1) v.y = lerp;
              if (c == 2) v.z = lerp;
            }
            interp_lds_write(mine + (uint32_t)(lane + 64 * j) * 16u, v);
          }
          f32x4 l[kInterpCols], rr[kInterpCols];
          interp_lds_read8(at_lo, at_hi, l, rr);
#pragma unroll
          for (int k = 0; k < kInterpCols; ++k) {
            const float xr = ax[k].ratio;
            const float m0 = l[k].x + (rr[k].x - l[k].x) * xr;
            const float m1 = l[k].y + (rr[k].y - l[k].y) * xr;
            const float m2 = l[k].z + (rr[k].z - l[k].z) * xr;
            out[k] = ((uint32_t)(int)m0 & 0xffu) | (((uint32_t)(int)m1 & 0xffu) << 8) |
                     (((uint32_t)(int)m2 & 0xffu) << 16);
          }
          if (ay.exact) {  // wave-uniform: rows that hit a reduced row exactly
#pragma unroll
            for (int k = 0; k < kInterpCols; ++k)
              if (ax[k].exact)  // exact hit on both axes -> plain copy
                out[k] = texel((uint32_t)ay.exact_idx * row_bytes + off_ex[k]) & 0x00ffffffu;
          }
          uint32_t *o = dst + (size_t)y * out_w + x_base;
          if (vec_store && x_base < out_w) {
            F360_STREAM_STORE(reinterpret_cast<u32x4_t *>(o),
                              (u32x4_t{out[0], out[1], out[2], out[3]}));
          } else {
#pragma unroll
            for (int k = 0; k < kInterpCols; ++k)
              if (x_base + k < out_w) o[k] = out[k];
          }
        }
       }
        return;
      }
      uint32_t top[kInterpSpan], bot[kInterpSpan];
      int row_lo = -1, row_hi = -1;
      for (int y = y0; y < y1; ++y) {
        const InterpAxis ay = interp_axis(y, cyp, ty[y - cyp + range_y], false, out_h, src_h, src_h);
        uint32_t out[kInterpCols];
        if (ay.exact && wave_exact_x) {  // :67-72 for the whole wave: plain copies
#pragma unroll
          for (int k = 0; k < kInterpCols; ++k)
            out[k] = texel((uint32_t)ay.exact_idx * row_bytes + off_ex[k]) & 0x00ffffffu;
        } else {
          if (ay.lo != row_lo || ay.hi != row_hi) {  // wave-uniform: the two reduced rows
            const uint32_t r_lo = (uint32_t)ay.lo * row_bytes, r_hi = (uint32_t)ay.hi * row_bytes;
#pragma unroll
            for (int j = 0; j < kInterpSpan; ++j) {
              const uint32_t c = (uint32_t)min(cmin + lane + 64 * j, cmax) * 4u;
              top[j] = texel(r_lo + c);
              bot[j] = texel(r_hi + c);
            }
            row_lo = ay.lo;
            row_hi = ay.hi;
          }
          const float yr = ay.ratio;
#pragma unroll
          for (int j = 0; j < kInterpSpan; ++j) {
            if (j >= ngroups) break;  // wave-uniform
            f32x4 v;
            v.w = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float ft = (float)((top[j] >> (8 * c)) & 0xffu);
              const float fb = (float)((bot[j] >> (8 * c)) & 0xffu);
              const float lerp = ft + (fb - ft) * yr;
              if (c == 0) v.x = lerp;
              if (c == 1) v.y = lerp;
              if (c == 2) v.z = lerp;
            }
            interp_lds_write(mine + (uint32_t)(lane + 64 * j) * 16u, v);
          }
          f32x4 l[kInterpCols], r[kInterpCols];
          interp_lds_read8(at_lo, at_hi, l, r);
#pragma unroll
          for (int k = 0; k < kInterpCols; ++k) {
            const float xr = ax[k].ratio;
            const float m0 = l[k].x + (r[k].x - l[k].x) * xr;
            const float m1 = l[k].y + (r[k].y - l[k].y) * xr;
            const float m2 = l[k].z + (r[k].z - l[k].z) * xr;
            out[k] = ((uint32_t)(int)m0 & 0xffu) | (((uint32_t)(int)m1 & 0xffu) << 8) |
                     (((uint32_t)(int)m2 & 0xffu) << 16);
          }
#pragma unroll
          for (int k = 0; k < kInterpCols; ++k)
            if (ay.exact && ax[k].exact)  // exact hit on both axes -> plain copy
              out[k] = texel((uint32_t)ay.exact_idx * row_bytes + off_ex[k]) & 0x00ffffffu;
        }
        uint32_t *o = dst + (size_t)y * out_w + x_base;
        if (vec_store && x_base < out_w) {
          F360_STREAM_STORE(reinterpret_cast<u32x4_t *>(o),
                            (u32x4_t{out[0], out[1], out[2], out[3]}));
        } else {
#pragma unroll
          for (int k = 0; k < kInterpCols; ++k)
            if (x_base + k < out_w) o[k] = out[k];
        }
      }
      return;
    }
  }
  for (int y = y0; y < y1; ++y) {
    const InterpAxis ay = interp_axis(y, cyp, ty[y - cyp + range_y], false, out_h, src_h, src_h);
    uint32_t out[kInterpCols];
    if (ay.exact && wave_exact_x) {  // :67-72 for the whole wave: plain copies
#pragma unroll
      for (int k = 0; k < kInterpCols; ++k)
        out[k] = texel((uint32_t)ay.exact_idx * row_bytes + off_ex[k]) & 0x00ffffffu;
    } else {
      if (ay.lo != held_lo || ay.hi != held_hi) {  // wave-uniform
        const uint32_t r_lo = (uint32_t)ay.lo * row_bytes, r_hi = (uint32_t)ay.hi * row_bytes;
#pragma unroll
        for (int k = 0; k < kInterpCols; ++k) {
          if (!same_as_left[k]) {
            tl[k] = texel(r_lo + off_lo[k]);
            tr[k] = texel(r_lo + off_hi[k]);
            bl[k] = texel(r_hi + off_lo[k]);
            br[k] = texel(r_hi + off_hi[k]);
          }
        }
#pragma unroll
        for (int k = 1; k < kInterpCols; ++k) {
          if (same_as_left[k]) {
            tl[k] = tl[k - 1];
            tr[k] = tr[k - 1];
            bl[k] = bl[k - 1];
            br[k] = br[k - 1];
          }
        }
        held_lo = ay.lo;
        held_hi = ay.hi;
      }
      // two pixels per packed-float instruction (v_pk_mul_f32 / v_pk_add_f32): the same IEEE
      // operations in the same order as the scalar form, so the results are unchanged
#pragma unroll
      for (int k = 0; k < kInterpCols; k += 2) {
        const f32x2 xr = {ax[k].ratio, ax[k + 1].ratio};
        const f32x2 yr = {ay.ratio, ay.ratio};
        uint32_t v0 = 0, v1 = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2 ftl = {(float)((tl[k] >> (8 * c)) & 0xffu), (float)((tl[k + 1] >> (8 * c)) & 0xffu)};
          const f32x2 fbl = {(float)((bl[k] >> (8 * c)) & 0xffu), (float)((bl[k + 1] >> (8 * c)) & 0xffu)};
          const f32x2 ftr = {(float)((tr[k] >> (8 * c)) & 0xffu), (float)((tr[k + 1] >> (8 * c)) & 0xffu)};
          const f32x2 fbr = {(float)((br[k] >> (8 * c)) & 0xffu), (float)((br[k + 1] >> (8 * c)) & 0xffu)};
          const f32x2 l = ftl + (fbl - ftl) * yr;
          const f32x2 r = ftr + (fbr - ftr) * yr;
          const f32x2 m = l + (r - l) * xr;
          v0 |= ((uint32_t)(int)m.x & 0xffu) << (8 * c);
          v1 |= ((uint32_t)(int)m.y & 0xffu) << (8 * c);
        }
        out[k] = v0;
        out[k + 1] = v1;
      }
#pragma unroll
      for (int k = 0; k < kInterpCols; ++k)
        if (ay.exact && ax[k].exact)  // exact hit on both axes -> plain copy
          out[k] = texel((uint32_t)ay.exact_idx * row_bytes + off_ex[k]) & 0x00ffffffu;
    }
    uint32_t *o = dst + (size_t)y * out_w + x_base;
    if (vec_store && x_base < out_w) {
      F360_STREAM_STORE(reinterpret_cast<u32x4_t *>(o), (u32x4_t{out[0], out[1], out[2], out[3]}));
    } else {
#pragma unroll
      for (int k = 0; k < kInterpCols; ++k)
        if (x_base + k < out_w) o[k] = out[k];
    }
  }
}

int upload(f360_ctx *ctx, f360::DevBuf &buf, const void *host, size_t bytes) {
  // The tables may still be read by kernels queued earlier on the stream.
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  int st = buf.reserve(bytes);
  if (st != F360_OK) return st;
  F360_HIP_TRY(hipMemcpy(buf.p, host, bytes, hipMemcpyHostToDevice));
  return F360_OK;
}

int ensure_interp_tables(f360_sat_decoder *dec, int w, int h, int rw, int rh,
                         int need_dx, int need_dy) {
  const bool same_geo =
      dec->it_w == w && dec->it_h == h && dec->it_rw == rw && dec->it_rh == rh;
  if (same_geo && dec->it_dx >= need_dx && dec->it_dy >= need_dy) return F360_OK;
  // Default coverage: every gaze centre in [-0.5, 1.5] x [-1, 2].
  const int rx = need_dx > w ? need_dx : w;
  const int ry = need_dy > 2 * h ? need_dy : 2 * h;
  std::vector<f360::InterpAxisEntry> t;
  f360::build_interp_axis(t, rx, w, rw);
  int st = upload(dec->ctx, dec->itx_dev, t.data(), t.size() * sizeof(t[0]));
  if (st != F360_OK) return st;
  f360::build_interp_axis(t, ry, h, rh);
  st = upload(dec->ctx, dec->ity_dev, t.data(), t.size() * sizeof(t[0]));
  if (st != F360_OK) return st;
  dec->it_w = w;
  dec->it_h = h;
  dec->it_rw = rw;
  dec->it_rh = rh;
  dec->it_dx = rx;
  dec->it_dy = ry;
  return F360_OK;
}

inline int iabs(int v) { return v < 0 ? -v : v; }
inline int imax(int a, int b) { return a > b ? a : b; }

}  // namespace

extern "C" {

int f360_satdec_create(f360_ctx *ctx, f360_sat_decoder **out) {
  F360_REQUIRE(ctx && out, "f360_satdec_create: null argument");
  f360_sat_decoder *d = new f360_sat_decoder();
  d->ctx = ctx;
  *out = d;
  return F360_OK;
}

int f360_satdec_destroy(f360_sat_decoder *dec) {
  if (!dec) return F360_OK;
  (void)hipStreamSynchronize(dec->ctx->stream);
  dec->gx_dev.release();
  dec->gy_dev.release();
  dec->itx_dev.release();
  dec->ity_dev.release();
  dec->lbx_dev.release();
  dec->fov_maps.release();
  dec->fov_corners.release();
  delete dec;
  return F360_OK;
}

int f360_satdec_initialize_grid(f360_sat_decoder *dec, int target_width,
                                int target_height, int source_width,
                                int source_height) {
  F360_REQUIRE(dec, "f360_satdec_initialize_grid: null decoder");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2,
               "f360_satdec_initialize_grid: bad geometry %dx%d <- %dx%d",
               target_width, target_height, source_width, source_height);
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_satdec_initialize_grid: a dimension exceeds 65536");
  if (dec->gw == target_width && dec->gh == target_height &&
      dec->sw == source_width && dec->sh == source_height && dec->gx_dev.p)
    return F360_OK;
  F360_BIND_DEVICE(dec->ctx);
  f360::build_satdec_grid_axis(dec->gx_host, target_width, source_width);
  f360::build_satdec_grid_axis(dec->gy_host, target_height, source_height);
  int st = upload(dec->ctx, dec->gx_dev, dec->gx_host.data(),
                  dec->gx_host.size() * sizeof(int16_t));
  if (st != F360_OK) return st;
  st = upload(dec->ctx, dec->gy_dev, dec->gy_host.data(),
              dec->gy_host.size() * sizeof(int16_t));
  if (st != F360_OK) return st;
  // inverse of the x grid for the tile streamer: lbx[d - dmin] = first grid index whose
  // offset is >= d; plus the largest step between neighbouring corners (its halo)
  {
    const std::vector<int16_t> &g = dec->gx_host;
    // A degenerate geometry (a reduced width of 2 or 3 for a wider source) overflows the
    // reference's float -> int conversion and leaves a grid that is not even monotonic
    // (e.g. {11668, -2, 1}); the gather kernels take any grid, the streamer needs the inverse.
    const bool monotonic = std::is_sorted(g.begin(), g.end());
    const int dmin = monotonic ? (int)g.front() : 0, dmax = monotonic ? (int)g.back() + 1 : 0;
    std::vector<int> lb((size_t)(dmax - dmin + 1));
    size_t gi = 0;
    for (int d = dmin; d <= dmax; ++d) {
      while (gi < g.size() && g[gi] < d) ++gi;
      lb[(size_t)(d - dmin)] = (int)gi;
    }
    int max_step = 1, max_step_y = 1;
    for (size_t k = 1; k < g.size(); ++k) max_step = std::max(max_step, g[k] - g[k - 1]);
    for (size_t k = 1; k < dec->gy_host.size(); ++k)
      max_step_y = std::max(max_step_y, dec->gy_host[k] - dec->gy_host[k - 1]);
    dec->halo = (max_step + 1 + 3) & ~3;
    dec->lb_dmin = dmin;
    dec->lb_n = (int)lb.size();
    // a tile's halo is at most 512 bytes (42 texels); box heights travel in 8 bits
    dec->stream_ok = monotonic && dec->halo <= 40 && max_step_y < 256 &&
                     std::is_sorted(dec->gy_host.begin(), dec->gy_host.end());
    st = upload(dec->ctx, dec->lbx_dev, lb.data(), lb.size() * sizeof(int));
    if (st != F360_OK) return st;
    dec->lbx_host = std::move(lb);
  }
  dec->gw = target_width;
  dec->gh = target_height;
  dec->sw = source_width;
  dec->sh = source_height;
  return F360_OK;
}

int f360_satdec_export_grid(f360_sat_decoder *dec, int16_t *grid_host) {
  F360_REQUIRE(dec && grid_host, "f360_satdec_export_grid: null argument");
  F360_BIND_DEVICE(dec->ctx);
  if (!dec->gx_dev.p) {
    f360::set_error("f360_satdec_export_grid: grid not initialised");
    return F360_ERR_NOT_INITIALIZED;
  }
  // read back what the kernels actually use
  std::vector<int16_t> gx(dec->gx_host.size()), gy(dec->gy_host.size());
  F360_HIP_TRY(hipStreamSynchronize(dec->ctx->stream));
  F360_HIP_TRY(hipMemcpy(gx.data(), dec->gx_dev.p, gx.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost));
  F360_HIP_TRY(hipMemcpy(gy.data(), dec->gy_dev.p, gy.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost));
  const int gw = dec->gw + 1;
  for (int ty = 0; ty <= dec->gh; ++ty)
    for (int tx = 0; tx <= dec->gw; ++tx) {
      grid_host[((size_t)ty * gw + tx) * 2 + 0] = gx[(size_t)tx];
      grid_host[((size_t)ty * gw + tx) * 2 + 1] = gy[(size_t)ty];
    }
  return F360_OK;
}

int f360_satdec_sample_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                            int target_width, int target_height,
                            int target_linesize, const uint32_t *sat_dev,
                            int source_width, int source_height, float center_x,
                            float center_y) {
  F360_REQUIRE(dec, "f360_satdec_sample_rect: null decoder");
  F360_BIND_DEVICE(dec->ctx);
  F360_REQUIRE(target_dev && sat_dev, "f360_satdec_sample_rect: null buffer");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2 && target_linesize >= 4 * target_width,
               "f360_satdec_sample_rect: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_satdec_sample_rect: a dimension exceeds 65536");
  F360_REQUIRE(std::fabs(center_x) <= 16.0f && std::fabs(center_y) <= 16.0f,
               "f360_satdec_sample_rect: gaze centre out of range");
  if (!dec->gx_dev.p) {  // src/sat_decoder.cc:312-317
    int st = f360_satdec_initialize_grid(dec, target_width, target_height,
                                         source_width, source_height);
    if (st != F360_OK) return st;
  }
  F360_REQUIRE(dec->gw == target_width && dec->gh == target_height,
               "f360_satdec_sample_rect: grid was initialised for %dx%d, not %dx%d",
               dec->gw, dec->gh, target_width, target_height);
  // (int2)(center.x * source_width, center.y * source_height): float multiply,
  // truncation (:176-179)
  const int cxp = (int)(center_x * (float)source_width);
  const int cyp = (int)(center_y * (float)source_height);
  f360_ctx *ctx = dec->ctx;
  SampleArgs sa;
  sa.dst = target_dev;
  sa.out_w = target_width;
  sa.out_h = target_height;
  sa.out_stride_px = target_linesize / 4;
  sa.sat = sat_dev;
  sa.src_w = source_width;
  sa.src_h = source_height;
  sa.gx = dec->gx_dev.as<int16_t>();
  sa.gy = dec->gy_dev.as<int16_t>();
  sa.cxp = cxp;
  sa.cyp = cyp;
  sa.lbx = dec->lbx_dev.as<int>();
  sa.lb_dmin = dec->lb_dmin;
  sa.lb_n = dec->lb_n;
  sa.halo = dec->halo;
  sa.ablate = ctx->opt_ablate;
  const int variant = ctx->opt_sample_variant;
  const bool can_stream = tile_stream_applies(dec, sat_dev, source_width, source_height,
                                              target_linesize, target_height);
  // host-side checks first: the span's start event must sit right in front of the launch
  const bool fits = variant == 2 && can_stream &&
                    tile_stream_fits(dec, cxp, source_width, target_width);
  f360::KernelSpan span(ctx, f360::kSampleRect, f360::take_profile_slot(ctx));
  bool streamed = false;
  if (variant == 2 && can_stream) {
    const int ntiles = (source_width + kTsTile - 1) / kTsTile;
    if (fits) {
      // reduced rows per wave ("sample.srows", 0 = 4): a single frame's launch has nothing else
      // to overlap its waves' set-up with, so shorter runs -- more waves -- pay at every size
      // (4 against 8 rows: 13.6 -> 11.7 us at 1080p, 15.1 -> 14.2 at 2560x1440, 53.4 -> 51.5 at
      // 8K, round 5; 12 rows 56.6, 16 rows 59.7); the batched launches keep 8
      const int rows = std::min(ctx->opt_stream_rows > 0 ? ctx->opt_stream_rows : 4, kTsMaxRows);
      const int nblocks = (target_height + rows - 1) / rows;
      // (a ring of 3 slots: two table rows in flight; rings of 4 and 6, a tile order spread over
      // the XCDs and 16-byte group stores were A/B switches until round 4: EXPERIMENTS.md 3)
      const dim3 sgrid((unsigned)((ntiles * nblocks + 3) / 4));
      hipLaunchKernelGGL(sample_rect_stream_kernel<3>, sgrid, dim3(256), 0, ctx->stream, sa, rows,
                         nblocks);
      streamed = true;
    }
  }
  if (streamed) {
    // launched above
  } else if (variant >= 1 && (size_t)source_width * source_height * 12 < ((size_t)1 << 32)) {
    const int rows = ctx->opt_walk_rows;
    const dim3 wgrid((target_width + 4 * kWalkCols - 1) / (4 * kWalkCols),
                     (target_height + rows - 1) / rows);
    hipLaunchKernelGGL(sample_rect_walk_kernel, wgrid, dim3(256), 0, ctx->stream, sa, rows);
  } else {
    const dim3 grid((target_width + 63) / 64, (target_height + 3) / 4);
    hipLaunchKernelGGL(sample_rect_kernel, grid, dim3(256), 0, ctx->stream, target_dev,
                       target_width, target_height, target_linesize / 4, sat_dev,
                       source_width, source_height, dec->gx_dev.as<int16_t>(),
                       dec->gy_dev.as<int16_t>(), cxp, cyp);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

// `sats_dev` null: every gaze samples `sat_dev` (clients of one video); else frame k's table
static int sample_rect_batch_impl(f360_sat_decoder *dec, uint8_t *const *targets_dev, int count,
                                  int target_width, int target_height, int target_linesize,
                                  const uint32_t *sat_dev, const uint32_t *const *sats_dev,
                                  int source_width, int source_height,
                                  const float *centers_xy, int profile = -1) {
  F360_REQUIRE(dec, "f360_satdec_sample_rect_batch: null decoder");
  F360_BIND_DEVICE(dec->ctx);
  F360_REQUIRE(targets_dev && (sat_dev || sats_dev) && centers_xy,
               "f360_satdec_sample_rect_batch: null buffer");
  F360_REQUIRE(count >= 1 && count <= kMaxBatch,
               "f360_satdec_sample_rect_batch: count %d outside 1..%d", count, kMaxBatch);
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2 && target_linesize >= 4 * target_width,
               "f360_satdec_sample_rect_batch: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_satdec_sample_rect_batch: a dimension exceeds 65536");
  F360_REQUIRE((size_t)source_width * source_height * 12 < ((size_t)1 << 32),
               "f360_satdec_sample_rect_batch: table too large");
  if (!dec->gx_dev.p) {
    int st = f360_satdec_initialize_grid(dec, target_width, target_height, source_width,
                                         source_height);
    if (st != F360_OK) return st;
  }
  F360_REQUIRE(dec->gw == target_width && dec->gh == target_height,
               "f360_satdec_sample_rect_batch: grid was initialised for %dx%d", dec->gw,
               dec->gh);
  SampleBatch b;
  for (int k = 0; k < kMaxBatch; ++k) {
    const int q = k < count ? k : 0;
    F360_REQUIRE(targets_dev[q], "f360_satdec_sample_rect_batch: null target %d", q);
    F360_REQUIRE(std::fabs(centers_xy[2 * q]) <= 16.0f && std::fabs(centers_xy[2 * q + 1]) <= 16.0f,
                 "f360_satdec_sample_rect_batch: gaze centre out of range");
    b.dst[k] = targets_dev[q];
    b.sat[k] = sats_dev ? sats_dev[q] : sat_dev;
    F360_REQUIRE(b.sat[k], "f360_satdec_sample_rect_batch: null table %d", q);
    b.cxp[k] = (int)(centers_xy[2 * q] * (float)source_width);
    b.cyp[k] = (int)(centers_xy[2 * q + 1] * (float)source_height);
  }
  SampleArgs sa = {};
  sa.out_w = target_width;
  sa.out_h = target_height;
  sa.out_stride_px = target_linesize / 4;
  sa.sat = b.sat[0];
  sa.src_w = source_width;
  sa.src_h = source_height;
  sa.gx = dec->gx_dev.as<int16_t>();
  sa.gy = dec->gy_dev.as<int16_t>();
  sa.lbx = dec->lbx_dev.as<int>();
  sa.lb_dmin = dec->lb_dmin;
  sa.lb_n = dec->lb_n;
  sa.halo = dec->halo;
  f360_ctx *ctx = dec->ctx;
  sa.ablate = ctx->opt_ablate;
  f360::KernelSpan span(ctx, f360::kSampleRect,
                        profile < 0 ? f360::take_profile_slot(ctx) : profile != 0, count);
  // the tile streamer when every client's gaze passes its host checks, else the walker for all
  bool stream = ctx->opt_sample_variant == 2;
  for (int k = 0; k < count && stream; ++k)
    stream = tile_stream_applies(dec, b.sat[k], source_width, source_height, target_linesize,
                                 target_height) &&
             tile_stream_fits(dec, b.cxp[k], source_width, target_width);
  if (stream) {
    const int ntiles = (source_width + kTsTile - 1) / kTsTile;
    const int rows = std::min(ctx->opt_stream_rows > 0 ? ctx->opt_stream_rows : 8, kTsMaxRows);
    const int nblocks = (target_height + rows - 1) / rows;
    const dim3 sgrid((unsigned)((ntiles * nblocks + 3) / 4), (unsigned)count);
    hipLaunchKernelGGL(sample_rect_stream_batch_kernel<3>, sgrid, dim3(256), 0, ctx->stream, sa, b,
                       rows, nblocks);
  } else {
    const int rows = ctx->opt_walk_rows;
    const dim3 grid((target_width + 4 * kWalkCols - 1) / (4 * kWalkCols),
                    (target_height + rows - 1) / rows, count);
    hipLaunchKernelGGL(sample_rect_walk_batch_kernel, grid, dim3(256), 0, ctx->stream, sa, b,
                       rows);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}


int f360_satdec_sample_rect_batch(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                  int count, int target_width, int target_height,
                                  int target_linesize, const uint32_t *sat_dev,
                                  int source_width, int source_height,
                                  const float *centers_xy) {
  F360_REQUIRE(sat_dev, "f360_satdec_sample_rect_batch: null table");
  F360_REQUIRE(count <= kMaxClients, "f360_satdec_sample_rect_batch: count %d outside 1..%d", count,
               kMaxClients);
  return sample_rect_batch_impl(dec, targets_dev, count, target_width, target_height,
                                target_linesize, sat_dev, nullptr, source_width, source_height,
                                centers_xy);
}

int f360_satdec_sample_rect_frames(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                   int count, int target_width, int target_height,
                                   int target_linesize, const uint32_t *const *sats_dev,
                                   int source_width, int source_height,
                                   const float *centers_xy) {
  F360_REQUIRE(dec && sats_dev && count >= 1, "f360_satdec_sample_rect_frames: bad arguments");
  const int prof = f360::take_profile_slot(dec->ctx) ? 1 : 0;  // one slot for the whole call
  // more frames than one launch takes ("sample.fpl", at most 64): consecutive launches
  const int per_launch = std::min(std::max(dec->ctx->opt_sample_fpl, 1), kMaxBatch);
  for (int k = 0; k < count; k += per_launch) {
    const int n = std::min(count - k, per_launch);
    const int st = sample_rect_batch_impl(dec, targets_dev + k, n, target_width, target_height,
                                          target_linesize, nullptr, sats_dev + k, source_width,
                                          source_height, centers_xy + 2 * k, prof);
    if (st != F360_OK) return st;
  }
  return F360_OK;
}

// Encode + sample in one pass over the frames: tables AND reduced frames, byte for byte what
// f360_sat_encode_batch followed by f360_satdec_sample_rect_frames leave (which is what this
// call does whenever the one-pass form does not apply: too few frames for the read-once encoder,
// a source the read-once encoder does not take, a grid whose offsets do not increase strictly).
// `planes` non-null: planar YUV 4:2:0 sources (sources_dev / source_linesize unused).
static int encode_sample_frames_impl(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                     uint32_t *const *sats_dev, const uint8_t *const *sources_dev,
                                     const f360::YuvPlanes *planes, int count, int target_width,
                                     int target_height, int target_linesize, int source_width,
                                     int source_height, int source_linesize,
                                     const float *centers_xy) {
  // `sats_dev` null: no tables wanted (f360_satdec_foveate_rect_frames[_yuv420p])
  F360_REQUIRE(dec && targets_dev && (sources_dev || planes) && centers_xy && count >= 1,
               "f360_satdec_encode_sample_frames: bad arguments");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2 && target_linesize >= 4 * target_width &&
                   (planes || source_linesize >= 1),
               "f360_satdec_encode_sample_frames: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}),
               "f360_satdec_encode_sample_frames: a dimension exceeds 65536");
  f360_ctx *ctx = dec->ctx;
  // two one-pass forms: the strip walker's wherever the read-once encoder takes the call, the
  // band writer's (RGB0 frames, tables wanted) for calls too small for that -- from one frame up
  const bool walks = ctx->opt_fuse_walk != 0 &&
                     f360::sat_encode_sample_applies(ctx, count, source_width, source_height,
                                                     source_linesize, target_width, target_height,
                                                     target_linesize, planes);
  const bool bands = !walks && sats_dev &&
                     f360::sat_encode_sample_band_applies(ctx, count, source_width, source_height,
                                                          source_linesize, target_width,
                                                          target_height, target_linesize, planes);
  bool one_pass = walks || bands;
  for (int k = 0; k < count && one_pass; ++k) {
    one_pass = targets_dev[k] && (!sats_dev || (sats_dev[k] && ((uintptr_t)sats_dev[k] % 16) == 0)) &&
               ((uintptr_t)targets_dev[k] % 4) == 0 && std::fabs(centers_xy[2 * k]) <= 16.0f &&
               std::fabs(centers_xy[2 * k + 1]) <= 16.0f;
    if (planes)
      one_pass = one_pass && planes[k].y && planes[k].u && planes[k].v &&
                 ((uintptr_t)planes[k].y % 4) == 0 && ((uintptr_t)planes[k].u % 2) == 0 &&
                 ((uintptr_t)planes[k].v % 2) == 0;
    else
      one_pass = one_pass && sources_dev[k] && ((uintptr_t)sources_dev[k] % 16) == 0;
  }
  if (one_pass) {
    F360_BIND_DEVICE(ctx);
    if (!dec->gx_dev.p) {
      int st = f360_satdec_initialize_grid(dec, target_width, target_height, source_width,
                                           source_height);
      if (st != F360_OK) return st;
    }
    F360_REQUIRE(dec->gw == target_width && dec->gh == target_height,
                 "f360_satdec_encode_sample_frames: grid was initialised for %dx%d", dec->gw,
                 dec->gh);
    // the row plan's reasoning (sat_fuse.hip: walk_fuse_plan_kernel) and its packed fields
    auto increasing = [](const std::vector<int16_t> &g, int max_step) {
      for (size_t k = 1; k < g.size(); ++k)
        if (g[k] <= g[k - 1] || g[k] - g[k - 1] > max_step) return false;
      return true;
    };
    one_pass = increasing(dec->gx_host, 1 << 20) && increasing(dec->gy_host, 1023);
  }
  if (one_pass) {
    F360_BIND_DEVICE(ctx);
    const f360::SatFuse fuse{targets_dev, centers_xy, dec->gx_dev.as<int16_t>(),
                             dec->gy_dev.as<int16_t>(), target_width, target_height,
                             target_linesize};
    if (bands)
      return f360::sat_encode_sample_band(ctx, count, sats_dev, sources_dev, planes, source_width,
                                          source_height, source_linesize, fuse,
                                          f360::take_profile_slot(ctx));
    return f360::sat_encode_sample_walk(ctx, count, sats_dev, sources_dev, planes, source_width,
                                        source_height, source_linesize, fuse,
                                        f360::take_profile_slot(ctx));
  }
  int st;
  if (!sats_dev) {  // reduced frames only: the single-frame fused call, frame by frame
    for (int k = 0; k < count; ++k) {
      st = planes ? f360_satdec_foveate_rect_yuv420p(
                        dec, targets_dev[k], target_width, target_height, target_linesize,
                        planes[k].y, planes[k].u, planes[k].v, planes[k].y_linesize,
                        planes[k].u_linesize, planes[k].v_linesize, source_width, source_height,
                        centers_xy[2 * k], centers_xy[2 * k + 1])
                  : f360_satdec_foveate_rect(dec, targets_dev[k], target_width, target_height,
                                             target_linesize, sources_dev[k], source_width,
                                             source_height, source_linesize, centers_xy[2 * k],
                                             centers_xy[2 * k + 1]);
      if (st != F360_OK) return st;
    }
    return F360_OK;
  }
  if (planes) {
    std::vector<const uint8_t *> y((size_t)count), u((size_t)count), v((size_t)count);
    for (int k = 0; k < count; ++k) {
      y[(size_t)k] = planes[k].y;
      u[(size_t)k] = planes[k].u;
      v[(size_t)k] = planes[k].v;
    }
    st = f360_sat_encode_yuv420p_batch(ctx, count, sats_dev, y.data(), u.data(), v.data(),
                                       planes[0].y_linesize, planes[0].u_linesize,
                                       planes[0].v_linesize, source_width, source_height);
  } else {
    st = f360_sat_encode_batch(ctx, count, sats_dev, sources_dev, source_width, source_height,
                               source_linesize);
  }
  if (st != F360_OK) return st;
  return f360_satdec_sample_rect_frames(dec, targets_dev, count, target_width, target_height,
                                        target_linesize, sats_dev, source_width, source_height,
                                        centers_xy);
}

int f360_satdec_encode_sample_frames(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                     uint32_t *const *sats_dev, const uint8_t *const *sources_dev,
                                     int count, int target_width, int target_height,
                                     int target_linesize, int source_width, int source_height,
                                     int source_linesize, const float *centers_xy) {
  return encode_sample_frames_impl(dec, targets_dev, sats_dev, sources_dev, nullptr, count,
                                   target_width, target_height, target_linesize, source_width,
                                   source_height, source_linesize, centers_xy);
}

int f360_satdec_foveate_rect_frames(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                    const uint8_t *const *sources_dev, int count,
                                    int target_width, int target_height, int target_linesize,
                                    int source_width, int source_height, int source_linesize,
                                    const float *centers_xy) {
  return encode_sample_frames_impl(dec, targets_dev, nullptr, sources_dev, nullptr, count,
                                   target_width, target_height, target_linesize, source_width,
                                   source_height, source_linesize, centers_xy);
}

int f360_satdec_foveate_rect_frames_yuv420p(
    f360_sat_decoder *dec, uint8_t *const *targets_dev, const uint8_t *const *y_dev,
    const uint8_t *const *u_dev, const uint8_t *const *v_dev, int y_linesize, int u_linesize,
    int v_linesize, int count, int target_width, int target_height, int target_linesize,
    int source_width, int source_height, const float *centers_xy) {
  F360_REQUIRE(y_dev && u_dev && v_dev && count >= 1,
               "f360_satdec_foveate_rect_frames_yuv420p: bad arguments");
  std::vector<f360::YuvPlanes> planes((size_t)count);
  for (int k = 0; k < count; ++k)
    planes[(size_t)k] =
        f360::YuvPlanes{y_dev[k], u_dev[k], v_dev[k], y_linesize, u_linesize, v_linesize};
  return encode_sample_frames_impl(dec, targets_dev, nullptr, nullptr, planes.data(), count,
                                   target_width, target_height, target_linesize, source_width,
                                   source_height, 0, centers_xy);
}

int f360_satdec_encode_sample_frames_yuv420p(
    f360_sat_decoder *dec, uint8_t *const *targets_dev, uint32_t *const *sats_dev,
    const uint8_t *const *y_dev, const uint8_t *const *u_dev, const uint8_t *const *v_dev,
    int y_linesize, int u_linesize, int v_linesize, int count, int target_width,
    int target_height, int target_linesize, int source_width, int source_height,
    const float *centers_xy) {
  F360_REQUIRE(y_dev && u_dev && v_dev && count >= 1,
               "f360_satdec_encode_sample_frames_yuv420p: bad arguments");
  std::vector<f360::YuvPlanes> planes((size_t)count);
  for (int k = 0; k < count; ++k)
    planes[(size_t)k] =
        f360::YuvPlanes{y_dev[k], u_dev[k], v_dev[k], y_linesize, u_linesize, v_linesize};
  return encode_sample_frames_impl(dec, targets_dev, sats_dev, nullptr, planes.data(), count,
                                   target_width, target_height, target_linesize, source_width,
                                   source_height, 0, centers_xy);
}

}  // extern "C"

namespace {
// shared by the RGB0 and the planar entry points; `yuv` non-null: pixels come from planes
int foveate_rect_impl(f360_sat_decoder *dec, uint8_t *target_dev, int target_width,
                      int target_height, int target_linesize, const uint8_t *source_dev,
                      int source_width, int source_height, int source_linesize,
                      float center_x, float center_y, const f360::YuvPlanes *yuv) {
  F360_REQUIRE(dec, "f360_satdec_foveate_rect: null decoder");
  F360_REQUIRE(target_dev && (source_dev || yuv), "f360_satdec_foveate_rect: null buffer");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2 && target_linesize >= 4 * target_width,
               "f360_satdec_foveate_rect: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_satdec_foveate_rect: a dimension exceeds 65536");
  F360_REQUIRE(std::fabs(center_x) <= 16.0f && std::fabs(center_y) <= 16.0f,
               "f360_satdec_foveate_rect: gaze centre out of range");
  if (!dec->gx_dev.p) {
    int st = f360_satdec_initialize_grid(dec, target_width, target_height, source_width,
                                         source_height);
    if (st != F360_OK) return st;
  }
  F360_REQUIRE(dec->gw == target_width && dec->gh == target_height,
               "f360_satdec_foveate_rect: grid was initialised for %dx%d", dec->gw, dec->gh);
  f360_ctx *ctx = dec->ctx;
  F360_BIND_DEVICE(ctx);
  // compact corner array: at most two distinct corners per reduced column / row
  const int cap_x = std::min(2 * target_width + 2, source_width);
  const int cap_y = std::min(2 * target_height + 2, source_height);
  F360_REQUIRE((size_t)cap_x * cap_y * 12 < ((size_t)1 << 32),
               "f360_satdec_foveate_rect: frame too large");
  const size_t n_maps = (size_t)source_width + source_height + 3 * (size_t)target_width +
                        3 * (size_t)target_height;
  if (dec->fov_maps.bytes < n_maps * sizeof(int) ||
      dec->fov_corners.bytes < (size_t)cap_x * cap_y * 12) {
    F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // earlier calls may still use the old ones
    int st = dec->fov_maps.reserve(n_maps * sizeof(int));
    if (st != F360_OK) return st;
    st = dec->fov_corners.reserve((size_t)cap_x * cap_y * 12);
    if (st != F360_OK) return st;
  }
  FovMaps m;
  m.gx = dec->gx_dev.as<int16_t>();
  m.gy = dec->gy_dev.as<int16_t>();
  m.cxp = (int)(center_x * (float)source_width);
  m.cyp = (int)(center_y * (float)source_height);
  m.src_w = source_width;
  m.src_h = source_height;
  m.out_w = target_width;
  m.out_h = target_height;
  int *w = dec->fov_maps.as<int>();
  m.xmap = w;  w += source_width;
  m.ymap = w;  w += source_height;
  m.ihx = w;   w += target_width;
  m.ilx = w;   w += target_width;
  m.dxw = w;   w += target_width;
  m.ihy = w;   w += target_height;
  m.ily = w;   w += target_height;
  m.dyw = w;
  const bool prof = f360::take_profile_slot(ctx);
  const bool piggyback = ctx->opt_fov_piggyback != 0;
  if (!piggyback) {
    f360::KernelSpan span(ctx, f360::kFovMaps, prof);
    hipLaunchKernelGGL(foveate_maps_kernel, dim3(2), dim3(f360::kFovThreads), 0, ctx->stream, m);
  }
  if (prof) ctx->prof_armed += 1;  // the encode below belongs to the same sampled call
  f360::SatEmit emit{m.xmap, m.ymap, dec->fov_corners.as<uint32_t>(), cap_x,
                     piggyback ? &m : nullptr};
  int st = f360::sat_encode_impl(ctx, nullptr, source_dev, source_width, source_height,
                                 source_linesize, &emit, yuv);
  if (st != F360_OK) return st;
  {
    f360::KernelSpan span(ctx, f360::kFovSample, prof);
    const int rows = ctx->opt_walk_rows;
    const dim3 grid((target_width + 4 * kWalkCols - 1) / (4 * kWalkCols),
                    (target_height + rows - 1) / rows);
    hipLaunchKernelGGL(sample_compact_walk_kernel, grid, dim3(256), 0, ctx->stream, target_dev,
                       target_width, target_height, target_linesize / 4,
                       dec->fov_corners.as<uint32_t>(), cap_x, m.ihx, m.ilx, m.dxw, m.ihy,
                       m.ily, m.dyw, rows);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
}  // namespace

extern "C" {

int f360_satdec_foveate_rect(f360_sat_decoder *dec, uint8_t *target_dev, int target_width,
                             int target_height, int target_linesize,
                             const uint8_t *source_dev, int source_width, int source_height,
                             int source_linesize, float center_x, float center_y) {
  return foveate_rect_impl(dec, target_dev, target_width, target_height, target_linesize,
                           source_dev, source_width, source_height, source_linesize, center_x,
                           center_y, nullptr);
}

int f360_satdec_foveate_rect_yuv420p(f360_sat_decoder *dec, uint8_t *target_dev,
                                     int target_width, int target_height, int target_linesize,
                                     const uint8_t *y_dev, const uint8_t *u_dev,
                                     const uint8_t *v_dev, int y_linesize, int u_linesize,
                                     int v_linesize, int source_width, int source_height,
                                     float center_x, float center_y) {
  const f360::YuvPlanes planes{y_dev, u_dev, v_dev, y_linesize, u_linesize, v_linesize};
  return foveate_rect_impl(dec, target_dev, target_width, target_height, target_linesize,
                           nullptr, source_width, source_height, 0, center_x, center_y,
                           &planes);
}

int f360_satdec_decode(f360_sat_decoder *dec, uint8_t *target_dev,
                       int target_linesize, const uint32_t *sat_dev, int width,
                       int height) {
  F360_REQUIRE(dec, "f360_satdec_decode: null decoder");
  F360_BIND_DEVICE(dec->ctx);
  F360_REQUIRE(target_dev && sat_dev, "f360_satdec_decode: null buffer");
  F360_REQUIRE(width >= 1 && height >= 1 && target_linesize / width >= 3,
               "f360_satdec_decode: bad geometry");
  const bool strip_path =
      width % 4 == 0 && target_linesize % 16 == 0 && target_linesize / width == 4 &&
      (size_t)target_linesize * height < ((size_t)1 << 32) &&
      (size_t)width * 12 * height < ((size_t)1 << 40) &&
      (reinterpret_cast<uintptr_t>(target_dev) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(sat_dev) & 15) == 0;
  if (strip_path) {
    const int strips = (width + 255) / 256;
    const int chunks = (height + kDecodeRows - 1) / kDecodeRows;
    const int waves = strips * chunks;
    hipLaunchKernelGGL(decode_strip_kernel, dim3((waves + 3) / 4), dim3(256), 0,
                       dec->ctx->stream, target_dev, target_linesize, sat_dev, width, height,
                       strips);
    F360_HIP_TRY(hipGetLastError());
    return F360_OK;
  }
  const dim3 grid((width + 63) / 64, (height + 3) / 4);
  hipLaunchKernelGGL(decode_kernel, grid, dim3(256), 0, dec->ctx->stream,
                     target_dev, target_linesize, target_linesize / width, sat_dev,
                     width, height);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

int f360_satdec_interpolate_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                                 int target_width, int target_height,
                                 int target_linesize, const uint8_t *source_dev,
                                 int source_width, int source_height,
                                 int source_linesize, float center_x,
                                 float center_y) {
  (void)target_linesize;  // the reference kernel never receives the linesizes
  (void)source_linesize;
  F360_REQUIRE(dec, "f360_satdec_interpolate_rect: null decoder");
  F360_REQUIRE(target_dev && source_dev, "f360_satdec_interpolate_rect: null buffer");
  F360_REQUIRE(target_width >= 2 && target_height >= 2 && source_width >= 1 &&
                   source_height >= 1,
               "f360_satdec_interpolate_rect: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_satdec_interpolate_rect: a dimension exceeds 65536");
  F360_REQUIRE(((uintptr_t)target_dev % 16) == 0 && ((uintptr_t)source_dev % 4) == 0,
               "f360_satdec_interpolate_rect: target must be 16-byte, source 4-byte aligned");
  F360_REQUIRE(std::fabs(center_x) <= 16.0f && std::fabs(center_y) <= 16.0f,
               "f360_satdec_interpolate_rect: gaze centre out of range");
  const int cxp = (int)(center_x * (float)target_width);
  const int cyp = (int)(center_y * (float)target_height);
  // largest |offset| any pixel can have after the single +-W wrap
  const int need_dx = imax(iabs(0 - cxp), iabs(target_width - 1 - cxp)) + 1;
  const int need_dy = imax(iabs(0 - cyp), iabs(target_height - 1 - cyp)) + 1;
  F360_BIND_DEVICE(dec->ctx);
  int st = ensure_interp_tables(dec, target_width, target_height, source_width,
                                source_height, need_dx, need_dy);
  if (st != F360_OK) return st;
  // rows per wave: enough to amortise the per-column set-up on large frames, few enough
  // that a small frame still yields thousands of waves (each row is a chain of dependent loads)
  const long px = (long)target_width * target_height;
  const int rows = dec->ctx->opt_interp_rows > 0 ? dec->ctx->opt_interp_rows
                   : px >= 16000000 ? 8 : px >= 6000000 ? 4 : 2;
  const dim3 grid((target_width + 64 * kInterpCols - 1) / (64 * kInterpCols),
                  (target_height + 4 * rows - 1) / (4 * rows));
  f360::KernelSpan span(dec->ctx, f360::kInterpolateRect,
                        f360::take_profile_slot(dec->ctx));
  if (dec->ctx->opt_interp_staged)
    hipLaunchKernelGGL(interpolate_rect_kernel<true>, grid, dim3(256), 0, dec->ctx->stream,
                       reinterpret_cast<uint32_t *>(target_dev), target_width,
                       target_height, reinterpret_cast<const uint32_t *>(source_dev),
                       source_width, source_height, dec->itx_dev.as<int4>(),
                       dec->it_dx, dec->ity_dev.as<int4>(), dec->it_dy, cxp, cyp, rows);
  else
    hipLaunchKernelGGL(interpolate_rect_kernel<false>, grid, dim3(256), 0, dec->ctx->stream,
                       reinterpret_cast<uint32_t *>(target_dev), target_width,
                       target_height, reinterpret_cast<const uint32_t *>(source_dev),
                       source_width, source_height, dec->itx_dev.as<int4>(),
                       dec->it_dx, dec->ity_dev.as<int4>(), dec->it_dy, cxp, cyp, rows);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // extern "C"
