// sat_encode.hip -- summed-area-table encode for gfx950 (MI355X).
//
// Replaces SATEncoder::EncodeFrameGPU (src/sat_encoder.cc:67-135) and its three
// OpenCL kernels copy_image / scan_rows / scan_columns
// (src/sat_encoder_encode_kernels.cl:1-20,44-58,60-74).  Output is bit-identical:
// uint32 addition is associative mod 2^32, so any summation order gives the
// reference's integers.
//
// Design (DESIGN.md "SAT encode"): reduce -> carry scan -> write.
//   A wave owns a strip of 256 pixels (64 lanes x 4 RGB0 pixels = one 16-byte
//   load per lane and row).  The frame is cut into bands of `band_rows` rows
//   and super-bands of `sb_bands` bands.
//   K1 sat_reduce : one wave per (strip, super-band) walks down the super-band
//                   and emits the column sums above each band (inside the
//                   super-band), the super-band column sums, every row's
//                   strip sum and every tile's sum.        reads 4 B/px
//   K2 sat_carry  : exclusive prefixes of those small arrays across
//                   super-bands / across strips.           ~1 % of the traffic
//   K3 sat_write  : one wave per (strip, band) re-reads its pixels, rebuilds
//                   the row prefix with a DPP wave scan, adds the carried-in
//                   column / row / corner sums and writes the final uint32x3
//                   table once.                     reads 4 B/px, writes 12 B/px
//   Total HBM traffic ~20.6 B/px against 16 B/px compulsory (the reference's
//   three passes move 64 B/px).
#include <algorithm>
#include <vector>

#include "f360_internal.h"

#include <algorithm>
#include <string>
#include <vector>
#include "fov_maps.h"
#include "host_tables.h"

namespace {

// Where K1 and K3 take their pixels from (template parameter SRC).  The planar sources
// convert in registers with libswscale's arithmetic (yuv_device.h), so the table equals the
// one of the RGB0 frame sws_scale would have produced, without that frame ever existing.
enum { kSrcBytes = 0, kSrcRgb0 = 1, kSrcYuvSwsC = 2, kSrcYuvSwsX86 = 3 };

constexpr int kLanePx = 4;                 // pixels per lane
constexpr int kStripPx = 64 * kLanePx;     // pixels per wave-row
constexpr int kWavesPerBlock = 4;
constexpr int kRowUnroll = 8;              // rows whose loads are issued together
#ifndef F360_REDUCE_DEPTH
#define F360_REDUCE_DEPTH 2
#endif
constexpr int kReduceDepth = F360_REDUCE_DEPTH;  // batches a reducer wave keeps in flight

// ---- wave64 DPP helpers ---------------------------------------------------
// dpp_ctrl: 0x110+n row_shr:n, 0x142 row_bcast:15, 0x143 row_bcast:31.
#define F360_DPP_ADD(v, ctrl, row_mask)                                        \
  (v) += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (row_mask), \
                                               0xf, false)

// Inclusive prefix sum over the 64 lanes of a wave (mod 2^32).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  F360_DPP_ADD(v, 0x111, 0xf);  // within each row of 16 lanes: Kogge-Stone
  F360_DPP_ADD(v, 0x112, 0xf);
  F360_DPP_ADD(v, 0x114, 0xf);
  F360_DPP_ADD(v, 0x118, 0xf);
  F360_DPP_ADD(v, 0x142, 0xa);  // rows 1,3 += last lane of rows 0,2
  F360_DPP_ADD(v, 0x143, 0xc);  // rows 2,3 += lane 31
  return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16 B
struct __attribute__((packed, aligned(4))) u32x3 {
  uint32_t x, y, z;
};

// LDS byte address in, 16 bytes per lane.  The reads carry their own wait
// (hipcc does not count memory operations issued from inline asm).
__device__ __forceinline__ void lds_write_b128(uint32_t addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read3_b128(uint32_t addr, u32x4 &a, u32x4 &b,
                                               u32x4 &c) {
  asm volatile(
      "ds_read_b128 %0, %3\n\t"
      "ds_read_b128 %1, %3 offset:1024\n\t"
      "ds_read_b128 %2, %3 offset:2048\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c)
      : "v"(addr)
      : "memory");
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write_b64(uint32_t addr, uint32_t a, uint32_t b) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(u32x2{a, b}) : "memory");
}
__device__ __forceinline__ uint32_t lds_read_b32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// A 16-byte global store the compiler does not see as a memory operation.  Loads and stores share
// one counter (vmcnt) on this hardware and complete out of order relative to each other, so
// with a store pending the compiler turns every wait for a load into vmcnt(0), i.e. into a wait
// for the acknowledgement of all earlier stores as well.  Hidden from its bookkeeping, a wait
// for a load is vmcnt(number of younger loads): still sufficient (loads complete in order; the
// extra pending stores can only make the wait longer, never shorter) and it no longer drains
// the stores.  The s_nop covers the store-data hazard (data > 8 bytes overwritten next).
__device__ __forceinline__ void global_store_b128_uncounted(uint32_t *p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

typedef uint32_t u32x3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void global_store_b96_uncounted(uint32_t *p, uint32_t x, uint32_t y,
                                                           uint32_t z) {
  asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(u32x3v{x, y, z})
               : "memory");
}

// The same, non-temporal: the table (354 MB at 8K) is written once and read back much later by
// another kernel, so its lines should not linger in L2 / Infinity Cache as dirty data -- their
// deferred write-back is otherwise paid by whichever kernel runs next (the next frame's reducer:
// 38-41 us in the pipeline against 26 us alone).  With `nt` the reducer takes 29 us, the writer
// itself 80 instead of 83 us and the whole path gains 8-12 %.  (Not for the fused path's corner
// array, which the compact sampler reads back at once: measured neutral to slightly worse.)
__device__ __forceinline__ void global_store_b128_uncounted_nt(uint32_t *p, u32x4 v) {
#ifdef F360_NO_NT_STORES
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
#ifndef F360_NT_BITS
#define F360_NT_BITS "sc0 sc1 nt"  // A/B: "nt" 174.3, "sc0 sc1 nt" 176.5, "sc0 sc1" alone 160.7 Gpix/s
#endif
  asm volatile("global_store_dwordx4 %0, %1, off " F360_NT_BITS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

constexpr int kEncBatch = 16;  // frames per batched launch (f360_sat_encode_batch)

struct EncodeArgs {
  uint32_t *sat;
  const uint8_t *src;
  int width, height, linesize, bpp;
  int band_rows, sb_bands, nstrips, nbands, nsb, wp3;
  uint32_t *lp, *sbtotal, *sbprefix, *rowsum, *rowcarry, *tiletotal, *tprefix;
  int ablate;  // timing experiments only (results are wrong when non-zero)
  // STORE == 2 (fused foveation): instead of the table, emit only the entries at the
  // lattice rows / columns a given gaze will sample
  const int *xmap, *ymap;  // source column / row -> compact index, -1 when unused
  uint32_t *corners;       // [compact row][corner_stride][3]
  int corner_stride;
  // SRC >= kSrcYuvSwsC: the three planes and the conversion constants
  f360::YuvPlanes yuv;
  f360::YuvConsts k;
  // fused foveation: workgroups past `reduce_blocks` of the reducer's grid compute the lattice
  // maps (one per axis) while the others reduce
  int reduce_blocks, has_maps;
  f360::FovMaps maps;
  // batched frames: blockIdx.y selects the frame -- its source, its table and its slice of the
  // scratch arrays (frame f's start f * ws_stride elements after frame 0's)
  int nbatch;
  size_t ws_stride;
  // read-once batched encoder (sat_walk_kernel): (frame, strip) units of the launch, 8-row
  // batches per strip, and the hand-off state (see WalkState)
  int walk_units, walk_nbatches;
  struct WalkState *walk;
  unsigned long long *walk_chain;  // [unit][batch][24] {tag:40 | row prefix:24}
  uint32_t *walk_err;              // host-visible: strips that gave up waiting and finished alone
  uint32_t walk_spin;              // polls a hand-off wait may take before the strip goes it alone
  int walk_mute;                   // test only ("debug.walk_mute"): this unit publishes nothing; -1
  unsigned long long *walk_stats;  // debug.ablate bit 8: per unit {start, end, slow polls, spins}
};
// (a kernel argument of its own: inside EncodeArgs the arrays keep the compiler from taking
// that struct apart, it lands in scratch memory and the row loops wait on vmcnt(0))
struct EncodeBatch {
  const uint8_t *src[kEncBatch];  // packed source, or the luma plane
  uint32_t *sat[kEncBatch];
  const uint8_t *u[kEncBatch], *v[kEncBatch];  // planar sources
};

// What a workgroup works on: the call's one frame, or frame blockIdx.y of a batch.
struct EncodeFrame {
  const uint8_t *src;
  uint32_t *sat;
  size_t ws;  // offset of the frame's scratch slice, in elements
  const uint8_t *y, *u, *v;  // planar sources (linesizes are the call's, in EncodeArgs::yuv)
};
// Constant indices and scalar selects: a dynamic index into a by-value argument makes the
// compiler copy it to scratch memory.
#define F360_ENCODE_FRAME(fr, a, b)                                          \
  EncodeFrame fr{a.src, a.sat, 0, a.yuv.y, a.yuv.u, a.yuv.v};                \
  if (a.nbatch != 0) {                                                       \
    const int f_ = (int)blockIdx.y;                                          \
    fr.src = b.src[0];                                                       \
    fr.sat = b.sat[0];                                                       \
    fr.u = b.u[0];                                                           \
    fr.v = b.v[0];                                                           \
    _Pragma("unroll") for (int k_ = 1; k_ < kEncBatch; ++k_) if (f_ == k_) { \
      fr.src = b.src[k_];                                                    \
      fr.sat = b.sat[k_];                                                    \
      fr.u = b.u[k_];                                                        \
      fr.v = b.v[k_];                                                        \
    }                                                                        \
    fr.y = fr.src;                                                           \
    fr.ws = (size_t)f_ * a.ws_stride;                                        \
  }

// kRowUnroll rows of a lane's four pixels as loaded; planar sources convert at use, so that
// the loads of a whole batch stay in flight.
template <bool YUV>
struct RowBatchT {
  uint4 raw[kRowUnroll];  // packed R | G<<8 | B<<16 dwords
};
template <>
struct RowBatchT<true> {
  uint32_t y4[kRowUnroll];      // four luma bytes per row
  uint32_t uv[kRowUnroll / 2];  // per row pair: U0 | U1<<8 | V0<<16 | V1<<24
};
template <int SRC>
using RowBatch = RowBatchT<(SRC >= kSrcYuvSwsC)>;

template <int SRC>
__device__ __forceinline__ void batch_pixels(const EncodeArgs &a, const RowBatch<SRC> &b,
                                             int r, uint32_t (&v)[4]) {
  if constexpr (SRC >= kSrcYuvSwsC) {
    f360::yuv_pixels4<SRC - kSrcYuvSwsC>(a.k, b.y4[r], b.uv[r >> 1], v);
  } else {
    v[0] = b.raw[r].x;
    v[1] = b.raw[r].y;
    v[2] = b.raw[r].z;
    v[3] = b.raw[r].w;
  }
}

// rows [y, y + kRowUnroll) of a planar source, y a multiple of kRowUnroll; rows past y_last
// (odd: the last row of the frame or of the caller's run of rows) re-read that row -- what lies
// past it is masked or never stored by the callers
template <int SRC>
__device__ __forceinline__ void load_yuv_batch(const EncodeArgs &a, const EncodeFrame &fr,
                                               RowBatch<SRC> &b, int y, int x0, int y_last) {
  if constexpr (SRC >= kSrcYuvSwsC) {
    const int xc = min(x0, a.width - kLanePx);
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.y4[r] = *reinterpret_cast<const uint32_t *>(
          fr.y + (size_t)min(y + r, y_last) * a.yuv.y_linesize + xc);
#pragma unroll
    for (int r = 0; r < kRowUnroll / 2; ++r) {
      const size_t crow = (size_t)min((y >> 1) + r, y_last >> 1);
      const uint32_t u =
          *reinterpret_cast<const uint16_t *>(fr.u + crow * a.yuv.u_linesize + (xc >> 1));
      const uint32_t v =
          *reinterpret_cast<const uint16_t *>(fr.v + crow * a.yuv.v_linesize + (xc >> 1));
      b.uv[r] = u | (v << 16);
    }
  }
}

// Four pixels of one row as packed R | G<<8 | B<<16 dwords (0 beyond the row).
template <int SRC>
__device__ __forceinline__ uint4 load_px4(const uint8_t *src, int width, int y,
                                          int x0, int linesize, int bpp) {
  if (SRC == kSrcRgb0) {
    if (x0 < width)
      return *reinterpret_cast<const uint4 *>(src + (size_t)y * linesize +
                                              (size_t)x0 * 4);
    return make_uint4(0, 0, 0, 0);
  }
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = 0;
    if (x0 + k < width) {
      const uint8_t *p = src + (size_t)y * linesize + (size_t)(x0 + k) * bpp;
      v[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
  }
  return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void unpack_px4(const uint4 &raw, uint32_t (&c)[12]) {
  const uint32_t v[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[3 * k + 0] = v[k] & 0xffu;
    c[3 * k + 1] = (v[k] >> 8) & 0xffu;
    c[3 * k + 2] = (v[k] >> 16) & 0xffu;
  }
}

__device__ __forceinline__ void store12(uint32_t *dst, const uint32_t (&a)[12]) {
  uint4 *d = reinterpret_cast<uint4 *>(dst);
  d[0] = make_uint4(a[0], a[1], a[2], a[3]);
  d[1] = make_uint4(a[4], a[5], a[6], a[7]);
  d[2] = make_uint4(a[8], a[9], a[10], a[11]);
}

__device__ __forceinline__ void load12(const uint32_t *src, uint32_t (&a)[12]) {
  const uint4 *s = reinterpret_cast<const uint4 *>(src);
  const uint4 q0 = s[0], q1 = s[1], q2 = s[2];
  a[0] = q0.x; a[1] = q0.y; a[2] = q0.z; a[3] = q0.w;
  a[4] = q1.x; a[5] = q1.y; a[6] = q1.z; a[7] = q1.w;
  a[8] = q2.x; a[9] = q2.y; a[10] = q2.z; a[11] = q2.w;
}

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

template <int SRC>
__device__ __forceinline__ void reduce_load_batch(const EncodeArgs &a, const EncodeFrame &fr,
                                                  RowBatch<SRC> &b, int y, int x0, int y_last) {
  if constexpr (SRC >= kSrcYuvSwsC) {
    load_yuv_batch<SRC>(a, fr, b, y, x0, y_last);
  } else if constexpr (SRC == kSrcRgb0) {
    // branch-free: rows past the wave's last row re-read that row (a cache hit), validity is
    // applied by the caller's masks
    const int xc = min(x0, a.width - kLanePx);
    const uint8_t *p = fr.src + (size_t)xc * 4;
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.raw[r] = *reinterpret_cast<const uint4 *>(p + (size_t)min(y + r, y_last) * a.linesize);
  } else {
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.raw[r] = (y + r < a.height)
                     ? load_px4<kSrcBytes>(fr.src, a.width, y + r, x0, a.linesize, a.bpp)
                     : make_uint4(0, 0, 0, 0);
  }
}

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

// ---- K2: exclusive prefixes of the carry arrays ------------------------------
// out[k][i] = sum_{k' < k} in[k'][i]   for i < n, k < K
struct ScanSeg {
  const uint32_t *in;
  uint32_t *out;
  int n, K, nblocks;
};
// (a batched launch: blockIdx.y = frame, whose arrays start `frame_stride` elements apart)

// One round of at most 32 loads per thread: a segment with more than 32 rows is split into
// `parts` row ranges handled by different threads of the workgroup (the workgroup then covers
// 256 / parts columns); a part's base is the sum of the parts before it, passed through LDS.
// (Two rounds in one thread: 8.8 us for the 60 super-bands of an 8K frame; 64 loads in flight: 11.4.)
__device__ __forceinline__ void carry_scan_segment(const ScanSeg &s, int blk, uint32_t *totals) {
  if (s.K > 128) {  // very tall / wide frames: rounds of 32 in one thread
    const int i = blk * 256 + (int)threadIdx.x;
    if (i >= s.n) return;
    uint32_t run = 0;
    for (int k = 0; k < s.K; k += 32) {
      uint32_t t[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) t[q] = (k + q < s.K) ? s.in[(size_t)(k + q) * s.n + i] : 0u;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        if (k + q < s.K) s.out[(size_t)(k + q) * s.n + i] = run;
        run += t[q];
      }
    }
    return;
  }
  const int parts = (s.K + 31) / 32;           // 1, 2 (8K), 3 or 4
  const int cols = 256 / parts;                // columns per workgroup
  const int part = (int)threadIdx.x / cols, c = (int)threadIdx.x - part * cols;
  const int i = blk * cols + c;
  const int rows = (s.K + parts - 1) / parts;  // rows per part, <= 32
  const int k0 = part * rows, k1 = min(k0 + rows, s.K);
  const bool live = part < parts && i < s.n;
  uint32_t t[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) t[q] = (live && k0 + q < k1) ? s.in[(size_t)(k0 + q) * s.n + i] : 0u;
  uint32_t base = 0;
  if (parts > 1) {  // wave-uniform
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) sum += t[q];
    totals[threadIdx.x] = sum;
    __syncthreads();
    for (int p = 0; p < part; ++p) base += totals[p * cols + c];
  }
  if (!live) return;
  uint32_t run = base;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    if (k0 + q < k1) s.out[(size_t)(k0 + q) * s.n + i] = run;
    run += t[q];
  }
}

__global__ __launch_bounds__(256) void sat_carry_kernel(ScanSeg a, ScanSeg b, ScanSeg c,
                                                        size_t frame_stride) {
  __shared__ uint32_t totals[256];
  const size_t shift = (size_t)blockIdx.y * frame_stride;
  a.in += shift; a.out += shift;
  b.in += shift; b.out += shift;
  c.in += shift; c.out += shift;
  int blk = blockIdx.x;
  if (blk < a.nblocks) {
    carry_scan_segment(a, blk, totals);
    return;
  }
  blk -= a.nblocks;
  if (blk < b.nblocks) {
    carry_scan_segment(b, blk, totals);
    return;
  }
  blk -= b.nblocks;
  carry_scan_segment(c, blk, totals);
}

// ---- K3: final table ----------------------------------------------------------
// STORE 0: three 16-byte stores per lane at a 48-byte lane stride.
// STORE 1: re-stage the row through wave-private LDS so that each store
//          instruction writes 1 KiB contiguous.
template <int SRC, int STORE>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sat_write_kernel(
    const EncodeArgs a, const EncodeBatch eb) {
  constexpr bool VEC = SRC != kSrcBytes;  // 16-byte accesses allowed (width % 4 == 0, aligned)
  __shared__ __attribute__((aligned(16))) uint32_t
      stage[STORE >= 1 ? kWavesPerBlock * 3 * kStripPx : 4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kWavesPerBlock + wave);
  if (tile >= a.nstrips * a.nbands) return;  // 1-D grid over the tiles, see sat_reduce_kernel
  F360_ENCODE_FRAME(fr, a, eb)
  const int band = tile / a.nstrips;
  const int strip = tile - band * a.nstrips;
  const int x0 = strip * kStripPx + lane * kLanePx;
  const int sb = band / a.sb_bands;
  int xm[4] = {-1, -1, -1, -1};
  bool dense = false;  // every pixel of the wave is a lattice column, ranks consecutive
  int xm_first = 0;
  if (STORE == 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (x0 + k < a.width) xm[k] = a.xmap[x0 + k];
    xm_first = __builtin_amdgcn_readfirstlane(xm[0]);
    dense = __all(xm[0] == xm_first + 4 * lane && xm[1] == xm[0] + 1 && xm[2] == xm[0] + 2 &&
                  xm[3] == xm[0] + 3 && xm_first >= 0);
  }

  const int y_end = min((band + 1) * a.band_rows, a.height);
  const uint32_t *rc = a.rowcarry + fr.ws + (size_t)strip * a.height * 3;
  // A batch = kRowUnroll rows of pixels plus their row carries (3 dwords per row, fetched by
  // lanes 0..23 in one load and broadcast with v_readlane: a per-row load of a wave-uniform
  // address would be one more vector-memory operation to wait for in every row).  No branch
  // around any load (addresses are clamped to the band's last row): see sat_reduce_kernel.
  auto load_batch = [&](RowBatch<SRC> &raw, uint32_t &carry, int y) {
    if constexpr (SRC == kSrcBytes) {
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r)
        raw.raw[r] = (y + r < y_end)
                         ? load_px4<SRC>(fr.src, a.width, y + r, x0, a.linesize, a.bpp)
                         : make_uint4(0, 0, 0, 0);
    } else {
      reduce_load_batch<SRC>(a, fr, raw, y, x0, y_end - 1);
    }
    const uint32_t *cp = rc + min(y * 3 + min(lane, 3 * kRowUnroll - 1), a.height * 3 - 1);
    if (STORE == 2 && lane >= 32)  // emit mode: lanes 32..39 fetch the rows' compact indices
      cp = reinterpret_cast<const uint32_t *>(a.ymap) +
           min(y + min(lane - 32, kRowUnroll - 1), a.height - 1);
    carry = *cp;
  };
  // The band's first two batches of pixels are requested BEFORE the prologue's own loads (the
  // carried-in column sums, the corner look-back): those used to be three memory round trips in
  // a row before the first pixel load was even issued; now everything is in flight together.
  RowBatch<SRC> buf_a, buf_b;
  uint32_t carry_a, carry_b;
  const int y_begin = band * a.band_rows;
  load_batch(buf_a, carry_a, y_begin);
  load_batch(buf_b, carry_b, y_begin + kRowUnroll);

  // --- table row just above the band, for this lane's 4 pixels -------------
  uint32_t acc[12];
  {
    uint32_t t0[12], t1[12];
    load12(a.sbprefix + fr.ws + (size_t)sb * a.wp3 + (size_t)x0 * 3, t1);
    if (a.sb_bands == 1) {  // a band is its own super-band: nothing above it inside
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t1[e];
    } else {
      load12(a.lp + fr.ws + (size_t)band * a.wp3 + (size_t)x0 * 3, t0);
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t0[e] + t1[e];
    }
  }
  // corner: every tile above and to the left
  uint32_t corner[3] = {0, 0, 0};
  for (int b = lane; b < band; b += 64) {
    const uint32_t *tp = a.tprefix + fr.ws + ((size_t)strip * a.nbands + b) * 3;
    corner[0] += tp[0];
    corner[1] += tp[1];
    corner[2] += tp[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    corner[c] = (uint32_t)__builtin_amdgcn_readlane(
        (int)wave_scan_incl(corner[c]), 63);
    // prefix along x of the column sums: inside the lane, then across lanes
    acc[3 + c] += acc[c];
    acc[6 + c] += acc[3 + c];
    acc[9 + c] += acc[6 + c];
    const uint32_t excl = wave_scan_incl(acc[9 + c]) - acc[9 + c] + corner[c];
    acc[c] += excl;
    acc[3 + c] += excl;
    acc[6 + c] += excl;
    acc[9 + c] += excl;
  }

  auto write_batch = [&](const RowBatch<SRC> &raw, uint32_t carry, int y) {
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) {
      if (y + r >= y_end) break;
      uint32_t c[12];
      {
        uint32_t v[4];
        batch_pixels<SRC>(a, raw, r, v);
        unpack_px4(make_uint4(v[0], v[1], v[2], v[3]), c);
      }
      // inclusive prefix over the lane's 4 pixels
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        c[3 * k + 0] += c[3 * k - 3];
        c[3 * k + 1] += c[3 * k - 2];
        c[3 * k + 2] += c[3 * k - 1];
      }
      const uint32_t inc_rg = wave_scan_incl(c[9] | (c[10] << 16));
      const uint32_t inc_b = wave_scan_incl(c[11]);
      const uint32_t base_r =
          (inc_rg & 0xffffu) - c[9] + (uint32_t)__builtin_amdgcn_readlane((int)carry, 3 * r);
      const uint32_t base_g =
          (inc_rg >> 16) - c[10] + (uint32_t)__builtin_amdgcn_readlane((int)carry, 3 * r + 1);
      const uint32_t base_b =
          inc_b - c[11] + (uint32_t)__builtin_amdgcn_readlane((int)carry, 3 * r + 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[3 * k + 0] += c[3 * k + 0] + base_r;
        acc[3 * k + 1] += c[3 * k + 1] + base_g;
        acc[3 * k + 2] += c[3 * k + 2] + base_b;
      }
      if (STORE == 2) {
        const int py = __builtin_amdgcn_readlane((int)carry, 32 + r);  // a.ymap[y + r]
        if (py >= 0) {
          uint32_t *crow = a.corners + (size_t)py * a.corner_stride * 3;
          if (dense) {
            // the wave's 768 dwords are contiguous in the compact row: re-stage through LDS as
            // the table writer does and store 1 KiB per instruction (4-byte aligned only)
            const uint32_t mine =
                (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * 3 * kStripPx * 4;
            lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
            lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
            lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
            u32x4 v[3];
            lds_read3_b128(mine + lane * 16, v[0], v[1], v[2]);
            uint32_t *dst = crow + (size_t)xm_first * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q)
              global_store_b128_uncounted(dst + q * 256 + lane * 4, v[q]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (xm[k] >= 0) {  // one 12-byte store per lattice texel
                global_store_b96_uncounted(crow + (size_t)xm[k] * 3, acc[3 * k + 0],
                                           acc[3 * k + 1], acc[3 * k + 2]);
              }
          }
        }
        continue;
      }
      uint32_t *row = fr.sat + (size_t)(y + r) * a.width * 3;
      if (VEC && STORE == 0) {
        if (x0 < a.width) store12(row + (size_t)x0 * 3, acc);
      } else if (VEC && STORE == 1) {
        // Lanes exchange data through wave-private LDS.  One wave's LDS
        // operations execute in order, so no s_barrier is needed; the accesses
        // are inline asm because the compiler's memory model is per lane (it
        // deletes plain LDS stores that only OTHER lanes read back).
        const uint32_t mine =
            (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * 3 * kStripPx * 4;
        lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
        lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
        lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
        u32x4 v[3];
        lds_read3_b128(mine + lane * 16, v[0], v[1], v[2]);
        const int row_dwords = a.width * 3;
        const int base = strip * kStripPx * 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int off = q * 256 + lane * 4;
          if (base + off < row_dwords)  // width % 4 == 0 -> whole 16 B in range
            global_store_b128_uncounted_nt(row + base + off, v[q]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (x0 + k < a.width) {
            row[(size_t)(x0 + k) * 3 + 0] = acc[3 * k + 0];
            row[(size_t)(x0 + k) * 3 + 1] = acc[3 * k + 1];
            row[(size_t)(x0 + k) * 3 + 2] = acc[3 * k + 2];
          }
      }
    }
  };
  // two batches alternate (a band is 2, 4 or 8 batches): the next one is in flight while this
  // one is scanned and stored
  for (int y = y_begin; y < y_end; y += 2 * kRowUnroll) {
    write_batch(buf_a, carry_a, y);
    load_batch(buf_a, carry_a, y + 2 * kRowUnroll);
    if (y + kRowUnroll < y_end) write_batch(buf_b, carry_b, y + kRowUnroll);
    load_batch(buf_b, carry_b, y + 3 * kRowUnroll);
  }
}

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

// ---- encode + sample in one pass (f360_satdec_encode_sample_frames) ---------------------------
// The gaze of every frame is known before its table is built (the server loop receives it before
// it encodes, src/video_server.cc:287-345), and a strip owner has, at every table row, the whole
// row of its strip in registers.  A reduced pixel is a box of the table,
//   (S[hi_y][hi_x] - S[lo_y][hi_x]) - (S[hi_y][lo_x] - S[lo_y][lo_x])) / area
// (src/sat_decoder_sample_rect_kernel.cl:206-217), so a strip that keeps a copy of its table row
// at row lo_y ("snapshot") can, at row hi_y, form D = row - snapshot and emit every reduced pixel
// whose two columns lie inside the strip: (D[hi_x] - D[lo_x]) / area.  The table is written
// exactly as before; what disappears is the sampler's pass over it (197 of its 233 MB per 8K
// frame are table rows read back).  Which rows snapshot and which emit is a per-frame row plan
// (walk_fuse_plan_kernel).  The gathering is not done by the strip owners -- they are the serial
// chain of the launch -- but by a helper wave per owner that takes the D rows from LDS
// (walk_fuse_helper); which pixels a strip owns the helper works out when it starts.  What this
// leaves out is finished by walk_fuse_fix_kernel: boxes that straddle two strips (three per strip
// boundary at most), from the halves the two strips' helpers export, and the reduced rows whose
// boxes overlap their neighbours' at the frame's top and bottom edge (none or one per frame),
// from the finished table.  Sources: RGB0 (source-pixel rows for the fovea, snapshot in
// registers) and planar YUV 4:2:0 (snapshot in LDS).  Null table pointers: the same launch
// without the table (f360_satdec_foveate_rect_frames).
constexpr uint32_t kFuseEmit = 1u << 31;  // row plan: this table row is the lower edge of a
                                          // reduced row (bits 0-15: which, bits 16-25: box height)
constexpr uint32_t kFuseSnap = 1u << 30;  // row plan: snapshot this table row (after emitting)
constexpr int kFuseOwners = kWalkWaves;  // strip owners per workgroup of the one-pass kernel, + as many helpers
constexpr int kFuseEntries = 3 * kStripPx;  // reduced pixels a strip can own: <= 256 per wrap class
constexpr int kFuseRawRows = kFuseEntries + kRowUnroll * 3 * kStripPx;  // (offset of the pixel rows)
constexpr int kFuseWaveDwords = kFuseRawRows + kRowUnroll * kStripPx;  // + a D row and a pixel row per batch row

struct WalkFuse {
  uint8_t *dst[kWalkFrames];
  int cxp[kWalkFrames], cyp[kWalkFrames];
  const int16_t *gx, *gy;
  uint32_t *rowplan;  // [frame][plan_stride]
  int plan_stride, out_w, out_h, dst_linesize;
  // boxes that straddle two strips: the list per frame ({count, then i, hi, lo per pixel}) and,
  // per emitted reduced row, the D values of their two columns ([row][pixel][hi | lo][3])
  uint32_t *spix, *side;
  int pmax;
  size_t side_stride;  // dwords per frame
};
constexpr int kFixCols = 256;  // straddling pixels of a frame: <= 3 per strip boundary
constexpr int kSpixWords = 1024;  // 1 + 3 * kFixCols, rounded up; the tail: reduced rows the walk
constexpr int kSpixLrows = 800;   // cannot emit ({count, rows}), at most kFixLrows listed
constexpr int kFixLrows = 16;
struct WalkNoFuse {};
template <bool FUSE> struct WalkFuseArg { typedef WalkNoFuse type; };
template <> struct WalkFuseArg<true> { typedef WalkFuse type; };

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// Three bytes of a reduced pixel (the fourth is not ours, sat_decoder_sample_rect_kernel.cl:212),
// hidden from the compiler's vmcnt bookkeeping like the table stores.
__device__ __forceinline__ void fuse_store_rgb(uint8_t *row, uint32_t off, uint32_t rg,
                                               uint32_t b) {
  asm volatile(
      "global_store_short %0, %1, %3 nt\n\t"
      "global_store_byte %0, %2, %3 offset:2 nt" ::"v"(off), "v"(rg), "v"(b), "s"(row)
      : "memory");
}

// Three consecutive dwords of LDS at a dword-aligned byte address, no wait: the caller issues a
// row's worth of these and waits once.
__device__ __forceinline__ void fuse_lds_read3(uint32_t addr, u32x2 &a01, uint32_t &a2) {
  asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8"
               : "=&v"(a01), "=&v"(a2)
               : "v"(addr)
               : "memory");
}

// The three quotients of one box, exact; operands of 2^22 and more (boxes of > 16k pixels) take
// the integer division inline -- a call would cost the walker its register allocation.
__device__ __forceinline__ uint3 fuse_div3(uint3 n, uint32_t d) {
  if (((n.x | n.y | n.z | d) >> 22) != 0) return make_uint3(n.x / d, n.y / d, n.z / d);
  const float inv = __builtin_amdgcn_rcpf((float)d);
  return make_uint3(f360::udiv_by_rcp(n.x, inv, d), f360::udiv_by_rcp(n.y, inv, d),
                    f360::udiv_by_rcp(n.z, inv, d));
}

// The helper wave of strip owner `unit`: the reduced pixels whose box lies inside the strip --
// {hi column : 8 | lo column : 8 | reduced column : 16}, worked out once -- and then, batch by
// batch, for every row the plan marks EMIT: wait for the owner's D row in slot r (mailbox word r
// = the plan word), one pixel per lane and round, hand the slot back.  The helper waits for
// nothing but its owner, and the owner only ever waits for a slot of the batch before.
//
// A row must take the helper less than the owner takes over its own (~0.45 us), so the first
// kFuseRounds * 64 pixels of the strip (all of them, outside pathological geometries) live in
// registers as LDS offsets, and a row's gathers are all issued before the first is used: one
// LDS round trip per row, not two per round.  Quotients: n <= 255 * area and q = n / area <= 255,
// so for area <= 2048 the float product n * (1/width) * (1/height), biased by 2^-12, truncates
// to q exactly -- every rounding together moves it by < 2^-13, a true fraction is at least
// 1/2048 below the next integer (tests/test_fuse_div.py walks every case); larger boxes divide.
constexpr int kFuseRounds = 5;

template <int NR, bool PIX>
__device__ __forceinline__ void walk_fuse_rows(const EncodeArgs &a, const WalkFuse &wf,
                                               const uint32_t *plan, uint8_t *dst, int lane,
                                               const uint32_t *ent, int n_ent,
                                               const uint32_t *drows, uint32_t mbox,
                                               uint32_t max_dxw, const int (&xcol)[3],
                                               const int (&xslot)[3], int npix, uint32_t *side,
                                               int unit) {
  const bool exports_k[3] = {(bool)__any(xslot[0] >= 0), (bool)__any(xslot[1] >= 0),
                             (bool)__any(xslot[2] >= 0)};
  const bool exports = exports_k[0] || exports_k[1] || exports_k[2];
  // this lane's pixel of round k: D-row byte offsets of its two columns, box width, target
  uint32_t eoff[NR], estore[NR];
  float einv[NR];
  bool valid[NR], unit_wide[NR];  // (unit_wide: every pixel of the round is one column wide)
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int e = lane + 64 * k;
    valid[k] = e < n_ent;
    const uint32_t en = valid[k] ? ent[e] : 0x00000100u;  // (hi 0, lo 1: width taken as 1 below)
    const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
    const uint32_t dxw = valid[k] ? hi - lo : 1u;
    eoff[k] = (hi * 12u) | ((lo * 12u) << 12) | (dxw << 24);
    estore[k] = ((en >> 16) * 4u) | (hi << 22);  // (reduced column * 4 < 2^18; the hi column again)
    einv[k] = __builtin_amdgcn_rcpf((float)dxw);
    unit_wide[k] = __all(dxw == 1u);
  }
  const bool timed = a.ablate & 256;
  unsigned long long wait_cycles = 0, work_cycles = 0, rows_done = 0;
  for (int t = 0; t < a.walk_nbatches; ++t) {
    // The plan words of the batch by a scalar load with its own wait: nothing in this loop may
    // make the compiler wait on the vector memory counter -- the pixel stores are hidden from
    // it, and a vmcnt(0) for a plan word waits for the stores of the row before as well (that
    // was 800 of a row's 1600 cycles).
    u32x8 pw;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(pw)
                 : "s"(plan + (size_t)t * kRowUnroll)
                 : "memory");
    uint32_t pv = 0;
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) pv = lane == r ? pw[r] : pv;
    for (int r = 0; r < kRowUnroll; ++r) {
      const uint32_t pr = (uint32_t)__builtin_amdgcn_readlane((int)pv, r);
      if (!(pr & kFuseEmit)) continue;
      const unsigned long long c0 = timed ? __builtin_amdgcn_s_memtime() : 0;
      // (two quick looks, then long naps: a helper of a sparse strip waits most of the time,
      // and every look costs its SIMD -- its owner's SIMD -- issue slots and an LDS access)
      for (int looks = 0; lds_read_b32(mbox + r * 4) != pr; ++looks) {
        if (looks < 2)
          __builtin_amdgcn_s_sleep(1);
        else
          __builtin_amdgcn_s_sleep(8);
      }
      const unsigned long long c1 = timed ? __builtin_amdgcn_s_memtime() : 0;
      const uint32_t dy = (pr >> 16) & 0x3ffu;
      uint8_t *orow = dst + (size_t)(pr & 0xffffu) * wf.dst_linesize;
      const uint32_t *d = drows + r * (3 * kStripPx);
      // every gather of the row -- two columns per pixel, the columns to export -- issued
      // before anything waits: left to the compiler the rounds wait one after the other, and a
      // row costs five LDS round trips instead of one (1600 against 1100 cycles for 267 boxes)
      const uint32_t dlds = (uint32_t)reinterpret_cast<uintptr_t>(d);
      u32x2 h01[NR], l01[NR], x01[3] = {};
      uint32_t h2[NR], l2[NR], x2[3] = {};
      const bool one_row = dy == 1u;
      const uint32_t plds = (uint32_t)reinterpret_cast<uintptr_t>(
          drows + kRowUnroll * 3 * kStripPx + r * kStripPx);  // the source pixels of the row
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if (PIX && one_row && unit_wide[k]) {  // the fovea: a reduced pixel IS a source pixel
          asm volatile("ds_read_b32 %0, %1" : "=v"(h2[k]) : "v"(plds + (estore[k] >> 22) * 4u) : "memory");
        } else {
          fuse_lds_read3(dlds + (eoff[k] & 0xfffu), h01[k], h2[k]);
          fuse_lds_read3(dlds + ((eoff[k] >> 12) & 0xfffu), l01[k], l2[k]);
        }
      }
      if (exports) {
#pragma unroll
        for (int k = 0; k < 3; ++k)  // (every lane reads -- column 0 if it has nothing to export)
          if (exports_k[k]) fuse_lds_read3(dlds + (uint32_t)xcol[k] * 12u, x01[k], x2[k]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NR; ++k)
        asm volatile("" : "+v"(h01[k]), "+v"(h2[k]), "+v"(l01[k]), "+v"(l2[k]));
#pragma unroll
      for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(x01[k]), "+v"(x2[k]));
      {
        uint3 n[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k)
          n[k] = make_uint3(h01[k].x - l01[k].x, h01[k].y - l01[k].y, h2[k] - l2[k]);
        if (dy * max_dxw <= 2048u) {
          const float inv_dy = __builtin_amdgcn_rcpf((float)dy);
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            if (one_row && unit_wide[k]) {
              if (PIX) {  // the posted source pixel: R, G from the low half, B from byte 2
                if (valid[k])
                  asm volatile(
                      "global_store_short %0, %1, %2 nt\n\t"
                      "global_store_byte_d16_hi %0, %1, %2 offset:2 nt" ::"v"(estore[k] & 0x3fffffu),
                      "v"(h2[k]), "s"(orow)
                      : "memory");
              } else if (valid[k]) {  // (planar sources post no pixel rows) a box of one: n itself
                fuse_store_rgb(orow, estore[k] & 0x3fffffu, n[k].x | (n[k].y << 8), n[k].z);
              }
              continue;
            }
            const float inv = einv[k] * inv_dy;
            const uint32_t qx = (uint32_t)__builtin_fmaf((float)n[k].x, inv, 0x1p-12f);
            const uint32_t qy = (uint32_t)__builtin_fmaf((float)n[k].y, inv, 0x1p-12f);
            const uint32_t qz = (uint32_t)__builtin_fmaf((float)n[k].z, inv, 0x1p-12f);
            if (valid[k]) fuse_store_rgb(orow, estore[k] & 0x3fffffu, qx | (qy << 8), qz);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const uint3 q = fuse_div3(n[k], (eoff[k] >> 24) * dy);
            if (valid[k]) fuse_store_rgb(orow, estore[k] & 0x3fffffu, (q.x & 0xffu) | ((q.y & 0xffu) << 8), q.z);
          }
        }
        for (int e = lane + 64 * NR; e < n_ent; e += 64) {  // (NR == kFuseRounds only)
          const uint32_t en = ent[e];
          const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
          const uint32_t *ph = d + hi * 3, *pl = d + lo * 3;
          const uint3 q = fuse_div3(make_uint3(ph[0] - pl[0], ph[1] - pl[1], ph[2] - pl[2]),
                                    (hi - lo) * dy);
          fuse_store_rgb(orow, (en >> 16) * 4, (q.x & 0xffu) | ((q.y & 0xffu) << 8), q.z);
        }
      }
      if (exports) {  // this strip's columns of the boxes that straddle two strips
        uint32_t *srow = side + (size_t)(pr & 0xffffu) * npix * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (xslot[k] >= 0)
            asm volatile("global_store_dwordx3 %0, %1, %2" ::"v"((uint32_t)xslot[k] * 12u),
                         "v"(u32x3v{x01[k].x, x01[k].y, x2[k]}), "s"(srow)
                         : "memory");
      }
      // (the stores above took their data from the D row: every read of it has returned)
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(mbox + r * 4), "v"(0u)
                   : "memory");
      if (timed) {
        wait_cycles += c1 - c0;
        work_cycles += __builtin_amdgcn_s_memtime() - c1;
        ++rows_done;
      }
    }
  }
  if (timed && lane == 0) {  // debug.ablate bit 8: the helper's half of the unit's statistics
    ulonglong2 *st = reinterpret_cast<ulonglong2 *>(a.walk_stats + (size_t)unit * 8 + 4);
    st[0] = make_ulonglong2(wait_cycles, work_cycles);
    st[1] = make_ulonglong2(rows_done, (unsigned long long)n_ent);
  }
}

template <bool PIX>
__device__ __forceinline__ void walk_fuse_helper(const EncodeArgs &a, const WalkFuse &wf,
                                                 int unit, int lane, uint32_t *ent,
                                                 uint32_t *box) {
  const int f = unit / a.nstrips;
  const int strip = unit - f * a.nstrips;
  const uint32_t *drows = ent + kFuseEntries;
  const uint32_t mbox = (uint32_t)reinterpret_cast<uintptr_t>(box);
  const uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  uint8_t *dst = wf.dst[f];
  const int cxp = wf.cxp[f];
  // (the boxes one column wide first, then the others, each group by reduced column: whole
  // rounds of the former take the fovea's short cut, and a round's stores stay consecutive)
  int n_ent = 0;
  uint32_t max_dxw = 1;
  for (int pass = 0; pass < 2; ++pass)
    for (int i0 = 0; i0 < wf.out_w; i0 += 64) {
      const int i = i0 + lane, ic = min(i, wf.out_w - 1);
      const f360::AxisBox bx = f360::sample_axis(cxp, wf.gx[ic + 1], wf.gx[ic], a.width, true);
      const bool own = i < wf.out_w && bx.ok && (bx.hi >> 8) == strip && (bx.lo >> 8) == strip &&
                       (bx.hi - bx.lo == 1) == (pass == 0);
      const unsigned long long m = __ballot(own);
      if (own) {
        ent[n_ent + __popcll(m & ((1ull << lane) - 1))] =
            (uint32_t)(bx.hi & 255) | ((uint32_t)(bx.lo & 255) << 8) | ((uint32_t)i << 16);
        max_dxw = max(max_dxw, (uint32_t)(bx.hi - bx.lo));
      }
      n_ent += __popcll(m);
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) max_dxw = max(max_dxw, (uint32_t)__shfl_xor((int)max_dxw, off, 64));
  max_dxw = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_dxw);
  // the straddling boxes with a column in this strip: lane, round -> {column, slot in the row}
  const uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  int npix = (int)sp[0];
  if (npix > wf.pmax) npix = 0;  // (more than the side rows hold: the fix-up takes every row)
  int xcol[3], xslot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = lane + 64 * k;
    xcol[k] = 0;
    xslot[k] = -1;
    if (q < npix) {
      const int hi = (int)sp[2 + 3 * q], lo = (int)sp[3 + 3 * q];
      if ((hi >> 8) == strip) {
        xcol[k] = hi & 255;
        xslot[k] = 2 * q;
      } else if ((lo >> 8) == strip) {
        xcol[k] = lo & 255;
        xslot[k] = 2 * q + 1;
      }
    }
  }
  uint32_t *side = wf.side + (size_t)f * wf.side_stride;
#define F360_FUSE_ROWS(NR)                                                                  \
  walk_fuse_rows<NR, PIX>(a, wf, plan, dst, lane, ent, n_ent, drows, mbox, max_dxw, xcol, xslot, \
                     npix, side, unit)
  if (n_ent <= 64) F360_FUSE_ROWS(1);
  else if (n_ent <= 128) F360_FUSE_ROWS(2);
  else if (n_ent <= 192) F360_FUSE_ROWS(3);
  else if (n_ent <= 256) F360_FUSE_ROWS(4);
  else F360_FUSE_ROWS(kFuseRounds);
#undef F360_FUSE_ROWS
}

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

// The row plan of encode + sample, one workgroup per frame.  Reduced row j is the box of table
// rows (lo, hi] (fov_maps.h: sample_axis); with the grid's offsets strictly increasing -- checked
// on the host -- lo and hi are non-decreasing in j and lo(j + 1) = hi(j) everywhere except where
// the clamps of sat_decoder_sample_rect_kernel.cl:201-204 bite, next to the frame's top and
// bottom edge.  A strip owner keeps ONE snapshot, so row j can be emitted by the walk iff
//   * no processed row snapshots strictly inside (lo, hi): with lo non-decreasing only the rows
//     just below can, and
//   * no earlier reduced row already emits at table row hi (two reduced rows clamped onto the
//     frame's last table row).
// Such rows get their marks -- EMIT | j | height at hi, SNAP at lo -- and every other processed
// row is left to walk_fuse_fix_kernel, which recognises it by the missing mark.
__global__ __launch_bounds__(256) void walk_fuse_plan_kernel(const int16_t *__restrict__ gy,
                                                             int out_h, int src_w, int src_h,
                                                             uint32_t *__restrict__ rowplan,
                                                             int plan_stride,
                                                             const WalkFuse wf) {
  uint32_t *plan = rowplan + (size_t)blockIdx.x * plan_stride;
  uint32_t *sp = wf.spix + (size_t)blockIdx.x * kSpixWords;
  const int cyp = wf.cyp[blockIdx.x];
  __shared__ int count, nleft;
  if (threadIdx.x == 0) count = nleft = 0;
  for (int y = threadIdx.x; y < plan_stride; y += 256) plan[y] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < out_h; j += 256) {
    const f360::AxisBox b = f360::sample_axis(cyp, gy[j + 1], gy[j], src_h, false);
    if (!b.ok) continue;
    bool fused = true;
    for (int d = 1; d <= 3; ++d) {
      if (j + d < out_h) {
        const f360::AxisBox n = f360::sample_axis(cyp, gy[j + d + 1], gy[j + d], src_h, false);
        if (n.ok && n.lo > b.lo && n.lo < b.hi) fused = false;
      }
      if (j - d >= 0) {
        const f360::AxisBox p = f360::sample_axis(cyp, gy[j - d + 1], gy[j - d], src_h, false);
        if (p.ok && p.hi == b.hi) fused = false;
      }
    }
    if (fused) {
      atomicOr(&plan[b.hi], kFuseEmit | (uint32_t)j | ((uint32_t)(b.hi - b.lo) << 16));
      atomicOr(&plan[b.lo], kFuseSnap);
    } else {
      const int k = atomicAdd(&nleft, 1);
      if (k < kFixLrows) sp[kSpixLrows + 1 + k] = (uint32_t)j;
    }
  }
  // the reduced columns whose box straddles two strips (in any order: the position in this
  // list is the pixel's slot in the side rows, for the helpers and for the fix-up alike)
  const int cxp = wf.cxp[blockIdx.x];
  for (int i = threadIdx.x; i < wf.out_w; i += 256) {
    const f360::AxisBox bx = f360::sample_axis(cxp, wf.gx[i + 1], wf.gx[i], src_w, true);
    if (bx.ok && (bx.hi >> 8) != (bx.lo >> 8)) {
      const int k = atomicAdd(&count, 1);
      if (k < kFixCols) {
        sp[1 + 3 * k] = (uint32_t)i;
        sp[2 + 3 * k] = (uint32_t)bx.hi;
        sp[3 + 3 * k] = (uint32_t)bx.lo;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sp[0] = (uint32_t)count;
    sp[kSpixLrows] = (uint32_t)nleft;
  }
}

// What the strip owners' helpers left out: in the reduced rows they emitted, the pixels whose box
// straddles two strips, from the D values the two strips' helpers put into the side rows; every
// processed pixel of the other reduced rows (the plan kernel's comment), from the finished
// table with sample_rect_kernel's arithmetic (sat_decoder.hip) -- or, when no table was asked
// for, as plain sums over the box's source pixels.
// (`yuv_model` -1: RGB0 sources; 0 / 1: planes, libswscale's C / x86 arithmetic)
__global__ __launch_bounds__(256) void walk_fuse_fix_kernel(const WalkBatch wb, const WalkFuse wf,
                                                            int src_w, int src_h,
                                                            int src_linesize, int yuv_model,
                                                            const f360::YuvPlanes yl,
                                                            const f360::YuvConsts yk) {
  const int f = blockIdx.y;
  const int cxp = wf.cxp[f], cyp = wf.cyp[f];
  const uint32_t *sat = wb.sat[f];
  uint8_t *dst = wf.dst[f];
  const uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  const uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  const uint32_t *side = wf.side + (size_t)f * wf.side_stride;
  const int npix = (int)sp[0];
  const int nleft = (int)sp[kSpixLrows];
  auto store = [&](int i, int j, uint3 q) {
    uint8_t *o = dst + (size_t)j * wf.dst_linesize + (size_t)i * 4;
    // (plain stores: these pixels' neighbours were written long ago, so each one is a partial
    // write of a cold line -- left in L2 they cost 3.2 us per 8K frame, written through 7.1)
    *reinterpret_cast<uint16_t *>(o) = (uint16_t)((q.x & 0xffu) | ((q.y & 0xffu) << 8));
    o[2] = (uint8_t)q.z;
  };
  auto from_table = [&](int i, int j, const f360::AxisBox &by) {
    const f360::AxisBox bx = f360::sample_axis(cxp, wf.gx[i + 1], wf.gx[i], src_w, true);
    if (!bx.ok) return;
    if (sat == nullptr) {  // no table was written: the box from the (converted) pixels
      const uint8_t *src = wb.src[f];
      uint3 n = make_uint3(0, 0, 0);
      for (int y = by.lo + 1; y <= by.hi; ++y)
        for (int x = bx.lo + 1; x <= bx.hi; ++x) {
          uint32_t v;
          if (yuv_model < 0) {
            v = *reinterpret_cast<const uint32_t *>(src + (size_t)y * src_linesize + 4 * x);
          } else {  // (wb.src is the luma plane; the chroma sample of the 2x2 block, yuv_device.h)
            const int Y = src[(size_t)y * yl.y_linesize + x];
            const int U = wb.u[f][(size_t)(y >> 1) * yl.u_linesize + (x >> 1)];
            const int V = wb.v[f][(size_t)(y >> 1) * yl.v_linesize + (x >> 1)];
            v = yuv_model == 0 ? f360::yuv_pixel<0>(yk, Y, f360::chroma_terms<0>(yk, U, V))
                               : f360::yuv_pixel<1>(yk, Y, f360::chroma_terms<1>(yk, U, V));
          }
          n.x += v & 0xffu;
          n.y += (v >> 8) & 0xffu;
          n.z += (v >> 16) & 0xffu;
        }
      store(i, j, f360::udiv3_exact(n, (uint32_t)((bx.hi - bx.lo) * (by.hi - by.lo))));
      return;
    }
    auto at = [&](int y, int x) {
      const uint32_t *p = sat + ((size_t)y * src_w + x) * 3;
      return make_uint3(p[0], p[1], p[2]);
    };
    const uint3 br = at(by.hi, bx.hi), tr = at(by.lo, bx.hi), tl = at(by.lo, bx.lo),
                bl = at(by.hi, bx.lo);
    store(i, j,
          f360::udiv3_exact(make_uint3(br.x - tr.x + tl.x - bl.x, br.y - tr.y + tl.y - bl.y,
                                       br.z - tr.z + tl.z - bl.z),
                            (uint32_t)((bx.hi - bx.lo) * (by.hi - by.lo))));
  };
  // workgroups past the straddling pixels': 256 columns of one listed leftover row each
  const int nsb = (wf.out_h * wf.pmax + 255) / 256;
  if ((int)blockIdx.x >= nsb) {
    const int ncc = (wf.out_w + 255) / 256;
    const int idx = blockIdx.x - nsb, lr = idx / ncc, i = (idx - lr * ncc) * 256 + threadIdx.x;
    if (lr >= min(nleft, kFixLrows) || nleft > kFixLrows || npix > wf.pmax || i >= wf.out_w) return;
    const int j = (int)sp[kSpixLrows + 1 + lr];
    from_table(i, j, f360::sample_axis(cyp, wf.gy[j + 1], wf.gy[j], src_h, false));
    return;
  }
  // one thread per (reduced row, straddling pixel): no thread waits for more than its own loads
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int j = t / wf.pmax, k = t - j * wf.pmax;
  if (j >= wf.out_h) return;
  const f360::AxisBox by = f360::sample_axis(cyp, wf.gy[j + 1], wf.gy[j], src_h, false);
  if (!by.ok) return;
  const uint32_t pr = plan[by.hi];
  const bool emitted = (pr & kFuseEmit) && (int)(pr & 0xffffu) == j && npix <= wf.pmax;
  if (emitted) {
    if (k >= npix) return;
    const uint32_t *v = side + ((size_t)j * npix + k) * 6;
    const uint32_t area = (sp[2 + 3 * k] - sp[3 + 3 * k]) * (uint32_t)(by.hi - by.lo);
    store((int)sp[1 + 3 * k], j,
          f360::udiv3_exact(make_uint3(v[0] - v[3], v[1] - v[4], v[2] - v[5]), area));
  } else if (nleft > kFixLrows || npix > wf.pmax) {
    // (more leftover rows than the list holds: this row's pmax threads take the whole row)
    for (int i = k; i < wf.out_w; i += wf.pmax) from_table(i, j, by);
  }
}

int ensure_plan(f360_ctx *ctx, int width, int height, bool planar = false, int frames = 1) {
  f360::SatEncodePlan &p = ctx->enc;
  // band height: the largest of 64 / 32 / 16 rows that still yields enough tiles (one wave
  // each in the writer) to fill 256 CUs -- 64 at 7680x3840, 16 at 3840x1920 and below
  int band_rows = ctx->opt_band_rows;
  if (band_rows == 0) {
    const int strips = (width + kStripPx - 1) / kStripPx;
    band_rows = 16;
    for (int cand : {64, 32})
      if ((long)strips * ((height + cand - 1) / cand) >= 1500) {
        band_rows = cand;
        break;
      }
  }
  // bands per reducer wave: at least 2, and few enough super-bands (<= 32) that the carry
  // kernel needs a single round of loads
  // (planar sources convert in the reducer, which makes it instruction-bound: one band per
  // wave doubles the waves, 45 -> 34 us at 8K)
  int sb_bands = ctx->opt_sb_bands;
  // (with 64-row bands one band per reducer wave also makes the reducer visit the frame in the
  // writer's tile order, and the writer's re-read then finds more of it in the caches: 80 -> 76 us
  // at 8K for +1.7 us in the carry kernel; with 16-row bands two bands per wave stay better)
  // (a small frame has too few reducer waves to fill the device with two bands each: 1080p
  // 272 waves of 32 rows against 544 of 16 -- reducer 11.6 -> 8.4 us, 11.8 -> 9.0 at 2560x1440)
  if (sb_bands < 0)
    sb_bands = (planar || band_rows == 64 || (long)width * height <= 4200000L) ? 1 : 2;
  if (sb_bands == 0) {
    const int nb = (height + band_rows - 1) / band_rows;
    sb_bands = (nb + 31) / 32;
    if (sb_bands < 2) sb_bands = 2;
  }
  // The scratch is carved for the layout with the most super-bands (one band each), so callers
  // that alternate planar and RGB0 sources on one context (different sb_bands at <= 4K) only
  // change two numbers: no re-carve, no synchronisation.
  if (p.width == width && p.height == height && p.band_rows == band_rows && p.ws.p &&
      p.frames >= frames) {
    p.sb_bands = sb_bands;
    p.nsb = (p.nbands + sb_bands - 1) / sb_bands;
    return F360_OK;
  }
  frames = std::max(frames, p.width == width && p.height == height ? p.frames : 1);
  // A geometry change re-carves the scratch; wait for work that may use it.
  if (p.ws.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  p.width = width;
  p.height = height;
  p.band_rows = band_rows;
  p.sb_bands = sb_bands;
  p.nstrips = (width + kStripPx - 1) / kStripPx;
  p.nbands = (height + p.band_rows - 1) / p.band_rows;
  p.nsb = (p.nbands + p.sb_bands - 1) / p.sb_bands;
  p.wp3 = p.nstrips * kStripPx * 3;
  auto align = [](size_t n) { return (n + 63) & ~(size_t)63; };
  const size_t n_lp = align((size_t)p.nbands * p.wp3);
  const size_t n_sb = align((size_t)p.nbands * p.wp3);  // room for sb_bands = 1
  const size_t n_row = align((size_t)p.nstrips * height * 3);
  const size_t n_tile = align((size_t)p.nstrips * p.nbands * 3);
  const size_t total = n_lp + 2 * n_sb + 2 * n_row + 2 * n_tile;
  int st = p.ws.reserve(total * sizeof(uint32_t) * (size_t)frames);
  if (st != F360_OK) {
    p.width = p.height = 0;
    p.frames = 0;
    return st;
  }
  p.frames = frames;
  p.ws_stride = total;
  uint32_t *w = p.ws.as<uint32_t>();
  p.lp = w;             w += n_lp;
  p.sbtotal = w;        w += n_sb;
  p.sbprefix = w;       w += n_sb;
  p.rowsum = w;         w += n_row;
  p.rowcarry = w;       w += n_row;
  p.tiletotal = w;      w += n_tile;
  p.tprefix = w;
  return F360_OK;
}

}  // namespace

extern "C" int f360_sat_encode_batch_max(void) { return kEncBatch; }

extern "C" int f360_sat_encode_prepare(f360_ctx *ctx, int width, int height) {
  F360_REQUIRE(ctx, "f360_sat_encode_prepare: null context");
  F360_REQUIRE(width >= 1 && height >= 1, "f360_sat_encode_prepare: bad size %dx%d",
               width, height);
  F360_BIND_DEVICE(ctx);
  return ensure_plan(ctx, width, height);
}

namespace f360 {

int sat_encode_impl(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *src_dev, int width,
                    int height, int linesize, const SatEmit *emit, const YuvPlanes *yuv,
                    int count, uint32_t *const *sats, const uint8_t *const *srcs, int profile,
                    const YuvPlanes *yuvs) {
  F360_REQUIRE(ctx, "f360_sat_encode: null context");
  F360_BIND_DEVICE(ctx);
  if (count > 0) {
    F360_REQUIRE(count <= kEncBatch && sats && (srcs || yuvs) && !emit && !yuv,
                 "f360_sat_encode_batch: count %d outside 1..%d, or null arrays", count, kEncBatch);
    for (int k = 0; k < count; ++k) {
      F360_REQUIRE(sats[k] && (yuvs ? yuvs[k].y && yuvs[k].u && yuvs[k].v : srcs[k] != nullptr),
                   "f360_sat_encode_batch: null buffer %d", k);
      if (yuvs)  // one set of linesizes and alignments for the whole launch
        F360_REQUIRE(yuvs[k].y_linesize == yuvs[0].y_linesize &&
                         yuvs[k].u_linesize == yuvs[0].u_linesize &&
                         yuvs[k].v_linesize == yuvs[0].v_linesize &&
                         ((uintptr_t)yuvs[k].y % 4) == 0 && ((uintptr_t)yuvs[k].u % 2) == 0 &&
                         ((uintptr_t)yuvs[k].v % 2) == 0 && ((uintptr_t)sats[k] % 16) == 0,
                     "f360_sat_encode_yuv420p_batch: frame %d: other linesizes than frame 0, "
                     "or a misaligned plane / table", k);
    }
    sat_dev = sats[0];
    if (yuvs) yuv = &yuvs[0];
    else src_dev = srcs[0];
  }
  F360_REQUIRE((sat_dev || emit) && (src_dev || yuv), "f360_sat_encode: null buffer");
  F360_REQUIRE(width >= 1 && height >= 1, "f360_sat_encode: bad size %dx%d", width,
               height);
  const int bpp = yuv ? 4 : linesize / width;  // src/sat_encoder_encode_kernels.cl:9
  F360_REQUIRE(bpp >= 3, "f360_sat_encode: linesize %d gives %d bytes per pixel (need >= 3)",
               linesize, bpp);
  if (yuv) {
    F360_REQUIRE(yuv->y && yuv->u && yuv->v, "f360_sat_encode_yuv420p: null plane");
    // the planar path loads 4 luma bytes and 2 + 2 chroma bytes per lane and row
    F360_REQUIRE(width % 4 == 0 && height % 2 == 0,
                 "f360_sat_encode_yuv420p: size %dx%d (need width %% 4 == 0, even height)",
                 width, height);
    F360_REQUIRE(yuv->y_linesize >= width && yuv->u_linesize >= width / 2 &&
                     yuv->v_linesize >= width / 2 && yuv->y_linesize % 4 == 0 &&
                     yuv->u_linesize % 2 == 0 && yuv->v_linesize % 2 == 0,
                 "f360_sat_encode_yuv420p: linesizes %d/%d/%d (need >= row, y %% 4, u,v %% 2)",
                 yuv->y_linesize, yuv->u_linesize, yuv->v_linesize);
    F360_REQUIRE(((uintptr_t)yuv->y % 4) == 0 && ((uintptr_t)yuv->u % 2) == 0 &&
                     ((uintptr_t)yuv->v % 2) == 0 && (emit || ((uintptr_t)sat_dev % 16) == 0),
                 "f360_sat_encode_yuv420p: misaligned plane or table");
  }
  F360_REQUIRE((size_t)width * height * 3 < ((size_t)1 << 31),
               "f360_sat_encode: frame too large for 32-bit element indices");
  int st = ensure_plan(ctx, width, height, yuv != nullptr, count > 0 ? count : 1);
  if (st != F360_OK) return st;
  const f360::SatEncodePlan &p = ctx->enc;

  EncodeArgs a;
  a.sat = sat_dev;
  a.src = src_dev;
  a.width = width;
  a.height = height;
  a.linesize = linesize;
  a.bpp = bpp;
  a.band_rows = p.band_rows;
  a.sb_bands = p.sb_bands;
  a.nstrips = p.nstrips;
  a.nbands = p.nbands;
  a.nsb = p.nsb;
  a.wp3 = p.wp3;
  a.lp = p.lp;
  a.sbtotal = p.sbtotal;
  a.sbprefix = p.sbprefix;
  a.rowsum = p.rowsum;
  a.rowcarry = p.rowcarry;
  a.tiletotal = p.tiletotal;
  a.tprefix = p.tprefix;
  a.ablate = ctx->opt_ablate;
  a.xmap = emit ? emit->xmap : nullptr;
  a.ymap = emit ? emit->ymap : nullptr;
  a.corners = emit ? emit->corners : nullptr;
  a.corner_stride = emit ? emit->corner_stride : 0;
  a.has_maps = emit && emit->maps ? 1 : 0;
  if (a.has_maps) a.maps = *emit->maps;
  else a.maps = FovMaps{};
  a.yuv = yuv ? *yuv : YuvPlanes{nullptr, nullptr, nullptr, 0, 0, 0};
  if (yuv)
    build_yuv2rgb_consts(a.k);
  else
    a.k = YuvConsts{};

  a.nbatch = count;
  a.ws_stride = p.ws_stride;
  EncodeBatch eb;
  for (int k = 0; k < kEncBatch; ++k) {
    const int q = k < count ? k : 0;
    eb.src[k] = count <= 0 ? nullptr : yuvs ? yuvs[q].y : srcs[q];
    eb.sat[k] = count > 0 ? sats[q] : nullptr;
    eb.u[k] = count > 0 && yuvs ? yuvs[q].u : nullptr;
    eb.v[k] = count > 0 && yuvs ? yuvs[q].v : nullptr;
  }
  const unsigned frames = count > 0 ? (unsigned)count : 1u;

  const bool prof = profile < 0 ? f360::take_profile_slot(ctx) : profile != 0;
  bool vec = !yuv && bpp == 4 && (width % 4) == 0 && (linesize % 16) == 0 &&
             ((uintptr_t)src_dev % 16) == 0 && (emit || ((uintptr_t)sat_dev % 16) == 0);
  for (int k = 1; k < count && !yuvs; ++k)  // one kernel flavour for the whole batch
    vec = vec && ((uintptr_t)srcs[k] % 16) == 0 && ((uintptr_t)sats[k] % 16) == 0;
  const int yuv_src = !yuv ? 0 : ctx->opt_yuv_model == 1 ? kSrcYuvSwsX86 : kSrcYuvSwsC;
  const dim3 block(64 * kWavesPerBlock);
  const int reduce_blocks = (p.nstrips * p.nsb + kWavesPerBlock - 1) / kWavesPerBlock;
  a.reduce_blocks = reduce_blocks;
  const int blocks1 = reduce_blocks + (a.has_maps ? 2 : 0);

  {
    f360::KernelSpan span(ctx, f360::kSatReduce, prof, (int)frames);
    if (yuv_src == kSrcYuvSwsX86)
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcYuvSwsX86>, dim3(blocks1, frames), block, 0,
                         ctx->stream, a, eb);
    else if (yuv_src == kSrcYuvSwsC)
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcYuvSwsC>, dim3(blocks1, frames), block, 0,
                         ctx->stream, a, eb);
    else if (vec)
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcRgb0>, dim3(blocks1, frames), block, 0, ctx->stream, a, eb);
    else
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcBytes>, dim3(blocks1, frames), block, 0,
                         ctx->stream, a, eb);
  }

  if (ctx->opt_ablate & 8) return F360_OK;  // timing experiments: reducer only
  auto seg = [](const uint32_t *in, uint32_t *out, int n, int K) {
    const int parts = K > 128 ? 1 : (K + 31) / 32, cols = 256 / (parts < 1 ? 1 : parts);
    return ScanSeg{in, out, n, K, (n + cols - 1) / cols};
  };
  ScanSeg sa = seg(p.sbtotal, p.sbprefix, p.wp3, p.nsb);
  ScanSeg sb = seg(p.rowsum, p.rowcarry, height * 3, p.nstrips);
  ScanSeg sc = seg(p.tiletotal, p.tprefix, p.nbands * 3, p.nstrips);
  {
    f360::KernelSpan span(ctx, f360::kSatCarry, prof, (int)frames);
    hipLaunchKernelGGL(sat_carry_kernel,
                       dim3(sa.nblocks + sb.nblocks + sc.nblocks, frames), dim3(256), 0,
                       ctx->stream, sa, sb, sc, p.ws_stride);
  }
  {
    f360::KernelSpan span(ctx, f360::kSatWrite, prof, (int)frames);
    const dim3 grid3((p.nstrips * p.nbands + kWavesPerBlock - 1) / kWavesPerBlock, frames);
    if (yuv_src == kSrcYuvSwsX86 && emit)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsX86, 2>), grid3, block, 0, ctx->stream, a, eb);
    else if (yuv_src == kSrcYuvSwsX86)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsX86, 1>), grid3, block, 0, ctx->stream, a, eb);
    else if (yuv_src == kSrcYuvSwsC && emit)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsC, 2>), grid3, block, 0, ctx->stream, a, eb);
    else if (yuv_src == kSrcYuvSwsC)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsC, 1>), grid3, block, 0, ctx->stream, a, eb);
    else if (emit && vec)
      hipLaunchKernelGGL((sat_write_kernel<kSrcRgb0, 2>), grid3, block, 0, ctx->stream, a, eb);
    else if (emit)
      hipLaunchKernelGGL((sat_write_kernel<kSrcBytes, 2>), grid3, block, 0, ctx->stream, a, eb);
    else if (!vec)
      hipLaunchKernelGGL((sat_write_kernel<kSrcBytes, 0>), grid3, block, 0, ctx->stream, a, eb);
    else  // (direct 48-byte-stride stores for RGB0 frames, "sat.store" = 0, were an A/B switch
          // until round 4: 136 against 88 us at 8K)
      hipLaunchKernelGGL((sat_write_kernel<kSrcRgb0, 1>), grid3, block, 0, ctx->stream, a, eb);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // namespace f360

namespace {

// Whether a batched call of `count` frames takes the read-once encoder ("sat.walk").
bool walk_wanted(const f360_ctx *ctx, int count, int width) {
  if (ctx->opt_walk == 0 || width > f360::kMaxDim) return false;
  if (ctx->opt_walk == 1) return true;
  const long strips = (width + kStripPx - 1) / kStripPx;
  return (long)count * strips >= ctx->opt_walk_units;
}

// f360_sat_encode_batch / _yuv420p_batch on the read-once encoder: launches of up to kWalkFrames
// frames.  The caller has checked the arguments and that every buffer allows 16-byte accesses.
int sat_encode_walk(f360_ctx *ctx, int count, uint32_t *const *sats, const uint8_t *const *srcs,
                    const f360::YuvPlanes *yuvs, int width, int height, int linesize,
                    bool prof, const f360::SatFuse *fuse = nullptr) {
  f360::SatEncodePlan &p = ctx->enc;
  const int nstrips = (width + kStripPx - 1) / kStripPx;
  const int nb = (height + kRowUnroll - 1) / kRowUnroll;
  // launches of equal size (65 frames: 33 + 32, not 64 + 1 -- a launch of one frame would be 30
  // strip owners on an empty device)
  // Frames per launch ("sat.walk_frames", 0 = automatic): one workgroup per CU -- one strip owner
  // per SIMD -- is the sweet spot (32 frames at 8K: 78.4 us per frame against 82.8 for 64 in one
  // launch on the same box: with two owners per SIMD the per-batch jitter that the hand-off
  // chain accumulates is five times larger), so a larger call runs as
  // several launches of about 1024 units, never fewer units than the call's own frames allow.
  // A launch must not exceed the device by a little: 35 frames (1050 units) take 3.8 ms -- the
  // 26 units of the second round walk their 480 batches alone -- where 32 take 2.6; and a launch
  // costs its serial chain (height / 8 batches of ~3.5 us: 1.7 ms at 8K) however few frames it
  // holds, so 40 frames as one oversubscribed launch (4.3 ms) or as two of 20 (4.1 ms) are both
  // no better than the three kernels (profiles/round4_few_frames.txt).
  int max_frames = ctx->opt_walk_frames > 0 ? ctx->opt_walk_frames : std::max(1024 / nstrips, 1);
  max_frames = std::min(max_frames, kWalkFrames);
  const int nlaunch = (count + max_frames - 1) / max_frames;
  const int per_launch = (count + nlaunch - 1) / nlaunch;
  // Everything below that allocates, clears or synchronises is illegal while the stream is
  // being captured into a hipGraph, and a captured launch keeps the hand-off buffer's address:
  // warm up eagerly with the largest geometry and frame count first.
  const size_t gran_bytes = (size_t)per_launch * nstrips * nb * kWalkLanes * 8;
  const size_t chain_bytes = gran_bytes + (size_t)per_launch * nstrips * 64;
  // encode + sample: a row plan per frame of a launch (one word per table row, whole batches)
  const int plan_stride = nb * kRowUnroll;
  const int pmax = std::max(1, std::min(3 * (nstrips - 1), kFixCols));
  const size_t side_stride = fuse ? (size_t)fuse->out_h * pmax * 6 : 0;  // dwords per frame
  const size_t plan_words = (size_t)per_launch * plan_stride;
  const size_t plan_bytes =
      fuse ? (plan_words + (size_t)per_launch * kSpixWords + (size_t)per_launch * side_stride) * 4
           : 0;
  if (!p.walk_state.p || !p.walk_err_host || chain_bytes > p.walk_chain.bytes ||
      plan_bytes > p.walk_plan.bytes) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    F360_HIP_TRY(hipStreamIsCapturing(ctx->stream, &cap));
    F360_REQUIRE(cap == hipStreamCaptureStatusNone,
                 "f360_sat_encode_batch: the read-once encoder must allocate its hand-off buffers "
                 "(%zu bytes) but the stream is being captured; run the same call once before "
                 "the capture", chain_bytes);
  }
  // state words: zero ticket / done, serial 1; the launches advance them
  if (!p.walk_state.p) {
    int st = p.walk_state.reserve(64);
    if (st != F360_OK) return st;
    const WalkState init{0u, 0u, 1ull};
    F360_HIP_TRY(hipMemsetAsync(p.walk_state.p, 0, 64, ctx->stream));
    F360_HIP_TRY(hipMemcpyAsync(p.walk_state.p, &init, sizeof(init), hipMemcpyHostToDevice,
                                ctx->stream));
    F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // `init` is a stack object
  }
  if (!p.walk_err_host) {
    void *h = nullptr, *d = nullptr;
    F360_HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
    *static_cast<uint32_t *>(h) = 0;
    F360_HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
    p.walk_err_host = static_cast<uint32_t *>(h);
    p.walk_err_dev = static_cast<uint32_t *>(d);
  }
  // granules: zeroed when (re)allocated -- a tag is never 0 -- and never again
  // (+ 32 bytes per unit behind the granules: the debug statistics of debug.ablate bit 8)
  if (chain_bytes > p.walk_chain.bytes) {
    if (p.walk_chain.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int st = p.walk_chain.reserve(chain_bytes);
    if (st != F360_OK) return st;
    F360_HIP_TRY(hipMemsetAsync(p.walk_chain.p, 0, p.walk_chain.bytes, ctx->stream));
  }

  if (plan_bytes > p.walk_plan.bytes) {
    if (p.walk_plan.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int st = p.walk_plan.reserve(plan_bytes);
    if (st != F360_OK) return st;
  }

  EncodeArgs a{};
  a.width = width;
  a.height = height;
  a.linesize = linesize;
  a.bpp = 4;
  a.nstrips = nstrips;
  a.ablate = ctx->opt_ablate;
  a.yuv = yuvs ? yuvs[0] : f360::YuvPlanes{nullptr, nullptr, nullptr, 0, 0, 0};
  if (yuvs)
    f360::build_yuv2rgb_consts(a.k);
  else
    a.k = f360::YuvConsts{};
  a.walk_nbatches = nb;
  a.walk = p.walk_state.as<WalkState>();
  a.walk_chain = p.walk_chain.as<unsigned long long>();
  a.walk_err = p.walk_err_dev;
  a.walk_spin = ctx->opt_walk_spin > 0 ? (uint32_t)ctx->opt_walk_spin : kWalkSpinDefault;
  a.walk_mute = ctx->opt_walk_mute - 1;
  a.walk_stats = reinterpret_cast<unsigned long long *>(p.walk_chain.as<uint8_t>() + gran_bytes);
  p.walk_stats_units = per_launch * nstrips;
  p.walk_stats_offset = gran_bytes;
  const int yuv_src = !yuvs ? 0 : ctx->opt_yuv_model == 1 ? kSrcYuvSwsX86 : kSrcYuvSwsC;

  for (int k0 = 0; k0 < count; k0 += per_launch) {
    const int n = std::min(count - k0, per_launch);
    WalkBatch wb;
    for (int k = 0; k < kWalkFrames; ++k) {
      const int q = k0 + (k < n ? k : 0);
      wb.src[k] = yuvs ? yuvs[q].y : srcs[q];
      wb.sat[k] = sats ? sats[q] : nullptr;  // (null: one pass without tables)
      wb.u[k] = yuvs ? yuvs[q].u : nullptr;
      wb.v[k] = yuvs ? yuvs[q].v : nullptr;
    }
    a.walk_units = n * nstrips;
    const dim3 grid((a.walk_units + kWalkWaves - 1) / kWalkWaves);
    const dim3 block(64 * kWalkWaves);
    if (fuse) {
      WalkFuse wf;
      for (int k = 0; k < kWalkFrames; ++k) {
        const int q = k0 + (k < n ? k : 0);
        wf.dst[k] = fuse->dsts[q];
        wf.cxp[k] = (int)(fuse->centers_xy[2 * q] * (float)width);  // sat_decoder_sample_rect_kernel.cl:176-179
        wf.cyp[k] = (int)(fuse->centers_xy[2 * q + 1] * (float)height);
      }
      wf.gx = fuse->gx;
      wf.gy = fuse->gy;
      wf.rowplan = p.walk_plan.as<uint32_t>();
      wf.plan_stride = plan_stride;
      wf.out_w = fuse->out_w;
      wf.out_h = fuse->out_h;
      wf.dst_linesize = fuse->dst_linesize;
      wf.spix = wf.rowplan + plan_words;
      wf.side = wf.spix + (size_t)per_launch * kSpixWords;
      wf.pmax = pmax;
      wf.side_stride = side_stride;
      {
        f360::KernelSpan span(ctx, f360::kWalkFusePlan, prof, n);
        hipLaunchKernelGGL(walk_fuse_plan_kernel, dim3(n), dim3(256), 0, ctx->stream, wf.gy,
                           wf.out_h, width, height, wf.rowplan, plan_stride, wf);
      }
      {
        f360::KernelSpan span(ctx, f360::kSatWalk, prof, n);
        const dim3 fgrid((a.walk_units + kFuseOwners - 1) / kFuseOwners), fblock(128 * kFuseOwners);
        if (yuv_src == kSrcYuvSwsX86)
          hipLaunchKernelGGL((sat_walk_kernel<kSrcYuvSwsX86, 2, true>), fgrid, fblock, 0,
                             ctx->stream, a, wb, wf);
        else if (yuv_src == kSrcYuvSwsC)
          hipLaunchKernelGGL((sat_walk_kernel<kSrcYuvSwsC, 2, true>), fgrid, fblock, 0,
                             ctx->stream, a, wb, wf);
        else
          hipLaunchKernelGGL((sat_walk_kernel<kSrcRgb0, 2, true>), fgrid, fblock, 0, ctx->stream,
                             a, wb, wf);
      }
      {
        f360::KernelSpan span(ctx, f360::kWalkFuseFix, prof, n);
        hipLaunchKernelGGL(walk_fuse_fix_kernel,
                           dim3((wf.out_h * pmax + 255) / 256 +
                                    kFixLrows * ((wf.out_w + 255) / 256), n),
                           dim3(256), 0, ctx->stream, wb, wf, width, height, linesize,
                           !yuvs ? -1 : ctx->opt_yuv_model == 1 ? 1 : 0, a.yuv, a.k);
      }
      continue;
    }
    f360::KernelSpan span(ctx, f360::kSatWalk, prof, n);
#define F360_WALK_LAUNCH(SRC)                                                                   \
  hipLaunchKernelGGL((sat_walk_kernel<SRC, 2>), grid, block, 0, ctx->stream, a, wb, WalkNoFuse{})
    if (yuv_src == kSrcYuvSwsX86)
      F360_WALK_LAUNCH(kSrcYuvSwsX86);
    else if (yuv_src == kSrcYuvSwsC)
      F360_WALK_LAUNCH(kSrcYuvSwsC);
    else
      F360_WALK_LAUNCH(kSrcRgb0);
#undef F360_WALK_LAUNCH
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // namespace

// Encode + sample in one pass (f360_satdec_encode_sample_frames, sat_decoder.hip).  False: the
// read-once encoder would not take this call, or the frames / grid are outside what the strip
// owners' packed bookkeeping holds; the caller then makes the two calls.
bool f360::sat_encode_sample_applies(const f360_ctx *ctx, int count, int width, int height,
                                     int linesize, int out_w, int out_h, int dst_linesize,
                                     const f360::YuvPlanes *yuv) {
  // the source side: f360_sat_encode_batch's / f360_sat_encode_yuv420p_batch's own rules
  const bool source_ok =
      yuv ? height % 2 == 0 && yuv->y_linesize >= width && yuv->u_linesize >= width / 2 &&
                yuv->v_linesize >= width / 2 && yuv->y_linesize % 4 == 0 &&
                yuv->u_linesize % 2 == 0 && yuv->v_linesize % 2 == 0
          : linesize / width == 4 && linesize % 16 == 0;
  return walk_wanted(ctx, count, width) && source_ok && width % 4 == 0 &&
         (size_t)width * height * 3 < ((size_t)1 << 31) &&
         out_w < 65536 && out_h < 65536 && (width + kStripPx - 1) / kStripPx <= kFixCols / 4 &&
         dst_linesize % 4 == 0 && dst_linesize >= 4 * out_w;
}
int f360::sat_encode_sample_walk(f360_ctx *ctx, int count, uint32_t *const *sats,
                                 const uint8_t *const *srcs, const f360::YuvPlanes *yuvs,
                                 int width, int height, int linesize, const f360::SatFuse &fuse,
                                 bool prof) {
  return sat_encode_walk(ctx, count, sats, srcs, yuvs, width, height, linesize, prof, &fuse);
}

// Debug: the per-unit statistics of the last read-once launch that ran with debug.ablate bit 8
// ({start, end} in 100 MHz ticks, slow-path waits | shader cycles << 16, polls spent waiting); returns the unit count.
extern "C" int f360_debug_walk_stats(f360_ctx *ctx, unsigned long long *out, int max_units) {
  F360_REQUIRE(ctx && out && max_units >= 0, "f360_debug_walk_stats: bad argument");
  F360_BIND_DEVICE(ctx);
  const f360::SatEncodePlan &p = ctx->enc;
  const int n = std::min(max_units, p.walk_stats_units);
  if (n <= 0 || !p.walk_chain.p) return 0;
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  F360_HIP_TRY(hipMemcpy(out, p.walk_chain.as<uint8_t>() + p.walk_stats_offset, (size_t)n * 64,
                         hipMemcpyDeviceToHost));
  return n;
}

// ------------------------------------------------------------------------------------------
// Tables for batched encodes, placed for the read-once encoder (f360_sat_tables_alloc).
//
// A launch of sat_walk_kernel writes `group` tables (32 at 8K) at the same time, and the rate at
// which this device takes those writes depends on what backs the tables: 5.5 to 7 TB/s for the
// write pattern alone, in two clusters, by ALLOCATION -- not by address, pitch, row phase or
// allocation API (profiles/round4_table_placement.txt, tools/frontbench).  User space can only
// measure it.  So the pool draws groups: every group allocated while all earlier ones are still
// held (so that it is backed by other memory), alternately as one allocation per table and as one
// slab, one encode launch of scratch frames timed into each (the second of two), the fastest
// ones kept, the rest given back.  Calls too small for the read-once encoder get plain allocations.
struct f360_table_pool {
  std::vector<void *> allocs;   // what f360_sat_tables_free gives back
  std::string report;
};

extern "C" int f360_sat_tables_alloc(f360_ctx *ctx, int width, int height, int count,
                                     uint32_t **tables_out, f360_table_pool **pool_out) {
  F360_REQUIRE(ctx && tables_out && pool_out && count >= 1 && width >= 1 && height >= 1,
               "f360_sat_tables_alloc: bad argument");
  F360_REQUIRE(f360::dims_ok({width, height}), "f360_sat_tables_alloc: a dimension exceeds 65536");
  F360_BIND_DEVICE(ctx);
  *pool_out = nullptr;
  const size_t tb = (size_t)width * height * 12;
  auto *pool = new f360_table_pool;
  auto fail = [&](int st) {
    for (void *p : pool->allocs) (void)hipFree(p);
    delete pool;
    return st;
  };
  const int strips = (width + kStripPx - 1) / kStripPx;
  // (the conditions under which f360_sat_encode_batch takes the read-once encoder)
  const bool walks = walk_wanted(ctx, count, width) && width % 4 == 0 &&
                     (size_t)width * height * 3 < ((size_t)1 << 31);
  char line[160];
  if (!walks) {
    for (int k = 0; k < count; ++k) {
      void *p = nullptr;
      if (hipMalloc(&p, tb) != hipSuccess) {
        (void)hipGetLastError();
        f360::set_error("f360_sat_tables_alloc: out of device memory");
        return fail(F360_ERR_HIP);
      }
      pool->allocs.push_back(p);
      tables_out[k] = static_cast<uint32_t *>(p);
    }
    pool->report = "one allocation per table (the call is below the read-once encoder's threshold)";
    *pool_out = pool;
    return F360_OK;
  }
  int max_frames = ctx->opt_walk_frames > 0 ? ctx->opt_walk_frames : std::max(1024 / strips, 1);
  max_frames = std::min(max_frames, kWalkFrames);
  const int nlaunch = (count + max_frames - 1) / max_frames;
  const int group = (count + nlaunch - 1) / nlaunch;
  struct Draw {
    float us = 0;
    bool slab = false;
    std::vector<void *> allocs;
    std::vector<uint32_t *> tabs;
  };
  std::vector<Draw> draws;
  auto drop = [&](Draw &d) {
    for (void *p : d.allocs) (void)hipFree(p);
    d.allocs.clear();
  };
  void *zero = nullptr;
  const size_t fb = (size_t)width * height * 4;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = F360_OK;
  // (sources: `group` frames' worth of scratch, contents irrelevant -- distinct frames, so that
  // the timed launch reads from memory as a real one does)
  if (hipMalloc(&zero, fb * group) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess)
    st = F360_ERR_HIP;
  const float good_us = 80.5f * (float)((double)width * height / (7680.0 * 3840.0));
  std::vector<const uint8_t *> srcs((size_t)group);
  for (int k = 0; k < group; ++k) srcs[(size_t)k] = static_cast<const uint8_t *>(zero) + (size_t)k * fb;
  // (up to nlaunch + 12 draws: on the boxes where most allocations draw badly -- 7 of 8 at
  // 86-94 us seen -- eight draws left a 1-in-4 chance of keeping a mediocre group; a draw is
  // 11 GB at 8K, held until the choice is made, and running out of memory just ends the drawing)
  for (int i = 0; st == F360_OK && i < nlaunch + 12; ++i) {
    Draw d;
    d.slab = i % 2 == 1;
    if (d.slab) {
      void *p = nullptr;
      if (hipMalloc(&p, tb * group) != hipSuccess) {
        (void)hipGetLastError();
        break;  // out of memory: make do with what has been drawn
      }
      d.allocs.push_back(p);
      for (int k = 0; k < group; ++k)
        d.tabs.push_back(reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(p) + (size_t)k * tb));
    } else {
      bool ok = true;
      for (int k = 0; k < group && ok; ++k) {
        void *p = nullptr;
        ok = hipMalloc(&p, tb) == hipSuccess;
        if (ok) {
          d.allocs.push_back(p);
          d.tabs.push_back(static_cast<uint32_t *>(p));
        }
      }
      if (!ok) {
        (void)hipGetLastError();
        drop(d);
        break;
      }
    }
    for (int rep = 0; rep < 2 && st == F360_OK; ++rep) {  // the second launch is the measurement
      if (hipEventRecord(e0, ctx->stream) != hipSuccess) st = F360_ERR_HIP;
      if (st == F360_OK)
        st = sat_encode_walk(ctx, group, d.tabs.data(), srcs.data(), nullptr, width, height,
                             4 * width, false);
      if (st == F360_OK && (hipEventRecord(e1, ctx->stream) != hipSuccess ||
                            hipEventSynchronize(e1) != hipSuccess))
        st = F360_ERR_HIP;
      float ms = 0;
      if (st == F360_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) st = F360_ERR_HIP;
      d.us = 1e3f * ms / (float)group;
    }
    snprintf(line, sizeof line, "%s%s %.1f", draws.empty() ? "" : ", ", d.slab ? "slab" : "separate",
             d.us);
    pool->report += line;
    draws.push_back(std::move(d));
    int good = 0;
    for (const Draw &x : draws) good += x.us <= good_us;
    if (good >= nlaunch) break;
  }
  if (zero) (void)hipFree(zero);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st != F360_OK || (int)draws.size() < nlaunch) {
    for (Draw &d : draws) drop(d);
    if (st == F360_OK) {
      f360::set_error("f360_sat_tables_alloc: out of device memory");
      st = F360_ERR_HIP;
    }
    return fail(st);
  }
  std::sort(draws.begin(), draws.end(), [](const Draw &a, const Draw &b) { return a.us < b.us; });
  pool->report = "us per frame of one launch into each drawn group of " + std::to_string(group) +
                 " tables: " + pool->report + "; kept:";
  int k = 0;
  for (int g = 0; g < (int)draws.size(); ++g) {
    if (g < nlaunch) {
      snprintf(line, sizeof line, " %s %.1f", draws[g].slab ? "slab" : "separate", draws[g].us);
      pool->report += line;
      for (void *p : draws[g].allocs) pool->allocs.push_back(p);
      for (uint32_t *t : draws[g].tabs)
        if (k < count) tables_out[k++] = t;
    } else {
      drop(draws[g]);
    }
  }
  *pool_out = pool;
  return F360_OK;
}

extern "C" int f360_sat_tables_free(f360_ctx *ctx, f360_table_pool *pool) {
  F360_REQUIRE(ctx, "f360_sat_tables_free: null context");
  if (!pool) return F360_OK;
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (void *p : pool->allocs) (void)hipFree(p);
  delete pool;
  return F360_OK;
}

extern "C" const char *f360_sat_tables_report(const f360_table_pool *pool) {
  return pool ? pool->report.c_str() : "";
}

// Strips of read-once launches that gave up waiting for their hand-off and finished alone (the
// tables are right either way) since the last call of this function; blocks until the stream
// has drained.
extern "C" int f360_debug_walk_recoveries(f360_ctx *ctx, unsigned *count_out) {
  F360_REQUIRE(ctx && count_out, "f360_debug_walk_recoveries: bad argument");
  F360_BIND_DEVICE(ctx);
  *count_out = 0;
  f360::SatEncodePlan &p = ctx->enc;
  if (!p.walk_err_host) return F360_OK;
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *count_out = *p.walk_err_host;
  *p.walk_err_host = 0;
  return F360_OK;
}

extern "C" int f360_sat_encode(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *src_dev,
                               int width, int height, int linesize) {
  F360_REQUIRE(sat_dev, "f360_sat_encode: null buffer");
  return f360::sat_encode_impl(ctx, sat_dev, src_dev, width, height, linesize, nullptr,
                               nullptr);
}

extern "C" int f360_sat_encode_batch(f360_ctx *ctx, int count, uint32_t *const *sat_dev,
                                     const uint8_t *const *src_dev, int width, int height,
                                     int linesize) {
  F360_REQUIRE(count >= 1 && sat_dev && src_dev && width >= 1 && height >= 1 && linesize >= 1,
               "f360_sat_encode_batch: bad arguments");
  // Frames per launch ("sat.batch_mb", 180 MB of source).  The frames of a launch are read
  // twice, by the reducer and then by the table writer, and the second read comes out of the
  // 256 MiB Infinity Cache only while they fit beside the rest of the traffic: two 8K RGB0
  // frames (236 MB) per launch and the writer takes 85-87 us per frame instead of 75, whereas
  // four 8K frames from planes (176 MB) keep it at 74 and six (264 MB) do not (80).  Sixteen
  // 1080p frames (133 MB) are 2.7 times faster than one at a time.
  const size_t frame_bytes = (size_t)linesize * height;
  const int per_launch = (int)std::min<size_t>(
      std::max<size_t>(((size_t)std::max(ctx ? ctx->opt_batch_mb : 180, 1) << 20) / frame_bytes, 1),
      (size_t)f360_sat_encode_batch_max());
  F360_REQUIRE(ctx, "f360_sat_encode_batch: null context");
  const int prof = f360::take_profile_slot(ctx) ? 1 : 0;  // one slot for the whole call
  // enough frames to fill the device with strip owners: the read-once encoder (sat_walk_kernel)
  if (walk_wanted(ctx, count, width) && linesize / width == 4 && width % 4 == 0 &&
      linesize % 16 == 0 && (size_t)width * height * 3 < ((size_t)1 << 31)) {
    bool ok = true;
    for (int k = 0; k < count && ok; ++k)
      ok = sat_dev[k] && src_dev[k] && ((uintptr_t)src_dev[k] % 16) == 0 &&
           ((uintptr_t)sat_dev[k] % 16) == 0;
    if (ok) {
      F360_BIND_DEVICE(ctx);
      return sat_encode_walk(ctx, count, sat_dev, src_dev, nullptr, width, height, linesize,
                             prof != 0);
    }
  }
  for (int k = 0; k < count; k += per_launch) {
    const int n = std::min(count - k, per_launch);
    const int st = f360::sat_encode_impl(ctx, nullptr, nullptr, width, height, linesize, nullptr,
                                         nullptr, n, sat_dev + k, src_dev + k, prof);
    if (st != F360_OK) return st;
  }
  return F360_OK;
}

extern "C" int f360_sat_encode_yuv420p_batch(f360_ctx *ctx, int count, uint32_t *const *sat_dev,
                                             const uint8_t *const *y_dev,
                                             const uint8_t *const *u_dev,
                                             const uint8_t *const *v_dev, int y_linesize,
                                             int u_linesize, int v_linesize, int width,
                                             int height) {
  F360_REQUIRE(ctx && count >= 1 && sat_dev && y_dev && u_dev && v_dev && width >= 1 &&
                   height >= 1 && y_linesize >= 1,
               "f360_sat_encode_yuv420p_batch: bad arguments");
  // frames per launch: the same cache budget as f360_sat_encode_batch, on 1.5 bytes per pixel
  // (8K: four frames per launch -- reducer 31 -> 23 us and carry pass 7.4 -> 3 us per frame)
  const size_t frame_bytes = (size_t)y_linesize * height * 3 / 2;
  const int per_launch = (int)std::min<size_t>(
      std::max<size_t>(((size_t)std::max(ctx->opt_batch_mb, 1) << 20) / frame_bytes, 1),
      (size_t)kEncBatch);
  const int prof = f360::take_profile_slot(ctx) ? 1 : 0;  // one slot for the whole call
  std::vector<f360::YuvPlanes> planes((size_t)count);
  for (int k = 0; k < count; ++k)
    planes[(size_t)k] = f360::YuvPlanes{y_dev[k], u_dev[k], v_dev[k], y_linesize, u_linesize,
                                        v_linesize};
  // enough frames to fill the device: the read-once encoder, converting in registers as the
  // three-kernel one does (the same argument rules; anything else takes the old path, whose
  // checks then report what is wrong)
  if (walk_wanted(ctx, count, width) && width % 4 == 0 && height % 2 == 0 &&
      y_linesize >= width && u_linesize >= width / 2 && v_linesize >= width / 2 &&
      y_linesize % 4 == 0 && u_linesize % 2 == 0 && v_linesize % 2 == 0 &&
      (size_t)width * height * 3 < ((size_t)1 << 31)) {
    bool ok = true;
    for (int k = 0; k < count && ok; ++k)
      ok = sat_dev[k] && y_dev[k] && u_dev[k] && v_dev[k] && ((uintptr_t)y_dev[k] % 4) == 0 &&
           ((uintptr_t)u_dev[k] % 2) == 0 && ((uintptr_t)v_dev[k] % 2) == 0 &&
           ((uintptr_t)sat_dev[k] % 16) == 0;
    if (ok) {
      F360_BIND_DEVICE(ctx);
      return sat_encode_walk(ctx, count, sat_dev, nullptr, planes.data(), width, height, 0,
                             prof != 0);
    }
  }
  for (int k = 0; k < count; k += per_launch) {
    const int n = std::min(count - k, per_launch);
    const int st = f360::sat_encode_impl(ctx, nullptr, nullptr, width, height, 0, nullptr, nullptr,
                                         n, sat_dev + k, nullptr, prof, planes.data() + k);
    if (st != F360_OK) return st;
  }
  return F360_OK;
}

extern "C" int f360_sat_encode_yuv420p(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *y_dev,
                                       const uint8_t *u_dev, const uint8_t *v_dev,
                                       int y_linesize, int u_linesize, int v_linesize,
                                       int width, int height) {
  F360_REQUIRE(sat_dev, "f360_sat_encode_yuv420p: null buffer");
  const f360::YuvPlanes planes{y_dev, u_dev, v_dev, y_linesize, u_linesize, v_linesize};
  return f360::sat_encode_impl(ctx, sat_dev, nullptr, width, height, 0, nullptr, &planes);
}
