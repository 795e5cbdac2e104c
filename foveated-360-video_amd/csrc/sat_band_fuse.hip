// sat_band_fuse.hip -- encode + sample in one pass for calls too small for the strip walker: up
// to the 22 8K frames below "sat.walk_units" (8 per rank when BASELINE config 4's 64 frames are
// sharded over 8 GPUs).  Taken from four frames per call on ("fuse.band" 1); a call of one to
// three frames -- the per-frame loop of the reference's offline tool,
// run_satlogrectilinear.cc:926-938 -- is faster as the two calls and stays that unless
// "fuse.band" is 2 (profiles/round5_band_one_pass.txt).
//
// The three-kernel encoder's table writer (sat_three.hip: one wave per band x strip tile) holds,
// at every row of its tile, the table row of its strip in registers -- exactly what a strip owner
// of the read-once encoder holds.  So the same arithmetic applies (sat_fuse_dev.h): a tile keeps a
// snapshot of its table row at a box's upper edge `lo`, forms D = row - snapshot at the lower edge
// `hi`, and a reduced pixel whose two columns lie in the strip is (D[hi_x] - D[lo_x]) / area
// (src/sat_decoder_sample_rect_kernel.cl:206-217, all in uint32).  The table is written exactly
// as by sat_write_kernel; what disappears is the sampler's pass over the table (233 MB of reads
// per 8K frame for 146.6 MB algorithmic).
//
// What differs from the strip walker's one pass:
//   * Tiles are independent -- no hand-off chain runs through them -- and a launch holds seven
//     waves per CU at 8K, so a tile does its own gathering (no helper wave, no mailbox): the rows
//     of the other waves on its SIMD hide it.  D rows go through the 3 KiB of LDS the table
//     row's store re-staging has just released.
//   * A tile starts with the table row above its band as its snapshot (the prologue computes
//     that row anyway), so every box whose rows lie inside one band is emitted; a box that
//     crosses a band boundary -- at most one per boundary, none in the fovea -- is left to the
//     fix-up kernel, which samples it from the finished table (band_fuse_plan_kernel).
//   * Which reduced pixels a strip owns is worked out once per frame by the plan kernel, not per
//     tile (a strip has 60 tiles at 8K) -- and a box one column wide (three quarters of an 8K
//     frame's reduced columns, all of the fovea) needs no gather at all: the lane that holds its
//     column has both of its columns in registers, or gets the left one from its neighbour.
//     Only the wider boxes go through a D row in LDS.
// Boxes that straddle two strips and the rows the plan could not mark go through the same side
// rows and the same fix-up kernel as the strip walker's (sat_fuse_kernels.h).
#include "sat_fuse_kernels.h"
#include "sat_walk.h"

using namespace f360::sat;

namespace {

#ifndef F360_BAND_PX_BITS
#define F360_BAND_PX_BITS " nt"  // cache bits of the reduced-pixel stores (A/B builds)
#endif
__device__ __forceinline__ void band_store_rgb(uint8_t *row, uint32_t off, uint32_t rg, uint32_t b,
                                               bool whole = false) {
  if (whole) {  // (timing experiment, debug.ablate 16384: ONE four-byte store, clobbers byte 3)
    asm volatile("global_store_dword %0, %1, %2" F360_BAND_PX_BITS ::"v"(off), "v"(rg | (b << 16)), "s"(row) : "memory");
    return;
  }
  asm volatile("global_store_short %0, %1, %3" F360_BAND_PX_BITS "\n\t"
               "global_store_byte %0, %2, %3 offset:2" F360_BAND_PX_BITS ::"v"(off),
               "v"(rg), "v"(b), "s"(row)
               : "memory");
}
// a source pixel as it is: R, G from the low half, B from byte 2
__device__ __forceinline__ void band_store_px(uint8_t *row, uint32_t off, uint32_t px,
                                              bool whole = false) {
  if (whole) {
    asm volatile("global_store_dword %0, %1, %2" F360_BAND_PX_BITS ::"v"(off), "v"(px), "s"(row) : "memory");
    return;
  }
  asm volatile("global_store_short %0, %1, %2" F360_BAND_PX_BITS "\n\t"
               "global_store_byte_d16_hi %0, %1, %2 offset:2" F360_BAND_PX_BITS ::"v"(off),
               "v"(px), "s"(row)
               : "memory");
}

// ---- the plan of a frame: row marks, straddling pixels, per-strip pixel maps -----------------
// One workgroup per frame; the decoder's two grid axes are staged in LDS first (every box rule
// below is a chain of grid look-ups, and from global memory each link is a round trip), the
// row plan is built in LDS and written out once.
//
// Rows (walk_fuse_plan_kernel's rule, sat_fuse_kernels.h, plus the band condition): reduced row j
// is the box of table rows (lo, hi]; a tile holds ONE snapshot and starts with the table row
// above its band, so row j is emitted by the writer iff no processed neighbour snapshots
// strictly inside (lo, hi), no earlier reduced row emits at the same table row, and all of
// lo + 1 .. hi lie in one band.  Such rows get EMIT | j | height at hi and SNAP at lo; the others
// are listed for the fix-up (one per band boundary at most, none where boxes are one row high).
//
// Columns, per strip of 256: a box one column wide that ends at column x (and begins in the same
// strip) is recorded AT x -- the lane that holds column x has everything such a pixel needs in
// registers -- wider boxes inside the strip are listed {hi : 8 | lo : 8 | reduced column : 16},
// boxes that straddle two strips go to the side-row list like the strip walker's.
constexpr uint32_t kNoPixel = 0xffffffffu;

constexpr int kPlanPerThread = 20;  // grid entries a thread stages at once: axes up to 5120

__global__ __launch_bounds__(256) void band_fuse_plan_kernel(const WalkFuse wf, int src_w,
                                                             int src_h) {
  extern __shared__ __attribute__((aligned(16))) uint32_t plds[];
  const int f = blockIdx.x, t = threadIdx.x;
  const int nstrips = (src_w + kStripPx - 1) / kStripPx;
  // LDS: the row plan | per column of every strip, the one-column box that ends there | the two
  // grids as int16 (each rounded up to whole dwords)
  uint32_t *lplan = plds;
  uint32_t *lunit = lplan + wf.plan_stride;
  int16_t *sgx = reinterpret_cast<int16_t *>(lunit + nstrips * 256);
  int16_t *sgy = sgx + ((wf.out_w + 2) & ~1);
  __shared__ uint32_t n_wide[kFixCols / 4], widest[kFixCols / 4];
  __shared__ int count, nleft;
  {
    // every grid load of a thread issued before the first is used: staged one after the other
    // the two axes are 26 dependent round trips (a third of this kernel's 26 us when it was
    // written that way)
    int16_t vx[kPlanPerThread], vy[kPlanPerThread];
#pragma unroll
    for (int k = 0; k < kPlanPerThread; ++k) {
      vx[k] = wf.gx[min(t + 256 * k, wf.out_w)];
      vy[k] = wf.gy[min(t + 256 * k, wf.out_h)];
    }
#pragma unroll
    for (int k = 0; k < kPlanPerThread; ++k) {
      if (t + 256 * k <= wf.out_w) sgx[t + 256 * k] = vx[k];
      if (t + 256 * k <= wf.out_h) sgy[t + 256 * k] = vy[k];
    }
    for (int i = t + 256 * kPlanPerThread; i <= wf.out_w; i += 256) sgx[i] = wf.gx[i];
    for (int j = t + 256 * kPlanPerThread; j <= wf.out_h; j += 256) sgy[j] = wf.gy[j];
  }
  for (int y = t; y < wf.plan_stride; y += 256) lplan[y] = 0;
  for (int k = t; k < nstrips * 256; k += 256) lunit[k] = kNoPixel;
  for (int s = t; s < nstrips; s += 256) {
    n_wide[s] = 0;
    widest[s] = 1;
  }
  if (t == 0) count = nleft = 0;
  __syncthreads();

  uint32_t *ent = wf.ent + (size_t)f * nstrips * kBandEntStride;
  uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  const int cyp = wf.cyp[f], cxp = wf.cxp[f];
  for (int j = t; j < wf.out_h; j += 256) {
    const f360::AxisBox b = f360::sample_axis(cyp, sgy[j + 1], sgy[j], src_h, false);
    if (!b.ok) continue;
    bool fused = (b.lo + 1) / wf.band_rows == b.hi / wf.band_rows;
    for (int d = 1; d <= 3; ++d) {
      if (j + d < wf.out_h) {
        const f360::AxisBox n = f360::sample_axis(cyp, sgy[j + d + 1], sgy[j + d], src_h, false);
        if (n.ok && n.lo > b.lo && n.lo < b.hi) fused = false;
      }
      if (j - d >= 0) {
        const f360::AxisBox p = f360::sample_axis(cyp, sgy[j - d + 1], sgy[j - d], src_h, false);
        if (p.ok && p.hi == b.hi) fused = false;
      }
    }
    if (fused) {
      atomicOr(&lplan[b.hi], kFuseEmit | (uint32_t)j | ((uint32_t)(b.hi - b.lo) << 16));
      atomicOr(&lplan[b.lo], kFuseSnap);
    } else {
      const int k = atomicAdd(&nleft, 1);
      if (k < wf.lrows_max) sp[kSpixLrows + 1 + k] = (uint32_t)j;
    }
  }
  // columns: reduced columns in chunks of 256, ascending, a barrier between chunks -- a strip's
  // list of wide boxes then ascends from chunk to chunk whatever order one chunk's atomics
  // come in, so a round of 64 of them stores to few lines.  Nothing in the loop waits for
  // global memory: the counters and the per-column map live in LDS, the list entries are
  // stores nobody waits for.
  for (int i0 = 0; i0 < wf.out_w; i0 += 256) {
    const int i = i0 + t;
    if (i < wf.out_w) {
      const f360::AxisBox bx = f360::sample_axis(cxp, sgx[i + 1], sgx[i], src_w, true);
      if (bx.ok) {
        const int s = bx.hi >> 8;
        if ((bx.lo >> 8) != s) {  // straddles two strips
          const int k = atomicAdd(&count, 1);
          if (k < kFixCols) {
            sp[1 + 3 * k] = (uint32_t)i;
            sp[2 + 3 * k] = (uint32_t)bx.hi;
            sp[3 + 3 * k] = (uint32_t)bx.lo;
          }
        } else if (bx.hi - bx.lo == 1 && atomicCAS(&lunit[bx.hi], kNoPixel, (uint32_t)i * 4u) == kNoPixel) {
          // (recorded at its column; a second one-column box ending at the same column -- two
          // wrap classes meeting -- goes to the list below)
        } else {
          const uint32_t k = atomicAdd(&n_wide[s], 1u);
          atomicMax(&widest[s], (uint32_t)(bx.hi - bx.lo));
          if (k < (uint32_t)kFuseEntries)  // (a strip cannot hold more: <= 256 per wrap class)
            ent[(size_t)s * kBandEntStride + kBandEntWide + k] =
                (uint32_t)(bx.hi & 255) | ((uint32_t)(bx.lo & 255) << 8) | ((uint32_t)i << 16);
        }
      }
    }
    // (a bare barrier: what must be ordered between chunks are the LDS atomics, whose results
    // the threads have waited for; __syncthreads() would also wait for the list stores above to
    // be acknowledged -- 17 memory round trips in a row, 20 of this kernel's 24 us)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __syncthreads();
  uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  for (int y = t; y < wf.plan_stride; y += 256) plan[y] = lplan[y];
  for (int k = t; k < nstrips * 256; k += 256)
    ent[(size_t)(k >> 8) * kBandEntStride + kBandEntUnit + (k & 255)] = lunit[k];
  for (int s = t; s < nstrips; s += 256) {
    uint32_t *e = ent + (size_t)s * kBandEntStride;
    e[0] = min(n_wide[s], (uint32_t)kFuseEntries);
    e[1] = widest[s];
  }
  if (t == 0) {
    sp[0] = (uint32_t)count;
    sp[kSpixLrows] = (uint32_t)nleft;
  }
}

// The wide boxes of one emitted row: one per lane and round, gathered from the wave's D row in
// LDS (`dlds`, 12 bytes per column, written by the caller just before).  Round 0's entries live
// in registers (`e0`: {hi * 12 | lo * 12 << 12 | width << 24, reduced column * 4}); a strip
// with more than 64 wide boxes reads the further rounds' entries from its list in LDS (`elds`,
// 8 bytes per box, same packing).  Every asm statement ends with its own wait: nothing is in
// flight between two of them, so the compiler may move or copy their results as it likes.
__device__ __forceinline__ void band_emit_wide(uint32_t pr, uint8_t *orow, uint32_t dlds,
                                               uint32_t elds, int lane, int n_ent, u32x2 e0,
                                               u32x2 es1, u32x2 h01, uint32_t h2, u32x2 l01,
                                               uint32_t l2, uint32_t max_dxw, bool no_px) {
  const uint32_t dy = (pr >> 16) & 0x3ffu;
  const bool quick = dy * max_dxw <= 2048u;  // the float quotient is exact (tests/test_fuse_div.py)
  const float inv_dy = __builtin_amdgcn_rcpf((float)dy);
  u32x2 es = e0;
  for (int b0 = 0; b0 < n_ent; b0 += 64) {
    const uint32_t eoff = es.x, estore = es.y;
    const bool valid = b0 + lane < n_ent;
    if (b0 == 0) {
      es = es1;  // (round 0's gathers and round 1's entries came with the caller's exchange)
    } else if (b0 + 64 < n_ent) {  // (wave-uniform) the next round's entry along with the gathers
      asm volatile(
          "ds_read_b64 %0, %5\n\t"
          "ds_read2_b32 %1, %6 offset1:1\n\t"
          "ds_read_b32 %2, %6 offset:8\n\t"
          "ds_read2_b32 %3, %7 offset1:1\n\t"
          "ds_read_b32 %4, %7 offset:8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(es), "=&v"(h01), "=&v"(h2), "=&v"(l01), "=&v"(l2)
          : "v"(elds + (uint32_t)(b0 + 64 + lane) * 8u), "v"(dlds + (eoff & 0xfffu)),
            "v"(dlds + ((eoff >> 12) & 0xfffu))
          : "memory");
    } else {
      asm volatile(
          "ds_read2_b32 %0, %4 offset1:1\n\t"
          "ds_read_b32 %1, %4 offset:8\n\t"
          "ds_read2_b32 %2, %5 offset1:1\n\t"
          "ds_read_b32 %3, %5 offset:8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(h01), "=&v"(h2), "=&v"(l01), "=&v"(l2)
          : "v"(dlds + (eoff & 0xfffu)), "v"(dlds + ((eoff >> 12) & 0xfffu))
          : "memory");
    }
    const uint3 n = make_uint3(h01.x - l01.x, h01.y - l01.y, h2 - l2);
    const uint32_t dxw = eoff >> 24;
    if (quick) {
      const float inv = __builtin_amdgcn_rcpf((float)dxw) * inv_dy;
      const uint32_t qx = (uint32_t)__builtin_fmaf((float)n.x, inv, 0x1p-12f);
      const uint32_t qy = (uint32_t)__builtin_fmaf((float)n.y, inv, 0x1p-12f);
      const uint32_t qz = (uint32_t)__builtin_fmaf((float)n.z, inv, 0x1p-12f);
      if (valid && !no_px) band_store_rgb(orow, estore, qx | (qy << 8), qz);
    } else {
      const uint3 q = fuse_div3(n, dxw * dy);
      if (valid && !no_px) band_store_rgb(orow, estore, (q.x & 0xffu) | ((q.y & 0xffu) << 8), q.z);
    }
  }
}

// The table writer of sat_three.hip (RGB0 frames, LDS-staged non-temporal stores) with the
// reduced pixels of its tile emitted on the way.  `frame0`: index of the launch's first frame in
// the call's per-frame arrays of `wf`.
template <int SRC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sat_write_fuse_kernel(
    const EncodeArgs a, const EncodeBatch eb, const WalkFuse wf, int frame0) {
  static_assert(SRC != kSrcBytes, "the band writer's one pass takes aligned RGB0 frames or planes");
  // per wave: 3 KiB store staging / D row, 1 KiB to turn a row of reduced pixels around (below),
  // then the strip's list of wide boxes past the first round (8 bytes each; 6 KiB, hardly ever
  // used)
  constexpr int kWaveDwords = 4 * kStripPx + 2 * kFuseEntries;
  __shared__ __attribute__((aligned(16))) uint32_t stage[kWavesPerBlock * kWaveDwords];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kWavesPerBlock + wave);
  if (tile >= a.nstrips * a.nbands) return;  // 1-D grid over the tiles, see sat_reduce_kernel
  F360_ENCODE_FRAME(fr, a, eb)
  const int band = tile / a.nstrips;
  const int strip = tile - band * a.nstrips;
  const int x0 = strip * kStripPx + lane * kLanePx;
  const int sb = band / a.sb_bands;
  const int f = frame0 + (a.nbatch != 0 ? (int)blockIdx.y : 0);
  const uint32_t mine = (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * kWaveDwords * 4;
  const uint32_t tlds = mine + 3 * kStripPx * 4;
  const uint32_t elds = tlds + kStripPx * 4;

  const int y_end = min((band + 1) * a.band_rows, a.height);
  const uint32_t *rc = a.rowcarry + fr.ws + (size_t)strip * a.height * 3;
  const uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  // A batch = kRowUnroll rows of pixels, their row carries (lanes 0..23) and their plan words
  // (lanes 32..39) in one more load, broadcast with v_readlane; no branch around any load
  auto load_batch = [&](RowBatch<SRC> &raw, uint32_t &carry, int y) {
    reduce_load_batch<SRC>(a, fr, raw, y, x0, y_end - 1);
    const uint32_t *cp = rc + min(y * 3 + min(lane, 3 * kRowUnroll - 1), a.height * 3 - 1);
    if (lane >= 32) cp = plan + min(y + min(lane - 32, kRowUnroll - 1), wf.plan_stride - 1);
    carry = *cp;
  };
  RowBatch<SRC> buf_a, buf_b;
  uint32_t carry_a, carry_b;
  const int y_begin = band * a.band_rows;
  load_batch(buf_a, carry_a, y_begin);
  load_batch(buf_b, carry_b, y_begin + kRowUnroll);

  // --- the strip's reduced pixels (band_fuse_plan_kernel) and its straddling columns --------
  const uint32_t *ent = wf.ent + ((size_t)f * a.nstrips + strip) * kBandEntStride;
  const int n_ent = (int)ent[0];
  const uint32_t max_dxw = ent[1];
  // Where the one-column box that ends at a column goes (byte offset in the reduced row, ~0 if
  // none).  A lane COMPUTES the pixels of its own four columns (4 * lane + k) but STORES those of
  // columns lane + 64 k: turned around through LDS, a store instruction writes 64 neighbouring
  // pixels (two lines) instead of every fourth pixel of 256 (all eight lines of the row segment,
  // each of which four instructions then fill piecemeal) -- the table rows' re-staging over again.
  uint32_t ux[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ux[k] = ent[kBandEntUnit + lane + 64 * k];
  const bool any_unit = __any((ux[0] & ux[1] & ux[2] & ux[3]) != kNoPixel);
  u32x2 e0;
  {
    const uint32_t en = lane < n_ent ? ent[kBandEntWide + lane] : 0x00000100u;  // (hi 0, lo 1: harmless)
    const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
    e0.x = (hi * 12u) | ((lo * 12u) << 12) | ((lane < n_ent ? hi - lo : 1u) << 24);
    e0.y = (en >> 16) * 4u;
  }
  for (int b0 = 64; b0 < n_ent; b0 += 64) {  // (more than one round of wide boxes: rare)
    const int e = b0 + lane;
    const uint32_t en = e < n_ent ? ent[kBandEntWide + e] : 0x00000100u;
    const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
    lds_write_b64(elds + (uint32_t)e * 8u,
                  (hi * 12u) | ((lo * 12u) << 12) | ((e < n_ent ? hi - lo : 1u) << 24),
                  (en >> 16) * 4u);
  }
  const uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  int npix = (int)sp[0];
  if (npix > wf.pmax) npix = 0;  // (more than the side rows hold: the fix-up takes every row)
  int xcol[3], xslot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = lane + 64 * k;
    xcol[k] = 0;
    xslot[k] = -1;
    const int hi = (int)sp[2 + 3 * min(q, kFixCols - 1)], lo = (int)sp[3 + 3 * min(q, kFixCols - 1)];
    if (q < npix) {
      if ((hi >> 8) == strip) {
        xcol[k] = hi & 255;
        xslot[k] = 2 * q;
      } else if ((lo >> 8) == strip) {
        xcol[k] = lo & 255;
        xslot[k] = 2 * q + 1;
      }
    }
  }
  const bool exports = __any(xslot[0] >= 0) || __any(xslot[1] >= 0) || __any(xslot[2] >= 0);
  const bool needs_d = n_ent > 0 || exports;  // (else the D row never goes through LDS)
  const bool no_px = a.ablate & 8192;         // (timing experiment: everything but the pixel stores)
  uint8_t *dst = wf.dst[f];
  uint32_t *side = wf.side + (size_t)f * wf.side_stride;

  // --- table row just above the band, for this lane's 4 pixels -------------
  uint32_t acc[12];
  {
    uint32_t t0[12], t1[12];
    load12(a.sbprefix + fr.ws + (size_t)sb * a.wp3 + (size_t)x0 * 3, t1);
    if (a.sb_bands == 1) {
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t1[e];
    } else {
      load12(a.lp + fr.ws + (size_t)band * a.wp3 + (size_t)x0 * 3, t0);
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t0[e] + t1[e];
    }
  }
  uint32_t corner[3] = {0, 0, 0};
  for (int b = lane; b < band; b += 64) {
    const uint32_t *tp = a.tprefix + fr.ws + ((size_t)strip * a.nbands + b) * 3;
    corner[0] += tp[0];
    corner[1] += tp[1];
    corner[2] += tp[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    corner[c] = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(corner[c]), 63);
    acc[3 + c] += acc[c];
    acc[6 + c] += acc[3 + c];
    acc[9 + c] += acc[6 + c];
    const uint32_t excl = wave_scan_incl(acc[9 + c]) - acc[9 + c] + corner[c];
    acc[c] += excl;
    acc[3 + c] += excl;
    acc[6 + c] += excl;
    acc[9 + c] += excl;
  }
  // the snapshot a box with its upper edge on the row above the band needs
  uint32_t snap[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) snap[e] = acc[e];

  const int row_dwords = a.width * 3;
  const int base = strip * kStripPx * 3;
  auto write_batch = [&](const RowBatch<SRC> &raw, uint32_t carry, int y) {
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) {
      if (y + r >= y_end) break;
      uint32_t c[12], px[4];
      batch_pixels<SRC>(a, raw, r, px);
      unpack_px4(make_uint4(px[0], px[1], px[2], px[3]), c);
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
      uint32_t *row = fr.sat + (size_t)(y + r) * a.width * 3;
      lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
      lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
      lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
      u32x4 v[3];
      lds_read3_b128(mine + lane * 16, v[0], v[1], v[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int off = q * 256 + lane * 4;
        if (base + off < row_dwords)  // width % 4 == 0 -> whole 16 B in range
          global_store_b128_uncounted_nt(row + base + off, v[q]);
      }
      const uint32_t pr = (uint32_t)__builtin_amdgcn_readlane((int)carry, 32 + r);  // plan[y + r]
      // (debug.ablate, timing experiments: 1024 no emit at all, 2048 no one-column boxes,
      // 4096 no wide boxes / side rows, 8192 no pixel stores)
      if ((pr & kFuseEmit) && !(a.ablate & 1024)) {
        const uint32_t dy = (pr >> 16) & 0x3ffu;
        uint8_t *orow = dst + (size_t)(pr & 0xffffu) * wf.dst_linesize;
        // (1) boxes one column wide, from registers: a lane has both of their columns, or takes
        // the left one from its neighbour (the first column of a strip never ends such a box:
        // its left column lies in the other strip)
        const bool do_unit = any_unit && !(a.ablate & 2048);
        const bool do_wide = needs_d && !(a.ablate & 4096);
        uint32_t mypx[4] = {0, 0, 0, 0};  // columns 4 * lane + k, packed R | G << 8 | B << 16
        if (do_unit) {
          if (dy == 1u) {  // the fovea: a reduced pixel IS a source pixel
#pragma unroll
            for (int k = 0; k < 4; ++k) mypx[k] = px[k];
          } else {
            // D = row - snapshot; box = D[x] - D[x - 1]; dy <= band height: the float quotient
            // is exact (tests/test_fuse_div.py)
            const float inv_dy = __builtin_amdgcn_rcpf((float)dy);
            uint32_t left[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {  // D of the lane to the left's fourth column
              left[ch] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(acc[9 + ch] - snap[9 + ch]),
                                                               0x138 /* wave_shr:1 */, 0xf, 0xf, false);
              // (kept a plain v_mov_dpp: hipcc 7.2 folds the move into the subtraction below as
              // v_subrev_u32_dpp, and that form returned wrong differences on gfx950 -- found by
              // the parity tests, like the byte-packing instruction check_isa.py bans)
              asm volatile("" : "+v"(left[ch]));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              uint32_t q[3];
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) {
                const uint32_t d = acc[3 * k + ch] - snap[3 * k + ch];
                const uint32_t n = d - left[ch];
                left[ch] = d;
                q[ch] = (uint32_t)__builtin_fmaf((float)n, inv_dy, 0x1p-12f);
              }
              mypx[k] = q[0] | (q[1] << 8) | (q[2] << 16);
            }
          }
        }
        // (2) wider boxes and the columns of boxes that straddle two strips need the D row: into
        // the staging slice (its reads have returned)
        if (do_wide) {
          lds_write_b128(mine + lane * 48, u32x4{acc[0] - snap[0], acc[1] - snap[1],
                                                  acc[2] - snap[2], acc[3] - snap[3]});
          lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4] - snap[4], acc[5] - snap[5],
                                                       acc[6] - snap[6], acc[7] - snap[7]});
          lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8] - snap[8], acc[9] - snap[9],
                                                       acc[10] - snap[10], acc[11] - snap[11]});
        }
        // ONE LDS round trip for everything the row needs back (they were five in a row: the
        // turned-around row, the wide boxes' first round, three straddling columns): the row of
        // one-column boxes goes in and comes back as columns lane + 64 k, the first 64 wide boxes
        // gather their two columns, the second round's entries and the straddling columns come
        // along.  Paths that are off read slots nobody looks at.  One statement, one wait.
        u32x2 t01, t23, h01, l01, es1, x01[3];
        uint32_t h2, l2, x2[3];
        asm volatile(
            "ds_write_b128 %13, %14\n\t"
            "ds_read2st64_b32 %0, %15 offset1:1\n\t"
            "ds_read2st64_b32 %1, %15 offset0:2 offset1:3\n\t"
            "ds_read2_b32 %2, %16 offset1:1\n\t"
            "ds_read_b32 %3, %16 offset:8\n\t"
            "ds_read2_b32 %4, %17 offset1:1\n\t"
            "ds_read_b32 %5, %17 offset:8\n\t"
            "ds_read_b64 %6, %18\n\t"
            "ds_read2_b32 %7, %19 offset1:1\n\t"
            "ds_read_b32 %8, %19 offset:8\n\t"
            "ds_read2_b32 %9, %20 offset1:1\n\t"
            "ds_read_b32 %10, %20 offset:8\n\t"
            "ds_read2_b32 %11, %21 offset1:1\n\t"
            "ds_read_b32 %12, %21 offset:8\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(t01), "=&v"(t23), "=&v"(h01), "=&v"(h2), "=&v"(l01), "=&v"(l2), "=&v"(es1),
              "=&v"(x01[0]), "=&v"(x2[0]), "=&v"(x01[1]), "=&v"(x2[1]), "=&v"(x01[2]), "=&v"(x2[2])
            : "v"(tlds + lane * 16), "v"(u32x4{mypx[0], mypx[1], mypx[2], mypx[3]}),
              "v"(tlds + lane * 4), "v"(mine + (e0.x & 0xfffu)), "v"(mine + ((e0.x >> 12) & 0xfffu)),
              "v"(elds + (uint32_t)(64 + lane) * 8u), "v"(mine + (uint32_t)xcol[0] * 12u),
              "v"(mine + (uint32_t)xcol[1] * 12u), "v"(mine + (uint32_t)xcol[2] * 12u)
            : "memory");
        if (do_unit) {
          const uint32_t out[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ux[k] != kNoPixel && !no_px) band_store_px(orow, ux[k], out[k], a.ablate & 16384);
        }
        if (do_wide) {
          if (n_ent > 0)
            band_emit_wide(pr, orow, mine, elds, lane, n_ent, e0, es1, h01, h2, l01, l2, max_dxw, no_px);
          if (exports) {
            uint32_t *srow = side + (size_t)(pr & 0xffffu) * npix * 6;
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (xslot[k] >= 0)
                asm volatile("global_store_dwordx3 %0, %1, %2" ::"v"((uint32_t)xslot[k] * 12u),
                             "v"(u32x3v{x01[k].x, x01[k].y, x2[k]}), "s"(srow)
                             : "memory");
          }
        }
      }
      if (pr & kFuseSnap) {
#pragma unroll
        for (int e = 0; e < 12; ++e) snap[e] = acc[e];
      }
    }
  };
  for (int y = y_begin; y < y_end; y += 2 * kRowUnroll) {
    write_batch(buf_a, carry_a, y);
    load_batch(buf_a, carry_a, y + 2 * kRowUnroll);
    if (y + kRowUnroll < y_end) write_batch(buf_b, carry_b, y + kRowUnroll);
    load_batch(buf_b, carry_b, y + 3 * kRowUnroll);
  }
}

}  // namespace

// The table writer's launch of a three-kernel encode (sat_three.hip: sat_encode_impl), one-pass
// form.  `grid` / frames as for sat_write_kernel.
void f360::sat::launch_write_fuse(f360_ctx *ctx, hipStream_t stream, const EncodeArgs &a,
                                  const EncodeBatch &eb, dim3 grid, const f360::SatBandFuse &bf,
                                  int src_kind) {
  (void)ctx;
  const dim3 block(64 * kWavesPerBlock);
  if (src_kind == kSrcYuvSwsX86)
    hipLaunchKernelGGL((sat_write_fuse_kernel<kSrcYuvSwsX86>), grid, block, 0, stream, a, eb, bf.wf, bf.frame0);
  else if (src_kind == kSrcYuvSwsC)
    hipLaunchKernelGGL((sat_write_fuse_kernel<kSrcYuvSwsC>), grid, block, 0, stream, a, eb, bf.wf, bf.frame0);
  else
    hipLaunchKernelGGL((sat_write_fuse_kernel<kSrcRgb0>), grid, block, 0, stream, a, eb, bf.wf, bf.frame0);
}

// Whether f360_satdec_encode_sample_frames can take the band writer's one pass for this call
// ("fuse.band"; the caller has already found the strip walker's form not applicable).
static size_t band_plan_lds_bytes(int height, int out_w, int out_h) {
  const int plan_stride = ((height + kRowUnroll - 1) / kRowUnroll) * kRowUnroll;
  return (size_t)plan_stride * 4 + (size_t)((out_w + 2) & ~1) * 2 + (size_t)((out_h + 2) & ~1) * 2;
}
static size_t band_plan_lds_bytes(int width, int height, int out_w, int out_h) {
  return band_plan_lds_bytes(height, out_w, out_h) + (size_t)((width + kStripPx - 1) / kStripPx) * 1024;
}
bool f360::sat_encode_sample_band_applies(const f360_ctx *ctx, int count, int width, int height,
                                          int linesize, int out_w, int out_h, int dst_linesize,
                                          const f360::YuvPlanes *yuv) {
  // the source side: f360_sat_encode_batch's / f360_sat_encode_yuv420p_batch's own fast-path rules
  const bool source_ok =
      yuv ? height % 2 == 0 && yuv->y_linesize >= width && yuv->u_linesize >= width / 2 &&
                yuv->v_linesize >= width / 2 && yuv->y_linesize % 4 == 0 &&
                yuv->u_linesize % 2 == 0 && yuv->v_linesize % 2 == 0
          : linesize / width == 4 && linesize % 16 == 0;
  return (ctx->opt_fuse_band == 2 || (ctx->opt_fuse_band == 1 && count >= 4)) && source_ok &&
         width % 4 == 0 && width <= f360::kMaxDim &&
         (size_t)width * height * 3 < ((size_t)1 << 31) && out_w < 65536 && out_h < 65536 &&
         (width + kStripPx - 1) / kStripPx <= kFixCols / 4 && dst_linesize % 4 == 0 &&
         dst_linesize >= 4 * out_w &&
         band_plan_lds_bytes(width, height, out_w, out_h) <= 62 * 1024;  // the plan kernel's LDS
}

// f360_satdec_encode_sample_frames on the three-kernel encoder: one plan launch and one fix-up
// launch per (up to kWalkFrames) frames, between them the encoder's launches of "sat.batch_mb"
// each -- reducer, carry pass, and the table writer in its one-pass form.
int f360::sat_encode_sample_band(f360_ctx *ctx, int count, uint32_t *const *sats,
                                 const uint8_t *const *srcs, const f360::YuvPlanes *yuvs, int width,
                                 int height, int linesize, const f360::SatFuse &fuse, bool prof) {
  f360::SatEncodePlan &p = ctx->enc;
  const int nstrips = (width + kStripPx - 1) / kStripPx;
  const int plan_stride = ((height + kRowUnroll - 1) / kRowUnroll) * kRowUnroll;
  const int pmax = (ctx->opt_fuse_force & 1) ? 1 : std::max(1, std::min(3 * (nstrips - 1), kFixCols));
  const size_t side_stride = ((size_t)fuse.out_h * pmax * 6 + 3) & ~(size_t)3;  // dwords per frame
  const size_t frame_bytes = yuvs ? (size_t)yuvs[0].y_linesize * height * 3 / 2 : (size_t)linesize * height;
  const int per_launch = (int)std::min<size_t>(
      std::max<size_t>(((size_t)std::max(ctx->opt_batch_mb, 1) << 20) / frame_bytes, 1),
      (size_t)kEncBatch);
  // (the band height is the encoder plan's: make sure it exists before the plan kernel runs)
  int st = f360::sat_encode_reserve(ctx, width, height, std::min(per_launch, count), yuvs != nullptr);
  if (st != F360_OK) return st;
  const int band_rows = p.band_rows, nbands = (height + band_rows - 1) / band_rows;
  // reduced rows the plan cannot mark: one per band boundary at most, plus the clamped edge rows
  const int lrows_max = (ctx->opt_fuse_force & 2) ? 0 : std::min(kFixLrowsBand, nbands + 4);
  for (int c0 = 0; c0 < count; c0 += kWalkFrames) {
    const int n = std::min(count - c0, kWalkFrames);
    // one buffer: pixel maps | row plans | straddling pixels | side rows (16-byte aligned parts)
    const size_t ent_words = (size_t)n * nstrips * kBandEntStride;
    const size_t words = ent_words + (size_t)n * plan_stride + (size_t)n * kSpixWords +
                         (size_t)n * side_stride;
    if (words * 4 > p.walk_plan.bytes) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      F360_HIP_TRY(hipStreamIsCapturing(ctx->stream, &cap));
      F360_REQUIRE(cap == hipStreamCaptureStatusNone,
                   "f360_satdec_encode_sample_frames: the plan buffers (%zu bytes) must be "
                   "allocated but the stream is being captured; run the same call once before "
                   "the capture", words * 4);
      if (p.walk_plan.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
      st = p.walk_plan.reserve(words * 4);
      if (st != F360_OK) return st;
    }
    f360::SatBandFuse bf;
    WalkFuse &wf = bf.wf;
    WalkBatch wb;
    walk_fill_batch(wb, c0, n, sats, srcs, yuvs);
    for (int k = 0; k < kWalkFrames; ++k) {
      const int q = c0 + (k < n ? k : 0);
      wf.dst[k] = fuse.dsts[q];
      wf.cxp[k] = (int)(fuse.centers_xy[2 * q] * (float)width);  // sat_decoder_sample_rect_kernel.cl:176-179
      wf.cyp[k] = (int)(fuse.centers_xy[2 * q + 1] * (float)height);
    }
    wf.gx = fuse.gx;
    wf.gy = fuse.gy;
    wf.ent = p.walk_plan.as<uint32_t>();
    wf.rowplan = wf.ent + ent_words;
    wf.plan_stride = plan_stride;
    wf.out_w = fuse.out_w;
    wf.out_h = fuse.out_h;
    wf.dst_linesize = fuse.dst_linesize;
    wf.spix = wf.rowplan + (size_t)n * plan_stride;
    wf.side = wf.spix + (size_t)n * kSpixWords;
    wf.pmax = pmax;
    wf.side_stride = side_stride;
    wf.lrows_max = lrows_max;
    wf.force_tail = 0;
    wf.band_rows = band_rows;
    {
      f360::KernelSpan span(ctx, f360::kWalkFusePlan, prof, n);
      hipLaunchKernelGGL(band_fuse_plan_kernel, dim3(n), dim3(256),
                         band_plan_lds_bytes(width, height, fuse.out_w, fuse.out_h), ctx->stream, wf,
                         width, height);
    }
    // (launch groups alternate between the context's stream and its side stream, see
    // sat_pipelined_groups: forked after the plan kernel, joined before the fix-up)
    st = f360::sat_pipelined_groups(
        ctx, width, height, yuvs != nullptr, (n + per_launch - 1) / per_launch, std::min(per_launch, n), true,
        [&](int g, const f360::SatLaunch &where) {
          const int k0 = g * per_launch, m = std::min(n - k0, per_launch);
          bf.frame0 = k0;
          const int e = f360::sat_encode_impl(ctx, nullptr, nullptr, width, height, linesize, nullptr,
                                              nullptr, m, sats + c0 + k0, srcs ? srcs + c0 + k0 : nullptr,
                                              prof ? 1 : 0, yuvs ? yuvs + c0 + k0 : nullptr, &bf, &where);
          if (e == F360_OK && ctx->enc.band_rows != band_rows) {  // (cannot happen: same geometry)
            f360::set_error("f360_satdec_encode_sample_frames: the encoder plan changed under the call");
            return (int)F360_ERR_INVALID_ARG;
          }
          return e;
        });
    if (st != F360_OK) return st;
    {
      f360::KernelSpan span(ctx, f360::kWalkFuseFix, prof, n);
      hipLaunchKernelGGL(walk_fuse_fix_kernel<0>,
                         dim3((wf.out_h * pmax + 255) / 256 +
                                  std::max(lrows_max, 1) * ((wf.out_w + 255) / 256), n),
                         dim3(256), 0, ctx->stream, wb, wf, width, height, linesize, -1,
                         f360::YuvPlanes{nullptr, nullptr, nullptr, 0, 0, 0}, f360::YuvConsts{});
      // (tables are always written on this path, so the fix-up samples them whatever the source)
    }
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
